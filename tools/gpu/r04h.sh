#!/bin/bash
# round 4: full GPU suite at HEAD; the 60-iteration oracle digests of the benchmarked job in 5 more processes; batch timings
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
( time timeout 600 python -m pytest tests -q -m gpu -x --durations=6 ) 2>&1 | tail -20 > $O/tests_gpu.log; cat $O/tests_gpu.log
for i in 1 2 3 4 5; do
  timeout 200 python -c "
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'spark-data-repair-plugin_amd')
import tests.test_gpu_bench_shapes as t
t0 = time.time(); t.test_bench_job_60_iterations_with_six_targets_in_flight_match_the_oracle_digests(); print('process $i: 60 iterations of c10 (K=64, five other targets in flight) and c0: every digest equals the oracle, %.1f s' % (time.time() - t0))" 2>&1 | tail -1 | tee -a $O/bench_job_digests_5_processes.log
done
RGBM_TIMING=1 timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; grep "\[rgbm\] batch" $O/bench_train_rows_10000.log | tail -3; tail -1 $O/bench_train_rows_10000.log > $O/bench_train_rows_10000.json; python -c "
import json; d=json.loads(open('$O/bench_train_rows_10000.json').read()); print({k:d[k] for k in ('model_train_sec','repair_sec','elapsed_sec')})"
RGBM_TIMING=1 HP_PROBE_RESIDENT_ONLY=1 timeout 200 python tools/hp_search_probe.py 2>&1 | tail -12 | tee $O/hp_search_probe.log
