#!/bin/bash
# round 4: batched small-table trainer wired into the search and the job
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_prep.py tests/test_pipeline.py tests/test_resident_path.py -q -m gpu -x ) 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log
timeout 200 python tools/hp_search_probe.py 2>&1 | tail -8 | tee $O/hp_search_probe.log
RGBM_SMALL_ROWS=0 timeout 200 python tools/hp_search_probe.py 2>&1 | tail -8 | tee $O/hp_search_probe_unbatched.log
timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; tail -1 $O/bench_train_rows_10000.log > $O/bench_train_rows_10000.json; cut -c1-700 $O/bench_train_rows_10000.json
RGBM_SMALL_ROWS=0 timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000_unbatched.log 2>&1; tail -1 $O/bench_train_rows_10000_unbatched.log | cut -c1-700
