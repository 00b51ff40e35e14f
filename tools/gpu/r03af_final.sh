#!/bin/bash
# round 3, final measurements at HEAD: GPU suite, smoke, bench line (+ traffic), kernel stats, reference-default job, the collective path with a
# world of one, two ranks on one GPU (gloo), 100M x 32 on one GPU, RepairModel.run() on a 1M-row frame, the 48-fit search
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03af; mkdir -p $O
J() { grep '^{"metric' "$1" | tail -1 > "$2"; }
( time timeout 1500 python -m pytest tests -q -m gpu --durations=6 ) 2>&1 | grep -v "NCCL\|RCCL\|^$" | tail -16 > $O/tests_gpu.log; grep -E "passed|failed|error" $O/tests_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 1200 python bench.py > $O/bench_default.log 2>&1; J $O/bench_default.log $O/bench_default.json; cut -c1-330 $O/bench_default.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; J $O/trace_seq.log $O/bench_steps10_seq.json
timeout 600 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; J $O/bench_train_rows_10000.log $O/bench_train_rows_10000.json; cut -c1-120 $O/bench_train_rows_10000.json
timeout 600 python bench.py --force-row-sharding --steps 10 --no-full-job --no-cpu-baseline --roofline-steps 1 > $O/bench_force_row_sharding.log 2>&1; J $O/bench_force_row_sharding.log $O/bench_force_row_sharding_steps10.json; cut -c1-120 $O/bench_force_row_sharding_steps10.json
BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --no-full-job --no-cpu-baseline --roofline-steps 1 > $O/bench_2ranks.log 2>&1; J $O/bench_2ranks.log $O/bench_2ranks_on_one_gpu_gloo_steps10.json; cut -c1-120 $O/bench_2ranks_on_one_gpu_gloo_steps10.json
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 2>&1 | grep "^run" | tee $O/resident_probe.log
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 --categorical 2>&1 | grep "^run" | tee -a $O/resident_probe.log
timeout 300 python tools/hp_search_probe.py 2>&1 | tail -6 | tee $O/hp_search_probe.log
timeout 1500 python bench.py --config 100m32 --steps 20 --no-cpu-baseline --no-full-job > $O/bench_100m32_steps20.log 2>&1; J $O/bench_100m32_steps20.log $O/bench_100m32_steps20.json; cut -c1-120 $O/bench_100m32_steps20.json
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
