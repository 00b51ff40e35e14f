#!/bin/bash
# round 4, final set at HEAD: GPU suite, smoke, bench lines (default = complete job; the driver's call), kernel stats of the bench command, reference-default job, 48-fit search, run() on 1M rows
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05x; mkdir -p $O
( time timeout 600 python -m pytest tests -q -m gpu --durations=6 ) > $O/tests_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/tests_gpu_full.log | tail -3 | tee $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -1 | tee $O/smoke.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-260 $O/bench_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log > $O/bench_steps20_warmup5.json; cut -c1-260 $O/bench_steps20_warmup5.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; grep "^{\"metric\"" $O/trace_seq.log | tail -1 > $O/bench_steps10_seq.json
timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; tail -1 $O/bench_train_rows_10000.log > $O/bench_train_rows_10000.json; cut -c1-200 $O/bench_train_rows_10000.json
