#!/bin/bash
# round 4: measurements at HEAD for DESIGN.md -- GPU suite, smoke, bench line (+ kernel stats of the same command), reference-default job, 48-fit search, 100M x 32 on one GPU
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
( time timeout 600 python -m pytest tests -q -m gpu --durations=6 ) 2>&1 | tail -22 > $O/tests_gpu.log; grep -E "passed|failed" $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-300 $O/bench_default.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; grep "^{\"metric\"" $O/trace_seq.log | tail -1 > $O/bench_steps10_seq.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_def -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --roofline-steps 1 > $OLDPWD/$O/trace_def.log 2>&1 )
f=$(find $O/trace_def -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_concurrent_kernel_stats.csv
timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; tail -1 $O/bench_train_rows_10000.log > $O/bench_train_rows_10000.json; cut -c1-200 $O/bench_train_rows_10000.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_small -- python $OLDPWD/bench.py --train-rows 10000 --no-cpu-baseline > $OLDPWD/$O/trace_small.log 2>&1 )
f=$(find $O/trace_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_train_rows_10000_kernel_stats.csv
HP_PROBE_RESIDENT_ONLY=1 timeout 200 python tools/hp_search_probe.py 2>&1 | grep -E "resident" | tee $O/hp_search_probe.log
timeout 600 python bench.py --config 100m32 --steps 20 --no-cpu-baseline --no-full-job > $O/bench_100m32_steps20.log 2>&1; tail -1 $O/bench_100m32_steps20.log > $O/bench_100m32_steps20.json; cut -c1-300 $O/bench_100m32_steps20.json
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 2>&1 | grep "^run" | tee $O/resident_probe.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
