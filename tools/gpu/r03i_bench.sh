#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'root GB/s', d['roofline']['root_scan_GBps_rank0'])"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --roofline-steps 1 > $OLDPWD/$O/trace_bench10.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_kernel_stats.csv && head -14 $O/bench_steps10_kernel_stats.csv | cut -c1-60,200-330
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
