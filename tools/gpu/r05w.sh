#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
( cd /tmp && timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps5_seq_kernel_stats.csv; tail -1 $O/trace_seq.log > $O/bench_steps5_seq.json
rm -rf $O/trace_seq; head -4 $O/bench_steps5_seq_kernel_stats.csv | cut -c1-150
