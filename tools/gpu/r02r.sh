set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02r; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 > $OLDPWD/$O/trace_bench10_seq.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv
grep '^{"metric"' $O/trace_bench10_seq.log > $O/bench_steps10_seq.json
head -6 $O/bench_steps10_seq_kernel_stats.csv | cut -c1-60,200-400
( time timeout 900 python bench.py --config 100m32 --no-cpu-baseline --roofline-steps 5 ) > $O/bench_100m32_full.log 2>&1; grep '^{"metric"' $O/bench_100m32_full.log | cut -c1-1200
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
