set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02w; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-600
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 > $OLDPWD/$O/trace_bench10_seq.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv
grep '^{"metric"' $O/trace_bench10_seq.log > $O/bench_steps10_seq.json
( time timeout 330 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_growers.py --deselect tests/test_gpu_rowshard.py --deselect tests/test_gpu_bench_shapes.py --timeout 200 --timeout-method=thread ) 2>&1 | tail -6 | tee $O/tests_rest.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
