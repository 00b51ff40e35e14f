set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25 > $O/tests_gpu.log
cat $O/tests_gpu.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log
RGBM_NO_PIN=1 timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-full-job 2>&1 | tail -1 | python -c "import sys,json; print('nopin upload', json.loads(sys.stdin.read())['upload'])"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job > $OLDPWD/$O/trace_bench10.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_kernel_stats.csv
tail -1 $O/trace_bench10.log > $O/bench_steps10.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_write.log 2>&1 )
for d in pmc_fetch pmc_write; do python tools/pmc_summary.py $O/$d --seq k_level_pass > $O/${d}_summary.txt 2>&1; done
grep -E "k_level_pass|k_level_route|k_grad_mc|k_level_final" $O/pmc_fetch_summary.txt | head; grep -E "k_level_pass|k_level_route|k_grad_mc|k_level_final" $O/pmc_write_summary.txt | head
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
