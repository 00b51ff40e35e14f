set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT RGBM_G_ONLY RGBM_LV_BLOCKS
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
( time timeout 1200 python -m pytest tests -x -q -m gpu --durations=12 ) 2>&1 | tail -30 > $O/tests_gpu.log
cat $O/tests_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-3000
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job > $OLDPWD/$O/trace_bench10.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_kernel_stats.csv
tail -1 $O/trace_bench10.log > $O/bench_steps10.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_write.log 2>&1 )
for d in pmc_fetch pmc_write; do python tools/pmc_summary.py $O/$d --seq k_level_pass > $O/${d}_summary.txt 2>&1; done
grep -E "k_level_pass|k_level_route|k_grad_mc|k_level_final" $O/pmc_fetch_summary.txt | head -12; grep -E "k_level_pass|k_level_route|k_grad_mc|k_level_final" $O/pmc_write_summary.txt | head -12
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 2>&1 | grep "^run" | tee $O/resident_probe.log
timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 2>&1 | grep "^target" | tee $O/probe.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
