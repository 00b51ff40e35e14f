set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm_nt.so timeout 300 python tools/probe.py --iters 5 --targets 7,10 > $O/probe_nt.log 2>&1
timeout 300 python tools/probe.py --iters 5 --targets 7,10 > $O/probe_plain.log 2>&1
grep target $O/probe_nt.log $O/probe_plain.log | awk 'NR%2==0'
( cd /tmp && RGBM_LIB_PATH=$OLDPWD/spark-data-repair-plugin_amd/lib/librepairgbm_nt.so timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch_nt -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_fetch_nt.log 2>&1 )
python tools/pmc_summary.py $O/pmc_fetch_nt > $O/pmc_fetch_nt_summary.txt 2>&1; grep -E "k_level_pass|k_level_route" $O/pmc_fetch_nt_summary.txt
for c in 1 2 4 8; do timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-full-job --concurrency $c > $O/bench20_c$c.log 2>&1; tail -1 $O/bench20_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conc', $c, 'ms_per_step', d['ms_per_step'], 'train', d['model_train_sec'])"; done
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -5 > $O/tests_shapes.log; cat $O/tests_shapes.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
