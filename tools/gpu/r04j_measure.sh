#!/bin/bash
# round 4: measurements at HEAD -- bench line, kernel stats, PMC traffic for both configs, 100M x 32 on one GPU
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-400 $O/bench_default.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; tail -1 $O/trace_seq.log > $O/bench_steps10_seq.json
ALL=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets $ALL --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python $OLDPWD/tools/probe.py --iters 1 --targets $ALL --stats 0 > $OLDPWD/$O/pmc_write.log 2>&1 )
python tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write r04j > $O/traffic_json.log 2>&1; tail -5 $O/traffic_json.log
T8=0,1,2,3,4,5,6,7
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc32_fetch -- python $OLDPWD/tools/probe.py --rows 100000000 --cols 32 --seed 43 --parallel 1 --iters 1 --targets $T8 --stats 0 > $OLDPWD/$O/pmc32_fetch.log 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc32_write -- python $OLDPWD/tools/probe.py --rows 100000000 --cols 32 --seed 43 --parallel 1 --iters 1 --targets $T8 --stats 0 > $OLDPWD/$O/pmc32_write.log 2>&1 )
python tools/make_traffic_json.py $O/pmc32_fetch $O/pmc32_write r04j_100m32 --rows 100000000 --cols 32 --targets $T8 > $O/traffic32_json.log 2>&1; tail -5 $O/traffic32_json.log
cp profiles/traffic.json $O/traffic.json; cp profiles/r04j*_hbm_traffic_pmc.txt $O/ 2>/dev/null
timeout 600 python bench.py --config 100m32 --steps 20 --no-cpu-baseline --no-full-job > $O/bench_100m32_steps20.log 2>&1; tail -1 $O/bench_100m32_steps20.log > $O/bench_100m32_steps20.json; cut -c1-300 $O/bench_100m32_steps20.json
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
