#!/bin/bash
# round 3: GPU suite, smoke, the bench line and the kernel stats at HEAD (count zeroing inside k_level_init, function attributes set once),
# then four more complete jobs: the models_md5 of every bench line must agree
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ab; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu --durations=6 ) 2>&1 | grep -v "NCCL\|RCCL\|^$" | tail -16 > $O/tests_gpu.log; grep -E "passed|failed|error" $O/tests_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 1200 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-700 $O/bench_default.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; tail -1 $O/trace_seq.log > $O/bench_steps10_seq.json
for i in 1 2 3 4; do
  timeout 600 python bench.py --no-cpu-baseline --roofline-steps 1 > $O/soak_$i.log 2>&1
  tail -1 $O/soak_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("soak", d["models_md5"], "%.10f" % d["repair_accuracy_vs_clean"], d["value"])'
done | tee $O/soak.log
python -c 'import json; d=json.load(open("'$O'/bench_default.json")); print("default", d["models_md5"])' | tee -a $O/soak.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
