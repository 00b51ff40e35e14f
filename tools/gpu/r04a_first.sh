#!/bin/bash
# round 4, first GPU call: numerics v2.1 + reworked k_level_mt (two-chunk pass, slot in the record byte, scalar uniforms) + predictor skip
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
nproc > $O/nproc.txt
( time timeout 900 python -m pytest tests -q -m gpu -x --durations=5 ) 2>&1 | tail -15 > $O/tests_gpu.log; tail -5 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
for v in default "RGBM_MT_THREADS=768"; do
  echo "== $v" >> $O/probe.log
  ( [ "$v" != default ] && export $v; timeout 300 python tools/probe.py --iters 4 --targets 0,4,7,10 2>&1 | grep "^target" >> $O/probe.log )
done
cat $O/probe.log
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log > $O/bench_steps20.json; cut -c1-900 $O/bench_steps20.json
RGBM_MT_THREADS=768 timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench_steps20_t768.log 2>&1; tail -1 $O/bench_steps20_t768.log > $O/bench_steps20_t768.json; cut -c1-600 $O/bench_steps20_t768.json
# two-chunk shape: 12.5M x 32 (one GPU's shard of configs[3]), K = 24 target, both variants
for v in "RGBM_MT_ACC2=1" "RGBM_MT_ACC2=0"; do
  echo "== $v" >> $O/probe32.log
  ( export $v; timeout 400 python tools/probe.py --rows 12500000 --cols 32 --iters 4 --targets 1,7 2>&1 | grep "^target" >> $O/probe32.log )
done
cat $O/probe32.log
timeout 1200 python bench.py --config 100m32 --steps 10 --no-cpu-baseline --no-full-job > $O/bench_100m32_steps10.log 2>&1; tail -1 $O/bench_100m32_steps10.log > $O/bench_100m32_steps10.json; cut -c1-500 $O/bench_100m32_steps10.json
