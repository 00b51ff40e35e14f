#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05a; mkdir -p $O; : > $O/probe.log; : > $O/bench.log
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_growers.py tests/test_gpu_rowshard.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 | tee $O/tests.log )
grep -q "passed" $O/tests.log || exit 1
grep -q "failed" $O/tests.log && exit 1
for v in 1 0; do
  echo "== RGBM_MT_SPARSE=$v" | tee -a $O/probe.log
  ( export RGBM_MT_SPARSE=$v; timeout 200 python tools/probe.py --rows 10000000 --iters 8 --targets 10,8,6,4 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-110 | tee -a $O/probe.log )
done
for v in 1 0; do
  echo "== RGBM_MT_SPARSE=$v" | tee -a $O/bench.log
  ( export RGBM_MT_SPARSE=$v; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:round(d[k],4) for k in ('ms_per_step','value')}, d['models_md5'], round(d['roofline']['frac'],4), round(d['roofline']['classes']['level']['avg_launch_us'],1))" | tee -a $O/bench.log )
done
