#!/bin/bash
# round 3: grouped record, routing decoded from the group codes (one record per row): parity of the grower tests, then A/B and the sweep
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ad; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py tests/test_gpu_bench_shapes.py -q -m gpu -x ) 2>&1 | grep -v "NCCL\|RCCL\|^$" | tail -30 > $O/tests_gpu.log; grep -E "passed|failed|error" $O/tests_gpu.log | tail -3
for v in "RGBM_VLEV=0" "RGBM_VLEV_CAP=64" "RGBM_VLEV_CAP=64 RGBM_MT_REP=4" "RGBM_VLEV_CAP=64 RGBM_MT_REP=2" "RGBM_VLEV_CAP=64 RGBM_MT_REP=1" "RGBM_VLEV_CAP=32 RGBM_MT_REP=2" "RGBM_VLEV_CAP=128 RGBM_MT_REP=2"; do
  echo "== $v"; env $v timeout 300 python tools/probe.py --iters 4 --targets 4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150
done 2>&1 | tee $O/probe_ab.log
