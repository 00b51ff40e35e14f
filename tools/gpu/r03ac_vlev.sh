#!/bin/bash
# round 3: level passes over the grouped record (RGBM_VLEV): GPU suite, then A/B on the K = 8 / 24 / 64 targets and the group cap sweep
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ac; mkdir -p $O
( time timeout 1200 python -m pytest tests -q -m gpu -x --durations=5 ) 2>&1 | grep -v "NCCL\|RCCL\|^$" | tail -30 > $O/tests_gpu.log; grep -E "passed|failed|error" $O/tests_gpu.log | tail -3
for v in "RGBM_VLEV=0" "RGBM_VLEV_CAP=32" "RGBM_VLEV_CAP=64" "RGBM_VLEV_CAP=128" "RGBM_VLEV_CAP=64 RGBM_MT_REP=4" "RGBM_VLEV_CAP=64 RGBM_MT_REP=2"; do
  echo "== $v"; env $v timeout 300 python tools/probe.py --iters 4 --targets 4,7,10 2>&1 | grep "^target" | awk 'NR%2==0'
done 2>&1 | tee $O/probe_ab.log
