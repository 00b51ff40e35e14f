#!/bin/bash
# round 3: row blocks per class tree of the level passes (RGBM_MT_BLOCKS): balance across workgroups when a workgroup holds one class tree
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ae; mkdir -p $O
for v in "RGBM_VLEV=0" "RGBM_VLEV=0 RGBM_MT_BLOCKS=16" "RGBM_VLEV=0 RGBM_MT_BLOCKS=32" "RGBM_VLEV=0 RGBM_MT_BLOCKS=64" "RGBM_VLEV=0 RGBM_MT_BLOCKS=128" "RGBM_MT_BLOCKS=32" "RGBM_MT_BLOCKS=64" "RGBM_MT_BLOCKS=32 RGBM_MT_REP=2" "RGBM_MT_BLOCKS=64 RGBM_MT_REP=2"; do
  echo "== $v"; env $v timeout 300 python tools/probe.py --iters 4 --targets 4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-130
done 2>&1 | tee $O/probe_blocks.log
