#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $O; : > $O/probe.log
for v in 1 0; do
  echo "== RGBM_MT_SPARSE=$v (1/16)" | tee -a $O/probe.log
  ( export RGBM_MT_SPARSE=$v; timeout 100 python tools/probe.py --rows 10000000 --iters 8 --targets 10,8 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-110 | tee -a $O/probe.log )
done
