#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05d; mkdir -p $O; : > $O/lds.txt
for lds in 0 153000 145000; do
  echo "== RGBM_LV_LDS=$lds (0 = default), sparse sweep off" | tee -a $O/lds.txt
  ( export RGBM_MT_SPARSE=0; [ $lds != 0 ] && export RGBM_LV_LDS=$lds; timeout 200 python tools/probe.py --rows 10000000 --iters 8 --targets 10,8,4 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-120 | tee -a $O/lds.txt )
done
