#!/bin/bash
# round 3: what a level pass spends its time on AT HEAD (T sized for replication, per-feature replication): production / no LDS atomics
# (-DMT_DBG=1) / routing only (-DMT_DBG=2); K = 8 / 24 / 64 targets, hist ms of 4 iterations (28-32 launches)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ai; mkdir -p $O
for v in "" "_dbg1" "_dbg2"; do
  echo "== library librepairgbm$v.so"; RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm$v.so timeout 300 python tools/probe.py --iters 4 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-140
done 2>&1 | tee $O/ablation.log
