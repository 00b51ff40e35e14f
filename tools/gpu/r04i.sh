#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
V=$PWD/spark-data-repair-plugin_amd/lib/variants
for v in "" sm1024 sm256; do
  echo "== $v" | tee -a $O/sm_threads.log
  ( [ -n "$v" ] && export RGBM_LIB_PATH=$V/librepairgbm_$v.so; RGBM_TIMING=1 timeout 120 python bench.py --train-rows 10000 --no-cpu-baseline --no-full-job --steps 300 2>&1 | grep "\[rgbm\] batch" | tail -1 | tee -a $O/sm_threads.log
    RGBM_TIMING=1 timeout 100 python tools/batch_probe.py 10000 300 2>&1 | grep -E "^target|batch of 48" | tail -6 | tee -a $O/sm_threads.log )
done
