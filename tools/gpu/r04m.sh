#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
V=$PWD/spark-data-repair-plugin_amd/lib/variants
RGBM_LIB_PATH=$V/librepairgbm_smprof.so RGBM_TIMING=1 timeout 120 python bench.py --train-rows 10000 --no-cpu-baseline --no-full-job --steps 300 2>&1 | grep -E "phases|batch of" | tail -4 | tee $O/small_tree_phases.log
RGBM_LIB_PATH=$V/librepairgbm_smprof.so RGBM_TIMING=1 timeout 100 python tools/batch_probe.py 10000 300 2>&1 | grep -E "phases|batch of|^target" | tail -16 | tee -a $O/small_tree_phases.log
