#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04n; mkdir -p $O
V=$PWD/spark-data-repair-plugin_amd/lib/variants
( timeout 200 python -m pytest tests/test_gpu_batch.py -q -m gpu -x ) 2>&1 | tail -6 | tee $O/tests_batch.log
( RGBM_SMALL_PACKED=0 timeout 200 python -m pytest tests/test_gpu_batch.py -q -m gpu -x ) 2>&1 | tail -3 | tee -a $O/tests_batch.log
RGBM_TIMING=1 timeout 120 python bench.py --train-rows 10000 --no-cpu-baseline --no-full-job --steps 300 2>&1 | grep -E "phases|batch of" | tail -2 | tee $O/small_tree.log
RGBM_LIB_PATH=$V/librepairgbm_smprof.so RGBM_TIMING=1 timeout 120 python bench.py --train-rows 10000 --no-cpu-baseline --no-full-job --steps 300 2>&1 | grep -E "phases|batch of" | tail -2 | tee -a $O/small_tree.log
RGBM_TIMING=1 timeout 100 python tools/batch_probe.py 10000 300 2>&1 | grep -E "phases|batch of 48|^target" | tail -9 | tee -a $O/small_tree.log
RGBM_TIMING=1 HP_PROBE_RESIDENT_ONLY=1 timeout 200 python tools/hp_search_probe.py 2>&1 | grep -E "resident|batch of 48" | tee $O/hp_search_probe.log
