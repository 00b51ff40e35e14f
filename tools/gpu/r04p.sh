#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
( timeout 200 python -m pytest tests/test_gpu_batch.py -q -m gpu -x ) 2>&1 | tail -12 | tee $O/tests_batch.log
RGBM_TIMING=1 HP_PROBE_RESIDENT_ONLY=1 timeout 200 python tools/hp_search_profile.py 2>&1 | grep -E "function calls|search_on_table|train_batch|repair_chain|count_codes|close|f1_score|gather_rows|__del__" | head -14 | tee $O/hp_search_profile.log
HP_PROBE_RESIDENT_ONLY=1 timeout 200 python tools/hp_search_probe.py 2>&1 | grep -E "resident" | tee $O/hp_search_probe.log
