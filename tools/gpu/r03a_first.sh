#!/bin/bash
# round 3, first GPU call: smoke + the grower / parity suites on the rewritten level grower (numerics v2)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python __graft_entry__.py smoke > gpurun_out/r03a_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03a_smoke.log
tail -3 gpurun_out/r03a_smoke.log
timeout 1500 python -m pytest tests/test_gpu_growers.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03a_tests.log
cat gpurun_out/r03a_tests.log
