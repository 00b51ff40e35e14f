#!/bin/bash
# round 3: the whole GPU suite on the rewritten level grower (numerics v2)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -60 > gpurun_out/r03b_tests.log
cat gpurun_out/r03b_tests.log
