#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python tools/determinism_check.py 10000000 60 3 2>&1 | grep -E "repeat|Error|error" | tee gpurun_out/r03r_determinism.log
