#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python tools/job_determinism.py 10000000 300 5 3 2>&1 | grep -E "repeat|Error|error" | tee gpurun_out/r03s_job_determinism.log
