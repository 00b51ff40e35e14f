#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O
timeout 200 python tools/sparse_debug2.py 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 | tee $O/debug.log
