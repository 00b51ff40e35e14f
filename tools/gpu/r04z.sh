#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_predictor.py -q -m gpu 2>&1 | grep -vE "^(HIP|ROCm|Hostname|Librccl)" | tail -30 | tee $O/tests.log )
