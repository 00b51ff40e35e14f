#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05z; mkdir -p $O
( timeout 50 python -m pytest tests/test_gpu_predictor.py -q -m gpu -x -k "chain_and_batch or one_word" 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $O/tests.log )
