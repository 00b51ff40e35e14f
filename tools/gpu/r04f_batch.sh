#!/bin/bash
# round 4: the batched small-table trainer -- parity tests, then timing of a 48-fit batch against 48 single calls
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
( timeout 240 python -m pytest tests/test_gpu_batch.py -q -m gpu -x ) 2>&1 | tail -25 > $O/tests_batch.log; cat $O/tests_batch.log
timeout 200 python tools/batch_probe.py 2>&1 | tail -12 | tee $O/batch_probe.log
