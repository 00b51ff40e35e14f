#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $O
( timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/pytest_gpu.log )
( timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log )
