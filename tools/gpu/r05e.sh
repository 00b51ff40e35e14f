#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O; : > $O/probe.log
( timeout 300 python -m pytest tests/test_gpu_growers.py tests/test_gpu_bench_shapes.py "tests/test_gpu_parity.py::test_build_model_hp_search_runs_on_gpu_and_matches_oracle_backend" -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 | tee $O/tests.log )
grep -q "passed" $O/tests.log || exit 1
grep -q "failed" $O/tests.log && exit 1
for v in 1 0; do
  echo "== RGBM_MT_SPARSE=$v" | tee -a $O/probe.log
  ( export RGBM_MT_SPARSE=$v; timeout 200 python tools/probe.py --rows 10000000 --iters 8 --targets 10,9,8 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-110 | tee -a $O/probe.log )
done
