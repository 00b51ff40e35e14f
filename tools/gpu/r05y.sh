#!/bin/bash
# round 4, last set at HEAD (sparse sweep of the level pass without LDS rings): GPU suite, smoke, the bench line
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
( time timeout 400 python -m pytest tests -q -m gpu -x ) > $O/tests_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/tests_gpu_full.log | tail -3 | tee $O/tests_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tail -1 | tee $O/smoke.log
timeout 300 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-260 $O/bench_default.json
