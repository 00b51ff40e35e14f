#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for i in 1 2 3; do echo "process $i"; timeout 900 python tools/bench_repeat.py 10000000 300 3 2>&1 | grep "^job"; done | tee gpurun_out/r03w_bench_repeat.log
