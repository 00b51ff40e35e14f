#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04o; mkdir -p $O
timeout 200 python tools/hp_search_profile.py 2>&1 | tail -50 | tee $O/hp_search_profile.log
