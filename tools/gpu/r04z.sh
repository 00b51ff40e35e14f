#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $O; : > $O/grower_shares.txt
for tk in "10 64" "8 32" "6 16" "4 8" "0 1"; do
  set -- $tk
  rm -rf /tmp/trace; mkdir -p /tmp/trace
  RGBM_TRACE=/tmp/trace RGBM_TRACE_K=$2 RGBM_TRACE_ITERS=2:4 timeout 200 python tools/probe.py --rows 2000000 --iters 6 --targets $1 --stats 0 > /tmp/probe.log 2>&1 || tail -3 /tmp/probe.log
  timeout 100 python tools/grower_live_fraction.py /tmp/trace $1 $2 2>&1 | tail -6 | tee -a $O/grower_shares.txt
done
