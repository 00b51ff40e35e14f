#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
for b in 0 128 512 1024 2048; do
  echo "== RGBM_MT_BLOCKS=$b" | tee -a $O/blocks32.log
  ( [ "$b" != 0 ] && export RGBM_MT_BLOCKS=$b; timeout 200 python bench.py --config 100m32 --steps 10 --roofline-steps 5 --no-cpu-baseline --no-full-job 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['classes']; print('ms_per_step %.1f | root %.0f us x %d | level %.0f us x %d | frac %.3f' % (d['ms_per_step'], r['root']['avg_launch_us'], r['root']['launches'], r['level']['avg_launch_us'], r['level']['launches'], d['roofline']['frac']))" | tee -a $O/blocks32.log )
done
for b in 0 64 256; do
  echo "== 10m16 RGBM_MT_BLOCKS=$b" | tee -a $O/blocks16.log
  ( [ "$b" != 0 ] && export RGBM_MT_BLOCKS=$b; timeout 100 python tools/probe.py --iters 4 --targets 4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-130 | tee -a $O/blocks16.log )
done
