#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
for rep in 8 4 2 1; do
  echo "== RGBM_MT_REP=$rep" | tee -a $O/rep32.log
  ( export RGBM_MT_REP=$rep; timeout 200 python bench.py --config 100m32 --steps 10 --roofline-steps 5 --no-cpu-baseline --no-full-job 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['classes']; print('ms_per_step %.1f | root %.0f us x %d | level %.0f us x %d | frac %.3f' % (d['ms_per_step'], r['root']['avg_launch_us'], r['root']['launches'], r['level']['avg_launch_us'], r['level']['launches'], d['roofline']['frac']))" | tee -a $O/rep32.log )
done
