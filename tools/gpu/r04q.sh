#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04q; mkdir -p $O
( time timeout 400 python -m pytest tests -q -m gpu -x ) 2>&1 | tail -8 | tee $O/tests_gpu.log
( RGBM_MT_SPEC=1 timeout 100 python -m pytest tests/test_gpu_growers.py -q -m gpu -x ) 2>&1 | tail -3 | tee -a $O/tests_gpu.log
for v in default "RGBM_MT_SPEC=1" "RGBM_MT_THREADS=768"; do
  echo "== $v" | tee -a $O/probe.log
  ( [ "$v" != default ] && export $v; timeout 100 python tools/probe.py --iters 4 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 | tee -a $O/probe.log )
done
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-full-job 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['classes']; print('ms_per_step %.1f | root %.0f us | level %.0f us | frac %.3f | repair %.3f s' % (d['ms_per_step'], r['root']['avg_launch_us'], r['level']['avg_launch_us'], d['roofline']['frac'], d['repair_sec']))" | tee $O/bench20.log
