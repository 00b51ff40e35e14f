#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
for v in qs1 qs2 walk; do
  [ $v = walk ] && export RGBM_PREDICTOR=walk
  timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 > $O/bench_$v.log 2>&1; tail -1 $O/bench_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'elapsed', d['elapsed_sec'], 'repair_sec', d['repair_sec'], 'acc %.10f' % d['repair_accuracy_vs_clean'])"
done
