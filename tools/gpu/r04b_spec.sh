#!/bin/bash
# round 4: wave-specialised level pass (producer waves route + fill rings, consumer waves run the LDS atomics) vs the plain pass
# (every command that runs the new kernel sits under a short timeout: a hung spin loop must not eat the GPU budget)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
( export RGBM_MT_SPEC=1; timeout 60 python -m pytest tests/test_gpu_growers.py -q -m gpu -x -k "many_classes or basic or three_way" ) 2>&1 | tail -5 > $O/tests_first.log; cat $O/tests_first.log
grep -q passed $O/tests_first.log || exit 1
( export RGBM_MT_SPEC=1; time timeout 240 python -m pytest tests -q -m gpu -x ) 2>&1 | tail -12 > $O/tests_gpu_spec.log; tail -5 $O/tests_gpu_spec.log
for v in default "RGBM_MT_SPEC=1"; do
  echo "== $v" >> $O/probe.log
  ( [ "$v" != default ] && export $v; timeout 120 python tools/probe.py --iters 4 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' >> $O/probe.log )
done
cat $O/probe.log
for v in "RGBM_MT_SPEC=0" "RGBM_MT_SPEC=1"; do
  echo "== $v" >> $O/probe32.log
  ( export $v; timeout 120 python tools/probe.py --rows 12500000 --cols 32 --iters 4 --targets 1,7 2>&1 | grep "^target" | awk 'NR%2==0' >> $O/probe32.log )
done
cat $O/probe32.log
RGBM_MT_SPEC=1 timeout 200 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench_steps20_spec.log 2>&1; tail -1 $O/bench_steps20_spec.log > $O/bench_steps20_spec.json; cut -c1-300 $O/bench_steps20_spec.json
