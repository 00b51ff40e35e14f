#!/bin/bash
# round 5: the fused last pass + gradients after its loads were batched: parity of the fused forms, A/B against the two kernels
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -q -m gpu -x ) > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
for f in 1 0; do
  echo "== RGBM_FUSE_GRAD=$f"; RGBM_FUSE_GRAD=$f timeout 300 python tools/probe.py --iters 8 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 | tee -a $O/probe_fuse_$f.txt
done
for f in 0 1; do
RGBM_FUSE_GRAD=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-job --roofline-steps 5 > $O/bench_fuse$f.log 2>&1; grep '^{"metric' $O/bench_fuse$f.log | tail -1 > $O/bench_fuse$f.json; python -c "import json; d=json.load(open('$O/bench_fuse$f.json')); print('fuse$f ms_per_step', d['ms_per_step'], d['models_md5'])"
done
