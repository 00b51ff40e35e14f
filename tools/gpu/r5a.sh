#!/bin/bash
# round 5, first GPU call: GPU suite at HEAD (fused last pass + gradients on by default), A/B of the fusion on the bench
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
( time timeout 900 python -m pytest tests -q -m gpu -x --durations=8 ) > $O/tests_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/tests_gpu_full.log | tail -3 | tee $O/tests_gpu.log
tail -30 $O/tests_gpu_full.log | grep -v "^$" | head -40
for f in 1 0; do
  echo "== RGBM_FUSE_GRAD=$f"; RGBM_FUSE_GRAD=$f timeout 300 python tools/probe.py --iters 8 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 | tee -a $O/probe_fuse_$f.txt
done
RGBM_FUSE_GRAD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-job --roofline-steps 5 > $O/bench_fuse0.log 2>&1; grep '^{"metric' $O/bench_fuse0.log | tail -1 > $O/bench_fuse0.json; python -c "import json; d=json.load(open('$O/bench_fuse0.json')); print('fuse0 ms_per_step', d['ms_per_step'], d['models_md5'])"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fuse1.log 2>&1; grep '^{"metric' $O/bench_fuse1.log | tail -1 > $O/bench_fuse1.json; python -c "import json; d=json.load(open('$O/bench_fuse1.json')); print('fuse1 ms_per_step', d['ms_per_step'], d['value'], d['elapsed_sec'], d['roofline']['frac'], d['models_md5'])"
