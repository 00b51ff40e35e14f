#!/bin/bash
# round 5: fusion group (several row-sharded targets of a rank in flight, one collective per step) -- tests, and the collective path of the bench
# with a world of one: fused against one target after another
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O; rm -f $O/*
( timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_rowshard.py tests/test_gpu_batch.py -q -m gpu -x ) > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3; grep -E "Error|assert" $O/tests.log | head -10
J() { grep '^{"metric' "$1" | tail -1 > "$2"; }
for f in 1 0; do
  RGBM_FUSION=$f timeout 400 python bench.py --force-row-sharding --steps 10 --warmup 3 --no-cpu-baseline --no-full-job --roofline-steps 2 > $O/bench_frs_fusion$f.log 2>&1; J $O/bench_frs_fusion$f.log $O/bench_frs_fusion$f.json
  python -c "import json; d=json.load(open('$O/bench_frs_fusion$f.json')); print('10m16 --force-row-sharding RGBM_FUSION=$f: ms_per_step %.1f md5 %s' % (d['ms_per_step'], d['models_md5']), d['config']['parallelism'][:60])" | tee -a $O/summary.txt
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-job --roofline-steps 2 > $O/bench_plain.log 2>&1; J $O/bench_plain.log $O/bench_plain.json
python -c "import json; d=json.load(open('$O/bench_plain.json')); print('10m16 six targets in flight (no collectives): ms_per_step %.1f md5 %s' % (d['ms_per_step'], d['models_md5']))" | tee -a $O/summary.txt
for f in 1 0; do
  RGBM_FUSION=$f timeout 900 python bench.py --config 100m32 --force-row-sharding --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --roofline-steps 2 > $O/bench_100m32_frs_fusion$f.log 2>&1; J $O/bench_100m32_frs_fusion$f.log $O/bench_100m32_frs_fusion$f.json
  python -c "import json; d=json.load(open('$O/bench_100m32_frs_fusion$f.json')); print('100m32 --force-row-sharding RGBM_FUSION=$f: ms_per_step %.1f md5 %s' % (d['ms_per_step'], d['models_md5']))" | tee -a $O/summary.txt
done
