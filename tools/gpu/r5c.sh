#!/bin/bash
# round 5: feature rotation of the level pass's LDS atomics (MT_ROT): parity, per-level durations and A/B against the MT_ROT=0 build
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O; rm -f $O/*
V=$GRAFT_REPO_ROOT/spark-data-repair-plugin_amd/lib/variants/librepairgbm_rot0.so
( timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_rowshard.py -q -m gpu -x ) > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
for tg in 10 6; do
  bash tools/gpu/levels.sh $O/levels.txt $tg X=rot1
  bash tools/gpu/levels.sh $O/levels.txt $tg RGBM_LIB_PATH=$V
done
cat $O/levels.txt
echo "== rot1"; timeout 300 python tools/probe.py --iters 8 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 | tee -a $O/probe_rot1.txt
echo "== rot0"; RGBM_LIB_PATH=$V timeout 300 python tools/probe.py --iters 8 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 | tee -a $O/probe_rot0.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-job --roofline-steps 5 > $O/bench_rot1.log 2>&1; grep '^{"metric' $O/bench_rot1.log | tail -1 > $O/bench_rot1.json; python -c "import json; d=json.load(open('$O/bench_rot1.json')); print('rot1 ms_per_step', d['ms_per_step'], d['roofline']['frac'], d['models_md5'])"
RGBM_LIB_PATH=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-job --roofline-steps 5 > $O/bench_rot0.log 2>&1; grep '^{"metric' $O/bench_rot0.log | tail -1 > $O/bench_rot0.json; python -c "import json; d=json.load(open('$O/bench_rot0.json')); print('rot0 ms_per_step', d['ms_per_step'], d['roofline']['frac'], d['models_md5'])"
timeout 600 python bench.py --config 100m32 --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --roofline-steps 3 > $O/bench_100m32_rot1.log 2>&1; grep '^{"metric' $O/bench_100m32_rot1.log | tail -1 > $O/bench_100m32_rot1.json; python -c "import json; d=json.load(open('$O/bench_100m32_rot1.json')); print('100m32 rot1 ms_per_step', d['ms_per_step'], d['roofline']['frac'], d['roofline']['classes']['level']['avg_launch_us'])"
