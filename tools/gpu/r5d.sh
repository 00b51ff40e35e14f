#!/bin/bash
# round 5: lock-step of the class-tree groups of a row block (wave-specialised two-chunk pass): parity, 100M x 32 step time for several windows
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O; rm -f $O/*
( timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_bench_shapes.py tests/test_gpu_rowshard.py -q -m gpu -x ) > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
for w in 0 16 4 64; do
  RGBM_MT_LOCK=$w timeout 600 python bench.py --config 100m32 --steps 6 --warmup 2 --no-cpu-baseline --no-full-job --roofline-steps 3 > $O/bench_100m32_lock$w.log 2>&1; grep '^{"metric' $O/bench_100m32_lock$w.log | tail -1 > $O/bench_100m32_lock$w.json
  python -c "import json; d=json.load(open('$O/bench_100m32_lock$w.json')); print('100m32 RGBM_MT_LOCK=$w ms_per_step %.1f frac %.4f level launch us %.0f md5 %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['classes']['level']['avg_launch_us'], d['models_md5']))" | tee -a $O/summary.txt
done
