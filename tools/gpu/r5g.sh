#!/bin/bash
# round 5: one rank's share of the 8-GPU north-star job on one GPU -- a 12.5M x 32 row shard, EVERY target on the collective path (RCCL world of
# one) -- with the fusion group (all eight targets in flight, one collective per step) against one target after another
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O; rm -f $O/*
J() { grep '^{"metric' "$1" | tail -1 > "$2"; }
for f in 1 0; do
  RGBM_FUSION=$f timeout 600 python bench.py --config 100m32 --rows 12500000 --force-row-sharding --row-shard-all --steps 20 --warmup 3 --no-cpu-baseline --no-full-job --roofline-steps 2 > $O/shard_fusion$f.log 2>&1; J $O/shard_fusion$f.log $O/shard_fusion$f.json
  python -c "import json; d=json.load(open('$O/shard_fusion$f.json')); print('12.5M x 32 shard, all 8 targets row-sharded (world of one), RGBM_FUSION=$f: ms_per_step %.1f md5 %s' % (d['ms_per_step'], d['models_md5']))" | tee -a $O/summary.txt
done
RGBM_FUSION=1 RGBM_TARGET_CONCURRENCY=8 timeout 600 python bench.py --config 100m32 --rows 12500000 --force-row-sharding --row-shard-all --steps 20 --warmup 3 --no-cpu-baseline --no-full-job --roofline-steps 2 > $O/shard_fusion8.log 2>&1; J $O/shard_fusion8.log $O/shard_fusion8.json
python -c "import json; d=json.load(open('$O/shard_fusion8.json')); print('... fusion, 8 in flight: ms_per_step %.1f md5 %s' % (d['ms_per_step'], d['models_md5']))" | tee -a $O/summary.txt
timeout 600 python bench.py --config 100m32 --rows 12500000 --steps 20 --warmup 3 --no-cpu-baseline --no-full-job --roofline-steps 2 > $O/shard_plain.log 2>&1; J $O/shard_plain.log $O/shard_plain.json
python -c "import json; d=json.load(open('$O/shard_plain.json')); print('12.5M x 32, no collectives, six targets in flight: ms_per_step %.1f md5 %s' % (d['ms_per_step'], d['models_md5']))" | tee -a $O/summary.txt
