#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04v; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_pipeline.py tests/test_resident_path.py tests/test_quality.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3 )
( export RGBM_TIMING=1; timeout 200 python bench.py --train-rows 10000 --no-cpu-baseline 2>&1 | grep "batch of\|chain of\|metric" | cut -c1-330 > $O/timing.log )
cat $O/timing.log | cut -c1-300
for i in 1 2; do
  ( timeout 200 python bench.py --train-rows 10000 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_small_$i.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:round(d[k],4) for k in ('model_train_sec','repair_sec','elapsed_sec','repair_accuracy_vs_clean')}, d['models_md5'])" | tee -a $O/repair.log )
done
( export RGBM_NO_PIN=1; timeout 200 python bench.py --train-rows 10000 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-pin', {k:round(d[k],4) for k in ('model_train_sec','repair_sec','elapsed_sec','repair_accuracy_vs_clean')}, d['models_md5'])" )
