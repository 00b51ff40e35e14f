#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.log
timeout 900 python bench.py --no-cpu-baseline --roofline-steps 2 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('elapsed', d['elapsed_sec'], 'train', d['model_train_sec'], 'repair_sec', d['repair_sec'], 'acc', d['repair_accuracy_vs_clean'])"
timeout 600 python bench.py --train-rows 10000 --no-cpu-baseline --roofline-steps 2 > $O/bench10k.log 2>&1; tail -1 $O/bench10k.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train10k: elapsed', d['elapsed_sec'], 'train', d['model_train_sec'], 'repair_sec', d['repair_sec'], 'acc', d['repair_accuracy_vs_clean'])"
