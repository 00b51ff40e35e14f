#!/bin/bash
# round 3: levels with <= 8 row blocks per class tree skip k_level_reduce (k_level_split sums the partials): GPU suite, reference-default job, 48-fit search
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ak; mkdir -p $O
( timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "NCCL\|RCCL\|^$" | tail -5 ) | tee $O/tests_gpu.log | grep -E "passed|failed|error"
timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline --roofline-steps 1 > $O/bench_train_rows_10000.log 2>&1; grep '^{"metric' $O/bench_train_rows_10000.log | tail -1 > $O/bench_train_rows_10000.json
python -c "import json; d=json.load(open('$O/bench_train_rows_10000.json')); print('train10k', d['elapsed_sec'], d['model_train_sec'], d['repair_sec'], d['models_md5'], d['repair_accuracy_vs_clean'])"
timeout 200 python tools/hp_search_probe.py 2>&1 | grep "batch_size" | cut -c1-110 | tee $O/hp_search_probe.log
timeout 200 python tools/probe.py --iters 4 --targets 0,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-120 | tee $O/probe.log
