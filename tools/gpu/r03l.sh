#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/tests.log
timeout 900 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'repair_sec', d['repair_sec'], 'roof', d['roofline']['frac'], {k:(round(v['avg_launch_us']),round(v['frac'],3)) for k,v in d['roofline']['classes'].items()}); print(d['config']['plan'])"
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-full-job --train-rows 10000 > $O/bench_train10k.log 2>&1; tail -1 $O/bench_train10k.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train10k: elapsed', d['elapsed_sec'], 'train', d['model_train_sec'], 'repair_sec', d['repair_sec'])"
