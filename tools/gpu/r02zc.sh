set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r02zc
timeout 300 python bench.py --train-rows 10000 --no-cpu-baseline --roofline-steps 5 2>&1 | tail -1 > gpurun_out/r02zc/bench_default_rows.json
python -c "
import json; d=json.load(open('gpurun_out/r02zc/bench_default_rows.json')); print('train-rows 10000:', d['elapsed_sec'], d['model_train_sec'], d['repair_sec'], d['repair_accuracy_vs_clean'], d['ms_per_step'])"
