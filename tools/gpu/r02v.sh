set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02v; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py tests/test_gpu_bench_shapes.py -x -q -m gpu --durations=5 --timeout 300 --timeout-method=thread ) 2>&1 | tail -14 | tee $O/t_joint.log
if grep -q "failed\|rror" $O/t_joint.log; then echo "tests failed: stop"; exit 1; fi
timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | tee $O/probe_joint.log
RGBM_JOINT_ROOT=0 timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | tee $O/probe_plain.log
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-full-job 2>&1 | tail -1 > $O/bench20_joint.json; python -c "
import json; d=json.load(open('$O/bench20_joint.json')); print('joint', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['repair_accuracy_vs_clean'])"
RGBM_JOINT_ROOT=0 timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-full-job 2>&1 | tail -1 > $O/bench20_plain.json; python -c "
import json; d=json.load(open('$O/bench20_plain.json')); print('plain', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['repair_accuracy_vs_clean'])"
