set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02t; mkdir -p $O
export TMPDIR=/tmp
( time BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --no-cpu-baseline --no-full-job --roofline-steps 5 ) > $O/bench_2ranks_shared_gpu.log 2>&1; tail -5 $O/bench_2ranks_shared_gpu.log | cut -c1-1500
( time timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 --timeout 600 --timeout-method=thread ) 2>&1 | tail -16 | tee $O/tests_gpu.log
