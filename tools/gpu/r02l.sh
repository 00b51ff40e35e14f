set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02l; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
( time RGBM_TIMING=1 timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -s --durations=5 ) > $O/t_parity.log 2>&1
grep -E "hp search|passed|failed|real|Thread|File|rgbm\] target" $O/t_parity.log | tail -60
timeout 300 python tools/probe.py --iters 5 --targets 0,1,4,7,10 > $O/probe.log 2>&1; grep target $O/probe.log | awk 'NR%2==0'
timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py -x -q -m gpu 2>&1 | grep -E "passed|failed" 
