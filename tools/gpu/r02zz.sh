set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r02zz
( time timeout 300 python -m pytest tests/test_quality.py tests/test_gpu_parity.py tests/test_gpu_prep.py tests/test_abi.py -x -q -m gpu --timeout 200 --timeout-method=thread ) 2>&1 | tail -5 | tee gpurun_out/r02zz/tests.log
