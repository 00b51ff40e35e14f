set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02a
unset RGBM_LEVEL_SPLIT
timeout 900 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02a/tests_growers.log
cat gpurun_out/r02a/tests_growers.log
for sp in 0 1; do
  RGBM_LEVEL_SPLIT=$sp timeout 600 python tools/probe.py --iters 5 --targets 0,4,7,10 > gpurun_out/r02a/probe_split$sp.log 2>&1
  cat gpurun_out/r02a/probe_split$sp.log
done
