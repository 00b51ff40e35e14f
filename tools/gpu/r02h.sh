set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -22 > $O/tests_gpu.log
cat $O/tests_gpu.log
timeout 600 python tools/hp_search_probe.py > $O/hp_search.log 2>&1; grep -E "batch_size|identical|Error|error" $O/hp_search.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log | cut -c1-700
