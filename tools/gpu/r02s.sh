set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02s; mkdir -p $O
export TMPDIR=/tmp
( time timeout 500 python -m pytest tests/test_gpu_growers.py -x -q -m gpu --durations=6 -p no:cacheprovider --timeout 240 --timeout-method=thread ) 2>&1 | tail -25 | tee $O/t_growers.log
if grep -q "failed\|error\|Timeout" $O/t_growers.log; then echo "growers failed: stop"; exit 1; fi
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_pipeline.py tests/test_quality.py tests/test_resident_path.py tests/test_gpu_prep.py tests/test_abi.py -x -q -m gpu --durations=6 --timeout 240 --timeout-method=thread ) 2>&1 | tail -15 | tee $O/t_rest.log
timeout 300 python tools/concurrency_check.py 3 2>&1 | tail -2 | tee $O/conc_small.log
RGBM_GROWER=level timeout 300 python tools/concurrency_check.py 3 2>&1 | tail -2 | tee $O/conc_level.log
HP_PROBE_RESIDENT_ONLY=1 timeout 300 python tools/hp_search_probe.py 2>&1 | grep resident | cut -c1-120 | tee $O/hp_small.log
timeout 600 python bench.py --train-rows 10000 --no-cpu-baseline --roofline-steps 5 2>&1 | tail -1 | cut -c1-900 | tee $O/bench_default_rows.log
RGBM_GROWER=level timeout 600 python bench.py --train-rows 10000 --no-cpu-baseline --roofline-steps 5 2>&1 | tail -1 | cut -c1-900 | tee $O/bench_default_rows_level.log
