set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
( time timeout 600 python -m pytest tests/test_gpu_growers.py -q -m gpu -k "depth or tile_edges or binary" --durations=12 ) > $O/t_growers.log 2>&1; tail -22 $O/t_growers.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_quality.py tests/test_resident_path.py tests/test_pipeline.py -q -m gpu --durations=15 ) > $O/t_new.log 2>&1; tail -30 $O/t_new.log
