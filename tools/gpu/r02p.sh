set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_GRAPH
timeout 300 python tools/concurrency_check.py 3 2>&1 | tail -2 | tee $O/conc_plain.log
RGBM_GRAPH=1 timeout 300 python tools/concurrency_check.py 6 2>&1 | tail -2 | tee $O/conc_graph.log
RGBM_GRAPH=1 GPU_MAX_HW_QUEUES=8 timeout 300 python tools/concurrency_check.py 3 2>&1 | tail -2 | tee $O/conc_graph_q8.log
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/concurrency_check.py 3 2>&1 | tail -2 | tee $O/conc_plain_q8.log
HP_PROBE_RESIDENT_ONLY=1 timeout 300 python tools/hp_search_probe.py 2>&1 | grep resident | cut -c1-120 | tee $O/hp_plain.log
RGBM_GRAPH=1 HP_PROBE_RESIDENT_ONLY=1 timeout 300 python tools/hp_search_probe.py 2>&1 | grep resident | cut -c1-120 | tee $O/hp_graph.log
RGBM_GRAPH=1 GPU_MAX_HW_QUEUES=8 HP_PROBE_RESIDENT_ONLY=1 timeout 300 python tools/hp_search_probe.py 2>&1 | grep resident | cut -c1-120 | tee $O/hp_graph_q8.log
( time timeout 400 python -m pytest tests/test_gpu_growers.py -x -q -m gpu --durations=5 ) 2>&1 | grep -E "passed|failed|real|s call" | tee $O/t_growers.log
