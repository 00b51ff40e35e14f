set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
( time timeout 900 python -m pytest tests/test_gpu_growers.py tests/test_gpu_parity.py tests/test_gpu_rowshard.py -x -q -m gpu --durations=4 ) > $O/tests.log 2>&1; tail -10 $O/tests.log
timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_gonly.log 2>&1; echo gonly; grep target $O/probe_gonly.log | awk 'NR%2==0'
RGBM_G_ONLY=0 timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_gh.log 2>&1; echo gh; grep target $O/probe_gh.log | awk 'NR%2==0'
timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench20_gonly.log 2>&1; tail -1 $O/bench20_gonly.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gonly ms_per_step', d['ms_per_step'], d['roofline']['frac'], d['roofline']['with_route']['frac'])"
RGBM_G_ONLY=0 timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench20_gh.log 2>&1; tail -1 $O/bench20_gh.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gh ms_per_step', d['ms_per_step'], d['roofline']['frac'], d['roofline']['with_route']['frac'])"
