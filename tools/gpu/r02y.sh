set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02y; mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python bench.py ) > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | cut -c1-500; tail -4 $O/bench_default.log | grep real
