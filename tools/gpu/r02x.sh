set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02x; mkdir -p $O
export TMPDIR=/tmp
for c in 4 6 8 6 4 8; do
  RGBM_TARGET_CONCURRENCY=$c timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-full-job --roofline-steps 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('concurrency $c: ms_per_step %.2f' % d['ms_per_step'])" | tee -a $O/conc.log
done
