set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r02zb
for b in 0 8 16 24 32 48; do
  if [ $b -eq 0 ]; then unset RGBM_LV_BLOCKS; else export RGBM_LV_BLOCKS=$b; fi
  timeout 120 python tools/probe.py --iters 5 --targets 4,7,9,10 2>&1 | grep "^target" | awk 'NR%2==0' | sed "s/^/blocks=$b /" | cut -c1-150 | tee -a gpurun_out/r02zb/blocks.log
done
