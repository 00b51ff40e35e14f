set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py --config 100m32 --steps 20 --warmup 2 --no-full-job --no-cpu-baseline --roofline-steps 5 ) > $O/bench_100m32_steps20.log 2>&1; tail -4 $O/bench_100m32_steps20.log | cut -c1-2500
rocm-smi --showmeminfo vram 2>/dev/null | head -8
