#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export RGBM_POISON=1
timeout 1500 python -m pytest tests/test_gpu_growers.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_gpu_rowshard.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r03u_poison.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03u_poison.log | tail -30
