#!/bin/bash
# A/B builds of librepairgbm.so for timing experiments (RGBM_LIB_PATH=<file> selects one): tools/build_variants.sh name "-DFLAG ..." [name flags ...]
cd "$(dirname "$0")/../spark-data-repair-plugin_amd/csrc" || exit 1
mkdir -p ../lib/variants
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc $f -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-result -shared \
     -o ../lib/variants/librepairgbm_$n.so rgbm.hip rgbm_prep.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" &
done
wait
ls -la ../lib/variants
