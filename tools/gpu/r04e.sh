#!/bin/bash
# round 4: wave-specialised pass with the ring sync words as real LDS accesses (they were FLAT: every access waited for all outstanding loads)
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
V=$PWD/spark-data-repair-plugin_amd/lib/variants
( export RGBM_MT_SPEC=1; timeout 90 python -m pytest tests/test_gpu_growers.py -q -m gpu -x ) 2>&1 | tail -5 > $O/tests_first.log; cat $O/tests_first.log
grep -q passed $O/tests_first.log || exit 1
run() { echo "== $1" >> $O/probe.log; ( export $2 $3; timeout 100 python tools/probe.py --iters 4 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 >> $O/probe.log ); }
run "plain"                    RGBM_MT_SPEC=0 X=1
run "spec (4 consumers)"       RGBM_MT_SPEC=1 X=1
run "spec, double atomics"     RGBM_MT_SPEC=1 RGBM_LIB_PATH=$V/librepairgbm_dbl.so
run "spec, ring 256"           RGBM_MT_SPEC=1 RGBM_LIB_PATH=$V/librepairgbm_r256.so
run "spec, MT_REP=4"           RGBM_MT_SPEC=1 RGBM_MT_REP=4
run "spec, MT_REP=16"          RGBM_MT_SPEC=1 RGBM_MT_REP=16
cat $O/probe.log
for v in "RGBM_MT_SPEC=0" "RGBM_MT_SPEC=1"; do
  echo "== $v" >> $O/probe32.log
  ( export $v; timeout 120 python tools/probe.py --rows 12500000 --cols 32 --iters 4 --targets 1,7 2>&1 | grep "^target" | awk 'NR%2==0' >> $O/probe32.log )
done
cat $O/probe32.log
( export RGBM_MT_SPEC=1; time timeout 240 python -m pytest tests -q -m gpu -x ) 2>&1 | tail -12 > $O/tests_gpu_spec.log; tail -5 $O/tests_gpu_spec.log
RGBM_MT_SPEC=1 timeout 200 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench_steps20_spec.log 2>&1; tail -1 $O/bench_steps20_spec.log > $O/bench_steps20_spec.json; cut -c1-300 $O/bench_steps20_spec.json
