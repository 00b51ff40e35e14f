#!/bin/bash
# round 4: what bounds the level pass?  perturbation builds (same results, more / fewer of one resource) x {plain, wave-specialised}
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
V=$PWD/spark-data-repair-plugin_amd/lib/variants
run() { echo "== $1" >> $O/probe.log; ( export $2 $3; timeout 100 python tools/probe.py --iters 4 --targets 7,10 2>&1 | grep "^target" | awk 'NR%2==0' | cut -c1-150 >> $O/probe.log ); }
run "plain"                    RGBM_MT_SPEC=0 X=1
run "plain, double atomics"    RGBM_MT_SPEC=0 RGBM_LIB_PATH=$V/librepairgbm_dbl.so
run "spec (4 consumers)"       RGBM_MT_SPEC=1 X=1
run "spec, double atomics"     RGBM_MT_SPEC=1 RGBM_LIB_PATH=$V/librepairgbm_dbl.so
run "spec, 8 consumers"        RGBM_MT_SPEC=1 RGBM_LIB_PATH=$V/librepairgbm_c8.so
run "spec, 2 consumers"        RGBM_MT_SPEC=1 RGBM_LIB_PATH=$V/librepairgbm_c2.so
run "spec, ring 256"           RGBM_MT_SPEC=1 RGBM_LIB_PATH=$V/librepairgbm_r256.so
run "plain, ring 256"          RGBM_MT_SPEC=0 RGBM_LIB_PATH=$V/librepairgbm_r256.so
run "plain, MT_TREES=1"        RGBM_MT_SPEC=0 RGBM_MT_TREES=1
run "spec, MT_TREES=1"         RGBM_MT_SPEC=1 RGBM_MT_TREES=1
run "spec, MT_REP=1"           RGBM_MT_SPEC=1 RGBM_MT_REP=1
cat $O/probe.log
run_pmc() { local name=$1; shift; ( cd /tmp && RGBM_MT_SPEC=1 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OLDPWD/$O/pmc_$name -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_$name.log 2>&1 ); python tools/pmc_summary.py $O/pmc_$name --seq k_level_ > $O/pmc_${name}_summary.txt 2>&1; }
run_pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR
for n in sq1 sq2; do echo "== $n"; grep -A400 "# per dispatch" $O/pmc_${n}_summary.txt | grep -E "k_level_mt|k_level_root" | tail -64; done > $O/pmc_per_dispatch.txt
f=$(find $O/pmc_sq2 -name "*kernel_trace.csv" | head -1); python - "$f" > $O/k64_level_durations.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
print("root + levels 1..6 (us):", " ".join("%7.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]][-7:]))
PY
cat $O/k64_level_durations.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
