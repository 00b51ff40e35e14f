#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/probe.py --iters 5 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | tee $O/probe.log
echo "JOINT_ROOT=0"; RGBM_JOINT_ROOT=0 timeout 600 python tools/probe.py --iters 5 --targets 4,10 2>&1 | grep "^target" | awk 'NR%2==0' | tee $O/probe_nojoint.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OLDPWD/$O/pmc -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc.log 2>&1 )
python tools/pmc_summary.py $O/pmc --seq k_level_ 2>&1 | grep -A400 "# per dispatch" | grep -E "k_level_mt|k_level_root" | grep -E "^(47|51) " | cut -c1-120 > $O/pmc_per_dispatch.txt; cat $O/pmc_per_dispatch.txt
f=$(find $O/pmc -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for r in [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]][-7:]:
    print("%-30s %9.1f us" % (r["Kernel_Name"][:30], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
