#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
for R in 1 2 4 8 16 32; do echo "MT_REP=$R"; RGBM_MT_REP=$R timeout 300 python tools/probe.py --iters 3 --targets 4,7,10 2>&1 | grep "^target" | awk 'NR%2==0'; done 2>&1 | tee $O/sweep_rep.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/tr -- python $OLDPWD/tools/probe.py --iters 2 --targets 10 --stats 0 > $OLDPWD/$O/tr.log 2>&1 )
f=$(find $O/tr -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%7.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]][-7:]))
PY
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
