#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
for v in "" 1 2; do
  lib=""; [ -n "$v" ] && lib="$PWD/spark-data-repair-plugin_amd/lib/dbg/librepairgbm_dbg$v.so"
  echo "== variant ${v:-production}"
  ( cd /tmp && RGBM_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/tr$v -- python $OLDPWD/tools/probe.py --iters 2 --targets 10 --stats 0 > $OLDPWD/$O/tr$v.log 2>&1 )
  f=$(find $O/tr$v -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%7.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]][-7:]))
PY
done 2>&1 | tee $O/variants.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
