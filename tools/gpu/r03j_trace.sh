#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
for t in 0 4 7; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/tr$t -- python $OLDPWD/tools/probe.py --iters 2 --targets $t --stats 0 > $OLDPWD/$O/tr$t.log 2>&1 )
  f=$(find $O/tr$t -name "*kernel_trace.csv" | head -1); echo "== target $t"; python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last iteration: from the last k_grad* to the end
gi = max(i for i, r in enumerate(rows) if "k_grad" in r["Kernel_Name"])
prev_end = None
for r in rows[gi:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-28s %8.1f us  gap %6.1f  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][-28:], (e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    prev_end = e
PY
done 2>&1 | tee $O/iteration_timeline.txt | tail -120
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
