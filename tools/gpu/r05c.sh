#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05c; mkdir -p $O; : > $O/levels.txt
for tg in 9 8; do
for v in 1 0; do
  rm -rf /tmp/tr; cd /tmp
  RGBM_MT_SPARSE=$v timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/probe.py --rows 10000000 --iters 6 --targets $tg --stats 0 > /tmp/p.log 2>&1
  T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
  python - "$T" $v $tg >> $O/levels.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]]
d = [(r["Kernel_Name"][:24], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
# the last 2 iterations of the second training call: 7 launches each (root + 6 levels)
tail = d[-14:]
print("target %s RGBM_MT_SPARSE=%s: root + levels 1..6 (us), last two iterations:" % (sys.argv[3], sys.argv[2]))
print("  ", [x[1] // 1000 for x in tail[:7]])
print("  ", [x[1] // 1000 for x in tail[7:]])
PY
  cd $GRAFT_REPO_ROOT
done
done
cat $O/levels.txt
