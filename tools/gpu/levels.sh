#!/bin/bash
# per-level durations of the level passes of one target (rocprofv3 kernel trace of tools/probe.py): levels.sh <out file> <target> [env assignments...]
O=$1; tg=$2; shift 2
rm -rf /tmp/tr; cd /tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/probe.py --rows 10000000 --iters 6 --targets $tg --stats 0 > /tmp/p.log 2>&1
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python - "$T" "$tg" "$*" >> $O <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]]
d = [("root" if "root" in r["Kernel_Name"] else "mt", int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
# the last iteration of the second training call: from the last root launch on
last = max(i for i, x in enumerate(d) if x[0] == "root")
prev = max(i for i, x in enumerate(d[:last]) if x[0] == "root")
print("target %s %s: root + level launches (us), last two iterations:" % (sys.argv[2], sys.argv[3]))
print("  ", [x[1] // 1000 for x in d[prev:last]], "sum", sum(x[1] for x in d[prev:last]) // 1000)
print("  ", [x[1] // 1000 for x in d[last:]], "sum", sum(x[1] for x in d[last:]) // 1000)
PY
cd $GRAFT_REPO_ROOT
