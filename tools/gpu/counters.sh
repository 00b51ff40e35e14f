#!/bin/bash
# SQ counters of the level passes of one target, per library build: counters.sh <out file> <target> <label> [VAR=val ...]
# (two rocprofv3 --pmc passes of tools/probe.py, as the guide prescribes: no trace domains next to --pmc except the kernel trace)
O=$1; tg=$2; label=$3; shift 3
cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pm
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/probe.py --rows 10000000 --iters 2 --targets $tg --stats 0 > /tmp/pm.log 2>&1
  python - "$label" "$grp" >> $O <<'PY'
import csv, glob, sys, collections
agg = collections.OrderedDict()
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_level_mt" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            agg["#dispatch_rows"] = agg.get("#dispatch_rows", 0) + 1
print("%-10s k_level_mt (all level launches of 2 x 2 iterations): %s" % (sys.argv[1], "  ".join("%s=%.4g" % kv for kv in agg.items())))
PY
done
cd $GRAFT_REPO_ROOT
