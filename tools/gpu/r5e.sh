#!/bin/bash
# round 5: does the lock-step change the level pass's HBM fetch?  FETCH_SIZE + duration per k_level_mt dispatch, K = 24 target of the 100M x 32 table
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O; rm -rf $O/*
for w in 0 16; do
  ( cd /tmp && RGBM_MT_LOCK=$w timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_lock$w -- python $GRAFT_REPO_ROOT/tools/probe.py --rows 100000000 --cols 32 --seed 43 --parallel 1 --iters 1 --targets 7 --stats 0 > $O/pmc_lock$w.log 2>&1 )
  python - $O/pmc_lock$w $w <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, os
d, w = sys.argv[1], sys.argv[2]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
fetch = {}
for f in cc:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_level_mt" in r["Kernel_Name"]:
            fetch[r["Dispatch_Id"]] = fetch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        if "k_level_mt" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(fetch, key=lambda x: int(x))
print("RGBM_MT_LOCK=%s: k_level_mt dispatches (FETCH_SIZE KB x 2 -> GB, us):" % w)
print("  ", [(round(fetch[i] * 2 * 1024 / 1e9, 1), int(dur.get(i, 0))) for i in ids[-8:]])
PY
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
