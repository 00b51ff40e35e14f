#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $O
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o small -- python $GRAFT_REPO_ROOT/bench.py --train-rows 10000 --no-cpu-baseline > $O/rocprof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$T" > $O/predict_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_predict", "k_pack_bins", "k_softmax", "k_fill_cells"))]
for r in keep[-80:]:
    print(r["Kernel_Name"][:34], int(r["Start_Timestamp"]) % 10**10, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
PY
rm -rf $O/prof
