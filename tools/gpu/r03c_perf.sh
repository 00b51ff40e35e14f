#!/bin/bash
# round 3: first performance survey of the rewritten level grower
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
timeout 600 python tools/probe.py --iters 5 --targets 0,4,7,10 2>&1 | grep "^target" | tee $O/probe.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 3 --targets 10 --stats 0 > $OLDPWD/$O/trace_probe.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/probe_k64_kernel_stats.csv && head -25 $O/probe_k64_kernel_stats.csv | cut -c1-200
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY' | tee $O/probe_k64_level_sequence.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]]
for r in sel[-16:]:
    print("%-40s %9.1f us grid=%s" % (r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
timeout 900 python bench.py --steps 20 --no-cpu-baseline > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log | cut -c1-2500
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
