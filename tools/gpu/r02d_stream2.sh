set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/tests_gpu.log
cat $O/tests_gpu.log
RGBM_LEVEL_SPLIT=1 timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_split1.log 2>&1
grep target $O/probe_split1.log | awk 'NR%2==0'
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 2 --targets 10 --stats 0 > $OLDPWD/$O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" > $O/trace_level_seq.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "k_level_" in n or "k_grad" in n:
        print("%9.1f us  grid=(%s,%s,%s)  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), n[:70]))
PY
head -45 $O/trace_level_seq.txt
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/bench20.log 2>&1; tail -2 $O/bench20.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
