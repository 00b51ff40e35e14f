set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
( time timeout 900 python -m pytest tests/test_gpu_growers.py tests/test_gpu_parity.py -x -q -m gpu --durations=5 ) > $O/tests.log 2>&1; tail -14 $O/tests.log
timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_gonly.log 2>&1; echo gonly; grep target $O/probe_gonly.log | awk 'NR%2==0'
RGBM_G_ONLY=0 timeout 300 python tools/probe.py --iters 5 --targets 7,10 > $O/probe_gh.log 2>&1; echo gh; grep target $O/probe_gh.log | awk 'NR%2==0'
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY' | tee $O/trace_level_seq.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "k_level_pass" in n or "k_level_route" in n or "k_level_final" in n or "k_grad" in n:
        print("%9.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n[:66]))
PY
( time timeout 600 python -m pytest tests/test_gpu_rowshard.py -x -q -m gpu --durations=5 ) > $O/tests_rowshard.log 2>&1; tail -12 $O/tests_rowshard.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
