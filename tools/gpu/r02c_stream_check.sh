set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
timeout 900 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_growers.log
cat $O/tests_growers.log
RGBM_LEVEL_SPLIT=1 timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_split1.log 2>&1
RGBM_LEVEL_SPLIT=0 timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_split0.log 2>&1
RGBM_LEVEL_SPLIT=1 timeout 400 python tools/probe.py --iters 5 --targets 4,7,10 --sort 1 > $O/probe_sorted_split1.log 2>&1
tail -n 8 $O/probe_*.log
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 2 --targets 10 --stats 0 > $OLDPWD/$O/trace.log 2>&1 )
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OLDPWD/$O/pmc_sq -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_sq.log 2>&1 )
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
for d in pmc_sq pmc_fetch; do python tools/pmc_summary.py $O/$d --seq k_level > $O/${d}_summary.txt 2>&1; done
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/trace_kernel_stats.csv
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" > $O/trace_level_seq.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "k_level_" in n or "k_grad" in n:
        print("%9.1f us  grid=(%s,%s,%s)  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), n[:70]))
PY
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
