set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
# 1. split mode after the prefetch fix, all targets of interest; then the two ablations
RGBM_LEVEL_SPLIT=1 timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_split1.log 2>&1
RGBM_LEVEL_SPLIT=1 RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm_noatom.so timeout 300 python tools/probe.py --iters 5 --targets 10 > $O/probe_split1_noatom.log 2>&1
RGBM_LEVEL_SPLIT=1 RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm_noappend.so timeout 300 python tools/probe.py --iters 5 --targets 10 > $O/probe_split1_noappend.log 2>&1
RGBM_LEVEL_SPLIT=0 RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm_noatom.so timeout 300 python tools/probe.py --iters 5 --targets 10 > $O/probe_split0_noatom.log 2>&1
# 2. rows clustered on the host (what a device-side sort would give)
RGBM_LEVEL_SPLIT=0 timeout 400 python tools/probe.py --iters 5 --targets 4,7,10 --sort 1 > $O/probe_sorted_split0.log 2>&1
RGBM_LEVEL_SPLIT=1 timeout 400 python tools/probe.py --iters 5 --targets 4,7,10 --sort 1 > $O/probe_sorted_split1.log 2>&1
RGBM_LEVEL_SPLIT=1 RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm_noatom.so timeout 400 python tools/probe.py --iters 5 --targets 10 --sort 1 > $O/probe_sorted_split1_noatom.log 2>&1
tail -n 3 $O/probe_*.log
# 3. per-kernel times and counters of the K=64 target in split mode
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 2 --targets 10 --stats 0 > $OLDPWD/$O/trace.log 2>&1 )
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OLDPWD/$O/pmc_sq -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_sq.log 2>&1 )
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_write.log 2>&1 )
find $O -name "*.csv" | head -30
for d in pmc_sq pmc_fetch pmc_write; do python tools/pmc_summary.py $O/$d --seq k_level > $O/${d}_summary.txt 2>&1; done
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
# keep the merged output small
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O
