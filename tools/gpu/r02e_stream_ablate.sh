set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
SEQ='
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "k_level_" in n or "k_grad" in n:
        print("%9.1f us  grid=(%s,%s,%s)  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), n[:70]))
'
( cd /tmp && RGBM_LEVEL_SPLIT=1 RGBM_DBG_STREAM=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python -c "$SEQ" "$f" > $O/trace_level_seq.txt
grep -E "k_level_pass|k_level_route" $O/trace_level_seq.txt | head -40
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OLDPWD/$O/pmc_sq2 -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_sq2.log 2>&1 )
( cd /tmp && RGBM_LEVEL_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$O/pmc_tcc -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_tcc.log 2>&1 )
for d in pmc_sq2 pmc_tcc; do python tools/pmc_summary.py $O/$d > $O/${d}_summary.txt 2>&1; grep -E "k_level_pass|k_level_route|k_level_final|k_grad_mc" $O/${d}_summary.txt; done
timeout 900 python bench.py --steps 20 --warmup 2 > $O/bench20.log 2>&1; tail -1 $O/bench20.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
