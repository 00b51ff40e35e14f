set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py -x -q -m gpu 2>&1 | tail -6 > $O/tests.log; cat $O/tests.log
timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_blocks.log 2>&1; grep target $O/probe_blocks.log | awk 'NR%2==0'
RGBM_LV_BLOCK_TILES=32 timeout 300 python tools/probe.py --iters 5 --targets 7,10 > $O/probe_blocks32.log 2>&1; grep target $O/probe_blocks32.log | awk 'NR%2==0'
RGBM_LV_BLOCK_TILES=512 timeout 300 python tools/probe.py --iters 5 --targets 7,10 > $O/probe_blocks512.log 2>&1; grep target $O/probe_blocks512.log | awk 'NR%2==0'
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
python tools/pmc_summary.py $O/pmc_fetch --seq k_level_pass > $O/pmc_fetch_summary.txt 2>&1; grep -E "k_level_pass|k_level_route" $O/pmc_fetch_summary.txt | head -12
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY' | tee $O/trace_level_seq.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "k_level_pass" in n or "k_level_route" in n or "k_level_final" in n or "k_grad" in n:
        print("%9.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n[:60]))
PY
timeout 600 python tools/hp_test_timing.py 2>&1 | grep -E "backend" | tee $O/hp_timing.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
