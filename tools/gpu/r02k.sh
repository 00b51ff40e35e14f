set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
unset RGBM_LEVEL_SPLIT
( time timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k hp_search -s --durations=3 ) > $O/t_hp.log 2>&1; grep -E "hp search|passed|failed|real" $O/t_hp.log
timeout 600 python -m pytest tests/test_gpu_growers.py tests/test_gpu_rowshard.py tests/test_gpu_parity.py -x -q -m gpu --deselect tests/test_gpu_parity.py::test_build_model_hp_search_runs_on_gpu_and_matches_oracle_backend 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
for b in default 8 16; do
  if [ $b = default ]; then unset RGBM_LV_BLOCKS; else export RGBM_LV_BLOCKS=$b; fi
  timeout 300 python tools/probe.py --iters 5 --targets 0,4,7,10 > $O/probe_b$b.log 2>&1; echo "blocks=$b"; grep target $O/probe_b$b.log | awk 'NR%2==0'
done
unset RGBM_LV_BLOCKS
RGBM_LIB_PATH=$PWD/spark-data-repair-plugin_amd/lib/librepairgbm_norot.so timeout 300 python tools/probe.py --iters 5 --targets 7,10 > $O/probe_norot.log 2>&1; echo norot; grep target $O/probe_norot.log | awk 'NR%2==0'
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
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
