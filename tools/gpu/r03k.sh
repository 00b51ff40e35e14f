#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_growers.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/probe.py --iters 3 --targets 0,4,7,10 2>&1 | grep "^target" | awk 'NR%2==0' | tee $O/probe.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/tr -- python $OLDPWD/tools/probe.py --iters 2 --targets 10 --stats 0 > $OLDPWD/$O/tr.log 2>&1 )
f=$(find $O/tr -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%7.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]][-7:]))
PY
timeout 900 python bench.py --steps 20 --no-cpu-baseline --no-full-job > $O/bench_steps20.log 2>&1; tail -1 $O/bench_steps20.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'root GB/s', d['roofline']['root_scan_GBps_rank0'])"
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
