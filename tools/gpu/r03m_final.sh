#!/bin/bash
# round 3: measurements at HEAD -- GPU suite, smoke, bench line, kernel stats, PMC traffic over all targets, counters, reference-default job,
# 100M x 32 on one GPU, RepairModel.run() on a 1M-row frame
set -x
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) 2>&1 | tail -22 > $O/tests_gpu.log; tail -4 $O/tests_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
timeout 1200 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; cut -c1-600 $O/bench_default.json
# kernel stats: one target model at a time (the roofline pass's conditions) and the default six in flight
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_seq -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $OLDPWD/$O/trace_seq.log 2>&1 )
f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; tail -1 $O/trace_seq.log > $O/bench_steps10_seq.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_def -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --roofline-steps 1 > $OLDPWD/$O/trace_def.log 2>&1 )
f=$(find $O/trace_def -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_kernel_stats.csv
# HBM traffic of every target model (FETCH_SIZE / WRITE_SIZE in separate passes)
ALL=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets $ALL --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python $OLDPWD/tools/probe.py --iters 1 --targets $ALL --stats 0 > $OLDPWD/$O/pmc_write.log 2>&1 )
python tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write r03m > $O/traffic_json.log 2>&1; tail -30 $O/traffic_json.log; cp profiles/traffic.json $O/traffic.json; cp profiles/r03m_hbm_traffic_pmc.txt $O/ 2>/dev/null
# counters of the K = 64 target at HEAD
run_pmc() { local name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OLDPWD/$O/pmc_$name -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_$name.log 2>&1 ); python tools/pmc_summary.py $O/pmc_$name --seq k_level_ > $O/pmc_${name}_summary.txt 2>&1; }
run_pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR
run_pmc tcc TCC_HIT_sum TCC_MISS_sum
for n in sq1 sq2 tcc; do echo "== $n"; grep -A400 "# per dispatch" $O/pmc_${n}_summary.txt | grep -E "k_level_mt|k_level_root" | tail -64; done > $O/pmc_per_dispatch.txt
f=$(find $O/pmc_tcc -name "*kernel_trace.csv" | head -1); python - "$f" > $O/k64_level_durations.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
print("root + levels 1..6 (us):", " ".join("%7.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in [r for r in rows if "k_level_mt" in r["Kernel_Name"] or "k_level_root" in r["Kernel_Name"]][-7:]))
PY
cat $O/k64_level_durations.txt
timeout 600 python tools/probe.py --iters 5 --targets 0,4,7,10 2>&1 | grep "^target" | tee $O/probe.log
# the reference's default job (10 000 training rows per model), the complete 300 iterations
timeout 600 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; tail -1 $O/bench_train_rows_10000.log > $O/bench_train_rows_10000.json; cut -c1-400 $O/bench_train_rows_10000.json
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 2>&1 | grep "^run" | tee $O/resident_probe.log
timeout 300 python tools/resident_probe.py --rows 1000000 --cols 8 --estimators 300 --categorical 2>&1 | grep "^run" | tee -a $O/resident_probe.log
# configs[3] on one GPU: 100M x 32, 20 boosting iterations (the complete job takes minutes; its per-step time is what the scaling curve starts from)
timeout 1500 python bench.py --config 100m32 --steps 20 --no-cpu-baseline --no-full-job > $O/bench_100m32_steps20.log 2>&1; tail -1 $O/bench_100m32_steps20.log > $O/bench_100m32_steps20.json; cut -c1-500 $O/bench_100m32_steps20.json
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
