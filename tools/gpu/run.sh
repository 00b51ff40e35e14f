#!/bin/bash
# The GPU-box runs of this repo, one parameterised script (the ~100 one-off rNN*.sh scripts of rounds 2-4 are gone; their outputs
# live under profiles/).  Usage, from the repo root, through gpurun:   gpurun --timeout N -- 'bash tools/gpu/run.sh <task> <tag> [args...]'
#   suite  <tag> [pytest args]        the -m gpu tests (default: all of tests/)
#   ab     <tag> VAR v1 v2 ... -- <command>   the command once per value of the environment variable VAR (A/B of a runtime switch / RGBM_LIB_PATH)
#   levels <tag> <target> [VAR=val ...]       per-launch durations of k_level_root / k_level_mt of one target (rocprofv3 kernel trace of tools/probe.py)
#   bench  <tag> [bench.py args]      one bench line -> gpurun_out/<tag>/bench.json
#   final  <tag>                      what a round ends with: suite, smoke, the driver's bench line, rocprofv3 kernel stats of the same command (one target
#                                     at a time), PMC traffic of both configs -> profiles/traffic.json, 100M x 32 on one GPU (20 steps + the complete job)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
task=$1; tag=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
J() { grep '^{"metric' "$1" | tail -1 > "$2"; }
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("%s: ms_per_step %.1f value %.0f elapsed %.2fs frac %.4f frac_needed %s md5 %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["elapsed_sec"], r.get("frac", 0), r.get("frac_needed"), d.get("models_md5")))
PY
}
case $task in
suite)
  ( time timeout 1500 python -m pytest ${@:-tests} -q -m gpu --durations=6 ) > $O/tests_gpu_full.log 2>&1
  grep -E "passed|failed|error" $O/tests_gpu_full.log | tail -3 | tee $O/tests_gpu.log; grep -E "^(FAILED|ERROR)|Error" $O/tests_gpu_full.log | head -10 ;;
ab)
  var=$1; shift; vals=(); while [ "$1" != "--" ]; do vals+=("$1"); shift; done; shift
  for v in "${vals[@]}"; do echo "== $var=$v" | tee -a $O/ab.txt; env "$var=$v" timeout 900 "$@" 2>&1 | tail -12 | tee -a $O/ab.txt; done ;;
levels)
  bash tools/gpu/levels.sh $O/levels.txt "$@"; cat $O/levels.txt ;;
bench)
  timeout 1500 python bench.py "$@" > $O/bench.log 2>&1; J $O/bench.log $O/bench.json; show $O/bench.json ;;
final)
  ( time timeout 1500 python -m pytest tests -q -m gpu --durations=6 ) > $O/tests_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/tests_gpu_full.log | tail -3 | tee $O/tests_gpu.log
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tail -1 | tee $O/smoke.log
  # (the PMC passes first: the bench line attaches profiles/traffic.json, which must describe THIS build)
  ALL=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
  for c in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/tools/probe.py --iters 1 --targets $ALL --stats 0 > $O/pmc_$c.log 2>&1 ); done
  python tools/make_traffic_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $tag > $O/traffic_json.log 2>&1; tail -4 $O/traffic_json.log
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; J $O/bench_default.log $O/bench_steps20_warmup5.json; show $O/bench_steps20_warmup5.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_seq -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-job --concurrency 1 --roofline-steps 1 > $O/trace_seq.log 2>&1 )
  f=$(find $O/trace_seq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_steps10_seq_kernel_stats.csv; J $O/trace_seq.log $O/bench_steps10_seq.json
  T8=0,1,2,3,4,5,6,7
  for c in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc32_$c -- python $GRAFT_REPO_ROOT/tools/probe.py --rows 100000000 --cols 32 --seed 43 --parallel 1 --iters 1 --targets $T8 --stats 0 > $O/pmc32_$c.log 2>&1 ); done
  python tools/make_traffic_json.py $O/pmc32_FETCH_SIZE $O/pmc32_WRITE_SIZE ${tag}_100m32 --rows 100000000 --cols 32 --targets $T8 > $O/traffic32_json.log 2>&1; tail -4 $O/traffic32_json.log
  cp profiles/traffic.json $O/traffic.json; cp profiles/${tag}*_hbm_traffic_pmc.txt $O/ 2>/dev/null
  timeout 1200 python bench.py --config 100m32 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_100m32.log 2>&1; J $O/bench_100m32.log $O/bench_100m32_steps20_and_complete_job.json; show $O/bench_100m32_steps20_and_complete_job.json
  BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-full-job --no-cpu-baseline --roofline-steps 1 > $O/bench_2ranks.log 2>&1; J $O/bench_2ranks.log $O/bench_2ranks_on_one_gpu_gloo_steps5.json; show $O/bench_2ranks_on_one_gpu_gloo_steps5.json
  timeout 600 python bench.py --train-rows 10000 --no-cpu-baseline > $O/bench_train_rows_10000.log 2>&1; J $O/bench_train_rows_10000.log $O/bench_train_rows_10000.json; show $O/bench_train_rows_10000.json
  find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete; rm -rf $O/pmc_* $O/pmc32_* $O/trace_seq ;;
*) echo "unknown task $task"; exit 2 ;;
esac
