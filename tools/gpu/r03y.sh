#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
export RGBM_GUARD=1
run() { # name, extra args
  n=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 --dump-labels /tmp/$n "$@" > $O/bench_$n.log 2>&1
  echo "$n $(md5sum /tmp/${n}_model_10.bin | cut -c1-8) acc $(grep '^{' $O/bench_$n.log | tail -1 | python -c 'import sys,json; print("%.10f" % json.loads(sys.stdin.read())["repair_accuracy_vs_clean"])') guard lines: $(grep -c 'rgbm guard' $O/bench_$n.log)"
  grep 'rgbm guard' $O/bench_$n.log | sort | uniq -c | sort -rn | head -8
}
for i in 1 2 3 4 5; do run g$i; done
