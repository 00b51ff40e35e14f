#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
run() { # name, extra args
  n=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 --dump-labels /tmp/$n "$@" > $O/bench_$n.log 2>&1
  echo "$n $(md5sum /tmp/${n}_model_10.bin | cut -c1-8) acc $(tail -1 $O/bench_$n.log | python -c 'import sys,json; print("%.10f" % json.loads(sys.stdin.read())["repair_accuracy_vs_clean"])')"
}
for i in 1 2 3 4; do run d$i; done
for i in 1 2 3; do run c$i --concurrency 1; done
ref=d1
for n in d2 d3 d4 c1 c2 c3; do
  if ! cmp -s /tmp/${ref}_model_10.bin /tmp/${n}_model_10.bin; then echo "== $ref vs $n"; python tools/model_diff.py /tmp/$ref /tmp/$n 16; break; fi
done
