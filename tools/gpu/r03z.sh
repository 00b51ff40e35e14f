#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
run() { # name, extra args
  n=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 --dump-labels /tmp/$n "$@" > $O/bench_$n.log 2>&1
  echo "$n $(md5sum /tmp/${n}_model_10.bin | cut -c1-8)"
}
run d1; ref=d1
for i in 2 3 4 5 6 7; do
  run d$i
  if ! cmp -s /tmp/${ref}_model_10.bin /tmp/d${i}_model_10.bin; then echo "== $ref vs d$i"; python tools/model_diff.py /tmp/$ref /tmp/d$i 16; break; fi
done
