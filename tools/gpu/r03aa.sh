#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03aa; mkdir -p $O
export RGBM_TRACE_K=64 RGBM_TRACE_ITERS=44:44
run() { # name
  n=$1; mkdir -p /tmp/tr_$n
  RGBM_TRACE=/tmp/tr_$n timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 --dump-labels /tmp/$n > $O/bench_$n.log 2>&1
  echo "$n $(md5sum /tmp/${n}_model_10.bin | cut -c1-8) trace $(ls -la /tmp/tr_$n/target10.bin | awk '{print $5}')"
}
run d1; ref=d1
for i in 2 3 4 5 6 7 8 9 10; do
  run d$i
  if ! cmp -s /tmp/${ref}_model_10.bin /tmp/d${i}_model_10.bin; then
    echo "== $ref vs d$i"; python tools/model_diff.py /tmp/$ref /tmp/d$i 16 2>&1 | head -14; python tools/trace_diff.py /tmp/tr_$ref /tmp/tr_d$i 10 64; break
  fi
  rm -rf /tmp/tr_d$i
done
