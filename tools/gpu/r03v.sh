#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
run() { # name
  timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 --dump-labels /tmp/$1 > $O/bench_$1.log 2>&1
  python - $1 <<'PY'
import sys, hashlib, json, numpy as np
name = sys.argv[1]
blob = open("/tmp/%s_models.bin" % name, "rb").read()
d = json.loads(open("gpurun_out/r03v/bench_%s.log" % name).read().strip().split("\n")[-1])
print(name, "acc %.10f" % d["repair_accuracy_vs_clean"], "models md5", hashlib.md5(blob).hexdigest(), "labels md5", hashlib.md5(np.load("/tmp/%s_labels.npy" % name).tobytes()).hexdigest())
PY
}
run normal1
export RGBM_POISON=1
run poison1
run poison2
run poison3
