#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03t; mkdir -p $O
for i in 1 2 3 4; do
  timeout 900 python bench.py --no-cpu-baseline --roofline-steps 1 --dump-labels /tmp/run$i > $O/bench_$i.log 2>&1
  tail -1 $O/bench_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i acc %.10f' % d['repair_accuracy_vs_clean'])"
done
python - <<'PY' | tee $O/diff.txt
import numpy as np, hashlib
L = [np.load("/tmp/run%d_labels.npy" % i) for i in (1, 2, 3, 4)]
P = [np.load("/tmp/run%d_probs.npy" % i) for i in (1, 2, 3, 4)]
M = [hashlib.md5(open("/tmp/run%d_models.bin" % i, "rb").read()).hexdigest() for i in (1, 2, 3, 4)]
print("model digests", M)
for i in range(1, 4):
    d = np.argwhere(L[0] != L[i]); dp = np.argwhere(P[0] != P[i])
    print("run 1 vs run %d: labels differ at %d places, probabilities at %d" % (i + 1, len(d), len(dp)), d[:5].tolist(), dp[:5].tolist())
    for t, r in dp[:5]:
        print("   target %d row %d: labels %d / %d, probs %.17g / %.17g" % (t, r, L[0][t, r], L[i][t, r], P[0][t, r], P[i][t, r]))
PY
