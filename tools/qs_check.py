"""Debug aid: bit-vector predictor (k_predict_qs) against the index-linked walk (RGBM_PREDICTOR=walk, separate process) on the bench workload's
dirty rows: first differing (row, class) of the raw probabilities."""
import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from repair import _native as N
rows, t, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dirty, clean, cards = make_table(rows, 16, seed=42)
feats = [c for c in range(16) if c != t]
r = dirty[t] >= 0
K = int(cards[t]); cw = balanced_weights(dirty[t], K)
m = N.train(np.ascontiguousarray(dirty[feats][:, r]), cards[feats], dirty[t][r], K, class_weight=cw, objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=iters)
X = np.ascontiguousarray(dirty[feats])
p = m.predict(X)
tag = os.environ.get("RGBM_PREDICTOR", "qs")
np.save("/tmp/qs_check_%s.npy" % tag, p)
if tag == "qs":
    subprocess.check_call([sys.executable] + sys.argv, env=dict(os.environ, RGBM_PREDICTOR="walk"))
    q = np.load("/tmp/qs_check_walk.npy")
    d = np.flatnonzero((p != q).any(axis=1))
    print("rows with different probabilities: %d of %d" % (len(d), len(p)))
    for i in d[:5]:
        print("row", i, "codes", X[:, i].tolist(), "qs", p[i][:6], "walk", q[i][:6])
