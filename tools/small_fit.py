"""Latency probe: the reference's default shape -- a 10 000-row training sample, 300 boosting iterations (train.py:53-55,
model.py:755-766) -- through the host-array entry point."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from repair import _native as N
for rows, cols, tgt in [(10000, 8, 4), (10000, 16, 10), (1000, 16, 4)]:
    dirty, clean, cards = make_table(rows, cols, seed=5)
    feats = [c for c in range(cols) if c != tgt]
    r = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, r]); y = dirty[tgt][r]; K = int(cards[tgt])
    cw = balanced_weights(y, K)
    for rep in range(3):
        t0 = time.perf_counter()
        m = N.train(X, cards[feats], y, K, class_weight=cw, objective=1, num_class=K, n_estimators=300)
        dt = time.perf_counter() - t0
    print("rows=%d cols=%d K=%d: fit 300 iterations %.1f ms (%.1f us/iteration), n_iter=%d" % (rows, cols, K, dt * 1e3, dt * 1e6 / 300, m.info()["n_iter"]), flush=True)
