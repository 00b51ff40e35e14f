"""Confirms that RGBM_LV_DRAIN_SHIFT really forces packed-slot drains: the same fit gets slower as the budget shrinks,
and the model bytes stay the same.  python tools/drain_probe.py  (GPU box)"""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from repair import _native as N

dirty, clean, cards = make_table(2_000_000, 8, seed=5)
for tgt in (0, 5):
    feats = [c for c in range(8) if c != tgt]
    rows = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, rows]); y = dirty[tgt][rows]; K = int(cards[tgt])
    kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=6, learning_rate=0.3)
    cw = balanced_weights(y, K)
    for shift in (None, 0, 4, 8, 12, 16):
        if shift is None:
            os.environ.pop("RGBM_LV_DRAIN_SHIFT", None)
        else:
            os.environ["RGBM_LV_DRAIN_SHIFT"] = str(shift)
        t = time.time()
        m = N.train(X, cards[feats], y, K, class_weight=cw, **kw)
        dt = time.time() - t
        print("target %d K=%d shift=%s: %.3f s  model %s" % (tgt, K, shift, dt, hashlib.sha1(m.save()).hexdigest()[:12]), flush=True)
