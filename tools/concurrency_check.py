"""Train the same model from many host threads at once (one HIP stream each): every result must be bit-identical."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from repair import _native as N
dirty, clean, cards = make_table(10000, 8, seed=5)
tgt = 4; feats = [c for c in range(8) if c != tgt]; r = dirty[tgt] >= 0
X = np.ascontiguousarray(dirty[feats][:, r]); y = dirty[tgt][r]; K = int(cards[tgt]); cw = balanced_weights(y, K)
kws = [dict(num_leaves=31), dict(num_leaves=7, min_data_in_leaf=40), dict(num_leaves=63, lambda_l2=2.0), dict(num_leaves=15, feature_fraction=0.5)]
ref = [N.train(X, cards[feats], y, K, class_weight=cw, objective=1, num_class=K, n_estimators=60, **kw).save() for kw in kws]
bad = 0
for rnd in range(3):
    out = [None] * 24
    def work(i):
        out[i] = N.train(X, cards[feats], y, K, class_weight=cw, objective=1, num_class=K, n_estimators=60, **kws[i % 4]).save()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(24)]
    [t.start() for t in ths]; [t.join() for t in ths]
    bad += sum(out[i] != ref[i % 4] for i in range(24))
print("graph=%s mismatches: %d of 72" % (os.environ.get("RGBM_NO_GRAPH") is None, bad))
