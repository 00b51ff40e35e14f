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
kws = [dict(num_leaves=31), dict(num_leaves=7, min_data_in_leaf=40), dict(num_leaves=63, lambda_l2=2.0), dict(num_leaves=15, feature_fraction=0.5),
       dict(num_leaves=31, bagging_fraction=0.7, bagging_freq=3), dict(num_leaves=20, bagging_fraction=0.9, bagging_freq=1, feature_fraction=0.6)]
NK = len(kws); NT = 24; ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
import time
genv = None
t0 = time.perf_counter()
ref = [N.train(X, cards[feats], y, K, class_weight=cw, objective=1, num_class=K, n_estimators=60, **kw).save() for kw in kws]
t_seq_plain = (time.perf_counter() - t0) / NK
t0 = time.perf_counter()
seq = [N.train(X, cards[feats], y, K, class_weight=cw, objective=1, num_class=K, n_estimators=60, **kw).save() for kw in kws]
t_seq = (time.perf_counter() - t0) / NK
bad = sum(a != b for a, b in zip(seq, ref))
t0 = time.perf_counter()
for rnd in range(ROUNDS):
    out = [None] * NT
    def work(i):
        out[i] = N.train(X, cards[feats], y, K, class_weight=cw, objective=1, num_class=K, n_estimators=60, **kws[i % NK]).save()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
    [t.start() for t in ths]; [t.join() for t in ths]
    bad += sum(out[i] != ref[i % NK] for i in range(NT))
t_conc = (time.perf_counter() - t0) / (ROUNDS * NT)
print("mismatches: %d of %d; per fit (60 iterations): first sequential pass %.1f ms, second %.1f ms, %d threads %.1f ms"
      % (bad, NK + ROUNDS * NT, 1e3 * t_seq_plain, 1e3 * t_seq, NT, 1e3 * t_conc))
