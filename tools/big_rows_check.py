"""One-off check of the 64-bit row arithmetic: train on >= 2^26 rows (more than 2^22 rows per workgroup, several carry-word
ranges, node-id buffers beyond 4 GB for K = 12) and compare the model bytes with the CPU oracle.
    python tools/big_rows_check.py [--rows 100000000]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spark-data-repair-plugin_amd")]
from tests.synth import make_table, balanced_weights
from repair import _native as N
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--oracle", type=int, default=1)
a = ap.parse_args()
t0 = time.time()
dirty, clean, cards = make_table(a.rows, 6, seed=9, null_ratio=0.01, cards=[2, 12, 8, 48, 6, 3])
print("gen %.1fs" % (time.time() - t0), flush=True)
tab = N.Table(dirty, cards)
for tgt, iters in ((0, 3), (1, 2)):
    feats = [c for c in range(6) if c != tgt]
    K = int(cards[tgt])
    cw = balanced_weights(dirty[tgt], K)
    kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=iters, learning_rate=0.3)
    t0 = time.time(); mg = tab.train(tgt, feats, class_weight=cw, **kw); tg = time.time() - t0
    if a.oracle:
        rows = dirty[tgt] >= 0
        t0 = time.time()
        mo = O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[tgt][rows], K, class_weight=cw, **kw)
        to = time.time() - t0
        print("target %d K=%d rows=%d: gpu %.2fs oracle %.1fs  identical=%s" % (tgt, K, int(rows.sum()), tg, to, mo.save() == mg.save()), flush=True)
    else:
        print("target %d K=%d: gpu %.2fs" % (tgt, K, tg), flush=True)
