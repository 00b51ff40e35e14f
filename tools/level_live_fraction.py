"""Which share of the (row, class tree) pairs a level pass streams is LIVE (the row sits in a node that is split at this level) and BUILT
(it falls into the smaller child, whose histogram is accumulated)?  From the trees of the oracle on a sample of the synthetic bench table:
python tools/level_live_fraction.py [rows] [iterations] [target ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from oracle import oracle as O
from repair.synth import make_table
from repair.engine import balanced_class_weight
from tests.numerics_bound import parse_trees

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
targets = [int(a) for a in sys.argv[3:]] or [10, 4, 1]
dirty, clean, cards = make_table(rows, 16, seed=42)
for t in targets:
    feats = [c for c in range(16) if c != t]
    r = dirty[t] >= 0
    K = int(cards[t])
    cw = balanced_class_weight(np.bincount(dirty[t][r], minlength=K))
    m = O.train(np.ascontiguousarray(dirty[feats][:, r]), cards[feats], dirty[t][r], K, class_weight=cw, objective=0 if K <= 2 else 1, num_class=max(K, 2),
                n_estimators=iters, num_leaves=31, max_depth=7, learning_rate=0.01, min_data_in_leaf=20, lambda_l2=0.0, seed=42)
    _, n_it, trees = parse_trees(m.save())
    n_train = int(r.sum())
    live = np.zeros(8); built = np.zeros(8); ntree = 0
    for tr in trees:
        L = len(tr["leaf_count"])
        if L < 2:
            continue
        ntree += 1
        cnt = {}
        def count(ref):
            if ref < 0:
                return int(tr["leaf_count"][~ref])
            c = count(int(tr["left"][ref])) + count(int(tr["right"][ref]))
            cnt[ref] = c
            return c
        count(0)
        def walk(ref, d):
            if ref < 0:
                return
            l, rr = int(tr["left"][ref]), int(tr["right"][ref])
            cl = cnt[l] if l >= 0 else int(tr["leaf_count"][~l]); cr = cnt[rr] if rr >= 0 else int(tr["leaf_count"][~rr])
            live[d] += cnt[ref]; built[d] += min(cl, cr)
            walk(l, d + 1); walk(rr, d + 1)
        walk(0, 0)
    tot = ntree * n_train
    print("target c%d K=%d: %d trees of %d iterations, %d training rows" % (t, K, ntree, n_it, n_train))
    print("  level pass      : " + " ".join("%6d" % (d + 1) for d in range(7)))
    print("  live share      : " + " ".join("%6.3f" % (live[d] / tot) for d in range(7)))
    print("  built share     : " + " ".join("%6.3f" % (built[d] / tot) for d in range(7)))
    print("  sum over levels : live %.2f, built %.2f pairs per (row, class tree) and iteration (of 6-7 streamed)" % (live[:7].sum() / tot, built[:7].sum() / tot))
