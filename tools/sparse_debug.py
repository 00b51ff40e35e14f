"""Debug aid: the level grower with and without the sparse sweep (RGBM_MT_SPARSE) must give the same model; prints the first tree that differs."""
import os, sys, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from tests.numerics_bound import parse_trees
from repair import _native as N

for rows, cols, tgt in ((600, 4, 2), (5000, 6, 3), (40000, 8, 5)):
    dirty, clean, cards = make_table(rows, cols, seed=29)
    feats = [c for c in range(cols) if c != tgt]
    r = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, r]); y = dirty[tgt][r]; K = int(cards[tgt])
    for bag, nl, mdl in itertools.product((None, (0.7, 1)), (31, 60), (1, 20)):
        kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=12, learning_rate=0.2, num_leaves=nl, min_data_in_leaf=mdl, max_depth=7)
        if bag:
            kw.update(bagging_fraction=bag[0], bagging_freq=bag[1])
        blobs = []
        for v in ("0", "1"):
            os.environ["RGBM_MT_SPARSE"] = v
            blobs.append(N.train(X, cards[feats], y, K, class_weight=balanced_weights(y, K), **kw).save())
        if blobs[0] == blobs[1]:
            print("rows %d K %d bag %s leaves %d mdl %d: same" % (rows, K, bag, nl, mdl)); continue
        Kt, n_it, ta = parse_trees(blobs[0]); _, _, tb = parse_trees(blobs[1])
        for i, (a, b) in enumerate(zip(ta, tb)):
            same = len(a["feat"]) == len(b["feat"]) and all(np.array_equal(a[n], b[n]) for n in ("feat", "theta", "left", "right", "leaf_value", "leaf_count"))
            if not same:
                print("rows %d K %d bag %s leaves %d mdl %d: DIFFER at iteration %d class tree %d: leaves %d / %d; leaf_count sum %d / %d; first counts %s / %s" % (
                    rows, K, bag, nl, mdl, i // Kt, i % Kt, len(a["leaf_count"]), len(b["leaf_count"]), a["leaf_count"].sum(), b["leaf_count"].sum(),
                    a["leaf_count"][:10].tolist(), b["leaf_count"][:10].tolist()))
                n = min(len(a["feat"]), len(b["feat"]))
                d = [j for j in range(n) if a["feat"][j] != b["feat"][j] or a["theta"][j] != b["theta"][j] or a["gain"][j] != b["gain"][j]]
                print("   first differing node %s" % (d[:3],), [(int(a["feat"][j]), int(a["theta"][j]), float(a["gain"][j]), int(b["feat"][j]), int(b["theta"][j]), float(b["gain"][j])) for j in d[:2]])
                break
