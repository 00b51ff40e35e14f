"""Timing of a hyper-parameter-search batch (16 trials x 3 folds, 300 iterations, 10 000-row table) through rgbm_table_train_batch
against the same 48 fits as single rgbm_table_train calls (sequential, and 24 host threads)."""
import os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from repair import _native as N

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for target in (4, 10, 0):
    dirty, _, cards = make_table(rows, 16, seed=42, null_ratio=0.01)
    K = int(cards[target]); feats = [c for c in range(16) if c != target]
    tab = N.Table(dirty, cards)
    rng = np.random.RandomState(42)
    fits = []
    for trial in range(16):
        kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=iters, num_leaves=int(rng.randint(2, 101)), bagging_fraction=float(rng.uniform(0.5, 1.0)),
                  bagging_freq=int(rng.randint(1, 21)), feature_fraction=float(rng.uniform(0.01, 1.0)), min_data_in_leaf=int(rng.randint(1, 51)),
                  min_sum_hessian_in_leaf=float(np.exp(rng.uniform(-3, 1))), lambda_l2=float(np.exp(rng.uniform(-2, 3))))
        for r in range(3):
            rws = np.flatnonzero(np.arange(rows) % 3 != r).astype(np.int64)
            fits.append(dict(table=tab.gather_rows(rws), target_col=target, feat_cols=feats, class_weight=balanced_weights(dirty[target][rws], K), **kw))
    def single(f):
        f = dict(f); ft = f.pop("table")
        return ft.train(f.pop("target_col"), f.pop("feat_cols"), class_weight=f.pop("class_weight"), **f)
    N.train_batch(fits[:3]); single(fits[0])                                    # warm-up
    t0 = time.time(); out = N.train_batch(fits); tb = time.time() - t0
    t0 = time.time(); out2 = N.train_batch(fits); tb2 = time.time() - t0
    t0 = time.time(); ref = [single(f) for f in fits]; ts = time.time() - t0
    with ThreadPoolExecutor(24) as ex:
        t0 = time.time(); ref2 = list(ex.map(single, fits)); tt = time.time() - t0
    bad = sum(1 for a, b in zip(out, ref) if not isinstance(a, N.Model) or a.save() != b.save())
    print("target c%d K=%d, 48 fits x %d iterations on %d rows: batch %.3f s (again %.3f s) | 48 single calls %.3f s | 24 threads %.3f s | differing models %d" % (target, K, iters, rows, tb, tb2, ts, tt, bad), flush=True)
