"""CPU: unit checks of the oracle's restated LightGBM semantics (numerics spec of DESIGN.md)."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from tests.synth import balanced_weights, make_table


def test_exp_close_to_libm():
    xs = np.concatenate([np.linspace(-745, 709, 4001), np.random.default_rng(0).normal(size=2000) * 5, [0.0, -0.0, 1e-300]])
    for x in xs:
        got, ref = O.lib().orc_exp(float(x)), math.exp(x) if x < 709.7 else float("inf")
        if ref == 0.0 or math.isinf(ref):
            continue
        assert abs(got - ref) <= 4e-16 * ref + 5e-324, (x, got, ref)
    assert O.lib().orc_exp(800.0) == float("inf") and O.lib().orc_exp(-800.0) == 0.0


def _bins(col, n_codes, **kw):
    import ctypes as C
    col = np.ascontiguousarray(col, np.int32)
    p = O.make_params(**kw)
    V, nan = C.c_int32(), C.c_int32()
    ub = np.zeros(512, np.int32)
    O.lib().orc_find_bin(col.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(len(col)), C.c_int32(n_codes), C.byref(p),
                         C.byref(V), C.byref(nan), ub.ctypes.data_as(C.POINTER(C.c_int32)))
    return V.value, nan.value, ub[:V.value].tolist()


def test_binning_small_domain_one_bin_per_code():
    col = np.repeat(np.arange(5), 10)
    V, nan, ub = _bins(col, 5)
    assert (V, nan) == (5, 0) and ub[:-1] == [0, 1, 2, 3] and ub[-1] == 2**31 - 1


def test_binning_min_data_in_bin_merges_rare_codes():
    col = np.array([0] * 10 + [1] * 1 + [2] * 1 + [3] * 10)   # codes 1,2 have < 3 rows
    V, nan, ub = _bins(col, 4)
    assert V == 2 and ub == [0, 2**31 - 1]   # rare codes 1,2 (< min_data_in_bin rows) share the last bin with 3


def test_binning_nan_bin_and_unseen_codes():
    col = np.array([0] * 5 + [4] * 5 + [-1] * 3)
    V, nan, ub = _bins(col, 6)
    assert nan == 1 and V == 2 and ub[0] == 2      # boundary = floor((0+4)/2): unseen 1,2 -> left bin, 3 -> right


def test_binning_more_codes_than_max_bin():
    rng = np.random.default_rng(1)
    col = rng.integers(0, 1000, 50000)
    V, nan, ub = _bins(col, 1000, max_bin=63)
    assert V <= 62 and nan == 0 and ub == sorted(ub) and len(set(ub)) == len(ub)
    V2, _, _ = _bins(np.concatenate([col, [-1] * 10]), 1000, max_bin=63)
    assert V2 <= 61


def test_constant_model_when_no_split_possible():
    X = np.array([[0, 1, 0, 1, 1, 0]], np.int32); y = np.array([0, 1, 0, 1, 1, 1], np.int32)
    m = O.train(X, [2], y, 2, objective=0, n_estimators=10)
    assert m.info()["n_iter"] == 1                      # LightGBM stops: "no more leaves that meet the split requirements"
    p = m.predict(X)
    assert np.allclose(p[:, 1], 4 / 6) and np.all(p == p[0])   # BoostFromScore prior


def test_training_fits_and_is_deterministic():
    dirty, clean, cards = make_table(6000, 6, seed=3)
    X = np.ascontiguousarray(dirty[:5]); y = clean[5]; K = int(cards[5])
    kw = dict(objective=1, num_class=K, n_estimators=15, learning_rate=0.2, class_weight=balanced_weights(y, K))
    a, b = O.train(X, cards[:5], y, K, **kw), O.train(X, cards[:5], y, K, **kw)
    assert a.save() == b.save()
    assert (a.predict(X).argmax(1) == y).mean() > 0.6
    assert np.allclose(a.predict(X).sum(1), 1.0)


def test_row_order_invariance():
    """Integer histograms make the model independent of the row order."""
    dirty, clean, cards = make_table(3000, 5, seed=4)
    X = np.ascontiguousarray(dirty[:4]); y = clean[4]; K = int(cards[4])
    perm = np.random.default_rng(0).permutation(X.shape[1])
    kw = dict(objective=1, num_class=K, n_estimators=8, learning_rate=0.3, min_data_in_leaf=5)
    assert O.train(X, cards[:4], y, K, **kw).save() == O.train(np.ascontiguousarray(X[:, perm]), cards[:4], y[perm], K, **kw).save()


def test_save_load_roundtrip_predicts_identically():
    dirty, clean, cards = make_table(2000, 5, seed=5)
    X = np.ascontiguousarray(dirty[:4]); y = clean[4] % 2
    m = O.train(X, cards[:4], y, 2, objective=0, n_estimators=10, learning_rate=0.3, min_data_in_leaf=5)
    m2 = O.OracleModel.load(m.save())
    assert m2.save() == m.save() and np.array_equal(m.predict(X), m2.predict(X))


def test_regression_l2_learns_and_uses_weights():
    rng = np.random.default_rng(6)
    X = rng.integers(0, 8, (3, 4000)).astype(np.int32)
    vals = np.linspace(-2, 2, 16)
    y = ((X[0] + X[1]) % 16).astype(np.int32)
    m = O.train(X, [8, 8, 8], y, 16, y_value=vals, objective=2, n_estimators=60, learning_rate=0.2, min_data_in_leaf=5)
    pred = m.predict(X)[:, 0]
    assert np.mean((pred - vals[y]) ** 2) < 0.1 * np.var(vals[y])


def test_bagging_and_feature_fraction_paths_run():
    dirty, clean, cards = make_table(3000, 6, seed=7)
    X = np.ascontiguousarray(dirty[:5]); y = clean[5] % 3
    m = O.train(X, cards[:5], y, 3, objective=1, num_class=3, n_estimators=10, learning_rate=0.3, min_data_in_leaf=5,
                bagging_fraction=0.7, bagging_freq=2, feature_fraction=0.6)
    assert m.info()["n_iter"] == 10 and (m.predict(X).argmax(1) == y).mean() > 0.5


def test_multiclass_accuracy_next_to_sklearn_hist_gradient_boosting():
    """Sanity of the multiclass / class-weight path next to an independent implementation (accuracy only: sklearn's softmax
    gradients are float32 and its class trees use a different hessian factor, so trees cannot be compared node by node here).
    The node-by-node pin of the tree growth lives in tests/test_oracle_split_pin.py."""
    from sklearn.ensemble import HistGradientBoostingClassifier
    dirty, clean, cards = make_table(8000, 8, seed=8)
    tgt = 5; feats = [c for c in range(8) if c != tgt]
    rows = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, rows]); y = dirty[tgt][rows]; K = int(cards[tgt])
    m = O.train(X, cards[feats], y, K, objective=1, num_class=K, n_estimators=40, learning_rate=0.1, class_weight=balanced_weights(y, K))
    Xs = X.T.astype(float); Xs[Xs < 0] = np.nan
    h = HistGradientBoostingClassifier(max_iter=40, learning_rate=0.1, max_depth=7, max_leaf_nodes=31, class_weight="balanced",
                                       early_stopping=False).fit(Xs, y)
    nul = ~rows
    Xd = np.ascontiguousarray(dirty[feats][:, nul]); Xds = Xd.T.astype(float); Xds[Xds < 0] = np.nan
    acc_o = (m.predict(Xd).argmax(1) == clean[tgt][nul]).mean(); acc_h = (h.predict(Xds) == clean[tgt][nul]).mean()
    assert abs(acc_o - acc_h) < 0.05 and acc_o > 0.7


def test_chain_fills_only_nulls_and_feeds_later_models():
    dirty, clean, cards = make_table(5000, 5, seed=9, null_ratio=0.05)
    targets = [1, 3]
    models, feats_l = [], []
    for t in targets:
        feats = [c for c in range(5) if c != t]; rows = dirty[t] >= 0; K = int(cards[t])
        models.append(O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[t][rows], K,
                              objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=10, learning_rate=0.3))
        feats_l.append(feats)
    tbl = dirty.copy()
    lab, prob = O.repair_chain(models, targets, feats_l, [list(range(int(cards[t]))) for t in targets], tbl)
    for i, t in enumerate(targets):
        was_null = dirty[t] < 0
        assert np.array_equal(tbl[t][~was_null], dirty[t][~was_null])          # non-NULL cells untouched
        assert np.array_equal(tbl[t][was_null], lab[i][was_null])              # NULL cells take the arg-max label
        assert (tbl[t] >= 0).all()
    # model 2 saw model 1's repairs: re-predicting with column 1 still NULL differs for some rows
    feats = feats_l[1]; both = (dirty[1] < 0) & (dirty[3] < 0)
    if both.any():
        p_chain = models[1].predict(np.ascontiguousarray(tbl[feats][:, both]))
        assert np.array_equal(p_chain.argmax(1), lab[1][both])


def test_model_trees_and_needed_built_rows_of_a_serialised_model():
    """repair._native.model_trees / needed_built_rows (bench.py's `roofline.frac_needed`): the blob parser against the test suite's own
    (tests/numerics_bound.parse_trees), and the rows the finished trees needed below their roots against a direct walk of one tree."""
    import numpy as np
    from oracle import oracle as O
    from repair import _native as N
    from tests.numerics_bound import parse_trees
    from tests.synth import make_table, balanced_weights
    d, _, cards = make_table(6000, 8, seed=3)
    t = 5
    feats = [c for c in range(8) if c != t]
    r = d[t] >= 0
    K = int(cards[t])
    blob = O.train(np.ascontiguousarray(d[feats][:, r]), cards[feats], d[t][r], K, class_weight=balanced_weights(d[t], K), objective=1, num_class=K,
                   n_estimators=3, max_depth=4, num_leaves=12, min_data_in_leaf=5).save()
    Ka, na, ta = N.model_trees(blob)
    Kb, nb, tb = parse_trees(blob)
    assert (Ka, na, len(ta)) == (Kb, nb, len(tb)) == (K, 3, 3 * K)
    for a, b in zip(ta, tb):
        for name in ("feat", "theta", "dleft", "left", "right", "gain", "leaf_value", "leaf_count"):
            assert np.array_equal(a[name], b[name])

    def walk(tr, max_depth):
        def size(ref):
            return int(tr["leaf_count"][~ref]) if ref < 0 else size(int(tr["left"][ref])) + size(int(tr["right"][ref]))

        def rec(ref, depth):
            if ref < 0:
                return 0
            a, b = int(tr["left"][ref]), int(tr["right"][ref])
            here = min(size(a), size(b)) if depth + 1 < max_depth else 0
            return here + rec(a, depth + 1) + rec(b, depth + 1)
        return rec(0, 0) if len(tr["feat"]) else 0
    assert N.needed_built_rows(blob, 4) == sum(walk(tr, 4) for tr in ta)
    assert 0 < N.needed_built_rows(blob, 4) < N.needed_built_rows(blob, 8) + 1
