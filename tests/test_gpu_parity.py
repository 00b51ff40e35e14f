"""-m gpu: HIP engine vs CPU oracle, bit-exact (serialised model bytes, probabilities, labels).

Parity bar (BASELINE.json north_star): arg-max labels bit-identical, probabilities within 1e-4 --
in practice identical bits, which is what is asserted here; the tolerance is the fall-back bar.
"""
import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


def _both(X, n_codes, y, K, obj, cw=None, yv=None, sw=None, **kw):
    from oracle import oracle as O
    from repair import _native as N
    kw = dict(kw)
    params = dict(objective=obj, num_class=max(K, 2), **kw)
    mo = O.train(X, n_codes, y, K, y_value=yv, class_weight=cw, sample_weight=sw, **params)
    mg = N.train(X, n_codes, y, K, y_value=yv, class_weight=cw, sample_weight=sw, **params)
    return mo, mg


def _assert_same(mo, mg, X):
    bo, bg = mo.save(), mg.save()
    assert mo.info() == mg.info()
    assert len(bo) == len(bg)
    assert bo == bg, "serialised models differ"
    po, pg = mo.predict(X), mg.predict(X)
    assert np.array_equal(po.argmax(1), pg.argmax(1))
    assert np.allclose(po, pg, rtol=0, atol=1e-4)
    assert np.array_equal(po, pg), "probabilities are not bit-identical"


@pytest.mark.parametrize("tgt", [0, 1, 3, 7])
def test_classifier_models_bit_exact(tgt):
    dirty, clean, cards = make_table(20000, 8, seed=1)
    feats = [c for c in range(8) if c != tgt]
    rows = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, rows])
    y = dirty[tgt][rows]
    K = int(cards[tgt])
    cw = balanced_weights(y, K)
    mo, mg = _both(X, cards[feats], y, K, 0 if K == 2 else 1, cw=cw, n_estimators=20, learning_rate=0.1)
    _assert_same(mo, mg, X)
    assert mg.info()["n_iter"] == 20


def test_default_reference_params_small():
    # reference fixed params (train.py:102-115): lr 0.01, depth 7, 31 leaves, min_child_samples 20
    dirty, clean, cards = make_table(5000, 6, seed=2)
    tgt = 4
    feats = [c for c in range(6) if c != tgt]
    rows = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, rows]); y = dirty[tgt][rows]; K = int(cards[tgt])
    mo, mg = _both(X, cards[feats], y, K, 1, cw=balanced_weights(y, K), n_estimators=30)
    _assert_same(mo, mg, X)


def test_regression_bit_exact():
    rng = np.random.default_rng(3)
    dirty, clean, cards = make_table(8000, 6, seed=3)
    X = np.ascontiguousarray(dirty[:5])
    vals = np.sort(rng.normal(size=40))
    y = (clean[5] % 40).astype(np.int32)
    mo, mg = _both(X, cards[:5], y, 40, 2, yv=vals, n_estimators=25, learning_rate=0.1, lambda_l2=0.5)
    _assert_same(mo, mg, X)


def test_tiny_and_ragged_inputs():
    # fewer rows than 2*min_data_in_leaf -> constant model; one row; all-NULL feature column
    from oracle import oracle as O
    from repair import _native as N
    X = np.array([[0, 1, 0, 1, 1], [-1, -1, -1, -1, -1]], np.int32)
    y = np.array([0, 1, 0, 1, 1], np.int32)
    for kw in (dict(min_data_in_leaf=20), dict(min_data_in_leaf=1, min_sum_hessian_in_leaf=1e-3)):
        mo = O.train(X, [2, 3], y, 2, objective=0, n_estimators=5, **kw)
        mg = N.train(X, [2, 3], y, 2, objective=0, n_estimators=5, **kw)
        assert mo.save() == mg.save()
        assert np.array_equal(mo.predict(X), mg.predict(X))


def test_high_cardinality_binning():
    # > max_bin distinct codes -> GreedyFindBin quantile path; rare codes merge (min_data_in_bin)
    rng = np.random.default_rng(5)
    n = 30000
    a = (rng.zipf(1.3, n) % 1000).astype(np.int32)
    b = rng.integers(0, 300, n).astype(np.int32)
    c = rng.integers(0, 5, n).astype(np.int32)
    y = ((a % 7 + b % 3 + c) % 4).astype(np.int32)
    X = np.ascontiguousarray(np.stack([a, b, c]))
    X[0][rng.random(n) < 0.02] = -1
    mo, mg = _both(X, [1000, 300, 5], y, 4, 1, cw=balanced_weights(y, 4), n_estimators=10, learning_rate=0.2, max_bin=63)
    _assert_same(mo, mg, X)


def test_more_than_16_features_two_chunks():
    dirty, clean, cards = make_table(12000, 21, seed=7)
    tgt = 20
    feats = list(range(20))
    rows = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, rows]); y = dirty[tgt][rows]; K = int(cards[tgt])
    mo, mg = _both(X, cards[feats], y, K, 1, cw=balanced_weights(y, K), n_estimators=8, learning_rate=0.2)
    _assert_same(mo, mg, X)


def test_determinism_run_twice():
    from repair import _native as N
    dirty, clean, cards = make_table(30000, 8, seed=11)
    X = np.ascontiguousarray(dirty[:7]); y = clean[7]
    a = N.train(X, cards[:7], y, int(cards[7]), objective=1, num_class=int(cards[7]), n_estimators=6, learning_rate=0.2).save()
    b = N.train(X, cards[:7], y, int(cards[7]), objective=1, num_class=int(cards[7]), n_estimators=6, learning_rate=0.2).save()
    assert a == b


def test_table_path_and_chain_match_oracle():
    """Resident-table training (rows with NULL target excluded on the device) + chained repair."""
    from oracle import oracle as O
    from repair import _native as N
    dirty, clean, cards = make_table(15000, 6, seed=13, null_ratio=0.03)
    targets = [1, 2, 4]
    tab = N.Table(dirty, cards)
    mo_l, mg_l, feats_l = [], [], []
    for t in targets:
        feats = [c for c in range(6) if c != t]
        rows = dirty[t] >= 0
        K = int(cards[t])
        cw = balanced_weights(dirty[t], K)
        kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=10, learning_rate=0.2)
        mo = O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[t][rows], K, class_weight=cw, **kw)
        mg = tab.train(t, feats, class_weight=cw, **kw)
        assert mo.save() == mg.save(), "table-path model differs for target %d" % t
        mo_l.append(mo); mg_l.append(mg); feats_l.append(feats)
    ref = dirty.copy()
    lab_o, prob_o = O.repair_chain(mo_l, targets, feats_l, [list(range(int(cards[t]))) for t in targets], ref)
    lab_g, prob_g = tab.repair_chain(mg_l, targets, feats_l)
    assert np.array_equal(lab_o, lab_g)
    assert np.array_equal(prob_o, prob_g)
    for t in targets:
        assert np.array_equal(tab.read_column(t), ref[t])
    # host-array chain entry point
    tbl = dirty.copy()
    lab_h, prob_h = N.repair_chain(mg_l, targets, feats_l, [list(range(int(cards[t]))) for t in targets], tbl)
    assert np.array_equal(lab_h, lab_o) and np.array_equal(tbl, ref) and np.array_equal(prob_h, prob_o)
    # repaired cells are mostly right (the generator has learnable structure)
    for i, t in enumerate(targets):
        nul = dirty[t] < 0
        assert (ref[t][nul] == clean[t][nul]).mean() > 0.7


@pytest.mark.parametrize("kw", [
    dict(bagging_fraction=0.7, bagging_freq=1),
    dict(bagging_fraction=0.55, bagging_freq=3, feature_fraction=0.5),
    dict(feature_fraction=0.3),
    dict(bagging_fraction=0.9, bagging_freq=2, feature_fraction=0.8, lambda_l1=0.5, lambda_l2=2.0, min_gain_to_split=0.01,
         num_leaves=63, max_depth=-1, min_data_in_leaf=7, min_sum_hessian_in_leaf=0.05),
])
def test_sampling_and_regularisation_params_bit_exact(kw):
    """The reference's searched parameters (train.py:148-156): subsample, subsample_freq, colsample_bytree, ..."""
    dirty, clean, cards = make_table(9000, 9, seed=17, null_ratio=0.02)
    tgt = 6
    feats = [c for c in range(9) if c != tgt]
    rows = dirty[tgt] >= 0
    X = np.ascontiguousarray(dirty[feats][:, rows]); y = dirty[tgt][rows]; K = int(cards[tgt])
    mo, mg = _both(X, cards[feats], y, K, 1, cw=balanced_weights(y, K), n_estimators=12, learning_rate=0.15, **kw)
    _assert_same(mo, mg, X)


def test_bagging_on_table_path_with_null_targets():
    from oracle import oracle as O
    from repair import _native as N
    dirty, clean, cards = make_table(7000, 6, seed=19, null_ratio=0.05)
    t = 2; feats = [c for c in range(6) if c != t]; rows = dirty[t] >= 0; K = int(cards[t])
    kw = dict(objective=1, num_class=K, n_estimators=9, learning_rate=0.2, bagging_fraction=0.6, bagging_freq=2, feature_fraction=0.7)
    cw = balanced_weights(dirty[t], K)
    mo = O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[t][rows], K, class_weight=cw, **kw)
    mg = N.Table(dirty, cards).train(t, feats, class_weight=cw, **kw)
    assert mo.save() == mg.save()


def test_sample_weight_path_bit_exact():
    rng = np.random.default_rng(23)
    dirty, clean, cards = make_table(5000, 6, seed=23)
    X = np.ascontiguousarray(dirty[:5]); y = clean[5]; K = int(cards[5])
    sw = rng.uniform(0.2, 3.0, X.shape[1])
    mo, mg = _both(X, cards[:5], y, K, 1, cw=balanced_weights(y, K), sw=sw, n_estimators=10, learning_rate=0.2)
    _assert_same(mo, mg, X)


def test_build_model_hp_search_runs_on_gpu_and_matches_oracle_backend():
    """train.build_model (hp search + CV + final fit) end to end on the HIP backend == oracle backend."""
    import pandas as pd
    from repair import gbm
    from repair.train import build_model
    from tests.helpers import OracleBackend
    rng = np.random.default_rng(29)
    X = pd.DataFrame({"a": rng.choice(list("xyz"), 600), "b": rng.choice(list("pqrs"), 600), "c": rng.integers(0, 6, 600)})
    y = pd.Series(np.where(X.a == "x", "A", np.where(X.b == "p", "B", "C")))
    opts = {"model.hp.max_evals": "4", "model.lgb.n_estimators": "25", "model.lgb.learning_rate": "0.2", "model.hp.no_progress_loss": "3"}
    import faulthandler
    import sys
    import time
    t0 = time.perf_counter()
    faulthandler.dump_traceback_later(20, repeat=False, file=sys.stderr)     # a search that takes longer than 20 s shows where its threads sit
    try:
        (mg, sg), _ = build_model(X, y, True, 3, n_jobs=-1, opts=opts)
    finally:
        faulthandler.cancel_dump_traceback_later()
    t1 = time.perf_counter()
    prev = gbm.set_backend(OracleBackend)
    try:
        (mo, so), _ = build_model(X, y, True, 3, n_jobs=-1, opts=opts)
    finally:
        gbm.set_backend(prev)
    print("hp search: HIP backend %.2fs, oracle backend %.2fs" % (t1 - t0, time.perf_counter() - t1))
    assert mg is not None and mo is not None and sg == so
    assert mg.booster_bytes_ == mo.booster_bytes_
    assert np.array_equal(mg.predict_proba(X), mo.predict_proba(X))


def test_numeric_columns_bin_at_value_midpoints_like_the_oracle():
    """rgbm_table_set_column_values: with the value dictionary of a numeric column the bin bounds sit at the midpoints of the VALUES
    (LightGBM on raw numbers), which only shows for codes that no training row holds.  HIP == oracle, and the bounds differ from the
    code-midpoint ones exactly where the values are unevenly spaced."""
    from oracle import oracle as O
    from repair import _native as N
    rng = np.random.default_rng(211)
    n = 30000
    vals0 = np.cumsum(rng.exponential(1.0, 40) ** 3 + 0.01)            # very unevenly spaced
    x0 = rng.integers(0, 40, n).astype(np.int32)
    x0[x0 % 3 == 1] = (x0[x0 % 3 == 1] + 1) % 40                       # codes = 1 mod 3 never occur among the training rows
    x1 = rng.integers(0, 6, n).astype(np.int32)
    y = ((vals0[x0] > np.median(vals0)).astype(np.int32) + x1) % 3
    tab = np.ascontiguousarray(np.stack([x0, x1, y.astype(np.int32)]))
    cards = np.asarray([40, 6, 3], np.int32)
    kw = dict(objective=1, num_class=3, n_estimators=6, learning_rate=0.3, min_data_in_leaf=5)
    cw = np.ones(3)
    t = N.Table(tab, cards)
    plain = t.train(2, [0, 1], class_weight=cw, **kw).save()
    t.set_column_values(0, vals0)
    withv = t.train(2, [0, 1], class_weight=cw, **kw).save()
    mo = O.train(tab[:2], cards[:2], tab[2], 3, class_weight=cw, feature_values={0: vals0}, **kw).save()
    assert withv == mo
    assert plain == O.train(tab[:2], cards[:2], tab[2], 3, class_weight=cw, **kw).save()
    assert withv != plain
    g = t.gather_rows(np.arange(0, n, 2))                               # row gathers inherit the dictionary
    assert g.train(2, [0, 1], class_weight=cw, **kw).save() == O.train(np.ascontiguousarray(tab[:2, ::2]), cards[:2], tab[2, ::2], 3, class_weight=cw,
                                                                         feature_values={0: vals0}, **kw).save()


def test_categorical_columns_record_unseen_codes_like_the_oracle():
    """rgbm_table_set_column_kind: a model trained from a table whose column is CATEGORICAL records the codes none of its training rows
    held (format version 2) and scores them as missing.  Same bytes and same predictions as the oracle; without the marking the
    model is version 1 and bins the unseen code next to its neighbours."""
    import struct
    from oracle import oracle as O
    from repair import _native as N
    rng = np.random.default_rng(223)
    n = 20000
    x0 = rng.integers(0, 12, n).astype(np.int32)
    x1 = rng.integers(0, 5, n).astype(np.int32)
    y = ((x0 // 3) + x1) % 4
    y = y.astype(np.int32)
    hold = (x0 == 7) | (x0 == 2)                                 # rows holding codes 2 / 7 of x0 lose their label: not training rows
    ytab = np.where(hold, -1, y).astype(np.int32)
    tab = np.ascontiguousarray(np.stack([x0, x1, ytab]))
    cards = np.asarray([12, 5, 4], np.int32)
    kw = dict(objective=1, num_class=4, n_estimators=5, learning_rate=0.3, min_data_in_leaf=5)
    cw = np.ones(4)
    t = N.Table(tab, cards)
    plain = t.train(2, [0, 1], class_weight=cw, **kw)
    t.set_column_kind(0, True)
    cat = t.train(2, [0, 1], class_weight=cw, **kw)
    rows = ~hold
    mo = O.train(np.ascontiguousarray(tab[:2][:, rows]), cards[:2], tab[2][rows], 4, class_weight=cw, categorical=[0], **kw)
    assert cat.save() == mo.save()
    assert struct.unpack_from("2i", cat.save(), 0)[1] == 2 and struct.unpack_from("2i", plain.save(), 0)[1] == 1
    X = np.ascontiguousarray(tab[:2])
    pg, po = cat.predict(X), mo.predict(X)
    assert np.array_equal(pg, po)
    Xm = X.copy(); Xm[0][hold] = -1                              # an unseen category scores exactly like a missing value
    assert np.array_equal(pg, cat.predict(Xm))
    assert not np.array_equal(pg[hold], plain.predict(X)[hold])
    assert N.Model.load(cat.save()).save() == cat.save()
