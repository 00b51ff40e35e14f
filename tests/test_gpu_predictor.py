"""-m gpu: every form of the predictor gives the oracle's bits.

`predict_device` (csrc/rgbm.hip) picks per model: the fixed-stride bit-vector scorer (`k_predict_qs<MW, FMAX, TW, TBN>`: tables of the
sizes that occur), the dynamic-stride one (any table that fits the LDS; `RGBM_QS_FIXED=0` forces it) or the walk over index-linked
nodes (`k_predict_raw`: more than 64 leaves or more than 32 features; `RGBM_PREDICTOR=walk` forces it).  The reference's side is one
call, `model.predict_proba(X)` (python/repair/model.py:1120): all forms must return what the oracle returns for it, bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(n, cards, seed, k):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.integers(0, c, n) for c in cards]).astype(np.int32)
    # a label that depends on several features, some of them through their high bins, plus noise
    s = (X[0] * 3 + X[1] * 5 + X[len(cards) // 2] * 7 + X[-1]) + rng.integers(0, 3, n)
    y = (s % k).astype(np.int32)
    nul = rng.random(X.shape) < 0.02
    X[nul] = -1                                   # NULL cells: the missing bin
    return np.ascontiguousarray(X), y


SHAPES = [
    # name, feature cardinalities, classes, trainer parameters, the form predict_device picks
    ("one word, 16 features", [3, 5, 9, 17, 4, 6, 33, 2], 5, dict(num_leaves=31), "fixed"),
    ("one word, 32 features", [3, 5, 9, 17, 4, 6, 12, 2] * 3, 4, dict(num_leaves=31), "fixed"),
    ("two words, 16 features", [3, 5, 9, 17, 4, 6, 33, 2, 7], 3, dict(num_leaves=63, max_depth=-1, min_data_in_leaf=5), "fixed"),
    ("two words, 32 features", [4, 6, 9, 17, 3] * 4, 3, dict(num_leaves=60, max_depth=-1, min_data_in_leaf=5), "fixed"),
    ("tables beyond the fixed variants", [70, 90, 120, 64, 80, 100, 75, 66] * 3, 4, dict(num_leaves=31, max_bin=255), "dynamic"),
    ("more than 64 leaves", [3, 5, 9, 17, 4, 6, 33, 2], 3, dict(num_leaves=100, max_depth=-1, min_data_in_leaf=3), "walk"),
    ("more than 32 features", [3, 4, 5, 6, 7] * 7, 3, dict(num_leaves=31), "walk"),
]


@pytest.mark.parametrize("name,cards,K,kw,form", SHAPES, ids=[s[0] for s in SHAPES])
def test_every_predictor_form_returns_the_oracle_bits(name, cards, K, kw, form, monkeypatch):
    from oracle import oracle as O
    from repair import _native as N
    X, y = _table(6000, cards, seed=len(cards) * 7 + K, k=K)
    params = dict(objective=1, num_class=K, n_estimators=11, learning_rate=0.2, **kw)
    mo = O.train(X, np.asarray(cards, np.int32), y, K, **params)
    mg = N.train(X, np.asarray(cards, np.int32), y, K, **params)
    assert mo.save() == mg.save()
    Xp, _ = _table(3000, cards, seed=99, k=K)      # rows the model has not seen, NULLs included
    Xp[0, :50] = np.asarray(cards)[0] + 3          # ... and codes outside the dictionary
    want = mo.predict(Xp)
    got = mg.predict(Xp)
    assert np.array_equal(want, got), "default form (%s)" % form
    # a model loaded from its blob builds its device tables afresh, under each forced form
    for env in ({"RGBM_QS_FIXED": "0"}, {"RGBM_PREDICTOR": "walk"}):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        m2 = N.Model.load(mg.save())
        assert np.array_equal(want, m2.predict(Xp)), "forced %s on a fresh model" % env
        assert np.array_equal(want, mg.predict(Xp)), "forced %s on a model whose tables exist" % env
        for k_ in env:
            monkeypatch.delenv(k_)
    assert np.array_equal(want, mg.predict(Xp))


def test_chain_and_batch_without_page_locked_staging_give_the_same_bits(monkeypatch):
    """RGBM_NO_PIN=1 keeps the direct paths (pageable tree harvest of a batch, one hipMemcpy per model of a chain): same bytes."""
    from repair import _native as N
    from tests.synth import make_table, balanced_weights
    dirty, clean, cards = make_table(8000, 7, seed=41, null_ratio=0.03)
    targets = [1, 3, 5, 6]
    feats_l = [[c for c in range(7) if c != t] for t in targets]

    def run():
        tab = N.Table(dirty, cards)
        specs = [dict(table=tab, target_col=t, feat_cols=f, class_weight=balanced_weights(dirty[t], int(cards[t])),
                      objective=0 if cards[t] == 2 else 1, num_class=max(int(cards[t]), 2), n_estimators=9, learning_rate=0.2)
                 for t, f in zip(targets, feats_l)]
        models = N.train_batch(specs)
        for m in models:
            assert not isinstance(m, Exception), m
        blobs = [m.save() for m in models]
        d = N.Table(dirty, cards)
        lab, prob = d.repair_chain(models, targets, feats_l)
        cols = [d.read_column(t) for t in targets]
        return blobs, lab, prob, cols

    a = run()
    N.release_cache()                        # gives the page-locked blocks back as well: the next calls allocate them again
    c = run()
    assert a[0] == c[0] and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
    monkeypatch.setenv("RGBM_NO_PIN", "1")
    b = run()
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(x, y)
