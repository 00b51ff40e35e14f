"""-m gpu: numerics v2.2 -- the fixed-point grid of every class tree of every boosting iteration -- on the HIP engine against the oracle.

On a table that trains in seconds the coarse gradient sums are tiny and every class tree gets the finest grid there is (2^50 per value): the
per-tree arithmetic would never show.  The test hook RGBM_TEST_HOOKS=1 + RGBM_FX_ROWS = R (read by library and oracle alike) sizes the grids
as if the table held R rows with ITS gradient distribution: then the exponent of a class tree follows the coarse sum of its gradients, differs
from class tree to class tree and changes from iteration to iteration as the fit improves -- and the model bytes only agree with the
oracle's when every place that turns a float32 (g, h) into an integer or an integer sum back into a double uses the grid of the RIGHT
class tree of the RIGHT iteration: the gradient kernels' measurement (k_grad_mc, k_grad_mc_rows, k_grad<0/1/2> + k_fx_measure), k_fx_reduce /
k_fx_scale, the root pass, the level pass (its per-node table), the split search, the leaf-wise grower, the batched small-table trainer
(which measures inside k_small_tree) and the row-sharded trainer (one more integer all-reduce per iteration).
Reference semantics pinned: python/repair/train.py:102-131 (the fit), python/repair/model.py:1120-1124 (what is read off the model).
"""
import threading

import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[10_000_000, 100_000_000], autouse=True)
def grid_of_rows(request, monkeypatch):
    monkeypatch.setenv("RGBM_TEST_HOOKS", "1")
    monkeypatch.setenv("RGBM_FX_ROWS", str(request.param))
    return request.param


def _oracle(dirty, cards, target, feats, **kw):
    from oracle import oracle as O
    rows = dirty[target] >= 0
    return O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[target][rows], int(cards[target]), **kw).save()


def _kw(dirty, cards, target, **over):
    K = int(cards[target])
    return dict(dict(objective=0 if K == 2 else 1, num_class=max(K, 2), class_weight=balanced_weights(dirty[target], K), n_estimators=12, learning_rate=0.2), **over)


@pytest.mark.parametrize("target", [0, 1, 5, 7, 10])         # binary (k_grad<0>), K = 3 and 12 (k_grad_mc_rows), K = 24 and 64 (k_grad_mc)
def test_level_grower_under_a_coarse_grid_equals_the_oracle(target):
    from repair import _native as N
    dirty, _, cards = make_table(30000, 12, seed=91, null_ratio=0.02)
    feats = [c for c in range(12) if c != target]
    tab = N.Table(dirty, cards)
    for over in (dict(), dict(bagging_fraction=0.6, bagging_freq=2, feature_fraction=0.8), dict(num_leaves=60, min_data_in_leaf=3, n_estimators=5)):
        kw = _kw(dirty, cards, target, **over)
        assert tab.train(target, feats, **kw).save() == _oracle(dirty, cards, target, feats, **kw), (target, over)


def test_skewed_many_class_attributes_under_a_coarse_grid_equal_the_oracle(monkeypatch):
    """hospital `Score` (55 classes: k_grad_mc), `Sample` (303 classes: k_grad<1> + k_fx_measure) and `City` -- class weights spread over two
    orders of magnitude, the tables numerics v2.1 failed on at these grids -- 40 iterations of the reference's parameters."""
    from repair import _native as N
    from repair.encode import TableEncoder
    from tests.helpers import frame, load_golden
    g = load_golden("hospital")
    df = frame(g["input"], dtypes=False); df["tid"] = df["tid"].astype(int)
    cells = frame(g["error_cells"], dtypes=False); cells["tid"] = cells["tid"].astype(int)
    cols = [c for c in df.columns if c != "tid"]
    pos = {v: i for i, v in enumerate(df["tid"].tolist())}
    for r, a in zip(cells["tid"].tolist(), cells["attribute"].tolist()):
        if a in cols and r in pos:
            df.loc[df.index[pos[r]], a] = None
    enc = TableEncoder(df, cols)
    codes = enc.encode(df); cards = np.asarray(enc.n_codes, np.int32)
    tab = N.Table(codes, cards)
    for name in ("Score", "Sample", "City"):
        t = cols.index(name)
        feats = [c for c in range(len(cols)) if c != t]
        kw = _kw(codes, cards, t, n_estimators=40, learning_rate=0.01)
        assert tab.train(t, feats, **kw).save() == _oracle(codes, cards, t, feats, **kw), name
    # the leaf-wise grower (k_hist / k_split_find read the same table)
    monkeypatch.setenv("RGBM_GROWER", "leafwise")
    t = cols.index("Score")
    feats = [c for c in range(len(cols)) if c != t]
    kw = _kw(codes, cards, t, n_estimators=10, learning_rate=0.05)
    assert tab.train(t, feats, **kw).save() == _oracle(codes, cards, t, feats, **kw)


def test_regression_and_unbounded_depth_under_a_coarse_grid():
    from repair import _native as N
    rng = np.random.default_rng(3)
    dirty, clean, cards = make_table(20000, 7, seed=93, null_ratio=0.02)
    t = 6
    feats = list(range(6))
    K = int(cards[t])
    yv = np.sort(rng.normal(size=K)) * 3.0
    tab = N.Table(dirty, cards)
    for over in (dict(), dict(max_depth=-1, num_leaves=40)):          # level grower / leaf-wise grower (max_depth <= 0)
        kw = dict(objective=2, y_value=yv, class_weight=None, n_estimators=15, learning_rate=0.2, lambda_l2=0.5, **over)
        assert tab.train(t, feats, **kw).save() == _oracle(dirty, cards, t, feats, **kw), over


def test_batched_small_table_trainer_under_a_coarse_grid():
    from repair import _native as N
    dirty, _, cards = make_table(9000, 12, seed=7, null_ratio=0.02)
    tab = N.Table(dirty, cards)
    trials = [dict(), dict(num_leaves=7, min_data_in_leaf=5), dict(bagging_fraction=0.7, bagging_freq=3), dict(feature_fraction=0.5, lambda_l1=0.2)]
    fits, want = [], []
    for j, target in enumerate([0, 4, 10, 5]):
        feats = [c for c in range(12) if c != target]
        kw = _kw(dirty, cards, target, n_estimators=8, **trials[j])
        fits.append(dict(table=tab, target_col=target, feat_cols=feats, **kw))
        want.append(_oracle(dirty, cards, target, feats, **kw))
    out = N.train_batch(fits)
    for j, m in enumerate(out):
        assert isinstance(m, N.Model), "fit %d failed: %r" % (j, m)
        assert m.save() == want[j], "fit %d (%r)" % (j, trials[j])


@pytest.mark.parametrize("target,bounds", [(10, [0, 9000, 30000]), (5, [0, 3000, 3500, 14000, 30000]), (0, [0, 1, 30000])])
def test_row_shards_under_a_coarse_grid_give_the_single_device_model(target, bounds):
    """Row-sharded training sums the coarse gradient sums over the ranks (one more integer all-reduce per iteration): every rank picks the
    grid the single-device trainer picks, so the models stay bit-identical for any row split (also with bagging)."""
    from repair import _native as N
    dirty, _, cards = make_table(30000, 12, seed=91, null_ratio=0.02)
    feats = [c for c in range(12) if c != target]
    for over in (dict(), dict(bagging_fraction=0.6, bagging_freq=2)):
        kw = _kw(dirty, cards, target, n_estimators=7, **over)
        cw = kw.pop("class_weight")
        single = N.Table(dirty, cards).train(target, feats, class_weight=cw, **kw).save()
        nr = len(bounds) - 1
        group = N.LocalGroup(nr)
        out, err = [None] * nr, [None] * nr

        def work(r):
            try:
                group.join(r)
                try:
                    out[r] = N.Table(np.ascontiguousarray(dirty[:, bounds[r]:bounds[r + 1]]), cards).train(target, feats, class_weight=cw, row_sharded=True, **kw).save()
                finally:
                    N.comm_finalize()
            except Exception as e:  # noqa: BLE001
                err[r] = e

        ths = [threading.Thread(target=work, args=(r,)) for r in range(nr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
        for e in err:
            if e is not None:
                raise e
        for r, b in enumerate(out):
            assert b == single, "rank %d of %d (%r)" % (r, nr, over)
        assert single == _oracle(dirty, cards, target, feats, class_weight=cw, **kw)
