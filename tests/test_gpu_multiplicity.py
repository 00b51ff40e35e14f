"""-m gpu: rows with multiplicities (rgbm_table_set_row_multiplicity, repair.pipeline.distinct_rows) -- a VARIANT of the workload.

Identical (features, label) rows take identical paths and gradients in every tree, and every sum of the trainer is an exact integer: training
on the DISTINCT rows with integer multiplicities must return, byte for byte, the model the whole table gives (and therefore the oracle's, which
only ever sees the whole table).  Checked for every form of the level grower the variant covers: binary / few-class / many-class targets
(k_grad<0>, k_grad_mc_rows, k_grad_mc: the coarse sums of numerics v2.2 weigh a row by its multiplicity), the joint-bin root pass and the plain
one, a two-chunk table (the multiplicity rides in the second record), rows that occur more than 255 times (split), the coarse-grid hook, thread-rank
row shards; and that what the variant cannot honour fails loudly instead of training something else.
Reference semantics pinned: the model of python/repair/train.py:102-131 on the frame of python/repair/model.py:768-815 (every row)."""
import threading

import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


def _kw(dirty, cards, t, **over):
    K = int(cards[t])
    return dict(dict(objective=0 if K == 2 else 1, num_class=max(K, 2), class_weight=balanced_weights(dirty[t], K), n_estimators=10, learning_rate=0.2), **over)


def _distinct_table(dirty, cards):
    from repair import _native as N
    from repair.pipeline import distinct_rows
    dist, mult, inv = distinct_rows(dirty, cards)
    assert int(mult.sum()) == dirty.shape[1] and np.array_equal(dist[:, inv], dirty)
    tab = N.Table(dist, cards)
    tab.set_row_multiplicity(mult)
    return tab, dist, mult


@pytest.mark.parametrize("rows,cols,targets", [(60000, 8, [0, 1, 4, 7]), (400000, 16, [10, 0]), (30000, 24, [7])])
def test_distinct_rows_with_multiplicities_give_the_whole_tables_model(rows, cols, targets, monkeypatch):
    from oracle import oracle as O
    from repair import _native as N
    dirty, _, cards = make_table(rows, cols, seed=97, null_ratio=0.01)
    if cols == 24:
        dirty = np.ascontiguousarray(np.concatenate([dirty, dirty[:, :20000]], axis=1))
    if cols == 8:
        dirty = np.ascontiguousarray(np.concatenate([dirty, dirty[:, :9000], np.repeat(dirty[:, :3], 400, axis=1)], axis=1))     # rows that occur > 255 times
    whole = N.Table(dirty, cards)
    tab, dist, mult = _distinct_table(dirty, cards)
    assert dist.shape[1] < dirty.shape[1] or cols > 16          # (a table of many columns may hold no duplicate at all: multiplicities of 1)
    for t in targets:
        feats = [c for c in range(cols) if c != t]
        for over, env in ((dict(), {}), (dict(num_leaves=60, min_data_in_leaf=3, n_estimators=4), {"RGBM_JOINT_ROOT": "0"}), (dict(n_estimators=6), {"RGBM_TEST_HOOKS": "1", "RGBM_FX_ROWS": "100000000"})):
            if env.get("RGBM_JOINT_ROOT") == "0" and cols > 16:
                continue          # (two chunks: the plain root pass has no record that carries the multiplicity -> refused, below)
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            kw = _kw(dirty, cards, t, **over)
            a = whole.train(t, feats, **kw).save()
            b = tab.train(t, feats, **kw).save()
            for k_ in env:
                monkeypatch.delenv(k_)
            assert a == b, (t, over, env)
            if not over and not env:
                r = dirty[t] >= 0
                assert b == O.train(np.ascontiguousarray(dirty[feats][:, r]), cards[feats], dirty[t][r], int(cards[t]), **kw).save()


def test_row_shards_of_distinct_rows():
    from repair import _native as N
    dirty, _, cards = make_table(50000, 10, seed=99, null_ratio=0.02)
    t, feats = 9, list(range(9))
    kw = _kw(dirty, cards, t, n_estimators=6)
    cw = kw.pop("class_weight")
    single = N.Table(dirty, cards).train(t, feats, class_weight=cw, **kw).save()
    from repair.pipeline import distinct_rows
    dist, mult, _ = distinct_rows(dirty, cards)
    bounds = [0, dist.shape[1] // 3, dist.shape[1]]
    group = N.LocalGroup(2)
    out, err = [None, None], [None, None]

    def work(r):
        try:
            group.join(r)
            try:
                tab = N.Table(np.ascontiguousarray(dist[:, bounds[r]:bounds[r + 1]]), cards)
                tab.set_row_multiplicity(mult[bounds[r]:bounds[r + 1]])
                out[r] = tab.train(t, feats, class_weight=cw, row_sharded=True, **kw).save()
            finally:
                N.comm_finalize()
        except Exception as e:  # noqa: BLE001
            err[r] = e
    ths = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=600)
    for e in err:
        if e is not None:
            raise e
    assert out[0] == single and out[1] == single


def test_what_the_variant_cannot_honour_fails_loudly(monkeypatch):
    from repair import _native as N
    dirty, _, cards = make_table(20000, 17, seed=101)
    tab, _, _ = _distinct_table(dirty, cards)
    feats16 = list(range(16))
    with pytest.raises(N.RepairGbmError):          # 16 features: no free byte in the record
        tab.train(16, feats16, **_kw(dirty, cards, 16, n_estimators=2))
    feats = list(range(15))
    with pytest.raises(N.RepairGbmError):          # bagging draws per ORIGINAL row
        tab.train(16, feats, **_kw(dirty, cards, 16, n_estimators=2, bagging_fraction=0.5, bagging_freq=1))
    with pytest.raises(N.RepairGbmError):          # the leaf-wise grower does not carry multiplicities
        tab.train(16, feats, **_kw(dirty, cards, 16, n_estimators=2, max_depth=-1))
    tab.set_row_multiplicity(None)                 # cleared: an ordinary table again
    assert tab.train(16, feats16, **_kw(dirty, cards, 16, n_estimators=2)).info()["n_iter"] == 2
