"""-m gpu: the batched small-table trainer (rgbm_table_train_batch, csrc/rgbm_small.h) -- many fits advancing through their boosting
iterations together, one workgroup per (fit, class tree) -- against the single-fit trainer (rgbm_table_train) and the CPU oracle.

What a batch is in the reference: the folds x trials of a hyper-parameter search (python/repair/train.py:158-209: every trial is
cross_val_score = n_splits fits on row subsets of one frame, with its own num_leaves / subsample / colsample / ... ) and the models of
a reference-default job (python/repair/model.py:755-766: <= 10 000 training rows each).  Every fit of a batch must come out as the
model the single call returns for the same arguments, bit for bit, whatever else shares the batch.
"""
import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


def _oracle_model(dirty, cards, target, feats, K, cw, **kw):
    from oracle import oracle as O
    rows = dirty[target] >= 0
    return O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[target][rows], K, class_weight=cw, **kw).save()


def test_batch_of_mixed_fits_equals_single_calls_and_oracle():
    from repair import _native as N
    dirty, _, cards = make_table(9000, 12, seed=7, null_ratio=0.02)
    tab = N.Table(dirty, cards)
    fits, single, oracle = [], [], []
    # the trials of a search draw all of these (train.py:148-156); one fit per row, deliberately different
    trials = [dict(), dict(num_leaves=7, min_data_in_leaf=5), dict(num_leaves=63, lambda_l2=2.0, min_sum_hessian_in_leaf=0.5),
              dict(bagging_fraction=0.7, bagging_freq=3), dict(feature_fraction=0.5, lambda_l1=0.2, min_gain_to_split=0.01),
              dict(bagging_fraction=0.55, bagging_freq=1, feature_fraction=0.8, num_leaves=20, max_depth=5)]
    for j, target in enumerate([0, 4, 7, 10, 1, 5]):          # binary, K = 8, 24, 64, 3, 12
        feats = [c for c in range(12) if c != target]
        K = int(cards[target])
        cw = balanced_weights(dirty[target], K)
        kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=6, learning_rate=0.2, **trials[j])
        fits.append(dict(table=tab, target_col=target, feat_cols=feats, class_weight=cw, **kw))
        single.append(tab.train(target, feats, class_weight=cw, **kw).save())
        oracle.append(_oracle_model(dirty, cards, target, feats, K, cw, **kw))
    out = N.train_batch(fits)
    assert len(out) == len(fits)
    for j, m in enumerate(out):
        assert isinstance(m, N.Model), "fit %d failed: %r" % (j, m)
        b = m.save()
        assert b == single[j], "fit %d (%r): the batched model differs from rgbm_table_train" % (j, trials[j])
        assert b == oracle[j], "fit %d (%r): the batched model differs from the oracle" % (j, trials[j])


def test_batch_over_fold_tables_regression_and_a_failing_fit():
    """Folds are row gathers of the training table (pipeline.search_on_table): different tables in one batch, a regression fit, and a
    fit whose target column is entirely NULL (fails alone: train.py:227-229 turns it into PoorModel, the batch goes on)."""
    from repair import _native as N
    rng = np.random.default_rng(3)
    n = 6000
    x0 = rng.integers(0, 30, n); x1 = rng.integers(0, 5, n); x2 = rng.integers(0, 40, n)
    yv = np.arange(50, dtype=np.float64) * 0.37
    y = np.clip(x0 + 3 * x1 + rng.integers(0, 4, n), 0, 49)
    dead = np.full(n, -1, np.int32)
    codes = np.ascontiguousarray(np.stack([x0, x1, x2, y, dead]).astype(np.int32))
    codes[0, rng.random(n) < 0.03] = -1
    cards = np.array([30, 5, 40, 50, 3], np.int32)
    tab = N.Table(codes, cards)
    folds = [np.flatnonzero(np.arange(n) % 3 != r).astype(np.int64) for r in range(3)]
    ftabs = [tab.gather_rows(f) for f in folds]
    fits, single = [], []
    for r, ft in enumerate(ftabs):
        kw = dict(objective=2, n_estimators=8, learning_rate=0.1, num_leaves=15, min_data_in_leaf=3 + r)
        fits.append(dict(table=ft, target_col=3, feat_cols=[0, 1, 2], y_value=yv, **kw))
        single.append(ft.train(3, [0, 1, 2], y_value=yv, **kw).save())
    fits.append(dict(table=tab, target_col=4, feat_cols=[0, 1, 2], objective=1, num_class=3, n_estimators=4))      # no training rows
    kw = dict(objective=1, num_class=5, n_estimators=5, learning_rate=0.3)
    fits.append(dict(table=tab, target_col=1, feat_cols=[0, 2, 3], class_weight=balanced_weights(codes[1], 5), **kw))
    single.append(None)
    single.append(tab.train(1, [0, 2, 3], class_weight=balanced_weights(codes[1], 5), **kw).save())
    out = N.train_batch(fits)
    for j, m in enumerate(out):
        if single[j] is None:
            assert isinstance(m, N.RepairGbmError) and "no training rows" in str(m)
        else:
            assert isinstance(m, N.Model), "fit %d failed: %r" % (j, m)
            assert m.save() == single[j], "fit %d: the batched model differs from rgbm_table_train" % j


def test_a_search_sized_batch_of_48_fits():
    """16 trials x 3 folds on a 10 000-row table (the reference's default training sample): every model equals the single call's."""
    from repair import _native as N
    dirty, _, cards = make_table(10000, 16, seed=42, null_ratio=0.01)
    target, K = 4, int(cards[4])
    feats = [c for c in range(16) if c != target]
    tab = N.Table(dirty, cards)
    n = dirty.shape[1]
    rng = np.random.RandomState(42)
    fits = []
    for trial in range(16):
        kw = dict(objective=1, num_class=K, n_estimators=12, num_leaves=int(rng.randint(2, 101)), bagging_fraction=float(rng.uniform(0.5, 1.0)),
                  bagging_freq=int(rng.randint(1, 21)), feature_fraction=float(rng.uniform(0.01, 1.0)), min_data_in_leaf=int(rng.randint(1, 51)),
                  min_sum_hessian_in_leaf=float(np.exp(rng.uniform(-3, 1))), lambda_l2=float(np.exp(rng.uniform(-2, 3))))
        for r in range(3):
            rows = np.flatnonzero(np.arange(n) % 3 != r).astype(np.int64)
            ft = tab.gather_rows(rows)
            cw = balanced_weights(dirty[target][rows], K)
            fits.append(dict(table=ft, target_col=target, feat_cols=feats, class_weight=cw, **kw))
    out = N.train_batch(fits)
    assert len(out) == 48
    for j in (0, 7, 13, 22, 31, 40, 47):                     # spot checks against the single-fit trainer (each is a full training call)
        f = dict(fits[j]); ft = f.pop("table")
        ref = ft.train(f.pop("target_col"), f.pop("feat_cols"), class_weight=f.pop("class_weight"), **f).save()
        assert isinstance(out[j], N.Model) and out[j].save() == ref, "fit %d of 48 differs from the single call" % j


def test_search_on_table_hip_engine_equals_the_oracle_engine():
    """pipeline.search_on_table (train.py:133-209 on resident tables) on the HIP engine -- the fold fits of every batch of
    evaluations in ONE rgbm_table_train_batch call -- against the same search on the oracle engine (one training call per fit,
    CPU): the same best parameters, and the final model trained with them is the same bytes."""
    from repair import pipeline
    from repair.engine import HipEngine, balanced_class_weight, model_params
    from tests.helpers import OracleEngine
    dirty, _, cards = make_table(4000, 8, seed=11, null_ratio=0.02)
    base = dict(num_leaves=31, max_depth=7, max_bin=255, min_data_in_leaf=20, min_data_in_bin=3, bagging_freq=0, seed=42, learning_rate=0.05,
                lambda_l1=0.0, lambda_l2=0.0, min_gain_to_split=0.0, min_sum_hessian_in_leaf=1e-3, bagging_fraction=1.0, feature_fraction=1.0, n_estimators=15)
    opts = {"model.hp.max_evals": "6", "model.hp.no_progress_loss": "4", "model.hp.batch_size": "3", "model.cv.n_splits": "3"}
    hip, orc = HipEngine(0), OracleEngine()
    th, to = hip.upload(dirty, cards), orc.upload(dirty, cards)
    for t in (2, 5):                                          # K = 4 and K = 12
        ph = pipeline.search_on_table(hip, th, t, cards, base, opts)
        po = pipeline.search_on_table(orc, to, t, cards, base, opts)
        assert ph == po, "target c%d: the batched search found %r, the sequential oracle search %r" % (t, ph, po)
        feats = [c for c in range(8) if c != t]
        cw = balanced_class_weight(np.bincount(dirty[t][dirty[t] >= 0], minlength=int(cards[t])))
        p = model_params(int(cards[t]), dict(base, **ph))
        assert hip.train(th, t, feats, cw, p).save() == orc.train(to, t, feats, cw, p).save()


def test_validation_rows_scored_while_training_equal_the_predictor():
    """cross_val_score (train.py:171-172) without a predictor: a fit with a valid_table returns, for that table's rows, the labels and
    values `repair_chain` of its finished model gives -- bit for bit (the trees are added to the validation scores in the predictor's
    order).  Multiclass with bagging, binary, regression; the validation fold holds NULL feature cells and a category no training row
    of the fold has (missing for the model)."""
    from repair import _native as N
    dirty, _, cards = make_table(7000, 9, seed=31, null_ratio=0.03)
    dirty[3, :40] = int(cards[3]) - 1
    dirty[3, 40:][dirty[3, 40:] == int(cards[3]) - 1] = 0            # the last category of c3 only occurs in rows 0..39
    tab = N.Table(dirty, cards)
    tab.set_column_kind(3, True)
    n = dirty.shape[1]
    va = np.arange(0, n, 3).astype(np.int64)                          # holds rows 0, 3, ..., 39: the unseen category
    tr = np.setdiff1d(np.arange(n), va).astype(np.int64)
    ttab, vtab = tab.gather_rows(tr), tab.gather_rows(va)
    yv = np.arange(int(cards[7]), dtype=np.float64) * 1.5 - 3.0
    specs = [dict(target_col=5, objective=1, num_class=int(cards[5]), bagging_fraction=0.7, bagging_freq=2, num_leaves=20),
             dict(target_col=0, objective=0, num_class=2, feature_fraction=0.7),
             dict(target_col=7, objective=2, y_value=yv, num_leaves=12)]
    fits = []
    for sp in specs:
        t = sp["target_col"]
        feats = [c for c in range(9) if c != t]
        cw = None if sp["objective"] == 2 else balanced_weights(dirty[t][tr], int(cards[t]))
        fits.append(dict(table=ttab, feat_cols=feats, class_weight=cw, valid_table=vtab, n_estimators=10, learning_rate=0.2, **sp))
    out = N.train_batch(fits)
    for sp, f, res in zip(specs, fits, out):
        assert isinstance(res, tuple), "fit failed: %r" % (res,)
        m, lab, val = res
        check = tab.gather_rows(va)                                   # repair_chain fills NULL target cells in place: a fresh copy
        ref_lab, ref_val = check.repair_chain([m], [sp["target_col"]], [f["feat_cols"]])
        assert np.array_equal(lab, ref_lab[0]) and np.array_equal(val, ref_val[0]), "objective %d: validation scores differ from the predictor" % sp["objective"]
        single = ttab.train(sp["target_col"], f["feat_cols"], class_weight=f["class_weight"], y_value=sp.get("y_value"),
                            **{k: v for k, v in sp.items() if k not in ("target_col", "y_value")}, n_estimators=10, learning_rate=0.2)
        assert m.save() == single.save()


def test_a_fit_is_frozen_after_its_first_iteration_without_a_split():
    """the model ends at the first boosting iteration in which no class tree could split (LightGBM stops training
    there), but with feature_fraction < 1 a LATER iteration draws other features and could split again -- its trees must not reach the
    validation scores of a CV fold, or cross_val_score would score a model that does not exist.  One informative feature among noise,
    min_gain_to_split high enough that noise never splits, two of six features per tree: most seeds stop within a few iterations.
    Every fit: validation labels / values == repair_chain of the returned model, model == the single-fit trainer's."""
    from repair import _native as N
    rng = np.random.default_rng(5)
    n = 6000
    codes = np.empty((7, n), np.int32)
    codes[1] = rng.integers(0, 4, n)
    codes[0] = ((codes[1] >= 2) ^ (rng.random(n) < 0.1)).astype(np.int32)
    for c in range(2, 7):
        codes[c] = rng.integers(0, 5, n)
    cards = np.asarray([2, 4, 5, 5, 5, 5, 5], np.int32)
    tab = N.Table(codes, cards)
    va = np.arange(0, n, 4).astype(np.int64)
    tr = np.setdiff1d(np.arange(n), va).astype(np.int64)
    ttab, vtab = tab.gather_rows(tr), tab.gather_rows(va)
    feats = list(range(1, 7))
    NE = 12
    kw = dict(objective=0, num_class=2, n_estimators=NE, learning_rate=0.3, feature_fraction=0.34, min_gain_to_split=25.0, min_data_in_leaf=50)
    fits = [dict(table=ttab, target_col=0, feat_cols=feats, class_weight=None, valid_table=vtab, seed=sd, **kw) for sd in range(1, 9)]
    out = N.train_batch(fits)
    stopped = 0
    for f, res in zip(fits, out):
        assert isinstance(res, tuple), "fit failed: %r" % (res,)
        m, lab, val = res
        n_iter = m.info()["n_iter"]
        stopped += 1 if n_iter < NE else 0
        check = tab.gather_rows(va)
        ref_lab, ref_val = check.repair_chain([m], [0], [feats])
        assert np.array_equal(lab, ref_lab[0]) and np.array_equal(val, ref_val[0]), "seed %d (model of %d iterations): CV scores are not the model's" % (f["seed"], n_iter)
        single = ttab.train(0, feats, class_weight=None, seed=f["seed"], **kw)
        assert m.save() == single.save()
    assert stopped >= 1, "no seed stopped early: the case this test is about did not occur"
