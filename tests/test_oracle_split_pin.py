"""CPU: the oracle's TREE GROWTH pinned from outside.

The reference-held golden labels (test_oracle_golden.py) all come from 6-20-row tables on which LightGBM
cannot split, so they pin BoostFromScore / class weights / arg-max but nothing about binning, gain,
tie-breaks or leaf values.  LightGBM itself is not installable here; two independent yardsticks are:

  1. a hand-computed two-level tree (numbers derived in the docstring of the test, from LightGBM's
     formulas: gain = GL^2/HL + GR^2/HR - G^2/H, output = -G/H, Tree::Split leaf numbering);
  2. scikit-learn's HistGradientBoostingRegressor -- a separate implementation of the same algorithm
     family (histogram GBDT, best-first growth, same gain / leaf-value formulas for L2 loss, same
     min_samples_leaf / max_leaf_nodes / max_depth semantics).  With one bin per code on both sides, unit
     hessians, no class weights, and labels chosen so that every first-iteration gradient is exactly
     representable in both gradient formats (sklearn: float32, oracle: 2^-k fixed point, D1), the first
     tree must agree EXACTLY: same split feature and threshold bin at every node, gains to 1e-9 relative,
     leaf values / predictions to 1e-9.  Later iterations see non-dyadic scores, where BOTH sides round
     (float32 vs fixed point), so they are held to the same tree structure and 1e-4 on predictions.
"""
import struct

import numpy as np
import pytest

from oracle import oracle as O


def _parse(blob):
    hdr = struct.unpack_from("7i", blob, 0)
    p, F, nt = 28, hdr[6], hdr[4] * hdr[5]
    feats = []
    for _ in range(F):
        nc, V, hn = struct.unpack_from("3i", blob, p); p += 12
        feats.append(dict(n_codes=nc, V=V, has_nan=hn, ub=np.frombuffer(blob, np.int32, V, p).copy())); p += 4 * V
    trees = []
    for _ in range(nt):
        L, = struct.unpack_from("i", blob, p); p += 4
        n = L - 1
        t = {}
        for name in ("feat", "theta", "dleft", "left", "right"):
            t[name] = np.frombuffer(blob, np.int32, n, p).copy(); p += 4 * n
        t["gain"] = np.frombuffer(blob, np.float64, n, p).copy(); p += 8 * n
        t["leaf_value"] = np.frombuffer(blob, np.float64, L, p).copy(); p += 8 * L
        t["leaf_count"] = np.frombuffer(blob, np.int32, L, p).copy(); p += 4 * L
        trees.append(t)
    return feats, trees


def test_hand_computed_two_level_tree():
    """80 rows, f0 in {0,1}, f1 in {0..3}, h = 1, learning_rate = 1, min_data_in_leaf = 20, num_leaves = 4.

        f0=0: f1 in {0,1} (10+10 rows) y = 0      f1 in {2,3} (10+10 rows) y = 4
        f0=1: f1 = 0      (20 rows)    y = 10     f1 in {1,2,3} (7+7+6 rows) y = 20

    mean = 8.5, g = 8.5 - y.  Root: f0 <= 0 gives GL = 260, GR = -260, gain = 2 * 260^2/40 = 3380 (f1 <= 1, the best other
    split, gives 59.5^2/47 + 59.5^2/33 = 182.6).  Right child (G = -260, 40 rows): only f1 <= 0 leaves 20 rows on both sides:
    30^2/20 + 230^2/20 - 260^2/40 = 1000.  Left child (G = 260): only f1 <= 1: 170^2/20 + 90^2/20 - 1690 = 160.
    Best-first: the right child (1000) is split before the left one (160).  Tree::Split numbering: the split leaf keeps its
    index for the left child, the right child gets the next free index: leaves 0 = (f0=0, f1<=1), 1 = (f0=1, f1=0),
    2 = (f0=1, f1>0), 3 = (f0=0, f1>1).  Outputs -G/H = -8.5, +1.5, +11.5, -4.5; AddBias puts the init score 8.5 into the
    first tree: 0, 10, 20, 4."""
    f0 = np.r_[np.zeros(40), np.ones(40)].astype(np.int32)
    f1 = np.r_[[0] * 10, [1] * 10, [2] * 10, [3] * 10, [0] * 20, [1] * 7, [2] * 7, [3] * 6].astype(np.int32)
    y = np.r_[[0.0] * 20, [4.0] * 20, [10.0] * 20, [20.0] * 20]
    vals = np.unique(y)
    m = O.train(np.stack([f0, f1]), [2, 4], np.searchsorted(vals, y).astype(np.int32), len(vals), y_value=vals, objective=2,
                n_estimators=1, learning_rate=1.0, num_leaves=4, max_depth=7, min_data_in_leaf=20)
    feats, (t,) = _parse(m.save())
    assert [f["V"] for f in feats] == [2, 4] and all(f["has_nan"] == 0 for f in feats)          # one bin per code
    assert t["feat"].tolist() == [0, 1, 1] and t["theta"].tolist() == [0, 0, 1]
    assert t["left"].tolist() == [2, ~1, ~0] and t["right"].tolist() == [1, ~2, ~3]
    assert np.allclose(t["gain"], [3380.0, 1000.0, 160.0], rtol=1e-12, atol=0)
    assert np.allclose(t["leaf_value"], [0.0, 10.0, 20.0, 4.0], rtol=0, atol=1e-12)
    assert t["leaf_count"].tolist() == [20, 20, 20, 20]
    assert np.allclose(m.predict(np.stack([f0, f1]))[:, 0], y, atol=1e-12)
    # one more leaf is not allowed by min_data_in_leaf: num_leaves = 31 grows the same tree
    m31 = O.train(np.stack([f0, f1]), [2, 4], np.searchsorted(vals, y).astype(np.int32), len(vals), y_value=vals, objective=2,
                  n_estimators=1, learning_rate=1.0, num_leaves=31, max_depth=7, min_data_in_leaf=20)
    assert m31.save() == m.save()


def _dyadic_regression_table(seed, n=4096, cards=(2, 3, 5, 8, 12, 20, 33), null_col=None):
    """Integer labels and n a power of two: the mean and every first-iteration gradient are multiples of 2^-12."""
    rng = np.random.default_rng(seed)
    z = rng.integers(0, 20, n)
    X = np.stack([(z * (j + 2)) % c for j, c in enumerate(cards)]).astype(np.int32)
    for j, c in enumerate(cards):
        X[j] = np.where(rng.random(n) < 0.15, rng.integers(0, c, n), X[j])
    y = (3 * X[5] % 7 + 2 * X[3] + X[1] * X[0] + (X[6] > 16) * 5 + rng.integers(0, 4, n)).astype(np.float64)
    return X, np.asarray(cards, np.int32), y


def _sk_nodes(pred):
    """(feature, bin threshold, gain) of the internal nodes and values of the leaves, in a canonical (pre-order) order."""
    nodes = pred.nodes
    internal, leaves = [], []

    def walk(i):
        nd = nodes[i]
        if nd["is_leaf"]:
            leaves.append((float(nd["value"]), int(nd["count"])))
            return
        internal.append((int(nd["feature_idx"]), int(nd["bin_threshold"]), float(nd["gain"])))
        walk(nd["left"]); walk(nd["right"])
    walk(0)
    return internal, leaves


def _orc_nodes(t):
    internal, leaves = [], []

    def walk(c):
        if c < 0:
            leaves.append((float(t["leaf_value"][~c]), int(t["leaf_count"][~c])))
            return
        internal.append((int(t["feat"][c]), int(t["theta"][c]), float(t["gain"][c])))
        walk(t["left"][c]); walk(t["right"][c])
    walk(0 if len(t["feat"]) else ~0)
    return internal, leaves


def _same_structure(oi, si, rtol):
    """Same split feature at every node (pre-order) and the same gain.  The threshold BIN may differ where bins of the feature are
    empty inside the node -- several thresholds then describe one partition: LightGBM scans the bins from the right and keeps the
    first maximum (the largest such threshold), sklearn scans from the left (the smallest).  So the oracle's threshold is never
    below sklearn's, and the partition (leaf counts, gains, every prediction -- asserted by the callers) is the same."""
    assert [f for f, _, _ in oi] == [f for f, _, _ in si], "split features differ from sklearn's tree"
    assert all(bo >= bs for (_, bo, _), (_, bs, _) in zip(oi, si)), "threshold bins below sklearn's"
    assert sum(bo == bs for (_, bo, _), (_, bs, _) in zip(oi, si)) >= len(oi) * 3 // 4      # ties are the exception
    assert np.allclose([g for _, _, g in oi], [g for _, _, g in si], rtol=rtol, atol=0)


@pytest.mark.parametrize("seed,leaves,depth,min_leaf,l2", [(3, 31, 7, 20, 0.0), (11, 31, 7, 20, 0.0), (5, 8, 3, 40, 0.0), (7, 63, 6, 5, 2.0)])
def test_first_tree_equals_sklearn_hist_gradient_boosting(seed, leaves, depth, min_leaf, l2):
    from sklearn.ensemble import HistGradientBoostingRegressor
    X, cards, y = _dyadic_regression_table(seed)
    vals = np.unique(y)
    lr = 0.5
    m = O.train(X, cards, np.searchsorted(vals, y).astype(np.int32), len(vals), y_value=vals, objective=2, n_estimators=1,
                learning_rate=lr, num_leaves=leaves, max_depth=depth, min_data_in_leaf=min_leaf, lambda_l2=l2)
    est = HistGradientBoostingRegressor(max_iter=1, learning_rate=lr, max_leaf_nodes=leaves, max_depth=depth, min_samples_leaf=min_leaf,
                                        l2_regularization=l2, max_bins=255, early_stopping=False).fit(X.T.astype(np.float64), y)
    feats, (t,) = _parse(m.save())
    assert [f["V"] for f in feats] == cards.tolist()                       # one bin per code on both sides
    si, sl = _sk_nodes(est._predictors[0][0])
    oi, ol = _orc_nodes(t)
    _same_structure(oi, si, 1e-9)
    assert [c for _, c in ol] == [c for _, c in sl]
    base = float(np.ravel(est._baseline_prediction)[0])
    assert np.allclose([v for v, _ in ol], [base + v for v, _ in sl], rtol=0, atol=1e-9)      # AddBias: the oracle's first tree carries the init score
    assert np.abs(m.predict(X)[:, 0] - est.predict(X.T.astype(np.float64))).max() <= 1e-9


def test_three_iterations_keep_sklearns_structure():
    from sklearn.ensemble import HistGradientBoostingRegressor
    X, cards, y = _dyadic_regression_table(13)
    vals = np.unique(y)
    m = O.train(X, cards, np.searchsorted(vals, y).astype(np.int32), len(vals), y_value=vals, objective=2, n_estimators=3,
                learning_rate=0.5, num_leaves=31, max_depth=7, min_data_in_leaf=20)
    est = HistGradientBoostingRegressor(max_iter=3, learning_rate=0.5, max_leaf_nodes=31, max_depth=7, min_samples_leaf=20,
                                        l2_regularization=0.0, max_bins=255, early_stopping=False).fit(X.T.astype(np.float64), y)
    _, trees = _parse(m.save())
    for it in range(3):
        si, sl = _sk_nodes(est._predictors[it][0])
        oi, ol = _orc_nodes(trees[it])
        _same_structure(oi, si, 1e-9 if it == 0 else 2e-4)   # scores are dyadic only before the first tree
        assert [c for _, c in ol] == [c for _, c in sl]
    # iterations 2+ start from non-dyadic scores: float32 gradients (sklearn) vs 2^-k fixed point (D1) differ by ~1e-6 per row
    assert np.abs(m.predict(X)[:, 0] - est.predict(X.T.astype(np.float64))).max() <= 1e-4


def _balanced_labels(X, K, seed):
    """Labels with a strong dependence on the features and EXACTLY n / K rows per class: the priors are 1 / K, so the init scores and
    every first-iteration gradient / hessian (p - y, p (1 - p) with p = 1 / K) are dyadic for K = 2, 4 -- exact in float32 (sklearn)
    and in the oracle's fixed point alike."""
    rng = np.random.default_rng(seed)
    n = X.shape[1]
    score = 3.0 * (X[3] % 4) + 2.0 * X[1] + (X[5] % 3) + 1.5 * X[0] * (X[2] > 2) + rng.random(n) * 2.5
    order = np.argsort(score, kind="stable")
    y = np.empty(n, np.int32)
    y[order] = np.repeat(np.arange(K, dtype=np.int32), n // K)
    return y


@pytest.mark.parametrize("seed,leaves,depth,min_leaf", [(21, 31, 7, 20), (23, 15, 4, 40), (29, 63, 6, 10)])
def test_binary_first_tree_equals_sklearn_classifier(seed, leaves, depth, min_leaf):
    """The binary objective (`binary_objective.hpp`: labels +-1, response = -label / (1 + exp(label score)), h = |r| (1 - |r|)) against
    scikit-learn's log-loss classifier (g = p - y, h = p (1 - p)): the same numbers.  Balanced classes: init score 0, g = +-0.5,
    h = 0.25, all exact; the first tree must agree at every node, in every leaf value and in every probability."""
    from sklearn.ensemble import HistGradientBoostingClassifier
    X, cards, _ = _dyadic_regression_table(seed)
    y = _balanced_labels(X, 2, seed)
    lr = 0.5
    m = O.train(X, cards, y, 2, objective=0, num_class=2, n_estimators=1, learning_rate=lr, num_leaves=leaves, max_depth=depth, min_data_in_leaf=min_leaf)
    est = HistGradientBoostingClassifier(max_iter=1, learning_rate=lr, max_leaf_nodes=leaves, max_depth=depth, min_samples_leaf=min_leaf,
                                         l2_regularization=0.0, max_bins=255, early_stopping=False).fit(X.T.astype(np.float64), y)
    _, (t,) = _parse(m.save())
    si, sl = _sk_nodes(est._predictors[0][0])
    oi, ol = _orc_nodes(t)
    assert len(oi) >= 7
    _same_structure(oi, si, 1e-9)
    assert [c for _, c in ol] == [c for _, c in sl]
    assert np.allclose([v for v, _ in ol], [v for v, _ in sl], rtol=0, atol=1e-9)             # init score 0: nothing to add
    assert np.abs(m.predict(X)[:, 1] - est.predict_proba(X.T.astype(np.float64))[:, 1]).max() <= 1e-9     # predict: [n][2] probabilities


@pytest.mark.parametrize("seed", [31, 33, 37, 41])
def test_multiclass_first_trees_match_sklearn_up_to_lightgbms_hessian_factor(seed):
    """Softmax, K = 4 balanced classes: p = 1/4, g = 1/4 or -3/4, p (1 - p) = 3/16 -- exact on both sides.  LightGBM multiplies the
    softmax hessian by K / (K - 1) (`multiclass_objective.hpp`, factor_), scikit-learn does not; a uniform hessian factor leaves every
    arg-max of the split search where it is, so the K first trees must have scikit-learn's structure node for node, with gains and
    leaf values scaled by (K - 1) / K = 3/4 and the init score log(1/4) added to the leaves (AddBias).
    Seed 31 holds a genuine TIE: in one class tree two thresholds of the last split give different partitions (48 | 24 and 24 | 48 rows)
    with the same gain 3.0.  LightGBM scans the bins from the right and keeps the first maximum, scikit-learn from the left: the oracle
    must show LightGBM's choice (the larger threshold) -- the only difference allowed."""
    from sklearn.ensemble import HistGradientBoostingClassifier
    K, lr = 4, 0.5
    X, cards, _ = _dyadic_regression_table(seed)
    y = _balanced_labels(X, K, seed)
    m = O.train(X, cards, y, K, objective=1, num_class=K, n_estimators=1, learning_rate=lr, num_leaves=31, max_depth=7, min_data_in_leaf=20)
    est = HistGradientBoostingClassifier(max_iter=1, learning_rate=lr, max_leaf_nodes=31, max_depth=7, min_samples_leaf=20,
                                         l2_regularization=0.0, max_bins=255, early_stopping=False).fit(X.T.astype(np.float64), y)
    _, trees = _parse(m.save())
    assert len(trees) == K
    ties = 0
    for k in range(K):
        si, sl = _sk_nodes(est._predictors[0][k])
        oi, ol = _orc_nodes(trees[k])
        assert len(oi) >= 5
        _same_structure([(f, b, g / 0.75) for f, b, g in oi], si, 1e-9)
        if [c for _, c in ol] == [c for _, c in sl]:
            assert np.allclose([v - np.log(0.25) for v, _ in ol], [0.75 * v for v, _ in sl], rtol=0, atol=1e-9)
        else:   # equal gains at every node (checked above), a larger threshold at the tied one, the same rows in total
            ties += 1
            assert sum(c for _, c in ol) == sum(c for _, c in sl)
            assert sum(bo > bs for (_, bo, _), (_, bs, _) in zip(oi, si)) >= 1
    assert ties == (1 if seed == 31 else 0)


@pytest.mark.parametrize("seed", [3, 5, 7, 11])
def test_missing_values_first_tree_is_sklearns_partition(seed):
    """NULL cells (code -1: the last, "NaN" bin of the feature, D3) against scikit-learn's missing-value handling: both try the
    missing rows on either side of every threshold and keep the better one.  Which child is called left can differ where a split
    separates the NULLs from everything else, so the comparison is on what the tree IS: the same gains, the same leaves (value, rows)
    and the same prediction for every training row."""
    from sklearn.ensemble import HistGradientBoostingRegressor
    X, cards, y = _dyadic_regression_table(seed)
    rng = np.random.default_rng(seed + 100)
    Xn = X.copy()
    for f, frac in ((3, 0.08), (5, 0.03)):      # NULLs that carry signal: mostly rows with a large label
        Xn[f] = np.where(rng.random(X.shape[1]) < np.where(y > np.median(y), 2 * frac, 0.2 * frac), -1, X[f])
    vals = np.unique(y)
    m = O.train(Xn, cards, np.searchsorted(vals, y).astype(np.int32), len(vals), y_value=vals, objective=2, n_estimators=1, learning_rate=0.5,
                num_leaves=31, max_depth=7, min_data_in_leaf=20)
    Xs = Xn.T.astype(np.float64)
    Xs[Xs < 0] = np.nan
    est = HistGradientBoostingRegressor(max_iter=1, learning_rate=0.5, max_leaf_nodes=31, max_depth=7, min_samples_leaf=20, l2_regularization=0.0,
                                        max_bins=255, early_stopping=False).fit(Xs, y)
    feats, (t,) = _parse(m.save())
    assert [f["has_nan"] for f in feats] == [0, 0, 0, 1, 0, 1, 0]
    si, sl = _sk_nodes(est._predictors[0][0])
    oi, ol = _orc_nodes(t)
    assert any(f in (3, 5) for f, _, _ in oi), "no split on a feature with NULLs: the case tests nothing"
    assert np.allclose(sorted(g for _, _, g in oi), sorted(g for _, _, g in si), rtol=1e-9, atol=0)
    base = float(np.ravel(est._baseline_prediction)[0])
    assert np.allclose(sorted(ol), sorted((base + v, c) for v, c in sl), rtol=0, atol=1e-9)
    assert np.abs(m.predict(Xn)[:, 0] - est.predict(Xs)).max() <= 1e-9
