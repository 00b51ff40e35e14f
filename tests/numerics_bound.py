"""How close the product's numerics (numerics v2: LightGBM's float32 g / h per row, EXACT integer histogram sums)
are to LightGBM 3.3.1's own arithmetic (the same float32 g / h, double histogram sums in row order) -- the two modes of
oracle/rgbm_oracle_train.inc.

For a table and a target attribute: both modes train on the rows whose target cell is non-NULL with the reference's fixed
parameters (python/repair/train.py:102-131: 300 iterations, learning rate 0.01, max_depth 7, class_weight balanced, ...) and
LightGBM's defaults for the searched ones (num_leaves 31, min_child_samples 20, ...), then score the rows to repair.  Reported:
arg-max disagreements, max |dp| over every (cell, class), and the first boosting iteration at which any tree differs in STRUCTURE
(split feature / threshold / child links; None = all trees of all iterations identical).  north_star's bar: arg-max identical,
probabilities within 1e-4.

History (profiles/r03a_numerics_v1_02_vs_float32.json): the round-1/2 numerics -- gradients quantised to 2^-20 of one per-model bound
(D1) and the hessian recomputed from the quantised gradient (D1b) -- FAILED this bar: 13 of 190 labels on hospital `Score`, 8 of 91
on `Sample`, max |dp| 0.17; 5e-2 on the K = 64 synthetic target.  A near-exact fixed point (2^-42) still differed from float32
LightGBM by up to 7e-3, because any difference in a gradient flips a near-tie between two splits sooner or later and the models
diverge from there.  Only the SAME float32 gradients with exact sums reproduce LightGBM's trees -- which is what v2 does.

Used by tests/test_numerics_bound.py (sizes that finish in seconds) and tools/numerics_bound.py (the full table of DESIGN.md 3).
"""
import struct

import numpy as np

from oracle import oracle as O

FIXED = dict(n_estimators=300, learning_rate=0.01, max_depth=7, max_bin=255, lambda_l1=0.0, min_gain_to_split=0.0, seed=42,
             num_leaves=31, min_data_in_leaf=20, min_sum_hessian_in_leaf=1e-3, lambda_l2=0.0, bagging_fraction=1.0, bagging_freq=0,
             feature_fraction=1.0)


def parse_trees(blob):
    hdr = struct.unpack_from("7i", blob, 0)
    ver, K, n_iter, F = hdr[1], hdr[4], hdr[5], hdr[6]
    p = 28
    for _ in range(F):
        _, V, _ = struct.unpack_from("3i", blob, p); p += 12 + 4 * V
        if ver == 2:
            nw, = struct.unpack_from("i", blob, p); p += 4 + 4 * nw
    trees = []
    for _ in range(K * n_iter):
        L, = struct.unpack_from("i", blob, p); p += 4
        n = L - 1
        t = {}
        for name in ("feat", "theta", "dleft", "left", "right"):
            t[name] = np.frombuffer(blob, np.int32, n, p); p += 4 * n
        t["gain"] = np.frombuffer(blob, np.float64, n, p); p += 8 * n
        t["leaf_value"] = np.frombuffer(blob, np.float64, L, p); p += 8 * L
        t["leaf_count"] = np.frombuffer(blob, np.int32, L, p); p += 4 * L
        trees.append(t)
    return K, n_iter, trees


def first_differing_iteration(blob_a, blob_b):
    """First boosting iteration with a tree whose structure differs (None: all trees have the same splits), and the largest
    |leaf value difference| over the iterations before it."""
    Ka, na, ta = parse_trees(blob_a)
    Kb, nb, tb = parse_trees(blob_b)
    assert Ka == Kb
    worst = 0.0
    for it in range(min(na, nb)):
        for k in range(Ka):
            a, b = ta[it * Ka + k], tb[it * Ka + k]
            same = len(a["feat"]) == len(b["feat"]) and all(np.array_equal(a[n], b[n]) for n in ("feat", "theta", "dleft", "left", "right"))
            if not same:
                return it, worst
            if len(a["leaf_value"]):
                worst = max(worst, float(np.abs(a["leaf_value"] - b["leaf_value"]).max()))
    return (None if na == nb else min(na, nb)), worst


def balanced(y, n_classes):
    cnt = np.bincount(y, minlength=n_classes).astype(np.float64)
    present = int((cnt > 0).sum())
    w = np.zeros(n_classes, np.float64)
    w[cnt > 0] = len(y) / (present * cnt[cnt > 0])
    return w


def _pair(pa, pb, ba, bb, regression):
    it, leaf_diff = first_differing_iteration(ba, bb)
    d = dict(first_diff_iteration=it, max_leaf_value_diff_before=leaf_diff)
    if regression:
        scale = max(float(np.abs(pb).max()), 1e-300) if len(pb) else 1.0
        d.update(max_abs_diff=float(np.abs(pa - pb).max()) if len(pa) else 0.0, max_rel_diff=float(np.abs(pa - pb).max() / scale) if len(pa) else 0.0,
                 rounded_mismatch=int((np.round(pa) != np.round(pb)).sum()))
    else:
        d.update(label_mismatch=int((pa.argmax(1) != pb.argmax(1)).sum()) if len(pa) else 0, max_dp=float(np.abs(pa - pb).max()) if len(pa) else 0.0)
    return d


def compare_target(codes, n_codes, target, feats, train_rows, score_rows, regression_values=None, threads=1, perm=True, **over):
    """codes [C][N] int32 (-1 = NULL).  Trains on `train_rows` of `target`, scores `score_rows`, in three ways:
         spec      the product's numerics (float32 g / h, exact integer sums)
         f32       LightGBM's arithmetic (float32 g / h, double sums in row order)
         f32_perm  f32 on a PERMUTATION of the training rows: same data, another summation order -- what LightGBM's own result
                   depends on (row order / thread count)
       and reports the pairs spec~f32 and f32~f32_perm."""
    O.lib().orc_set_threads(int(threads))
    try:
        tr = np.asarray(train_rows)
        ny = int(n_codes[target])
        params = dict(FIXED, **over)
        y_all = codes[target][tr]
        if regression_values is not None:   # class_weight reaches regressors too (LGBMModel.fit)
            kw = dict(y_value=np.asarray(regression_values, np.float64), class_weight=balanced(y_all, ny), objective=2)
        else:
            kw = dict(class_weight=balanced(y_all, ny), objective=0 if ny <= 2 else 1, num_class=max(ny, 2))
        Xs = np.ascontiguousarray(codes[feats][:, score_rows])

        def fit(numerics, rows):
            m = O.train(np.ascontiguousarray(codes[feats][:, rows]), n_codes[feats], np.ascontiguousarray(codes[target][rows]), ny, numerics=numerics, **kw, **params)
            return m.predict(Xs), m.save(), m.info()["n_iter"]

        r_spec, r_f32 = fit("spec", tr), fit("lightgbm_f32", tr)
        reg = regression_values is not None
        out = dict(target=int(target), K=ny, train_rows=int(len(tr)), cells=int(len(score_rows)), iterations=(r_spec[2], r_f32[2]),
                   spec_vs_f32=_pair(r_spec[0], r_f32[0], r_spec[1], r_f32[1], reg))
        if perm:
            r_perm = fit("lightgbm_f32", tr[np.random.default_rng(7).permutation(len(tr))])
            out["f32_vs_f32_perm"] = _pair(r_f32[0], r_perm[0], r_f32[1], r_perm[1], reg)
        if not reg and len(score_rows):   # how decisive the closest call is: gap between the two best classes, minimum over the cells
            top2 = np.sort(r_f32[0], axis=1)[:, -2:]
            out["min_top2_gap"] = float((top2[:, 1] - top2[:, 0]).min())
        return out
    finally:
        O.lib().orc_set_threads(1)


def frame_case(df, row_id, targets, error_cells=None, numeric_targets=(), threads=1, perm=True, **over):
    """A DataFrame through repair.encode (the boundary's label encoding): error cells (default: the NULL cells) are NULLed, every
    target trains on its non-NULL rows and scores the rows of its error cells."""
    from repair.encode import TableEncoder
    cols = [c for c in df.columns if c != row_id]
    work = df.copy()
    if error_cells is not None:
        pos = {v: i for i, v in enumerate(work[row_id].tolist())}
        for r, a in zip(error_cells[row_id].tolist(), error_cells["attribute"].tolist()):
            if a in cols and r in pos:
                work.loc[work.index[pos[r]], a] = None
    enc = TableEncoder(work, cols)
    codes = enc.encode(work)
    n_codes = enc.n_codes
    res = []
    for t in targets:
        j = cols.index(t)
        score = np.flatnonzero(codes[j] < 0)
        train = np.flatnonzero(codes[j] >= 0)
        if len(score) == 0 or len(train) == 0 or (t not in numeric_targets and n_codes[j] < 2):
            continue
        feats = [i for i in range(len(cols)) if i != j]
        r = compare_target(codes, n_codes, j, feats, train, score,
                           regression_values=enc.dicts[t].values.astype(np.float64) if t in numeric_targets else None, threads=threads, perm=perm, **over)
        r["attribute"] = t
        res.append(r)
    return res


def iteration_digests(blob):
    """md5 of every boosting iteration's K trees (split features, thresholds, default directions, child links, gains, leaf values,
    leaf counts -- every byte the model stores for them), in iteration order: the handle by which tests/golden/bench_job_digests.json
    pins a long training run without storing the model."""
    import hashlib
    K, n_iter, trees = parse_trees(blob)
    out = []
    for it in range(n_iter):
        h = hashlib.md5()
        for k in range(K):
            t = trees[it * K + k]
            for name in ("feat", "theta", "dleft", "left", "right", "gain", "leaf_value", "leaf_count"):
                h.update(np.ascontiguousarray(t[name]).tobytes())
        out.append(h.hexdigest())
    return out
