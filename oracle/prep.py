"""CPU restatement (numpy) of the relational steps either side of the repair models, in code space.
TEST INFRASTRUCTURE ONLY -- the product path is csrc/rgbm_prep.hip; nothing under the package imports this.

A table is an int32 array ``codes[c][n]`` (column-major, -1 = NULL); a cell is (row position, column index).
Each function cites the reference code it follows (read-only study copy under /root/reference):

  detect_nulls        src/main/scala/org/apache/spark/api/python/ErrorDetectorApi.scala:128-157
  constraint_rows     ErrorDetectorApi.scala:189-244 + python/DenialConstraints.scala:66-225 (two-tuple EQ / IQ predicates)
  null_cells          src/main/scala/org/apache/spark/api/python/RepairApi.scala:171-211
  rows_of_cells       python/repair/model.py:549-553
  count_codes         python/repair/train.py:39-40,105 (class weights need the rows per class)
  encode_dictionaries python/repair/model.py:701-729 (replaced by sorted-rank codes)
  flatten_cells       src/main/scala/org/apache/spark/api/python/RepairMiscApi.scala:41-49 (flattenTable) + model.py:1398-1401
  top_k_pmf           python/repair/model.py:1174-1225

Pinned against the reference's own golden rows in tests/test_prep_oracle.py (RepairSuite.scala:205-235,
ErrorDetectorSuite.scala:118-202, RepairMiscSuite.scala:49-64) -- they are stated there in value space and run
through repair.encode to get here.
"""
import numpy as np


def detect_nulls(codes, cols):
    """NULL cells of ``cols``, ordered by position in ``cols`` then ascending row.

    The reference emits one `SELECT rowId, attr WHERE attr IS NULL` per target attribute and UNION ALLs them
    (ErrorDetectorApi.scala:139-146); a Spark result has no order, this one fixes it.
    """
    rows, out_cols = [], []
    for c in cols:
        r = np.flatnonzero(codes[c] < 0)
        rows.append(r.astype(np.int64))
        out_cols.append(np.full(len(r), c, np.int32))
    if not rows:
        return np.zeros(0, np.int64), np.zeros(0, np.int32)
    return np.concatenate(rows), np.concatenate(out_cols)


def constraint_rows(codes, eq_cols, iq_col):
    """Ascending rows t1 for which a row t2 exists with  AND_x (t1.x <=> t2.x)  AND  NOT (t1.y <=> t2.y).

    ErrorDetectorApi.scala:214-224 runs  `SELECT t1.rowId FROM input t1 WHERE EXISTS (SELECT .. FROM input t2 WHERE preds)`
    and DenialConstraints.scala:75-79 renders EQ as `l <=> r` and IQ as `NOT(l <=> r)`: NULL-safe on both sides, so NULLs
    form a group of their own and NULL is a Y value of its own (golden: ErrorDetectorSuite.scala:118-160, rows "1","2","3").
    """
    n = codes.shape[1]
    key = np.zeros(n, np.int64)
    for c in eq_cols:
        key = key * (int(codes[c].max(initial=-1)) + 2) + (codes[c].astype(np.int64) + 1)
    _, gid = np.unique(key, return_inverse=True)
    ng = int(gid.max()) + 1 if n else 0
    y = codes[iq_col].astype(np.int64) + 1
    span = int(y.max(initial=0)) + 1
    pair = np.unique(gid.astype(np.int64) * span + y)
    distinct = np.bincount((pair // span).astype(np.int64), minlength=ng)
    return np.flatnonzero(distinct[gid] > 1).astype(np.int64)


def constraint_cells(codes, eq_cols, iq_col, cell_cols):
    """Violating rows x cell_cols, column-major (`explode(array(attrs))`, ErrorDetectorApi.scala:217)."""
    rows = constraint_rows(codes, eq_cols, iq_col)
    if len(cell_cols) == 0:
        return rows, None
    return np.tile(rows, len(cell_cols)), np.repeat(np.asarray(cell_cols, np.int32), len(rows))


def null_cells(codes, rows, cols, target_cols):
    """`IF(array_contains(errors, attr), NULL, attr)` for the target attributes only (RepairApi.scala:193-199); error cells
    that name a row or column outside the table disappear in the LEFT OUTER JOIN (RepairApi.scala:202-206)."""
    out = codes.copy()
    c, n = codes.shape
    tg = set(int(t) for t in target_cols)
    for r, cc in zip(np.asarray(rows, np.int64), np.asarray(cols, np.int64)):
        if 0 <= r < n and 0 <= cc < c and int(cc) in tg:
            out[cc, r] = -1
    return out


def rows_of_cells(n, rows):
    """Left-semi join of the table with the error cells' row ids (model.py:549-553): ascending, distinct."""
    r = np.asarray(rows, np.int64)
    r = r[(r >= 0) & (r < n)]
    return np.unique(r)


def count_codes(codes, col, n_codes):
    v = codes[col]
    ok = (v >= 0) & (v < n_codes)
    return np.bincount(v[ok], minlength=n_codes).astype(np.int64), int((~ok).sum())


def encode_dictionaries(indices, remaps):
    """indices [c][n] (Arrow dictionary indices, < 0 = NULL) -> codes through per-column remap tables."""
    out = np.full(indices.shape, -1, np.int32)
    for c in range(indices.shape[0]):
        m = np.asarray(remaps[c], np.int32)
        ok = (indices[c] >= 0) & (indices[c] < len(m))
        out[c, ok] = m[indices[c][ok]]
    return out


def flatten_cells(labels, probs, target_cols, dirty_rows, cell_rows, cell_cols):
    """(row, attribute) error cells -> (repaired code, probability): the reference flattens the repaired dirty frame to
    (rowId, attribute, value) and inner-joins it with the error cells (RepairMiscApi.scala:41-49, model.py:1398-1401).
    labels/probs: [T][D] over the ascending ``dirty_rows``; cells of non-target columns or clean rows drop out (-1 / nan)."""
    tpos = {int(c): i for i, c in enumerate(target_cols)}
    pos = np.searchsorted(dirty_rows, cell_rows)
    pos = np.clip(pos, 0, max(len(dirty_rows) - 1, 0))
    hit = (len(dirty_rows) > 0) & (np.asarray(dirty_rows)[pos] == cell_rows) if len(dirty_rows) else np.zeros(len(cell_rows), bool)
    out_l = np.full(len(cell_rows), -1, np.int32)
    out_p = np.full(len(cell_rows), np.nan, np.float64)
    for i, (r, c) in enumerate(zip(cell_rows, cell_cols)):
        if hit[i] and int(c) in tpos:
            out_l[i] = labels[tpos[int(c)], pos[i]]
            if probs is not None:
                out_p[i] = probs[tpos[int(c)], pos[i]]
    return out_l, out_p


def top_k_pmf(proba, top_k, threshold):
    """Per row: classes sorted by descending probability, then `prob > threshold`, then the first top_k
    (model.py:1196-1212: `array_sort` with a comparator that returns 0 on ties -- a stable sort, so ties keep class
    order -- followed by `slice(filter(pmf, x -> x.prob > threshold), 1, top_k)`).
    Returns (classes [n][top_k] int32, -1 padded; probs [n][top_k], 0 padded)."""
    n, K = proba.shape
    cls = np.full((n, top_k), -1, np.int32)
    pr = np.zeros((n, top_k), np.float64)
    for i in range(n):
        order = sorted(range(K), key=lambda j: -proba[i, j])
        keep = [j for j in order if proba[i, j] > threshold][:top_k]
        cls[i, :len(keep)] = keep
        pr[i, :len(keep)] = proba[i, keep]
    return cls, pr
