"""`RepairModel.run()` on a label-encoded table that stays resident in HBM from detection to the repaired cells.

The reference walks  detect errors -> convertErrorCellsToNull -> clean/dirty split -> per-attribute models ->
chained repair -> flatten + join with the error cells  through Spark SQL and pandas (python/repair/model.py:1293-1419).
Here every step is a call on the engine's table object (csrc/rgbm_prep.hip for the relational steps,
csrc/rgbm.hip for the models); the host only merges the (small) cell lists.  The engine is an argument so the same
job logic runs on the CPU oracle engine in the tests.

Cells are (row position, column index); values are codes (-1 = NULL).  `repair.encode` / `Table.from_dictionaries`
map between values and codes.
"""
import time

import numpy as np

from repair.engine import run_job


def _merge_cells(n, parts):
    """Union of cell lists, ordered by (column, row), duplicates removed (detectors overlap: model.py `_detect_errors`
    + ErrorModel concatenates the detectors' frames and drops duplicates)."""
    if not parts:
        return np.zeros(0, np.int64), np.zeros(0, np.int32)
    key = np.concatenate([c.astype(np.int64) * n + r for r, c in parts])
    key = np.unique(key)
    return key % n, (key // n).astype(np.int32)


def detect_error_cells(table, targets, constraints=(), detect_nulls=True, error_cells=None):
    """Error cells of the target attributes.

    constraints : [(eq_cols, iq_col)] -- denial constraints  EQ(X1)..EQ(Xm) & IQ(Y)  (ErrorDetectorApi.scala:189-244); a
                  violating row contributes the constraint's attributes that are targets (`attrs`, line 211)
    error_cells : (rows, cols) given by the caller (RepairModel.setErrorCells); non-target attributes are dropped
    """
    tg = [int(t) for t in targets]
    parts = []
    if error_cells is not None:
        r, c = np.asarray(error_cells[0], np.int64), np.asarray(error_cells[1], np.int32)
        keep = np.isin(c, tg) & (r >= 0) & (r < table.n)
        parts.append((r[keep], c[keep]))
    if detect_nulls and tg:
        parts.append(table.detect_nulls(tg))
    for eq, iq in constraints:
        attrs = []
        for a in list(eq) + [iq]:
            if a in tg and a not in attrs:
                attrs.append(a)
        if attrs:
            parts.append(table.detect_constraint(list(eq), int(iq), cell_cols=attrs))
    return _merge_cells(table.n, parts)


def repair_table(engine, table, targets, base_params, constraints=(), detect_nulls=True, error_cells=None,
                 want_pmf=False, top_k=32, threshold=0.0, want_stats=False):
    """Detect, NULL out, split, train, repair, shape.  ``table`` is modified in place (error cells become NULL).

    Returns dict(rows, cols, current, repaired, prob[, pmf_class, pmf_prob, current_prob], dirty_rows, models, times, stats):
    one entry per error cell, ordered by (column, row).
    """
    t0 = time.perf_counter()
    targets = [int(t) for t in targets]
    n_codes = np.asarray(table.n_codes, np.int32)
    rows, cols = detect_error_cells(table, targets, constraints, detect_nulls, error_cells)
    t_detect = time.perf_counter() - t0
    t0 = time.perf_counter()
    current = table.read_cells(rows, cols)
    table.null_cells(rows, cols, targets)                       # convertErrorCellsToNull (RepairApi.scala:171-211)
    dirty_rows = table.rows_of_cells(rows)                      # model.py:549-553
    out = dict(rows=rows, cols=cols, current=current, dirty_rows=dirty_rows, models={}, stats=[])
    if len(rows) == 0:
        out.update(repaired=np.zeros(0, np.int32), prob=np.zeros(0, np.float64), times=dict(detect=t_detect, prepare=time.perf_counter() - t0))
        return out
    dirty_tab = table.gather_rows(dirty_rows)
    pmf_tab = table.gather_rows(dirty_rows) if want_pmf else None      # stays un-repaired: pmf mode does not chain
    label_counts = {}
    for t in targets:
        cnt, _ = table.count_codes(t)
        if int((cnt > 0).sum()) < 2:
            raise ValueError("target column %d has fewer than two classes among its non-NULL rows; the reference short-cuts such "
                             "attributes with a constant model (model.py:1008-1017) -- drop it from `targets`" % t)
        label_counts[t] = cnt
    t_prep = time.perf_counter() - t0
    res = run_job(engine, table, dirty_tab, n_codes, targets, label_counts, base_params, want_stats=want_stats)
    # flatten + join with the error cells (RepairMiscApi.scala:41-49, model.py:1398-1401)
    t0 = time.perf_counter()
    tpos = np.full(table.c, -1, np.int64)
    tpos[targets] = np.arange(len(targets))
    pos = np.searchsorted(dirty_rows, rows)
    repaired = res["labels"][tpos[cols], pos].astype(np.int32)
    prob = res["probs"][tpos[cols], pos] if res["probs"] is not None else None
    out.update(repaired=repaired, prob=prob, models=res["models"], stats=res["stats"])
    if want_pmf:
        pc = np.full((len(rows), top_k), -1, np.int32)
        pp = np.zeros((len(rows), top_k), np.float64)
        cp = np.zeros(len(rows), np.float64)
        for t in targets:
            sel = np.flatnonzero(cols == t)
            if len(sel) == 0:
                continue
            feats = [c for c in range(table.c) if c != t]
            model = engine.load_model(res["models"][t])
            # NULL cells of t in the dirty frame: a superset of this target's error cells when NULL detection is off
            drows, _ = pmf_tab.detect_nulls([t])
            j = np.searchsorted(dirty_rows[drows], rows[sel])
            cur_for = np.full(len(drows), -1, np.int32)
            cur_for[j] = current[sel]                          # the value the cell held (model.py:1196-1199: its probability)
            _, dcls, dpr, dcp = pmf_tab.repair_pmf(model, t, feats, top_k=top_k, threshold=threshold, cur_codes=cur_for)
            pc[sel], pp[sel], cp[sel] = dcls[j], dpr[j], dcp[j]
        out.update(pmf_class=pc, pmf_prob=pp, current_prob=cp)
    t_shape = time.perf_counter() - t0
    times = dict(res["times"])
    times.update(detect=t_detect, prepare=t_prep, shape=t_shape)
    out["times"] = times
    return out
