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
from repair.utils import setup_logger

_logger = setup_logger()


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


# searched parameter (sklearn alias, train.py:148-156) -> LightGBM core name of rgbm_params
_POINT_TO_CORE = dict(num_leaves="num_leaves", subsample="bagging_fraction", subsample_freq="bagging_freq", colsample_bytree="feature_fraction",
                      min_child_samples="min_data_in_leaf", min_child_weight="min_sum_hessian_in_leaf", reg_lambda="lambda_l2")


def search_on_table(engine, table, t, n_codes, base_params, opts, y_value=None, rows=None):
    """The hyper-parameter search of one target (train.py:133-209) on RESIDENT tables: every CV fold is a pair of device row gathers
    of the training table (no pandas slicing, no re-encoding, no upload per fit -- the value-space search pays all three for each
    of its >= 150 fits), trained and scored through the table API; the folds of `model.hp.batch_size` evaluations are in flight
    together, one HIP stream per fit.  Same points (RandomState(42)), same folds (StratifiedKFold / KFold shuffled with the
    evaluation index) and same stopping rule as repair.train.run_search -- it IS that loop.  Returns core parameters of the best point."""
    from sklearn.metrics import f1_score, mean_squared_error
    from sklearn.model_selection import KFold, StratifiedKFold
    from repair.engine import balanced_class_weight, model_params
    from repair.train import _opt_n_splits, run_search
    from repair.utils import get_option_value
    col = table.read_column(t)
    rows = np.flatnonzero(col >= 0) if rows is None else np.asarray(rows, np.int64)
    y = col[rows]
    feats = [c for c in range(table.c) if c != t]
    discrete = y_value is None
    n_splits = int(get_option_value(opts, *_opt_n_splits))

    def folds(seed):
        cv = StratifiedKFold(n_splits=n_splits, shuffle=True, random_state=seed) if discrete else KFold(n_splits=n_splits, shuffle=True, random_state=seed)
        return list(cv.split(np.zeros(len(rows)), y))

    def core(point):
        p = dict(base_params)
        for k, v in point.items():
            p[_POINT_TO_CORE[k]] = int(v) if k in ("num_leaves", "subsample_freq", "min_child_samples") else float(v)
        return p

    def fold_score(point, fold):
        tr, va = fold
        ttab, vtab = table.gather_rows(np.sort(rows[tr])), table.gather_rows(np.sort(rows[va]))
        cnt = ttab.count_codes(t)[0]
        m = engine.train(ttab, t, feats, balanced_class_weight(cnt), model_params(int(n_codes[t]), core(point), continuous=not discrete), y_value=y_value)
        lab, pr = engine.repair_chain(vtab, [m], [t], [feats], 0, vtab.n)
        truth = vtab.read_column(t)
        if discrete:
            return float(f1_score(truth, lab[0], average="macro"))
        return -float(mean_squared_error(np.asarray(y_value, np.float64)[truth], pr[0]))

    def score(m, vtab):
        lab, pr = engine.repair_chain(vtab, [m], [t], [feats], 0, vtab.n)
        truth = vtab.read_column(t)
        if discrete:
            return float(f1_score(truth, lab[0], average="macro"))
        return -float(mean_squared_error(np.asarray(y_value, np.float64)[truth], pr[0]))

    def batch_scores(jobs):
        """All fold fits of a batch of evaluations through ONE batched training call (engine.train_many ->
        rgbm_table_train_batch); a fit that fails yields its exception, like the fold_score future would."""
        fits, truths = [], []
        for point, (tr, va) in jobs:
            rtr, rva = np.sort(rows[tr]), np.sort(rows[va])
            ttab, vtab = table.gather_rows(rtr), table.gather_rows(rva)
            cnt = np.bincount(col[rtr], minlength=int(n_codes[t]))        # the fold's label counts (the host holds the column already)
            fits.append((ttab, t, feats, balanced_class_weight(cnt), dict(model_params(int(n_codes[t]), core(point), continuous=not discrete), want_model=False), y_value, vtab))
            truths.append(col[rva])
        out = []
        for res, truth in zip(engine.train_many(fits), truths):     # the validation rows were scored while the fits trained
            try:
                if isinstance(res, Exception):
                    out.append(res)
                elif discrete:
                    out.append(float(f1_score(truth, res[1], average="macro")))
                else:
                    out.append(-float(mean_squared_error(np.asarray(y_value, np.float64)[truth], res[2])))
            except Exception as e:   # noqa: BLE001
                out.append(e)
        return out

    use_batch = hasattr(engine, "train_many") and table.n <= engine.small_rows()
    point, _, _ = run_search(opts, folds, fold_score, batch_scores=batch_scores if use_batch else None)
    return {_POINT_TO_CORE[k]: (int(v) if k in ("num_leaves", "subsample_freq", "min_child_samples") else float(v)) for k, v in point.items()}


def distinct_rows(codes, n_codes, max_mult=255):
    """The DISTINCT rows of a label-encoded table and how often each occurs: codes [C][N] int32 (-1 = NULL) ->
    (distinct [C][M] int32, mult [M] uint8, inverse [N] int64: the distinct row every original row maps to).
    A row that occurs more than `max_mult` times is kept as several rows whose multiplicities add up (the trainer carries a multiplicity in one byte).
    What it is for: `_native.Table(distinct, n_codes).set_row_multiplicity(mult)` trains byte for byte the models of the whole table
    (identical rows take identical paths and gradients in every tree; all sums are exact integers) at the cost of its distinct rows -- a quarter
    of the rows on the categorical tables this plugin repairs (the synthetic 10M x 16 benchmark table holds 2.48M distinct rows).  A VARIANT of
    the workload: bench.py --dedup reports it apart from the row-for-row line.  The reference's frames hold every row (model.py:768-815)."""
    codes = np.ascontiguousarray(codes, np.int32)
    C, N = codes.shape
    # mixed-radix keys over (code + 1) in [0, n_codes]: as few uint64 words per row as the radices need
    words, cur, room = [], np.zeros(N, np.uint64), 1
    for c in range(C):
        radix = int(n_codes[c]) + 1
        if room * radix >= (1 << 63):
            words.append(cur); cur, room = np.zeros(N, np.uint64), 1
        cur = cur * np.uint64(radix) + (codes[c] + 1).astype(np.uint64)
        room *= radix
    words.append(cur)
    if len(words) == 1:
        _, first, inverse, counts = np.unique(words[0], return_index=True, return_inverse=True, return_counts=True)
    else:
        key = np.ascontiguousarray(np.stack(words, axis=1)).view([("w%d" % i, np.uint64) for i in range(len(words))]).reshape(N)
        _, first, inverse, counts = np.unique(key, return_index=True, return_inverse=True, return_counts=True)
    reps = (counts + max_mult - 1) // max_mult                      # distinct rows kept more than once
    rows = np.repeat(first, reps)
    mult = np.full(len(rows), max_mult, np.int64)
    last = np.cumsum(reps) - 1
    mult[last] = counts - (reps - 1) * max_mult
    start = np.concatenate([[0], np.cumsum(reps)[:-1]])
    return np.ascontiguousarray(codes[:, rows]), mult.astype(np.uint8), start[inverse].astype(np.int64)


class NotResidentEligible(ValueError):
    """The run is not the plain per-attribute model loop after all (only known once the error cells are NULLed): a discrete target
    is left with fewer than two classes, a continuous one with no value.  The reference short-cuts those with `PoorModel`
    (model.py:1008-1017, 779-783); `RepairModel._run` catches this and takes the value-space path, which has that short-cut."""


class DeadClasses(NotResidentEligible):
    """Detection ran on the device and found error cells that hold the ONLY occurrences of some values of a target attribute: the
    dictionaries were built before the cells were known, so those values would stay behind as classes without rows.  Carries the
    detected cells (row positions, column indices): `repair_frame` NULLs them in the dictionary indices, drops the dead values,
    uploads again and goes on WITHOUT detecting a second time (and without leaving the resident path)."""

    def __init__(self, msg, rows, cols):
        super().__init__(msg)
        self.rows, self.cols = np.asarray(rows, np.int64), np.asarray(cols, np.int32)


class UnseenCategories(NotResidentEligible):
    """A dirty row holds a categorical feature value that none of the target's training rows has (see `unseen_categories`)."""


def unseen_categories(table, dirty_tab, targets, ordered_cols=(), train_tables=None):
    """[(target, feature, codes)]: categorical feature codes that occur in the rows a target model has to repair but in none of the
    rows it trains on.  The value-space path gives such a value the code -1 (the model's own dictionary does not hold it: it is
    MISSING for that model); the resident table codes it like any other value and the trees would bin it next to its dictionary
    neighbours.  Callers that promise the value-space result (RepairModel.run) check this and fall back."""
    C = table.c
    d = np.stack([dirty_tab.read_column(c) for c in range(C)])
    total = [table.count_codes(c)[0] for c in range(C)]
    out = []
    for t in targets:
        need = d[t] < 0                                   # the rows this model's prediction is used for; all of them are dirty rows
        if not need.any():
            continue
        for f in range(C):
            if f == t or f in ordered_cols:
                continue
            v = d[f][need]
            v = v[v >= 0]
            if len(v) == 0:
                continue
            held = np.bincount(v, minlength=len(total[f]))   # rows with this code whose target cell is NULL: not training rows
            if train_tables and t in train_tables:            # the model trains on a row sample (its target cells are all non-NULL)
                seen = train_tables[t].count_codes(f)[0]
            else:
                seen = total[f] - held
            gone = np.flatnonzero((held > 0) & (seen <= 0))
            if len(gone):
                out.append((t, f, gone))
    return out


def repair_table(engine, table, targets, base_params, constraints=(), detect_nulls=True, error_cells=None,
                 want_pmf=False, top_k=32, threshold=0.0, want_stats=False, continuous=None, train_rows=None,
                 check_unseen=False, search_opts=None, only_noisy_targets=False):
    """Detect, NULL out, split, train, repair, shape.  ``table`` is modified in place (error cells become NULL).

    continuous : {column: (ascending distinct values, is_integral)} -- CONTINUOUS target attributes (byte/short/int/long/float/
                 double in the reference, RepairBase.scala:41-44): repaired by an L2 regressor on the values behind their codes
                 (train.py:97-100), integral ones rounded (model.py:1130-1132); `repaired_value` carries the prediction.
    only_noisy_targets: train models for the target attributes that hold at least one error cell only (the reference's
                 `target_columns` are the noisy columns, errors.py:472-477) and insist that every label of such a target still has a
                 training row once the cells are NULLed -- `RepairModel.run()` with detection on the device: the cells are not known
                 when the dictionaries are built, so a value that only error cells held would stay behind as a class without rows
                 (num_class enters the softmax hessian factor); NotResidentEligible then sends the run to the value-space path.
    search_opts: reference option dict (model.hp.*, model.cv.n_splits): run the hyper-parameter search of every target on the
                 resident tables before its final fit (`search_on_table`); None = the fixed parameters.
    train_rows : {target: row positions} or a callable (target, positions of its non-NULL rows) -> positions -- train that
                 target's model on these rows only (model.max_training_row_num sampling, model.py:755-766); default: every row
                 whose target cell is not NULL.
    Returns dict(rows, cols, current, repaired, repaired_value, prob[, pmf_class, pmf_prob, current_prob], dirty_rows, models,
    times, stats): one entry per error cell, ordered by (column, row).
    """
    continuous = dict(continuous or {})
    t0 = time.perf_counter()
    targets = [int(t) for t in targets]
    n_codes = np.asarray(table.n_codes, np.int32)
    rows, cols = detect_error_cells(table, targets, constraints, detect_nulls, error_cells)
    if only_noisy_targets:
        noisy = set(int(c) for c in np.unique(cols))
        targets = [t for t in targets if t in noisy]
    t_detect = time.perf_counter() - t0
    t0 = time.perf_counter()
    current = table.read_cells(rows, cols)
    table.null_cells(rows, cols, targets)                       # convertErrorCellsToNull (RepairApi.scala:171-211)
    dirty_rows = table.rows_of_cells(rows)                      # model.py:549-553
    out = dict(rows=rows, cols=cols, current=current, dirty_rows=dirty_rows, models={}, stats=[])
    if len(rows) == 0:
        out.update(repaired=np.zeros(0, np.int32), repaired_value=np.zeros(0, np.float64), prob=np.zeros(0, np.float64),
                   times=dict(detect=t_detect, prepare=time.perf_counter() - t0))
        return out
    dirty_tab = table.gather_rows(dirty_rows)
    pmf_tab = table.gather_rows(dirty_rows) if want_pmf else None      # stays un-repaired: pmf mode does not chain
    label_counts = {}
    for t in targets:
        cnt, _ = table.count_codes(t)
        if t in continuous:
            if int((cnt > 0).sum()) < 1:
                raise NotResidentEligible("continuous target column %d has no non-NULL row to learn from" % t)
        elif only_noisy_targets and int((cnt > 0).sum()) < len(cnt):
            raise DeadClasses("target column %d: %d of its %d values are held by error cells only" % (t, int((cnt <= 0).sum()), len(cnt)), rows, cols)
        elif int((cnt > 0).sum()) < 2:
            raise NotResidentEligible("target column %d has fewer than two classes among its non-NULL rows; the reference short-cuts such "
                             "attributes with a constant model (model.py:1008-1017) -- drop it from `targets`" % t)
        label_counts[t] = cnt
    t_prep = time.perf_counter() - t0
    y_values = {t: np.asarray(continuous[t][0], np.float64) for t in targets if t in continuous}
    integral = {t for t in targets if t in continuous and continuous[t][1]}
    train_tables = {}
    if callable(train_rows):
        picked = {}
        for t in targets:
            r = train_rows(t, np.flatnonzero(table.read_column(t) >= 0))
            if r is not None:
                picked[t] = r
        train_rows = picked
    if train_rows:
        for t, r in train_rows.items():
            train_tables[t] = table.gather_rows(np.sort(np.asarray(r, np.int64)))
            label_counts[t] = train_tables[t].count_codes(t)[0]
    if check_unseen:
        gone = unseen_categories(table, dirty_tab, targets, train_tables=train_tables, ordered_cols=set(continuous) | set(check_unseen if not isinstance(check_unseen, bool) else ()))
        if gone:
            raise UnseenCategories("%d (target, feature) pairs hold categories no training row has, e.g. target column %d / feature column %d"
                                   % (len(gone), gone[0][0], gone[0][1]))
    search = None
    if search_opts is not None:
        def search(t, tab):
            return search_on_table(engine, tab, t, n_codes, base_params, search_opts, y_value=y_values.get(t))
        # fold fits a search keeps in flight per target (run_search: model.hp.batch_size evaluations x model.cv.n_splits folds): run_job
        # budgets device memory with it
        from repair.train import search_fits_in_flight
        search.fits_in_flight = search_fits_in_flight(search_opts)
    res = run_job(engine, table, dirty_tab, n_codes, targets, label_counts, base_params, want_stats=want_stats,
                  y_values=y_values, integral=integral, train_tables=train_tables, param_search=search)
    # flatten + join with the error cells (RepairMiscApi.scala:41-49, model.py:1398-1401)
    t0 = time.perf_counter()
    tpos = np.full(table.c, -1, np.int64)
    tpos[targets] = np.arange(len(targets))
    pos = np.searchsorted(dirty_rows, rows)
    repaired = res["labels"][tpos[cols], pos].astype(np.int32)
    prob = res["probs"][tpos[cols], pos] if res["probs"] is not None else None
    repaired_value = res["values"][tpos[cols], pos] if res.get("values") is not None else np.full(len(rows), np.nan)
    out.update(repaired=repaired, repaired_value=repaired_value, prob=prob, models=res["models"], stats=res["stats"])
    if want_pmf:
        pc = np.full((len(rows), top_k), -1, np.int32)
        pp = np.zeros((len(rows), top_k), np.float64)
        cp = np.zeros(len(rows), np.float64)
        for t in targets:
            sel = np.flatnonzero(cols == t)
            if len(sel) == 0 or t in continuous:       # continuous attributes have no pmf (model.py:1215-1222: the value with prob 1.0)
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


def encode_frame(df, columns):
    """Dictionary encoding of the listed columns: (indices [C][n] int32 with -1 = NULL, remaps, dictionaries).

    `remaps[c][i]` is the code of dictionary entry i: its rank among the column's distinct values in ascending order
    (strings by code point, numbers numerically) -- the contract of repair.encode, so codes mean the same on both paths.
    The per-row work (value -> dictionary index in order of first appearance) is one hash pass per column (`pandas.factorize`:
    42 ms per 10^6 string cells, against 67-105 ms for an Arrow `dictionary_encode` of the same column); the index -> code gather
    runs on the device (`Table.from_dictionaries`)."""
    import pandas as pd
    idx, remaps, dicts = [], [], []
    for c in columns:
        s = df[c]
        numeric = pd.api.types.is_numeric_dtype(s) and not pd.api.types.is_bool_dtype(s)
        if numeric:
            codes, uniq = pd.factorize(s.to_numpy(dtype="float64", na_value=np.nan), use_na_sentinel=True)
        else:
            # one hash pass per column and run: inside RepairModel.run() the NULL detector / domain statistics have factorised the column
            # already (repair.utils.column_code_cache); categorical / Arrow-backed columns factorise through their own dictionary or buffer
            from repair.utils import column_factorize
            codes, uniq = column_factorize(df, c)
        vals = np.asarray(uniq, dtype=np.float64 if numeric else object)
        order = np.argsort(vals, kind="stable")
        remap = np.empty(len(vals), np.int32)
        remap[order] = np.arange(len(vals), dtype=np.int32)
        idx.append(codes.astype(np.int32, copy=False))
        remaps.append(remap)
        dicts.append(vals[order])
    return (np.stack(idx) if idx else np.zeros((0, len(df)), np.int32)), remaps, dicts


def repair_frame(engine, df, row_id, targets=None, constraints=(), base_params=None, want_pmf=False, top_k=32, threshold=0.0,
                 error_cells=None, detect_nulls=True, continuous_columns=(), train_rows=None, want_details=False,
                 check_unseen=False, search_opts=None, only_noisy_targets=False):
    """DataFrame in, the reference's result frame out: (row_id, attribute, current_value, repaired, prob[, pmf]) -- the
    shape of `RepairModel.run()` / `run(compute_repair_candidate_prob=True)` (python/repair/model.py:1398-1419).

    Columns are discrete (one class per distinct value) unless named in `continuous_columns` (numeric columns; those targets
    get regressors and `repaired` is the predicted number, rounded for integer columns); `constraints` are `X1,..,Xm -> Y`
    dependencies given as ([x names], y name); `error_cells` is a frame with `row_id` and `attribute` columns
    (RepairModel.setErrorCells).  Regex / outlier detectors, rule-based repairs and cost functions stay with
    `repair.model.RepairModel` (the value-space API)."""
    import pandas as pd
    cols = [c for c in df.columns if c != row_id]
    targets = list(targets) if targets is not None else list(cols)
    unknown = [t for t in targets if t not in cols]
    if unknown:
        raise ValueError("Target attributes not found in the input: %s" % ",".join(unknown))
    indices, remaps, dicts = encode_frame(df, cols)
    pos = {c: i for i, c in enumerate(cols)}
    cells, given_current = None, None

    def null_known_cells(kr, kc):
        """The cells (row positions kr, column indices kc) are known before the table is (re)built: NULL them in the dictionary
        indices and drop the values that ONLY they held from the dictionaries of the target attributes -- the reference counts a
        target's classes over the frame with the error cells removed (model.py:1005, count(distinct y)), and num_class enters the
        softmax hessian factor K / (K - 1): a dead class would change every tree.  Returns (keys, values): what the cells held."""
        tset = {pos[t] for t in targets}
        sel = np.isin(kc, list(tset))
        gr, gc = kr[sel], kc[sel]
        gv = np.empty(len(gr), object)                          # the values the cells hold now: `current_value` of the result
        for j in np.unique(gc):
            m = gc == j
            idx = indices[j, gr[m]]
            v = np.empty(int(m.sum()), object)
            v[idx >= 0] = dicts[j][remaps[j][idx[idx >= 0]]]
            v[idx < 0] = None
            gv[m] = v
        indices[gc, gr] = -1
        for j in sorted(tset):
            live = np.bincount(indices[j][indices[j] >= 0], minlength=len(remaps[j])) > 0
            if not live.all():
                live_codes = np.sort(remaps[j][live])                     # surviving old codes, ascending
                new = np.zeros(len(remaps[j]), np.int32)
                new[live] = np.searchsorted(live_codes, remaps[j][live]).astype(np.int32)
                dicts[j] = dicts[j][live_codes]
                remaps[j] = new
        return gc.astype(np.int64) * len(df) + gr, gv

    if error_cells is not None:
        if row_id not in error_cells.columns or "attribute" not in error_cells.columns:
            raise ValueError("Error cells should have `%s` and `attribute` in columns" % row_id)
        rpos = pd.Series(np.arange(len(df)), index=df[row_id].to_numpy()).reindex(error_cells[row_id].to_numpy()).to_numpy(np.float64)
        cpos = np.array([pos.get(a, -1) for a in error_cells["attribute"]], np.int64)
        ok = ~np.isnan(rpos) & (cpos >= 0)                      # cells of unknown rows / attributes drop out (join semantics)
        cells = (rpos[ok].astype(np.int64), cpos[ok].astype(np.int32))
        given_current = null_known_cells(cells[0], cells[1])
    if error_cells is not None and not detect_nulls and not constraints:
        # every error cell is known and NULLed: the class counts the models will see are final.  Say so BEFORE anything is uploaded
        # (a constraint / regex / user-given cell may hold the only occurrence of a class; an all-NULL numeric column has no value)
        for t in targets:
            j = pos[t]
            live = len(np.unique(indices[j][indices[j] >= 0]))
            if live < (1 if t in continuous_columns else 2):
                raise NotResidentEligible("target `%s` is left with %d distinct value(s) once the error cells are removed" % (t, live))

    def build_and_run(cells_, detect_nulls_, constraints_):
        table = engine.upload_dictionaries(indices, remaps)
        for j, c in enumerate(cols):               # numeric columns: bin bounds at the midpoints of the values, like LightGBM on raw numbers
            if dicts[j].dtype != object and len(dicts[j]) > 0:
                table.set_column_values(j, dicts[j])
            elif dicts[j].dtype == object:
                table.set_column_kind(j, True)     # categories a model's training rows do not show are missing for that model
        cons = [([pos[x] for x in xs], pos[y]) for xs, y in constraints_]
        cont_ = {}
        for c in continuous_columns:
            if c in pos and c in targets:
                cont_[pos[c]] = (np.asarray(dicts[pos[c]], np.float64), pd.api.types.is_integer_dtype(df[c]))
        return cont_, repair_table(engine, table, [pos[t] for t in targets], dict(base_params or {}), constraints=cons, detect_nulls=detect_nulls_,
                                   error_cells=cells_, want_pmf=want_pmf, top_k=top_k, threshold=threshold, continuous=cont_, search_opts=search_opts,
                                   only_noisy_targets=only_noisy_targets,
                                   check_unseen=([pos[c] for c in cols if pd.api.types.is_numeric_dtype(df[c]) and not pd.api.types.is_bool_dtype(df[c])] or True) if check_unseen else False,
                                   train_rows=(lambda t, r: train_rows(cols[t], r)) if callable(train_rows) else train_rows)

    try:
        cont, res = build_and_run(cells, detect_nulls, constraints)
    except DeadClasses as e:
        # Detection ran on the device; some values of a target were held by error cells only (a typo occurs once).  The cells are
        # known now: NULL them on the host side of the encoding, drop the dead values, upload the (cached) encoding again and go on
        # with the cells as GIVEN -- no second detection, no pandas detectors, no value-space fallback.
        _logger.info("%s: re-encoding the target dictionaries without them" % e)
        cells = (e.rows, e.cols)
        first = given_current
        given_current = null_known_cells(cells[0], cells[1])
        if first is not None:
            # cells of a user-given error_cells frame were NULLed before the first attempt: the second pass sees None there, the first
            # one knows what they held
            fresh = ~np.isin(given_current[0], first[0])
            given_current = (np.concatenate([first[0], given_current[0][fresh]]), np.concatenate([first[1], given_current[1][fresh]]))
        cont, res = build_and_run(cells, False, ())
    rows, ccols = res["rows"], res["cols"]

    def decode(codes, col_idx):
        out = np.empty(len(codes), object)
        for j in np.unique(col_idx):
            sel = col_idx == j
            d = dicts[j]
            c = codes[sel]
            v = np.empty(len(c), object)
            ok = c >= 0
            v[ok] = d[c[ok]]
            v[~ok] = None
            out[sel] = v
        return out

    repaired = decode(res["repaired"], ccols)
    for j, (_, is_int) in cont.items():                       # continuous attributes: the regressor's value, not a dictionary entry
        sel = ccols == j
        v = res["repaired_value"][sel]
        repaired[sel] = [int(x) for x in v] if is_int else [float(x) for x in v]
    current = decode(res["current"], ccols)
    if given_current is not None and len(rows):
        key = ccols.astype(np.int64) * len(df) + rows
        order = np.argsort(given_current[0], kind="stable")
        at = np.searchsorted(given_current[0][order], key)
        at = np.clip(at, 0, max(len(order) - 1, 0))
        hit = given_current[0][order][at] == key if len(order) else np.zeros(len(key), bool)
        current[hit] = given_current[1][order][at[hit]]
    frame = pd.DataFrame({row_id: df[row_id].to_numpy()[rows], "attribute": np.asarray(cols, object)[ccols],
                          "current_value": current, "repaired": repaired})
    if res.get("prob") is not None:
        frame["prob"] = res["prob"]
    if want_pmf and len(rows):
        pc, pp = res["pmf_class"], res["pmf_prob"]
        pmf = []
        for i in range(len(rows)):
            d = dicts[ccols[i]]
            pmf.append([{"class": d[k], "prob": float(p)} for k, p in zip(pc[i], pp[i]) if k >= 0])
        frame["pmf"] = pmf
        frame["current_prob"] = res["current_prob"]
    if want_details:
        return frame, dict(times=res.get("times", {}), stats=res.get("stats", []), models=res.get("models", {}), columns=cols)
    return frame


def constraint_to_columns(preds, columns):
    """Parsed denial constraint (repair.errors.parse_constraint) -> (eq column indices, iq column index) when it has the form
    the device detector handles -- two tuples, same attribute on both sides, EQ predicates plus exactly one IQ -- else None
    (single-tuple constant predicates, LT/GT, several IQs stay with the pandas detector)."""
    pos = {c: i for i, c in enumerate(columns)}
    eq, iq = [], []
    for p in preds:
        if p.constant is not None or p.right is None or p.left != p.right or p.left not in pos:
            return None
        if p.op == "EQ":
            eq.append(pos[p.left])
        elif p.op == "IQ":
            iq.append(pos[p.left])
        else:
            return None
    if len(iq) != 1 or len(eq) > 12:
        return None
    return eq, iq[0]
