"""The repair job on label-encoded tables: train one model per target attribute, then run the
chained repair over the dirty rows -- `_build_repair_stat_models_in_series/_in_parallel` +
`_repair` of the reference (python/repair/model.py:768-926, 1062-1143) without Spark.

``HipEngine`` is the only engine the product ships: it keeps the int32 table resident in HBM and
drives librepairgbm.so.  The job logic takes the engine as an argument so that the multi-rank tests
can run the same sharding code on CPU ranks (gloo) with the test-only oracle engine.
"""
import time

import numpy as np

from repair import _native, dist


def balanced_class_weight(counts):
    """sklearn ``class_weight='balanced'`` (reference: train.py:39-40,105): n / (n_present * count)."""
    counts = np.asarray(counts, np.float64)
    n, present = counts.sum(), int((counts > 0).sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(counts > 0, n / (present * counts), 0.0)


class HipEngine:
    """HBM-resident tables + the HIP trainer/predictor (fails loudly without a GPU)."""

    name = "hip"

    def __init__(self, device_id=0):
        if _native.device_count() <= device_id:
            raise _native.RepairGbmError("HIP device %d not available (found %d); there is no CPU fallback" %
                                         (device_id, _native.device_count()))
        self.device_id = device_id

    def upload(self, codes, n_codes):
        return _native.Table(codes, n_codes, device_id=self.device_id)

    def upload_dictionaries(self, indices, remaps):
        """Encode on the device: Arrow dictionary indices + sorted-rank remap tables -> resident code table."""
        return _native.Table.from_dictionaries(indices, remaps, device_id=self.device_id)

    def train(self, table, target, feats, class_weight, params, y_value=None, want_stats=False):
        return table.train(target, feats, y_value=y_value, class_weight=class_weight, want_stats=want_stats, **params)

    @staticmethod
    def small_rows():
        """Tables up to this many rows train through the batched small-table trainer when several fits are at hand."""
        import os
        return int(os.environ.get("RGBM_SMALL_ROWS", "65536"))

    def train_many(self, fits):
        """Many fits in one batched call (rgbm_table_train_batch; include/rgbm.h): fits = [(table, target, feats, class_weight,
        params, y_value[, valid_table]), ...] -> [model or the exception of that fit].  Every model equals `train` on the same
        arguments; a fit with a valid_table comes back as (model, labels, values) of that table's rows, scored while it trained."""
        return _native.train_batch([dict(table=f[0], target_col=f[1], feat_cols=f[2], y_value=f[5], class_weight=f[3],
                                         **(dict(valid_table=f[6]) if len(f) > 6 and f[6] is not None else {}), **f[4]) for f in fits])

    def train_row_sharded(self, shard_table, target, feats, class_weight, params, y_value=None, want_stats=False):
        """Collective: every rank calls this for the same target with its own row shard (needs dist.init_row_comm)."""
        return shard_table.train(target, feats, y_value=y_value, class_weight=class_weight, want_stats=want_stats, row_sharded=True, **params)

    def fusion_group(self, n_members):
        """Several row-sharded targets of this rank in flight at once, ONE collective per step for all of them (_native.FusionGroup).  Needs the
        rank's communicator on the calling thread; None when there is none (then the targets train one after another, as before)."""
        if _native.comm_info()["kind"] not in (1, 2) or n_members < 2:
            return None
        return _native.FusionGroup(n_members)

    def load_model(self, blob):
        return _native.Model.load(blob)

    def device_memory_bytes(self):
        try:
            import torch
            return float(torch.cuda.get_device_properties(self.device_id).total_memory)
        except Exception:  # noqa: BLE001
            return 256e9

    def repair_chain(self, table, models, targets, feats, row_begin, n_rows):
        return table.repair_chain(models, targets, feats, row_begin=row_begin, n_rows=n_rows)

    def repair_chain_gather(self, table, models, targets, feats, row_begin, n_rows):
        """C2 on device buffers: the chain over this rank's rows, labels / probabilities all-gathered over the rank's communicator before
        they leave the device (include/rgbm.h rgbm_table_repair_chain_gather) -> (labels, probs of ALL ranks' rows in rank order, first row of this rank)."""
        lab, prob, row0, _ = table.repair_chain_gather(models, targets, feats, row_begin=row_begin, n_rows=n_rows)
        return lab, prob, row0


def model_params(n_classes, base, continuous=False):
    """objective pick of the reference (train.py:97-100): regression for continuous targets, else binary / multiclass."""
    p = dict(base)
    p["objective"] = 2 if continuous else (0 if n_classes <= 2 else 1)
    p["num_class"] = max(int(n_classes), 2)
    return p


def nearest_code(values, v):
    """Code of the dictionary value nearest to v (ties to the lower one): what repair.encode gives a number it has not seen."""
    values = np.asarray(values, np.float64)
    v = np.asarray(v, np.float64)
    pos = np.clip(np.searchsorted(values, v, side="left"), 0, len(values) - 1)
    lo = np.clip(pos - 1, 0, len(values) - 1)
    pick_lo = np.abs(v - values[lo]) <= np.abs(values[pos] - v)
    return np.where(pick_lo, lo, pos).astype(np.int32)


def chained_repair(engine, table, models, targets, feats_l, row_begin, n_rows, y_values, integral):
    """`_repair` of the reference (model.py:1107-1133) on rows [row_begin, row_begin + n_rows) of a resident table.

    Discrete targets are scored and filled by the device chain (`repair_chain`), run by run.  A CONTINUOUS target is one more
    model of the same chain: its regressor scores every row, integral attributes are rounded (np.round, half to even,
    model.py:1131-1132), and the NULL cells of the column receive the code of the nearest dictionary value so that the
    models behind it see the repaired cell -- as the reference's later models see the predicted number.
    Returns (labels [T][n] int32 (-1 for continuous targets), probs [T][n] float64, values [T][n] float64 (NaN for discrete))."""
    T = len(targets)
    labels = np.full((T, n_rows), -1, np.int32)
    probs = np.zeros((T, n_rows), np.float64)
    values = np.full((T, n_rows), np.nan, np.float64)
    i = 0
    while i < T:
        j = i
        if targets[i] not in y_values:
            while j < T and targets[j] not in y_values:
                j += 1
            lab, prob = engine.repair_chain(table, models[i:j], targets[i:j], feats_l[i:j], row_begin, n_rows)
            labels[i:j] = lab
            if prob is not None:
                probs[i:j] = prob
        else:
            j = i + 1
            t = targets[i]
            _, pred = engine.repair_chain(table, models[i:j], [t], feats_l[i:j], row_begin, n_rows)
            v = np.asarray(pred[0], np.float64)
            if t in integral:
                v = np.round(v)
            values[i] = v
            probs[i] = 1.0
            col = table.read_column(t)[row_begin:row_begin + n_rows]
            nul = np.flatnonzero(col < 0)
            if len(nul):
                table.write_cells(nul + row_begin, np.full(len(nul), t, np.int32), nearest_code(y_values[t], v[nul]))
        i = j
    return labels, probs, values


def _train_concurrency(engine, table, costs, requested, search_fits=0):
    """How many target models train at once: `requested` (or RGBM_TARGET_CONCURRENCY, default 6: measured on the 10M x 16 job,
    60 iterations: 79.7 ms per step with 4 in flight, 76.2-76.6 with 6, 76.3-76.7 with 8), capped so that the largest `n` models
    together stay within half of the device memory.  A target with a hyper-parameter search in front of its final fit keeps
    `search_fits` fold fits in flight of its own (run_search), each on two thirds of the rows next to its two gathered fold tables
    (4 B per cell): they are charged to the target."""
    import os
    n = int(requested if requested is not None else os.environ.get("RGBM_TARGET_CONCURRENCY", "6"))
    if n <= 1 or getattr(engine, "name", "") != "hip":
        return 1
    budget = 0.5 * getattr(engine, "device_memory_bytes", lambda: 256e9)()
    per_unit = 26.0 * (1.0 + 0.67 * max(0, int(search_fits)))
    fold_tables = 4.0 * float(getattr(table, "n", 0)) * float(getattr(table, "c", 0)) * max(0, int(search_fits))
    need = sorted((per_unit * c + fold_tables for _, c in costs), reverse=True)       # cost = class trees x rows
    while n > 1 and sum(need[:n]) > budget:
        n -= 1
    return n


def run_job(engine, train_table, dirty_table, n_codes, targets, label_counts, base_params, want_stats=False, row_table=None,
            force_row_sharding=False, train_concurrency=None, y_values=None, integral=(), train_tables=None, param_search=None,
            dirty_is_shard=False, row_shard_all=False):
    """Train + repair, sharded over the ranks of the current torch.distributed group (if any).

    train_table / dirty_table : engine tables (all rows with error cells NULLed / the dirty rows)
    label_counts[t]           : per-code row counts of target t over its non-NULL rows (GLOBAL counts)
    row_table                 : this rank's row shard of the training table; when given (and more than one rank),
                                the expensive targets are trained row-sharded over ALL ranks (dist.split_targets)
    row_shard_all             : every target is row-sharded (with `dirty_is_shard`: a job whose ranks hold nothing but their shard)
    dirty_is_shard            : the dirty table holds the dirty rows of THIS rank's row shard only (a rank that never sees the whole
                                table: every target is then row-sharded, train_table may be None); the rank repairs all of them and the
                                all-gather concatenates the ranks' rows in rank order (`dirty_row0` = position of this rank's first row)
    y_values[t]               : CONTINUOUS targets only -- the ascending distinct values behind the codes of column t: the target gets
                                an L2 regressor (train.py:97-100) on those values; `integral` names the ones rounded after prediction
    Returns dict(labels [T][D], probs [T][D], values [T][D] or None, models {target: bytes}, times, stats).
    """
    y_values = dict(y_values or {})
    integral = set(integral)
    train_tables = dict(train_tables or {})     # {target: table}: that target trains on its own (sampled) table, model.py:755-766
    # param_search(target, table) -> {LightGBM core parameter: value} found by the hyper-parameter search for that target (train.py:133-209),
    # run by the rank that owns the target right before its final fit; None = the fixed parameters
    rank, ws = dist.world()
    n_cols = len(n_codes)
    costs = []
    for t in targets:
        k = int(n_codes[t])
        costs.append((t, (1 if (k <= 2 or t in y_values) else k) * float(np.sum(label_counts[t]))))
    big, small = dist.split_targets(costs, ws, row_table is not None, force=force_row_sharding, all_targets=row_shard_all)
    mine = dist.assign_targets(small, ws)[rank]
    t0 = time.perf_counter()
    blobs, stats, shared, fusion_stats = {}, [], {}, {}
    trained = {}     # target -> the model object this rank trained itself (load(save(m)) is m: no need to parse its own blob again)

    def one(t, table, fn):
        feats = [c for c in range(n_cols) if c != t]
        cw = balanced_class_weight(label_counts[t])
        base = dict(base_params)
        if param_search is not None and fn == engine.train:
            base.update(param_search(t, table) or {})
        res = fn(table, t, feats, cw, model_params(int(n_codes[t]), base, continuous=t in y_values), y_value=y_values.get(t),
                 want_stats=want_stats)
        if want_stats:
            res, st = res
            st["target"] = t
            stats.append(st)
        trained[t] = res
        return res.save()

    # This rank's own (target-sharded) models train CONCURRENTLY, with each other and with the row-sharded ones: every training call
    # owns a HIP stream and ctypes releases the GIL.  A few-class target is a chain of small dependent kernels (plan / split-find /
    # replay at 10-40 us around 0.1 ms passes) that leaves the GPU idle two thirds of the time; next to a many-class target those
    # kernels fill the gaps between its passes.  Largest first (LPT), a bounded number in flight (device memory: ~26 B per row and
    # class tree each).  The row-sharded targets run next to them: in a fusion group (below: member j of the group trains big[j],
    # big[j + M], ... on a thread and a stream of its own, the same assignment on every rank, ONE collective per step for all of them) or,
    # without one, one after another on THIS thread in the same order on every rank.  The models do not depend on any of this.
    conc = _train_concurrency(engine, train_table, [c for c in costs if c[0] in set(mine)] + list(big), train_concurrency,
                              search_fits=getattr(param_search, "fits_in_flight", 0) if param_search is not None else 0)
    # The reference's default job trains every model on a <= 10 000-row sample (model.py:755-766): such fits are chains of tiny
    # dependent kernels, so all of this rank's small fits go through ONE batched training call (rgbm_table_train_batch: one
    # workgroup per (fit, class tree), three launches per boosting iteration for the whole batch); same models, bit for bit.
    batched = []
    if param_search is None and not want_stats and hasattr(engine, "train_many"):
        batched = [t for t in mine if getattr(train_tables.get(t, train_table), "n", 1 << 62) <= engine.small_rows()]
        if len(batched) < 2:
            batched = []
    if batched:
        fits = []
        for t in batched:
            feats = [c for c in range(n_cols) if c != t]
            fits.append((train_tables.get(t, train_table), t, feats, balanced_class_weight(label_counts[t]),
                         model_params(int(n_codes[t]), dict(base_params), continuous=t in y_values), y_values.get(t)))
        ms = engine.train_many(fits)
        for m in ms:
            if isinstance(m, Exception):
                raise m
        # the blobs of a batch are tens of MB (300 x K trees a model): serialised side by side (ctypes releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(ms))) as ex:
            for t, m, b in zip(batched, ms, ex.map(lambda m: m.save(), ms)):
                blobs[t] = b
                trained[t] = m
    mine_single = [t for t in mine if t not in set(batched)]
    pool, futs = None, {}
    if conc > 1 and len(mine_single) + len(big) > 1 and mine_single:
        from concurrent.futures import ThreadPoolExecutor
        cost_of = dict(costs)
        order = sorted(mine_single, key=lambda t: -cost_of[t])
        pool = ThreadPoolExecutor(max_workers=max(1, conc - (1 if big else 0)))
        futs = {t: pool.submit(one, t, train_tables.get(t, train_table), engine.train) for t in order}
    try:
        # Row-sharded targets (collective: every rank trains the same targets, identical model everywhere).  With a fusion group the rank
        # keeps several of them in flight -- member j of the group trains big[j], big[j + M], ... on a thread and a stream of its own, the
        # same assignment on every rank -- and the i-th collective of all members is ONE all-reduce (include/rgbm.h "Fusion group").
        # Without one (oracle engine of the CPU tests, a single target, RGBM_FUSION=0) they train one after another on this thread.
        import os
        # (members in flight: measured on one rank's share of the 100M x 32 job -- a 12.5M-row shard, eight targets -- 91.8 ms per step one
        # after another, 79.9 with six in flight, 75.6 with all eight, 72.5 without any collective: profiles/r5g_*.)  Up to eight, CAPPED by the
        # same memory budget that caps `conc`: a member's fit holds ~26 B per (row of THIS rank's shard, class tree) = cost / ws, and
        # the pool beside it keeps up to conc - 1 whole-table fits in flight.  The number fixes the collective schedule (big[j::n_members]), so
        # the ranks AGREE on it -- it derives from rank-local inputs (this rank's share of the small targets, its free memory, its environment):
        # the minimum over the ranks, one tiny all-reduce that every rank reaches because `big` is the same list everywhere.
        n_members = 1
        if big:
            want = min(len(big), 8, max(1, int(os.environ.get("RGBM_FUSION_MEMBERS", "8")))) if (os.environ.get("RGBM_FUSION", "1") != "0" and conc > 1) else 1
            budget = 0.5 * getattr(engine, "device_memory_bytes", lambda: 256e9)()
            cost_of_ = dict(costs)
            pool_need = sum(sorted((26.0 * cost_of_[t] for t in mine_single), reverse=True)[:max(0, conc - 1)])
            shard_need = sorted((26.0 * c / max(ws, 1) for _, c in big), reverse=True)
            while want > 1 and pool_need + sum(shard_need[:want]) > budget:
                want -= 1
            n_members = max(1, int(dist.min_over_ranks(want)))
        group = engine.fusion_group(n_members) if (n_members > 1 and hasattr(engine, "fusion_group")) else None
        if group is not None:
            from concurrent.futures import ThreadPoolExecutor

            def member(j):
                with group.member(j):
                    return [(t, one(t, row_table, engine.train_row_sharded)) for t, _ in big[j::n_members]]
            try:
                with ThreadPoolExecutor(max_workers=n_members) as ex:
                    for part in ex.map(member, range(n_members)):
                        shared.update(part)
                fusion_stats.update(group.info(), members=n_members)
            finally:
                group.close()
        else:
            for t, _ in big:
                shared[t] = one(t, row_table, engine.train_row_sharded)
        for t in mine_single:
            blobs[t] = futs[t].result() if pool is not None else one(t, train_tables.get(t, train_table), engine.train)
    except BaseException:
        # a failing fit must surface now: do not sit out the queued fits first (the peers of a collective job are waiting)
        if pool is not None:
            pool.shutdown(wait=False, cancel_futures=True)
            pool = None
        raise
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    t_train = time.perf_counter() - t0
    # C1: all-gather of the serialised models (Spark broadcast, model.py:1069)
    t0 = time.perf_counter()
    all_blobs = dist.exchange_blobs(blobs)
    all_blobs.update(shared)
    todo = [t for t in targets if t not in trained]
    if len(todo) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            trained.update(zip(todo, ex.map(lambda t: engine.load_model(all_blobs[t]), todo)))
    else:
        trained.update((t, engine.load_model(all_blobs[t])) for t in todo)
    models = [trained[t] for t in targets]
    t_xchg = time.perf_counter() - t0
    # data-parallel chained inference on this rank's row shard
    t0 = time.perf_counter()
    # (a rank of a shard-only job whose shard holds no dirty row passes dirty_table = None: it repairs nothing and still takes part in the gathers)
    D = dirty_table.n if dirty_table is not None else 0
    b, c = (0, D) if dirty_is_shard else dist.shard_rows(D, ws, rank)
    feats_l = [[cc for cc in range(n_cols) if cc != t] for t in targets]
    row0 = 0
    # C2 = the union of the ranks' repaired cells (the UDF outputs, model.py:1142).  When the rank's librepairgbm communicator spans the
    # job (RCCL; the thread group of the tests) the chain's outputs stay on the device, are all-gathered there (ncclAllGather over xGMI on
    # the ONE communicator the rank holds) and leave it once; otherwise they are gathered from the host through the process group.
    if not y_values and ws > 1 and hasattr(engine, "repair_chain_gather") and dist._lib_comm_spans_world():
        labels, probs, row0 = engine.repair_chain_gather(dirty_table if dirty_table is not None else row_table, models, targets, feats_l, b, c)
        values = None
        dist._note_gather("librepairgbm communicator (device buffers, all-gather)")
        t_infer = time.perf_counter() - t0
        t_gather = 0.0           # (inside the call: config.rccl.gather has its seconds)
        if not dirty_is_shard:
            row0 = 0
    else:
        if D == 0:
            lab, prob = np.zeros((len(targets), 0), np.int32), np.zeros((len(targets), 0), np.float64)
            val = np.zeros((len(targets), 0), np.float64) if y_values else None
        elif y_values:
            lab, prob, val = chained_repair(engine, dirty_table, models, targets, feats_l, b, c, y_values, integral)
        else:
            lab, prob = engine.repair_chain(dirty_table, models, targets, feats_l, b, c)
            val = None
        t_infer = time.perf_counter() - t0
        # C2: all-gather of the repaired cells
        t0 = time.perf_counter()
        if dirty_is_shard:
            labels, row0 = dist.gather_rows_var(lab)
            probs = dist.gather_rows_var(prob)[0] if prob is not None else None
            values = dist.gather_rows_var(val)[0] if val is not None else None
        else:
            labels = dist.gather_rows(lab, D)
            probs = dist.gather_rows(prob, D) if prob is not None else None
            values = dist.gather_rows(val, D) if val is not None else None
        t_gather = time.perf_counter() - t0
    return dict(labels=labels, probs=probs, values=values, models=all_blobs, stats=stats, my_targets=mine, row_sharded_targets=[t for t, _ in big], fusion=fusion_stats,
                dirty_row0=row0, times=dict(train=t_train, exchange=t_xchg, infer=t_infer, gather=t_gather))
