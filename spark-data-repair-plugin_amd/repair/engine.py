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

    def train_row_sharded(self, shard_table, target, feats, class_weight, params, y_value=None, want_stats=False):
        """Collective: every rank calls this for the same target with its own row shard (needs dist.init_row_comm)."""
        return shard_table.train(target, feats, y_value=y_value, class_weight=class_weight, want_stats=want_stats, row_sharded=True, **params)

    def load_model(self, blob):
        return _native.Model.load(blob)

    def repair_chain(self, table, models, targets, feats, row_begin, n_rows):
        return table.repair_chain(models, targets, feats, row_begin=row_begin, n_rows=n_rows)


def model_params(n_classes, base):
    p = dict(base)
    p["objective"] = 0 if n_classes <= 2 else 1
    p["num_class"] = max(int(n_classes), 2)
    return p


def run_job(engine, train_table, dirty_table, n_codes, targets, label_counts, base_params, want_stats=False, row_table=None,
            force_row_sharding=False):
    """Train + repair, sharded over the ranks of the current torch.distributed group (if any).

    train_table / dirty_table : engine tables (all rows with error cells NULLed / the dirty rows)
    label_counts[t]           : per-code row counts of target t over its non-NULL rows (GLOBAL counts)
    row_table                 : this rank's row shard of the training table; when given (and more than one rank),
                                the expensive targets are trained row-sharded over ALL ranks (dist.split_targets)
    Returns dict(labels [T][D], probs [T][D], models {target: bytes}, times, stats).
    """
    rank, ws = dist.world()
    n_cols = len(n_codes)
    costs = []
    for t in targets:
        k = int(n_codes[t])
        costs.append((t, (1 if k <= 2 else k) * float(np.sum(label_counts[t]))))
    big, small = dist.split_targets(costs, ws, row_table is not None, force=force_row_sharding)
    mine = dist.assign_targets(small, ws)[rank]
    t0 = time.perf_counter()
    blobs, stats, shared = {}, [], {}

    def one(t, table, fn):
        feats = [c for c in range(n_cols) if c != t]
        cw = balanced_class_weight(label_counts[t])
        res = fn(table, t, feats, cw, model_params(int(n_codes[t]), base_params), want_stats=want_stats)
        if want_stats:
            res, st = res
            st["target"] = t
            stats.append(st)
        return res.save()

    for t, _ in big:                      # collective: same order on every rank, identical model everywhere
        shared[t] = one(t, row_table, engine.train_row_sharded)
    for t in mine:
        blobs[t] = one(t, train_table, engine.train)
    t_train = time.perf_counter() - t0
    # C1: all-gather of the serialised models (Spark broadcast, model.py:1069)
    t0 = time.perf_counter()
    all_blobs = dist.exchange_blobs(blobs)
    all_blobs.update(shared)
    models = [engine.load_model(all_blobs[t]) for t in targets]
    t_xchg = time.perf_counter() - t0
    # data-parallel chained inference on this rank's row shard
    t0 = time.perf_counter()
    D = dirty_table.n
    b, c = dist.shard_rows(D, ws, rank)
    feats_l = [[cc for cc in range(n_cols) if cc != t] for t in targets]
    lab, prob = engine.repair_chain(dirty_table, models, targets, feats_l, b, c)
    t_infer = time.perf_counter() - t0
    # C2: all-gather of the repaired cells
    t0 = time.perf_counter()
    labels = dist.gather_rows(lab, D)
    probs = dist.gather_rows(prob, D) if prob is not None else None
    t_gather = time.perf_counter() - t0
    return dict(labels=labels, probs=probs, models=all_blobs, stats=stats, my_targets=mine, row_sharded_targets=[t for t, _ in big],
                times=dict(train=t_train, exchange=t_xchg, infer=t_infer, gather=t_gather))
