"""Shared test helpers: golden fixture loading and the oracle-backed compute backend.

The oracle backend exposes the same train()/Model API as repair._native so the product's host code
(repair.gbm / repair.model) can be exercised on CPU-only machines.  It lives under tests/ because
only tests may touch oracle/.
"""
import gzip
import json
import os

import numpy as np
import pandas as pd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name + ".json.gz"), "rt", encoding="utf-8") as f:
        return json.load(f)


def frame(t, dtypes=True):
    df = pd.DataFrame(t["rows"], columns=t["columns"])
    if dtypes and "dtypes" in t:
        for c, d in t["dtypes"].items():
            if d.startswith("int") and df[c].isna().any():
                df[c] = df[c].astype("Int64")
            elif d.startswith(("int", "float")):
                df[c] = df[c].astype(d)
    return df


class OracleBackend:
    """Drop-in for the `repair._native` module surface used by repair.gbm (test infrastructure)."""
    from oracle import oracle as _O

    class Model(_O.OracleModel):
        def importance(self, kind="gain"):
            # feature importances are not part of the oracle; split counts from the serialised trees
            return np.zeros(self.info()["F"], np.float64)

        @staticmethod
        def load(b):
            m = OracleBackend._O.OracleModel.load(b)
            m.__class__ = OracleBackend.Model
            return m

        def predict(self, X, device_id=None):
            return OracleBackend._O.OracleModel.predict(self, X)

    @staticmethod
    def train(X, n_codes, y_code, n_y_codes, y_value=None, class_weight=None, sample_weight=None, **params):
        params.pop("device_id", None)
        m = OracleBackend._O.train(X, n_codes, y_code, n_y_codes, y_value=y_value, class_weight=class_weight,
                                   sample_weight=sample_weight, **params)
        m.__class__ = OracleBackend.Model
        return m


class OracleEngine:
    """repair.engine-compatible engine on the CPU oracle (for the gloo multi-rank tests)."""
    name = "oracle"

    class _Table:
        def __init__(self, codes, n_codes, values=None, kinds=None):
            self.codes = np.ascontiguousarray(codes, np.int32).copy()
            self.n_codes = np.asarray(n_codes, np.int32)
            self.c, self.n = self.codes.shape
            self.values = dict(values or {})        # numeric columns: value dictionaries (set_column_values)
            self.kinds = set(kinds or ())           # categorical columns (set_column_kind)

        def set_column_kind(self, col, categorical=True):
            (self.kinds.add if categorical else self.kinds.discard)(int(col))

        def set_column_values(self, col, values):
            self.values[int(col)] = np.asarray(values, np.float64)

        # the relational steps of repair.pipeline, restated by oracle/prep.py (same method names as repair._native.Table)
        def detect_nulls(self, cols):
            from oracle import prep as P
            return P.detect_nulls(self.codes, list(cols))

        def detect_constraint(self, eq_cols, iq_col, cell_cols=()):
            from oracle import prep as P
            rows, cols = P.constraint_cells(self.codes, list(eq_cols), iq_col, list(cell_cols))
            return rows if cols is None else (rows, cols)

        def read_cells(self, rows, cols):
            return self.codes[np.asarray(cols, np.int64), np.asarray(rows, np.int64)].astype(np.int32)

        def write_cells(self, rows, cols, codes):
            self.codes[np.asarray(cols, np.int64), np.asarray(rows, np.int64)] = np.asarray(codes, np.int32)

        def read_column(self, col):
            return self.codes[col].copy()

        def null_cells(self, rows, cols, target_cols):
            from oracle import prep as P
            self.codes = P.null_cells(self.codes, rows, cols, target_cols)

        def rows_of_cells(self, rows):
            from oracle import prep as P
            return P.rows_of_cells(self.n, rows)

        def gather_rows(self, rows):
            return OracleEngine._Table(self.codes[:, np.asarray(rows, np.int64)], self.n_codes, self.values, self.kinds)

        def count_codes(self, col):
            from oracle import prep as P
            return P.count_codes(self.codes, col, int(self.n_codes[col]))

        def repair_pmf(self, model, target_col, feat_cols, top_k=32, threshold=0.0, cur_codes=None, want_cur_prob=False):
            from oracle import prep as P
            rows = np.flatnonzero(self.codes[target_col] < 0).astype(np.int64)
            K = model.info()["num_class"]
            proba = model.predict(np.ascontiguousarray(self.codes[list(feat_cols)][:, rows])) if len(rows) else np.zeros((0, K))
            cls, pr = P.top_k_pmf(proba, top_k, threshold)
            if cur_codes is None and not want_cur_prob:
                return rows, cls, pr
            cur = np.full(len(rows), -1, np.int32) if cur_codes is None else np.asarray(cur_codes, np.int32)
            cp = np.where(cur >= 0, proba[np.arange(len(rows)), np.maximum(cur, 0)], 0.0) if len(rows) else np.zeros(0)
            return rows, cls, pr, cp

    def upload(self, codes, n_codes):
        return OracleEngine._Table(codes, n_codes)

    def upload_dictionaries(self, indices, remaps):
        from oracle import prep as P
        codes = P.encode_dictionaries(np.asarray(indices, np.int32), remaps)
        return OracleEngine._Table(codes, [max(int(np.max(m)) + 1 if len(m) else 0, 1) for m in remaps])

    def train(self, table, target, feats, class_weight, params, y_value=None, want_stats=False):
        from oracle import oracle as O
        rows = table.codes[target] >= 0
        p = {k: v for k, v in params.items() if k != "device_id"}
        fv = {i: table.values[f] for i, f in enumerate(feats) if f in table.values}
        m = O.train(np.ascontiguousarray(table.codes[feats][:, rows]), table.n_codes[feats], table.codes[target][rows],
                    int(table.n_codes[target]), y_value=y_value, class_weight=class_weight, feature_values=fv or None,
                    categorical=[i for i, f in enumerate(feats) if f in table.kinds] or None, **p)
        return (m, {"hist_ms": 0.0, "hist_bytes": 0, "hist_launches": 0, "root_ms": 0.0, "root_rows": 0}) if want_stats else m

    def train_row_sharded(self, shard_table, target, feats, class_weight, params, y_value=None, want_stats=False):
        """CPU stand-in for the collective: all-gather the row shards (so every rank must take part, in the same
        order) and train on their concatenation -- the product's row-sharded HIP training gives exactly the
        single-device model (tests/test_gpu_rowshard.py), which is what this returns."""
        import torch
        import torch.distributed as dist
        ws = dist.get_world_size()
        n = torch.tensor([shard_table.n], dtype=torch.int64)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(ws)]
        dist.all_gather(sizes, n)
        mx = int(max(int(x.item()) for x in sizes))
        pad = np.full((shard_table.c, mx), -1, np.int32)
        pad[:, :shard_table.n] = shard_table.codes
        outs = [torch.zeros((shard_table.c, mx), dtype=torch.int32) for _ in range(ws)]
        dist.all_gather(outs, torch.from_numpy(pad))
        full = np.concatenate([outs[r].numpy()[:, :int(sizes[r].item())] for r in range(ws)], axis=1)
        return self.train(OracleEngine._Table(full, shard_table.n_codes), target, feats, class_weight, params, y_value=y_value, want_stats=want_stats)

    def load_model(self, blob):
        from oracle import oracle as O
        return O.OracleModel.load(blob)

    def repair_chain(self, table, models, targets, feats, row_begin, n_rows):
        from oracle import oracle as O
        sub = np.ascontiguousarray(table.codes[:, row_begin:row_begin + n_rows])
        # an empty class list marks a regression model (its column is left untouched by the chain, like the C-ABI's)
        lab, prob = O.repair_chain(models, targets, feats, [list(range(int(table.n_codes[t]))) if m.info()["objective"] != 2 else []
                                                            for t, m in zip(targets, models)], sub)
        table.codes[:, row_begin:row_begin + n_rows] = sub
        return lab, prob
