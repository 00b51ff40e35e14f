"""ctypes binding of librepairgbm.so (include/rgbm.h) -- the only door to the HIP engine.

There is deliberately no fallback: if the shared library is missing or no HIP device is usable
the calls raise ``RepairGbmError`` (the caller in train.py turns a failed build into
``PoorModel(None)`` exactly like the reference does, python/repair/train.py:227-229).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "lib", "librepairgbm.so")


class RepairGbmError(RuntimeError):
    pass


class RgbmParams(C.Structure):
    _fields_ = [
        ("objective", C.c_int32), ("num_class", C.c_int32),
        ("n_estimators", C.c_int32), ("num_leaves", C.c_int32), ("max_depth", C.c_int32), ("max_bin", C.c_int32),
        ("min_data_in_leaf", C.c_int32), ("min_data_in_bin", C.c_int32), ("bagging_freq", C.c_int32), ("seed", C.c_int32),
        ("device_id", C.c_int32), ("reserved", C.c_int32),
        ("learning_rate", C.c_double), ("lambda_l1", C.c_double), ("lambda_l2", C.c_double), ("min_gain_to_split", C.c_double),
        ("min_sum_hessian_in_leaf", C.c_double), ("bagging_fraction", C.c_double), ("feature_fraction", C.c_double),
    ]


class RgbmTrainStats(C.Structure):
    _fields_ = [
        ("hist_ms", C.c_double), ("total_ms", C.c_double), ("hist_launches", C.c_int64), ("hist_rows", C.c_int64),
        ("hist_bytes", C.c_int64), ("root_rows", C.c_int64), ("root_ms", C.c_double), ("trees", C.c_int64),
        ("route_ms", C.c_double), ("route_launches", C.c_int64), ("root_atomics_per_row", C.c_int64), ("level_atomics_per_row", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


PARAM_DEFAULTS = dict(objective=1, num_class=2, n_estimators=300, num_leaves=31, max_depth=7, max_bin=255,
                      min_data_in_leaf=20, min_data_in_bin=3, bagging_freq=0, seed=42, device_id=0, reserved=0,
                      learning_rate=0.01, lambda_l1=0.0, lambda_l2=0.0, min_gain_to_split=0.0,
                      min_sum_hessian_in_leaf=1e-3, bagging_fraction=1.0, feature_fraction=1.0)


FLAG_ROW_SHARDED = 1
FLAG_NO_MODEL = 2


def make_params(**kw):
    d = dict(PARAM_DEFAULTS)
    kw = dict(kw)
    if kw.pop("row_sharded", False):
        d["reserved"] = d.get("reserved", 0) | FLAG_ROW_SHARDED
    if not kw.pop("want_model", True):
        d["reserved"] = d.get("reserved", 0) | FLAG_NO_MODEL
    for k, v in kw.items():
        if k not in d:
            raise TypeError("unknown parameter %r" % k)
        d[k] = v
    return RgbmParams(**d)


_lib = None


def lib():
    """Load librepairgbm.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        path = os.path.abspath(os.environ.get("RGBM_LIB_PATH") or LIB_PATH)   # RGBM_LIB_PATH: A/B runs of differently built libraries
        if not os.path.exists(path):
            raise RepairGbmError("librepairgbm.so is not built (%s); run `python __graft_entry__.py` or `make -C "
                                 "spark-data-repair-plugin_amd/csrc`" % path)
        l = C.CDLL(path)
        l.rgbm_last_error.restype = C.c_char_p
        for name in ("rgbm_device_count", "rgbm_version", "rgbm_release_cache", "rgbm_train", "rgbm_predict", "rgbm_repair_chain",
                     "rgbm_table_create", "rgbm_table_train", "rgbm_table_train_batch", "rgbm_table_repair_chain", "rgbm_table_read_column",
                     "rgbm_model_save", "rgbm_model_load", "rgbm_model_info", "rgbm_model_importance",
                     "rgbm_comm_unique_id", "rgbm_comm_init", "rgbm_comm_finalize", "rgbm_comm_info", "rgbm_comm_count",
                     "rgbm_local_group_create", "rgbm_comm_init_local",
                     "rgbm_table_detect_nulls", "rgbm_table_detect_constraint", "rgbm_table_rows_of_cells", "rgbm_table_cells_fetch",
                     "rgbm_table_null_cells", "rgbm_table_gather_rows", "rgbm_table_count_codes", "rgbm_table_create_dict",
                     "rgbm_table_shape", "rgbm_table_repair_pmf", "rgbm_table_read_cells"):
            getattr(l, name).restype = C.c_int
        l.rgbm_local_group_free.restype = None
        l.rgbm_table_free.restype = None
        l.rgbm_model_free.restype = None
        l.rgbm_host_free.restype = None
        _lib = l
    return _lib


EXPORTED_SYMBOLS = [
    "rgbm_device_count", "rgbm_last_error", "rgbm_version", "rgbm_release_cache", "rgbm_train", "rgbm_predict", "rgbm_repair_chain",
    "rgbm_table_create", "rgbm_table_free", "rgbm_table_train", "rgbm_table_train_batch", "rgbm_table_repair_chain", "rgbm_table_read_column",
    "rgbm_model_save", "rgbm_model_load", "rgbm_model_free", "rgbm_model_info", "rgbm_model_importance",
    "rgbm_comm_unique_id", "rgbm_comm_init", "rgbm_comm_finalize", "rgbm_comm_info", "rgbm_comm_count",
    "rgbm_local_group_create", "rgbm_local_group_free", "rgbm_comm_init_local",
    "rgbm_comm_all_gather_sizes", "rgbm_comm_all_gather_bytes", "rgbm_table_repair_chain_gather", "rgbm_comm_gather_stats",
    "rgbm_fusion_create", "rgbm_fusion_join", "rgbm_fusion_leave", "rgbm_fusion_info", "rgbm_fusion_free",
    "rgbm_table_detect_nulls", "rgbm_table_detect_constraint", "rgbm_table_rows_of_cells", "rgbm_table_cells_fetch",
    "rgbm_table_null_cells", "rgbm_table_gather_rows", "rgbm_table_count_codes", "rgbm_table_create_dict", "rgbm_table_shape",
    "rgbm_table_repair_pmf", "rgbm_table_read_cells", "rgbm_table_write_cells", "rgbm_host_alloc", "rgbm_host_free",
    "rgbm_table_set_column_values", "rgbm_table_set_column_kind", "rgbm_table_set_row_multiplicity",
]

COMM_ID_BYTES = 128


def comm_unique_id():
    """128 opaque bytes (an ncclUniqueId) created on one rank; every rank passes them to ``comm_init``."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib().rgbm_comm_unique_id(buf), "rgbm_comm_unique_id")
    return bytes(buf)


def comm_init(uid, rank, nranks, device_id=0):
    """RCCL communicator of the calling thread for row-sharded training (include/rgbm.h)."""
    buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
    _check(lib().rgbm_comm_init(buf, C.c_int32(rank), C.c_int32(nranks), C.c_int32(device_id)), "rgbm_comm_init")


def comm_finalize():
    _check(lib().rgbm_comm_finalize(), "rgbm_comm_finalize")


def comm_info():
    a = np.zeros(3, np.int32)
    _check(lib().rgbm_comm_info(_p(a, C.c_int32)), "rgbm_comm_info")
    return dict(kind=int(a[0]), rank=int(a[1]), nranks=int(a[2]))


def model_trees(blob):
    """The trees of a serialised model (rgbm_model_save), in (iteration, class tree) order: dicts of the arrays the blob stores -- feat, theta,
    dleft, left, right (child >= 0: internal node, < 0: ~leaf), gain, leaf_value, leaf_count.  Returns (K, n_iter, trees)."""
    import struct
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


def needed_built_rows(blob, max_depth):
    """Rows whose (g, h) the FINISHED trees of a model needed in a histogram below the root: for every split whose children can still be
    split (depth + 1 < max_depth) the rows of its smaller child (LightGBM constructs the smaller leaf and derives the sibling by
    subtraction) -- what a grower that knew the final trees in advance would accumulate.  The level grower expands a superset."""
    _, _, trees = model_trees(blob)
    total = 0
    for tr in trees:
        n = len(tr["feat"])
        if n == 0:
            continue
        left, right, lc = tr["left"], tr["right"], tr["leaf_count"]
        cnt = np.zeros(n, np.int64)
        for j in range(n - 1, -1, -1):          # children have larger indices than their parent (Tree::Split numbering)
            a, b = int(left[j]), int(right[j])
            cnt[j] = (cnt[a] if a >= 0 else int(lc[~a])) + (cnt[b] if b >= 0 else int(lc[~b]))
        depth = np.zeros(n, np.int32)
        for j in range(n):
            a, b = int(left[j]), int(right[j])
            if a >= 0:
                depth[a] = depth[j] + 1
            if b >= 0:
                depth[b] = depth[j] + 1
            if depth[j] + 1 < max_depth:
                ca = cnt[a] if a >= 0 else int(lc[~a]); cb = cnt[b] if b >= 0 else int(lc[~b])
                total += int(min(ca, cb))
    return total


def comm_count():
    """Ranks the calling thread's communicator really spans (ncclCommCount); 1 without a communicator."""
    n = C.c_int32(0)
    _check(lib().rgbm_comm_count(C.byref(n)), "rgbm_comm_count")
    return int(n.value)


def comm_all_gather_sizes(values):
    """Every rank of the calling thread's communicator contributes the int64 vector `values`; returns [nranks][len(values)]."""
    v = np.ascontiguousarray(values, np.int64).reshape(-1)
    out = np.zeros((max(1, comm_info()["nranks"]), v.size), np.int64)
    _check(lib().rgbm_comm_all_gather_sizes(_p(v, C.c_int64), C.c_int32(v.size), _p(out, C.c_int64)), "rgbm_comm_all_gather_sizes")
    return out


def comm_all_gather_bytes(payload):
    """Variable-size all-gather of byte strings over the calling thread's communicator (device buffers + ncclAllGather inside the library:
    include/rgbm.h "C1 / C2"): returns the list of every rank's payload, in rank order."""
    mine = np.frombuffer(bytes(payload), np.uint8) if not isinstance(payload, np.ndarray) else np.ascontiguousarray(payload).view(np.uint8).reshape(-1)
    sizes = comm_all_gather_sizes([mine.size])[:, 0]
    block = int(max(int(sizes.max()), 1))
    recv = np.zeros((len(sizes), block), np.uint8)
    _check(lib().rgbm_comm_all_gather_bytes(_p(mine, C.c_uint8) if mine.size else None, C.c_int64(mine.size), C.c_int64(block), _p(recv, C.c_uint8)),
           "rgbm_comm_all_gather_bytes")
    return [recv[r, :int(sizes[r])].copy() for r in range(len(sizes))]


def comm_gather_stats():
    a = np.zeros(3, np.int64)
    _check(lib().rgbm_comm_gather_stats(_p(a, C.c_int64)), "rgbm_comm_gather_stats")
    return dict(bytes=int(a[0]), seconds=float(a[1]) * 1e-9, collectives=int(a[2]))


class FusionGroup:
    """The row-sharded training calls this rank makes AT THE SAME TIME (include/rgbm.h "Fusion group"): created on the thread that holds
    the rank's communicator, which moves into the group until ``close()``.  Every member thread runs ``with group.member(j): ...`` around its
    ``Table.train(..., row_sharded=True)`` calls (the same j on every rank); the i-th collective of all members is one all-reduce."""

    def __init__(self, n_members):
        self.h = C.c_void_p()
        _check(lib().rgbm_fusion_create(C.c_int32(n_members), C.byref(self.h)), "rgbm_fusion_create")
        self.n_members = n_members

    class _Member:
        def __init__(self, group, j):
            self.group, self.j = group, j

        def __enter__(self):
            _check(lib().rgbm_fusion_join(self.group.h, C.c_int32(self.j)), "rgbm_fusion_join")
            return self

        def __exit__(self, et, ev, tb):
            lib().rgbm_fusion_leave(C.c_int32(0 if et is None else 1))
            return False

    def member(self, j):
        return FusionGroup._Member(self, j)

    def info(self):
        a = np.zeros(4, np.int64)
        _check(lib().rgbm_fusion_info(self.h, _p(a, C.c_int64)), "rgbm_fusion_info")
        return dict(collectives=int(a[0]), parts=int(a[1]), members_in=int(a[2]), broken=bool(a[3]))

    def close(self):
        """Hands the communicator back to the calling thread (call it on the thread that created the group, after every member left)."""
        if self.h:
            _check(lib().rgbm_fusion_free(self.h), "rgbm_fusion_free")
            self.h = C.c_void_p()


class LocalGroup:
    """Test transport: ``nranks`` host threads of this process act as the ranks of a row-sharded job on one device."""

    def __init__(self, nranks, device_id=0):
        self.h = C.c_void_p()
        _check(lib().rgbm_local_group_create(C.c_int32(nranks), C.c_int32(device_id), C.byref(self.h)), "rgbm_local_group_create")
        self.nranks = nranks

    def join(self, rank):
        """Call from the thread that plays ``rank``; pair with ``comm_finalize()`` in the same thread."""
        _check(lib().rgbm_comm_init_local(self.h, C.c_int32(rank)), "rgbm_comm_init_local")

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            _lib.rgbm_local_group_free(h)


def _check(rc, what):
    if rc != 0:
        raise RepairGbmError("%s failed (%d): %s" % (what, rc, lib().rgbm_last_error().decode("utf-8", "replace")))


def device_count():
    return int(lib().rgbm_device_count())


def release_cache():
    """Return the device blocks parked in the library's caching pool to the driver (include/rgbm.h)."""
    _check(lib().rgbm_release_cache(), "rgbm_release_cache")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, np.int32)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, np.float64)


class Model:
    """Owning handle of an ``rgbm_model`` (host-resident trees; device mirrors are lazy)."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            _lib.rgbm_model_free(h)

    def info(self):
        a = np.zeros(5, np.int32)
        _check(lib().rgbm_model_info(self.h, _p(a, C.c_int32)), "rgbm_model_info")
        return dict(objective=int(a[0]), num_class=int(a[1]), K=int(a[2]), n_iter=int(a[3]), F=int(a[4]))

    def save(self):
        n = C.c_size_t(0)
        _check(lib().rgbm_model_save(self.h, None, C.byref(n)), "rgbm_model_save")
        buf = bytearray(max(n.value, 1))
        _check(lib().rgbm_model_save(self.h, (C.c_char * len(buf)).from_buffer(buf), C.byref(n)), "rgbm_model_save")
        return bytes(buf[:n.value]) if n.value != len(buf) else bytes(buf)

    @staticmethod
    def load(b):
        h = C.c_void_p()
        arr = np.frombuffer(b, np.uint8)
        _check(lib().rgbm_model_load(arr.ctypes.data_as(C.c_void_p), C.c_size_t(len(b)), C.byref(h)), "rgbm_model_load")
        return Model(h)

    def importance(self, kind="gain"):
        out = np.zeros(self.info()["F"], np.float64)
        _check(lib().rgbm_model_importance(self.h, C.c_int32(0 if kind == "split" else 1), _p(out, C.c_double)), "rgbm_model_importance")
        return out

    def predict(self, X, device_id=0):
        """X: [F][n] int32 codes (column-major). Returns [n][ncol] float64 (regression: [n][1])."""
        X = _i32(X)
        F, n = X.shape
        inf = self.info()
        ncol = 1 if inf["objective"] == 2 else inf["num_class"]
        out = np.zeros((n, ncol), np.float64)
        _check(lib().rgbm_predict(self.h, _p(X, C.c_int32), C.c_int64(n), C.c_int32(F), C.c_int32(device_id), _p(out, C.c_double)), "rgbm_predict")
        return out


def train(X, n_codes, y_code, n_y_codes, y_value=None, class_weight=None, sample_weight=None, want_stats=False, **params):
    """Fit one model from host arrays. X: [F][N] int32 column-major codes."""
    X = _i32(X)
    F, N = X.shape
    n_codes, y_code = _i32(n_codes), _i32(y_code)
    yv, cw, sw = _f64(y_value), _f64(class_weight), _f64(sample_weight)
    p = make_params(**params)
    h = C.c_void_p()
    st = RgbmTrainStats()
    _check(lib().rgbm_train(_p(X, C.c_int32), C.c_int64(N), C.c_int32(F), _p(n_codes, C.c_int32), _p(y_code, C.c_int32),
                            C.c_int32(n_y_codes), _p(yv, C.c_double), _p(cw, C.c_double), _p(sw, C.c_double),
                            C.byref(p), C.byref(h), C.byref(st) if want_stats else None), "rgbm_train")
    m = Model(h)
    return (m, st.as_dict()) if want_stats else m


class RgbmFitSpec(C.Structure):
    _fields_ = [("table", C.c_void_p), ("target_col", C.c_int32), ("n_features", C.c_int32), ("feat_cols", C.POINTER(C.c_int32)),
                ("y_value", C.POINTER(C.c_double)), ("class_weight", C.POINTER(C.c_double)), ("params", C.POINTER(RgbmParams)),
                ("valid_table", C.c_void_p), ("valid_label_out", C.POINTER(C.c_int32)), ("valid_value_out", C.POINTER(C.c_double))]


def train_batch(fits):
    """Many fits in one go (include/rgbm.h rgbm_table_train_batch): ``fits`` is a list of dicts with the arguments of
    ``Table.train`` plus ``table`` -- dict(table=Table, target_col=int, feat_cols=[...], y_value=None, class_weight=None, **params).
    Returns one entry per fit: the ``Model``, or the ``RepairGbmError`` of a fit that failed (a failing fit does not fail the
    batch: the reference turns a failing build into PoorModel, python/repair/train.py:227-229).  Every model is the one
    ``Table.train`` returns for the same arguments, bit for bit.
    A fit with ``valid_table=Table`` is scored on that table WHILE it trains (cross_val_score, train.py:171-172, without a predictor):
    its entry is ``(Model, labels [rows] int32, values [rows] float64)`` -- what ``repair_chain`` of the model gives for those rows;
    with ``want_model=False`` no model is built for such a fit (``None`` in its place)."""
    n = len(fits)
    if n == 0:
        return []
    specs = (RgbmFitSpec * n)()
    keep = []
    for i, f in enumerate(fits):
        f = dict(f)
        tab = f.pop("table")
        fc = _i32(f.pop("feat_cols"))
        yv, cw = _f64(f.pop("y_value", None)), _f64(f.pop("class_weight", None))
        target = int(f.pop("target_col"))
        vtab = f.pop("valid_table", None)
        vlab = np.zeros(vtab.n, np.int32) if vtab is not None else None
        vval = np.zeros(vtab.n, np.float64) if vtab is not None else None
        f.setdefault("device_id", tab.device_id)
        p = make_params(**f)
        keep.append((tab, fc, yv, cw, p, vtab, vlab, vval))
        specs[i].valid_table = vtab.h if vtab is not None else None
        specs[i].valid_label_out = _p(vlab, C.c_int32)
        specs[i].valid_value_out = _p(vval, C.c_double)
        specs[i].table = tab.h
        specs[i].target_col = target
        specs[i].n_features = len(fc)
        specs[i].feat_cols = _p(fc, C.c_int32)
        specs[i].y_value = _p(yv, C.c_double)
        specs[i].class_weight = _p(cw, C.c_double)
        specs[i].params = C.pointer(p)
    handles = (C.c_void_p * n)()
    status = np.zeros(n, np.int32)
    _check(lib().rgbm_table_train_batch(specs, C.c_int32(n), handles, _p(status, C.c_int32)), "rgbm_table_train_batch")
    out = []
    # rgbm_last_error() is ONE thread-local string: it describes the LAST failing fit of the batch; the others report their status code only
    failing = [i for i in range(n) if status[i] != 0]
    last_msg = lib().rgbm_last_error().decode("utf-8", "replace") if failing else ""
    for i in range(n):
        if status[i] == 0 and (handles[i] or (keep[i][5] is not None and keep[i][4].reserved & FLAG_NO_MODEL)):
            m = Model(C.c_void_p(handles[i])) if handles[i] else None          # want_model=False: a CV fold, wanted for its scores only
            out.append(m if keep[i][5] is None else (m, keep[i][6], keep[i][7]))
        else:
            why = last_msg if (failing and i == failing[-1]) else ("status %d" % int(status[i]) if status[i] != 0 else "no model returned")
            out.append(RepairGbmError("fit %d of the batch failed (%d): %s" % (i, int(status[i]), why)))
    return out


def _chain_args(models, feat_cols, class_codes):
    T = len(models)
    arr = (C.c_void_p * max(T, 1))(*[m.h for m in models])
    fo = np.zeros(T + 1, np.int32)
    co = np.zeros(T + 1, np.int32)
    for t in range(T):
        fo[t + 1] = fo[t] + len(feat_cols[t])
        if class_codes is not None:
            co[t + 1] = co[t] + len(class_codes[t])
    fc = _i32(np.concatenate([np.asarray(f, np.int32) for f in feat_cols])) if T else np.zeros(1, np.int32)
    cc = None
    if class_codes is not None:
        cc = _i32(np.concatenate([np.asarray(c, np.int32) for c in class_codes] + [np.zeros(1, np.int32)]))
    return arr, fc, fo, cc, co


def repair_chain(models, target_col, feat_cols, class_codes, table, device_id=0):
    """Host-array chained repair. table: [C][n] int32, modified in place."""
    T = len(models)
    Cc, n = table.shape
    assert table.dtype == np.int32 and table.flags.c_contiguous
    arr, fc, fo, cc, co = _chain_args(models, feat_cols, class_codes)
    tc = _i32(target_col)
    lab = np.zeros((T, n), np.int32)
    prob = np.zeros((T, n), np.float64)
    _check(lib().rgbm_repair_chain(arr, C.c_int32(T), _p(tc, C.c_int32), _p(fc, C.c_int32), _p(fo, C.c_int32), _p(cc, C.c_int32),
                                   _p(co, C.c_int32), _p(table, C.c_int32), C.c_int64(n), C.c_int32(Cc), C.c_int32(device_id),
                                   _p(lab, C.c_int32), _p(prob, C.c_double)), "rgbm_repair_chain")
    return lab, prob


class _PinnedBlock:
    """Owner of a page-locked host block (rgbm_host_alloc); the numpy view keeps it alive through `.base`."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(lib().rgbm_host_alloc(C.c_size_t(nbytes), C.byref(p)), "rgbm_host_alloc")
        self.p, self.nbytes = p, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (p.value, False), "version": 3}

    def __del__(self):
        if getattr(self, "p", None):
            lib().rgbm_host_free(self.p)
            self.p = None


def pinned_empty(shape, dtype=np.int32):
    """Uninitialised numpy array in page-locked host memory: encoders that fill it hand rgbm_table_create a block the copy engine
    reads directly (no pageable staging)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.asarray(_PinnedBlock(max(n, 1)))[:n].view(dtype).reshape(shape)


class Table:
    """An int32 code table resident in HBM (``rgbm_table``)."""

    def __init__(self, codes, n_codes, device_id=0):
        codes = _i32(codes)
        self.c, self.n = codes.shape
        self.n_codes = _i32(n_codes)
        self.device_id = device_id
        h = C.c_void_p()
        _check(lib().rgbm_table_create(_p(codes, C.c_int32), C.c_int64(self.n), C.c_int32(self.c), _p(self.n_codes, C.c_int32),
                                       C.c_int32(device_id), C.byref(h)), "rgbm_table_create")
        self.h = h

    @classmethod
    def _adopt(cls, handle, device_id):
        """Wrap a table the library has just created (gather_rows / from_dictionaries)."""
        t = cls.__new__(cls)
        t.h, t.device_id = handle, device_id
        n, c = C.c_int64(0), C.c_int32(0)
        _check(lib().rgbm_table_shape(handle, C.byref(n), C.byref(c), None), "rgbm_table_shape")
        t.n, t.c = int(n.value), int(c.value)
        t.n_codes = np.zeros(t.c, np.int32)
        _check(lib().rgbm_table_shape(handle, None, None, _p(t.n_codes, C.c_int32)), "rgbm_table_shape")
        return t

    @classmethod
    def from_dictionaries(cls, indices, remaps, device_id=0):
        """Encode on the device: ``indices`` [C][n] int32 dictionary indices (< 0 = NULL), ``remaps[c][i]`` = code of
        dictionary entry i of column c (its rank among the sorted distinct values; -1 = NULL)."""
        idx = _i32(indices)
        c, n = idx.shape
        maps = [np.ascontiguousarray(m, np.int32) for m in remaps]
        if len(maps) != c:
            raise ValueError("one remap table per column expected")
        ptrs = (C.POINTER(C.c_int32) * c)(*[_p(m if len(m) else np.zeros(1, np.int32), C.c_int32) for m in maps])
        sizes = np.asarray([len(m) for m in maps], np.int32)
        h = C.c_void_p()
        _check(lib().rgbm_table_create_dict(_p(idx, C.c_int32), C.c_int64(n), C.c_int32(c), ptrs, _p(sizes, C.c_int32), C.c_int32(device_id),
                                            C.byref(h)), "rgbm_table_create_dict")
        return cls._adopt(h, device_id)

    # ---- relational steps around the models (csrc/rgbm_prep.hip)
    def _fetch_cells(self, n, want_cols):
        rows = np.zeros(n, np.int64)
        cols = np.zeros(n, np.int32) if want_cols else None
        if n:
            _check(lib().rgbm_table_cells_fetch(self.h, _p(rows, C.c_int64), _p(cols, C.c_int32)), "rgbm_table_cells_fetch")
        return (rows, cols) if want_cols else rows

    def detect_nulls(self, cols):
        """NULL cells of ``cols`` as (rows, cols), ordered by position in ``cols`` then row (ErrorDetectorApi.scala:128-157)."""
        cc = _i32(np.asarray(cols, np.int32).reshape(-1))
        n = C.c_int64(0)
        _check(lib().rgbm_table_detect_nulls(self.h, _p(cc, C.c_int32), C.c_int32(len(cc)), C.byref(n)), "rgbm_table_detect_nulls")
        return self._fetch_cells(int(n.value), True)

    def detect_constraint(self, eq_cols, iq_col, cell_cols=()):
        """Rows violating  EQ(eq_cols...) & IQ(iq_col)  (ErrorDetectorApi.scala:189-244).  Returns the ascending violating rows
        when ``cell_cols`` is empty, else the cells (rows, cols) = violating rows x cell_cols, column-major."""
        eq = _i32(np.asarray(eq_cols, np.int32).reshape(-1))
        cc = _i32(np.asarray(cell_cols, np.int32).reshape(-1))
        nr, nc = C.c_int64(0), C.c_int64(0)
        _check(lib().rgbm_table_detect_constraint(self.h, _p(eq, C.c_int32), C.c_int32(len(eq)), C.c_int32(iq_col),
                                                  _p(cc, C.c_int32), C.c_int32(len(cc)), C.byref(nr), C.byref(nc)), "rgbm_table_detect_constraint")
        return self._fetch_cells(int(nc.value), len(cc) > 0)

    def rows_of_cells(self, rows):
        """Ascending positions of the rows that hold at least one of the given cells (the dirty rows, model.py:549-553)."""
        r = np.ascontiguousarray(rows, np.int64)
        n = C.c_int64(0)
        _check(lib().rgbm_table_rows_of_cells(self.h, _p(r, C.c_int64), C.c_int64(len(r)), C.byref(n)), "rgbm_table_rows_of_cells")
        return self._fetch_cells(int(n.value), False)

    def null_cells(self, rows, cols, target_cols):
        """convertErrorCellsToNull (RepairApi.scala:171-211), in place in HBM."""
        r, c2, tc = np.ascontiguousarray(rows, np.int64), _i32(cols), _i32(np.asarray(target_cols, np.int32).reshape(-1))
        if len(r) != len(c2):
            raise ValueError("rows and cols must have the same length")
        _check(lib().rgbm_table_null_cells(self.h, _p(r, C.c_int64), _p(c2, C.c_int32), C.c_int64(len(r)), _p(tc, C.c_int32), C.c_int32(len(tc))),
               "rgbm_table_null_cells")

    def read_cells(self, rows, cols):
        r, c2 = np.ascontiguousarray(rows, np.int64), _i32(cols)
        out = np.zeros(len(r), np.int32)
        _check(lib().rgbm_table_read_cells(self.h, _p(r, C.c_int64), _p(c2, C.c_int32), C.c_int64(len(r)), _p(out, C.c_int32)), "rgbm_table_read_cells")
        return out

    def set_column_kind(self, col, categorical=True):
        """CATEGORICAL column: codes that no training row of a model holds are missing for that model (per-model dictionaries)."""
        _check(lib().rgbm_table_set_column_kind(self.h, C.c_int32(col), C.c_int32(1 if categorical else 0)), "rgbm_table_set_column_kind")

    def set_column_values(self, col, values):
        """NUMERIC column: the ascending distinct values behind its codes (bin bounds at value midpoints)."""
        v = np.ascontiguousarray(values, np.float64)
        _check(lib().rgbm_table_set_column_values(self.h, C.c_int32(col), _p(v, C.c_double), C.c_int32(len(v))), "rgbm_table_set_column_values")

    def write_cells(self, rows, cols, codes):
        r, c2, v = np.ascontiguousarray(rows, np.int64), _i32(cols), _i32(codes)
        _check(lib().rgbm_table_write_cells(self.h, _p(r, C.c_int64), _p(c2, C.c_int32), _p(v, C.c_int32), C.c_int64(len(r))), "rgbm_table_write_cells")

    def gather_rows(self, rows):
        r = np.ascontiguousarray(rows, np.int64)
        h = C.c_void_p()
        _check(lib().rgbm_table_gather_rows(self.h, _p(r, C.c_int64), C.c_int64(len(r)), C.byref(h)), "rgbm_table_gather_rows")
        return Table._adopt(h, self.device_id)

    def repair_pmf(self, model, target_col, feat_cols, top_k=32, threshold=0.0, cur_codes=None, want_cur_prob=False):
        """Candidate distributions of the NULL cells of ``target_col`` (model.py:1196-1212).  Returns
        (rows [m], classes [m][top_k] (-1 padded), probs [m][top_k] (0 padded)[, cur_prob [m]])."""
        fc = _i32(feat_cols)
        _, n_null = self.count_codes(target_col)
        rows = np.zeros(n_null, np.int64)
        cls = np.zeros((n_null, top_k), np.int32)
        pr = np.zeros((n_null, top_k), np.float64)
        cur = _i32(cur_codes)
        if cur is not None and len(cur) != n_null:
            raise ValueError("cur_codes must hold one code per NULL cell of the target (%d)" % n_null)
        cp = np.zeros(n_null, np.float64) if (want_cur_prob or cur is not None) else None
        n = C.c_int64(0)
        _check(lib().rgbm_table_repair_pmf(self.h, model.h, C.c_int32(target_col), _p(fc, C.c_int32), C.c_int32(len(fc)), C.c_int32(top_k),
                                           C.c_double(threshold), _p(cur, C.c_int32), C.c_int64(n_null), C.byref(n), _p(rows, C.c_int64),
                                           _p(cls, C.c_int32), _p(pr, C.c_double), _p(cp, C.c_double)), "rgbm_table_repair_pmf")
        assert int(n.value) == n_null
        return (rows, cls, pr, cp) if cp is not None else (rows, cls, pr)

    def count_codes(self, col):
        """(rows per code [n_codes[col]], NULL rows) of one column."""
        out = np.zeros(int(self.n_codes[col]), np.int64)
        nn = C.c_int64(0)
        _check(lib().rgbm_table_count_codes(self.h, C.c_int32(col), _p(out, C.c_int64), C.byref(nn)), "rgbm_table_count_codes")
        return out, int(nn.value)

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            _lib.rgbm_table_free(h)

    __del__ = close

    def train(self, target_col, feat_cols, y_value=None, class_weight=None, want_stats=False, **params):
        fc = _i32(feat_cols)
        yv, cw = _f64(y_value), _f64(class_weight)
        params.setdefault("device_id", self.device_id)
        p = make_params(**params)
        h = C.c_void_p()
        st = RgbmTrainStats()
        _check(lib().rgbm_table_train(self.h, C.c_int32(target_col), _p(fc, C.c_int32), C.c_int32(len(fc)), _p(yv, C.c_double),
                                      _p(cw, C.c_double), C.byref(p), C.byref(h), C.byref(st) if want_stats else None), "rgbm_table_train")
        m = Model(h)
        return (m, st.as_dict()) if want_stats else m

    def repair_chain(self, models, target_col, feat_cols, row_begin=0, n_rows=None, want_prob=True):
        T = len(models)
        n_rows = self.n - row_begin if n_rows is None else n_rows
        arr, fc, fo, _, _ = _chain_args(models, feat_cols, None)
        tc = _i32(target_col)
        lab = np.zeros((T, n_rows), np.int32)
        prob = np.zeros((T, n_rows), np.float64) if want_prob else None
        _check(lib().rgbm_table_repair_chain(self.h, arr, C.c_int32(T), _p(tc, C.c_int32), _p(fc, C.c_int32), _p(fo, C.c_int32),
                                             C.c_int64(row_begin), C.c_int64(n_rows), _p(lab, C.c_int32), _p(prob, C.c_double)),
               "rgbm_table_repair_chain")
        return lab, prob

    def set_row_multiplicity(self, mult):
        """Row i stands for mult[i] (1..255) identical rows of a larger table (repair.pipeline.distinct_rows): every later train() on this
        table returns the model of the EXPANDED table, byte for byte.  None clears."""
        m = None if mult is None else np.ascontiguousarray(mult, np.uint8)
        if m is not None and m.shape != (self.n,):
            raise ValueError("one multiplicity per row")
        _check(lib().rgbm_table_set_row_multiplicity(self.h, _p(m, C.c_uint8)), "rgbm_table_set_row_multiplicity")

    def repair_chain_gather(self, models, target_col, feat_cols, row_begin=0, n_rows=None):
        """The chained repair of THIS rank's rows, the outputs all-gathered over the calling thread's communicator on the device (C2):
        returns (labels [T][sum of the ranks' rows] in rank order, probs likewise, first row of this rank, rows of every rank)."""
        T = len(models)
        n_rows = self.n - row_begin if n_rows is None else n_rows
        counts = comm_all_gather_sizes([n_rows])[:, 0]
        mx = int(max(int(counts.max()), 1))
        nr = len(counts)
        arr, fc, fo, _, _ = _chain_args(models, feat_cols, None)
        tc = _i32(target_col)
        lab = np.zeros((nr, T, mx), np.int32)
        prob = np.zeros((nr, T, mx), np.float64)
        _check(lib().rgbm_table_repair_chain_gather(self.h, arr, C.c_int32(T), _p(tc, C.c_int32), _p(fc, C.c_int32), _p(fo, C.c_int32),
                                                    C.c_int64(row_begin), C.c_int64(n_rows), C.c_int64(mx), _p(lab, C.c_int32), _p(prob, C.c_double)),
               "rgbm_table_repair_chain_gather")
        labels = np.concatenate([lab[r, :, :int(counts[r])] for r in range(nr)], axis=1)
        probs = np.concatenate([prob[r, :, :int(counts[r])] for r in range(nr)], axis=1)
        me = comm_info()["rank"]
        return labels, probs, int(counts[:me].sum()), [int(x) for x in counts]

    def read_column(self, col):
        out = np.zeros(self.n, np.int32)
        _check(lib().rgbm_table_read_column(self.h, C.c_int32(col), _p(out, C.c_int32)), "rgbm_table_read_column")
        return out
