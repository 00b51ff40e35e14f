"""ctypes loader for the CPU oracle (oracle/rgbm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Mirrors the product's C-ABI (include/rgbm.h) one-to-one with an ``orc_`` prefix so that parity
tests call both sides with the same arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "librgbm_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("objective", C.c_int32), ("num_class", C.c_int32),
        ("n_estimators", C.c_int32), ("num_leaves", C.c_int32), ("max_depth", C.c_int32), ("max_bin", C.c_int32),
        ("min_data_in_leaf", C.c_int32), ("min_data_in_bin", C.c_int32), ("bagging_freq", C.c_int32), ("seed", C.c_int32),
        ("device_id", C.c_int32), ("reserved", C.c_int32),
        ("learning_rate", C.c_double), ("lambda_l1", C.c_double), ("lambda_l2", C.c_double), ("min_gain_to_split", C.c_double),
        ("min_sum_hessian_in_leaf", C.c_double), ("bagging_fraction", C.c_double), ("feature_fraction", C.c_double),
    ]


DEFAULTS = dict(objective=1, num_class=2, n_estimators=300, num_leaves=31, max_depth=7, max_bin=255,
                min_data_in_leaf=20, min_data_in_bin=3, bagging_freq=0, seed=42, device_id=-1, reserved=0,
                learning_rate=0.01, lambda_l1=0.0, lambda_l2=0.0, min_gain_to_split=0.0,
                min_sum_hessian_in_leaf=1e-3, bagging_fraction=1.0, feature_fraction=1.0)


def make_params(**kw):
    d = dict(DEFAULTS)
    d.update(kw)
    return OrcParams(**d)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("rgbm_oracle.c", "rgbm_oracle_train.inc")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_train.restype = C.c_int
        _lib.orc_train2.restype = C.c_int
        _lib.orc_train2_f32.restype = C.c_int
        _lib.orc_predict.restype = C.c_int
        _lib.orc_model_save.restype = C.c_int
        _lib.orc_model_load.restype = C.c_int
        _lib.orc_repair_chain.restype = C.c_int
        _lib.orc_exp.restype = C.c_double
        _lib.orc_exp.argtypes = [C.c_double]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class OracleModel:
    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_free(self.h)
            self.h = None

    def info(self):
        a = np.zeros(5, np.int32)
        lib().orc_model_info(self.h, _p(a, C.c_int32))
        return dict(objective=int(a[0]), num_class=int(a[1]), K=int(a[2]), n_iter=int(a[3]), F=int(a[4]))

    def save(self):
        n = C.c_size_t(0)
        assert lib().orc_model_save(self.h, None, C.byref(n)) == 0
        buf = np.zeros(n.value, np.uint8)
        assert lib().orc_model_save(self.h, buf.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
        return buf.tobytes()

    @staticmethod
    def load(b):
        h = C.c_void_p()
        arr = np.frombuffer(b, np.uint8)
        rc = lib().orc_model_load(arr.ctypes.data_as(C.c_void_p), C.c_size_t(len(b)), C.byref(h))
        if rc:
            raise ValueError("orc_model_load failed: %d" % rc)
        return OracleModel(h)

    def predict(self, X):
        """X: [F][n] int32 codes (column-major).  Returns [n][ncol] float64."""
        X = np.ascontiguousarray(X, np.int32)
        F, n = X.shape
        inf = self.info()
        ncol = 1 if inf["objective"] == 2 else inf["num_class"]
        out = np.zeros((n, ncol), np.float64)
        rc = lib().orc_predict(self.h, _p(X, C.c_int32), C.c_int64(n), C.c_int32(F), _p(out, C.c_double))
        if rc:
            raise RuntimeError("orc_predict failed: %d" % rc)
        return out


def train(X, n_codes, y_code, n_y_codes, y_value=None, class_weight=None, sample_weight=None, feature_values=None,
          categorical=None, numerics="spec", **params):
    """X: [F][N] int32 codes column-major; y_code: [N] int32.  numerics: "spec" = the numerics the product implements (DESIGN.md
    section 3: LightGBM's float32 g / h per row, EXACT integer histogram sums on a fixed-point grid), "lightgbm_f32" = LightGBM's own
    arithmetic (the same float32 g / h, double histogram sums in row order) -- the two modes of oracle/rgbm_oracle_train.inc, compared by
    tests/test_numerics_bound.py.  feature_values: {feature index: ascending distinct values} of the
    NUMERIC features (bin bounds at value midpoints, like Table.set_column_values on the product side); categorical: indices of the
    CATEGORICAL features (codes no training row holds are missing at prediction time, like Table.set_column_kind)."""
    X = np.ascontiguousarray(X, np.int32)
    F, N = X.shape
    n_codes = np.ascontiguousarray(n_codes, np.int32)
    y_code = np.ascontiguousarray(y_code, np.int32)
    yv = None if y_value is None else np.ascontiguousarray(y_value, np.float64)
    cw = None if class_weight is None else np.ascontiguousarray(class_weight, np.float64)
    sw = None if sample_weight is None else np.ascontiguousarray(sample_weight, np.float64)
    p = make_params(**params)
    h = C.c_void_p()
    fv_arr, keep = None, []
    if feature_values:
        fv_arr = (C.POINTER(C.c_double) * F)()
        for f, v in feature_values.items():
            a = np.ascontiguousarray(v, np.float64)
            assert len(a) == int(n_codes[f]), "feature_values[%d] must list every code of the column" % f
            keep.append(a)
            fv_arr[f] = a.ctypes.data_as(C.POINTER(C.c_double))
    kinds = None
    if categorical:
        kinds = np.zeros(F, np.int32)
        kinds[list(categorical)] = 1
    fn = {"spec": lib().orc_train2, "lightgbm_f32": lib().orc_train2_f32}[numerics]
    rc = fn(_p(X, C.c_int32), C.c_int64(N), C.c_int32(F), _p(n_codes, C.c_int32),
            _p(y_code, C.c_int32), C.c_int32(n_y_codes), _p(yv, C.c_double),
            _p(cw, C.c_double), _p(sw, C.c_double), C.byref(p), fv_arr, _p(kinds, C.c_int32), C.byref(h))
    if rc:
        raise RuntimeError("orc_train failed: %d" % rc)
    return OracleModel(h)


def repair_chain(models, target_col, feat_cols, class_codes, table):
    """table: [C][n] int32 (modified in place). Returns (labels [T][n], probs [T][n])."""
    T = len(models)
    Cc, n = table.shape
    assert table.dtype == np.int32 and table.flags.c_contiguous
    arr = (C.c_void_p * T)(*[m.h for m in models])
    tc = np.ascontiguousarray(target_col, np.int32)
    fo = np.zeros(T + 1, np.int32)
    co = np.zeros(T + 1, np.int32)
    for t in range(T):
        fo[t + 1] = fo[t] + len(feat_cols[t])
        co[t + 1] = co[t] + len(class_codes[t])
    fc = np.ascontiguousarray(np.concatenate([np.asarray(f, np.int32) for f in feat_cols]) if T else np.zeros(0), np.int32)
    cc = np.ascontiguousarray(np.concatenate([np.asarray(c, np.int32) for c in class_codes]) if T else np.zeros(0), np.int32)
    if cc.size == 0:
        cc = np.zeros(1, np.int32)
    lab = np.zeros((T, n), np.int32)
    prob = np.zeros((T, n), np.float64)
    rc = lib().orc_repair_chain(arr, C.c_int32(T), _p(tc, C.c_int32), _p(fc, C.c_int32), _p(fo, C.c_int32),
                                _p(cc, C.c_int32), _p(co, C.c_int32), _p(table, C.c_int32), C.c_int64(n), C.c_int32(Cc),
                                _p(lab, C.c_int32), _p(prob, C.c_double))
    if rc:
        raise RuntimeError("orc_repair_chain failed: %d" % rc)
    return lab, prob
