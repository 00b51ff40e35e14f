"""scikit-learn-protocol estimators over librepairgbm.so -- the drop-in for
``lgb.LGBMClassifier`` / ``lgb.LGBMRegressor`` at the reference's call sites:

  train.py:121-131   model_class(**p)              -> RepairGBMClassifier/Regressor(**p)
  train.py:171-172   cross_val_score(model, X, y)  -> get_params/set_params/fit/predict (clone-able)
  train.py:216,219   fit, feature_name_, feature_importances_
  model.py:1120-1130 predict_proba, classes_, predict
  model.py:910,921   pickle.dumps / loads          -> __getstate__/__setstate__ (serialised trees)

Constructor keywords are LightGBM's sklearn names; they are mapped onto the core names of
``rgbm_params`` exactly as lightgbm/sklearn.py does (subsample->bagging_fraction, ...).
"""
from typing import Any, Dict, Optional

import numpy as np
import pandas as pd

from repair import _native
from repair.encode import TableEncoder

_backend = _native   # tests may swap in a module with the same train()/Model API (set_backend)


def set_backend(mod: Any) -> Any:
    """Select the compute backend module (default: the HIP library).  Returns the previous one."""
    global _backend
    prev, _backend = _backend, mod
    return prev


def get_backend() -> Any:
    return _backend


_SK_DEFAULTS: Dict[str, Any] = dict(
    boosting_type="gbdt", num_leaves=31, max_depth=-1, learning_rate=0.1, n_estimators=100,
    subsample_for_bin=200000, objective=None, class_weight=None, min_split_gain=0.0, min_child_weight=1e-3,
    min_child_samples=20, subsample=1.0, subsample_freq=0, colsample_bytree=1.0, reg_alpha=0.0, reg_lambda=0.0,
    random_state=None, n_jobs=-1, importance_type="split", max_bin=255, min_data_in_bin=3, num_class=None, device_id=0)


def _as_frame(X: Any) -> pd.DataFrame:
    if isinstance(X, pd.DataFrame):
        return X
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError("X should be 2-dimensional")
    return pd.DataFrame(X, columns=["Column_%d" % i for i in range(X.shape[1])])


class _RepairGBMBase:
    _is_classifier = True
    _estimator_type = "classifier"

    def __init__(self, **kwargs: Any) -> None:
        unknown = set(kwargs) - set(_SK_DEFAULTS)
        if unknown:
            raise TypeError("unknown parameter(s): %s" % sorted(unknown))
        for k, v in _SK_DEFAULTS.items():
            setattr(self, k, kwargs.get(k, v))
        self._model = None
        self._encoder: Optional[TableEncoder] = None

    # -- sklearn.base.clone protocol
    def get_params(self, deep: bool = True) -> Dict[str, Any]:
        return {k: getattr(self, k) for k in _SK_DEFAULTS}

    def set_params(self, **params: Any) -> "_RepairGBMBase":
        for k, v in params.items():
            if k not in _SK_DEFAULTS:
                raise ValueError("Invalid parameter %s" % k)
            setattr(self, k, v)
        return self

    def __sklearn_tags__(self):  # sklearn >= 1.6
        from sklearn.utils import Tags, TargetTags, ClassifierTags, RegressorTags, InputTags
        return Tags(estimator_type="classifier" if self._is_classifier else "regressor",
                    target_tags=TargetTags(required=True),
                    classifier_tags=ClassifierTags() if self._is_classifier else None,
                    regressor_tags=None if self._is_classifier else RegressorTags(),
                    input_tags=InputTags(allow_nan=True, string=True))

    def _core_params(self, objective: int, num_class: int) -> Dict[str, Any]:
        if self.boosting_type != "gbdt":
            raise ValueError("only boosting_type='gbdt' is implemented by the HIP engine, got %r" % self.boosting_type)
        return dict(objective=objective, num_class=max(int(num_class), 2), n_estimators=int(self.n_estimators),
                    num_leaves=int(self.num_leaves), max_depth=int(self.max_depth), max_bin=int(self.max_bin),
                    min_data_in_leaf=int(self.min_child_samples), min_data_in_bin=int(self.min_data_in_bin),
                    bagging_freq=int(self.subsample_freq), seed=int(self.random_state) if self.random_state is not None else 0,
                    device_id=int(self.device_id), learning_rate=float(self.learning_rate), lambda_l1=float(self.reg_alpha),
                    lambda_l2=float(self.reg_lambda), min_gain_to_split=float(self.min_split_gain),
                    min_sum_hessian_in_leaf=float(self.min_child_weight), bagging_fraction=float(self.subsample),
                    feature_fraction=float(self.colsample_bytree))

    def _label_weights(self, counts: np.ndarray, labels: np.ndarray) -> Optional[np.ndarray]:
        cw = self.class_weight
        if cw is None:
            return None
        if isinstance(cw, str):
            if cw != "balanced":
                raise ValueError("class_weight must be 'balanced', a dict or None")
            present = int((counts > 0).sum())
            with np.errstate(divide="ignore"):
                return np.where(counts > 0, counts.sum() / (present * counts.astype(np.float64)), 0.0)
        return np.array([float(cw.get(l, 1.0)) for l in labels.tolist()], np.float64)

    def _encode_X(self, X: Any) -> np.ndarray:
        df = _as_frame(X)
        if list(df.columns) != self._encoder.columns:
            df = df.set_axis(self._encoder.columns, axis=1) if df.shape[1] == len(self._encoder.columns) else df
        return self._encoder.encode(df)

    @property
    def feature_name_(self):
        return list(self._encoder.columns)

    @property
    def n_features_in_(self) -> int:
        return len(self._encoder.columns)

    @property
    def feature_importances_(self) -> np.ndarray:
        return self._model.importance("gain" if self.importance_type == "gain" else "split")

    @property
    def booster_bytes_(self) -> bytes:
        return self._model.save()

    def __getstate__(self) -> Dict[str, Any]:
        st = {k: v for k, v in self.__dict__.items() if k != "_model"}
        st["_model_bytes"] = self._model.save() if self._model is not None else None
        return st

    def __setstate__(self, st: Dict[str, Any]) -> None:
        blob = st.pop("_model_bytes", None)
        self.__dict__.update(st)
        # trees are host-resident in the handle; the device mirror is re-created lazily in this process
        self._model = _backend.Model.load(blob) if blob is not None else None


class RepairGBMClassifier(_RepairGBMBase):
    _is_classifier = True
    _estimator_type = "classifier"

    def fit(self, X: Any, y: Any, sample_weight: Any = None) -> "RepairGBMClassifier":
        df = _as_frame(X)
        self._encoder = TableEncoder(df, list(df.columns))
        codes = self._encoder.encode(df)
        y = np.asarray(y.to_numpy() if hasattr(y, "to_numpy") else y)
        if pd.isna(y).any():
            raise ValueError("y contains NULLs")
        self.classes_, y_code = np.unique(y, return_inverse=True)
        k = len(self.classes_)
        self.n_classes_ = k
        num_class = int(self.num_class) if self.num_class else k
        objective = 0 if (self.objective == "binary" or (self.objective is None and k <= 2)) else 1
        if objective == 0 and k > 2:
            raise ValueError("binary objective with %d classes" % k)
        if objective == 1 and num_class < max(k, 2):
            raise ValueError("num_class (%d) is smaller than the number of labels in y (%d)" % (num_class, k))
        cw = self._label_weights(np.bincount(y_code, minlength=k), self.classes_)
        sw = None if sample_weight is None else np.asarray(sample_weight, np.float64)
        self._model = _backend.train(codes, self._encoder.n_codes, y_code.astype(np.int32), k, class_weight=cw,
                                     sample_weight=sw, **self._core_params(objective, num_class))
        self._objective = objective
        return self

    def predict_proba(self, X: Any) -> np.ndarray:
        p = self._model.predict(self._encode_X(X), **({"device_id": int(self.device_id)} if _backend is _native else {}))
        return p[:, :max(len(self.classes_), 2)] if self._objective == 1 else p

    def predict(self, X: Any) -> np.ndarray:
        p = self.predict_proba(X)
        idx = np.argmax(p[:, :len(self.classes_)], axis=1) if len(self.classes_) > 1 else np.zeros(len(p), np.int64)
        return self.classes_[idx]

    def score(self, X: Any, y: Any) -> float:
        return float((self.predict(X) == np.asarray(y)).mean())


class RepairGBMRegressor(_RepairGBMBase):
    _is_classifier = False
    _estimator_type = "regressor"

    def fit(self, X: Any, y: Any, sample_weight: Any = None) -> "RepairGBMRegressor":
        df = _as_frame(X)
        self._encoder = TableEncoder(df, list(df.columns))
        codes = self._encoder.encode(df)
        y = np.asarray(y.to_numpy() if hasattr(y, "to_numpy") else y, np.float64)
        if np.isnan(y).any():
            raise ValueError("y contains NULLs")
        values, y_code = np.unique(y, return_inverse=True)
        # LGBMModel.fit applies class_weight to regressors too: every distinct y is a "class"
        cw = self._label_weights(np.bincount(y_code, minlength=len(values)), values)
        sw = None if sample_weight is None else np.asarray(sample_weight, np.float64)
        self._model = _backend.train(codes, self._encoder.n_codes, y_code.astype(np.int32), len(values), y_value=values,
                                     class_weight=cw, sample_weight=sw, **self._core_params(2, 1))
        return self

    def predict(self, X: Any) -> np.ndarray:
        return self._model.predict(self._encode_X(X), **({"device_id": int(self.device_id)} if _backend is _native else {}))[:, 0]

    def score(self, X: Any, y: Any) -> float:
        y = np.asarray(y, np.float64)
        r = y - self.predict(X)
        return float(1.0 - (r * r).sum() / max(((y - y.mean()) ** 2).sum(), 1e-300))
