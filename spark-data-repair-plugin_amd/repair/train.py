"""Builds ONE statistical repair model -- the mirror of python/repair/train.py of the reference:
objective pick (train.py:97-100), fixed LightGBM parameters (102-115), hyper-parameter search over
the same 7-parameter space (148-156) scored by shuffled k-fold CV (158-173), early stop after
`model.hp.no_progress_loss` non-improving evaluations or `model.hp.timeout` seconds (181-195),
final fit on all rows (215-216), and the "any failure -> (None, 0.0)" contract (227-229).

hyperopt is not installed in this image (and its TPE draws cannot be reproduced offline), so the
sampler is a seeded random search whose first evaluation is LightGBM's defaults for the searched
parameters; `RandomState(42)` as in train.py:207.  Everything else keeps the reference's option
names: model.lgb.*, model.cv.n_splits, model.hp.{timeout,max_evals,no_progress_loss}.
"""
import copy
import time
from collections import namedtuple
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd

from repair.utils import elapsed_time, get_option_value, setup_logger

_logger = setup_logger()

_option = namedtuple("_option", "key default_value type_class validator err_msg")

_opt_boosting_type = _option("model.lgb.boosting_type", "gbdt", str,
                             lambda v: v in ["gbdt", "dart", "goss", "rf"], "`{}` should be in ['gbdt', 'dart', 'goss', 'rf']")
_opt_class_weight = _option("model.lgb.class_weight", "balanced", str, None, None)
_opt_learning_rate = _option("model.lgb.learning_rate", 0.01, float, lambda v: v > 0.0, "`{}` should be positive")
_opt_max_depth = _option("model.lgb.max_depth", 7, int, None, None)
_opt_max_bin = _option("model.lgb.max_bin", 255, int, None, None)
_opt_reg_alpha = _option("model.lgb.reg_alpha", 0.0, float, lambda v: v >= 0.0, "`{}` should be greater than or equal to 0.0")
_opt_min_split_gain = _option("model.lgb.min_split_gain", 0.0, float, lambda v: v >= 0.0, "`{}` should be greater than or equal to 0.0")
_opt_n_estimators = _option("model.lgb.n_estimators", 300, int, lambda v: v > 0, "`{}` should be positive")
_opt_importance_type = _option("model.lgb.importance_type", "gain", str, lambda v: v in ["split", "gain"], "`{}` should be in ['split', 'gain']")
_opt_n_splits = _option("model.cv.n_splits", 3, int, lambda v: v >= 3, "`{}` should be greater than 2")
_opt_timeout = _option("model.hp.timeout", 0, int, None, None)
_opt_max_evals = _option("model.hp.max_evals", 100000000, int, lambda v: v > 0, "`{}` should be positive")
_opt_no_progress_loss = _option("model.hp.no_progress_loss", 50, int, lambda v: v > 0, "`{}` should be positive")

train_option_keys = [o.key for o in (
    _opt_boosting_type, _opt_class_weight, _opt_learning_rate, _opt_max_depth, _opt_max_bin, _opt_reg_alpha,
    _opt_min_split_gain, _opt_n_estimators, _opt_importance_type, _opt_n_splits, _opt_timeout, _opt_max_evals,
    _opt_no_progress_loss)]

# LightGBM defaults of the searched parameters = evaluation #0 of the search
_DEFAULT_POINT = dict(num_leaves=31, subsample=1.0, subsample_freq=0, colsample_bytree=1.0, min_child_samples=20,
                      min_child_weight=1e-3, reg_lambda=0.0)


def _sample_point(rs: np.random.RandomState) -> Dict[str, Any]:
    """One draw from the reference's search space (train.py:148-156)."""
    return dict(num_leaves=int(round(rs.uniform(2, 100))), subsample=float(rs.uniform(0.5, 1.0)),
                subsample_freq=int(round(rs.uniform(1, 20))), colsample_bytree=float(rs.uniform(0.01, 1.0)),
                min_child_samples=int(round(rs.uniform(1, 50))), min_child_weight=float(np.exp(rs.uniform(-3, 1))),
                reg_lambda=float(np.exp(rs.uniform(-2, 3))))


def fixed_params(opts: Dict[str, str], is_discrete: bool, num_class: int, n_jobs: int) -> Dict[str, Any]:
    g = lambda o: get_option_value(opts, *o)  # noqa: E731
    objective = ("binary" if num_class <= 2 else "multiclass") if is_discrete else "regression"
    p = {"boosting_type": g(_opt_boosting_type), "objective": objective, "class_weight": g(_opt_class_weight),
         "learning_rate": g(_opt_learning_rate), "max_depth": g(_opt_max_depth), "max_bin": g(_opt_max_bin),
         "reg_alpha": g(_opt_reg_alpha), "min_split_gain": g(_opt_min_split_gain), "n_estimators": g(_opt_n_estimators),
         "importance_type": g(_opt_importance_type), "random_state": 42, "n_jobs": n_jobs}
    if objective == "multiclass":
        p["num_class"] = num_class
    return p


def _cv_loss(model: Any, X: pd.DataFrame, y: pd.Series, is_discrete: bool, n_splits: int, seed: int) -> float:
    from sklearn.base import clone
    from sklearn.metrics import f1_score, mean_squared_error
    from sklearn.model_selection import KFold, StratifiedKFold
    cv = StratifiedKFold(n_splits=n_splits, shuffle=True, random_state=seed) if is_discrete \
        else KFold(n_splits=n_splits, shuffle=True, random_state=seed)
    scores = []
    for tr, va in cv.split(X, y):
        m = clone(model).fit(X.iloc[tr], y.iloc[tr])
        pred = m.predict(X.iloc[va])
        scores.append(f1_score(y.iloc[va], pred, average="macro") if is_discrete else -mean_squared_error(y.iloc[va], pred))
    return -float(np.mean(scores))


@elapsed_time  # type: ignore
def _build_gbm_model(X: pd.DataFrame, y: pd.Series, is_discrete: bool, num_class: int, n_jobs: int,
                     opts: Dict[str, str]) -> Tuple[Any, float]:
    from repair.gbm import RepairGBMClassifier, RepairGBMRegressor
    g = lambda o: get_option_value(opts, *o)  # noqa: E731
    base = fixed_params(opts, is_discrete, num_class, n_jobs)
    model_class = RepairGBMClassifier if is_discrete else RepairGBMRegressor

    def _create_model(params: Dict[str, Any]) -> Any:
        p = copy.deepcopy(base)
        p.update(params)
        for k in ("num_leaves", "subsample_freq", "min_child_samples"):
            if k in p:
                p[k] = int(p[k])
        return model_class(**p)

    n_splits = int(g(_opt_n_splits))
    max_evals = int(g(_opt_max_evals))
    patience = int(g(_opt_no_progress_loss))
    timeout = int(g(_opt_timeout))
    try:
        rs = np.random.RandomState(42)
        trials: List[Tuple[float, Dict[str, Any]]] = []
        best_loss, since_best, start = None, 0, time.time()
        while len(trials) < max_evals:
            point = dict(_DEFAULT_POINT) if not trials else _sample_point(rs)
            if max_evals == 1:
                loss = 0.0   # a single evaluation decides nothing: skip the CV fits
            else:
                try:
                    loss = _cv_loss(_create_model(point), X, y, is_discrete, n_splits, seed=len(trials))
                except Exception as e:   # e.g. a fold misses a label (train.py:175-179)
                    _logger.warning("%s: %s" % (e.__class__, e))
                    loss = 0.0
            trials.append((loss, point))
            if best_loss is None or loss < best_loss:
                best_loss, since_best = loss, 0
            else:
                since_best += 1
            if since_best >= patience or (timeout > 0 and time.time() - start > timeout):
                break
        _logger.info("hyperopt: #eval=%d/%d" % (len(trials), max_evals))
        best = min(trials, key=lambda t: t[0])
        model = _create_model(best[1])
        model.fit(X, y)
        imp = sorted(((n, v) for n, v in zip(model.feature_name_, model.feature_importances_) if v > 0.0), key=lambda x: -x[1])
        _logger.debug("repairgbm: feature_importances=%s" % imp)
        return model, -best[0]
    except Exception as e:
        _logger.warning("Failed to build a stat model because: %s" % e)
        return None, 0.0


def build_model(X: pd.DataFrame, y: pd.Series, is_discrete: bool, num_class: int, n_jobs: int,
                opts: Dict[str, str]) -> Tuple[Any, float]:
    """((model, score), elapsed_seconds) -- same shape as the reference's build_model (train.py:232-234)."""
    return _build_gbm_model(X, y, is_discrete, num_class, n_jobs, opts)


def compute_class_nrow_stdv(y: pd.Series, is_discrete: bool) -> Optional[float]:
    from collections import Counter
    return float(np.std(list(Counter(y).values()))) if is_discrete else None


def rebalance_training_data(X: pd.DataFrame, y: pd.Series, target: str) -> Tuple[pd.DataFrame, pd.Series]:
    """reference train.py:242-293 uses imbalanced-learn's SMOTEN + RandomUnderSampler (not installed here and
    default-off: model.py:180).  This keeps the reference's intent -- every class resampled to the median
    class size -- with seeded duplication / sub-sampling."""
    from collections import Counter
    rs = np.random.RandomState(42)
    hist = Counter(y)
    median = int(np.median(list(hist.values())))
    parts = []
    for label, cnt in hist.items():
        idx = np.flatnonzero((y == label).to_numpy())
        if cnt > median:
            idx = rs.choice(idx, median, replace=False)
        elif cnt < median and cnt > 5:
            idx = np.concatenate([idx, rs.choice(idx, median - cnt, replace=True)])
        parts.append(idx)
    sel = np.sort(np.concatenate(parts))
    _logger.info("Rebalanced training data (y=%s, median=%d): #rows=%d -> #rows=%d" % (target, median, len(X), len(sel)))
    return X.iloc[sel], y.iloc[sel]
