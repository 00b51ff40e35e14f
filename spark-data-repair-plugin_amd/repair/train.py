"""Builds ONE statistical repair model -- the mirror of python/repair/train.py of the reference:
objective pick (train.py:97-100), fixed LightGBM parameters (102-115), hyper-parameter search over
the same 7-parameter space (148-156) scored by shuffled k-fold CV (158-173), early stop after
`model.hp.no_progress_loss` non-improving evaluations or `model.hp.timeout` seconds (181-195),
final fit on all rows (215-216), and the "any failure -> (None, 0.0)" contract (227-229).

hyperopt is not installed in this image (and its TPE draws cannot be reproduced offline), so the
sampler is a seeded random search whose first evaluation is LightGBM's defaults for the searched
parameters; `RandomState(42)` as in train.py:207.  Everything else keeps the reference's option
names: model.lgb.*, model.cv.n_splits, model.hp.{timeout,max_evals,no_progress_loss}.

On the device the search is BATCHED: every fit of a CV fold is an independent training call on its own HIP stream
(librepairgbm creates one per call and ctypes releases the GIL), and at the reference's default sample size
(<= 10 000 rows, model.py:755-766) one fit is launch/latency bound and uses a sliver of the GPU.  So the folds of
`model.hp.batch_size` consecutive trials are trained concurrently from a thread pool.  The outcome is identical to the
sequential search: points are drawn from the same RandomState in the same order and the no-progress / timeout rule is
applied to the losses in trial order (trials past the stopping point are evaluated in vain and discarded).
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
_opt_batch_size = _option("model.hp.batch_size", 8, int, lambda v: v > 0, "`{}` should be positive")
# new in this engine: which HIP device trains / predicts (one process per GPU: every rank names its own)
_opt_gpu_device_id = _option("model.gpu.device_id", 0, int, lambda v: v >= 0, "`{}` should be non-negative")

train_option_keys = [o.key for o in (
    _opt_boosting_type, _opt_class_weight, _opt_learning_rate, _opt_max_depth, _opt_max_bin, _opt_reg_alpha,
    _opt_min_split_gain, _opt_n_estimators, _opt_importance_type, _opt_n_splits, _opt_timeout, _opt_max_evals,
    _opt_no_progress_loss, _opt_batch_size)]

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
         "importance_type": g(_opt_importance_type), "random_state": 42, "n_jobs": n_jobs,
         "device_id": g(_opt_gpu_device_id)}
    if objective == "multiclass":
        p["num_class"] = num_class
    return p


def _cv_folds(X: pd.DataFrame, y: pd.Series, is_discrete: bool, n_splits: int, seed: int) -> List[Tuple[np.ndarray, np.ndarray]]:
    from sklearn.model_selection import KFold, StratifiedKFold
    cv = StratifiedKFold(n_splits=n_splits, shuffle=True, random_state=seed) if is_discrete \
        else KFold(n_splits=n_splits, shuffle=True, random_state=seed)
    return list(cv.split(X, y))


def _fold_score(model: Any, X: pd.DataFrame, y: pd.Series, is_discrete: bool, tr: np.ndarray, va: np.ndarray) -> float:
    from sklearn.base import clone
    from sklearn.metrics import f1_score, mean_squared_error
    m = clone(model).fit(X.iloc[tr], y.iloc[tr])
    pred = m.predict(X.iloc[va])
    return float(f1_score(y.iloc[va], pred, average="macro") if is_discrete else -mean_squared_error(y.iloc[va], pred))


def _cv_loss(model: Any, X: pd.DataFrame, y: pd.Series, is_discrete: bool, n_splits: int, seed: int) -> float:
    scores = [_fold_score(model, X, y, is_discrete, tr, va) for tr, va in _cv_folds(X, y, is_discrete, n_splits, seed)]
    return -float(np.mean(scores))


def search_fits_in_flight(opts: Dict[str, str]) -> int:
    """Fold fits `run_search` keeps in flight at once: `model.hp.batch_size` evaluations x `model.cv.n_splits` folds."""
    g = lambda o: get_option_value(opts, *o)  # noqa: E731
    return max(1, int(g(_opt_batch_size))) * int(g(_opt_n_splits))


def run_search(opts: Dict[str, str], n_folds_of: Any, fold_score: Any) -> Tuple[Dict[str, Any], float, int]:
    """The hyper-parameter search loop of train.py:133-209, independent of WHERE a fit runs.

    n_folds_of(seed) -> the CV folds of one evaluation (a list; shuffled with `seed`, train.py:158-173)
    fold_score(point, fold) -> the score of one fit (macro-F1 / -MSE) -- called from a thread pool, several at a time
    Returns (best point, best score, evaluations).  Evaluation #0 is LightGBM's defaults for the searched parameters; the folds of
    `model.hp.batch_size` consecutive evaluations are in flight together (every fit owns a HIP stream); the no-progress / timeout
    rule is applied to the losses in trial order, so the outcome equals the sequential search."""
    from concurrent.futures import ThreadPoolExecutor
    g = lambda o: get_option_value(opts, *o)  # noqa: E731
    n_splits = int(g(_opt_n_splits))
    max_evals = int(g(_opt_max_evals))
    patience = int(g(_opt_no_progress_loss))
    timeout = int(g(_opt_timeout))
    batch = max(1, int(g(_opt_batch_size)))
    rs = np.random.RandomState(42)
    trials: List[Tuple[float, Dict[str, Any]]] = []
    best_loss, since_best, start = None, 0, time.time()
    stop = False
    with ThreadPoolExecutor(max_workers=batch * n_splits) as pool:
        while len(trials) < max_evals and not stop:
            # draw the next `batch` points (same RandomState stream as a sequential search) ...
            nb = min(batch, max_evals - len(trials))
            points = [dict(_DEFAULT_POINT) if (not trials and i == 0) else _sample_point(rs) for i in range(nb)]
            futs: List[Any] = []
            if max_evals > 1:   # a single evaluation decides nothing: skip the CV fits
                # ... train every CV fold of every point of the batch concurrently (one HIP stream per fit) ...
                for i, point in enumerate(points):
                    try:
                        futs.append([pool.submit(fold_score, point, fold) for fold in n_folds_of(len(trials) + i)])
                    except Exception as e:   # noqa: BLE001
                        futs.append(e)
            # ... and account for the losses in trial order, exactly like the sequential loop
            for i, point in enumerate(points):
                if max_evals == 1:
                    loss = 0.0
                else:
                    try:
                        if isinstance(futs[i], Exception):
                            raise futs[i]
                        loss = -float(np.mean([f.result() for f in futs[i]]))
                    except Exception as e:   # e.g. a fold misses a label (train.py:175-179)
                        _logger.warning("%s: %s" % (e.__class__, e))
                        loss = 0.0
                if stop:
                    continue   # evaluated in vain: past the stopping point of the sequential search
                trials.append((loss, point))
                if best_loss is None or loss < best_loss:
                    best_loss, since_best = loss, 0
                else:
                    since_best += 1
                if since_best >= patience or (timeout > 0 and time.time() - start > timeout):
                    stop = True
    _logger.info("hyperopt: #eval=%d/%d" % (len(trials), max_evals))
    best = min(trials, key=lambda t: t[0])
    return best[1], -best[0], len(trials)


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
    try:
        point, score, _ = run_search(opts, lambda seed: _cv_folds(X, y, is_discrete, n_splits, seed=seed),
                                     lambda point, fold: _fold_score(_create_model(point), X, y, is_discrete, fold[0], fold[1]))
        model = _create_model(point)
        model.fit(X, y)
        imp = sorted(((n, v) for n, v in zip(model.feature_name_, model.feature_importances_) if v > 0.0), key=lambda x: -x[1])
        _logger.debug("repairgbm: feature_importances=%s" % imp)
        return model, score
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
