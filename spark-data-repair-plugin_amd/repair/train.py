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


def run_search(opts: Dict[str, str], n_folds_of: Any, fold_score: Any, batch_scores: Any = None) -> Tuple[Dict[str, Any], float, int]:
    """The hyper-parameter search loop of train.py:133-209, independent of WHERE a fit runs.

    n_folds_of(seed) -> the CV folds of one evaluation (a list; shuffled with `seed`, train.py:158-173)
    fold_score(point, fold) -> the score of one fit (macro-F1 / -MSE) -- called from a thread pool, several at a time
    batch_scores([(point, fold), ...]) -> [score or Exception, ...] (optional): ALL fits of a batch of evaluations in one call -- the
        resident-table search hands them to the batched device trainer (rgbm_table_train_batch: one launch sequence for the whole
        batch) instead of one training call per fit; same fits, same scores, same accounting
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
                jobs: List[Any] = []
                for i, point in enumerate(points):
                    try:
                        fl = n_folds_of(len(trials) + i)
                        if batch_scores is not None:
                            futs.append([len(jobs) + j for j in range(len(fl))])
                            jobs.extend((point, fold) for fold in fl)
                        else:
                            futs.append([pool.submit(fold_score, point, fold) for fold in fl])
                    except Exception as e:   # noqa: BLE001
                        futs.append(e)
                if batch_scores is not None and jobs:
                    scored = batch_scores(jobs)

                    class _Done:   # the future protocol of the loop below
                        def __init__(self, v: Any) -> None:
                            self.v = v

                        def result(self) -> Any:
                            if isinstance(self.v, Exception):
                                raise self.v
                            return self.v
                    futs = [f if isinstance(f, Exception) else [_Done(scored[j]) for j in f] for f in futs]
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


def _ordinal_codes(col: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """OrdinalEncoder of one column: (codes, sorted categories)."""
    try:
        cats, codes = np.unique(col, return_inverse=True)
    except TypeError:      # mixed types: order by the string form
        keys = np.asarray([str(v) for v in col], dtype=object)
        _, first, codes = np.unique(keys, return_index=True, return_inverse=True)
        cats = col[first]
    return codes.astype(np.int64), cats


def smoten_resample(X: pd.DataFrame, y: pd.Series, targets: Dict[Any, int], k_neighbors: int = 5, random_state: int = 42) -> Tuple[pd.DataFrame, pd.Series]:
    """Restatement of imbalanced-learn 0.8.0's `SMOTEN(random_state, sampling_strategy=dict, k_neighbors).fit_resample` (the
    reference pins imbalanced-learn==0.8.0, bin/requirements.txt:10, and calls it at train.py:271-274; the package is not installed
    here, so this follows the published algorithm, Chawla et al. 2002 section 6.2, as that release implements it):

      * every column is nominal: ordinal codes of its sorted distinct values;
      * Value Difference Metric fitted on ALL rows: p_f(c | v) = share of class c among the rows with value v in feature f;
        d(a, b) = sum over features of (sum over classes |p_f(c | a_f) - p_f(c | b_f)|) ** 2   (k = 1, r = 2);
      * per class to grow (`targets`: class -> number of rows it should end up with), in ascending class order (as imbalanced-learn sorts a dict strategy): the k nearest neighbours of
        every row of the class AMONG the rows of the class (the row itself excluded); a fresh RandomState(random_state) draws, with
        replacement, the rows to grow from; a new row takes, feature by feature, the most common value among the drawn row's
        neighbours (ties: the smallest code);
      * the new rows are appended class by class.
    Neighbour ties are broken by row position (scikit-learn's brute-force k-neighbours leaves their order to an unstable sort)."""
    cols = list(X.columns)
    Xv = X.to_numpy(dtype=object)
    yv = y.to_numpy()
    n, F = Xv.shape
    codes = np.empty((n, F), np.int64)
    cats = []
    for j in range(F):
        codes[:, j], c = _ordinal_codes(Xv[:, j])
        cats.append(c)
    classes, y_idx = np.unique(yv, return_inverse=True)
    proba = []
    for j in range(F):
        cnt = np.zeros((len(cats[j]), len(classes)), np.float64)
        np.add.at(cnt, (codes[:, j], y_idx), 1.0)
        proba.append(cnt / cnt.sum(axis=1, keepdims=True))
    new_X, new_y = [], []
    for klass, n_target in sorted(targets.items(), key=lambda kv: kv[0]):     # imbalanced-learn sorts a dict strategy by class (check_sampling_strategy)
        rows = np.flatnonzero(yv == klass)
        n_new = int(n_target) - len(rows)
        if n_new <= 0 or len(rows) <= k_neighbors:
            continue
        Xc = codes[rows]
        dist = np.zeros((len(rows), len(rows)), np.float64)
        for j in range(F):
            pj = proba[j][Xc[:, j]]
            dj = np.zeros((len(rows), len(rows)), np.float64)
            for c in range(pj.shape[1]):          # class by class: peak memory stays O(m^2), not O(m^2 x classes)
                dj += np.abs(pj[:, c][:, None] - pj[:, c][None, :])
            dist += dj ** 2
        np.fill_diagonal(dist, -1.0)          # the row itself comes first and is dropped
        nn = np.argsort(dist, axis=1, kind="stable")[:, 1:k_neighbors + 1]
        rs = np.random.RandomState(random_state)
        picks = rs.choice(np.arange(len(rows)), size=n_new, replace=True)
        neigh = Xc[nn[picks]]                 # [n_new][k][F]
        out = np.empty((n_new, F), dtype=object)
        for j in range(F):
            counts = np.zeros((n_new, len(cats[j])), np.int64)
            np.add.at(counts, (np.repeat(np.arange(n_new), k_neighbors), neigh[:, :, j].ravel()), 1)
            out[:, j] = cats[j][counts.argmax(axis=1)]      # argmax: first maximum = smallest code
        new_X.append(out); new_y.append(np.full(n_new, klass, dtype=yv.dtype))
    if not new_X:
        return X, y
    Xn = pd.DataFrame(np.concatenate(new_X), columns=cols).astype(X.dtypes.to_dict(), errors="ignore")
    return pd.concat([X, Xn], ignore_index=True), pd.concat([y, pd.Series(np.concatenate(new_y), name=y.name)], ignore_index=True)


def random_under_sample(X: pd.DataFrame, y: pd.Series, targets: Dict[Any, int], random_state: int = 42) -> Tuple[pd.DataFrame, pd.Series]:
    """imbalanced-learn 0.8.0's `RandomUnderSampler(random_state, sampling_strategy=dict).fit_resample` (reference train.py:284-286): one
    RandomState, the classes in sorted order, `targets[c]` rows of class c drawn without replacement, the other classes whole."""
    rs = np.random.RandomState(random_state)
    yv = y.to_numpy()
    keep = []
    for klass in np.unique(yv):
        rows = np.flatnonzero(yv == klass)
        if klass in targets:
            rows = rows[rs.choice(np.arange(len(rows)), size=int(targets[klass]), replace=False)]
        keep.append(rows)
    idx = np.concatenate(keep)
    return X.iloc[idx].reset_index(drop=True), y.iloc[idx].reset_index(drop=True)


def rebalance_training_data(X: pd.DataFrame, y: pd.Series, target: str) -> Tuple[pd.DataFrame, pd.Series]:
    """reference train.py:242-293: every class is resampled to the MEDIAN class size -- SMOTEN over the rows without NULLs for the
    classes below it (those with more than k = 5 clean rows; the others are left alone with a warning), RandomUnderSampler for the
    classes above it.  imbalanced-learn is not installed here: `smoten_resample` / `random_under_sample` restate the two samplers."""
    from collections import Counter
    prev_nrows, prev_stdv = len(X), compute_class_nrow_stdv(y, is_discrete=True)
    hist = dict(Counter(y).items())
    median = int(np.median(list(hist.values())))
    X = X.reset_index(drop=True); y = y.reset_index(drop=True)
    has_na = X.isnull().any(axis=1).to_numpy()
    X_notna, y_notna, X_na, y_na = X[~has_na], y[~has_na], X[has_na], y[has_na]
    hist_na = dict(Counter(y_na).items())
    kn = 5
    smote_targets: Dict[Any, int] = {}
    for key, count in hist.items():
        if count < median:
            nna = hist_na.get(key, 0)
            if count - nna > kn:
                smote_targets[key] = median - nna
            else:
                _logger.warning("Over-sampling of '%s' in y='%s' failed because the number of the clean rows is too small: %d" % (key, target, count - nna))
    if smote_targets:
        X_notna, y_notna = smoten_resample(X_notna.reset_index(drop=True), y_notna.reset_index(drop=True), smote_targets, k_neighbors=kn, random_state=42)
    X = pd.concat([X_notna, X_na], ignore_index=True)
    y = pd.concat([y_notna, y_na], ignore_index=True)
    rus_targets = {k: median for k, c in hist.items() if c > median}
    if rus_targets:
        X, y = random_under_sample(X, y, rus_targets, random_state=42)
    _logger.info("Rebalanced training data (y=%s, median=%d): #rows=%d(stdv=%s) -> #rows=%d(stdv=%s)" % (
        target, median, prev_nrows, prev_stdv, len(X), compute_class_nrow_stdv(y, is_discrete=True)))
    return X, y
