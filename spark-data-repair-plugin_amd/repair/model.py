"""RepairModel -- the reference's fluent repair API (python/repair/model.py) on a Spark-free host,
with its hot path (`_build_repair_models` 955-1052, `_build_repair_stat_models_in_series` 768-815,
`_repair` 1062-1143) running on the MI355X engine through `repair.gbm` / librepairgbm.so.

Kept verbatim from the reference: setter names and validation messages, `option()` keys, `run()`
flags and their exclusivity rules, the output schemas
  (row_id, attribute, current_value, repaired[, prob | pmf | score]),
the estimator protocol of the per-attribute models (PoorModel / FunctionalDepModel / stat model),
"train on all rows with error cells NULLed, drop rows whose target is NULL", the fill-only-NULL
chain over target attributes, integral rounding, and "a failed build becomes PoorModel(None)".
Inputs are pandas DataFrames (or names registered with `repair.session.register_table`).
"""
import copy
import heapq
import json
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import pandas as pd

from repair import session
from repair.costs import UpdateCostFunction
from repair.encode import is_integral_column, is_numeric_column
from repair.errors import (ConstraintErrorDetector, ErrorDetector, ErrorModel, NullErrorDetector, RegExErrorDetector, parse_constraint, load_constraints,
                           parse_and_verify_constraints)
from repair.train import build_model, compute_class_nrow_stdv, rebalance_training_data, train_option_keys
from repair.train import _opt_gpu_device_id as _train_opt_gpu_device_id
from repair.utils import argtype_check, elapsed_time, get_option_value, job_group, setup_logger, to_list_str

_logger = setup_logger()

DataFrame = pd.DataFrame


class PoorModel():
    """Model to return the same value regardless of an input value (reference model.py:44-61)."""

    def __init__(self, v: Any) -> None:
        self.v = v

    @property
    def classes_(self) -> Any:
        return np.array([self.v])

    def predict(self, X: pd.DataFrame) -> Any:
        return [self.v] * len(X)

    def predict_proba(self, X: pd.DataFrame) -> Any:
        return [np.array([1.0])] * len(X)


class FunctionalDepModel():
    """sklearn-like model that follows a functional dependency x -> y (reference model.py:64-100)."""

    def __init__(self, x: str, fd_map: Dict[str, str]) -> None:
        self.fd_map = fd_map
        self.classes = sorted(set(fd_map.values()), key=str)   # deterministic (the reference's set order is hash-dependent)
        self.x = x
        self.fd_keypos_map = {c: i for i, c in enumerate(self.classes)}

    @property
    def classes_(self) -> Any:
        return np.array(self.classes)

    def predict(self, X: pd.DataFrame) -> Any:
        return [self.fd_map[x] if x in self.fd_map else None for x in X[self.x]]

    def predict_proba(self, X: pd.DataFrame) -> Any:
        pmf = []
        for x in X[self.x]:
            if x in self.fd_map:
                probs = np.zeros(len(self.classes))
                probs[self.fd_keypos_map[self.fd_map[x]]] = 1.0
                pmf.append(probs)
            else:
                _logger.warning('Unknown "%s" domain value found: %s' % (self.x, x))
                pmf.append(None)
        return pmf


class RepairModel():
    """Interface to detect error cells in given input data and build statistical models to repair them."""

    from collections import namedtuple
    _option = namedtuple("_option", "key default_value type_class validator err_msg")

    _opt_max_training_row_num = _option("model.max_training_row_num", 10000, int, lambda v: v >= 10, "`{}` should be greater than and equal to 10")
    _opt_max_training_column_num = _option("model.max_training_column_num", 65536, int, lambda v: v >= 2, "`{}` should be greater than 1")
    _opt_small_domain_threshold = _option("model.small_domain_threshold", 12, int, lambda v: v >= 3, "`{}` should be greater than 2")
    _opt_repair_by_regex_disabled = _option("model.rule.repair_by_regex.disabled", True, bool, None, None)
    _opt_repair_by_nearest_values_disabled = _option("model.rule.repair_by_nearest_values.disabled", True, bool, None, None)
    _opt_merge_threshold = _option("model.rule.merge_threshold", 2.0, float, None, None)
    _opt_repair_by_functional_deps_disabled = _option("model.rule.repair_by_functional_deps.disabled", False, bool, None, None)
    _opt_max_domain_size = _option("model.rule.max_domain_size", 1000, int, lambda v: v > 10, "`{}` should be greater than 10")
    _opt_cost_weight = _option("repair.pmf.cost_weight", 0.1, float, lambda v: v > 0.0, "`{}` should be positive")
    _opt_prob_threshold = _option("repair.pmf.prob_threshold", 0.0, float, None, None)
    _opt_prob_top_k = _option("repair.pmf.prob_top_k", 32, int, lambda v: v >= 3, "`{}` should be greater than 2")
    # new in this engine: which HIP device trains/predicts (read by repair.train.fixed_params)
    _opt_gpu_device_id = _train_opt_gpu_device_id

    option_keys = set([o.key for o in (
        _opt_max_training_row_num, _opt_max_training_column_num, _opt_small_domain_threshold, _opt_repair_by_regex_disabled,
        _opt_repair_by_nearest_values_disabled, _opt_merge_threshold, _opt_repair_by_functional_deps_disabled,
        _opt_max_domain_size, _opt_cost_weight, _opt_prob_threshold, _opt_prob_top_k, _opt_gpu_device_id)] +
        list(ErrorModel.option_keys) + list(train_option_keys))

    def __init__(self) -> None:
        self.db_name: str = ""
        self.input: Optional[Union[str, DataFrame]] = None
        self.row_id: Optional[str] = None
        self.targets: List[str] = []
        self.error_cells: Optional[Union[str, DataFrame]] = None
        self.error_detectors: List[ErrorDetector] = []
        self.discrete_thres: int = 80
        self.parallel_stat_training_enabled: bool = False
        self.training_data_rebalancing_enabled: bool = False
        self.repair_by_rules: bool = False
        self.repair_delta: Optional[int] = None
        self.repair_validation_enabled: bool = False
        self.cf: Optional[UpdateCostFunction] = None
        self.opts: Dict[str, str] = {}

    # ------------------------------------------------------------------ fluent setters
    @argtype_check  # type: ignore
    def setDbName(self, db_name: str) -> "RepairModel":
        if isinstance(self.input, DataFrame):
            raise ValueError("Can not specify a database name when input is `DataFrame`")
        self.db_name = db_name
        return self

    @argtype_check  # type: ignore
    def setTableName(self, table_name: str) -> "RepairModel":
        if not table_name:
            raise ValueError("`table_name` should have at least character")
        self.input = table_name
        return self

    @argtype_check  # type: ignore
    def setInput(self, input: Union[str, DataFrame]) -> "RepairModel":
        if type(input) is str:
            self.setTableName(input)
        else:
            self.db_name = ""
            self.input = input
        return self

    @argtype_check  # type: ignore
    def setRowId(self, row_id: str) -> "RepairModel":
        if not row_id:
            raise ValueError("`row_id` should have at least character")
        self.row_id = row_id
        return self

    @argtype_check  # type: ignore
    def setTargets(self, attrs: List[str]) -> "RepairModel":
        if len(attrs) == 0:
            raise ValueError("`attrs` should have at least one attribute")
        self.targets = attrs
        return self

    @argtype_check  # type: ignore
    def setErrorCells(self, error_cells: Union[str, DataFrame]) -> "RepairModel":
        if type(error_cells) is str and not error_cells:
            raise ValueError("`error_cells` should have at least character")
        if self.row_id is None:
            raise ValueError("`setRowId` should be called before specifying error cells")
        df = session.resolve(error_cells)
        if not all(c in df.columns for c in [self._row_id, "attribute"]):
            raise ValueError("Error cells should have `%s` and `attribute` in columns" % self.row_id)
        self.error_cells = error_cells
        return self

    @argtype_check  # type: ignore
    def setErrorDetectors(self, detectors: List[ErrorDetector]) -> "RepairModel":
        self.error_detectors = detectors
        return self

    @argtype_check  # type: ignore
    def setDiscreteThreshold(self, thres: int) -> "RepairModel":
        if int(thres) < 2:
            raise ValueError("`thres` should be bigger than 1, got %s" % thres)
        self.discrete_thres = thres
        return self

    @argtype_check  # type: ignore
    def setParallelStatTrainingEnabled(self, enabled: bool) -> "RepairModel":
        self.parallel_stat_training_enabled = enabled
        return self

    @argtype_check  # type: ignore
    def setTrainingDataRebalancingEnabled(self, enabled: bool) -> "RepairModel":
        self.training_data_rebalancing_enabled = enabled
        return self

    @argtype_check  # type: ignore
    def setRepairByRules(self, enabled: bool) -> "RepairModel":
        self.repair_by_rules = enabled
        return self

    @argtype_check  # type: ignore
    def setRepairDelta(self, delta: int) -> "RepairModel":
        if delta <= 0:
            raise ValueError("Repair delta should be positive, got %s" % delta)
        self.repair_delta = int(delta)
        return self

    @argtype_check  # type: ignore
    def setUpdateCostFunction(self, cf: UpdateCostFunction) -> "RepairModel":
        self.cf = cf
        return self

    @argtype_check  # type: ignore
    def option(self, key: str, value: str) -> "RepairModel":
        if key not in self.option_keys:
            raise ValueError("Non-existent key specified: key=%s" % key)
        self.opts[key] = value
        return self

    def _get_option_value(self, *args) -> Any:  # type: ignore
        return get_option_value(self.opts, *args)

    @property
    def _row_id(self) -> str:
        return str(self.row_id)

    @property
    def _repair_by_nearest_values_enabled(self) -> bool:
        return not bool(self._get_option_value(*self._opt_repair_by_nearest_values_disabled)) and self.repair_by_rules and self.cf is not None

    @property
    def _repair_by_regex_enabled(self) -> bool:
        return not bool(self._get_option_value(*self._opt_repair_by_regex_disabled)) and self.repair_by_rules

    @property
    def _repair_by_functional_deps_enabled(self) -> bool:
        return not bool(self._get_option_value(*self._opt_repair_by_functional_deps_disabled)) and self.repair_by_rules

    # ------------------------------------------------------------------ input checks
    def _qualified_input_name(self) -> str:
        """`db.table` / `table` as the reference's messages print it; a DataFrame input has no name of its own."""
        if isinstance(self.input, str):
            return ("%s.%s" % (self.db_name, self.input)) if self.db_name else self.input
        return "input"

    def _check_input_table(self) -> Tuple[DataFrame, List[str]]:
        """RepairApi.checkInputTable (RepairApi.scala:34-67)."""
        df = session.resolve(self.input)
        rid = self._row_id
        name = self._qualified_input_name()
        if rid not in df.columns:
            raise ValueError("Column '%s' does not exist in '%s'." % (rid, name))
        unsupported = [t for t in (_sql_type_name(df[c]) for c in df.columns) if t not in _SUPPORTED_TYPES]
        if unsupported:
            raise ValueError("Supported types are %s, but unsupported ones found: %s" % (",".join(_SUPPORTED_TYPES), ",".join(unsupported)))
        if len(df.columns) < 3:
            raise ValueError("A least three columns (`%s` columns + two more ones) in table '%s'" % (rid, name))
        if not df[rid].is_unique:
            raise ValueError("Uniqueness does not hold in column '%s' of table '%s' (# of distinct '%s': %d, # of rows: %d)" % (
                rid, name, rid, df[rid].nunique(dropna=False), len(df)))
        continous = [c for c in df.columns if c != rid and is_numeric_column(df[c])]
        _logger.info("input_table: (%d rows x %d columns)" % (len(df), len(df.columns) - 1))
        # positional index without copying the table (a frame of 10^6 object cells takes > 1 s to copy + consolidate): nothing
        # downstream writes into `input_df`, every step that changes cells works on its own copy
        if isinstance(df.index, pd.RangeIndex) and df.index.start == 0 and df.index.step == 1:
            return df, continous
        return df.set_axis(pd.RangeIndex(len(df)), axis=0, copy=False), continous

    def _detect_errors(self, input_df: DataFrame, continous_columns: List[str]) -> Any:
        ec = None
        if self.error_cells is not None:
            ec = session.resolve(self.error_cells)[[self._row_id, "attribute"]]
        em = ErrorModel(row_id=self._row_id, targets=self.targets, discrete_thres=self.discrete_thres,
                        error_detectors=self.error_detectors, error_cells=ec, opts=self.opts)
        return em.detect(input_df, continous_columns)

    def _prepare_repair_base_cells(self, input_df: DataFrame, error_cells_df: DataFrame, target_columns: List[str]) -> DataFrame:
        """RepairApi.convertErrorCellsToNull (RepairApi.scala:171-211)."""
        base = input_df.copy()
        pos = pd.Series(np.arange(len(base)), index=base[self._row_id].to_numpy())
        for a, grp in error_cells_df.groupby("attribute"):
            if a in target_columns:
                rows = pos.reindex(grp[self._row_id].to_numpy()).dropna().to_numpy(np.int64)
                col = base[a].astype(object) if not is_numeric_column(base[a]) else base[a].astype("float64") if not is_integral_column(base[a]) else base[a].astype("Int64")
                col.iloc[rows] = None if not is_numeric_column(base[a]) else pd.NA if is_integral_column(base[a]) else np.nan
                base[a] = col
        return base

    # ------------------------------------------------------------------ rule models
    def _get_functional_deps(self, target_columns: List[str]) -> Optional[Dict[str, List[str]]]:
        dets = [d for d in self.error_detectors if isinstance(d, ConstraintErrorDetector)]
        if len(dets) != 1:
            if len(dets) > 1:
                _logger.warning("Multiple constraint classes not supported for detecting functional deps")
            return None
        ced = dets[0]
        tg = [c for c in target_columns if c in ced.targets] if ced.targets else target_columns
        deps: Dict[str, List[str]] = {}
        for stmt in load_constraints(ced.constraint_path, ced.constraints):
            try:
                ps = parse_constraint(stmt)
            except Exception:
                continue
            # X -> Y  ==  EQ(t1.X,t2.X) & IQ(t1.Y,t2.Y)
            if len(ps) == 2 and ps[0].op == "EQ" and ps[1].op == "IQ" and ps[0].constant is None and ps[1].constant is None \
                    and ps[0].left == ps[0].right and ps[1].left == ps[1].right and ps[1].left in tg:
                deps.setdefault(ps[1].left, []).append(ps[0].left)
        return deps or None

    def _build_rule_model(self, train_df: DataFrame, x: str, y: str) -> Any:
        sub = train_df[[x, y]].dropna()
        g = sub.groupby(x)[y].agg(lambda s: s.iloc[0] if s.nunique() == 1 else None).dropna()
        return FunctionalDepModel(x, {k: v for k, v in g.items()})

    # ------------------------------------------------------------------ hot path: training
    def _select_features(self, pairwise_attr_stats: Dict[str, Any], y: str, features: List[str]) -> List[str]:
        max_cols = int(self._get_option_value(*self._opt_max_training_column_num))
        if max_cols < len(features) and y in pairwise_attr_stats:
            heap: List[Tuple[float, str]] = []
            for f, corr in map(tuple, pairwise_attr_stats[y]):
                if f in features:
                    heapq.heappush(heap, (float(corr), f))
            top: List[Tuple[float, str]] = []
            for corr, f in [heapq.heappop(heap) for _ in range(len(heap))]:
                if len(top) <= 1 or (corr >= 0.0 and len(top) < max_cols):
                    top.append((corr, f))
            features = [f for _, f in top]
        return features

    def _sample_training_data_from(self, df: DataFrame) -> DataFrame:
        """model.py:755-766; the reference's Bernoulli sample is unseeded, this one is seeded (42)."""
        max_rows = int(self._get_option_value(*self._opt_max_training_row_num))
        if len(df) > max_rows:
            _logger.info("To reduce training data, extracts %s%% samples from %d rows" % (100.0 * max_rows / len(df), len(df)))
            return df.sample(n=max_rows, random_state=42)
        return df

    def _build_repair_stat_models_in_series(self, models: Dict[str, Any], train_df: DataFrame, target_columns: List[str],
                                            continous_columns: List[str], num_class_map: Dict[str, int],
                                            feature_map: Dict[str, List[str]]) -> Dict[str, Any]:
        opts = dict(self.opts)
        for y in [c for c in target_columns if c not in models]:
            index = len(models) + 1
            df = train_df[train_df[y].notna()]
            if len(df) == 0:
                _logger.info("Skipping %d/%d model... type=classfier y=%s num_class=%s" % (index, len(target_columns), y, num_class_map[y]))
                models[y] = (PoorModel(None), feature_map[y], None)
                continue
            train_pdf = self._sample_training_data_from(df)
            is_discrete = y not in continous_columns
            X = train_pdf[feature_map[y]]
            X, y_ = rebalance_training_data(X, train_pdf[y], y) if is_discrete and self.training_data_rebalancing_enabled else (X, train_pdf[y])
            _logger.info("Building %d/%d model... type=%s y=%s features=%s #rows=%d%s" % (
                index, len(target_columns), "classfier" if is_discrete else "regressor", y, to_list_str(feature_map[y]), len(train_pdf),
                " #class=%d" % num_class_map[y] if num_class_map[y] > 0 else ""))
            (model, score), elapsed = build_model(X, y_, is_discrete, num_class_map[y], n_jobs=-1, opts=opts)
            if model is None:
                model = PoorModel(None)
            _logger.info("Finishes building '%s' model...  score=%s elapsed=%ss (class stdv=%s)" % (
                y, score, elapsed, compute_class_nrow_stdv(y_, is_discrete)))
            models[y] = (model, feature_map[y], None)
        return models

    def _resolve_prediction_order(self, models: Dict[str, Any], target_columns: List[str]) -> List[Any]:
        ordered, pending = [], list(target_columns)
        for y in target_columns:
            if not isinstance(models[y][0], FunctionalDepModel):
                ordered.append((y, models[y])); pending.remove(y)
        while pending:
            before = len(pending)
            for y in list(pending):
                if models[y][1][0] not in pending:
                    ordered.append((y, models[y])); pending.remove(y)
            assert len(pending) < before
        return ordered

    @job_group(name="repair model training")
    def _build_repair_models(self, train_df: DataFrame, target_columns: List[str], continous_columns: List[str],
                             domain_stats: Dict[str, int], pairwise_attr_stats: Dict[str, Any]) -> List[Any]:
        train_df = train_df.drop(columns=[self._row_id])
        functional_deps = self._get_functional_deps(target_columns) if self._repair_by_functional_deps_enabled else None
        _logger.info("[Repair Model Training Phase] Building %d models to repair the cells in %s" % (len(target_columns), to_list_str(target_columns)))
        models: Dict[str, Any] = {}
        num_class_map: Dict[str, int] = {}
        for y in target_columns:
            input_columns = [c for c in train_df.columns if c != y]
            is_discrete = y not in continous_columns
            num_class_map[y] = int(train_df[y].nunique(dropna=True)) if is_discrete else 0
            if is_discrete and num_class_map[y] <= 1:
                v = train_df[y].dropna().iloc[0] if num_class_map[y] == 1 else None
                models[y] = (PoorModel(v), input_columns, None)
            if y not in models and functional_deps is not None and y in functional_deps:
                max_dom = int(self._get_option_value(*self._opt_max_domain_size))
                fx = [x for x in functional_deps[y] if x in domain_stats and int(domain_stats[x]) < max_dom]
                if fx:
                    models[y] = (self._build_rule_model(train_df, fx[0], y), [fx[0]], None)
        if len(models) != len(target_columns):
            feature_map = {y: self._select_features(pairwise_attr_stats, y, [c for c in train_df.columns if c != y])
                           for y in target_columns if y not in models}
            # one process drives one GPU: the reference's "parallel" mode (model.py:817-926) is a
            # scheduling choice with identical maths, so both settings train here in series.
            models = self._build_repair_stat_models_in_series(models, train_df, target_columns, continous_columns, num_class_map, feature_map)
        assert len(models) == len(target_columns)
        if any(isinstance(m, FunctionalDepModel) for m, _, _ in models.values()):
            return self._resolve_prediction_order(models, target_columns)
        return [(y, models[y]) for y in target_columns]

    # ------------------------------------------------------------------ hot path: inference
    @job_group(name="repairing")
    def _repair(self, models: List[Any], continous_columns: List[str], dirty_rows_df: DataFrame, error_cells_df: DataFrame,
                compute_repair_candidate_prob: bool, maximal_likelihood_repair: bool) -> Tuple[DataFrame, Dict[str, Any]]:
        """The body of the reference's grouped-map UDF `repair` (model.py:1095-1135) on the whole dirty frame."""
        _logger.info("[Repairing Phase] Computing %d repair updates in %d rows..." % (len(error_cells_df), len(dirty_rows_df)))
        pdf = dirty_rows_df.copy()
        need_pmf = compute_repair_candidate_prob or maximal_likelihood_repair
        pmfs: Dict[str, Any] = {}
        for y, (model, features, _) in models:
            X = pdf[features]
            isnull = pdf[y].isna().to_numpy()
            if need_pmf and y not in continous_columns:
                predicted = model.predict_proba(X)
                # the reference stores a JSON pmf in the cell; later models then see an unknown category,
                # i.e. the cell stays "missing" for them -- so the column is left NULL here
                pmfs[y] = (list(np.asarray(model.classes_).tolist()), predicted, isnull)
            else:
                predicted = np.asarray(model.predict(X), dtype=object if y not in continous_columns else None)
                if y in continous_columns and is_integral_column(dirty_rows_df[y]):
                    predicted = np.round(predicted.astype(np.float64))
                col = pdf[y].astype(object) if y not in continous_columns else pdf[y].astype("float64")
                col = col.where(~isnull, pd.Series(predicted, index=col.index))
                if y in continous_columns and is_integral_column(dirty_rows_df[y]):
                    col = col.astype("Int64")
                pdf[y] = col
        return pdf, pmfs

    # ------------------------------------------------------------------ result shaping
    def _flatten_join(self, repaired_rows_df: DataFrame, error_cells_df: DataFrame) -> DataFrame:
        rid = self._row_id
        pos = pd.Series(np.arange(len(repaired_rows_df)), index=repaired_rows_df[rid].to_numpy())
        rp = pos.reindex(error_cells_df[rid].to_numpy()).to_numpy(np.int64)
        vals = np.empty(len(error_cells_df), object)
        for a, idx in error_cells_df.groupby("attribute").indices.items():
            v = repaired_rows_df[a].to_numpy(dtype=object)[rp[idx]]
            vals[idx] = [None if pd.isna(x) else _to_str(x) for x in v]
        return error_cells_df.assign(repaired=vals)

    def _compute_repair_pmf(self, pmfs: Dict[str, Any], repaired_rows_df: DataFrame, dirty_rows_df: DataFrame,
                            error_cells_df: DataFrame, continous_columns: List[str]) -> DataFrame:
        rid = self._row_id
        thres = float(self._get_option_value(*self._opt_prob_threshold))
        top_k = int(self._get_option_value(*self._opt_prob_top_k))
        weight = float(self._get_option_value(*self._opt_cost_weight))
        pos = pd.Series(np.arange(len(dirty_rows_df)), index=dirty_rows_df[rid].to_numpy())
        rows = []
        for r in error_cells_df.itertuples(index=False):
            rowid, attr, cur = r[0], r[1], r[2]
            i = int(pos[rowid])
            if attr in continous_columns:
                v = repaired_rows_df[attr].iloc[i]
                rows.append((rowid, attr, {"value": cur, "prob": 0.0}, [{"class": None if pd.isna(v) else _to_str(v), "prob": 1.0}]))
                continue
            classes, predicted, _ = pmfs[attr]
            probs = predicted[i]
            if probs is None:
                cls, pr = [], []
            else:
                # a PoorModel(None) target (every label NULL) has the single class NULL, not the string 'None'
                cls = [None if c is None else _to_str(c) for c in classes]
                pr = [float(p) for p in np.asarray(probs)[:len(classes)]]
            if self.cf is not None and pr:
                # _compute_weighted_probs (model.py:1145-1165): costs weigh the probs of the cost function's targets only,
                # but the renormalisation runs for every attribute once a cost function is set
                if not self.cf.targets or attr in self.cf.targets:
                    costs = [self.cf.compute(cur, c) for c in cls] if cur else None
                    if costs is not None:
                        pr = [p * (1.0 / (1.0 + weight * c)) if c is not None else p for p, c in zip(pr, costs)]
                norm = sum(pr)
                pr = [p / norm for p in pr] if norm > 0 else pr
            cur_prob = pr[cls.index(cur)] if cur in cls else 0.0
            order = sorted(range(len(cls)), key=lambda j: -pr[j])   # stable: ties keep class order
            pmf = [{"class": cls[j], "prob": pr[j]} for j in order if pr[j] > thres][:top_k]
            rows.append((rowid, attr, {"value": cur, "prob": cur_prob}, pmf))
        return pd.DataFrame(rows, columns=[rid, "attribute", "current_value", "pmf"])

    def _compute_score(self, pmf_df: DataFrame) -> DataFrame:
        assert self.cf is not None
        out = []
        for r in pmf_df.itertuples(index=False):
            cur, pmf = r.current_value, r.pmf
            rep = pmf[0] if pmf else {"class": None, "prob": 0.0}
            base = cur["value"] if cur["value"] is not None else rep["class"]
            cost = self.cf.compute(base, rep["class"])
            p_cur = cur["prob"] if cur["prob"] > 0.0 else 1e-6
            score = float(np.log(rep["prob"] / p_cur)) * (1.0 / (1.0 + (cost if cost is not None else 256.0))) if rep["prob"] > 0 else float("-inf")
            out.append((r[0], r.attribute, cur["value"], rep["class"], score))
        return pd.DataFrame(out, columns=[self._row_id, "attribute", "current_value", "repaired", "score"])

    def _maximal_likelihood_repair(self, score_df: DataFrame, error_cells_df: DataFrame) -> DataFrame:
        assert self.repair_delta is not None
        n = len(error_cells_df)
        percent = min(1.0, 1.0 - self.repair_delta / n)
        thres = float(np.percentile(score_df["score"].to_numpy(np.float64), percent * 100.0)) if n else 0.0
        top = score_df[score_df["score"] >= thres].drop(columns=["score"])
        _logger.info("[Repairing Phase] %d repair updates (delta=%d) selected among %d candidates" % (len(top), self.repair_delta, n))
        return top

    def _repair_attrs(self, updates: DataFrame, base: DataFrame) -> DataFrame:
        """RepairMiscApi.repairAttrsFrom (RepairMiscApi.scala:184-247): apply (row_id, attribute, repaired) updates to a table.
        `repaired` arrives as strings; continuous attributes are cast back (`CAST(.. AS DOUBLE)` accepts Java's `3.1D`
        spelling), integral ones through `round` (half-up, as Spark rounds doubles) first (lines 224-229)."""
        out = base.copy()
        pos = pd.Series(np.arange(len(out)), index=out[self._row_id].to_numpy())
        for a, grp in updates.groupby("attribute"):
            if a not in out.columns:
                continue
            rows = pos.reindex(grp[self._row_id].to_numpy()).to_numpy(np.float64)
            keep = ~np.isnan(rows)                            # updates of unknown rows vanish in the LEFT OUTER JOIN
            rows = rows[keep].astype(np.int64)
            vals = grp["repaired"].to_numpy(dtype=object)[keep]
            if is_numeric_column(base[a]):
                num = np.array([_to_double(v) for v in vals], np.float64)
                if is_integral_column(base[a]):
                    num = np.where(np.isnan(num), np.nan, np.sign(num) * np.floor(np.abs(num) + 0.5))
                    col = out[a].astype("Int64")
                    col.iloc[rows] = pd.array([pd.NA if np.isnan(v) else int(v) for v in num], dtype="Int64")
                else:
                    col = out[a].astype("float64")
                    col.iloc[rows] = num
            else:
                col = out[a].astype(object)
                col.iloc[rows] = vals
            out[a] = col
        return out

    def _repair_by_nearest_values(self, repair_base_df: DataFrame, error_cells_df: DataFrame, target_columns: List[str]) -> Tuple[DataFrame, DataFrame]:
        assert self.cf is not None
        targets = [c for c in target_columns if c in self.cf.targets] if self.cf.targets else target_columns
        merge_thres = float(self._get_option_value(*self._opt_merge_threshold))
        rep = np.empty(len(error_cells_df), object)
        dom = {c: list(pd.unique(repair_base_df[c].dropna())) for c in targets}
        for i, r in enumerate(error_cells_df.itertuples(index=False)):
            rep[i] = None
            r = type('R', (), dict(attribute=r[1], current_value=r[2]))
            if r.attribute in dom and r.current_value:
                costs = sorted(((self.cf.compute(r.current_value, _to_str(v)), _to_str(v)) for v in dom[r.attribute]),
                               key=lambda t: (t[0] is None, t[0]))
                costs = [c for c in costs if c[0] is not None]
                if costs and costs[0][0] <= merge_thres and (len(costs) == 1 or costs[0][0] < costs[1][0]):
                    rep[i] = costs[0][1]
        done = pd.notna(pd.Series(rep)).to_numpy()
        repaired = error_cells_df[done].assign(repaired=rep[done])
        return error_cells_df[~done], repaired

    # ------------------------------------------------------------------ resident (device) path
    def _resident_engine(self) -> Any:
        """The engine of the HBM-resident path, or None: no HIP device, the estimator backend was swapped (CPU tests), or
        REPAIR_RESIDENT=0.  Tests inject one through `_engine_override`."""
        import os
        hook = getattr(self, "_engine_override", None)
        if hook is not None:
            return hook
        if os.environ.get("REPAIR_RESIDENT", "1") == "0":
            return None
        from repair import _native, gbm
        if gbm.get_backend() is not _native:
            return None
        try:
            if _native.device_count() < 1:
                return None
            from repair.engine import HipEngine
            return HipEngine(int(self._get_option_value(*self._opt_gpu_device_id)))
        except Exception:  # noqa: BLE001 - any doubt: the value-space path
            return None

    def _resident_plan(self, input_df: DataFrame, target_columns: List[str], continous_columns: List[str], domain_stats: Dict[str, int],
                       compute_repair_candidate_prob: bool, maximal_likelihood_repair: bool) -> Optional[Dict[str, Any]]:
        """Decides whether this run can take the HBM-resident pipeline (repair.pipeline) and with which parameters.

        It can when everything between error detection and the result frame is the per-attribute model loop itself:
        no rule-based repairs, no functional-dependency rule models, no rebalancing, no cost function, plain repair output
        (cells or repaired data), no feature selection, and every discrete target has at least two classes.  The
        hyper-parameter search (`model.hp.max_evals` > 1) runs on the resident tables too (pipeline.search_on_table).
        Otherwise the value-space path (pandas + one estimator per attribute) below handles the run."""
        if compute_repair_candidate_prob or maximal_likelihood_repair or self.repair_by_rules or self.cf is not None:
            return None
        if self.training_data_rebalancing_enabled:
            return None
        from repair.train import (_opt_learning_rate, _opt_max_bin, _opt_max_depth, _opt_max_evals, _opt_min_split_gain,
                                  _opt_n_estimators, _opt_reg_alpha, _opt_boosting_type, _opt_class_weight)
        g = lambda o: get_option_value(self.opts, *o)  # noqa: E731
        if g(_opt_boosting_type) != "gbdt" or g(_opt_class_weight) != "balanced":
            return None
        features = len(input_df.columns) - 2
        if int(self._get_option_value(*self._opt_max_training_column_num)) < features:
            return None
        if self._repair_by_functional_deps_enabled and self._get_functional_deps(target_columns):
            return None
        for y in target_columns:
            if y not in continous_columns and int(domain_stats.get(y, 0)) < 2:
                return None
        max_depth = int(g(_opt_max_depth))
        if not 1 <= max_depth:
            return None
        engine = self._resident_engine()
        if engine is None:
            return None
        params = dict(n_estimators=int(g(_opt_n_estimators)), learning_rate=float(g(_opt_learning_rate)), max_depth=max_depth,
                      max_bin=int(g(_opt_max_bin)), lambda_l1=float(g(_opt_reg_alpha)), min_gain_to_split=float(g(_opt_min_split_gain)),
                      num_leaves=31, min_data_in_leaf=20, min_sum_hessian_in_leaf=1e-3, lambda_l2=0.0, bagging_fraction=1.0, bagging_freq=0,
                      feature_fraction=1.0, seed=42)
        return dict(engine=engine, params=params, search=int(g(_opt_max_evals)) > 1)

    def _device_detection_plan(self, input_df: DataFrame, continous_columns: List[str], compute_repair_candidate_prob: bool,
                               maximal_likelihood_repair: bool) -> Optional[Dict[str, Any]]:
        """Can error DETECTION run on the resident table as well (reference python/repair/errors.py:545-582)?  Yes when the error
        cells come from NULL / denial-constraint detectors only (no own target lists; every constraint in the form the device
        detector handles, `pipeline.constraint_to_columns`) and the rest of the run qualifies for the resident pipeline: the
        frame is then encoded once and `pipeline.repair_frame` detects, NULLs, trains and repairs without an `error_cells_df` ever
        being built in pandas."""
        from repair.pipeline import constraint_to_columns
        if self.error_cells is not None or not self.error_detectors:
            return None
        cols = [c for c in input_df.columns if c != self._row_id]
        cons: List[Tuple[List[str], str]] = []
        has_null = False
        for d in self.error_detectors:
            if getattr(d, "targets", None):
                return None
            if type(d) is NullErrorDetector:
                has_null = True
            elif type(d) is ConstraintErrorDetector:
                stmts = load_constraints(d.constraint_path, d.constraints)
                try:
                    plist = parse_and_verify_constraints(stmts, list(input_df.columns)) if stmts else []
                except Exception:  # noqa: BLE001 - the pandas detector reports malformed constraints
                    return None
                for preds in plist:
                    cc = constraint_to_columns(preds, cols)
                    if cc is None:
                        return None
                    cons.append(([cols[i] for i in cc[0]], cols[cc[1]]))
            else:
                return None
        from repair.utils import column_nunique
        domain_stats = {c: column_nunique(input_df, c) for c in cols}
        discretized = [c for c in cols if c in continous_columns or 1 < domain_stats[c] <= self.discrete_thres]
        cands = [c for c in (self.targets if self.targets else cols) if c in discretized]
        if not cands:
            return None
        # Error cells OUTSIDE the discretizable candidates must be seen by the pandas detectors: with noisy cells there and none in a
        # candidate the reference raises "At least one valid discretizable feature ..." (model.py `_run`), which the device path --
        # detecting in the candidates only -- would report as "already clean" (ADVICE r3).  Cheap host checks, before anything is uploaded.
        others = [c for c in (self.targets if self.targets else cols) if c in cols and c not in cands]
        if others and has_null and bool(input_df[others].isna().to_numpy().any()):
            return None
        if others and any(a in others for xs, y in cons for a in list(xs) + [y]):
            return None
        plan = self._resident_plan(input_df, cands, continous_columns, domain_stats, compute_repair_candidate_prob, maximal_likelihood_repair)
        if plan is None:
            return None
        plan.update(candidates=cands, constraints=cons, detect_nulls=has_null)
        return plan

    def _run_resident_detect(self, plan: Dict[str, Any], input_df: DataFrame, continous_columns: List[str], repair_data: bool) -> DataFrame:
        """`_run` with detection, NULLing, training and repair on the resident table (`_device_detection_plan`)."""
        from repair.pipeline import NotResidentEligible, repair_frame
        from repair.errors import _to_sql_string
        rid = self._row_id
        cands = plan["candidates"]
        if repair_data and not plan["detect_nulls"]:
            raise NotResidentEligible("repair_data without a NULL detector: NULL target cells of dirty rows would stay NULL")
        max_rows = int(self._get_option_value(*self._opt_max_training_row_num))

        def sample(_attr: str, rows: np.ndarray) -> Optional[np.ndarray]:
            if len(rows) <= max_rows:
                return None
            _logger.info("To reduce training data, extracts %s%% samples from %d rows" % (100.0 * max_rows / len(rows), len(rows)))
            return rows[np.random.RandomState(42).choice(len(rows), max_rows, replace=False)]

        _logger.info("[Error Detection + Repair Model Training Phase] on the HBM-resident table, candidate attributes %s" % to_list_str(cands))
        frame, info = repair_frame(plan["engine"], input_df, rid, targets=cands, base_params=plan["params"], constraints=plan["constraints"],
                                   detect_nulls=plan["detect_nulls"], continuous_columns=[c for c in continous_columns if c in cands],
                                   train_rows=sample, want_details=True, search_opts=dict(self.opts) if plan.get("search") else None,
                                   only_noisy_targets=True)
        self._last_resident_info = info
        self._last_detection_on_device = True
        if len(frame) == 0:
            _logger.info("Any error cell not found, so the input data is already clean")
            return input_df if repair_data else pd.DataFrame({rid: pd.Series([], dtype=input_df[rid].dtype), "attribute": pd.Series([], dtype=object),
                                                              "current_value": pd.Series([], dtype=object), "repaired": pd.Series([], dtype=object)})
        cur = [None if v is None or (isinstance(v, float) and np.isnan(v)) else _to_sql_string(v) for v in frame["current_value"].tolist()]
        rep = [None if v is None or (isinstance(v, float) and np.isnan(v)) else _to_str(v) for v in frame["repaired"].tolist()]
        cand = pd.DataFrame({rid: frame[rid].to_numpy(), "attribute": frame["attribute"].to_numpy(), "current_value": np.asarray(cur, object),
                             "repaired": np.asarray(rep, object)})
        # integral attributes: CAST(int AS STRING) of the current value is '2', never '2.0' (ErrorModel.detect keeps nullable ints integral)
        for a in set(cand["attribute"]):
            if is_integral_column(input_df[a]):
                m = (cand["attribute"] == a).to_numpy() & cand["current_value"].notna().to_numpy()
                cand.loc[m, "current_value"] = [str(int(float(v))) for v in cand.loc[m, "current_value"]]
        if repair_data:
            base = self._prepare_repair_base_cells(input_df, cand, sorted(set(cand["attribute"])))
            is_dirty = base[rid].isin(set(cand[rid].tolist())).to_numpy()
            dirty = self._repair_attrs(cand[[rid, "attribute", "repaired"]], base[is_dirty].reset_index(drop=True))
            return pd.concat([base[~is_dirty], dirty], ignore_index=True)
        keep = cand["repaired"].isna() | ~((cand["current_value"] == cand["repaired"]) | (cand["current_value"].isna() & cand["repaired"].isna()))
        return cand[keep.to_numpy()].reset_index(drop=True)

    def _run_resident(self, plan: Dict[str, Any], input_df: DataFrame, error_cells_df: DataFrame, target_columns: List[str],
                      continous_columns: List[str], repair_data: bool) -> DataFrame:
        """Steps 2 and 3 of `_run` on the device: the table is encoded once (dictionary indices -> codes, on the device), error
        cells are NULLed, the dirty rows split off, one model per target attribute trained and the chained repair run without
        the table leaving HBM (repair.pipeline.repair_frame); the host only shapes the (small) list of repaired cells."""
        from repair.pipeline import NotResidentEligible, repair_frame
        rid = self._row_id
        max_rows = int(self._get_option_value(*self._opt_max_training_row_num))
        if repair_data:
            # The repair UDF fills EVERY NULL target cell of a dirty row (model.py:1128,1133), the cell list below only the error
            # cells: with setErrorCells / non-NULL detectors a dirty row may hold further NULLs -- those runs keep the value-space path.
            dirty = input_df[rid].isin(set(error_cells_df[rid].tolist())).to_numpy()
            nulls_in_dirty = int(input_df.loc[dirty, target_columns].isna().to_numpy().sum())
            null_error_cells = int(error_cells_df["current_value"].isna().to_numpy().sum())
            if nulls_in_dirty > null_error_cells:
                raise NotResidentEligible("%d NULL target cells of dirty rows are not error cells" % (nulls_in_dirty - null_error_cells))

        def sample(_attr: str, rows: np.ndarray) -> Optional[np.ndarray]:
            # `_sample_training_data_from`: the same seeded sample as DataFrame.sample(n, random_state=42) on the attribute's non-NULL rows
            if len(rows) <= max_rows:
                return None
            _logger.info("To reduce training data, extracts %s%% samples from %d rows" % (100.0 * max_rows / len(rows), len(rows)))
            return rows[np.random.RandomState(42).choice(len(rows), max_rows, replace=False)]

        _logger.info("[Repair Model Training Phase] Building %d models on the HBM-resident table to repair the cells in %s" % (
            len(target_columns), to_list_str(target_columns)))
        frame, info = repair_frame(plan["engine"], input_df, rid, targets=target_columns, base_params=plan["params"],
                                   error_cells=error_cells_df[[rid, "attribute"]], detect_nulls=False,
                                   continuous_columns=[c for c in continous_columns if c in target_columns], train_rows=sample,
                                   want_details=True, search_opts=dict(self.opts) if plan.get("search") else None)
        self._last_resident_info = info
        rep = pd.Series(frame["repaired"].to_numpy(dtype=object),
                        index=pd.MultiIndex.from_arrays([frame[rid].to_numpy(), frame["attribute"].to_numpy()]))
        key = pd.MultiIndex.from_arrays([error_cells_df[rid].to_numpy(), error_cells_df["attribute"].to_numpy()])
        values = rep.reindex(key).to_numpy(dtype=object)
        if repair_data:
            base = self._prepare_repair_base_cells(input_df, error_cells_df, target_columns)
            is_dirty = base[rid].isin(set(error_cells_df[rid].tolist())).to_numpy()
            upd = error_cells_df[[rid, "attribute"]].assign(repaired=[None if v is None or (isinstance(v, float) and np.isnan(v)) else _to_str(v) for v in values])
            dirty = self._repair_attrs(upd, base[is_dirty].reset_index(drop=True))
            return pd.concat([base[~is_dirty], dirty], ignore_index=True)
        cand = error_cells_df.assign(repaired=[None if v is None or (isinstance(v, float) and np.isnan(v)) else _to_str(v) for v in values])
        keep = cand["repaired"].isna() | ~((cand["current_value"] == cand["repaired"]) | (cand["current_value"].isna() & cand["repaired"].isna()))
        return cand[keep.to_numpy()].reset_index(drop=True)

    # ------------------------------------------------------------------ pipeline
    @elapsed_time  # type: ignore
    def _run(self, input_df: DataFrame, continous_columns: List[str], detect_errors_only: bool,
             compute_repair_candidate_prob: bool, compute_repair_prob: bool, compute_repair_score: bool,
             repair_data: bool, maximal_likelihood_repair: bool) -> DataFrame:
        rid = self._row_id
        self._last_detection_on_device = False
        # 0. Everything on the HBM-resident table, detection included, when the detectors and the run allow it
        if not detect_errors_only:
            dplan = self._device_detection_plan(input_df, continous_columns, compute_repair_candidate_prob, maximal_likelihood_repair)
            if dplan is not None:
                from repair.pipeline import NotResidentEligible
                try:
                    return self._run_resident_detect(dplan, input_df, continous_columns, repair_data)
                except NotResidentEligible as e:
                    self._last_resident_info = None
                    self._last_detection_on_device = False
                    _logger.info("device-side detection not taken: %s" % e)
        # 1. Error Detection Phase
        _logger.info("[Error Detection Phase] Detecting errors in a table... ")
        error_cells_df, target_columns, pairwise_attr_stats, domain_stats = self._detect_errors(input_df, continous_columns)
        if detect_errors_only:
            return error_cells_df
        if len(error_cells_df) == 0:
            _logger.info("Any error cell not found, so the input data is already clean")
            return input_df if repair_data else error_cells_df.assign(repaired=pd.Series([], dtype=object))
        if len(target_columns) == 0:
            raise ValueError("At least one valid discretizable feature is needed to repair error cells, but no such feature found")
        error_cells_df = error_cells_df[error_cells_df["attribute"].isin(target_columns)].reset_index(drop=True)

        # 2. + 3. on the HBM-resident table when the run is the plain per-attribute model loop (the hot path of this engine)
        plan = self._resident_plan(input_df, target_columns, continous_columns, domain_stats, compute_repair_candidate_prob,
                                   maximal_likelihood_repair)
        if plan is not None:
            from repair.pipeline import NotResidentEligible
            try:
                return self._run_resident(plan, input_df, error_cells_df, target_columns, continous_columns, repair_data)
            except NotResidentEligible as e:
                # only known once the error cells are NULLed: a target left with a single class / no value (the reference's PoorModel
                # short-cut, model.py:1008-1017), or NULL target cells in dirty rows that are not error cells under repair_data.
                # (Unseen categories of a dirty row need no fallback: the table marks categorical columns, so a model treats a
                # category its training rows do not show as missing, like the reference's per-model encoders.)
                self._last_resident_info = None
                _logger.info("resident path not taken: %s" % e)

        # 2. Repair Model Training Phase
        repair_base_df = self._prepare_repair_base_cells(input_df, error_cells_df, target_columns)
        repaired_by_rules_df = None
        if self.repair_by_rules:
            repaired_by_rules_df = error_cells_df.iloc[0:0].assign(repaired=pd.Series([], dtype=object))
            if self._repair_by_regex_enabled and any(isinstance(d, RegExErrorDetector) for d in self.error_detectors):
                _logger.warning("regex-structure repair (RegexStructureRepair.scala) is outside the accelerated path and not available")
            if self._repair_by_nearest_values_enabled:
                error_cells_df, by_nv = self._repair_by_nearest_values(repair_base_df, error_cells_df, target_columns)
                repaired_by_rules_df = pd.concat([repaired_by_rules_df, by_nv], ignore_index=True)
            repair_base_df = self._repair_attrs(repaired_by_rules_df, repair_base_df)
        dirty_ids = set(error_cells_df[rid].tolist())
        is_dirty = repair_base_df[rid].isin(dirty_ids).to_numpy()
        clean_rows_df, dirty_rows_df = repair_base_df[~is_dirty], repair_base_df[is_dirty].reset_index(drop=True)
        models = self._build_repair_models(repair_base_df, target_columns, continous_columns, domain_stats, pairwise_attr_stats)

        # 3. Repair Phase
        repaired_rows_df, pmfs = self._repair(models, continous_columns, dirty_rows_df, error_cells_df,
                                              compute_repair_candidate_prob, maximal_likelihood_repair)
        if compute_repair_candidate_prob and not maximal_likelihood_repair:
            assert not self._repair_by_nearest_values_enabled, "repairing data by nearest values not supported in this path"
            pmf_df = self._compute_repair_pmf(pmfs, repaired_rows_df, dirty_rows_df, error_cells_df, continous_columns)
            pmf_df = pmf_df.assign(current_value=[c["value"] for c in pmf_df["current_value"]])
            if compute_repair_prob:
                return pd.DataFrame({rid: pmf_df[rid], "attribute": pmf_df["attribute"], "current_value": pmf_df["current_value"],
                                     "repaired": [p[0]["class"] if p else None for p in pmf_df["pmf"]],
                                     "prob": [p[0]["prob"] if p else None for p in pmf_df["pmf"]]})
            return pmf_df
        if maximal_likelihood_repair:
            assert len(continous_columns) == 0
            pmf_df = self._compute_repair_pmf(pmfs, repaired_rows_df, dirty_rows_df, error_cells_df, [])
            score_df = self._compute_score(pmf_df)
            if compute_repair_score:
                return score_df
            top = self._maximal_likelihood_repair(score_df, error_cells_df)
            if not repair_data:
                return top
            repaired_rows_df = self._repair_attrs(top, dirty_rows_df)
        if repair_data:
            clean_df = pd.concat([clean_rows_df, repaired_rows_df], ignore_index=True)
            assert len(clean_df) == len(input_df)
            return clean_df
        cand = self._flatten_join(repaired_rows_df, error_cells_df)
        keep = cand["repaired"].isna() | ~((cand["current_value"] == cand["repaired"]) | (cand["current_value"].isna() & cand["repaired"].isna()))
        cand = cand[keep.to_numpy()].reset_index(drop=True)
        if self.repair_by_rules and repaired_by_rules_df is not None:
            cand = pd.concat([cand, repaired_by_rules_df], ignore_index=True)
        return cand

    def run(self, detect_errors_only: bool = False, compute_repair_candidate_prob: bool = False,
            compute_repair_prob: bool = False, compute_repair_score: bool = False, repair_data: bool = False,
            maximal_likelihood_repair: bool = False) -> DataFrame:
        """Starts processing to detect error cells in given input data and build a statistical model to
        repair them (reference model.py:1421-1537: same flags, same exclusivity rules, same messages)."""
        if self.input is None or self.row_id is None:
            raise ValueError("`setInput` and `setRowId` should be called before repairing")
        if maximal_likelihood_repair and self.repair_delta is None:
            raise ValueError("`setRepairDelta` should be called when enabling maximal likelihood repairing")
        if maximal_likelihood_repair and self.cf is None:
            raise ValueError("`setUpdateCostFunction` should be called when enabling maximal likelihood repairing")
        if maximal_likelihood_repair and len(self.cf.targets) > 0:  # type: ignore
            raise ValueError("`UpdateCostFunction.targets` cannot be used when enabling maximal likelihood repairing")
        flags = [("detect_errors_only", detect_errors_only), ("compute_repair_candidate_prob", compute_repair_candidate_prob),
                 ("compute_repair_prob", compute_repair_prob), ("compute_repair_score", compute_repair_score), ("repair_data", repair_data)]
        selected = [n for n, v in flags if v]
        if len(selected) > 1:
            raise ValueError("%s cannot be set to true simultaneously" % to_list_str(selected, sep="/", quote=True))
        if self._repair_by_nearest_values_enabled and (maximal_likelihood_repair or compute_repair_candidate_prob or compute_repair_prob or compute_repair_score):
            raise ValueError("Cannot repair data by nearest values when enabling `maximal_likelihood_repair`, "
                             "`compute_repair_candidate_prob`, `compute_repair_prob`, or `compute_repair_score`")
        if compute_repair_prob or compute_repair_score:
            compute_repair_candidate_prob = True
        if compute_repair_score:
            maximal_likelihood_repair = True
        input_df, continous_columns = self._check_input_table()
        if maximal_likelihood_repair and len(continous_columns) != 0:
            raise ValueError("Cannot enable the maximal likelihood repair mode when continous attributes found")
        if self.targets and len(set(self.targets) & set(input_df.columns)) == 0:
            # the reference names the (qualified) input table here (model.py:1525-1526); a DataFrame input has no name
            raise ValueError("Target attributes not found in %s: %s" % (self._qualified_input_name(), to_list_str(self.targets)))
        self.opts.setdefault("model.gpu.device_id", self.opts.get("model.gpu.device_id", "0"))
        from repair.utils import column_code_cache
        with column_code_cache(input_df):     # NULL detection, domain statistics and the resident encoding share one hash pass per column
            df, elapsed = self._run(input_df, continous_columns, detect_errors_only, compute_repair_candidate_prob,
                                    compute_repair_prob, compute_repair_score, repair_data, maximal_likelihood_repair)
        _logger.info("!!!Total Processing time is %s(s)!!!" % elapsed)
        return df


# RepairBase.scala:41-47: the column types the reference accepts, in the order its message prints them
_SUPPORTED_TYPES = ["tinyint", "float", "smallint", "string", "double", "int", "bigint"]


def _sql_type_name(s: Any) -> str:
    """Spark SQL name of a pandas column's type (only what `checkInputTable` needs to tell apart)."""
    import datetime
    k = s.dtype.kind
    if k == "M":
        return "timestamp"
    if k == "m":
        return "interval"
    if k == "b" or str(s.dtype) == "boolean":
        return "boolean"
    if k in "iu" or str(s.dtype).startswith(("Int", "UInt")):
        bits = "".join(ch for ch in str(s.dtype) if ch.isdigit())
        return {"8": "tinyint", "16": "smallint", "32": "int"}.get(bits, "bigint")
    if k == "f" or str(s.dtype).startswith("Float"):
        return "float" if "32" in str(s.dtype) else "double"
    if isinstance(s.dtype, pd.CategoricalDtype):         # a dictionary-encoded column has the type of its dictionary
        return _sql_type_name(s.cat.categories.to_series())
    kind = pd.api.types.infer_dtype(s, skipna=True)      # one C pass; only the unusual answers need the element-wise look below
    if kind in ("string", "empty"):
        return "string"
    vals = s.dropna()
    if len(vals) and all(isinstance(v, datetime.datetime) for v in vals):
        return "timestamp"
    if len(vals) and all(isinstance(v, datetime.date) for v in vals):
        return "date"
    if len(vals) and all(isinstance(v, (bool, np.bool_)) for v in vals):
        return "boolean"
    return "string"


def _to_double(v: Any) -> float:
    """Spark's CAST(string AS DOUBLE): Java `Double.parseDouble` after trimming, which also takes a d/D/f/F suffix; anything
    else is NULL."""
    if v is None or (isinstance(v, float) and np.isnan(v)):
        return float("nan")
    if isinstance(v, (int, float, np.integer, np.floating)):
        return float(v)
    t = str(v).strip()
    if t[-1:] in "dDfF" and len(t) > 1:
        t = t[:-1]
    try:
        return float(t)
    except ValueError:
        return float("nan")


def _to_str(v: Any) -> str:
    if isinstance(v, (float, np.floating)):
        return repr(float(v))
    if isinstance(v, (np.integer,)):
        return str(int(v))
    return str(v)
