"""The ErrorDetector plugin API of the reference (python/repair/errors.py:37-82) and its detectors,
re-stated on pandas (the reference pushes them to Spark SQL in ErrorDetectorApi.scala:128-300).

Kept: class names, constructor arguments, `setUp(row_id, qualified_input_name, continous_cols,
targets)`, `_detect_impl()` -> DataFrame[row_id, attribute], `detect()`.  `qualified_input_name` may
be a registered table name or the DataFrame itself.  Error detection is NOT the accelerated hot
path; these run on the host.
"""
import re
from abc import ABCMeta, abstractmethod
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd

from repair import session
from repair.utils import get_option_value, setup_logger, to_list_str

_logger = setup_logger()


class ErrorDetector(metaclass=ABCMeta):

    def __init__(self, targets: List[str] = []) -> None:
        self.row_id: Optional[str] = None
        self.qualified_input_name: Any = None
        self.continous_cols: List[str] = []
        self.targets: List[str] = targets

    def setUp(self, row_id: str, qualified_input_name: Any, continous_cols: List[str], targets: List[str]) -> "ErrorDetector":
        self.row_id = row_id
        self.qualified_input_name = qualified_input_name
        self.continous_cols = continous_cols
        self._targets = [t for t in targets if t in set(self.targets)] if self.targets else list(targets)
        return self

    @abstractmethod
    def _detect_impl(self) -> pd.DataFrame:
        pass

    def _input(self) -> pd.DataFrame:
        return session.resolve(self.qualified_input_name)

    def _empty_dataframe(self) -> pd.DataFrame:
        df = self._input()
        return pd.DataFrame({str(self.row_id): pd.Series([], dtype=df[str(self.row_id)].dtype), "attribute": pd.Series([], dtype=object)})

    def _cells(self, mask: Any, attr: str) -> pd.DataFrame:
        ids = self._input()[str(self.row_id)].to_numpy()[np.asarray(mask, bool)]
        return pd.DataFrame({str(self.row_id): ids, "attribute": attr})

    def detect(self) -> pd.DataFrame:
        assert self.row_id is not None and self.qualified_input_name is not None
        out = self._detect_impl()
        assert isinstance(out, pd.DataFrame)
        return out


def _concat(frames: List[pd.DataFrame], empty: pd.DataFrame) -> pd.DataFrame:
    frames = [f for f in frames if len(f)]
    return pd.concat(frames, ignore_index=True) if frames else empty


class NullErrorDetector(ErrorDetector):
    """NULL cells of the target attributes (ErrorDetectorApi.scala:128-157)."""

    def __init__(self) -> None:
        ErrorDetector.__init__(self)

    def __str__(self) -> str:
        return "%s()" % self.__class__.__name__

    def _detect_impl(self) -> pd.DataFrame:
        df = self._input()
        from repair.utils import column_isna
        return _concat([self._cells(column_isna(df, c), c) for c in df.columns if c != self.row_id and c in self._targets],
                       self._empty_dataframe())


class RegExErrorDetector(ErrorDetector):
    """`CAST(attr AS STRING) NOT RLIKE regex OR attr IS NULL` (ErrorDetectorApi.scala:159-187)."""

    def __init__(self, attr: str, regex: str) -> None:
        ErrorDetector.__init__(self)
        self.attr = attr
        self.regex = regex

    def __str__(self) -> str:
        return '%s(pattern="%s")' % (self.__class__.__name__, self.regex)

    def _detect_regex(self, attr: str, regex: str) -> pd.DataFrame:
        df = self._input()
        if attr not in self._targets or attr not in df.columns or not regex or not regex.strip():
            return self._empty_dataframe()
        pat = re.compile(regex)
        s = df[attr]
        bad = s.isna() | ~s.astype(str).map(lambda v: pat.search(v) is not None)
        return self._cells(bad, attr)

    def _detect_impl(self) -> pd.DataFrame:
        return self._detect_regex(self.attr, self.regex)


class DomainValues(RegExErrorDetector):
    """Cells outside a value domain; `autofill` takes the values seen more than `min_count_thres` times."""

    def __init__(self, attr: str, values: List[str] = [], autofill: bool = False, min_count_thres: int = 12) -> None:
        ErrorDetector.__init__(self)
        self.attr = attr
        self.values = values if not autofill else []
        self.autofill = autofill
        self.min_count_thres = min_count_thres

    def __str__(self) -> str:
        return '%s(attr="%s",size=%d,autofill=%s,min_count_thres=%d)' % (
            self.__class__.__name__, self.attr, len(self.values), self.autofill, self.min_count_thres)

    def _detect_impl(self) -> pd.DataFrame:
        if self.attr in self.continous_cols:
            return self._empty_dataframe()
        values = self.values
        if self.autofill and self.attr in self._input().columns:
            vc = self._input()[self.attr].dropna().value_counts()
            freq = [str(v) for v in vc[vc > self.min_count_thres].index.tolist()]
            if freq:
                values = freq
        # the reference builds the alternation un-escaped and un-anchored (errors.py:127)
        return self._detect_regex(self.attr, "(%s)" % "|".join(values) if values else "$^")


# ---- denial constraints (HoloClean syntax; src/main/scala/.../python/DenialConstraints.scala:66-225)
_OPS = ("EQ", "IQ", "LT", "GT")
_IDENT = re.compile(r"^[a-zA-Z]+[a-zA-Z0-9]*$")


class Predicate:
    def __init__(self, op: str, left: str, right: Optional[str] = None, constant: Optional[str] = None) -> None:
        self.op, self.left, self.right, self.constant = op, left, right, constant

    @property
    def references(self) -> List[str]:
        return [self.left] + ([self.right] if self.right is not None and self.right != self.left else [])

    def __repr__(self) -> str:
        return "%s(%s,%s)" % (self.op, self.left, self.right if self.constant is None else self.constant)


def parse_constraint(c: str) -> List[Predicate]:
    """One statement -> predicates.  `t1&t2&EQ(t1.A,t2.A)&IQ(t1.B,t2.B)`, `t1&EQ(t1.A,"x")&...`, or `X->Y`."""
    parts = [p.strip() for p in c.split("&")]
    parts = [p for p in parts if p != ""] if len(parts) > 1 else parts

    def _alt() -> List[Predicate]:
        xy = [p.strip() for p in c.split("->") if p.strip()]
        if len(xy) == 2:
            return [Predicate("EQ", xy[0], xy[0]), Predicate("IQ", xy[1], xy[1])]
        if xy:
            raise ValueError("Failed to parse an input string: '%s'" % c)
        return []

    try:
        if len(parts) >= 2 and _IDENT.match(parts[0]) and _IDENT.match(parts[1]):
            t1, t2, body = parts[0], parts[1], parts[2:]
            if len(body) < 2:
                raise ValueError("At least two predicate candidates should be given, but %d candidates found: %s" % (len(body), c))
            pat = re.compile(r"^(%s)\s*\(\s*%s\.(.*)\s*,\s*%s\.(.*)\s*\)$" % ("|".join(_OPS), re.escape(t1), re.escape(t2)))
            preds = []
            for b in body:
                m = pat.match(b)
                if not m:
                    raise ValueError("Illegal predicates found: %s" % b)
                preds.append(Predicate(m.group(1), m.group(2).strip(), m.group(3).strip()))
            return preds
        if len(parts) >= 1 and _IDENT.match(parts[0]) and len(parts) > 1:
            t1, body = parts[0], parts[1:]
            if len(body) < 2:
                raise ValueError("At least two predicate candidates should be given, but %d candidates found: %s" % (len(body), c))
            pat = re.compile(r"^(%s)\s*\(\s*%s\.(.*)\s*,\s*(.*)\)$" % ("|".join(_OPS), re.escape(t1)))
            preds = []
            for b in body:
                m = pat.match(b)
                if not m:
                    raise ValueError("Illegal predicates found: %s" % b)
                preds.append(Predicate(m.group(1), m.group(2).strip(), None, m.group(3).strip()))
            return preds
        raise ValueError("Failed to parse an input string: '%s'" % c)
    except ValueError:
        return _alt()


def load_constraints(path: str, text: str) -> List[str]:
    stmts: List[str] = []
    if path and path.strip():
        try:
            with open(path[7:] if path.startswith("file://") else path) as f:
                stmts += [ln.rstrip("\n") for ln in f]
        except Exception:
            _logger.warning("Failed to load constrains from '%s'" % path)
    if text:
        stmts += [s.strip() for s in text.split(";") if s.strip()]
    return stmts


def parse_and_verify_constraints(stmts: List[str], table_attrs: List[str]) -> List[List[Predicate]]:
    out = []
    for c in stmts:
        try:
            ps = parse_constraint(c)
        except Exception:
            _logger.warning("Illegal constraint format found: %s" % c)
            continue
        if ps:
            out.append(ps)
    attrs = set(table_attrs)
    return [ps for ps in out if all(r in attrs for p in ps for r in p.references)]


def _key(s: pd.Series) -> np.ndarray:
    """NULL-safe equality keys (`<=>`): NULLs compare equal to each other."""
    return pd.factorize(s.astype(object).where(s.notna(), "\0__NULL__"), sort=False)[0]


def _violating_rows(df: pd.DataFrame, preds: List[Predicate]) -> np.ndarray:
    n = len(df)
    if all(p.constant is not None for p in preds):   # single-tuple constraint
        ok = np.ones(n, bool)
        for p in preds:
            const = p.constant.strip()
            const = const[1:-1] if len(const) >= 2 and const[0] == const[-1] and const[0] in "\"'" else const
            col = df[p.left]
            sv = col.astype(str).where(col.notna(), None)
            if p.op == "EQ":
                ok &= (sv == const).fillna(False).to_numpy(bool)
            elif p.op == "IQ":
                ok &= ~((sv == const).fillna(False).to_numpy(bool))
            else:
                num = pd.to_numeric(col, errors="coerce")
                cv = float(const)
                ok &= ((num < cv) if p.op == "LT" else (num > cv)).fillna(False).to_numpy(bool)
        return ok
    eq = [p for p in preds if p.op == "EQ"]
    iq = [p for p in preds if p.op == "IQ"]
    other = [p for p in preds if p.op in ("LT", "GT")]
    same_attr = all(p.left == p.right for p in preds)
    if same_attr and not other and len(iq) <= 1:
        # group by the EQ attributes; a row violates iff its group holds another IQ value
        if eq:
            gk = np.zeros(n, np.int64)
            for p in eq:
                k = _key(df[p.left]); gk = gk * (int(k.max()) + 2 if n else 1) + k
            gk = pd.factorize(gk)[0]
        else:
            gk = np.zeros(n, np.int64)
        if not iq:
            # EQ predicates only: the reference's EXISTS sub-query ranges over ALL rows, t1 itself included
            # (ErrorDetectorApi.scala:218-223), so every row satisfies it
            return np.ones(n, bool)
        vk = _key(df[iq[0].left])
        pair = pd.DataFrame({"g": gk, "v": vk}).drop_duplicates()
        nd = np.bincount(pair["g"].to_numpy(), minlength=int(gk.max()) + 1 if n else 0)
        return nd[gk] > 1
    # general fallback: O(n^2) in blocks (small tables only)
    if n > 20000:
        raise ValueError("constraint %s needs the quadratic fallback; table too large" % preds)
    cols: Dict[str, np.ndarray] = {}
    for p in preds:
        for a in p.references:
            if a not in cols:
                cols[a] = _key(df[a]) if p.op in ("EQ", "IQ") else pd.to_numeric(df[a], errors="coerce").to_numpy(np.float64)
    viol = np.zeros(n, bool)
    for i in range(n):
        m = np.ones(n, bool)
        for p in preds:
            l, r = cols[p.left][i], cols[p.right]
            if p.op == "EQ":
                m &= (r == l)
            elif p.op == "IQ":
                m &= (r != l)
            elif p.op == "LT":
                m &= (l < r)
            else:
                m &= (l > r)
        viol[i] = m.any()
    return viol


class ConstraintErrorDetector(ErrorDetector):
    """Cells of rows that violate a denial constraint (ErrorDetectorApi.scala:189-244)."""

    def __init__(self, constraint_path: str = "", constraints: str = "", targets: List[str] = []) -> None:
        ErrorDetector.__init__(self, targets)
        if not constraint_path and not constraints:
            raise ValueError("At least one of `constraint_path` or `constraints` should be specified")
        self.constraint_path = constraint_path
        self.constraints = constraints

    def __str__(self) -> str:
        params = []
        if self.constraint_path:
            params.append("constraint_path=%s" % self.constraint_path)
        if self.constraints:
            params.append("constraints=%s" % self.constraints)
        if self.targets:
            params.append("targets=%s" % ",".join(self.targets))
        return "%s(%s)" % (self.__class__.__name__, ",".join(params))

    def _detect_impl(self) -> pd.DataFrame:
        df = self._input()
        stmts = load_constraints(self.constraint_path, self.constraints)
        plist = parse_and_verify_constraints(stmts, list(df.columns)) if stmts else []
        frames = []
        for preds in plist:
            attrs = []
            for p in preds:
                for r in p.references:
                    if r in self._targets and r not in attrs:
                        attrs.append(r)
            if not attrs:
                continue
            viol = pd.Series(_violating_rows(df, preds))
            frames += [self._cells(viol, a) for a in attrs]
        out = _concat(frames, self._empty_dataframe())
        return out.drop_duplicates(ignore_index=True)


class GaussianOutlierErrorDetector(ErrorDetector):
    """Tukey fences on the continuous attributes (ErrorDetectorApi.scala:249-300)."""

    def __init__(self, approx_enabled: bool = False) -> None:
        ErrorDetector.__init__(self)
        self.approx_enabled = approx_enabled

    def __str__(self) -> str:
        return "%s(approx_enabled=%s)" % (self.__class__.__name__, self.approx_enabled)

    def _detect_impl(self) -> pd.DataFrame:
        df = self._input()
        frames = []
        for c in [c for c in self.continous_cols if c in self._targets]:
            v = pd.to_numeric(df[c], errors="coerce")
            if v.notna().sum() == 0:
                continue
            q1, q3 = np.percentile(v.dropna(), [25, 75])
            lo, hi = q1 - 1.5 * (q3 - q1), q3 + 1.5 * (q3 - q1)
            frames.append(self._cells((v < lo) | (v > hi), c))
        return _concat(frames, self._empty_dataframe())


class ScikitLearnBackedErrorDetector(ErrorDetector):
    """`fit_predict`-style outlier detectors on each continuous attribute (errors.py:193-299)."""

    def __init__(self, error_detector_cls: Callable[[], Any], parallel_mode_threshold: int = 10000,
                 num_parallelism: Optional[int] = None) -> None:
        ErrorDetector.__init__(self)
        if num_parallelism is not None and int(num_parallelism) <= 0:
            raise ValueError("`num_parallelism` must be positive, got %s" % num_parallelism)
        if not hasattr(error_detector_cls, "__call__"):
            raise ValueError("`error_detector_cls` should be callable")
        if not hasattr(error_detector_cls(), "fit_predict"):
            raise ValueError("An instance that `error_detector_cls` returns should have a `fit_predict` method")
        self.error_detector_cls = error_detector_cls
        self.parallel_mode_threshold = parallel_mode_threshold
        self.num_parallelism = num_parallelism

    def __str__(self) -> str:
        return "%s()" % self.__class__.__name__

    def _detect_impl(self) -> pd.DataFrame:
        df = self._input()
        cols = [c for c in self.continous_cols if c in self._targets] if self._targets else self.continous_cols
        frames = []
        for c in cols:
            v = pd.to_numeric(df[c], errors="coerce")
            med = float(np.median(v.dropna())) if v.notna().any() else 0.0
            pred = self.error_detector_cls().fit_predict(v.fillna(med).to_frame())
            frames.append(self._cells(pd.Series(np.asarray(pred) < 0), c))
        return _concat(frames, self._empty_dataframe())


class LOFOutlierErrorDetector(ScikitLearnBackedErrorDetector):

    def __init__(self, parallel_mode_threshold: int = 10000, num_parallelism: Optional[int] = None) -> None:
        from sklearn.neighbors import LocalOutlierFactor
        ScikitLearnBackedErrorDetector.__init__(self, lambda: LocalOutlierFactor(novelty=False), parallel_mode_threshold, num_parallelism)


class ErrorModel:
    """detect -> error cells with current values + the repairable target columns (errors.py:545-582).

    The reference additionally prunes "weak-labelled" cells with a naive-Bayes domain analysis in the
    JVM (RepairApi.computeDomainInErrorCells); that relational step is outside the accelerated path
    and is not re-stated: every detected cell is handed to the repair models (out of scope).
    """
    from collections import namedtuple
    _option = namedtuple("_option", "key default_value type_class validator err_msg")
    _opt_attr_freq_ratio_threshold = _option("error.attr_freq_ratio_threshold", 0.0, float, lambda v: 0.0 <= v <= 1.0, "`{}` should be in [0.0, 1.0]")
    _opt_pairwise_freq_ratio_threshold = _option("error.pairwise_freq_ratio_threshold", 0.05, float, lambda v: 0.0 <= v <= 1.0, "`{}` should be in [0.0, 1.0]")
    _opt_max_attrs_to_compute_pairwise_stats = _option("error.max_attrs_to_compute_pairwise_stats", 3, int, lambda v: v >= 2, "`{}` should be greater than 1")
    _opt_max_attrs_to_compute_domains = _option("error.max_attrs_to_compute_domains", 2, int, lambda v: v >= 2, "`{}` should be greater than 1")
    _opt_domain_threshold_alpha = _option("error.domain_threshold_alpha", 0.0, float, lambda v: 0.0 <= v < 1.0, "`{}` should be in [0.0, 1.0)")
    _opt_domain_threshold_beta = _option("error.domain_threshold_beta", 0.70, float, lambda v: 0.0 <= v < 1.0, "`{}` should be in [0.0, 1.0)")
    option_keys = set(o.key for o in (_opt_attr_freq_ratio_threshold, _opt_pairwise_freq_ratio_threshold,
                                      _opt_max_attrs_to_compute_pairwise_stats, _opt_max_attrs_to_compute_domains,
                                      _opt_domain_threshold_alpha, _opt_domain_threshold_beta))

    def __init__(self, row_id: str, targets: List[str], discrete_thres: int, error_detectors: List[ErrorDetector],
                 error_cells: Optional[pd.DataFrame], opts: Dict[str, str]) -> None:
        self.row_id = str(row_id)
        self.targets = targets
        self.discrete_thres = discrete_thres
        self.error_detectors = error_detectors
        self.error_cells = error_cells
        self.opts = opts

    def _default_detectors(self, df: pd.DataFrame) -> List[ErrorDetector]:
        dets: List[ErrorDetector] = [NullErrorDetector()]
        for c in (self.targets if self.targets else [c for c in df.columns if c != self.row_id]):
            dets.append(DomainValues(attr=c, autofill=True, min_count_thres=4))
        return dets

    def _target_attrs(self, columns: List[str]) -> List[str]:
        attrs = [c for c in columns if c != self.row_id]
        return [c for c in attrs if c in set(self.targets)] if self.targets else attrs

    def _checked_options(self) -> Dict[str, Any]:
        """The error-model options are cast and validated as the reference does when it reads them (errors.py:440-470,
        `_get_option_value`): a malformed value fails the run under testing and falls back to the default otherwise."""
        from repair.utils import get_option_value
        return {o.key: get_option_value(self.opts, *o) for o in (
            self._opt_attr_freq_ratio_threshold, self._opt_pairwise_freq_ratio_threshold, self._opt_max_attrs_to_compute_pairwise_stats,
            self._opt_max_attrs_to_compute_domains, self._opt_domain_threshold_alpha, self._opt_domain_threshold_beta)}

    def detect(self, input_df: pd.DataFrame, continous_columns: List[str]) -> Tuple[pd.DataFrame, List[str], Dict[str, Any], Dict[str, int]]:
        rid = self.row_id
        self._checked_options()
        if self.error_cells is not None:
            cells = self.error_cells[[rid, "attribute"]]
            if cells[rid].dtype != input_df[rid].dtype:   # e.g. 'tid STRING' cells against an int row id
                try:
                    cells = cells.assign(**{rid: cells[rid].astype(input_df[rid].dtype)})
                except Exception:
                    pass
            keep = self.targets if self.targets else list(input_df.columns)
            cells = cells[cells["attribute"].isin(keep)]
        else:
            dets = self.error_detectors or self._default_detectors(input_df)
            _logger.info("[Error Detection Phase] Used error detectors: %s" % to_list_str(dets))
            tattrs = self._target_attrs(list(input_df.columns))
            frames = [d.setUp(rid, input_df, continous_columns, tattrs).detect() for d in dets]
            cells = _concat(frames, pd.DataFrame({rid: pd.Series([], dtype=input_df[rid].dtype), "attribute": pd.Series([], dtype=object)}))
        cells = cells.drop_duplicates(ignore_index=True)
        cells = cells[cells["attribute"].isin([c for c in input_df.columns if c != rid])]
        # RepairApi.withCurrentValues is an INNER join on the row id (RepairApi.scala:91-101): cells of rows the input does not
        # hold are dropped silently
        cells = cells[cells[rid].isin(input_df[rid])].reset_index(drop=True)
        if len(cells) == 0:
            cells = cells.assign(current_value=pd.Series([], dtype=object))
            return cells, [], {}, {}
        # current values (RepairApi.withCurrentValues): CAST(value AS STRING)
        pos = pd.Series(np.arange(len(input_df)), index=input_df[rid].to_numpy())
        cur = np.empty(len(cells), object)
        rpos = pos.reindex(cells[rid].to_numpy()).to_numpy()
        for a, idx in cells.groupby("attribute").indices.items():
            col = input_df[a]
            # object dtype keeps nullable integers integral (CAST(int AS STRING) gives '2', never '2.0')
            vals = col.to_numpy(dtype=object)[rpos[idx].astype(np.int64)]
            null = np.asarray(pd.isna(vals), bool)               # one array pass; NULL cells (the usual error cell) need no formatting
            out = np.empty(len(vals), object)
            out[~null] = [_to_sql_string(v) for v in vals[~null]]
            cur[idx] = out
        cells = cells.assign(current_value=cur)
        noisy_columns = [c for c in input_df.columns if c in set(cells["attribute"])]
        from repair.utils import column_nunique
        domain_stats = {c: column_nunique(input_df, c) for c in input_df.columns if c != rid}
        # discretizable attributes (RepairApi.discretizeTable): continuous ones, or 1 < |domain| <= threshold
        discretized = [c for c in input_df.columns if c != rid and (c in continous_columns or 1 < domain_stats[c] <= self.discrete_thres)]
        target_columns = [c for c in noisy_columns if c in discretized]
        return cells, target_columns, {}, domain_stats


def _to_sql_string(v: Any) -> str:
    if isinstance(v, (float, np.floating)) and float(v).is_integer() and not isinstance(v, bool):
        return repr(float(v))
    return str(v)
