"""Label encoding: pandas columns -> the int32 column-major code matrix the C-ABI takes.

This replaces category_encoders' SumEncoder/OrdinalEncoder of the reference
(python/repair/model.py:701-729) as mandated by BASELINE.json's north_star ("Spark DataFrames
label-encoded into a pinned int32 column-major feature matrix").  Contract:
  * per column, the distinct non-NULL values sorted ascending get codes 0..n-1 (strings by code
    point, numbers numerically) -- deterministic, unlike first-appearance ordinal codes;
  * NULL / NaN / a value never seen when the dictionary was built -> -1 (LightGBM's NaN);
  * numeric columns keep their order (rank codes), so `code <= threshold` is `value <= threshold`;
    an unseen number maps to its nearest dictionary value (ties to the lower one), which is where
    a LightGBM midpoint threshold would send it.
"""
from typing import Dict, List, Optional

import numpy as np
import pandas as pd


class ColumnDict:
    def __init__(self, values: np.ndarray, numeric: bool) -> None:
        self.values = values          # sorted distinct values (object for strings, float64 for numbers)
        self.numeric = numeric
        self._index = None if numeric else {v: i for i, v in enumerate(values.tolist())}

    def __len__(self) -> int:
        return len(self.values)

    def encode(self, s: pd.Series) -> np.ndarray:
        out = np.full(len(s), -1, np.int32)
        notna = s.notna().to_numpy()
        if not notna.any() or len(self.values) == 0:
            return out
        if self.numeric:
            v = pd.to_numeric(s[notna], errors="coerce").to_numpy(np.float64)
            ok = ~np.isnan(v)
            pos = np.searchsorted(self.values, v, side="left")
            pos = np.clip(pos, 0, len(self.values) - 1)
            lo = np.clip(pos - 1, 0, len(self.values) - 1)
            # nearest dictionary value, ties to the lower one
            pick_lo = np.abs(v - self.values[lo]) <= np.abs(self.values[pos] - v)
            code = np.where(pick_lo, lo, pos).astype(np.int32)
            code[~ok] = -1
            out[notna] = code
        else:
            # hashed in C: position of each value among the (sorted) dictionary entries, -1 for NULL / unseen values
            out[:] = pd.Categorical(s.astype(object), categories=pd.Index(self.values, dtype=object)).codes
        return out

    def decode(self, codes: np.ndarray) -> np.ndarray:
        res = np.empty(len(codes), object)
        ok = (codes >= 0) & (codes < len(self.values))
        res[~ok] = None
        res[ok] = self.values[codes[ok]]
        return res


def is_numeric_column(s: pd.Series) -> bool:
    """Continuous types = byte/short/int/long/float/double (reference RepairBase.scala:41-44)."""
    return pd.api.types.is_numeric_dtype(s) and not pd.api.types.is_bool_dtype(s)


def is_integral_column(s: pd.Series) -> bool:
    return pd.api.types.is_integer_dtype(s)


class TableEncoder:
    """Dictionaries for every listed column of a DataFrame + the encoded matrix."""

    def __init__(self, df: pd.DataFrame, columns: List[str], numeric: Optional[List[str]] = None) -> None:
        self.columns = list(columns)
        self.dicts: Dict[str, ColumnDict] = {}
        for c in self.columns:
            s = df[c]
            num = (c in numeric) if numeric is not None else is_numeric_column(s)
            vals = s.dropna()
            if num:
                arr = np.unique(pd.to_numeric(vals, errors="coerce").dropna().to_numpy(np.float64))
            else:
                arr = np.array(sorted(pd.unique(vals.astype(object)).tolist()), dtype=object)
            self.dicts[c] = ColumnDict(arr, num)

    @property
    def n_codes(self) -> np.ndarray:
        return np.array([max(len(self.dicts[c]), 1) for c in self.columns], np.int32)

    def encode(self, df: pd.DataFrame) -> np.ndarray:
        """-> [C][N] int32, C-contiguous (column-major table)."""
        out = np.empty((len(self.columns), len(df)), np.int32)
        for i, c in enumerate(self.columns):
            out[i] = self.dicts[c].encode(df[c])
        return out
