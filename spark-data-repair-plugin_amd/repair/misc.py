"""`RepairMisc` (reference python/repair/misc.py:27-131,159-260): the helper API around the repair path, on pandas frames.

Only the helpers that touch the path's inputs and outputs are here -- `repair` (apply predicted updates,
RepairMiscApi.repairAttrsFrom), `flatten` (RepairMiscApi.flattenTable), `injectNull` (RepairMiscApi.injectNullAt, the error
injector the synthetic benchmark tables use) and `splitInputTable`'s argument checks; `describe`, `toHistogram`,
`toErrorMap` and `generateDepGraph` are analysis / plotting utilities outside the rebuilt path.
"""
from typing import Dict, List

import numpy as np
import pandas as pd

from repair import session
from repair.utils import argtype_check

DataFrame = pd.DataFrame


class RepairMisc():
    """Interface to provide helper functionalities."""

    def __init__(self) -> None:
        self.opts: Dict[str, str] = {}

    @argtype_check  # type: ignore
    def option(self, key: str, value: str) -> "RepairMisc":
        self.opts[str(key)] = str(value)
        return self

    @argtype_check  # type: ignore
    def options(self, options: Dict[str, str]) -> "RepairMisc":
        self.opts.update(options)
        return self

    @property
    def _db_name(self) -> str:
        return self.opts.get("db_name", "")

    @property
    def _target_attr_list(self) -> str:
        return self.opts.get("target_attr_list", "")

    def _check_required_options(self, required: List[str]) -> None:
        if not all(opt in self.opts.keys() for opt in required):
            raise ValueError("Required options not found: {}".format(", ".join(required)))

    def _qualified(self, name: str) -> str:
        return "%s.%s" % (self._db_name, name) if self._db_name else name

    def _input(self) -> DataFrame:
        return session.resolve(self.opts["table_name"])

    def _check_attrs(self, df: DataFrame, attrs: List[str]) -> None:
        missing = [a for a in attrs if a not in df.columns]
        if missing:
            raise ValueError("Columns '%s' do not exist in '%s'" % (", ".join(missing), self._qualified(self.opts["table_name"])))

    def repair(self) -> DataFrame:
        """Applies predicted repair updates into an input table (RepairMiscApi.scala:184-247)."""
        self._check_required_options(["repair_updates", "table_name", "row_id"])
        from repair.model import RepairModel
        updates = session.resolve(self.opts["repair_updates"])
        rid = self.opts["row_id"]
        if not {rid, "attribute", "repaired"} <= set(updates.columns):
            raise ValueError("Table '%s' must have '%s', 'attribute', and 'repaired' columns" % (self.opts["repair_updates"], rid))
        return RepairModel().setRowId(rid)._repair_attrs(updates[[rid, "attribute", "repaired"]], self._input())

    def flatten(self) -> DataFrame:
        """<row_id, attribute, value> with `CAST(value AS STRING)` (RepairMiscApi.scala:41-49); attribute-major like the
        reference's `INLINE(ARRAY(STRUCT..))` is row-major -- callers sort, so is the order here (row, then column)."""
        self._check_required_options(["table_name", "row_id"])
        from repair.errors import _to_sql_string
        df, rid = self._input(), self.opts["row_id"]
        cols = [c for c in df.columns if c != rid]
        ids = np.repeat(df[rid].to_numpy(), len(cols))
        attrs = np.tile(np.asarray(cols, object), len(df))
        vals = np.empty(len(df) * len(cols), object)
        for j, c in enumerate(cols):
            vals[j::len(cols)] = [None if pd.isna(v) else _to_sql_string(v) for v in df[c].to_numpy(dtype=object)]
        return pd.DataFrame({rid: ids, "attribute": attrs, "value": vals})

    def splitInputTable(self) -> DataFrame:
        self._check_required_options(["table_name", "row_id", "k"])
        if not self.opts["k"].isdigit():
            raise ValueError("Option 'k' must be an integer, but '%s' found" % self.opts["k"])
        raise NotImplementedError("splitInputTable (k-means over q-gram features) is outside the rebuilt path")

    def injectNull(self) -> DataFrame:
        """Randomly injects NULL into the given attributes: `IF(rand() > ratio, col, NULL)` per cell
        (RepairMiscApi.scala:155-182).  The reference's `rand()` is unseeded; option `seed` (default 0) fixes it here."""
        self._check_required_options(["table_name", "target_attr_list"])
        if "null_ratio" in self.opts.keys():
            try:
                ratio = float(self.opts["null_ratio"])
                ok = True
            except ValueError:
                ok = False
            if not (ok and 0.0 < ratio <= 1.0):
                raise ValueError("Option 'null_ratio' must be a float in (0.0, 1.0], but '%s' found" % self.opts["null_ratio"])
        else:
            ratio = 0.01
        df = self._input().copy()
        attrs = [a.strip() for a in self._target_attr_list.split(",") if a.strip()]
        self._check_attrs(df, attrs)
        rng = np.random.Generator(np.random.PCG64(int(self.opts.get("seed", "0"))))
        for a in attrs:
            keep = rng.random(len(df)) > ratio
            col = df[a]
            if pd.api.types.is_integer_dtype(col) and not str(col.dtype).startswith(("Int", "UInt")):
                col = col.astype("Int64")
            df[a] = col.where(keep, other=pd.NA if str(col.dtype).startswith(("Int", "UInt")) else None)
        return df

    def describe(self) -> DataFrame:
        raise NotImplementedError("describe (column statistics) is outside the rebuilt path")

    def toHistogram(self) -> DataFrame:
        raise NotImplementedError("toHistogram is outside the rebuilt path")

    def toErrorMap(self) -> DataFrame:
        raise NotImplementedError("toErrorMap is outside the rebuilt path")

    def generateDepGraph(self) -> None:
        raise NotImplementedError("generateDepGraph is outside the rebuilt path")
