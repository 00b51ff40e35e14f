"""A tiny in-process table catalog standing in for the Spark session catalog, so that the
reference's name-based entry points (`setTableName`, `setInput("adult")`, `setErrorCells("cells")`)
keep working on a Spark-free host."""
from typing import Dict

import pandas as pd

_tables: Dict[str, pd.DataFrame] = {}


def register_table(name: str, df: pd.DataFrame) -> None:
    _tables[name] = df


def drop_table(name: str) -> None:
    _tables.pop(name, None)


def table(name: str) -> pd.DataFrame:
    if name not in _tables:
        raise ValueError("Table or view not found: %s" % name)
    return _tables[name]


def resolve(obj) -> pd.DataFrame:  # type: ignore
    return obj if isinstance(obj, pd.DataFrame) else table(str(obj))
