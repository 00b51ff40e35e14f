"""`Delphi` facade (reference python/repair/api.py:26-63): `Delphi.getOrCreate().repair` gives a fresh
RepairModel, `.version()` the package version.  `register_table` stands in for Spark's catalog."""
from typing import Any

import pandas as pd

from repair import session
from repair.misc import RepairMisc
from repair.model import RepairModel


class Delphi():
    """A Delphi API set for data repairing."""

    _instance: Any = None
    __version__ = "0.1.0-mi355x"

    def __new__(cls, *args, **kwargs):  # type: ignore
        if cls._instance is None:
            cls._instance = super(Delphi, cls).__new__(cls)
        return cls._instance

    @staticmethod
    def getOrCreate() -> "Delphi":
        return Delphi()

    @property
    def repair(self) -> RepairModel:
        """Returns :class:`RepairModel` to repair input data."""
        return RepairModel()

    @property
    def misc(self) -> RepairMisc:
        """Returns :class:`RepairMisc` for misc helper functions."""
        return RepairMisc()

    @staticmethod
    def version() -> str:
        return Delphi.__version__

    @staticmethod
    def register_table(name: str, df: pd.DataFrame) -> None:
        """Registers a DataFrame under a table name (stands in for createOrReplaceTempView)."""
        session.register_table(name, df)


delphi = Delphi.getOrCreate()
