"""Entry point kept from the reference: `Delphi.getOrCreate().repair` hands out a fresh `RepairModel`, `.misc` a
`RepairMisc`, `.version()` the package version (reference python/repair/api.py:26-63).  Tables the reference would find in
Spark's catalog are pandas frames registered by name with `Delphi.register_table`."""
import pandas as pd

from repair import session
from repair.misc import RepairMisc
from repair.model import RepairModel

__version__ = "0.1.0-mi355x"


class Delphi():
    """Process-wide facade; every call of `Delphi()` / `Delphi.getOrCreate()` yields the same object."""

    _the_one = None

    def __new__(cls):  # type: ignore
        if Delphi._the_one is None:
            Delphi._the_one = object.__new__(cls)
        return Delphi._the_one

    @classmethod
    def getOrCreate(cls) -> "Delphi":
        return cls()

    @staticmethod
    def version() -> str:
        return __version__

    @staticmethod
    def register_table(name: str, df: pd.DataFrame) -> None:
        """Stands in for `createOrReplaceTempView` / `saveAsTable`: makes `df` reachable as `setTableName(name)`."""
        session.register_table(name, df)

    @property
    def repair(self) -> RepairModel:
        return RepairModel()

    @property
    def misc(self) -> RepairMisc:
        return RepairMisc()


delphi = Delphi.getOrCreate()
