"""Update-cost functions: how expensive is it to change a cell from `x` to `y`.

Same public surface as the reference's `repair.costs` (python/repair/costs.py:25-78 -- `UpdateCostFunction` with a
`targets` list and `compute(x, y)`, `Levenshtein`, `UserDefinedUpdateCostFunction`), so detectors / models written
against it keep working.  They sit off the accelerated path: only the pmf re-weighting, the score and the
maximal-likelihood modes consult them.  The `Levenshtein` package is not part of this image; the edit distance is a
two-row dynamic programme with the same value.
"""
from abc import ABCMeta, abstractmethod
from typing import Callable, List, Optional, Union

Value = Union[str, int, float]


def edit_distance(a: str, b: str) -> int:
    """Levenshtein distance (insert / delete / substitute, unit costs)."""
    if len(b) > len(a):
        a, b = b, a
    row = list(range(len(b) + 1))
    for i, ca in enumerate(a, start=1):
        nxt = [i] + [0] * len(b)
        for j, cb in enumerate(b, start=1):
            nxt[j] = min(row[j] + 1, nxt[j - 1] + 1, row[j - 1] + (0 if ca == cb else 1))
        row = nxt
    return row[len(b)]


class UpdateCostFunction(metaclass=ABCMeta):
    """Base class of the plug-in: subclasses implement `_compute_impl`; `targets` restricts the attributes it applies to
    (empty = all)."""

    def __init__(self, targets: List[str] = []) -> None:
        self.targets: List[str] = targets

    def __str__(self) -> str:
        inner = ("targets=" + ",".join(self.targets)) if self.targets else ""
        return type(self).__name__ + "(" + inner + ")"

    def compute(self, x: Optional[Value], y: Optional[Value]) -> Optional[float]:
        # a missing (or empty) side has no defined cost -- callers treat None as "leave the probability alone"
        if not x or not y:
            return None
        return self._compute_impl(x, y)

    @abstractmethod
    def _compute_impl(self, x: Value, y: Value) -> Optional[float]:
        ...


class Levenshtein(UpdateCostFunction):
    """Edit distance between the string forms of the two values."""

    def __init__(self, targets: List[str] = []) -> None:
        super().__init__(targets)

    def _compute_impl(self, x: Value, y: Value) -> Optional[float]:
        return float(edit_distance(str(x), str(y)))


class UserDefinedUpdateCostFunction(UpdateCostFunction):
    """Wraps a callable `f(str, str) -> float`; a callable that fails on a pair yields no cost for that pair."""

    def __init__(self, f: Callable[[str, str], float], targets: List[str] = []) -> None:
        super().__init__(targets)
        if not self._returns_float(f):
            raise ValueError("`f` should take two values and return a float cost value")
        self._f = f

    @staticmethod
    def _returns_float(f: Callable[[str, str], float]) -> bool:
        try:
            return type(f("x", "y")) is float
        except Exception:
            return False

    def _compute_impl(self, x: Value, y: Value) -> Optional[float]:
        try:
            return float(self._f(str(x), str(y)))
        except Exception:
            return None
