"""Update-cost functions (reference python/repair/costs.py:25-78).  Off the hot path: only the pmf
re-weighting and maximal-likelihood modes consult them.  python-Levenshtein is not installed in this
image, so the edit distance is a small pure-Python DP with the same value."""
from abc import ABCMeta, abstractmethod
from typing import Callable, List, Optional, Union


class UpdateCostFunction(metaclass=ABCMeta):

    def __init__(self, targets: List[str] = []) -> None:
        self.targets: List[str] = targets

    @abstractmethod
    def _compute_impl(self, x: Union[str, int, float], y: Union[str, int, float]) -> Optional[float]:
        pass

    def compute(self, x: Optional[Union[str, int, float]], y: Optional[Union[str, int, float]]) -> Optional[float]:
        return self._compute_impl(x, y) if x and y else None


def _edit_distance(a: str, b: str) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class Levenshtein(UpdateCostFunction):

    def __init__(self, targets: List[str] = []) -> None:
        UpdateCostFunction.__init__(self, targets)

    def __str__(self) -> str:
        params = "targets=%s" % ",".join(self.targets) if self.targets else ""
        return "%s(%s)" % (self.__class__.__name__, params)

    def _compute_impl(self, x: Union[str, int, float], y: Union[str, int, float]) -> Optional[float]:
        return float(_edit_distance(str(x), str(y)))


class UserDefinedUpdateCostFunction(UpdateCostFunction):

    def __init__(self, f: Callable[[str, str], float], targets: List[str] = []) -> None:
        UpdateCostFunction.__init__(self, targets)
        try:
            ok = type(f("x", "y")) is float
        except Exception:
            ok = False
        if not ok:
            raise ValueError("`f` should take two values and return a float cost value")
        self._f = f

    def __str__(self) -> str:
        params = "targets=%s" % ",".join(self.targets) if self.targets else ""
        return "%s(%s)" % (self.__class__.__name__, params)

    def _compute_impl(self, x: Union[str, int, float], y: Union[str, int, float]) -> Optional[float]:
        try:
            return float(self._f(str(x), str(y)))
        except Exception:
            return None
