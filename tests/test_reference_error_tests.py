"""Mirrors of python/repair/tests/test_errors.py and test_costs.py: the detector plug-in API and the update cost functions,
same calls, same expected rows / messages (pandas frames instead of Spark views)."""
import numpy as np
import pandas as pd
import pytest

from repair.api import Delphi
from repair.errors import (ConstraintErrorDetector, DomainValues, GaussianOutlierErrorDetector, LOFOutlierErrorDetector, NullErrorDetector,
                           RegExErrorDetector, ScikitLearnBackedErrorDetector)
from tests.helpers import frame, load_golden


@pytest.fixture
def adult(tmp_path):
    g = load_golden("adult")
    Delphi.register_table("adult", frame(g["input"]))
    path = tmp_path / "adult_constraints.txt"
    path.write_text(g["constraints"])
    return str(path)


def _cells(df, rid="tid"):
    return sorted((int(t), a) for t, a in df[[rid, "attribute"]].values.tolist())


def test_NullErrorDetector(adult):
    assert _cells(NullErrorDetector().setUp("tid", "adult", [], ["Sex", "Age", "Income"]).detect()) == [
        (3, "Sex"), (5, "Age"), (5, "Income"), (7, "Sex"), (12, "Age"), (12, "Sex"), (16, "Income")]
    assert _cells(NullErrorDetector().setUp("tid", "adult", [], ["Sex"]).detect()) == [(3, "Sex"), (7, "Sex"), (12, "Sex")]
    assert _cells(NullErrorDetector().setUp("tid", "adult", [], ["Age", "Income"]).detect()) == [(5, "Age"), (5, "Income"), (12, "Age"), (16, "Income")]
    assert _cells(NullErrorDetector().setUp("tid", "adult", [], ["Income", "Unknown"]).detect()) == [(5, "Income"), (16, "Income")]
    assert len(NullErrorDetector().setUp("tid", "adult", [], ["Non-existent"]).detect()) == 0


def test_DomainValues(adult):
    assert _cells(DomainValues("Country", []).setUp("tid", "adult", [], ["Country"]).detect()) == [(i, "Country") for i in range(20)]
    assert _cells(DomainValues("Country", ["United-States"]).setUp("tid", "adult", [], ["Country"]).detect()) == [(7, "Country"), (19, "Country")]
    assert _cells(DomainValues("Income", ["LessThan50K", "MoreThan50K"]).setUp("tid", "adult", [], ["Income"]).detect()) == [(5, "Income"), (16, "Income")]
    assert _cells(DomainValues("Country", autofill=True, min_count_thres=4).setUp("tid", "adult", [], ["Country"]).detect()) == [(7, "Country"), (19, "Country")]
    assert _cells(DomainValues("Income", autofill=True, min_count_thres=1).setUp("tid", "adult", [], ["Income"]).detect()) == [(5, "Income"), (16, "Income")]
    assert len(DomainValues("Country", []).setUp("tid", "adult", [], ["Non-existent"]).detect()) == 0


def test_RegExErrorDetector(adult):
    for targets in (["Country"], ["Unknown", "Country"]):
        assert _cells(RegExErrorDetector("Country", "United-States").setUp("tid", "adult", [], targets).detect()) == [(7, "Country"), (19, "Country")]
    Delphi.register_table("tempView", pd.DataFrame([(1, 12), (2, 123), (3, 1234), (4, 12345)], columns=["tid", "v"]))
    assert _cells(RegExErrorDetector("v", "123.+").setUp("tid", "tempView", [], ["v"]).detect()) == [(1, "v"), (2, "v")]
    assert len(RegExErrorDetector("Country", "United-States").setUp("tid", "adult", [], ["Non-existent"]).detect()) == 0


def test_ConstraintErrorDetector(adult):
    both = [(4, "Relationship"), (4, "Sex"), (11, "Relationship"), (11, "Sex")]
    assert _cells(ConstraintErrorDetector(adult).setUp("tid", "adult", [], ["Relationship", "Sex"]).detect()) == both
    assert _cells(ConstraintErrorDetector(adult, targets=["Relationship"]).setUp("tid", "adult", [], ["Relationship", "Sex"]).detect()) == [c for c in both if c[1] == "Relationship"]
    assert _cells(ConstraintErrorDetector(adult).setUp("tid", "adult", [], ["Relationship"]).detect()) == [c for c in both if c[1] == "Relationship"]
    assert _cells(ConstraintErrorDetector(adult).setUp("tid", "adult", [], ["Sex", "Relationship"]).detect()) == both
    assert _cells(ConstraintErrorDetector(adult).setUp("tid", "adult", [], ["Unknown", "Sex"]).detect()) == [c for c in both if c[1] == "Sex"]
    with pytest.raises(ValueError, match="At least one of `constraint_path` or `constraints` should be specified"):
        ConstraintErrorDetector()
    assert len(ConstraintErrorDetector(adult).setUp("tid", "adult", [], ["Non-existent"]).detect()) == 0
    assert len(ConstraintErrorDetector(adult).setUp("tid", "adult", [], ["Income"]).detect()) == 0


def test_GaussianOutlierErrorDetector():
    Delphi.register_table("tempView", pd.DataFrame([(1, 1.0), (2, 1.0), (3, 1.0), (4, 1000.0), (5, None)], columns=["tid", "v"]))
    for approx in (True, False):
        for targets in (["v"], ["Unknown", "v"]):
            assert _cells(GaussianOutlierErrorDetector(approx).setUp("tid", "tempView", ["v"], targets).detect()) == [(4, "v")]
        assert len(GaussianOutlierErrorDetector(approx).setUp("tid", "tempView", ["v"], ["Non-existent"]).detect()) == 0


@pytest.mark.parametrize("nrows", [3000, 10000])
def test_LOF_and_ScikitLearnBackedErrorDetector(nrows):
    from sklearn.neighbors import LocalOutlierFactor
    ids = np.arange(nrows)
    df = pd.DataFrame({"id": ids, "v1": (ids % 2).astype("float64"), "v2": (ids % 3).astype("float64")})
    dirty = pd.DataFrame([(1000000, 1, 1000), (1000001, 1000, 1), (1000002, None, None)], columns=["id", "v1", "v2"])
    Delphi.register_table("tempView", pd.concat([df, dirty], ignore_index=True))
    with pytest.raises(ValueError, match="`num_parallelism` must be positive, got 0"):
        LOFOutlierErrorDetector(5000, num_parallelism=0)
    with pytest.raises(ValueError, match="`error_detector_cls` should be callable"):
        ScikitLearnBackedErrorDetector(error_detector_cls=1, parallel_mode_threshold=5000, num_parallelism=1)
    with pytest.raises(ValueError, match="An instance that `error_detector_cls` returns should have a `fit_predict` method"):
        ScikitLearnBackedErrorDetector(error_detector_cls=lambda: 1, parallel_mode_threshold=5000, num_parallelism=1)
    with pytest.raises(ValueError, match="`num_parallelism` must be positive, got 0"):
        ScikitLearnBackedErrorDetector(error_detector_cls=lambda: LocalOutlierFactor(novelty=False), parallel_mode_threshold=5000, num_parallelism=0)
    for make in (lambda: LOFOutlierErrorDetector(5000, num_parallelism=1),
                 lambda: ScikitLearnBackedErrorDetector(error_detector_cls=lambda: LocalOutlierFactor(novelty=False), parallel_mode_threshold=5000, num_parallelism=1)):
        assert _cells(make().setUp("id", "tempView", ["v1", "v2"], ["v1", "v2"]).detect(), "id") == [(1000000, "v2"), (1000001, "v1")]
        assert _cells(make().setUp("id", "tempView", ["v1", "v2"], ["v1"]).detect(), "id") == [(1000001, "v1")]
        assert _cells(make().setUp("id", "tempView", ["v1", "v2"], ["Unknown", "v1"]).detect(), "id") == [(1000001, "v1")]
        assert len(make().setUp("id", "tempView", ["v1", "v2"], ["Non-existent"]).detect()) == 0


def test_cost_functions_like_test_costs():
    """python/repair/tests/test_costs.py:26-68 (the user-defined distance there uses the `Levenshtein` package; its edit distance
    is restated inline here)."""
    from repair.costs import Levenshtein, UserDefinedUpdateCostFunction

    def lev(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]

    f = Levenshtein()
    for x, y in (("111", "123"), (111, 123), ("111", 123), (111, "123"), (1.11, 1.23), ("1.11", 1.23), (1.11, "1.23")):
        assert f.compute(x, y) == pytest.approx(2.0)
    assert f.compute(None, "123") is None and f.compute("111", None) is None and f.compute(None, None) is None
    assert f.compute("1xx%", "100%") < f.compute("1xx%", "abcdefg")
    assert f.compute("1xx%", "100%") == pytest.approx(f.compute("1xx%", "12%")) == pytest.approx(f.compute("1xx%", "1%"))
    assert f.compute("1xx%", "100%") < f.compute("1xx%", "2%")
    g = UserDefinedUpdateCostFunction(f=lambda x, y: float(abs(len(str(x)) - len(str(y))) + lev(str(x), str(y))))
    for x, y in (("111", "123"), (111, 123), ("111", 123), (111, "123"), (1.11, 1.23), ("1.11", 1.23), (1.11, "1.23")):
        assert g.compute(x, y) == pytest.approx(2.0)
    assert g.compute(None, "123") is None and g.compute("111", None) is None and g.compute(None, None) is None
    for other in ("abcdefg", "12%", "1%", "2%"):
        assert g.compute("1xx%", "100%") < g.compute("1xx%", other)
    with pytest.raises(ValueError, match="`f` should take two values and return a float cost value"):
        UserDefinedUpdateCostFunction(f=lambda x, y: lev(str(x), str(y)))
    with pytest.raises(ValueError, match="`f` should take two values and return a float cost value"):
        UserDefinedUpdateCostFunction(f=lambda x: x)
