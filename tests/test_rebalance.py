"""`rebalance_training_data` (reference python/repair/train.py:242-293): SMOTEN for the classes below the median class size,
RandomUnderSampler for the ones above it -- restated from imbalanced-learn 0.8.0 (pinned by the reference, not installed here)."""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))

from repair.train import random_under_sample, rebalance_training_data, smoten_resample  # noqa: E402


def _frame(seed=0, n=400):
    rng = np.random.default_rng(seed)
    y = pd.Series(rng.choice(["a", "b", "c", "d"], n, p=[0.55, 0.25, 0.15, 0.05]), name="t")
    X = pd.DataFrame({"f1": rng.choice(["x", "y", "z"], n), "f2": rng.integers(0, 5, n).astype(object), "f3": rng.choice(["p", "q"], n)})
    X.loc[rng.choice(n, 20, replace=False), "f2"] = None
    return X, y


def test_every_class_ends_at_the_median():
    X, y = _frame()
    median = int(np.median(list(y.value_counts().to_dict().values())))
    Xr, yr = rebalance_training_data(X.copy(), y.copy(), "t")
    assert set(yr.value_counts().to_dict().values()) == {median}
    assert len(Xr) == len(yr) == 4 * median
    # deterministic
    Xr2, yr2 = rebalance_training_data(X.copy(), y.copy(), "t")
    assert Xr.equals(Xr2) and yr.equals(yr2)


def test_rows_with_nulls_are_never_synthesised_from():
    X, y = _frame(seed=3)
    has_na = X.isnull().any(axis=1)
    Xr, yr = rebalance_training_data(X.copy(), y.copy(), "t")
    small = [c for c, n in y.value_counts().items() if n < int(np.median(list(y.value_counts().values)))]
    for c in small:      # the over-sampled classes keep exactly their original rows with NULLs
        assert int(Xr[yr == c].isnull().any(axis=1).sum()) == int((has_na & (y == c)).sum())


def test_a_class_with_too_few_clean_rows_is_left_alone():
    rng = np.random.default_rng(5)
    y = pd.Series(["a"] * 60 + ["b"] * 30 + ["c"] * 4, name="t")
    X = pd.DataFrame({"f1": rng.choice(["x", "y"], 94), "f2": rng.choice(["u", "v", "w"], 94)})
    Xr, yr = rebalance_training_data(X, y, "t")
    cnt = yr.value_counts().to_dict()
    assert cnt == {"a": 30, "b": 30, "c": 4}


def test_smoten_by_hand():
    """Six rows of class 'm' (+ six of class 'o' that shape the value-difference metric).  With one feature that separates two groups
    of three inside 'm', every row's neighbours start with the two rows of its own group; with k = 2 a new row copies its group."""
    X = pd.DataFrame({"g": ["A", "A", "A", "B", "B", "B"] + ["A"] * 5 + ["C"], "h": ["u", "u", "v", "w", "w", "w"] + ["u", "v", "w", "u", "v", "w"]})
    y = pd.Series(["m"] * 6 + ["o"] * 6, name="t")
    Xn, yn = smoten_resample(X, y, {"m": 10}, k_neighbors=2, random_state=42)
    assert len(Xn) == 16 and yn.tolist()[-4:] == ["m"] * 4 and Xn.iloc[:12].equals(X)
    picks = np.random.RandomState(42).choice(np.arange(6), size=4, replace=True)
    for row, src in zip(Xn.iloc[12:].itertuples(index=False), picks):
        group = "A" if src < 3 else "B"
        assert row.g == group                                    # both neighbours come from the row's own group
        assert row.h in set(X["h"][:3] if group == "A" else X["h"][3:6])


def test_random_under_sampler_order_and_counts():
    y = pd.Series(list("bbbbbaaaaaaccc"), name="t")
    X = pd.DataFrame({"i": np.arange(len(y))})
    Xu, yu = random_under_sample(X, y, {"a": 2, "b": 3}, random_state=42)
    assert yu.tolist() == ["a"] * 2 + ["b"] * 3 + ["c"] * 3      # classes in sorted order, untouched classes whole
    rs = np.random.RandomState(42)
    a_rows = np.flatnonzero(y.to_numpy() == "a"); b_rows = np.flatnonzero(y.to_numpy() == "b")
    exp = list(a_rows[rs.choice(np.arange(6), size=2, replace=False)]) + list(b_rows[rs.choice(np.arange(5), size=3, replace=False)]) + [11, 12, 13]
    assert Xu["i"].tolist() == exp
