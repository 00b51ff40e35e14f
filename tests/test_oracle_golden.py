"""Pins the CPU oracle (and the Python host pipeline above it) to EVERY golden label the reference's
own tests hold for this path:

  * bin/testdata/adult_repair.csv -- 7 labels (test_model.py:87-89, used by 6 tests)
  * test_error_cells_having_no_existent_attribute (test_model.py:493-508) -- 2 labels
  * test_integer_input (test_model.py:1121-1146) -- 4 labels (regressors on integral columns + rounding)
  * test_escaped_column_names (test_model.py:687-721) -- 2 labels + repaired rows

The reference builds these with `model.hp.max_evals=1` (one hyperopt sample).  On 6..20-row tables
LightGBM cannot split for any min_child_samples >= 9 (5 for the 9-row table), so the answers are
decided by BoostFromScore on the float32 'balanced' class weights, the constant first tree,
first-max arg-max and half-even integral rounding -- exactly the semantics restated in
oracle/rgbm_oracle.c.  (They do not pin tree growth; the GPU<->oracle bit-exactness tests and the
accuracy floors in test_quality.py cover that.)
"""
import numpy as np
import pandas as pd
import pytest

from tests.helpers import frame, load_golden


def _model(oracle_backend):
    from repair.model import RepairModel
    from repair.errors import NullErrorDetector
    # `_build_model()` of the reference's test suite (test_model.py:268-273)
    return RepairModel().setErrorDetectors([NullErrorDetector()]).option("model.hp.max_evals", "1")


def _rows(df, rid):
    out = df.sort_values([rid, "attribute"]).reset_index(drop=True)
    return [[r[0], r[1], None if pd.isna(r[2]) else r[2], r[3]] for r in out[[rid, "attribute", "current_value", "repaired"]].itertuples(index=False)]


def test_adult_repair_golden(oracle_backend):
    g = load_golden("adult")
    df = frame(g["input"])
    out = _model(oracle_backend).setInput(df).setRowId("tid").run()
    exp = frame(g["expected_repair"], dtypes=False)
    exp = sorted([[int(r[0]), r[1], None, r[3]] for r in exp.itertuples(index=False)], key=lambda r: (r[0], r[1]))
    assert _rows(out, "tid") == exp
    assert len(exp) == 7


def test_adult_repair_data_matches_adult_clean(oracle_backend):
    g = load_golden("adult")
    df = frame(g["input"])
    out = _model(oracle_backend).setInput(df).setRowId("tid").run(repair_data=True).sort_values("tid").reset_index(drop=True)
    clean = frame(g["clean"]).sort_values("tid").reset_index(drop=True)
    assert list(out.columns) == list(clean.columns)
    assert out.astype(str).values.tolist() == clean.astype(str).values.tolist()


def test_adult_table_name_and_targets(oracle_backend):
    from repair.api import Delphi
    g = load_golden("adult")
    Delphi.register_table("adult", frame(g["input"]))
    out = _model(oracle_backend).setTableName("adult").setRowId("tid").setTargets(["Sex"]).run()
    assert _rows(out, "tid") == [[3, "Sex", None, "Male"], [7, "Sex", None, "Male"], [12, "Sex", None, "Male"]]


def test_error_cells_having_no_existent_attribute(oracle_backend):
    g = load_golden("adult"); ig = load_golden("inline_goldens")["error_cells_no_existent_attribute"]
    cells = pd.DataFrame(ig["error_cells"], columns=["tid", "attribute"]).astype({"tid": str})
    out = _model(oracle_backend).setInput(frame(g["input"])).setRowId("tid").setErrorCells(cells).run()
    assert _rows(out, "tid") == ig["expected"]


def test_integer_input(oracle_backend):
    ig = load_golden("inline_goldens")["integer_input"]
    df = pd.DataFrame(ig["rows"], columns=ig["columns"]).astype({c: "Int64" for c in ig["columns"][1:]})
    out = _model(oracle_backend).setInput(df).setRowId("tid").run()
    assert _rows(out, "tid") == ig["expected"]


def test_escaped_column_names(oracle_backend):
    ig = load_golden("inline_goldens")["escaped_column_names"]
    df = pd.DataFrame(ig["rows"], columns=ig["columns"])
    m = _model(oracle_backend).setInput(df).setRowId("t i d").setDiscreteThreshold(ig["discrete_threshold"])
    assert _rows(m.run(), "t i d") == ig["expected"]
    keys = m.run(compute_repair_candidate_prob=True).sort_values(["t i d", "attribute"])[["t i d", "attribute"]].values.tolist()
    assert keys == [[1, "y y"], [2, "x x"]]
    keys = m.run(compute_repair_prob=True).sort_values(["t i d", "attribute"])[["t i d", "attribute"]].values.tolist()
    assert keys == [[1, "y y"], [2, "x x"]]
    rep = m.run(repair_data=True)
    rep = rep[rep["t i d"].isin([1, 2])].sort_values("t i d").values.tolist()
    assert rep == ig["expected_repair_data_rows_1_2"]


def test_estimator_protocol_goldens():
    from repair.model import FunctionalDepModel, PoorModel
    ep = load_golden("inline_goldens")["estimator_protocol"]
    fd = ep["fd_model"]
    m = FunctionalDepModel(fd["x"], {k: v for k, v in fd["fd_map"]})
    pdf = pd.DataFrame([[v] for v in fd["inputs"]], columns=[fd["x"]])
    assert m.classes_.tolist() == fd["classes"]
    assert m.predict(pdf) == fd["predict"]
    pmf = m.predict_proba(pdf)
    assert [None if p is None else p.tolist() for p in pmf] == fd["proba"]
    for v in ep["poor_model"]["values"]:
        pm = PoorModel(v)
        assert pm.classes_.tolist() == [v]
        assert pm.predict(pdf) == [v] * 4
        assert [p.tolist() for p in pm.predict_proba(pdf)] == [[1.0]] * 4
