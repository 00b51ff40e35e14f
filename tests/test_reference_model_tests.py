"""Mirrors of the reference's own API tests (python/repair/tests/test_model.py) on the adult fixture: same calls, same
expected rows / messages, pandas frames instead of Spark tables, the CPU oracle as the compute backend.  Each test names
the reference test it follows."""
import os

import numpy as np
import pandas as pd
import pytest

from tests.helpers import frame, load_golden

ADULT_ERRORS = [(3, "Sex"), (5, "Age"), (5, "Income"), (7, "Sex"), (12, "Age"), (12, "Sex"), (16, "Income")]


@pytest.fixture
def adult(oracle_backend):
    from repair.api import Delphi
    g = load_golden("adult")
    df = frame(g["input"])
    Delphi.register_table("adult", df)
    dirty = pd.DataFrame(ADULT_ERRORS, columns=["tid", "attribute"])
    Delphi.register_table("adult_dirty", dirty)
    exp = frame(g["expected_repair"], dtypes=False)
    expected = sorted([[int(r[0]), r[1], None, r[3]] for r in exp.itertuples(index=False)], key=lambda r: (r[0], r[1]))
    return dict(df=df, dirty=dirty, expected=expected, constraints=g["constraints"])


def _build_model():
    from repair.errors import NullErrorDetector
    from repair.model import RepairModel
    return RepairModel().setErrorDetectors([NullErrorDetector()]).option("model.hp.max_evals", "1")


def _rows(df, cols=("tid", "attribute", "current_value", "repaired")):
    out = df.sort_values(["tid", "attribute"]).reset_index(drop=True)
    return [[None if (not isinstance(v, str) and pd.isna(v)) else (int(v) if c == "tid" else v) for c, v in zip(cols, r)]
            for r in out[list(cols)].itertuples(index=False)]


def test_options_keys_and_messages():
    """test_options / test_invalid_internal_options (test_model.py:275-328)."""
    from repair.model import RepairModel
    with pytest.raises(ValueError, match="Non-existent key specified: key=non-existent"):
        RepairModel().option("non-existent", "1")
    for key, value in [
            ("error.domain_threshold_alpha", "0.0"), ("error.domain_threshold_beta", "0.7"), ("error.max_attrs_to_compute_pairwise_stats", "3"),
            ("error.max_attrs_to_compute_domains", "2"), ("error.attr_freq_ratio_threshold", "0.0"), ("error.pairwise_freq_ratio_threshold", "0.05"),
            ("model.max_training_row_num", "100000"), ("model.max_training_column_num", "65536"), ("model.small_domain_threshold", "12"),
            ("model.rule.repair_by_nearest_values.disabled", "1"), ("model.rule.merge_threshold", "2.0"), ("model.rule.repair_by_regex.disabled", ""),
            ("model.rule.repair_by_functional_deps.disabled", ""), ("model.rule.max_domain_size", "1000"), ("repair.pmf.cost_weight", "0.1"),
            ("repair.pmf.prob_threshold", "0.0"), ("repair.pmf.prob_top_k", "80"), ("model.lgb.boosting_type", "gbdt"),
            ("model.lgb.class_weight", "balanced"), ("model.lgb.learning_rate", "0.01"), ("model.lgb.max_depth", "7"), ("model.lgb.max_bin", "255"),
            ("model.lgb.reg_alpha", "0.0"), ("model.lgb.min_split_gain", "0.0"), ("model.lgb.n_estimators", "300"), ("model.lgb.importance_type", "gain"),
            ("model.cv.n_splits", "3"), ("model.hp.timeout", "0"), ("model.hp.max_evals", "10000000"), ("model.hp.no_progress_loss", "50")]:
        RepairModel().option(key, value)


def test_invalid_internal_options(adult):
    with pytest.raises(ValueError, match='Failed to cast "invalid" into float data: key=error.attr_freq_ratio_threshold'):
        _build_model().setTableName("adult").setRowId("tid").option("error.attr_freq_ratio_threshold", "invalid").run()


def test_multiple_run_with_every_option_spelled_out(adult):
    """test_multiple_run (test_model.py:330-372): two runs of the same configuration give the golden rows."""
    def run():
        m = _build_model().setTableName("adult").setRowId("tid")
        for k, v in [("error.domain_threshold_alpha", "0.0"), ("error.domain_threshold_beta", "0.70"), ("error.max_attrs_to_compute_pairwise_stats", "3"),
                     ("error.max_attrs_to_compute_domains", "2"), ("error.attr_freq_ratio_threshold", "0.0"), ("error.pairwise_freq_ratio_threshold", "0.05"),
                     ("model.max_training_row_num", "10000"), ("model.max_training_column_num", "65536"), ("model.small_domain_threshold", "12"),
                     ("model.lgb.boosting_type", "gbdt"), ("model.lgb.class_weight", "balanced"), ("model.lgb.learning_rate", "0.01"),
                     ("model.lgb.max_depth", "7"), ("model.lgb.max_bin", "255"), ("model.lgb.reg_alpha", "0.0"), ("model.lgb.min_split_gain", "0.0"),
                     ("model.lgb.n_estimators", "300"), ("model.lgb.importance_type", "gain"), ("model.cv.n_splits", "3"), ("model.hp.timeout", "0"),
                     ("model.hp.max_evals", "1"), ("model.hp.no_progress_loss", "50")]:
            m = m.option(k, v)
        return _rows(m.run())
    assert run() == adult["expected"]
    assert run() == adult["expected"]


def test_parallel_stat_training_and_inputs(adult):
    """test_parallel_stat_training, test_setInput, test_input_overwrite (test_model.py:374-416)."""
    assert _rows(_build_model().setTableName("adult").setRowId("tid").setParallelStatTrainingEnabled(True).run()) == adult["expected"]
    assert _rows(_build_model().setInput("adult").setRowId("tid").run()) == adult["expected"]
    assert _rows(_build_model().setInput(adult["df"]).setRowId("tid").run()) == adult["expected"]
    assert _rows(_build_model().setDbName("default").setTableName("adult").setInput(adult["df"]).setRowId("tid").run()) == adult["expected"]


def test_setTargets(adult):
    """test_setTargets (test_model.py:418-448)."""
    for targets in (["Sex"], ["Sex", "Income"], ["Age", "Sex"], ["Non-Existent", "Age"]):
        got = _rows(_build_model().setInput("adult").setRowId("tid").setTargets(targets).run(), cols=("tid", "attribute"))
        assert got == [[t, a] for t, a in ADULT_ERRORS if a in targets]
    with pytest.raises(ValueError, match="Target attributes not found in adult: Non-Existent"):
        _build_model().setInput("adult").setRowId("tid").setTargets(["Non-Existent"]).run()


def test_setErrorCells(adult):
    """test_setErrorCells, test_setErrorCells_and_detect_errors_only (test_model.py:450-487)."""
    with pytest.raises(ValueError, match="`setRowId` should be called before specifying error cells"):
        _build_model().setErrorCells("adult_dirty").setInput("adult").setRowId("tid").run()
    with pytest.raises(ValueError, match="Error cells should have `tid` and `attribute` in columns"):
        _build_model().setInput("adult").setRowId("tid").setErrorCells("adult").run()
    for cells in ("adult_dirty", adult["dirty"], adult["dirty"].assign(unrelated=1)):
        assert _rows(_build_model().setTableName("adult").setRowId("tid").setErrorCells(cells).run()) == adult["expected"]
    got = _build_model().setTableName("adult").setRowId("tid").setErrorCells(adult["dirty"].assign(unrelated=1)).run(detect_errors_only=True)
    assert _rows(got, cols=("tid", "attribute", "current_value")) == [[t, a, None] for t, a in ADULT_ERRORS]


def test_detect_errors_only_every_detector(adult, tmp_path):
    """test_detect_errors_only (test_model.py:510-646)."""
    from repair.errors import ConstraintErrorDetector, DomainValues, RegExErrorDetector
    cols3 = ("tid", "attribute", "current_value")
    base = [[t, a, None] for t, a in ADULT_ERRORS]
    assert _rows(_build_model().setInput("adult").setRowId("tid").run(detect_errors_only=True), cols3) == base
    assert _rows(_build_model().setInput("adult").setRowId("tid").setTargets(["Sex", "Age", "Income"]).run(detect_errors_only=True), cols3) == base
    assert _rows(_build_model().setInput("adult").setRowId("tid").setTargets(["Sex"]).run(detect_errors_only=True), cols3) == [r for r in base if r[1] == "Sex"]
    assert _rows(_build_model().setInput("adult").setRowId("tid").setTargets(["Age", "Income"]).run(detect_errors_only=True), cols3) == [r for r in base if r[1] != "Sex"]
    assert _rows(_build_model().setInput("adult").setRowId("tid").setTargets(["Unknown", "Age"]).run(detect_errors_only=True), cols3) == [r for r in base if r[1] == "Age"]
    dets = [DomainValues("Country", ["United-States"]), DomainValues("Income", ["LessThan50K", "MoreThan50K"])]
    assert _rows(_build_model().setInput("adult").setRowId("tid").setErrorDetectors(dets).run(detect_errors_only=True), cols3) == [
        [5, "Income", None], [7, "Country", "India"], [16, "Income", None], [19, "Country", "Iran"]]
    dets = [RegExErrorDetector("Country", "United-States"), RegExErrorDetector("Relationship", "(Husband|Own-child|Not-in-family)")]
    got = _build_model().setInput("adult").setRowId("tid").setTargets(["Country", "Relationship"]).setErrorDetectors(dets).run(detect_errors_only=True)
    assert _rows(got, cols3) == [[7, "Country", "India"], [14, "Relationship", "Unmarried"], [16, "Relationship", "Unmarried"], [19, "Country", "Iran"]]
    path = tmp_path / "adult_constraints.txt"
    path.write_text(adult["constraints"])
    got = _build_model().setInput("adult").setRowId("tid").setTargets(["Sex", "Relationship"]) \
        .setErrorDetectors([ConstraintErrorDetector(str(path))]).run(detect_errors_only=True)
    assert _rows(got, cols3) == [[4, "Relationship", "Husband"], [4, "Sex", "Female"], [11, "Relationship", "Husband"], [11, "Sex", "Female"]]
    got = _build_model().setInput("adult").setRowId("tid").setTargets(["Sex", "Relationship"]) \
        .setErrorDetectors([ConstraintErrorDetector(str(path), targets=["Sex"])]).run(detect_errors_only=True)
    assert _rows(got, cols3) == [[4, "Sex", "Female"], [11, "Sex", "Female"]]
    m = _build_model().setInput("adult").setRowId("tid").setTargets(["Sex"]).setErrorDetectors([ConstraintErrorDetector(str(path), targets=["Sex", "Relationship"])])
    assert _rows(m.run(detect_errors_only=True), cols3) == [[4, "Sex", "Female"], [11, "Sex", "Female"]]
    assert _rows(m.setTargets(["Relationship"]).run(detect_errors_only=True), cols3) == [[4, "Relationship", "Husband"], [11, "Relationship", "Husband"]]


def test_DomainValues_against_continous_values(oracle_backend):
    """test_DomainValues_against_continous_values (test_model.py:648-675)."""
    from repair.errors import DomainValues, NullErrorDetector
    df = pd.DataFrame([(1, 1.0, 1.0, 1.0), (2, 1.1, 1.1, 1.1), (3, 1.0, 1.0, None), (4, 1.1, 1.0, 1.0), (5, 1.1, 1.1, 1.1), (6, 1.0, 1.0, None)],
                      columns=["tid", "x", "y", "z"])
    dets = [DomainValues("x", autofill=True, min_count_thres=2), DomainValues("y", autofill=True, min_count_thres=2),
            DomainValues("z", autofill=True, min_count_thres=2), NullErrorDetector()]
    got = _build_model().setInput(df).setRowId("tid").setErrorDetectors(dets).run(detect_errors_only=True)
    assert _rows(got, ("tid", "attribute", "current_value")) == [[3, "z", None], [6, "z", None]]


def test_max_training_column_num(adult):
    """test_max_training_column_num (test_model.py:749-758)."""
    out = _build_model().setTableName("adult").setRowId("tid").setDiscreteThreshold(5).option("model.max_training_column_num", "2").run()
    assert _rows(out) == adult["expected"]


def test_input_shape_checks(oracle_backend):
    """test_table_has_no_enough_columns, test_rowid_uniqueness (test_model.py:760-792)."""
    import re
    df = pd.DataFrame([(1, None), (2, "test-1"), (3, "test-1")], columns=["tid", "x"])
    from repair.api import Delphi
    Delphi.register_table("inputView", df)
    with pytest.raises(Exception, match=re.escape("A least three columns (`tid` columns + two more ones) in table 'inputView'")):
        _build_model().setTableName("inputView").setRowId("tid").run()
    df = pd.DataFrame([(1, 1, None), (1, 1, "test-1"), (1, 2, "test-1")], columns=["tid", "x", "y"])
    Delphi.register_table("inputView", df)
    with pytest.raises(Exception, match=re.escape("Uniqueness does not hold in column 'tid' of table 'inputView' (# of distinct 'tid': 1, # of rows: 3)")):
        _build_model().setTableName("inputView").setRowId("tid").run()
    # test_unsupported_types (test_model.py:737-747)
    import datetime
    Delphi.register_table("inputView", pd.DataFrame({"tid": [0], "x": [1], "y": [datetime.date(2021, 8, 1)]}))
    with pytest.raises(Exception, match="Supported types are tinyint,float,smallint,string,double,int,bigint, but unsupported ones found: date"):
        _build_model().setTableName("inputView").setRowId("tid").run()


def test_no_valid_discrete_feature_and_no_repairable_cell(oracle_backend):
    """test_no_valid_discrete_feature_exists_1/2, test_no_repairable_cell_exists (test_model.py:794-857)."""
    msg = "At least one valid discretizable feature is needed to repair error cells"
    df = pd.DataFrame([(1, "1", None), (2, "1", None), (3, "1", "test-1"), (4, "1", "test-1"), (5, "1", "test-1"), (6, "1", None)], columns=["tid", "x", "y"])
    with pytest.raises(ValueError, match=msg):
        _build_model().setInput(df).setRowId("tid").run()
    df = pd.DataFrame([(1, "1", None)] + [(i, str(i), "test-%d" % i) for i in range(2, 7)], columns=["tid", "x", "y"])
    m = _build_model().setInput(df).setRowId("tid").setDiscreteThreshold(3)
    with pytest.raises(ValueError, match=msg):
        m.run(detect_errors_only=False)
    assert _rows(m.run(detect_errors_only=True), ("tid", "attribute", "current_value")) == [[1, "y", None]]
    df = pd.DataFrame([(1, "1", None), (2, "2", None), (3, "1", "test-1"), (4, "1", "test-1"), (5, "1", "test-1"), (6, "1", None)], columns=["tid", "x", "y"])
    m = _build_model().setInput(df).setRowId("tid")
    with pytest.raises(ValueError, match=msg + ", but no such feature found"):
        m.run(detect_errors_only=False)
    assert _rows(m.run(detect_errors_only=True), ("tid", "attribute", "current_value")) == [[1, "y", None], [2, "y", None], [6, "y", None]]


def test_regressor_model(oracle_backend):
    """test_regressor_model (test_model.py:859-881)."""
    df = pd.DataFrame([(1, 1.0, 1.0, 1.0), (2, 1.5, 1.5, 1.5), (3, 1.4, 1.4, None), (4, 1.3, 1.3, 1.3), (5, 1.1, 1.1, 1.1), (6, 1.2, 1.2, None)],
                      columns=["tid", "x", "y", "z"])
    out = _build_model().setInput(df).setRowId("tid").run()
    assert _rows(out, ("tid", "attribute", "current_value")) == [[3, "z", None], [6, "z", None]]
    assert out["repaired"].notna().all()


def test_repair_by_functional_deps(oracle_backend, tmp_path):
    """test_repair_by_functional_deps (test_model.py:883-928)."""
    from repair.errors import ConstraintErrorDetector, NullErrorDetector
    df = pd.DataFrame([(1, "1", "test-1"), (2, "2", "test-2"), (3, "1", None), (4, "2", "test-2"), (5, "2", None), (6, "3", None)], columns=["tid", "x", "y"])
    cells = pd.DataFrame([(3, "y"), (5, "y"), (6, "y")], columns=["tid", "attribute"])
    path = tmp_path / "c.txt"
    path.write_text("t1&t2&EQ(t1.x,t2.x)&IQ(t1.y,t2.y)")
    m = _build_model().setInput(df).setRowId("tid").setErrorCells(cells).setErrorDetectors([NullErrorDetector(), ConstraintErrorDetector(str(path))]) \
        .setRepairByRules(True).option("model.rule.max_domain_size", "1000")
    assert _rows(m.run()) == [[3, "y", None, "test-1"], [5, "y", None, "test-2"], [6, "y", None, None]]


def test_repair_by_nearest_values(oracle_backend):
    """test_repair_by_nearest_values (test_model.py:930-983)."""
    from repair.costs import Levenshtein
    df = pd.DataFrame([(1, "100%", 100, "a", 1.0), (3, "32%", 101, "b", 1.1), (4, "1xx%", 1, "a", 1.3), (5, "100x", 2, "b", 0.6), (6, "12x", 300, "a", 0.8)],
                      columns=["tid", "v0", "v1", "v2", "v3"])
    cells = pd.DataFrame([(4, "v0"), (5, "v0"), (6, "v0"), (3, "v1"), (5, "v1"), (6, "v1"), (5, "v2")], columns=["tid", "attribute"])

    def model():
        return _build_model().setInput(df).setRowId("tid").setErrorCells(cells).setRepairByRules(True) \
            .setUpdateCostFunction(Levenshtein(targets=["v0", "v1"])).option("model.rule.repair_by_nearest_values.disabled", "") \
            .option("model.rule.merge_threshold", "2.0")
    full = [[3, "v1", "101", "100"], [4, "v0", "1xx%", "100%"], [5, "v0", "100x", "100%"], [5, "v1", "2", "1"], [5, "v2", "b", "a"],
            [6, "v0", "12x", "32%"], [6, "v1", "300", "100"]]
    assert _rows(model().run()) == full
    assert _rows(model().setTargets(["v0", "v1"]).run()) == [r for r in full if r[1] != "v2"]


def test_output_schemas_of_the_probability_modes(adult):
    """test_compute_repair_candidate_prob / _prob / _score, test_maximal_likelihood_repair (test_model.py:1003-1093)."""
    from repair.costs import Levenshtein
    base = [[t, a, None] for t, a in ADULT_ERRORS]
    out = _build_model().setTableName("adult").setRowId("tid").option("repair.pmf.cost_weight", "0.1").option("repair.pmf.prob_threshold", "0.0") \
        .option("repair.pmf.prob_top_k", "80").run(compute_repair_candidate_prob=True)
    assert list(out.columns) == ["tid", "attribute", "current_value", "pmf"]
    assert sorted([int(t), a] for t, a in zip(out["tid"], out["attribute"])) == [r[:2] for r in base]
    assert all(isinstance(p, list) and set(p[0]) == {"class", "prob"} for p in out["pmf"])
    out = _build_model().setTableName("adult").setRowId("tid").run(compute_repair_prob=True)
    assert list(out.columns) == ["tid", "attribute", "current_value", "repaired", "prob"]
    assert _rows(out, ("tid", "attribute", "current_value")) == base
    out = _build_model().setTableName("adult").setRowId("tid").setUpdateCostFunction(Levenshtein()).setRepairDelta(1).run(compute_repair_score=True)
    assert list(out.columns) == ["tid", "attribute", "current_value", "repaired", "score"]
    assert _rows(out, ("tid", "attribute", "current_value")) == base
    out = _build_model().setTableName("adult").setRowId("tid").setUpdateCostFunction(Levenshtein()).setRepairDelta(3).run(maximal_likelihood_repair=True)
    assert _rows(out) == [[3, "Sex", None, "Male"], [7, "Sex", None, "Male"], [12, "Sex", None, "Male"]]


def test_timeout_option_stops_the_search(adult):
    """test_timeout (test_model.py:1183-1195): a 3 s budget ends an otherwise unbounded hyper-parameter search."""
    import time
    t0 = time.time()
    out = _build_model().setTableName("adult").setRowId("tid").setErrorCells("adult_dirty").option("model.hp.max_evals", "10000000") \
        .option("model.hp.no_progress_loss", "100000").option("model.hp.timeout", "3").run()
    assert len(out) == 7 and time.time() - t0 < 120


MIXED = [(1, 0, 1.0, 1.0, "a"), (2, 1, 1.5, 1.5, "b"), (3, 0, 1.4, None, "b"), (4, 1, 1.3, 1.3, "b"), (5, 1, 1.2, 1.1, "b"), (6, 1, 1.1, 1.2, "b"),
         (7, 0, None, 1.4, "b"), (8, 1, 1.4, 1.0, "b"), (9, 0, 1.2, 1.1, "b"), (10, None, 1.3, 1.2, "b"), (11, 0, 1.0, 1.9, "b"), (12, 0, 1.9, 1.2, "b"),
         (13, 0, 1.2, 1.3, "b"), (14, 0, 1.8, 1.2, None), (15, 0, 1.3, 1.1, "b"), (16, 1, 1.3, 1.0, "b"), (17, 0, 1.3, 1.0, "b")]
MIXED_ERRORS = [[3, "v3", None], [7, "v2", None], [10, "v1", None], [14, "v4", None]]


@pytest.fixture
def mixed(oracle_backend):
    from repair.api import Delphi
    df = pd.DataFrame(MIXED, columns=["tid", "v1", "v2", "v3", "v4"]).astype({"v1": "Int64"})
    Delphi.register_table("mixed_input", df)
    return df


def test_invalid_running_modes(mixed, adult):
    """test_invalid_running_modes (test_model.py:231-266)."""
    from repair.costs import Levenshtein
    from repair.model import RepairModel
    m = RepairModel().setTableName("mixed_input").setRowId("tid").setRepairDelta(1).setUpdateCostFunction(Levenshtein())
    with pytest.raises(ValueError, match="Cannot enable the maximal likelihood repair mode when continous attributes found"):
        m.run(maximal_likelihood_repair=True)
    m = RepairModel().setTableName("adult").setRowId("tid").setRepairByRules(True).setUpdateCostFunction(Levenshtein()).setRepairDelta(3) \
        .option("model.rule.repair_by_nearest_values.disabled", "")
    msg = "Cannot repair data by nearest values when enabling `maximal_likelihood_repair`, `compute_repair_candidate_prob`, `compute_repair_prob`, or `compute_repair_score`"
    for kw in (dict(maximal_likelihood_repair=True), dict(compute_repair_candidate_prob=True), dict(compute_repair_prob=True), dict(compute_repair_score=True)):
        with pytest.raises(ValueError, match=msg):
            m.run(**kw)


def test_compute_repair_prob_for_continouos_values(mixed):
    """test_compute_repair_prob_for_continouos_values (test_model.py:1095-1118)."""
    from repair.costs import Levenshtein
    for f in (lambda m: m, lambda m: m.setUpdateCostFunction(Levenshtein())):
        m = f(_build_model().setTableName("mixed_input").setRowId("tid"))
        out = m.run(compute_repair_candidate_prob=True)
        assert list(out.columns) == ["tid", "attribute", "current_value", "pmf"]
        got = sorted([int(t), a] for t, a in zip(out["tid"], out["attribute"]))
        assert got == [r[:2] for r in MIXED_ERRORS]
        out = m.run(compute_repair_prob=True)
        assert list(out.columns) == ["tid", "attribute", "current_value", "repaired", "prob"]
        assert _rows(out, ("tid", "attribute", "current_value")) == MIXED_ERRORS


def test_training_data_rebalancing(mixed):
    """test_training_data_rebalancing (test_model.py:1197-1214)."""
    out = _build_model().setTableName("mixed_input").setRowId("tid").setTrainingDataRebalancingEnabled(True).run()
    assert _rows(out, ("tid", "attribute", "current_value")) == MIXED_ERRORS
    assert out["repaired"].notna().all()


def test_rule_based_model_and_poor_model():
    """test_rule_based_model, test_PoorModel (test_model.py:1148-1181)."""
    from repair.model import FunctionalDepModel, PoorModel
    model = FunctionalDepModel("x", {1: "test-1", 2: "test-1", 3: "test-2"})
    pdf = pd.DataFrame([[3], [1], [2], [4]], columns=["x"])
    assert model.classes_.tolist() == ["test-1", "test-2"]
    assert model.predict(pdf) == ["test-2", "test-1", "test-1", None]
    pmf = model.predict_proba(pdf)
    assert len(pmf) == 4 and pmf[0].tolist() == [0.0, 1.0] and pmf[1].tolist() == [1.0, 0.0] and pmf[2].tolist() == [1.0, 0.0] and pmf[3] is None
    for v in (None, "test"):
        model = PoorModel(v)
        assert model.classes_.tolist() == [v] and model.predict(pdf) == [v] * 4
        pmf = model.predict_proba(pdf)
        assert len(pmf) == 4 and all(p.tolist() == [1.0] for p in pmf)


def test_compute_weighted_probs_for_target_attributes(adult, tmp_path):
    """test_compute_weighted_probs_for_target_attributes (test_model.py:1017-1052): a huge cost weight on `Sex` drives the top
    candidate's probability to 1 there and leaves `Relationship` alone."""
    from repair.costs import Levenshtein
    from repair.errors import ConstraintErrorDetector
    path = tmp_path / "adult_constraints.txt"
    path.write_text(adult["constraints"])
    m = _build_model().setTableName("adult").setRowId("tid").setTargets(["Sex", "Relationship"]).setErrorDetectors([ConstraintErrorDetector(str(path))]) \
        .option("model.hp.max_evals", "40").option("model.hp.no_progress_loss", "20")

    def top(df):
        df = df.sort_values(["tid", "attribute"])
        return [(int(t), a, p[0]["class"], p[0]["prob"]) for t, a, p in zip(df["tid"], df["attribute"], df["pmf"])]
    base = top(m.run(compute_repair_candidate_prob=True))
    weighted = top(m.setUpdateCostFunction(Levenshtein(targets=["Sex"])).option("repair.pmf.cost_weight", "100000000.0").run(compute_repair_candidate_prob=True))
    assert [r[:2] for r in base] == [r[:2] for r in weighted] == [(4, "Relationship"), (4, "Sex"), (11, "Relationship"), (11, "Sex")]
    for r1, r2 in zip(base, weighted):
        if r1[1] == "Sex":
            assert r1[3] < 0.95 and r2[3] > 0.9999
        else:
            assert r1[3] < 0.95 and r2[3] < 0.95


def test_repair_updates_through_misc(adult):
    """test_repair_updates (test_model.py:985-1001): `RepairMisc().repair()` applied to the predicted updates gives adult_clean."""
    from repair.api import Delphi
    from repair.misc import RepairMisc
    updates = _build_model().setTableName("adult").setRowId("tid").run()
    Delphi.register_table("repair_updates_view", updates)
    out = RepairMisc().option("repair_updates", "repair_updates_view").option("table_name", "adult").option("row_id", "tid").repair()
    clean = frame(load_golden("adult")["clean"]).sort_values("tid").reset_index(drop=True)
    assert out.sort_values("tid").reset_index(drop=True).astype(str).values.tolist() == clean.astype(str).values.tolist()
    assert isinstance(Delphi.getOrCreate().misc, RepairMisc)
    with pytest.raises(ValueError, match="Table 'adult' must have 'tid', 'attribute', and 'repaired' columns"):
        RepairMisc().options({"repair_updates": "adult", "table_name": "adult", "row_id": "tid"}).repair()


def test_misc_flatten_inject_null_and_argument_checks():
    """test_misc.py:49-111 (test_argtype_check, test_flatten, test_splitInputTable_invalid_params, test_injectNull) and
    RepairMiscSuite.scala:49-64,100-122 (flattenTable, injectNullAt)."""
    from repair.api import Delphi
    from repair.misc import RepairMisc
    with pytest.raises(TypeError, match="`key` should be provided as str, got int"):
        RepairMisc().option(1, "value")
    with pytest.raises(TypeError, match="`value` should be provided as str, got int"):
        RepairMisc().option("key", 1)
    with pytest.raises(TypeError, match=r"`options` should be provided as dict\[str,str\], got int"):
        RepairMisc().options(1)
    with pytest.raises(TypeError, match=r"`options` should be provided as dict\[str,str\], got int in keys"):
        RepairMisc().options({"1": "v1", 2: "v2"})
    with pytest.raises(TypeError, match=r"`options` should be provided as dict\[str,str\], got float in values"):
        RepairMisc().options({"1": "v1", "2": 1.1})
    Delphi.register_table("tempView", pd.DataFrame([(1, "a"), (2, "b"), (3, "c")], columns=["tid", "v"]))
    out = RepairMisc().options({"table_name": "tempView", "row_id": "tid"}).flatten()
    assert out.sort_values("tid").values.tolist() == [[1, "v", "a"], [2, "v", "b"], [3, "v", "c"]]
    Delphi.register_table("t", pd.DataFrame({"tid": ["1", "2", "3", "4"], "v1": [100000, 200000, 300000, 400000], "v2": ["test-1", "test-2", "test-3", "test-4"]}))
    out = RepairMisc().options({"table_name": "t", "row_id": "tid"}).flatten()
    assert sorted(map(tuple, out.values.tolist())) == sorted([("1", "v1", "100000"), ("2", "v1", "200000"), ("3", "v1", "300000"), ("4", "v1", "400000"),
                                                              ("1", "v2", "test-1"), ("2", "v2", "test-2"), ("3", "v2", "test-3"), ("4", "v2", "test-4")])
    with pytest.raises(ValueError, match="Required options not found: table_name, row_id, k"):
        RepairMisc().splitInputTable()
    with pytest.raises(ValueError, match="Option 'k' must be an integer, but 'x' found"):
        RepairMisc().options({"table_name": "t", "row_id": "tid", "k": "x"}).splitInputTable()
    Delphi.register_table("tempView", pd.DataFrame([(1, "a", 1), (2, "b", 1), (3, "c", 1), (4, "d", 2)], columns=["tid", "v1", "v2"]))
    out = RepairMisc().options({"table_name": "tempView", "target_attr_list": "v1", "null_ratio": "1.0"}).injectNull()
    assert out["v1"].isna().all() and out["v2"].tolist() == [1, 1, 1, 2] and out["tid"].tolist() == [1, 2, 3, 4]
    out = RepairMisc().options({"table_name": "t", "target_attr_list": "v1", "null_ratio": "1.0"}).injectNull()
    assert out["v1"].isna().all() and out["v2"].tolist() == ["test-1", "test-2", "test-3", "test-4"]
    with pytest.raises(ValueError, match=r"Option 'null_ratio' must be a float in \(0.0, 1.0\], but '0.0' found"):
        RepairMisc().options({"table_name": "t", "target_attr_list": "v2", "null_ratio": "0.0"}).injectNull()
    with pytest.raises(ValueError, match="Columns 'non-existent' do not exist in 'default.t'"):
        RepairMisc().options({"db_name": "default", "table_name": "t", "target_attr_list": "non-existent", "null_ratio": "1.0"}).injectNull()
    big = pd.DataFrame({"tid": np.arange(20000), "x": np.arange(20000) % 7, "y": ["s"] * 20000})
    Delphi.register_table("big", big)
    out = RepairMisc().options({"table_name": "big", "target_attr_list": "x,y", "null_ratio": "0.1"}).injectNull()
    assert 0.08 < out["x"].isna().mean() < 0.12 and 0.08 < out["y"].isna().mean() < 0.12 and not out["tid"].isna().any()


def test_escaped_column_names_every_mode(oracle_backend):
    """test_escaped_column_names (test_model.py:687-735): column names with blanks through every run mode."""
    from repair.costs import Levenshtein
    rows = [(1, "1", None, 1.0), (2, None, "test-2", 2.0), (3, "1", "test-1", 1.0), (4, "2", "test-2", 2.0), (5, "2", "test-2", 1.0), (6, "1", "test-1", 1.0)]
    df = pd.DataFrame(rows, columns=["t i d", "x x", "y y", "z z"])
    m = _build_model().setInput(df).setRowId("t i d").setDiscreteThreshold(10)
    out = m.run().sort_values(["t i d", "attribute"])
    assert out.values.tolist() == [[1, "y y", None, "test-1"], [2, "x x", None, "2"]]
    for kw in (dict(compute_repair_candidate_prob=True), dict(compute_repair_prob=True)):
        got = m.run(**kw).sort_values(["t i d", "attribute"])
        assert got[["t i d", "attribute"]].values.tolist() == [[1, "y y"], [2, "x x"]]
    rep = m.run(repair_data=True)
    rep = rep[rep["t i d"].isin([1, 2])].sort_values("t i d")
    assert rep.values.tolist() == [[1, "1", "test-1", 1.0], [2, "2", "test-2", 2.0]]
    m2 = _build_model().setInput(df[["t i d", "x x", "y y"]]).setRowId("t i d").setDiscreteThreshold(10).setUpdateCostFunction(Levenshtein()).setRepairDelta(3)
    got = m2.run(compute_repair_score=True).sort_values(["t i d", "attribute"])
    assert got[["t i d", "attribute"]].values.tolist() == [[1, "y y"], [2, "x x"]]
