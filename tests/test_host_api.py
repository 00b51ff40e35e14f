"""CPU: the drop-in Python surface -- setter validation and error messages asserted by the reference's
own tests (python/repair/tests/test_model.py:98-266), option handling (utils.py:50-75), the
ErrorDetector plugin API and detectors (test_errors.py / ErrorDetectorSuite.scala goldens)."""
import re

import numpy as np
import pandas as pd
import pytest

from repair.api import Delphi
from repair.costs import Levenshtein, UserDefinedUpdateCostFunction
from repair.errors import (ConstraintErrorDetector, DomainValues, ErrorDetector, GaussianOutlierErrorDetector,
                           NullErrorDetector, RegExErrorDetector, parse_constraint)
from repair.model import RepairModel
from tests.helpers import frame, load_golden


def _adult():
    return frame(load_golden("adult")["input"])


def test_invalid_params_messages():
    with pytest.raises(ValueError, match="`setInput` and `setRowId` should be called before repairing"):
        RepairModel().run()
    with pytest.raises(ValueError, match="`setInput` and `setRowId` should be called before repairing"):
        RepairModel().setTableName("dummyTab").run()
    with pytest.raises(ValueError, match="Can not specify a database name when input is `DataFrame`"):
        RepairModel().setInput(_adult()).setDbName("default")
    with pytest.raises(ValueError, match="`setRepairDelta` should be called when enabling maximal likelihood repairing"):
        RepairModel().setInput("dummyTab").setRowId("dummyId").run(maximal_likelihood_repair=True)
    with pytest.raises(ValueError, match="`setUpdateCostFunction` should be called when enabling maximal likelihood repairing"):
        RepairModel().setInput("dummyTab").setRowId("dummyId").setRepairDelta(3).run(maximal_likelihood_repair=True)
    with pytest.raises(ValueError, match="`UpdateCostFunction.targets` cannot be used when enabling maximal likelihood repairing"):
        RepairModel().setInput("dummyTab").setRowId("dummyId").setRepairDelta(3) \
            .setUpdateCostFunction(Levenshtein(targets=["non-existent"])).run(maximal_likelihood_repair=True)
    with pytest.raises(ValueError, match="`table_name` should have at least character"):
        RepairModel().setTableName("")
    with pytest.raises(ValueError, match="`row_id` should have at least character"):
        RepairModel().setRowId("")
    with pytest.raises(ValueError, match="`attrs` should have at least one attribute"):
        RepairModel().setTargets([])
    with pytest.raises(ValueError, match="`thres` should be bigger than 1, got 0"):
        RepairModel().setDiscreteThreshold(0)
    with pytest.raises(ValueError, match="Repair delta should be positive, got -1"):
        RepairModel().setRepairDelta(-1)
    with pytest.raises(ValueError, match="`error_cells` should have at least character"):
        RepairModel().setRowId("tid").setErrorCells("")
    with pytest.raises(ValueError, match="`setRowId` should be called before specifying error cells"):
        RepairModel().setErrorCells(pd.DataFrame({"tid": [1], "attribute": ["a"]}))
    with pytest.raises(ValueError, match="Non-existent key specified: key=non.existent"):
        RepairModel().option("non.existent", "1")


def test_argtype_check_messages():
    cases = [
        (lambda: RepairModel().setDbName(1), "`db_name` should be provided as str, got int"),
        (lambda: RepairModel().setTableName(1), "`table_name` should be provided as str, got int"),
        (lambda: RepairModel().setInput(1), "`input` should be provided as str/DataFrame, got int"),
        (lambda: RepairModel().setTargets(1), "`attrs` should be provided as list[str], got int"),
        (lambda: RepairModel().setTargets(["a", 1]), "`attrs` should be provided as list[str], got int in elements"),
        (lambda: RepairModel().setErrorDetectors(1), "`detectors` should be provided as list[ErrorDetector], got int"),
        (lambda: RepairModel().setErrorDetectors([1]), "`detectors` should be provided as list[ErrorDetector], got int in elements"),
        (lambda: RepairModel().setDiscreteThreshold("a"), "`thres` should be provided as int, got str"),
        (lambda: RepairModel().setUpdateCostFunction(1), "`cf` should be provided as UpdateCostFunction, got int"),
        (lambda: RepairModel().setUpdateCostFunction([1]), "`cf` should be provided as UpdateCostFunction, got list"),
    ]
    for fn, msg in cases:
        with pytest.raises(TypeError, match=re.escape(msg)):
            fn()


def test_exclusive_run_flags_and_nearest_values_rule():
    api = RepairModel().setTableName("dummyTab").setRowId("dummyId")
    for kw in (dict(detect_errors_only=True, repair_data=True), dict(compute_repair_candidate_prob=True, compute_repair_prob=True),
               dict(compute_repair_score=True, repair_data=True)):
        with pytest.raises(ValueError, match="cannot be set to true simultaneously"):
            api.run(**kw)
    m = RepairModel().setTableName("dummyTab").setRowId("dummyId").setRepairByRules(True).setUpdateCostFunction(Levenshtein()) \
        .option("model.rule.repair_by_nearest_values.disabled", "")
    with pytest.raises(ValueError, match="Cannot repair data by nearest values when enabling"):
        m.run(compute_repair_prob=True)


def test_input_table_checks():
    df = _adult()
    with pytest.raises(ValueError, match="Uniqueness does not hold in column 'Sex'"):
        RepairModel().setInput(df).setRowId("Sex").run()
    with pytest.raises(ValueError, match="A least three columns"):
        RepairModel().setInput(df[["tid", "Sex"]]).setRowId("tid").run()
    with pytest.raises(ValueError, match="Target attributes not found"):
        RepairModel().setInput(df).setRowId("tid").setTargets(["nope"]).run()
    num = pd.DataFrame({"tid": [1, 2, 3], "a": [1.0, None, 2.0], "b": ["x", "y", "x"]})
    with pytest.raises(ValueError, match="Cannot enable the maximal likelihood repair mode when continous attributes found"):
        RepairModel().setInput(num).setRowId("tid").setRepairDelta(1).setUpdateCostFunction(Levenshtein()).run(maximal_likelihood_repair=True)


def test_option_values_validated_under_testing():
    from repair.utils import get_option_value
    assert get_option_value({}, "k", 3, int) == 3
    assert get_option_value({"k": "5"}, "k", 3, int) == 5
    with pytest.raises(ValueError, match='Failed to cast "x" into int data: key=k'):
        get_option_value({"k": "x"}, "k", 3, int)
    with pytest.raises(ValueError, match="`k` should be positive, got -1"):
        get_option_value({"k": "-1"}, "k", 3, int, lambda v: v > 0, "`{}` should be positive")
    keys = RepairModel.option_keys
    for k in ("model.lgb.n_estimators", "model.cv.n_splits", "model.hp.max_evals", "model.hp.no_progress_loss", "model.hp.timeout",
              "model.max_training_row_num", "model.max_training_column_num", "model.small_domain_threshold",
              "repair.pmf.prob_top_k", "error.domain_threshold_beta", "model.gpu.device_id"):
        assert k in keys


def test_detect_errors_only_null_detector():
    out = RepairModel().setInput(_adult()).setRowId("tid").setErrorDetectors([NullErrorDetector()]).run(detect_errors_only=True)
    got = sorted(map(tuple, out[["tid", "attribute"]].values.tolist()))
    assert got == [(3, "Sex"), (5, "Age"), (5, "Income"), (7, "Sex"), (12, "Age"), (12, "Sex"), (16, "Income")]
    assert out["current_value"].isna().all()


def test_regex_and_domain_detectors():
    df = _adult()
    d = RegExErrorDetector("Sex", "^(Male|Female)$").setUp("tid", df, [], ["Sex", "Age"])
    assert sorted(d.detect()["tid"].tolist()) == [3, 7, 12]          # NULLs never match
    d = RegExErrorDetector("Sex", "^Male$").setUp("tid", df, [], ["Sex"])
    assert len(d.detect()) == 3 + int((df["Sex"] == "Female").sum())
    d = DomainValues("Income", values=["LessThan50K"]).setUp("tid", df, [], ["Income"])
    assert len(d.detect()) == int((df["Income"] != "LessThan50K").sum())
    assert len(RegExErrorDetector("Sex", "x").setUp("tid", df, [], ["Age"]).detect()) == 0   # not a target
    assert str(NullErrorDetector()) == "NullErrorDetector()"


def test_constraint_parser_and_detector():
    ps = parse_constraint("t1&t2&EQ(t1.a,t2.a)&IQ(t1.b,t2.b)")
    assert [(p.op, p.left, p.right) for p in ps] == [("EQ", "a", "a"), ("IQ", "b", "b")]
    ps = parse_constraint('t1&EQ(t1.Sex,"Female")&EQ(t1.Relationship,"Husband")')
    assert [(p.op, p.left, p.constant) for p in ps] == [("EQ", "Sex", '"Female"'), ("EQ", "Relationship", '"Husband"')]
    ps = parse_constraint("X->Y")
    assert [(p.op, p.left) for p in ps] == [("EQ", "X"), ("IQ", "Y")]
    with pytest.raises(ValueError, match="At least one of `constraint_path` or `constraints` should be specified"):
        ConstraintErrorDetector()
    df = pd.DataFrame({"tid": range(6), "k": ["a", "a", "b", "b", "c", None], "v": ["1", "2", "3", "3", "4", "5"]})
    det = ConstraintErrorDetector(constraints="t1&t2&EQ(t1.k,t2.k)&IQ(t1.v,t2.v)").setUp("tid", df, [], ["k", "v"])
    got = sorted(map(tuple, det.detect().values.tolist()))
    assert got == [(0, "k"), (0, "v"), (1, "k"), (1, "v")]
    g = load_golden("adult")
    det = ConstraintErrorDetector(constraints=g["constraints"].replace("\n", ";")).setUp("tid", frame(g["input"]), [], ["Sex", "Relationship"])
    out = det.detect()
    adult = frame(g["input"])
    bad = adult[((adult.Sex == "Female") & (adult.Relationship == "Husband")) | ((adult.Sex == "Male") & (adult.Relationship == "Wife"))]
    assert sorted(out["tid"].unique().tolist()) == sorted(bad["tid"].tolist())


def test_hospital_constraint_detector_recall():
    """ConstraintErrorDetector on hospital finds most ground-truth error cells of the constrained attributes."""
    g = load_golden("hospital")
    df = frame(g["input"], dtypes=False)
    df["tid"] = df["tid"].astype(int)
    det = ConstraintErrorDetector(constraints=";".join(l for l in g["constraints"].splitlines() if l.strip()))
    cols = [c for c in df.columns if c != "tid"]
    out = det.setUp("tid", df, [], cols).detect()
    truth = frame(g["error_cells"], dtypes=False); truth["tid"] = truth["tid"].astype(int)
    found = set(map(tuple, out[["tid", "attribute"]].values.tolist()))
    constrained = set(out["attribute"])
    tset = set((r.tid, r.attribute) for r in truth.itertuples() if r.attribute in constrained)
    assert len(tset) > 300 and len(found & tset) / len(tset) > 0.9


def test_gaussian_outlier_detector():
    df = pd.DataFrame({"tid": range(12), "x": [1.0, 1.1, 0.9, 1.2, 1.0, 0.8, 1.1, 1.0, 0.95, 1.05, 50.0, None], "s": ["a"] * 12})
    d = GaussianOutlierErrorDetector().setUp("tid", df, ["x"], ["x", "s"])
    assert d.detect().values.tolist() == [[10, "x"]]


def test_cost_functions():
    assert Levenshtein().compute("kitten", "sitting") == 3.0
    assert Levenshtein().compute(None, "x") is None
    assert UserDefinedUpdateCostFunction(lambda x, y: float(abs(len(x) - len(y)))).compute("ab", "abcd") == 2.0
    with pytest.raises(ValueError, match="`f` should take two values and return a float cost value"):
        UserDefinedUpdateCostFunction(lambda x, y: 1)


def test_delphi_facade():
    assert Delphi.getOrCreate() is Delphi.getOrCreate()
    assert isinstance(Delphi.getOrCreate().repair, RepairModel)
    assert Delphi.getOrCreate().repair is not Delphi.getOrCreate().repair
    assert isinstance(Delphi.version(), str)


def test_custom_error_detector_plugin(oracle_backend):
    class EvenTidDetector(ErrorDetector):
        def _detect_impl(self):
            df = self._input()
            return self._cells(df["tid"] % 10 == 0, "Sex")
    out = RepairModel().setInput(_adult()).setRowId("tid").setErrorDetectors([EvenTidDetector()]).option("model.hp.max_evals", "1").run()
    assert set(out["attribute"]) <= {"Sex"} and set(out["tid"]) <= {0, 10}


def test_pmf_prob_and_score_outputs(oracle_backend):
    m = RepairModel().setInput(_adult()).setRowId("tid").setErrorDetectors([NullErrorDetector()]).option("model.hp.max_evals", "1")
    pmf = m.run(compute_repair_candidate_prob=True)
    assert list(pmf.columns) == ["tid", "attribute", "current_value", "pmf"] and len(pmf) == 7
    for p in pmf["pmf"]:
        probs = [e["prob"] for e in p]
        assert probs == sorted(probs, reverse=True) and abs(sum(probs) - 1.0) < 1e-9
    prob = m.run(compute_repair_prob=True)
    assert list(prob.columns) == ["tid", "attribute", "current_value", "repaired", "prob"]
    assert set(prob[prob.attribute == "Sex"]["repaired"]) == {"Male"}
    sc = m.setRepairDelta(3).setUpdateCostFunction(Levenshtein()).run(compute_repair_score=True)
    assert list(sc.columns) == ["tid", "attribute", "current_value", "repaired", "score"] and len(sc) == 7
    top = m.run(maximal_likelihood_repair=True)
    assert 3 <= len(top) <= 7 and list(top.columns) == ["tid", "attribute", "current_value", "repaired"]


def test_estimators_are_sklearn_cloneable_and_picklable(oracle_backend):
    import pickle
    from sklearn.base import clone
    from repair.gbm import RepairGBMClassifier, RepairGBMRegressor
    rng = np.random.default_rng(0)
    X = pd.DataFrame({"a": rng.choice(list("xyz"), 400), "b": rng.integers(0, 5, 400), "c": rng.normal(size=400).round(1)})
    y = pd.Series(np.where((X.a == "x") ^ (X.b > 2), "pos", "neg"))
    clf = RepairGBMClassifier(objective="binary", class_weight="balanced", learning_rate=0.2, n_estimators=20, min_child_samples=5, random_state=42)
    c2 = clone(clf).fit(X, y)
    assert c2.classes_.tolist() == ["neg", "pos"] and (c2.predict(X) == y).mean() > 0.9
    p = c2.predict_proba(X)
    assert p.shape == (400, 2) and np.allclose(p.sum(1), 1.0)
    c3 = pickle.loads(pickle.dumps(c2))
    assert np.array_equal(c3.predict_proba(X), p) and c3.feature_name_ == ["a", "b", "c"]
    reg = RepairGBMRegressor(objective="regression", learning_rate=0.2, n_estimators=30, min_child_samples=5).fit(X, X.b * 1.5 + (X.a == "y"))
    assert reg.score(X, X.b * 1.5 + (X.a == "y")) > 0.9
    with pytest.raises(TypeError):
        RepairGBMClassifier(not_a_param=1)
    with pytest.raises(ValueError, match="only boosting_type='gbdt'"):
        RepairGBMClassifier(boosting_type="dart").fit(X, y)


def test_build_model_contract(oracle_backend):
    from repair.train import build_model
    rng = np.random.default_rng(1)
    X = pd.DataFrame({"a": rng.choice(list("xyz"), 300), "b": rng.choice(list("pq"), 300)})
    y = pd.Series(np.where(X.a == "x", "A", np.where(X.b == "p", "B", "C")))
    (model, score), elapsed = build_model(X, y, True, 3, n_jobs=-1, opts={"model.hp.max_evals": "3", "model.lgb.n_estimators": "30",
                                                                          "model.lgb.learning_rate": "0.2", "model.hp.no_progress_loss": "2"})
    assert model is not None and elapsed >= 0 and 0.0 <= score <= 1.0
    assert (model.predict(X) == y).mean() > 0.9
    # any failure inside the build -> (None, 0.0)  (reference train.py:227-229)
    (model, score), _ = build_model(X, pd.Series([None] * 300), True, 3, n_jobs=-1, opts={"model.hp.max_evals": "1"})
    assert model is None and score == 0.0


def test_batched_hp_search_equals_sequential(oracle_backend):
    """model.hp.batch_size only changes HOW MANY CV fits are in flight (threads / HIP streams), never the outcome."""
    from repair.train import build_model
    rng = np.random.default_rng(7)
    X = pd.DataFrame({"a": rng.choice(list("xyz"), 400), "b": rng.choice(list("pqrs"), 400), "c": rng.integers(0, 5, 400)})
    y = pd.Series(np.where(X.a == "x", "A", np.where(X.b == "p", "B", "C")))
    base = {"model.hp.max_evals": "7", "model.lgb.n_estimators": "15", "model.lgb.learning_rate": "0.2", "model.hp.no_progress_loss": "3"}
    outs = []
    for bs in ("1", "3", "8"):
        (m, score), _ = build_model(X, y, True, 3, n_jobs=-1, opts=dict(base, **{"model.hp.batch_size": bs}))
        assert m is not None
        outs.append((score, m.booster_bytes_, m.get_params()["num_leaves"]))
    assert outs[0] == outs[1] == outs[2]


def test_repair_attrs_from_golden():
    """RepairMiscSuite.scala:124-144 ("repairAttrsFrom"): integral attributes are rounded, doubles parsed Java-style."""
    from repair.model import RepairModel
    base = pd.DataFrame({"tid": [1, 2, 3], "x": pd.array([None, None, 1], dtype="Int64"), "y": ["test-1", None, "test-2"], "z": [1.0, 2.0, None]})
    updates = pd.DataFrame({"tid": [1, 2, 2, 3, 9], "attribute": ["x", "x", "y", "z", "x"], "repaired": ["2.4", "2.6", "test-3", "3.1D", "7"]})
    m = RepairModel().setRowId("tid")
    out = m._repair_attrs(updates, base)
    assert out["x"].tolist() == [2, 3, 1] and out["y"].tolist() == ["test-1", "test-3", "test-2"] and out["z"].tolist() == [1.0, 2.0, 3.1]
    assert str(out["x"].dtype) == "Int64" and base["x"].isna().sum() == 2              # the input frame is left alone
    # half-up like Spark's round(), unparsable text becomes NULL
    upd = pd.DataFrame({"tid": [1, 2, 3], "attribute": ["x"] * 3, "repaired": ["2.5", "-2.5", "abc"]})
    assert m._repair_attrs(upd, base)["x"].tolist()[:2] == [3, -3] and m._repair_attrs(upd, base)["x"].isna().tolist() == [False, False, True]


def test_error_cells_of_unknown_rows_are_dropped(oracle_backend):
    """RepairApi.withCurrentValues is an inner join on the row id (RepairApi.scala:91-101): a caller-supplied
    error cell whose row the input does not hold vanishes instead of crashing the run."""
    df = _adult()
    cells = pd.DataFrame({"tid": [3, 12, 999], "attribute": ["Sex", "Age", "Sex"]})
    m = RepairModel().setInput(df).setRowId("tid").setErrorCells(cells).option("model.hp.max_evals", "1")
    det = m.run(detect_errors_only=True)
    assert sorted(zip(det["tid"], det["attribute"])) == [(3, "Sex"), (12, "Age")]
    out = m.run()
    assert set(out["tid"]) <= {3, 12}


def test_current_value_of_nullable_integers_is_integral():
    """CAST(int AS STRING) renders 2 as '2' even when the column holds NULLs (pandas would widen it to 2.0)."""
    from repair.errors import ErrorModel
    df = pd.DataFrame({"tid": [0, 1, 2, 3], "v": pd.array([2, None, 5, 2], dtype="Int64"), "w": ["a", "b", None, "a"]})
    cells = pd.DataFrame({"tid": [0, 1, 2], "attribute": ["v", "v", "w"]})
    em = ErrorModel("tid", [], 80, [], cells, {})
    got, _, _, _ = em.detect(df, ["v"])
    assert dict(zip(zip(got["tid"], got["attribute"]), got["current_value"])) == {(0, "v"): "2", (1, "v"): None, (2, "w"): None}


def test_gpu_device_option_reaches_the_estimator():
    from repair.train import fixed_params
    from repair.gbm import RepairGBMClassifier
    p = fixed_params({"model.gpu.device_id": "3"}, True, 4, -1)
    assert p["device_id"] == 3 and RepairGBMClassifier(**p).device_id == 3
    assert fixed_params({}, False, 0, -1)["device_id"] == 0
    with pytest.raises(ValueError, match="should be non-negative"):
        fixed_params({"model.gpu.device_id": "-1"}, True, 4, -1)


def test_column_code_cache_agrees_with_pandas():
    """`repair.utils.column_isna / column_nunique / column_factorize` (one shared hash pass per object column inside
    `column_code_cache`) must say what pandas says, for every kind of column `RepairModel.run()` can meet."""
    import numpy as np
    import pandas as pd
    from repair.utils import column_code_cache, column_factorize, column_isna, column_nunique
    rng = np.random.default_rng(5)
    n = 500
    df = pd.DataFrame({
        "s": rng.choice(np.array(["a", "bb", "ccc", None], object), n),
        "mixed_nan": rng.choice(np.array(["x", "y", np.nan, None], object), n),
        "f": np.where(rng.random(n) < 0.1, np.nan, rng.integers(0, 7, n).astype(float)),
        "i": rng.integers(0, 5, n),
        "nullable_int": pd.array(np.where(rng.random(n) < 0.1, None, rng.integers(0, 4, n)), dtype="Int64"),
        "b": rng.choice(np.array([True, False, None], object), n),
        "cat": pd.Categorical(rng.choice(np.array(["p", "q", None], object), n)),
        "all_null": np.array([None] * n, object),
    })
    for inside in (False, True):
        ctx = column_code_cache(df) if inside else None
        if ctx:
            ctx.__enter__()
        try:
            for c in df.columns:
                assert np.array_equal(column_isna(df, c), df[c].isna().to_numpy()), c
                assert column_nunique(df, c) == df[c].nunique(dropna=True), c
            for c in ("s", "mixed_nan", "b", "cat", "all_null"):
                codes, uniq = column_factorize(df, c)
                assert codes.dtype == np.int32 and np.array_equal(codes < 0, df[c].isna().to_numpy())
                back = np.array([None if k < 0 else uniq[k] for k in codes], object)
                want = df[c].astype(object).where(~df[c].isna(), None).to_numpy()
                assert all((a is None and b is None) or a == b for a, b in zip(back, want)), c
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
    # the cache belongs to ONE frame: another frame with the same columns is not served from it
    other = df.copy()
    other.loc[0, "s"] = None if df.loc[0, "s"] is not None else "a"
    with column_code_cache(df):
        column_isna(df, "s")
        assert np.array_equal(column_isna(other, "s"), other["s"].isna().to_numpy())
