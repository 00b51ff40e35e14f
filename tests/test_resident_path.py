"""`RepairModel.run()` takes the HBM-resident pipeline (repair.pipeline) whenever the run is the plain per-attribute model loop
(missing 2 / 3): same result frame as the value-space path (pandas + one estimator per attribute), for discrete
and CONTINUOUS target attributes, with training-row sampling, for `run()` and `run(repair_data=True)`.

CPU: the job logic runs on the oracle engine (tests/helpers.OracleEngine) and the value-space path on the oracle estimator
backend, so both sides share the oracle's arithmetic and must agree cell for cell.  -m gpu: the same comparison with the HIP
engine against the HIP estimators, plus a 1M-row synthetic frame whose labels are checked against the oracle."""
import numpy as np
import pandas as pd
import pytest

from repair.errors import ConstraintErrorDetector, NullErrorDetector
from repair.model import RepairModel
from tests.helpers import OracleEngine, frame, load_golden
from tests.synth import make_table


def _synthetic_frame(n, cols, seed, null_ratio=0.02):
    dirty, clean, cards = make_table(n, cols, seed=seed, null_ratio=null_ratio)
    df = pd.DataFrame({"tid": np.arange(n)})
    for c in range(cols):
        v = np.array(["c%d_v%02d" % (c, k) for k in range(int(cards[c]))], object)[np.maximum(dirty[c], 0)]
        v[dirty[c] < 0] = None
        df["c%d" % c] = v
    return df, dirty, clean, cards


def _model(df, **opts):
    m = RepairModel().setInput(df).setRowId("tid").setErrorDetectors([NullErrorDetector()])
    for k, v in dict({"model.hp.max_evals": "1", "model.lgb.n_estimators": "12", "model.lgb.learning_rate": "0.2"}, **opts).items():
        m = m.option(k, str(v))
    return m


def _sorted(df):
    return df.sort_values(["tid", "attribute"]).reset_index(drop=True)


def _both_paths(df, engine, repair_data=False, **opts):
    slow = _model(df, **opts)
    slow._engine_override = None
    fast = _model(df, **opts)
    fast._engine_override = engine
    import os
    os.environ["REPAIR_RESIDENT"] = "0"
    try:
        a = slow.run(repair_data=repair_data)
    finally:
        os.environ.pop("REPAIR_RESIDENT", None)
    b = fast.run(repair_data=repair_data)
    assert getattr(fast, "_last_resident_info", None) is not None, "the run did not take the resident path"
    return a, b


def test_discrete_targets_equal_the_value_space_path(oracle_backend):
    df, _, _, _ = _synthetic_frame(3000, 6, seed=3)
    a, b = _both_paths(df, OracleEngine())
    assert len(a) > 100 and list(a.columns) == list(b.columns) == ["tid", "attribute", "current_value", "repaired"]
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))


def test_repair_data_equal(oracle_backend):
    df, _, _, _ = _synthetic_frame(2000, 5, seed=5)
    a, b = _both_paths(df, OracleEngine(), repair_data=True)
    pd.testing.assert_frame_equal(a.sort_values("tid").reset_index(drop=True), b.sort_values("tid").reset_index(drop=True))
    assert not b.drop(columns=["tid"]).isna().any().any()


def test_training_row_sampling_is_the_same_sample(oracle_backend):
    df, _, _, _ = _synthetic_frame(4000, 5, seed=7)
    a, b = _both_paths(df, OracleEngine(), **{"model.max_training_row_num": "1500"})
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))


def test_continuous_and_integral_targets_boston(oracle_backend):
    """configs[4]: CHAS / RAD are string attributes (classifiers), CRIM / LSTAT doubles and ZN / TAX ints (regressors + rounding,
    model.py:1130-1132)."""
    g = load_golden("boston")
    df = frame(g["input"])
    a, b = _both_paths(df, OracleEngine(), **{"model.lgb.n_estimators": "20"})
    assert set(a["attribute"]) >= {"CRIM", "RAD"} and len(a) > 20
    # Numeric FEATURES: the value-space path builds one dictionary per model (the distinct values of its training rows; a number it has
    # not seen goes to the nearest entry), the resident table one per column.  A dirty row whose CRIM occurs nowhere else is then a code
    # BETWEEN two training codes; the trainer puts its bin bounds at the midpoints of the values (rgbm_table_set_column_values), which
    # is the nearest-entry rule -- so both paths give the same numbers, not merely close ones.
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))


def test_runs_outside_the_plain_loop_keep_the_value_space_path(oracle_backend):
    df, _, _, _ = _synthetic_frame(600, 5, seed=9)
    m = _model(df)
    m._engine_override = OracleEngine()
    m.run(compute_repair_candidate_prob=True)
    assert getattr(m, "_last_resident_info", None) is None                  # pmf output: value-space path
    m2 = _model(df).setRepairByRules(True)
    m2._engine_override = OracleEngine()
    m2.run()
    assert getattr(m2, "_last_resident_info", None) is None                 # rule-based repairs: value-space path


def test_hyperparameter_search_on_resident_tables_equals_the_value_space_search(oracle_backend):
    """model.hp.max_evals > 1: the same points, folds and stopping rule, with every CV fit a pair of device row gathers + a table
    training call.  Same best point per target, hence the same repairs as the value-space search."""
    df, _, _, _ = _synthetic_frame(1500, 5, seed=19)
    opts = {"model.hp.max_evals": "4", "model.hp.no_progress_loss": "3", "model.lgb.n_estimators": "8"}
    a, b = _both_paths(df, OracleEngine(), **opts)
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))


def _given_cells_model(df, cells, **opts):
    m = RepairModel().setInput(df).setRowId("tid").setErrorCells(cells)
    for k, v in dict({"model.hp.max_evals": "1", "model.lgb.n_estimators": "12", "model.lgb.learning_rate": "0.2"}, **opts).items():
        m = m.option(k, str(v))
    return m


def _both_paths_given_cells(df, cells, engine, repair_data=False):
    import os
    slow = _given_cells_model(df, cells)
    os.environ["REPAIR_RESIDENT"] = "0"
    try:
        a = slow.run(repair_data=repair_data)
    finally:
        os.environ.pop("REPAIR_RESIDENT", None)
    fast = _given_cells_model(df, cells)
    fast._engine_override = engine
    b = fast.run(repair_data=repair_data)
    return a, b, fast


def test_error_cell_holding_the_only_occurrence_of_a_class_falls_back(oracle_backend):
    """`domain_stats` is counted before the error cells are NULLed.  A given error cell that holds the only occurrence of
    one of a target's two classes leaves one class: the reference short-cuts that attribute with PoorModel (model.py:1008-1017); the
    resident pipeline says NotResidentEligible before uploading anything and `_run` takes the value-space path."""
    df, _, _, _ = _synthetic_frame(400, 5, seed=23, null_ratio=0.0)
    df["c0"] = "a"
    df.loc[7, "c0"] = "b"                                            # the only "b"
    cells = pd.DataFrame({"tid": [7, 11, 12], "attribute": ["c0", "c1", "c2"]})
    a, b, fast = _both_paths_given_cells(df, cells, OracleEngine())
    assert getattr(fast, "_last_resident_info", None) is None        # fell back
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))
    assert set(b["attribute"]) <= {"c0", "c1", "c2"} and (b[b["attribute"] == "c0"]["repaired"] == "a").all()


def test_repair_data_with_null_cells_that_are_not_error_cells_falls_back(oracle_backend):
    """the repair UDF fills every NULL target cell of a dirty row (model.py:1128,1133), not only the listed error cells."""
    df, _, _, _ = _synthetic_frame(500, 5, seed=29, null_ratio=0.0)
    df.loc[20, "c1"] = None                                          # a NULL in a target column of a dirty row, NOT an error cell
    cells = pd.DataFrame({"tid": [20, 21, 30], "attribute": ["c2", "c1", "c2"]})
    a, b, fast = _both_paths_given_cells(df, cells, OracleEngine(), repair_data=True)
    assert getattr(fast, "_last_resident_info", None) is None
    pd.testing.assert_frame_equal(a.sort_values("tid").reset_index(drop=True), b.sort_values("tid").reset_index(drop=True))
    assert b.loc[b["tid"] == 20, "c1"].notna().all()                 # filled, as the reference's UDF does
    # without such cells the same call stays resident
    df2, _, _, _ = _synthetic_frame(500, 5, seed=29, null_ratio=0.0)
    a2, b2, fast2 = _both_paths_given_cells(df2, cells, OracleEngine(), repair_data=True)
    assert getattr(fast2, "_last_resident_info", None) is not None
    pd.testing.assert_frame_equal(a2.sort_values("tid").reset_index(drop=True), b2.sort_values("tid").reset_index(drop=True))


def test_unseen_category_in_a_dirty_row_is_missing_on_both_paths(oracle_backend):
    """a dirty row carries a category of feature c3 that no training row of target c1 shows.  The value-space path's per-model
    dictionary has no code for it (missing); the resident table marks categorical columns (`set_column_kind`) so the model treats it as
    missing too: same repairs, no fallback."""
    df, _, _, _ = _synthetic_frame(1500, 5, seed=31, null_ratio=0.0)
    df.loc[40, "c3"] = "zz_never_seen"                               # only this row holds it ...
    df.loc[40, "c1"] = None                                          # ... and its c1 is NULL: not a training row of the c1 model
    df.loc[[50, 60, 70], "c1"] = None
    a, b = _both_paths(df, OracleEngine())
    assert (a["tid"] == 40).any()
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))


def test_detection_runs_on_the_resident_table_too(oracle_backend):
    """with NULL / denial-constraint detectors `run()` never builds an `error_cells_df` in pandas: the frame is
    encoded once and detection, NULLing, training and repair all happen on the table (reference python/repair/errors.py:545-582 ->
    pipeline.detect_error_cells).  Same frames as the value-space path: NULL detector, NULL + constraint detectors, repair_data, and a
    clean table."""
    df, _, _, _ = _synthetic_frame(2500, 6, seed=41)
    a, b = _both_paths(df, OracleEngine())
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))
    # constraint c4 -> c5 on top of the NULL detector (c5 is a noisy function of the latent class, so many rows violate it)
    import os

    def model(engine):
        m = RepairModel().setInput(df).setRowId("tid").setErrorDetectors([NullErrorDetector(), ConstraintErrorDetector(constraints="c4->c5")])
        for k, v in {"model.hp.max_evals": "1", "model.lgb.n_estimators": "8", "model.lgb.learning_rate": "0.2"}.items():
            m = m.option(k, v)
        m._engine_override = engine
        return m
    os.environ["REPAIR_RESIDENT"] = "0"
    try:
        slow = model(None).run()
    finally:
        os.environ.pop("REPAIR_RESIDENT", None)
    fast_m = model(OracleEngine())
    fast = fast_m.run()
    pd.testing.assert_frame_equal(_sorted(slow), _sorted(fast))
    assert len(fast) > 500
    # (either the device detected, or a class that only error cells held sent the run back to the value-space path: same frame both ways)
    m2 = _model(df); m2._engine_override = OracleEngine()
    out = m2.run(repair_data=True)
    assert m2._last_detection_on_device and not out.drop(columns=["tid"]).isna().any().any() and len(out) == len(df)
    clean = df.dropna().reset_index(drop=True)
    m3 = _model(clean); m3._engine_override = OracleEngine()
    assert len(m3.run()) == 0 and m3._last_detection_on_device


@pytest.mark.gpu
def test_gpu_run_takes_the_resident_path_and_matches_both_references():
    """HIP engine vs the HIP estimators of the value-space path on 20 000 rows; then 1M rows through `run()`: every repaired label
    equals what the oracle's models give for the same encoded table, and the host-side (pandas / Arrow) share of the run is reported."""
    import time
    from oracle import oracle as O
    from repair.engine import HipEngine, balanced_class_weight
    eng = HipEngine(0)
    df, _, _, _ = _synthetic_frame(20000, 6, seed=11)
    a, b = _both_paths(df, eng)
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))

    n, cols = 1_000_000, 8
    df, dirty, clean, cards = _synthetic_frame(n, cols, seed=13, null_ratio=0.01)
    m = _model(df, **{"model.max_training_row_num": str(n), "model.lgb.n_estimators": "6"})
    t0 = time.perf_counter()
    out = m.run()
    wall = time.perf_counter() - t0
    info = m._last_resident_info
    assert info is not None
    # the oracle on the same codes: models per target on all non-NULL rows, then the chained repair of the dirty rows
    targets = list(range(cols))
    feats_l = [[c for c in range(cols) if c != t] for t in targets]
    models = []
    O.lib().orc_set_threads(16)
    try:
        for t in targets:
            r = dirty[t] >= 0
            K = int(cards[t])
            models.append(O.train(np.ascontiguousarray(dirty[feats_l[t]][:, r]), cards[feats_l[t]], dirty[t][r], K,
                                  class_weight=balanced_class_weight(np.bincount(dirty[t][r], minlength=K)), objective=0 if K == 2 else 1,
                                  num_class=max(K, 2), n_estimators=6, learning_rate=0.2, max_depth=7, num_leaves=31))
    finally:
        O.lib().orc_set_threads(1)
    mask = (dirty < 0).any(axis=0)
    sub = np.ascontiguousarray(dirty[:, mask])
    lab, _ = O.repair_chain(models, targets, feats_l, [list(range(int(cards[t]))) for t in targets], sub)
    pos = np.flatnonzero(mask)
    expect = {}
    for i, t in enumerate(targets):
        nul = np.flatnonzero(dirty[t][mask] < 0)
        for j in nul:
            expect[(int(pos[j]), "c%d" % t)] = "c%d_v%02d" % (t, lab[i][j])
    got = {(int(r.tid), r.attribute): r.repaired for r in out.itertuples(index=False)}
    assert got == expect                                       # NULL current values: every cell is kept in the result
    device = sum(info["times"].get(k, 0.0) for k in ("train", "infer", "detect", "prepare", "exchange", "gather"))
    print("resident run(): %.2fs wall, %.2fs on the device pipeline, host share %.0f %%" % (wall, device, 100 * (1 - device / wall)))


@pytest.mark.parametrize("kind", ["category", "arrow"])
def test_dictionary_encoded_columns_give_the_same_repairs(oracle_backend, kind):
    """Columns handed over as pandas Categorical or Arrow-backed strings (what a Parquet / Arrow reader produces) take the cheap
    encoding path -- their dictionary is factorised, not 10^6 Python objects -- and must repair exactly like object columns."""
    df, _, _, _ = _synthetic_frame(4000, 6, seed=23)
    fast = df.copy()
    for c in df.columns:
        if c == "tid":
            continue
        if kind == "category":
            fast[c] = pd.Categorical(df[c])
        else:
            pa = pytest.importorskip("pyarrow")
            fast[c] = df[c].astype(pd.ArrowDtype(pa.string()))
    ma, mb = _model(df), _model(fast)
    ma._engine_override = mb._engine_override = OracleEngine()
    a, b = ma.run(), mb.run()
    assert mb._last_resident_info is not None, "the run did not take the resident path"
    assert len(a) > 100
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b.astype({"current_value": object, "repaired": object})), check_dtype=False)


def test_unseen_category_in_a_dirty_row_is_missing_for_the_model(oracle_backend):
    """A dirty row with a feature value no training row of that target shows.  The value-space path treats it as missing (it is not in
    the model's dictionary).  The table-wide dictionary does hold it, so the column is marked CATEGORICAL (rgbm_table_set_column_kind)
    and the model records the codes its training rows never showed: same prediction on both paths, through the resident one."""
    df, _, _, _ = _synthetic_frame(1500, 5, seed=17)
    for k, i in enumerate(np.flatnonzero(df["c3"].isna().to_numpy())[:12]):
        df.loc[int(i), "c1"] = "only-here-%d" % (k % 3)
    a, b = _both_paths(df, OracleEngine())
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))
    # ... and the guard that callers without categorical marking can use still sees them
    from repair.pipeline import UnseenCategories, repair_frame
    with pytest.raises(UnseenCategories):
        repair_frame(OracleEngine(), df, "tid", targets=["c3"], base_params=dict(n_estimators=2), check_unseen=True)


def test_hospital_config1_through_the_resident_path(oracle_backend):
    """configs[1]: 1000 rows of typos.  Dirty rows carry categories their target's training rows never show (missing for that model:
    unseen-category bitmap), and erroneous values that only the error cells held must not count as classes (the dictionaries are built
    with the given cells already NULLed: num_class enters the softmax hessian factor).  177 repairs, identical on both paths."""
    import os
    import tests.test_quality as Q
    from repair.model import RepairModel
    os.environ["REPAIR_RESIDENT"] = "0"
    try:
        a, _ = Q._run_hospital()
    finally:
        os.environ.pop("REPAIR_RESIDENT", None)
    prev = RepairModel._resident_engine
    taken = []
    RepairModel._resident_engine = lambda self: (taken.append(1), OracleEngine())[1]
    try:
        b, _ = Q._run_hospital()
    finally:
        RepairModel._resident_engine = prev
    assert taken and len(a) == len(b) > 150
    pd.testing.assert_frame_equal(_sorted(a), _sorted(b))


def test_typo_values_held_by_error_cells_only_stay_on_the_resident_path(oracle_backend):
    """denial-constraint violations usually are values that occur once (a typo).  Detection runs on the device, so
    such a value is still in the target's dictionary when its cell is NULLed -- a class without rows.  The run used to throw the
    device work away and start over with the pandas detectors; now the detected cells are handed back, the dead values leave the
    dictionaries on the host side of the (cached) encoding, and the run goes on without a second detection.  Same frame as the
    value-space path."""
    import os
    df, _, _, _ = _synthetic_frame(3000, 6, seed=47, null_ratio=0.0)
    df["c5"] = ["g%02d" % (int(v[4:]) % 7) for v in df["c4"]]             # c4 -> c5 holds exactly ...
    for i, r in enumerate([5, 77, 901, 1500, 2222]):                      # ... until five cells get values nobody else has
        df.loc[r, "c5"] = "typo-%d" % i

    def model(engine):
        m = RepairModel().setInput(df).setRowId("tid").setErrorDetectors([ConstraintErrorDetector(constraints="c4->c5")])
        for k, v in {"model.hp.max_evals": "1", "model.lgb.n_estimators": "8", "model.lgb.learning_rate": "0.2"}.items():
            m = m.option(k, v)
        m._engine_override = engine
        return m
    os.environ["REPAIR_RESIDENT"] = "0"
    try:
        slow = model(None).run()
    finally:
        os.environ.pop("REPAIR_RESIDENT", None)
    fast_m = model(OracleEngine())
    fast = fast_m.run()
    assert fast_m._last_detection_on_device and getattr(fast_m, "_last_resident_info", None) is not None     # never left the device path
    pd.testing.assert_frame_equal(_sorted(slow), _sorted(fast))
    typos = fast[fast["current_value"].astype(str).str.startswith("typo-")]
    assert len(typos) == 5 and not typos["repaired"].astype(str).str.startswith("typo-").any()
