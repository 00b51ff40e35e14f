"""repair.pipeline: detection -> null-out -> split -> models -> chained repair -> shaped cells on a code table.
CPU: the job logic on the oracle engine.  GPU (-m gpu): the HIP engine gives the very same cells, values, probabilities."""
import numpy as np
import pytest

from tests.synth import make_table

PARAMS = dict(n_estimators=12, learning_rate=0.2, num_leaves=31, max_depth=7)


def _job(engine, seed=51, n=6000, want_pmf=True, detect_nulls=True, with_cells=True, with_constraint=True):
    from repair.pipeline import repair_table
    dirty, clean, cards = make_table(n, 7, seed=seed, null_ratio=0.02, cards=[2, 3, 4, 6, 8, 64, 16])
    rng = np.random.default_rng(seed)
    noisy = dirty.copy()
    # a dependency c5 -> c6 with a few wrong (non-NULL) cells that only the constraint detector can see
    dep = (clean[5] * 3 + 1) % cards[6]
    noisy[6] = dep.astype(np.int32)                             # no NULLs here: a NULL is a value of its own for IQ and would flag its whole group
    bad = rng.choice(n, 6, replace=False)          # 6 of the 64 c5 groups become violating: all their rows are flagged
    noisy[6][bad] = (dep[bad] + 1) % cards[6]
    noisy[5] = clean[5]                                         # the determinant is clean
    cells = (rng.integers(0, n, 25), rng.choice([1, 2, 9], 25)) if with_cells else None     # column 9 does not exist
    table = engine.upload(noisy, cards)
    res = repair_table(engine, table, [1, 2, 4, 6], PARAMS, constraints=[([5], 6)] if with_constraint else (), detect_nulls=detect_nulls,
                       error_cells=cells, want_pmf=want_pmf, top_k=4, threshold=0.01)
    return res, noisy, clean, dep, bad


def _check_shape_of_result(res, noisy, n):
    rows, cols = res["rows"], res["cols"]
    key = cols.astype(np.int64) * n + rows
    assert (np.diff(key) > 0).all()                              # ordered by (column, row), no duplicates
    assert set(np.unique(cols).tolist()) <= {1, 2, 4, 6}
    assert np.array_equal(res["current"], noisy[cols, rows])
    assert np.array_equal(res["dirty_rows"], np.unique(rows))
    assert (res["repaired"] >= 0).all() and ((res["prob"] > 0) & (res["prob"] <= 1)).all()


def test_pipeline_on_the_oracle_engine():
    from tests.helpers import OracleEngine
    res, noisy, clean, dep, bad = _job(OracleEngine())
    n = noisy.shape[1]
    _check_shape_of_result(res, noisy, n)
    rows, cols = res["rows"], res["cols"]
    # every NULL cell of a target is an error cell; the constraint finds the planted violations (whole groups are flagged)
    for t in (1, 2, 4):
        assert set(np.flatnonzero(noisy[t] < 0)) <= set(rows[cols == t])
    assert set(bad) <= set(rows[cols == 6])
    # a violated group is flagged as a whole (its c5 and c6 cells; c5 is not a target), nothing else of column 6 is
    grp = np.isin(noisy[5], noisy[5][bad])
    assert np.array_equal(np.sort(rows[cols == 6]), np.flatnonzero(grp))
    # repairs of the NULL cells are mostly right (the columns follow a latent cluster)
    for t in (1, 2, 4):
        sel = cols == t
        assert (res["repaired"][sel] == clean[t][rows[sel]]).mean() > 0.7
    # candidate distributions: descending, above the threshold, first candidate = arg-max of the un-chained prediction
    pc, pp = res["pmf_class"], res["pmf_prob"]
    assert pc.shape == (len(rows), 4) and (np.diff(pp, axis=1) <= 0).all()
    assert ((pp > 0.01) == (pc >= 0)).all()
    single = np.isin(rows, np.flatnonzero((np.stack([noisy[t] < 0 for t in (1, 2, 4, 6)]).sum(0) + np.isin(np.arange(n), rows[cols == 6])) <= 1))
    same = pc[single, 0] == res["repaired"][single]
    assert same.mean() > 0.99                                    # rows with one error cell: chain == no chain
    known = res["current"] >= 0
    assert (res["current_prob"][~known] == 0).all() and (res["current_prob"][known] > 0).any()


def test_pipeline_without_any_error_cell_and_degenerate_target():
    from repair.pipeline import repair_table
    from tests.helpers import OracleEngine
    dirty, clean, cards = make_table(500, 4, seed=3, null_ratio=0.0)
    eng = OracleEngine()
    res = repair_table(eng, eng.upload(clean, cards), [1, 2], PARAMS)
    assert len(res["rows"]) == 0 and len(res["repaired"]) == 0 and len(res["dirty_rows"]) == 0
    const = clean.copy(); const[2] = 1; const[1][:5] = -1
    with pytest.raises(ValueError, match="fewer than two classes"):
        repair_table(eng, eng.upload(const, cards), [1, 2], PARAMS)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(want_pmf=False, with_cells=False), dict(detect_nulls=False, with_constraint=False), dict(seed=52, n=20011)])
def test_pipeline_hip_engine_equals_oracle_engine(kw):
    from repair.engine import HipEngine
    from tests.helpers import OracleEngine
    a = _job(HipEngine(), **kw)[0]
    b, noisy = _job(OracleEngine(), **kw)[:2]
    _check_shape_of_result(a, noisy, noisy.shape[1])
    for k in ("rows", "cols", "current", "repaired", "prob", "dirty_rows") + (("pmf_class", "pmf_prob", "current_prob") if "pmf_class" in b else ()):
        assert np.array_equal(a[k], b[k]), k
    assert a["models"] == b["models"]


def _people(n=1500, seed=7):
    import pandas as pd
    rng = np.random.default_rng(seed)
    city = rng.choice(["Lyon", "Paris", "Nice", "Lille", "Brest", "Metz"], n)
    zipc = np.array([{"Lyon": "69000", "Paris": "75000", "Nice": "06000", "Lille": "59000", "Brest": "29200", "Metz": "57000"}[c] for c in city], object)
    region = np.array([{"Lyon": "ARA", "Paris": "IDF", "Nice": "PACA", "Lille": "HDF", "Brest": "BRE", "Metz": "GES"}[c] for c in city], object)
    grade = np.where(np.isin(city, ["Lyon", "Paris"]), rng.choice([3, 4], n), rng.choice([1, 2], n))
    df = pd.DataFrame({"tid": np.arange(n) + 100, "city": city, "zip": zipc, "region": region, "grade": grade})
    truth = df.copy()
    for c, k in (("zip", 30), ("grade", 25)):             # no NULL in `region`: a NULL is a value of its own for IQ and would flag its whole city
        df.loc[rng.choice(n, k, replace=False), c] = None
    wrong = rng.choice(np.flatnonzero(df["region"].notna().to_numpy() & (city == "Metz")), 3, replace=False)
    df.loc[wrong, "region"] = "IDF"                      # violates city -> region; the whole Metz group gets flagged
    return df, truth, wrong


def test_repair_frame_value_space_round_trip():
    from repair.pipeline import encode_frame, repair_frame
    from repair.encode import TableEncoder
    from tests.helpers import OracleEngine
    df, truth, wrong = _people()
    cols = ["city", "zip", "region", "grade"]
    idx, remaps, dicts = encode_frame(df, cols)
    enc = TableEncoder(df, cols)
    from oracle import prep as P
    assert np.array_equal(P.encode_dictionaries(idx, remaps), enc.encode(df))            # same codes as the pandas encoder
    out = repair_frame(OracleEngine(), df, "tid", targets=["zip", "region"], constraints=[(["city"], "region")],
                       base_params=dict(n_estimators=15, learning_rate=0.3, min_data_in_leaf=5), want_pmf=True, top_k=3)
    assert list(out.columns) == ["tid", "attribute", "current_value", "repaired", "prob", "pmf", "current_prob"]
    by = {(int(r.tid), r.attribute): r for r in out.itertuples()}
    nulls = [(int(t), "zip") for t in df["tid"][df["zip"].isna()]]
    assert set(nulls) <= set(by)
    pos = {int(t): i for i, t in enumerate(df["tid"])}
    right = sum(by[k].repaired == truth[k[1]].iloc[pos[k[0]]] for k in nulls)
    assert right / len(nulls) > 0.9                                                       # zip / region follow the city
    for k in nulls:
        assert by[k].current_value is None and by[k].pmf[0]["prob"] >= by[k].pmf[-1]["prob"] and 1 <= len(by[k].pmf) <= 3
    # the Metz rows are flagged by the constraint and carry their current value (and its probability under the model)
    metz = [int(t) for t in df["tid"][(df["city"] == "Metz") & df["region"].notna()]]
    assert all((t, "region") in by for t in metz)
    assert sorted(by[(int(df["tid"].iloc[w]), "region")].current_value for w in wrong) == ["IDF"] * 3
    with pytest.raises(ValueError, match="Target attributes not found"):
        repair_frame(OracleEngine(), df, "tid", targets=["nope"])


@pytest.mark.gpu
def test_repair_frame_hip_engine_equals_oracle_engine():
    from repair.engine import HipEngine
    from repair.pipeline import repair_frame
    from tests.helpers import OracleEngine
    df, truth, wrong = _people(4000, seed=11)
    kw = dict(targets=["zip", "region", "grade"], constraints=[(["city"], "region")], base_params=dict(n_estimators=10, learning_rate=0.3), want_pmf=True, top_k=4)
    a = repair_frame(HipEngine(), df, "tid", **kw)
    b = repair_frame(OracleEngine(), df, "tid", **kw)
    assert a.equals(b)


def test_hospital_through_the_pipeline_on_the_oracle_engine():
    """BASELINE configs[1] (hospital.csv + its 15 denial constraints) through repair.pipeline: the constraints all take the
    device form and detection equals the pandas ConstraintErrorDetector cell for cell; with the ground-truth error cells
    handed over (as the reference's benchmark does) the repairs reach the precision floor of tests/test_quality.py."""
    from repair.errors import ConstraintErrorDetector, parse_and_verify_constraints
    from repair.pipeline import constraint_to_columns, repair_frame
    from tests.helpers import OracleEngine, frame, load_golden
    g = load_golden("hospital")
    df = frame(g["input"], dtypes=False)
    df["tid"] = df["tid"].astype(int)
    cols = [c for c in df.columns if c != "tid"]
    stmts = [l for l in g["constraints"].splitlines() if l.strip()]
    plist = parse_and_verify_constraints(stmts, cols)
    forms = [constraint_to_columns(ps, cols) for ps in plist]
    constraints = [([cols[i] for i in eq], cols[iq]) for eq, iq in forms]
    host = ConstraintErrorDetector(constraints=";".join(stmts)).setUp("tid", df, [], cols).detect()
    targets = sorted(set(host["attribute"]))
    out = repair_frame(OracleEngine(), df, "tid", targets=targets, constraints=constraints,
                       base_params=dict(n_estimators=40, learning_rate=0.1, min_data_in_leaf=5))
    got = set(map(tuple, out[["tid", "attribute"]].values.tolist()))
    want = set(map(tuple, host.values.tolist())) | set((int(t), a) for a in targets for t in df["tid"][df[a].isna()])
    assert got == want
    # the reference's own hospital benchmark hands the ground-truth error cells over (test_model_perf.py:296-309):
    # same call through the pipeline, same precision floor as tests/test_quality.py
    cells = frame(g["error_cells"], dtypes=False)
    cells["tid"] = cells["tid"].astype(int)
    from tests.test_quality import HOSPITAL_TARGETS
    out = repair_frame(OracleEngine(), df, "tid", targets=HOSPITAL_TARGETS, error_cells=cells[["tid", "attribute"]],
                       base_params=dict(n_estimators=300, learning_rate=0.01, min_data_in_leaf=20))
    clean = frame(g["clean"], dtypes=False)
    clean["tid"] = clean["tid"].astype(int)
    c = out.merge(clean, on=["tid", "attribute"], how="inner")
    assert len(c) > 150
    assert (c["repaired"] == c["correct_val"]).mean() > 0.9
