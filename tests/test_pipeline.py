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
