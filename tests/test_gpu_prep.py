"""-m gpu: the device versions of the relational steps around the models (csrc/rgbm_prep.hip)
against the numpy oracle (oracle/prep.py) -- exact equality, ordered results included."""
import numpy as np
import pytest

from tests.synth import make_table

pytestmark = pytest.mark.gpu


def _table(n, cols, seed, null_ratio=0.02, cards=None):
    from repair import _native as N
    dirty, clean, cards = make_table(n, cols, seed=seed, null_ratio=null_ratio, cards=cards)
    return N.Table(dirty, cards), dirty, cards


@pytest.mark.parametrize("n", [1, 63, 64, 255, 4095, 4096, 4097, 12289, 100003])
def test_detect_nulls_sizes_around_the_block_edges(n):
    from oracle import prep as P
    tab, dirty, _ = _table(n, 5, seed=n, null_ratio=0.07)
    for cols in ([0, 1, 2, 3, 4], [3], [4, 0], [2, 2], []):
        rows, ccols = tab.detect_nulls(cols)
        er, ec = P.detect_nulls(dirty, cols)
        assert np.array_equal(rows, er) and np.array_equal(ccols, ec), "n=%d cols=%r" % (n, cols)


@pytest.mark.parametrize("ratio", [0.0, 1.0, 0.5])
def test_detect_nulls_dense_and_empty(ratio):
    from oracle import prep as P
    tab, dirty, _ = _table(20000, 3, seed=3, null_ratio=ratio)
    rows, ccols = tab.detect_nulls([0, 1, 2])
    er, ec = P.detect_nulls(dirty, [0, 1, 2])
    assert np.array_equal(rows, er) and np.array_equal(ccols, ec)
    assert len(rows) == int((dirty < 0).sum())


def test_constraint_detector_matches_oracle():
    from oracle import prep as P
    from repair import _native as N
    rng = np.random.default_rng(11)
    for trial in range(25):
        n = int(rng.integers(1, 30000))
        ncol = int(rng.integers(2, 7))
        cards = rng.choice([2, 3, 5, 17, 300, 5000], ncol).tolist()
        X = np.stack([rng.integers(0, c, n) for c in cards]).astype(np.int32)
        if trial % 3 == 0:     # a real dependency with a few violations: y = f(x0) except for noise
            X[ncol - 1] = (X[0] * 7 + 1) % cards[ncol - 1]
            noise = rng.random(n) < 0.01
            X[ncol - 1][noise] = rng.integers(0, cards[ncol - 1], int(noise.sum()))
        X[rng.integers(0, ncol)][rng.random(n) < 0.05] = -1
        tab = N.Table(X, cards)
        n_eq = int(rng.integers(0, ncol))
        eq = rng.choice(ncol - 1, size=min(n_eq, ncol - 1), replace=False).tolist()
        iq = ncol - 1
        want = P.constraint_rows(X, eq, iq)
        got = tab.detect_constraint(eq, iq)
        assert np.array_equal(got, want), "trial %d n=%d eq=%r cards=%r" % (trial, n, eq, cards)
        cc = [iq] + eq[:1]
        r2, c2 = tab.detect_constraint(eq, iq, cell_cols=cc)
        er, ec = P.constraint_cells(X, eq, iq, cc)
        assert np.array_equal(r2, er) and np.array_equal(c2, ec)


def test_constraint_detector_reference_golden_rows():
    """ErrorDetectorSuite.scala:118-160 in code space: v1 = [1,1,1,2,2,3,4,4], v2 = [t1,t1,NULL,t2,tX,t3,t4,t4]."""
    from repair import _native as N
    v1 = np.array([0, 0, 0, 1, 1, 2, 3, 3], np.int32)
    v2 = np.array([0, 0, -1, 1, 5, 2, 3, 3], np.int32)
    tab = N.Table(np.stack([v1, v2]), [4, 6])
    assert tab.detect_constraint([0], 1).tolist() == [0, 1, 2, 3, 4]
    rows, cols = tab.detect_constraint([0], 1, cell_cols=[0, 1])
    assert rows.tolist() == [0, 1, 2, 3, 4] * 2 and cols.tolist() == [0] * 5 + [1] * 5


def test_constraint_detector_rejects_bad_arguments():
    from repair import _native as N
    tab, dirty, cards = _table(100, 3, seed=1)
    with pytest.raises(N.RepairGbmError):
        tab.detect_constraint([0], -1)
    with pytest.raises(N.RepairGbmError):
        tab.detect_constraint([7], 1)
    big = N.Table(np.zeros((13, 4), np.int32), [2] * 13)
    with pytest.raises(N.RepairGbmError):
        big.detect_constraint(list(range(13)), 0)


def test_null_cells_dirty_rows_gather_and_counts():
    from oracle import prep as P
    from repair import _native as N
    rng = np.random.default_rng(17)
    n = 50021
    tab, dirty, cards = _table(n, 6, seed=17, null_ratio=0.0)
    m = 4000
    rows = rng.integers(-3, n + 3, m).astype(np.int64)           # a few cells name rows outside the table
    cols = rng.integers(-1, 7, m).astype(np.int32)               # ... or columns outside it
    targets = [1, 2, 4]
    tab.null_cells(rows, cols, targets)
    want = P.null_cells(dirty, rows, cols, targets)
    got = np.stack([tab.read_column(c) for c in range(6)])
    assert np.array_equal(got, want)
    # dirty rows = rows holding at least one error cell (whatever its attribute)
    dr = tab.rows_of_cells(rows)
    assert np.array_equal(dr, P.rows_of_cells(n, rows))
    assert len(tab.rows_of_cells(np.zeros(0, np.int64))) == 0
    # the dirty-row frame
    sub = tab.gather_rows(dr)
    assert (sub.n, sub.c) == (len(dr), 6) and np.array_equal(sub.n_codes, cards)
    assert np.array_equal(np.stack([sub.read_column(c) for c in range(6)]), want[:, dr])
    # rows per code (+ NULLs) of every column
    for c in range(6):
        cnt, nn = tab.count_codes(c)
        ec, en = P.count_codes(want, c, int(cards[c]))
        assert np.array_equal(cnt, ec) and nn == en
    with pytest.raises(N.RepairGbmError):
        tab.gather_rows(np.array([n], np.int64))


def test_count_codes_with_a_domain_larger_than_the_lds_histogram():
    from oracle import prep as P
    from repair import _native as N
    rng = np.random.default_rng(19)
    X = rng.integers(0, 20000, (1, 300000)).astype(np.int32)
    X[0][rng.random(300000) < 0.01] = -1
    tab = N.Table(X, [20000])
    cnt, nn = tab.count_codes(0)
    ec, en = P.count_codes(X, 0, 20000)
    assert np.array_equal(cnt, ec) and nn == en and cnt.sum() + nn == 300000


def test_encode_on_device_equals_oracle_and_pandas_encoder():
    import pandas as pd
    import pyarrow as pa
    from oracle import prep as P
    from repair import _native as N
    from repair.encode import TableEncoder
    rng = np.random.default_rng(23)
    n = 70001
    df = pd.DataFrame({"s": rng.choice(["b", "a", "zz", "c", None], n), "x": rng.choice([3.5, -1.0, 7.25, np.nan], n), "k": rng.integers(0, 900, n)})
    cols = ["s", "x", "k"]
    enc = TableEncoder(df, cols)
    idx, remaps = [], []
    for c in cols:
        arr = pa.array(df[c], from_pandas=True).dictionary_encode()
        idx.append(np.asarray(arr.indices.fill_null(-1), np.int32))
        vals = np.asarray(arr.dictionary.to_pylist(), dtype=object if c == "s" else np.float64)
        remaps.append(np.argsort(np.argsort(vals, kind="stable"), kind="stable").astype(np.int32))
    idx = np.stack(idx)
    tab = N.Table.from_dictionaries(idx, remaps)
    got = np.stack([tab.read_column(c) for c in range(3)])
    assert np.array_equal(got, P.encode_dictionaries(idx, remaps))
    assert np.array_equal(got, enc.encode(df))
    assert np.array_equal(tab.n_codes, enc.n_codes)


def test_millions_of_rows_properties():
    """Full-size properties that need no oracle: the cell list is ordered, complete and consistent with the counts;
    nulling the detected cells of a table and detecting again is idempotent; dirty rows are the distinct rows."""
    from repair import _native as N
    n = 4_000_000
    dirty, clean, cards = make_table(n, 8, seed=29, null_ratio=0.01)
    tab = N.Table(dirty, cards)
    rows, cols = tab.detect_nulls(list(range(8)))
    assert len(rows) == int((dirty < 0).sum())
    key = cols.astype(np.int64) * n + rows
    assert (np.diff(key) > 0).all()                                   # strictly ascending (column, row)
    assert (dirty[cols, rows] < 0).all()
    for c in range(8):
        cnt, nn = tab.count_codes(c)
        assert nn == int((cols == c).sum()) and cnt.sum() + nn == n
    dr = tab.rows_of_cells(rows)
    assert np.array_equal(dr, np.unique(rows))
    clean_tab = N.Table(clean, cards)
    clean_tab.null_cells(rows, cols, list(range(8)))
    r2, c2 = clean_tab.detect_nulls(list(range(8)))
    assert np.array_equal(r2, rows) and np.array_equal(c2, cols)
    # FD check at size: column 1 is a function of the latent cluster only up to 10 % noise, so x0 -> x1 is violated
    # by every row whose x0 group holds two x1 values; compare against a numpy group-by
    viol = tab.detect_constraint([0], 1)
    from oracle import prep as P
    assert np.array_equal(viol, P.constraint_rows(dirty, [0], 1))


@pytest.mark.parametrize("tgt,top_k,thres", [(5, 32, 0.0), (5, 3, 0.05), (0, 2, 0.0), (0, 1, 0.6), (7, 5, 0.0)])
def test_candidate_distributions_of_null_cells(tgt, top_k, thres):
    """rgbm_table_repair_pmf == oracle model's predict_proba on the NULL rows, sorted / filtered / sliced by the oracle."""
    from oracle import oracle as O
    from oracle import prep as P
    from repair import _native as N
    from tests.synth import balanced_weights
    dirty, clean, cards = make_table(9000, 8, seed=41, null_ratio=0.04)
    feats = [c for c in range(8) if c != tgt]
    K = int(cards[tgt])
    rows_tr = dirty[tgt] >= 0
    kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=12, learning_rate=0.2)
    cw = balanced_weights(dirty[tgt], K)
    tab = N.Table(dirty, cards)
    mg = tab.train(tgt, feats, class_weight=cw, **kw)
    mo = O.train(np.ascontiguousarray(dirty[feats][:, rows_tr]), cards[feats], dirty[tgt][rows_tr], K, class_weight=cw, **kw)
    assert mo.save() == mg.save()
    null_rows = np.flatnonzero(dirty[tgt] < 0)
    proba = mo.predict(np.ascontiguousarray(dirty[feats][:, null_rows]))
    want_cls, want_pr = P.top_k_pmf(proba, top_k, thres)
    cur = clean[tgt][null_rows].astype(np.int32)
    cur[::7] = -1
    rows, cls, pr, cp = tab.repair_pmf(mg, tgt, feats, top_k=top_k, threshold=thres, cur_codes=cur)
    assert np.array_equal(rows, null_rows)
    assert np.array_equal(cls, want_cls)
    assert np.array_equal(pr, want_pr)
    assert np.array_equal(cp, np.where(cur >= 0, proba[np.arange(len(cur)), np.maximum(cur, 0)], 0.0))


def test_candidate_distributions_edge_cases():
    from repair import _native as N
    dirty, clean, cards = make_table(3000, 5, seed=43, null_ratio=0.0)
    tab = N.Table(dirty, cards)
    m = tab.train(3, [0, 1, 2, 4], objective=1, num_class=int(cards[3]), n_estimators=3)
    rows, cls, pr = tab.repair_pmf(m, 3, [0, 1, 2, 4])                  # no NULL cell at all
    assert rows.shape == (0,) and cls.shape == (0, 32) and pr.shape == (0, 32)
    reg = N.train(dirty[:4], cards[:4], clean[4] % 5, 5, y_value=np.arange(5.0), objective=2, num_class=2, n_estimators=2)
    tab.null_cells([1, 2], [3, 3], [3])
    with pytest.raises(N.RepairGbmError):
        tab.repair_pmf(reg, 3, [0, 1, 2, 4])
    with pytest.raises(N.RepairGbmError):
        tab.repair_pmf(m, 3, [0, 1, 2])                                # wrong feature count
    rows, cls, pr = tab.repair_pmf(m, 3, [0, 1, 2, 4], top_k=40, threshold=0.0)     # top_k > num_class: padded
    K = int(cards[3])
    assert rows.tolist() == [1, 2] and (cls[:, K:] == -1).all() and (pr[:, K:] == 0).all() and (np.sort(cls[:, :K], axis=1) == np.arange(K)).all()
    assert (np.diff(pr[:, :K], axis=1) <= 0).all() and np.allclose(pr.sum(1), 1.0)


def test_concurrent_gathers_and_counts_on_one_table():
    """The folds of a resident hyper-parameter search gather from ONE training table on a thread pool (pipeline.search_on_table;
    ctypes drops the GIL).  The table's stream and scratch are shared, so the library serialises these calls per table:
    24 threads x 6 rounds of different row sets must each get exactly their rows."""
    from concurrent.futures import ThreadPoolExecutor
    tab, dirty, cards = _table(300000, 6, seed=21, null_ratio=0.03)
    rng = np.random.default_rng(5)
    jobs = [np.sort(rng.choice(300000, size=int(rng.integers(1, 200000)), replace=False)) for _ in range(48)]

    def run(i):
        rows = jobs[i]
        sub = tab.gather_rows(rows)
        cnt, n_null = tab.count_codes(i % 6)
        got = np.stack([sub.read_column(c) for c in range(6)])
        return i, got, cnt, n_null

    with ThreadPoolExecutor(24) as ex:
        for _ in range(3):
            for i, got, cnt, n_null in ex.map(run, range(48)):
                assert np.array_equal(got, dirty[:, jobs[i]]), "gather %d returned other rows" % i
                col = dirty[i % 6]
                assert n_null == int((col < 0).sum()) and np.array_equal(cnt, np.bincount(col[col >= 0], minlength=int(cards[i % 6])))
