"""CPU: the numpy oracle of the relational steps (oracle/prep.py) against the reference's own golden rows, and the
host (pandas) detectors against the same oracle.  The goldens are stated in value space exactly as the Scala suites
state them and are run through repair.encode to reach the code space the device works in."""
import numpy as np
import pandas as pd
import pytest

from oracle import prep as P
from repair.encode import TableEncoder
from repair.errors import ConstraintErrorDetector, NullErrorDetector
from tests.synth import make_table


def _encode(df, cols):
    enc = TableEncoder(df, cols)
    return enc, enc.encode(df)


def test_convert_error_cells_to_null_golden():
    """RepairSuite.scala:205-235 ("convertErrorCellsToNull"), both column-name variants."""
    for tid, c0, c1, c2 in (("tid", "c0", "c1", "c2"), ("t i d", "c 0", "c 1", "c 2")):
        df = pd.DataFrame({tid: [1, 2, 3, 4, 5], c0: [100, 200, 300, 400, 500], c1: ["abc", "def", "ghi", "jkl", "mno"],
                           c2: [1.2, 3.2, 2.1, 1.9, 0.5]})
        cols = [c0, c1, c2]
        enc, codes = _encode(df, cols)
        err = [(2, c1), (2, c2), (3, c0), (5, c2)]
        pos = {v: i for i, v in enumerate(df[tid])}
        rows = np.array([pos[r] for r, _ in err], np.int64)
        ccols = np.array([cols.index(a) for _, a in err], np.int32)
        out = P.null_cells(codes, rows, ccols, [0, 1, 2])
        got = [[enc.dicts[c].decode(out[j])[i] for j, c in enumerate(cols)] for i in range(5)]
        assert got == [[100.0, "abc", 1.2], [200.0, None, None], [None, "ghi", 2.1], [400.0, "jkl", 1.9], [500.0, "mno", None]]
        # only the listed target attributes are NULLed (RepairApi.scala:196-197)
        out = P.null_cells(codes, rows, ccols, [1])
        assert (out[0] >= 0).all() and (out[2] >= 0).all() and (out[1] < 0).tolist() == [False, True, False, False, False]
        # cells outside the table vanish in the join
        assert np.array_equal(P.null_cells(codes, [7, -1, 0], [0, 0, 9], [0, 1, 2]), codes)


def _constraint_table():
    """ErrorDetectorSuite.scala:118-131: v1 -> v2 with a NULL and a conflicting value."""
    return pd.DataFrame({"tid": ["1", "2", "3", "4", "5", "6", "7", "8"], "v1": [1, 1, 1, 2, 2, 3, 4, 4],
                         "v2": ["test-1", "test-1", None, "test-2", "test-X", "test-3", "test-4", "test-4"]})


@pytest.mark.parametrize("targets,expected", [
    (["v1", "v2"], [("1", "v1"), ("1", "v2"), ("2", "v1"), ("2", "v2"), ("3", "v1"), ("3", "v2"), ("4", "v1"), ("4", "v2"), ("5", "v1"), ("5", "v2")]),
    (["v1"], [("1", "v1"), ("2", "v1"), ("3", "v1"), ("4", "v1"), ("5", "v1")]),
    (["v2", "v1"], [("1", "v1"), ("1", "v2"), ("2", "v1"), ("2", "v2"), ("3", "v1"), ("3", "v2"), ("4", "v1"), ("4", "v2"), ("5", "v1"), ("5", "v2")]),
    (["v2", "v3"], [("1", "v2"), ("2", "v2"), ("3", "v2"), ("4", "v2"), ("5", "v2")]),
])
def test_constraint_detector_golden(targets, expected):
    """ErrorDetectorSuite.scala:140-186: EQ(t1.v1,t2.v1)&IQ(t1.v2,t2.v2); the NULL of row 3 is a value of its own."""
    df = _constraint_table()
    cols = ["v1", "v2"]
    _, codes = _encode(df, cols)
    attrs = [a for a in ["v1", "v2"] if a in targets]          # preds.flatMap(_.references).filter(targetAttrs.contains).distinct
    rows, ccols = P.constraint_cells(codes, [0], 1, [cols.index(a) for a in attrs])
    got = sorted((df["tid"][r], cols[c]) for r, c in zip(rows, ccols))
    assert got == sorted(expected)
    # the host (pandas) detector gives the same cells
    det = ConstraintErrorDetector(constraints="t1&t2&EQ(t1.v1,t2.v1)&IQ(t1.v2,t2.v2)").setUp("tid", df, [], targets)
    assert sorted(map(tuple, det.detect().values.tolist())) == sorted(expected)


def test_null_detector_golden():
    """ErrorDetectorSuite.scala "NULL-based error detector" shape: one (rowId, attribute) per NULL cell of a target."""
    df = pd.DataFrame({"tid": [1, 2, 3, 4], "a": ["x", None, "y", None], "b": [1.0, 2.0, None, 4.0], "c": ["p", "q", "r", "s"]})
    cols = ["a", "b", "c"]
    _, codes = _encode(df, cols)
    rows, ccols = P.detect_nulls(codes, [0, 1, 2])
    assert [(int(df["tid"][r]), cols[c]) for r, c in zip(rows, ccols)] == [(2, "a"), (4, "a"), (3, "b")]
    host = NullErrorDetector().setUp("tid", df, ["b"], cols).detect()
    assert sorted(map(tuple, host.values.tolist())) == sorted([(2, "a"), (4, "a"), (3, "b")])
    rows, ccols = P.detect_nulls(codes, [1])                      # only the targets are scanned
    assert rows.tolist() == [2] and ccols.tolist() == [1]
    assert len(P.detect_nulls(codes, [])[0]) == 0


def test_host_constraint_detector_equals_oracle_on_random_tables():
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = int(rng.integers(1, 400))
        dirty, clean, cards = make_table(n, 5, seed=100 + trial, null_ratio=0.1, cards=[3, 4, 2, 7, 5])
        df = pd.DataFrame({"tid": np.arange(n)})
        for c in range(5):
            df["c%d" % c] = pd.Series(np.where(dirty[c] < 0, None, dirty[c].astype(object)), dtype=object)
        eq = sorted(rng.choice(4, size=int(rng.integers(1, 3)), replace=False).tolist())
        stmt = "t1&t2&" + "&".join("EQ(t1.c%d,t2.c%d)" % (c, c) for c in eq) + "&IQ(t1.c4,t2.c4)"
        det = ConstraintErrorDetector(constraints=stmt).setUp("tid", df, [], ["c4"])
        host_rows = sorted(det.detect()["tid"].tolist())
        assert host_rows == P.constraint_rows(dirty, eq, 4).tolist()


def test_dirty_rows_and_flatten_and_pmf():
    n = 50
    cell_rows = np.array([7, 3, 7, 49, 3, 120, -4], np.int64)
    dirty = P.rows_of_cells(n, cell_rows)
    assert dirty.tolist() == [3, 7, 49]
    labels = np.array([[1, 2, 3], [4, 5, 6]], np.int32)          # [T=2][D=3]
    probs = labels / 10.0
    lab, pr = P.flatten_cells(labels, probs, [5, 8], dirty, np.array([7, 3, 49, 7, 10]), np.array([8, 5, 5, 2, 5]))
    assert lab.tolist() == [5, 1, 3, -1, -1]
    assert np.allclose(pr[:3], [0.5, 0.1, 0.3]) and np.isnan(pr[3:]).all()
    proba = np.array([[0.2, 0.5, 0.2, 0.1], [0.25, 0.25, 0.25, 0.25], [0.0, 1.0, 0.0, 0.0]])
    cls, p = P.top_k_pmf(proba, 3, 0.0)
    assert cls.tolist() == [[1, 0, 2], [0, 1, 2], [1, -1, -1]]     # ties keep class order; prob > threshold
    assert p[0].tolist() == [0.5, 0.2, 0.2] and p[2].tolist() == [1.0, 0.0, 0.0]
    cls, p = P.top_k_pmf(proba, 32 if False else 2, 0.2)
    assert cls.tolist() == [[1, -1], [0, 1], [1, -1]]


def test_encode_dictionaries_equals_table_encoder():
    """Arrow-style dictionary indices + sorted-rank remap == repair.encode (the pandas encoder the device step replaces)."""
    import pyarrow as pa
    rng = np.random.default_rng(9)
    df = pd.DataFrame({"s": rng.choice(["b", "a", "zz", "c", None], 300), "x": rng.choice([3.5, -1.0, 7.25, np.nan], 300),
                       "k": rng.integers(0, 9, 300)})
    cols = ["s", "x", "k"]
    enc, want = _encode(df, cols)
    idx, remaps = [], []
    for c in cols:
        arr = pa.array(df[c], from_pandas=True).dictionary_encode()
        idx.append(np.asarray(arr.indices.fill_null(-1), np.int32))
        vals = arr.dictionary.to_pylist()
        remaps.append(np.argsort(np.argsort(np.asarray(vals, dtype=object if c == "s" else np.float64), kind="stable"), kind="stable").astype(np.int32))
    got = P.encode_dictionaries(np.stack(idx), remaps)
    assert np.array_equal(got, want)
