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


def _sorted_cells(df):
    return sorted(map(tuple, df.values.tolist()))


def test_null_detector_scala_golden_rows():
    """ErrorDetectorSuite.scala:50-71 ("NULL-based error detector"), host detector and code-space oracle."""
    df = pd.DataFrame({"tid": ["1", "2", "3", "4"], "v1": pd.array([100000, None, 300000, 400000], dtype="Int64"),
                       "v2": [3.0, 8.0, 1.0, None], "v3": ["test-1", "test-2", None, "test-4"]})
    cols = ["v1", "v2", "v3"]
    _, codes = _encode(df, cols)
    for targets, expected in ((["v1", "v2", "v3"], [("2", "v1"), ("3", "v3"), ("4", "v2")]), (["v1"], [("2", "v1")]),
                              (["v2", "v3"], [("3", "v3"), ("4", "v2")]), (["v3", "v1"], [("2", "v1"), ("3", "v3")]),
                              (["v3", "v2", "v5"], [("3", "v3"), ("4", "v2")])):
        host = NullErrorDetector().setUp("tid", df, ["v1", "v2"], targets).detect()
        assert _sorted_cells(host) == sorted(expected)
        rows, ccols = P.detect_nulls(codes, [cols.index(t) for t in targets if t in cols])
        assert sorted((df["tid"][r], cols[c]) for r, c in zip(rows, ccols)) == sorted(expected)


def test_regex_detector_scala_golden_rows():
    """ErrorDetectorSuite.scala:73-101 ("RegEx-based error detector"): `CAST(attr AS STRING) NOT RLIKE regex OR attr IS NULL`."""
    from repair.errors import RegExErrorDetector
    df = pd.DataFrame({"tid": ["1", "2", "3", "4"], "v1": [123, 123456, 123000, 987654321], "v2": [53.0, 123.0, 456.0, None],
                       "v3": ["123-abc", "456-efg", None, "123-hij"]})
    for targets, attr, regex, expected in (
            (["v1", "v2", "v3"], "v3", "123-hij", [("1", "v3"), ("2", "v3"), ("3", "v3")]),
            (["v1", "v2", "v3"], "v3", "123.*", [("2", "v3"), ("3", "v3")]),
            (["v1", "v2", "v3"], "v1", "123.*", [("4", "v1")]),
            (["v3"], "v3", "123.*", [("2", "v3"), ("3", "v3")]),
            (["v2", "v3"], "v2", "123.*", [("1", "v2"), ("3", "v2"), ("4", "v2")])):
        got = RegExErrorDetector(attr, regex).setUp("tid", df, ["v1", "v2"], targets).detect()
        assert _sorted_cells(got) == sorted(expected), (targets, attr, regex)


def test_constraint_detector_adult_golden_rows():
    """ErrorDetectorSuite.scala:188-201 ("Constraint-based error detector - adult"): tids 4 and 11, Sex and Relationship."""
    from tests.helpers import frame, load_golden
    g = load_golden("adult")
    df = frame(g["input"])
    det = ConstraintErrorDetector(constraints=g["constraints"].replace("\n", ";")).setUp("tid", df, [], ["Sex", "Relationship"])
    got = sorted((str(t), a) for t, a in det.detect().values.tolist())
    assert got == sorted([("4", "Relationship"), ("4", "Sex"), ("11", "Relationship"), ("11", "Sex")])


def test_outlier_detector_scala_golden_rows():
    """ErrorDetectorSuite.scala:219-233 ("Outlier-based error detector"): 1000 x 100.0 and one 0.0."""
    from repair.errors import GaussianOutlierErrorDetector
    df = pd.DataFrame({"tid": np.arange(1001), "value": [100.0] * 1000 + [0.0]})
    for approx in (False, True):
        for targets in (["value"], ["v", "value"]):
            got = GaussianOutlierErrorDetector(approx_enabled=approx).setUp("tid", df, ["value"], targets).detect()
            assert got.values.tolist() == [[1000, "value"]]


def test_constraint_parser_scala_goldens():
    """DenialConstraintsSuite.scala:27-93: valid syntax (with blanks), the invalid cases and their log line."""
    import logging
    from repair.errors import parse_and_verify_constraints, parse_constraint

    def render(p):
        if p.constant is not None:
            return {"EQ": "t1.%s <=> %s", "IQ": "NOT(t1.%s <=> %s)", "LT": "t1.%s < %s", "GT": "t1.%s > %s"}[p.op] % (p.left, p.constant)
        return {"EQ": "t1.%s <=> t2.%s", "IQ": "NOT(t1.%s <=> t2.%s)", "LT": "t1.%s < t2.%s", "GT": "t1.%s > t2.%s"}[p.op] % (p.left, p.right)

    for stmt, preds, refs in (
            ('t1&EQ(t1.v1,"abc")&EQ(t1.v2,"def")', {'t1.v1 <=> "abc"', 't1.v2 <=> "def"'}, {"v1", "v2"}),
            ("t1&t2&EQ(t1.v1,t2.v1)&IQ(t1.v2,t2.v2)", {"t1.v1 <=> t2.v1", "NOT(t1.v2 <=> t2.v2)"}, {"v1", "v2"}),
            ("t1&t2&LT(t1.v1,t2.v1)&GT(t1.v2,t2.v2)&EQ(t1.v1,t2.v1)", {"t1.v1 < t2.v1", "t1.v2 > t2.v2", "t1.v1 <=> t2.v1"}, {"v1", "v2"}),
            (' t1 & EQ ( t1.v1 , "abc") & EQ ( t1.v2 , "def" ) ', {'t1.v1 <=> "abc"', 't1.v2 <=> "def"'}, {"v1", "v2"}),
            ("t1 & t2 & EQ ( t1.v1 , t2.v1 ) & IQ ( t1.v2 , t2.v2 ) ", {"t1.v1 <=> t2.v1", "NOT(t1.v2 <=> t2.v2)"}, {"v1", "v2"})):
        ps = parse_constraint(stmt)
        assert {render(p) for p in ps} == preds, stmt
        assert {r for p in ps for r in p.references} == refs
    records = []
    handler = logging.Handler()
    handler.emit = records.append
    logging.getLogger("repair").addHandler(handler)
    try:
        for bad in ('EQ(t1.v1,"abc")', '1a&IQ(1a.v,"abc")', 'key&EQ(noexistent.v1,"abc")&EQ(key,"def")', 't1&1a&EQ(t1.v,"abc")&IQ(1a.v,"def")',
                    't1&EQ(t1.v1,"abc")&IL(t1.v1, "def")&EQ(t1.v2,"ghi")', 't1&t2&GT(t3.v0,"abc")&EQ(t1.v1,t2.v1)&IQ(t1.v2,t2.v2)',
                    't1&t2&GT(t3.v0,"abc")&EQ(t1.v1,t2.v1)&IL(t1.v2,t2.v2)', 't1&EQ(t1.v1,"abc")', "t1&", "t1", "a&b&", "k1&k2"):
            with pytest.raises(ValueError, match="Failed to parse an input string|Illegal predicates found|At least two predicate candidates should be given"):
                parse_constraint(bad)
            del records[:]
            assert parse_and_verify_constraints([bad], ["v1", "v2"]) == []
            assert sum(("Illegal constraint format found: %s" % bad) in r.getMessage() for r in records) == 1, bad
    finally:
        logging.getLogger("repair").removeHandler(handler)


def test_hospital_constraints_take_the_device_form():
    """DenialConstraintsSuite.scala:107-168 ("constraint parsing - hospital"): all 15 are EQ.. & IQ on two tuples, i.e. what
    rgbm_table_detect_constraint evaluates; the adult ones (single tuple, constants) are not."""
    from repair.errors import parse_and_verify_constraints
    from repair.pipeline import constraint_to_columns
    from tests.helpers import frame, load_golden
    g = load_golden("hospital")
    cols = [c for c in frame(g["input"], dtypes=False).columns if c != "tid"]
    plist = parse_and_verify_constraints([l for l in g["constraints"].splitlines() if l.strip()], cols)
    assert len(plist) == 15
    forms = [constraint_to_columns(ps, cols) for ps in plist]
    assert all(f is not None for f in forms)
    assert sorted(len(f[0]) for f in forms) == [1] * 13 + [2, 3]
    ga = load_golden("adult")
    acols = [c for c in frame(ga["input"]).columns if c != "tid"]
    aps = parse_and_verify_constraints(ga["constraints"].splitlines(), acols)
    assert len(aps) == 2 and all(constraint_to_columns(ps, acols) is None for ps in aps)
    # the device-form detector on the encoded hospital table == the pandas detector, constraint by constraint
    df = frame(g["input"], dtypes=False)
    _, codes = _encode(df, cols)
    from repair.errors import _violating_rows
    for ps, (eq, iq) in zip(plist, forms):
        assert np.array_equal(P.constraint_rows(codes, eq, iq), np.flatnonzero(_violating_rows(df, ps)))
