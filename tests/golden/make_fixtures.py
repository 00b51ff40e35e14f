#!/usr/bin/env python3
"""Generates tests/golden/*.json.gz from the reference's own test data and expected outputs.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_fixtures.py
Inputs  : testdata/{adult,hospital,boston,iris}.csv, testdata/hospital_constraints.txt,
          testdata/adult_constraints.txt, bin/testdata/{adult_repair,adult_clean,hospital_clean,
          hospital_error_cells,boston_clean,iris_clean}.csv
Inline goldens are transcribed from python/repair/tests/test_model.py (line numbers in each entry).
"""
import gzip
import json
import os

import pandas as pd

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def table(path, **kw):
    df = pd.read_csv(os.path.join(REF, path), dtype=kw.pop("dtype", None), **kw)
    return {"columns": list(df.columns), "dtypes": {c: str(t) for c, t in df.dtypes.items()},
            "rows": json.loads(df.to_json(orient="values"))}


def dump(name, obj):
    with gzip.open(os.path.join(OUT, name + ".json.gz"), "wt", encoding="utf-8") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".json.gz")), "bytes")


def main():
    dump("adult", {
        "source": "testdata/adult.csv, bin/testdata/adult_repair.csv, bin/testdata/adult_clean.csv, testdata/adult_constraints.txt",
        "input": table("testdata/adult.csv"),
        "expected_repair": table("bin/testdata/adult_repair.csv", keep_default_na=False, dtype=str),   # test_model.py:87-89
        "clean": table("bin/testdata/adult_clean.csv"),
        "constraints": open(os.path.join(REF, "testdata/adult_constraints.txt")).read(),
    })
    dump("hospital", {
        "source": "testdata/hospital.csv, testdata/hospital_constraints.txt, bin/testdata/hospital_clean.csv, bin/testdata/hospital_error_cells.csv",
        "input": table("testdata/hospital.csv", dtype=str),
        "clean": table("bin/testdata/hospital_clean.csv", dtype=str),
        "error_cells": table("bin/testdata/hospital_error_cells.csv", dtype=str),
        "constraints": open(os.path.join(REF, "testdata/hospital_constraints.txt")).read(),
    })
    dump("boston", {
        "source": "bin/testdata/boston.csv, testdata/boston.csv, bin/testdata/boston_clean.csv; schema python/repair/tests/test_model_perf.py:74-76",
        "input": table("bin/testdata/boston.csv"),             # what test_model_perf.py loads ($REPAIR_TESTDATA)
        "input_testdata": table("testdata/boston.csv"),        # BASELINE.json configs[4]
        "clean": table("bin/testdata/boston_clean.csv"),
    })
    dump("iris", {
        "source": "bin/testdata/iris.csv, bin/testdata/iris_clean.csv",
        "input": table("bin/testdata/iris.csv"),
        "clean": table("bin/testdata/iris_clean.csv"),
    })
    dump("inline_goldens", {
        "integer_input": {   # python/repair/tests/test_model.py:1121-1146
            "columns": ["tid", "v1", "v2", "v3", "v4"],
            "rows": [[1, 1, 1, 3, 0], [2, 2, None, 2, 1], [3, 3, 2, 2, 0], [4, 2, 2, 3, 1], [5, None, 1, 3, 0],
                     [6, 2, 2, 3, 0], [7, 3, 1, None, 0], [8, 2, 1, 2, 1], [9, 1, 1, 2, None]],
            "expected": [[2, "v2", None, "2"], [5, "v1", None, "2"], [7, "v3", None, "2"], [9, "v4", None, "1"]],
        },
        "escaped_column_names": {   # python/repair/tests/test_model.py:687-721
            "columns": ["t i d", "x x", "y y", "z z"],
            "rows": [[1, "1", None, 1.0], [2, None, "test-2", 2.0], [3, "1", "test-1", 1.0], [4, "2", "test-2", 2.0],
                     [5, "2", "test-2", 1.0], [6, "1", "test-1", 1.0]],
            "discrete_threshold": 10,
            "expected": [[1, "y y", None, "test-1"], [2, "x x", None, "2"]],
            "expected_repair_data_rows_1_2": [[1, "1", "test-1", 1.0], [2, "2", "test-2", 2.0]],
        },
        "error_cells_no_existent_attribute": {   # test_model.py:493-508: cells (5,Income),(16,Income) -> MoreThan50K
            "error_cells": [[1, "NoExistent"], [5, "Income"], [16, "Income"]],
            "expected": [[5, "Income", None, "MoreThan50K"], [16, "Income", None, "MoreThan50K"]],
        },
        "estimator_protocol": {   # test_model.py:1148-1181
            "poor_model": {"values": [None, "test"], "n": 4, "proba": [1.0]},
            "fd_model": {"x": "x", "fd_map": [[1, "test-1"], [2, "test-1"], [3, "test-2"]], "inputs": [3, 1, 2, 4],
                         "classes": ["test-1", "test-2"], "predict": ["test-2", "test-1", "test-1", None],
                         "proba": [[0.0, 1.0], [1.0, 0.0], [1.0, 0.0], None]},
        },
    })


if __name__ == "__main__":
    main()
