"""Throughput of the relational steps around the models (csrc/rgbm_prep.hip) on the bench table shape
(10M rows x 16 columns, 1 % NULLs): wall-clock per call (includes the stream / scratch setup and the copy of the result)
next to the algorithmic bytes each step has to touch.  Kernel-level times come from running this under
`rocprofv3 --kernel-trace --stats` (profiles/r01j_prep_kernel_stats.csv).

    python tools/prep_bench.py [--rows 10000000] [--cols 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "spark-data-repair-plugin_amd")]

from repair import _native as N          # noqa: E402
from tests.synth import make_table       # noqa: E402


def timed(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=16)
    a = ap.parse_args()
    n, c = a.rows, a.cols
    dirty, clean, cards = make_table(n, c, seed=7, null_ratio=0.01)
    tab = N.Table(dirty, cards)
    res = {}
    cols = list(range(c))
    t, (rows, ccols) = timed(lambda: tab.detect_nulls(cols))
    res["detect_nulls"] = dict(ms=t * 1e3, cells=int(len(rows)), alg_bytes=n * c * 4, gbps=n * c * 4 / t / 1e9)
    t, viol = timed(lambda: tab.detect_constraint([0, 1], 2))
    res["detect_constraint(2 EQ low-cardinality, IQ)"] = dict(ms=t * 1e3, rows=int(len(viol)), alg_bytes=n * 3 * 4 * 2, gbps=n * 24 / t / 1e9)
    key = np.arange(n, dtype=np.int32)[None, :] // 3                 # ~3.3M groups of 3 rows: the hash table is really used
    tab2 = N.Table(np.concatenate([key, dirty[:1]]), [int(key.max()) + 1, int(cards[0])])
    t, viol2 = timed(lambda: tab2.detect_constraint([0], 1))
    res["detect_constraint(1 EQ, n/3 groups)"] = dict(ms=t * 1e3, rows=int(len(viol2)), alg_bytes=n * 2 * 4 * 2, gbps=n * 16 / t / 1e9)
    del tab2
    t, _ = timed(lambda: tab.null_cells(rows, ccols, cols))
    res["null_cells"] = dict(ms=t * 1e3, cells=int(len(rows)), alg_bytes=int(len(rows)) * 16)
    t, dr = timed(lambda: tab.rows_of_cells(rows))
    res["rows_of_cells"] = dict(ms=t * 1e3, rows=int(len(dr)), alg_bytes=int(len(rows)) * 8 + n * 2)
    t, sub = timed(lambda: tab.gather_rows(dr))
    res["gather_rows"] = dict(ms=t * 1e3, rows=int(len(dr)), alg_bytes=int(len(dr)) * c * 8 + int(len(dr)) * 8, gbps=len(dr) * c * 8 / t / 1e9)
    t, _ = timed(lambda: [tab.count_codes(j) for j in range(c)])
    res["count_codes(all columns)"] = dict(ms=t * 1e3, alg_bytes=n * c * 4, gbps=n * c * 4 / t / 1e9)
    # candidate distributions of one K=64 target's NULL cells with a small model (the scoring dominates)
    tgt = 10
    feats = [j for j in range(c) if j != tgt]
    m = tab.train(tgt, feats, objective=1, num_class=int(cards[tgt]), n_estimators=10, learning_rate=0.1)
    t, (prow, pcls, ppr) = timed(lambda: sub.repair_pmf(m, tgt, feats, top_k=32, threshold=0.0), reps=2)
    res["repair_pmf(K=%d, 10 iterations)" % int(cards[tgt])] = dict(ms=t * 1e3, cells=int(len(prow)))
    print(json.dumps(dict(rows=n, cols=c, steps=res), indent=1))


if __name__ == "__main__":
    main()
