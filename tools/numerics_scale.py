"""The fixed-point grids at the BENCHMARKED row counts.

The grid of a class tree shrinks with the table: v2.1's worst-case floor, e = 62 - ceil_log2(bound * sum_w / w_max), leaves 38 bits below the
bound at 10M equally weighted rows and 35 at 100M; numerics v2.2 lifts it per class tree and iteration by the measured coarse sum of that
tree's gradient magnitudes (csrc/rgbm_numerics.h).  This tool trains the oracle's `spec` mode under the test hook RGBM_TEST_HOOKS=1 +
RGBM_FX_ROWS = R (the grids a table of R rows with THIS table's gradient distribution gets) and LightGBM's own arithmetic (`lightgbm_f32`:
float32 g / h, double sums in row order) ONCE, and reports per R: first boosting iteration with a differing tree, arg-max disagreements,
max |dp|.

    python tools/numerics_scale.py [--rows 200000] [--targets 10,7,0] [--grids 10000000,100000000] [--iters 300] [--threads 8] [--out profiles/...json]
    python tools/numerics_scale.py --rows 10000000 --targets 0,4 --grids 0 --iters 30      # the REAL grids of a 10M-row table (0: no hook)
    python tools/numerics_scale.py --hospital Score,Sample                                   # the skewed many-class attributes of the reference's hospital table
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spark-data-repair-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import oracle as O  # noqa: E402
from tests import numerics_bound as NB  # noqa: E402
from tests.synth import make_table  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200000)
    ap.add_argument("--cols", type=int, default=16)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--targets", default="10,7,0")
    ap.add_argument("--grids", default="10000000,100000000", help="row counts whose grids the spec mode is given (0 = the table's own)")
    ap.add_argument("--hospital", default="", help="comma-separated hospital attributes instead of the synthetic table")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    grids = [int(x) for x in a.grids.split(",")]

    def with_grid(R, fn):
        for k in ("RGBM_FX_ROWS", "RGBM_TEST_HOOKS"):
            os.environ.pop(k, None)
        if R:
            os.environ["RGBM_TEST_HOOKS"] = "1"; os.environ["RGBM_FX_ROWS"] = str(R)
        try:
            return fn()
        finally:
            for k in ("RGBM_FX_ROWS", "RGBM_TEST_HOOKS"):
                os.environ.pop(k, None)

    if a.hospital:
        from tests.helpers import frame, load_golden
        g = load_golden("hospital")
        df = frame(g["input"], dtypes=False); df["tid"] = df["tid"].astype(int)
        cells = frame(g["error_cells"], dtypes=False); cells["tid"] = cells["tid"].astype(int)
        res = []
        for R in grids:
            for r in with_grid(R, lambda: NB.frame_case(df, "tid", a.hospital.split(","), error_cells=cells, threads=a.threads, perm=False, n_estimators=a.iters)):
                d = r["spec_vs_f32"]
                print("hospital %s K=%d rows=%d cells=%d grid of %s rows: %r" % (r["attribute"], r["K"], r["train_rows"], r["cells"], R or "its own", d), flush=True)
                res.append(dict(attribute=r["attribute"], K=r["K"], grid_rows=R, **d))
        if a.out:
            with open(a.out, "w") as f:
                json.dump(dict(table="hospital", results=res), f, indent=1)
        return
    dirty, _, cards = make_table(a.rows, a.cols, seed=a.seed, null_ratio=0.01)
    O.lib().orc_set_threads(a.threads)
    res = []
    for t in [int(x) for x in a.targets.split(",")]:
        feats = [c for c in range(a.cols) if c != t]
        tr, sc = np.flatnonzero(dirty[t] >= 0), np.flatnonzero(dirty[t] < 0)
        K = int(cards[t])
        kw = dict(NB.FIXED, n_estimators=a.iters, class_weight=NB.balanced(dirty[t][tr], K), objective=0 if K <= 2 else 1, num_class=max(K, 2))
        X, y, Xs = np.ascontiguousarray(dirty[feats][:, tr]), np.ascontiguousarray(dirty[t][tr]), np.ascontiguousarray(dirty[feats][:, sc])

        def fit(numerics, R):
            t0 = time.time()
            m = with_grid(R, lambda: O.train(X, cards[feats], y, K, numerics=numerics, **kw))
            return m.predict(Xs), m.save(), round(time.time() - t0, 1)

        pf, bf, sf = fit("lightgbm_f32", 0)
        row = dict(target=t, K=K, train_rows=int(len(tr)), cells=int(len(sc)), iterations=a.iters, f32_seconds=sf, grids={})
        top2 = np.sort(pf, axis=1)[:, -2:]
        row["min_top2_gap"] = float((top2[:, 1] - top2[:, 0]).min())
        for R in grids:
            ps, bs, ss = fit("spec", R)
            d = NB._pair(ps, pf, bs, bf, False)
            d["spec_seconds"] = ss
            row["grids"]["grid of %d rows" % R if R else "table's own grid"] = d
            print("c%d K=%d rows=%d %s: %r" % (t, K, len(tr), "grid of %d rows" % R if R else "own grid", d), flush=True)
        res.append(row)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(dict(rows=a.rows, cols=a.cols, seed=a.seed, results=res), f, indent=1)
    O.lib().orc_set_threads(1)


if __name__ == "__main__":
    main()
