"""The fixed-point grid at the BENCHMARKED row counts.

numerics v2.1 sizes the grid by the table: e = 62 - ceil_log2(bound * sum_w / w_max) leaves 38 bits for a value equal to the bound at
10M equally weighted rows and 35 bits at 100M -- a float32 gradient below 2^-14 (2^-11) of the bound is ROUNDED there, while every table
of tests/test_numerics_bound.py (<= 200 000 rows) has the full 2^50 grid.  This tool trains the oracle's `spec` mode under the test hook
RGBM_FX_E = 38 / 35 (the grid a 10M / 100M-row table gets, on a table of any size) and LightGBM's own arithmetic (`lightgbm_f32`: float32
g / h, double sums in row order) ONCE, and reports per E: first boosting iteration with a differing tree, arg-max disagreements, max |dp|.

    python tools/numerics_scale.py [--rows 200000] [--targets 10,7,0] [--E 38,35] [--iters 300] [--threads 8] [--out profiles/...json]
    python tools/numerics_scale.py --rows 10000000 --targets 0,4 --E 0 --iters 30      # the REAL grid of a 10M-row table (E = 0: no hook)
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
    ap.add_argument("--E", default="38,35")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dirty, _, cards = make_table(a.rows, a.cols, seed=a.seed, null_ratio=0.01)
    O.lib().orc_set_threads(a.threads)
    res = []
    for t in [int(x) for x in a.targets.split(",")]:
        feats = [c for c in range(a.cols) if c != t]
        tr, sc = np.flatnonzero(dirty[t] >= 0), np.flatnonzero(dirty[t] < 0)
        K = int(cards[t])
        kw = dict(NB.FIXED, n_estimators=a.iters, class_weight=NB.balanced(dirty[t][tr], K), objective=0 if K <= 2 else 1, num_class=max(K, 2))
        X, y, Xs = np.ascontiguousarray(dirty[feats][:, tr]), np.ascontiguousarray(dirty[t][tr]), np.ascontiguousarray(dirty[feats][:, sc])

        def fit(numerics, E):
            os.environ.pop("RGBM_FX_E", None)
            if E:
                os.environ["RGBM_FX_E"] = str(E)
            t0 = time.time()
            try:
                m = O.train(X, cards[feats], y, K, numerics=numerics, **kw)
            finally:
                os.environ.pop("RGBM_FX_E", None)
            return m.predict(Xs), m.save(), round(time.time() - t0, 1)

        pf, bf, sf = fit("lightgbm_f32", 0)
        row = dict(target=t, K=K, train_rows=int(len(tr)), cells=int(len(sc)), iterations=a.iters, f32_seconds=sf, grids={})
        top2 = np.sort(pf, axis=1)[:, -2:]
        row["min_top2_gap"] = float((top2[:, 1] - top2[:, 0]).min())
        for E in [int(x) for x in a.E.split(",")]:
            ps, bs, ss = fit("spec", E)
            d = NB._pair(ps, pf, bs, bf, False)
            d["spec_seconds"] = ss
            row["grids"]["E=%d" % E if E else "table's own grid"] = d
            print("c%d K=%d rows=%d %s: %r" % (t, K, len(tr), "E=%d" % E if E else "own grid", d), flush=True)
        res.append(row)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(dict(rows=a.rows, cols=a.cols, seed=a.seed, results=res), f, indent=1)
    O.lib().orc_set_threads(1)


if __name__ == "__main__":
    main()
