"""Full-size run of tests/numerics_bound.py (DESIGN.md section 3's table): the product's numerics (float32 g / h, exact integer sums)
against LightGBM's own arithmetic (double sums in row order), reference-fixed parameters, 300 iterations, on adult / hospital (discrete threshold 400: every attribute below 400
distinct values, the 100+-class ones included) / boston / a synthetic table with a K = 64 target.  CPU only (the oracle, both modes).

    python tools/numerics_bound.py [--rows 200000] [--threads 8] [--out profiles/r03_numerics_bound.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spark-data-repair-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from tests.helpers import frame, load_golden  # noqa: E402
from tests import numerics_bound as NB  # noqa: E402
from tests.synth import make_table  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200000)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    out = {}
    t0 = time.time()
    want = set(a.only.split(",")) if a.only else None

    def on(name):
        return want is None or name in want

    if on("adult"):
        g = load_golden("adult")
        out["adult"] = NB.frame_case(frame(g["input"]), "tid", ["Age", "Sex", "Income"], threads=a.threads)
    if on("hospital"):
        g = load_golden("hospital")
        df = frame(g["input"], dtypes=False); df["tid"] = df["tid"].astype(int)
        cells = frame(g["error_cells"], dtypes=False); cells["tid"] = cells["tid"].astype(int)
        targets = [c for c in df.columns if c != "tid" and 2 <= df[c].nunique() < 400 and c in set(cells["attribute"])]
        out["hospital"] = NB.frame_case(df, "tid", targets, error_cells=cells, threads=a.threads)
    if on("boston"):
        g = load_golden("boston")
        df = frame(g["input"])
        df["CHAS"] = df["CHAS"].astype("Int64").astype(str).where(df["CHAS"].notna(), None)
        df["RAD"] = df["RAD"].astype("Int64").astype(str).where(df["RAD"].notna(), None)
        out["boston"] = NB.frame_case(df, "tid", ["CRIM", "RAD", "TAX", "LSTAT", "CHAS"], numeric_targets=("CRIM", "TAX", "LSTAT"), threads=a.threads)
    if on("synthetic"):
        dirty, clean, cards = make_table(a.rows, 16, seed=42, null_ratio=0.01)
        res = []
        for t in (10, 0, 7):          # K = 64, the binary target, K = 24
            feats = [c for c in range(16) if c != t]
            r = NB.compare_target(dirty, cards, t, feats, np.flatnonzero(dirty[t] >= 0), np.flatnonzero(dirty[t] < 0), threads=a.threads)
            r["attribute"] = "c%d" % t
            res.append(r)
            print("synthetic", r, flush=True)
        out["synthetic_%d_rows" % a.rows] = res
    out["seconds"] = time.time() - t0
    for k, v in out.items():
        if isinstance(v, list):
            for r in v:
                head = "%-10s %-16s K=%-3d rows=%-6d cells=%-5d gap=%s" % (k[:10], r["attribute"], r["K"], r["train_rows"], r["cells"], ("%.2e" % r["min_top2_gap"]) if "min_top2_gap" in r else "-")
                print(head)
                for pair in ("spec_vs_f32", "f32_vs_f32_perm"):
                    if pair not in r:
                        continue
                    d = r[pair]
                    if "max_dp" in d:
                        print("    %-16s first differing iteration %-5s label mismatches %-3d max|dp| %.3e" % (pair, d["first_diff_iteration"], d["label_mismatch"], d["max_dp"]))
                    else:
                        print("    %-16s first differing iteration %-5s rounded mismatches %-3d max rel diff %.3e" % (pair, d["first_diff_iteration"], d["rounded_mismatch"], d["max_rel_diff"]))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
