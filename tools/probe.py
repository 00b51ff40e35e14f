"""Perf probe: train a few boosting iterations on a synthetic table through the resident-table path."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from repair import _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--cols", type=int, default=16)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--targets", type=str, default="0,4,10")
ap.add_argument("--stats", type=int, default=1)
ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--parallel", type=int, default=0, help="1: the chunked generator bench.py uses for the 100M-row table (make_table_parallel)")
ap.add_argument("--sort", type=int, default=0, help="1: cluster the rows on the host first (lexicographic by descending cardinality)")
a = ap.parse_args()
t0 = time.time()
if a.parallel:
    from repair.synth import make_table_parallel
    dirty, _, cards = make_table_parallel(a.rows, a.cols, seed=a.seed, threads=min(32, os.cpu_count() or 1))
else:
    dirty, clean, cards = make_table(a.rows, a.cols, seed=a.seed)
print("gen %.1fs" % (time.time() - t0), flush=True)
if a.sort:
    t0 = time.time()
    order = np.lexsort([dirty[c] for c in np.argsort(cards, kind="stable")])      # last key = primary = the largest cardinality
    dirty = np.ascontiguousarray(dirty[:, order])
    print("host sort %.1fs" % (time.time() - t0), flush=True)
t0 = time.time(); tab = N.Table(dirty, cards); print("upload %.2fs" % (time.time() - t0), flush=True)
for t in [int(x) for x in a.targets.split(",")]:
    feats = [c for c in range(a.cols) if c != t]
    K = int(cards[t]); cw = balanced_weights(dirty[t], K)
    for rep in range(2):
        t0 = time.time()
        res = tab.train(t, feats, class_weight=cw, want_stats=bool(a.stats), objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=a.iters)
        dt = time.time() - t0
        ktrees = 1 if K == 2 else K
        if not a.stats:
            print("target %d K=%d: %.3fs wall, %.1f ms/iter, %.2f ms/tree" % (t, K, dt, dt * 1e3 / a.iters, dt * 1e3 / a.iters / ktrees), flush=True)
            continue
        m, st = res
        print("target %d K=%d: %.3fs wall, %.1f ms/iter, %.2f ms/tree | hist %.1f ms (%d launches) root %.1f ms | rows %.3g bytes %.3g -> hist %.1f GB/s, root %.1f GB/s" % (
            t, K, dt, dt * 1e3 / a.iters, dt * 1e3 / a.iters / ktrees, st["hist_ms"], st["hist_launches"], st["root_ms"], st["hist_rows"], st["hist_bytes"],
            st["hist_bytes"] / max(st["hist_ms"], 1e-9) * 1e-6, st["root_rows"] * (len(feats) + 8) / max(st["root_ms"], 1e-9) * 1e-6), flush=True)
