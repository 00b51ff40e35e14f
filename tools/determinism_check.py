"""Trains the same models several times -- sequentially and with six training calls in flight, as engine.run_job does -- and compares the
serialised bytes: every repeat must give the same model (exact integer sums; nothing depends on scheduling)."""
import os, sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table, balanced_weights
from tests.numerics_bound import first_differing_iteration
from repair import _native as N
rows, iters, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dirty, clean, cards = make_table(rows, 16, seed=42)
tab = N.Table(dirty, cards)
targets = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "10,9,8,7,6,5,4,0").split(",")]


def fit(t):
    feats = [c for c in range(16) if c != t]
    K = int(cards[t]); cw = balanced_weights(dirty[t], K)
    return tab.train(t, feats, class_weight=cw, objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=iters).save()


ref = {}
for mode in ("sequential", "six in flight"):
    for rep in range(reps):
        t0 = time.time()
        if mode == "sequential":
            got = {t: fit(t) for t in targets}
        else:
            with ThreadPoolExecutor(6) as ex:
                got = dict(zip(targets, ex.map(fit, targets)))
        bad = []
        for t in targets:
            if t not in ref:
                ref[t] = got[t]
            elif got[t] != ref[t]:
                bad.append((t, first_differing_iteration(ref[t], got[t])[0]))
        print("%s repeat %d: %.1fs, differing models (target, first differing iteration): %s" % (mode, rep, time.time() - t0, bad or "none"), flush=True)
