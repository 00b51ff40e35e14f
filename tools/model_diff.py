"""Debug aid: where two dumps of `bench.py --dump-labels` differ: per target, the first boosting iteration / class tree whose bytes differ,
and what differs in it."""
import os, sys, hashlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.numerics_bound import parse_trees
a, b, nt = sys.argv[1], sys.argv[2], int(sys.argv[3])
for t in range(nt):
    ba, bb = open("%s_model_%d.bin" % (a, t), "rb").read(), open("%s_model_%d.bin" % (b, t), "rb").read()
    if ba == bb:
        continue
    K, n_it, ta = parse_trees(ba); _, _, tb = parse_trees(bb)
    print("target %d: K %d, %d iterations, blobs differ (%d vs %d bytes)" % (t, K, n_it, len(ba), len(bb)))
    shown = 0
    first_it = None
    for i, (x, y) in enumerate(zip(ta, tb)):
        if len(x["feat"]) != len(y["feat"]) or any(not np.array_equal(x[n], y[n]) for n in ("feat", "theta", "dleft", "left", "right", "gain", "leaf_value", "leaf_count")):
            first_it = i // K; break
    if first_it is not None:
        ks = []
        for k in range(K):
            x, y = ta[first_it * K + k], tb[first_it * K + k]
            if len(x["feat"]) != len(y["feat"]) or any(not np.array_equal(x[n], y[n]) for n in ("feat", "theta", "dleft", "left", "right", "gain", "leaf_value", "leaf_count")):
                ks.append(k)
        print("  first differing iteration %d: %d of %d class trees differ: %s" % (first_it, len(ks), K, ks[:70]))
        for k in ks[:3]:
            x, y = ta[first_it * K + k], tb[first_it * K + k]
            n = min(len(x["gain"]), len(y["gain"]))
            gd = np.flatnonzero(x["gain"][:n] != y["gain"][:n])
            print("   class %d: root gain a %.17g b %.17g; first node with another gain: %s; leaf counts a (sum %d) %s" % (
                k, x["gain"][0], y["gain"][0], ("%d (a %.17g, b %.17g; split a f%d<=%d b f%d<=%d)" % (gd[0], x["gain"][gd[0]], y["gain"][gd[0]], x["feat"][gd[0]], x["theta"][gd[0]], y["feat"][gd[0]], y["theta"][gd[0]])) if len(gd) else "none",
                int(x["leaf_count"].sum()), x["leaf_count"][:8].tolist()))
            print("            leaf counts b (sum %d) %s" % (int(y["leaf_count"].sum()), y["leaf_count"][:8].tolist()))
    for i, (x, y) in enumerate(zip(ta, tb)):
        same_struct = len(x["feat"]) == len(y["feat"]) and all(np.array_equal(x[n], y[n]) for n in ("feat", "theta", "dleft", "left", "right"))
        same_val = same_struct and np.array_equal(x["leaf_value"], y["leaf_value"])
        if same_val:
            continue
        it, k = divmod(i, K)
        if same_struct:
            d = np.abs(x["leaf_value"] - y["leaf_value"])
            print("  iteration %d class %d: same structure, %d of %d leaf values differ, max |d| %.3g" % (it, k, int((d > 0).sum()), len(d), d.max()))
        else:
            print("  iteration %d class %d: STRUCTURE differs: leaves %d vs %d" % (it, k, len(x["leaf_value"]), len(y["leaf_value"])))
            n = min(len(x["feat"]), len(y["feat"]))
            for j in range(n):
                if any(x[nm][j] != y[nm][j] for nm in ("feat", "theta", "dleft", "left", "right")):
                    print("    first differing node %d: a (f %d theta %d dl %d l %d r %d)  b (f %d theta %d dl %d l %d r %d)" % (
                        j, x["feat"][j], x["theta"][j], x["dleft"][j], x["left"][j], x["right"][j], y["feat"][j], y["theta"][j], y["dleft"][j], y["left"][j], y["right"][j]))
                    break
        shown += 1
        if shown >= 6:
            break
