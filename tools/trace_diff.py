"""Debug aid: first differences between two RGBM_TRACE dumps (rgbm.hip) of the same training call."""
import sys, numpy as np
a, b, tgt, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])


def load(d):
    idx = [l.split() for l in open("%s/target%d.idx" % (d, tgt))]
    return [(n, int(i), int(l), int(o), int(sz)) for n, i, l, o, sz in idx], np.memmap("%s/target%d.bin" % (d, tgt), np.uint8, "r")


ia, da = load(a); ib, db = load(b)
assert [r[:3] for r in ia] == [r[:3] for r in ib], "different record sequences"
shown = 0
for (n, it, lv, o, sz), (_, _, _, o2, _) in zip(ia, ib):
    x, y = da[o:o + sz], db[o2:o2 + sz]
    if np.array_equal(x, y):
        continue
    d = np.flatnonzero(x != y)
    msg = "%s it %d level %d: %d of %d bytes differ, first at byte %d" % (n, it, lv, len(d), sz, d[0])
    if n == "count":
        xi, yi = x.view(np.int32).reshape(K, 256), y.view(np.int32).reshape(K, 256)
        ks, ns = np.nonzero(xi != yi)
        msg += "; (class tree, node): a / b = " + ", ".join("(%d,%d): %d / %d" % (k, nn, xi[k, nn], yi[k, nn]) for k, nn in list(zip(ks, ns))[:8]) + " ... class trees %s" % sorted(set(ks.tolist()))[:70]
    elif n in ("plan", "snodes", "lpool"):
        per = sz // K
        ks = sorted(set((d // per).tolist()))
        msg += "; per class tree %d bytes; class trees %s; first: tree %d offset %d" % (per, ks[:70], d[0] // per, d[0] % per)
        if n == "snodes":
            per_node = per // 256
            msg += " = node %d field byte %d (node size %d)" % ((d[0] % per) // per_node, (d[0] % per) % per_node, per_node)
        if n == "lpool":
            w = per // 127
            msg += " = hist slot %d, bin %d, %s" % ((d[0] % per) // w, ((d[0] % per) % w) // 16, "g" if (d[0] % 16) < 8 else "h")
            xs, ys = x.view(np.int64), y.view(np.int64)
            j = d[0] // 8
            msg += " (a %d, b %d)" % (xs[j], ys[j])
    else:
        msg += "; a %d b %d" % (x.view(np.uint64)[0], y.view(np.uint64)[0])
    print(msg)
    shown += 1
    if shown >= 14:
        break
if not shown:
    print("traces identical")
