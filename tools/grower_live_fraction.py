"""What the LEVEL GROWER routes and accumulates per level: from an RGBM_TRACE dump (csrc/rgbm.hip), the share of (row, class tree) pairs in nodes
it expands at each level (live) and in their smaller children (built).  Counterpart of tools/level_live_fraction.py (what the FINISHED trees need).
python tools/grower_live_fraction.py <trace dir> <target column> <class trees K>"""
import sys
import numpy as np
d, tgt, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
idx = [l.split() for l in open("%s/target%d.idx" % (d, tgt))]
data = np.memmap("%s/target%d.bin" % (d, tgt), np.uint8, "r")
snode = np.dtype([("Gq", "<i8"), ("Hq", "<i8"), ("pmin", "<f8"), ("count", "<i4"), ("depth", "<i4"), ("parent", "<i4"), ("is_left", "<i4"),
                  ("left", "<i4"), ("right", "<i4"), ("best_feature", "<i4"), ("searched", "<i4"), ("hslot", "<i4"), ("pad", "<i4"),
                  ("gain", "<f8"), ("theta", "<i4"), ("dleft", "<i4"), ("left_gq", "<i8"), ("left_hq", "<i8"), ("left_out", "<f8"), ("right_out", "<f8")])
last = {}
for n, it, lv, o, sz in idx:
    if n in ("snodes", "count", "plan"):
        last[(n, int(it))] = (int(lv), int(o), int(sz))            # the dump after the last level of the iteration comes last
live = np.zeros(9); built = np.zeros(9); used = np.zeros(9); n_it = 0; tot = 0
for (n, it), (lv, o, sz) in sorted(last.items()):
    if n != "snodes":
        continue
    assert sz == K * 256 * snode.itemsize, (sz, K, snode.itemsize)
    S = np.frombuffer(data[o:o + sz], snode).reshape(K, 256)
    _, oc, szc = last[("count", it)]
    C = np.frombuffer(data[oc:oc + szc], np.int32).reshape(K, 256)
    _, op, szp = last[("plan", it)]
    n_nodes = np.frombuffer(data[op:op + szp], np.int32).reshape(K, szp // K // 4)[:, 0]        # LvPlan::n_nodes
    n_it += 1
    for k in range(K):
        cnt = np.where(C[k] != 0, C[k], S[k]["count"]).astype(np.int64)      # child counts by node id; the root's is in its node record
        cnt[0] = int(S[k, 0]["count"])
        tot += int(cnt[0])
        for nd in range(int(n_nodes[k])):
            l, r = int(S[k, nd]["left"]), int(S[k, nd]["right"])
            if l < 0:
                continue
            dp = int(S[k, nd]["depth"])
            live[dp] += cnt[nd]; built[dp] += min(cnt[l], cnt[r])
print("target c%d, %d class trees, %d iterations traced" % (tgt, K, n_it))
print("  level pass      : " + " ".join("%6d" % (dd + 1) for dd in range(7)))
print("  live share      : " + " ".join("%6.3f" % (live[dd] / tot) for dd in range(7)))
print("  built share     : " + " ".join("%6.3f" % (built[dd] / tot) for dd in range(7)))
print("  sum over levels : live %.2f, built %.2f pairs per (row, class tree) and iteration" % (live[:7].sum() / tot, built[:7].sum() / tot))
