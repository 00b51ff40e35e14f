"""Summarise rocprofv3 --pmc output (rocpd sqlite or csv): per-kernel sums of each counter, in launch order when --seq.
Usage: python tools/pmc_summary.py <dir> [--seq SUBSTR]"""
import argparse, glob, os, sqlite3, csv, collections
ap = argparse.ArgumentParser(); ap.add_argument("dir"); ap.add_argument("--seq", default=None)
a = ap.parse_args()
dbs = glob.glob(os.path.join(a.dir, "**", "*.db"), recursive=True)
csvs = glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)
agg = collections.OrderedDict(); seq = []
if csvs:
    for f in csvs:
        for r in csv.DictReader(open(f)):
            n, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
            d = agg.setdefault(n, collections.OrderedDict()); d[c] = d.get(c, 0.0) + v
            d["#"] = d.get("#", 0) + 1
            if a.seq and a.seq in n: seq.append((int(r["Dispatch_Id"]), n, c, v))
elif dbs:
    for f in dbs:
        c = sqlite3.connect(f)
        tabs = [t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')")]
        print("# tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()][:10])
        try:
            q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by 1, 2 order by 1"
            for n, cn, v, k in c.execute(q):
                d = agg.setdefault(n, collections.OrderedDict()); d[cn] = v; d["#"] = k
            if a.seq:
                for did, n, cn, v in c.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection where kernel_name like ? group by 1, 3 order by 1", ("%" + a.seq + "%",)):
                    seq.append((did, n, cn, v))
        except Exception as e:   # noqa: BLE001
            print("query failed:", e)
for n, d in agg.items():
    print("%-70s %s" % (n[:70], "  ".join("%s=%.6g" % kv for kv in d.items())))
if seq:
    print("# per dispatch")
    for did, n, cn, v in sorted(seq): print(did, n[:50], cn, "%.6g" % v)
