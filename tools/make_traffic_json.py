"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md prescribes) over

    python tools/probe.py --iters 1 --targets 0,1,...,15 --stats 0        (every target model of the 10M x 16 workload, one after another)

    python tools/make_traffic_json.py <fetch_dir> <write_dir> <tag> [--rows 10000000] [--cols 16]

Per kernel class of the histogram build -- root = rg::k_level_root, level = rg::k_level_mt -- the HBM-side bytes per launch, summed
over the SAME set of launches the bench line's algorithmic bytes describe (all 16 target models).  Both counters are calibrated on
rg::k_grad_mc of the same run, whose reads (8 B of score per (row, class tree) + 4 B of label per row) and writes (8 B of (g, h) per
(training row, class tree) + 1 B of node id per (row, class tree)) are known exactly: rocprofv3 reports KB, and on gfx950 tallies wide
coalesced reads at half their bytes (the guide's x2), which the calibration factor absorbs.  bench.py puts these next to the
algorithmic bytes it measures itself (`roofline.classes.<class>.traffic`)."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))


def load(dirname, counter):
    """{kernel class: [sum of counter over dispatches, dispatches]} in KB as rocprofv3 reports them."""
    out = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])] = per_dispatch.get((r["Dispatch_Id"], r["Kernel_Name"]), 0.0) + float(r["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            cls = "root" if "k_level_root" in name else "level" if "k_level_mt" in name else "grad_mc" if "k_grad_mc(" in name else None
            if cls:
                e = out.setdefault(cls, [0.0, 0])
                e[0] += v; e[1] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("tag")
    ap.add_argument("--rows", type=int, default=10_000_000); ap.add_argument("--cols", type=int, default=16)
    ap.add_argument("--reps", type=int, default=2, help="training calls per target in the profiled command (tools/probe.py runs every target twice)")
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--targets", default="", help="comma-separated target columns of the profiled command (default: every column)")
    a = ap.parse_args()
    from repair.synth import CARDS
    cards = [CARDS[c % len(CARDS)] for c in range(a.cols)]
    fetch, write = load(a.fetch_dir, "FETCH_SIZE"), load(a.write_dir, "WRITE_SIZE")
    # calibration on k_grad_mc: the multiclass targets with 16 <= K <= 112 use it, one launch per boosting iteration
    tcols = [int(x) for x in a.targets.split(",")] if a.targets else list(range(a.cols))
    ks = [cards[c] for c in tcols if 16 <= cards[c] <= 112]
    n_launch = len(ks) * a.reps * a.iters
    exp_read = sum(8.0 * k * a.rows + 4.0 * a.rows for k in ks) * a.reps * a.iters
    ns = (a.rows + 255) // 256 * 256          # node-id / (g, h) rows are padded to whole wave tiles (numerics v2.1 build); the padding is never written
    # (numerics v2.2: + the coarse gradient sums k_grad_mc leaves per 64-row workgroup, 16 B per class tree)
    exp_write = sum(8.0 * k * a.rows * 0.99 + 1.0 * k * a.rows + 16.0 * k * ((a.rows + 63) // 64) for k in ks) * a.reps * a.iters
    del ns
    assert fetch["grad_mc"][1] == n_launch == write["grad_mc"][1], ("k_grad_mc launches", fetch.get("grad_mc"), n_launch)
    f_cal = exp_read / (fetch["grad_mc"][0] * 1024.0)
    w_cal = exp_write / (write["grad_mc"][0] * 1024.0)
    out = {"source": "profiles/%s_hbm_traffic_pmc.txt: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over python tools/probe.py --iters %d "
                     "--targets %s --stats 0 (every target model of the %dM x %d workload, one after another); counters x calibration factor (FETCH %.3f, WRITE %.3f: known "
                     "reads / writes of rg::k_grad_mc in the same run)" % (a.tag, a.iters, a.targets or "0..%d" % (a.cols - 1), a.rows // 1_000_000, a.cols, f_cal, w_cal),
           "rows": a.rows, "cols": a.cols,
           "calibration": {"fetch_factor": f_cal, "write_factor": w_cal, "kernel": "rg::k_grad_mc", "launches": n_launch}, "classes": {}}
    for cls in ("root", "level"):
        fb, fl = fetch[cls][0] * 1024.0 * f_cal, fetch[cls][1]
        wb, wl = write[cls][0] * 1024.0 * w_cal, write[cls][1]
        assert fl == wl
        out["classes"][cls] = {"launches": fl, "fetch_bytes_per_launch": fb / fl, "write_bytes_per_launch": wb / wl, "bytes_per_launch": (fb + wb) / fl}
    tot_l = sum(c["launches"] for c in out["classes"].values())
    out["bytes_per_launch"] = sum(c["bytes_per_launch"] * c["launches"] for c in out["classes"].values()) / tot_l
    name = "%dm%d" % (a.rows // 1_000_000, a.cols)
    path = os.path.join(ROOT, "profiles", "traffic.json")
    allj = {}
    if os.path.exists(path):
        try:
            allj = json.load(open(path))
        except Exception:  # noqa: BLE001
            allj = {}
    allj = {k: v for k, v in allj.items() if isinstance(v, dict) and "classes" in v}   # drop entries of the old format
    allj[name] = out
    json.dump(allj, open(path, "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", "%s_hbm_traffic_pmc.txt" % a.tag), "w") as f:
        f.write("# %s\n" % out["source"])
        for label, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for cls, (v, n) in sorted(d.items()):
                f.write("%-10s %-8s launches=%-5d sum_KB=%.6g per_launch_KB=%.6g\n" % (label, cls, n, v, v / max(n, 1)))
        f.write(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
