"""profiles/traffic.json + profiles/<tag>_hbm_traffic_pmc.txt from the two rocprofv3 --pmc summaries of tools/probe.py
(tools/pmc_summary.py output: FETCH_SIZE and WRITE_SIZE passes, KB as rocprofv3 reports them).

    python tools/make_traffic_json.py gpurun_out/r02q r02q

FETCH_SIZE is doubled (gfx950: wide coalesced reads are tallied at half their bytes; calibrated on k_grad_mc, whose reads are
known exactly); WRITE_SIZE is taken as reported."""
import json, os, re, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K, F = 10_000_000, 64, 15


def per_dispatch(path, counter):
    root, levels = None, []
    for line in open(path):
        m = re.match(r"^(\d+) (.*) %s ([0-9.e+]+)$" % counter, line.strip())
        if not m:
            continue
        name, v = m.group(2), float(m.group(3)) * 1024.0
        is_root = "k_level_pass<true" in name
        if is_root:
            root = v
        elif "k_level_pass<false" in name:
            levels.append(v)
    return root, levels


fr, fl = per_dispatch(os.path.join(src, "pmc_fetch_summary.txt"), "FETCH_SIZE")
wr, wl = per_dispatch(os.path.join(src, "pmc_write_summary.txt"), "WRITE_SIZE")
fr, fl = fr * 2, [v * 2 for v in fl]
launches = 1 + len(fl)
out = {"10m16": {
    "source": "profiles/%s_hbm_traffic_pmc.txt: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over python tools/probe.py --iters 1 "
              "--targets 10 (the K=64 target of the 10M x 16 workload, one boosting iteration = 1 root + %d level launches, each covering the 64 class trees); FETCH_SIZE x2 "
              "(gfx950: wide coalesced reads are tallied at half their bytes; calibrated on k_grad_mc), WRITE_SIZE as reported (uncalibrated)" % (tag, len(fl)),
    "bytes_per_launch": (fr + wr + sum(fl) + sum(wl)) / launches,
    "root_pass": {"fetch_bytes": fr, "write_bytes": wr, "algorithmic_bytes": float(N) * K * (F + 8) * 0.99},
    "level_pass_stream": {"fetch_bytes_avg": sum(fl) / len(fl), "write_bytes_avg": sum(wl) / len(wl), "fetch_bytes_per_launch": fl,
                          "note": "gradient-only layout: a level pass streams node id (1 B) + g (4 B) of every row of every class tree = 3.2 GB whatever the share of built "
                                  "rows; the rest is the bin records the batches re-read past L2"}}}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", "%s_hbm_traffic_pmc.txt" % tag), "w") as f:
    f.write("# %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; KB as reported) -- python tools/probe.py --iters 1 --targets 10 --stats 0\n" % tag)
    f.write("# K=64 target of the 10M x 16 workload, split level pass (k_level_route + k_level_pass<STREAM, GONLY>), one boosting iteration\n")
    for name in ("pmc_fetch_summary.txt", "pmc_write_summary.txt"):
        for line in open(os.path.join(src, name)):
            if line.startswith("#") or not line.strip():
                continue
            if re.match(r"^\d+ ", line) or any(k in line for k in ("k_level_pass", "k_level_route", "k_grad_mc", "k_level_final")):
                f.write(line)
print(json.dumps(out, indent=1)[:1500])
