"""Static instruction mix of a kernel's loops (gfx950 ISA from hipcc --save-temps): python tools/isa_loop_mix.py <substring of the mangled kernel name>"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "spark-data-repair-plugin_amd", "csrc", "rgbm.hip")
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-c", src, "-o", "x.o", "--save-temps"] + sys.argv[2:], cwd=d, capture_output=True)
lines = open(os.path.join(d, "rgbm-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
key = sys.argv[1]
starts = [i for i, l in enumerate(lines) if key in l and ": " in l and l.startswith("_Z")]
for st in starts:
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st:end]
    print(subprocess.run(["c++filt", lines[st].split(":")[0]], capture_output=True, text=True).stdout.strip()[:110], len(body), "lines")
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i, m.group(1)))
    seen = set()
    for ln, a, b, t in sorted(loops, reverse=True)[:8]:
        if t in seen: continue
        seen.add(t)
        c = collections.Counter()
        for l in body[a:b]:
            tt = l.strip().split(" ")[0]
            if re.match(r"(global|scratch|buffer|flat)_", tt): c["vmem"] += 1
            elif tt.startswith("ds_"): c["ds_atomic" if "add" in tt else "ds_rw"] += 1
            elif tt.startswith("v_"): c["valu"] += 1
            elif tt.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif tt.startswith("s_"): c["salu"] += 1
        print("  loop %-12s %5d lines %s" % (t, ln, dict(c)))
