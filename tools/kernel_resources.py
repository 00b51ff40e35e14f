"""Register / scratch / occupancy table of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kernel_resources.py [name filter]      (cross-compiles rgbm.hip for gfx950; no GPU needed)"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "spark-data-repair-plugin_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-c", "rgbm.hip", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}; rows.append(cur); continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-70s %5s %5s %7s %7s %7s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "occ"))
for r in rows:
    if flt in r["name"]:
        print("%-70s %5d %5d %7d %7d %7d %4d" % (r["name"][:70], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1)))
