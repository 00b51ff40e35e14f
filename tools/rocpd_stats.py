"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals, and optionally the per-dispatch
sequence of one kernel.  Usage: python tools/rocpd_stats.py <results.db> [--seq SUBSTR] [--limit N]"""
import argparse, sqlite3
ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--seq", default=None); ap.add_argument("--limit", type=int, default=80)
a = ap.parse_args()
c = sqlite3.connect(a.db)
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
print("%-64s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for n, k, t, av, mn, mx in rows:
    print("%-64s %7d %12.3f %10.2f %10.2f %10.2f %6.2f" % (n[:64], k, t / 1e6, av / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
if a.seq:
    q = list(c.execute("select name, start, end, grid_x, grid_y, grid_z, lds_size from kernels where name like ? order by start", ("%" + a.seq + "%",)))
    print("\n# per-dispatch sequence of *%s* (first %d): duration_us grid lds" % (a.seq, a.limit))
    for n, s, e, gx, gy, gz, lds in q[:a.limit]:
        print("%10.2f  grid=(%d,%d,%d) lds=%d  %s" % ((e - s) / 1e3, gx, gy, gz, lds, n[:40]))
