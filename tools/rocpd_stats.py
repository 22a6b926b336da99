"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / avg / min / max (us)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>8s} {'total_us':>12s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for name, n, s, a, mn, mx in rows:
    print(f"{name[:70]:70s} {n:8d} {s / 1e3:12.1f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * s / tot:6.2f}")
print(f"TOTAL kernel time {tot / 1e3:.1f} us")
