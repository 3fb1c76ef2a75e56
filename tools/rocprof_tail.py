#!/usr/bin/env python
"""Per-kernel totals of the LAST `ms` milliseconds of a rocprofv3 kernel trace (rocpd database): the steady state of a run
whose warm-up (MIOpen's solver search, hipBLASLt heuristics) would otherwise dominate the summary.
    python tools/rocprof_tail.py DB.db 2000 [top]"""
import collections, sqlite3, sys
db, ms = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("no kernels view; tables:", tabs[:40]); sys.exit(1)
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
rows = con.execute(f"select name, start, end from {view}").fetchall()
t_end = max(r[2] for r in rows)
t0 = t_end - ms * 1e6
acc = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    if s >= t0:
        acc[n][0] += 1; acc[n][1] += (e - s) * 1e-3
tot = sum(v[1] for v in acc.values())
print(f"last {ms:.0f} ms: {tot / 1e3:.1f} ms of kernel time in {sum(v[0] for v in acc.values())} dispatches")
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / 1e3:9.2f} ms {100 * t / tot:5.1f} %  calls={c:6d}  {n[:110]}")
