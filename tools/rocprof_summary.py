#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME -> DIR/NAME_results.db) into the
per-kernel summary CSV that is committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_bench_e8_kernel_stats.csv
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if len(name) > 120:
        name = name[:117] + "..."
    return name


def main(db: str, out: str) -> None:
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, total, avg, pct in rows:
            w.writerow([short(name), calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
    ours = [(n, c, a) for n, c, t, a, p in rows if "vlfm::" in n]
    print(f"{len(rows)} kernels -> {out}")
    for n, c, a in ours:
        print(f"  {short(n)[:70]:70s} calls={c:5d} avg={a:9.3f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
