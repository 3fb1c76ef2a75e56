"""Per-launch averages of the counters in a rocprofv3 --pmc database for the kernels whose name contains a fragment
(tools/pmc_sq.sh).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)."""
import glob
import os
import sqlite3
import sys


def main(d, frag):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        print("no database under", d)
        return
    con = sqlite3.connect(dbs[0])
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    acc = {}
    for k, c, v in con.execute(f"select {name_col}, counter_name, value from counters_collection"):
        if frag in k:
            acc.setdefault((k.split("(")[0][-60:], c), []).append(float(v))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:62s} {c:22s} {sum(v) / len(v):16.1f}  (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
