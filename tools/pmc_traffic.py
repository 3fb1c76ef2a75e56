#!/usr/bin/env python
"""Turn the two rocprofv3 PMC databases written by tools/pmc_traffic.sh into profiles/pmc_traffic.json.

Units and corrections (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE are reported in
KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B read request for wide (16 B/lane) coalesced streams, i.e. exactly half
of the bytes -- both the raw and the doubled figure are stored, and which one applies is stated per kernel (the depth
streams are 16 B/lane; the fuse kernel's map reads are 4 B/lane read-modify-write, uncalibrated, so its raw value is
kept as the lower bound and the doubled one as the upper bound)."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = con.execute(f"select {name_col}, counter_name, value from counters_collection").fetchall()
    acc = {}
    for k, c, v in rows:
        acc.setdefault((k, c), []).append(float(v))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main(E, W, H, sync=False):
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    fetch = per_kernel(os.path.join(ROOT, "gpurun_out", "pmc_fetch", "pmc_results.db"))
    write = per_kernel(os.path.join(ROOT, "gpurun_out", "pmc_write", "pmc_results.db"))
    names = {"value_map_update_fused_kernel": "value_map_update_fused_kernel",
             "depth_ingest_kernel<false>": "depth_ingest_kernel", "depth_ingest_kernel<true>": "depth_ingest_scatter_kernel"}
    for frag, label in names.items():
        f = [(v, n) for (k, c), (v, n) in fetch.items() if frag in k and c == "FETCH_SIZE"]
        w = [(v, n) for (k, c), (v, n) in write.items() if frag in k and c == "WRITE_SIZE"]
        if not f or not w:
            continue
        fetch_b, write_b = f[0][0] * 1024.0, w[0][0] * 1024.0
        wide = label.startswith("depth")
        rec = {"fetch_size_bytes_raw": round(fetch_b), "write_size_bytes": round(write_b), "launches": f[0][1],
               "fetch_correction": "x2 (16 B/lane coalesced stream, gfx950 half-count)" if wide
               else "none applied (4 B/lane RMW: uncalibrated; x2 is the upper bound)",
               "bytes_per_launch": round((2 * fetch_b if wide else fetch_b) + write_b),
               "bytes_per_launch_upper": round(2 * fetch_b + write_b)}
        out[f"{label}@E={E},{W}x{H}" + (",sync" if sync else "")] = rec
        print(label, rec)
    try:   # which tree produced these counters (bench.py quotes it next to `roofline.traffic`)
        import subprocess

        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001 -- the GPU box has no .git: the caller passes VLFM_COMMIT
        head = os.environ.get("VLFM_COMMIT", "unknown")
    out["_commit"] = head
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if len(a) > 0 else 256, int(a[1]) if len(a) > 1 else 640, int(a[2]) if len(a) > 2 else 480,
         sync=len(a) > 3 and a[3] != "")
