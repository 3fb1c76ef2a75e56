#!/bin/bash
# tools/gemm_pmc.sh -- on the GPU box: SQ counters of the hand-written GEMM next to the library's (separate passes).
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  d=$REPO/gpurun_out/gemm_pmc_$i; rm -rf $d
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $d -o pmc -- python $REPO/tools/gemm_pmc_driver.py > $REPO/gpurun_out/gemm_pmc_$i.log 2>&1
  python - <<PY
import sqlite3, glob
for db in glob.glob("$d/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='view' or type='table'")]
    t = "counters_collection" if "counters_collection" in tabs else None
    if not t: print("no counters table", tabs[:20]); continue
    cols = [r[1] for r in con.execute(f"pragma table_info({t})")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    acc = {}
    for k, c, v in con.execute(f"select {name_col}, counter_name, value from {t}"):
        acc.setdefault((k[:60], c), []).append(float(v))
    for (k, c), v in sorted(acc.items()):
        if "gemm" in k.lower() or "cijk" in k.lower():
            print(f"{k:60s} {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  i=$((i+1))
done
