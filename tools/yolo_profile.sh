#!/bin/bash
# tools/yolo_profile.sh [batch] [nchw-only|hip-only] -- rocprofv3 kernel stats of the YOLOv7-E6E forward at 448x640 f16 (what YOLOv7.predict_batch runs)
B=${1:-32}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_yolo
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_yolo -o p -- python $R/tools/yolo_probe.py $B ${2:-nchw-only} > $R/gpurun_out/prof_yolo.log 2>&1
cd $R && python tools/rocprof_summary.py gpurun_out/prof_yolo/p_results.db gpurun_out/yolo_e6e_b${B}_kernel_stats.csv | head -5
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/yolo_e6e_b${B}_kernel_stats.csv")))
tot=sum(float(r["total_us"]) for r in rows)
for r in rows[:25]:
    print(f"{r['kernel'][:90]:90s} calls={r['calls']:>6s} avg={float(r['avg_us']):9.1f} us {float(r['total_us'])/tot*100:6.2f}%")
PY
find gpurun_out/prof_yolo -name "*.db" -size +20M -delete
