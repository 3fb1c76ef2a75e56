#!/bin/bash
# round-4 session 1: where GroundingDINO's time goes at 64 frames (before any change)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_gd -o p -- python $R/tools/gdino_profile_probe.py 64 3 > $R/$O/gdino_probe.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_gd/p_results.db $(cat /tmp/gdino_window_ms) 70 > $O/r04_gdino_b64_before.txt 2>&1
tail -3 $O/gdino_probe.log; head -40 $O/r04_gdino_b64_before.txt
