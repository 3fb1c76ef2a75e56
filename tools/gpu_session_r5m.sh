#!/bin/bash
# GroundingDINO at 8 frames: kernel table of the steady state + sections + full-step parts at 8 environments
set -u
R=$(pwd); O=gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_gd8 -o p -- python $R/tools/gdino_profile_probe.py 8 6 > $R/$O/gdino8_rocprof.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_gd8/p_results.db $(cat /tmp/gdino_window_ms) 60 > $O/gdino_b8_kernels.txt 2>&1
(timeout 300 python tools/gdino_sections_probe.py 8 1 split 2>&1 | grep -v amdgpu.ids) > $O/gdino_sections_b8.txt
(timeout 500 python tools/full_step_parts_probe.py 8 2>&1 | grep -v amdgpu.ids | tail -8) > $O/full_step_parts_e8.txt
head -50 $O/gdino_b8_kernels.txt | cut -c1-170; cat $O/gdino_sections_b8.txt | tail -12; cat $O/full_step_parts_e8.txt
