#!/bin/bash
# MobileSAM at 32 frames: kernel table of the steady state
set -u
R=$(pwd); O=gpurun_out/r5v; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_sam -o p -- python $R/tools/sam_probe.py 32 > $R/$O/sam_rocprof.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_sam/p_results.db 224 45 > $O/sam_b32_kernels.txt 2>&1
tail -1 $O/sam_rocprof.log; head -48 $O/sam_b32_kernels.txt | cut -c1-190
