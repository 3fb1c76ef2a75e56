#!/bin/bash
# tools/pmc_sq.sh <tag> <kernel-name-fragment> -- <command...>   (on the GPU box)
# One rocprofv3 pass with SQ counters (8 slots) + GRBM, --kernel-trace only (no other trace domain: gpurun refuses that
# combination; PMC_COUNTERS overrides the counter list), then the per-launch averages of the kernels whose name contains the fragment -> gpurun_out/pmc_sq_<tag>.txt
TAG=$1; FRAG=$2; shift 3
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
D=$REPO/gpurun_out/pmc_sq_$TAG
rm -rf $D; mkdir -p $REPO/gpurun_out
cd /tmp
COUNTERS=${PMC_COUNTERS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"}
timeout 300 rocprofv3 --pmc $COUNTERS \
    --kernel-trace -d $D -o pmc -- "$@" > $REPO/gpurun_out/pmc_sq_$TAG.log 2>&1
cd $REPO && python tools/pmc_sq.py $D "$FRAG" | tee $REPO/gpurun_out/pmc_sq_$TAG.txt
