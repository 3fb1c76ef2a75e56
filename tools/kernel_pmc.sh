#!/bin/bash
# tools/kernel_pmc.sh <tag> <kernel-name-fragment> -- <command...>   (on the GPU box)
# SQ / LDS / TA counters of the kernels whose name contains the fragment: separate rocprofv3 --pmc passes (--kernel-trace only; gpurun
# refuses --pmc combined with other trace domains), per-launch averages -> gpurun_out/pmc_<tag>.txt
TAG=$1; FRAG=$2; shift 3
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp
: > $REPO/gpurun_out/pmc_$TAG.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  d=/tmp/pmc_${TAG}_$i; rm -rf $d
  timeout 150 rocprofv3 --pmc $set --kernel-trace -d $d -o pmc -- "$@" > $REPO/gpurun_out/pmc_${TAG}_$i.log 2>&1
  (cd $REPO && python tools/pmc_sq.py $d "$FRAG") >> $REPO/gpurun_out/pmc_$TAG.txt
  i=$((i+1))
done
cat $REPO/gpurun_out/pmc_$TAG.txt
