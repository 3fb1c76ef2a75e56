#!/bin/bash
# tools/attn_pmc.sh [images] -- on the GPU box: SQ / LDS / TA counters of vit_attention_kernel (separate rocprofv3 --pmc passes,
# --kernel-trace only), per-launch averages -> gpurun_out/attn_pmc.txt.  The study VERDICT r2 #6 asked for.
B=${1:-256}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp
: > $REPO/gpurun_out/attn_pmc.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  d=$REPO/gpurun_out/attn_pmc_$i; rm -rf $d
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $d -o pmc -- python $REPO/tools/vit_attn_probe.py $B > $REPO/gpurun_out/attn_pmc_$i.log 2>&1
  (cd $REPO && python tools/pmc_sq.py $d vit_attention_kernel) >> $REPO/gpurun_out/attn_pmc.txt
  i=$((i+1))
done
cat $REPO/gpurun_out/attn_pmc.txt
find $REPO/gpurun_out -name "*.db" -size +20M -delete
