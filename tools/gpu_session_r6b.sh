#!/bin/bash
# round 6, GPU session B: SQ counters of k_hrb (and k_fused_block2 for comparison, SGX_DET_HRB=0) on the per-step harness: what the first version waits for
set -u
R=$PWD; O=$R/gpurun_out/r6b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B=512
P="python $R/tools/prof_det_ops.py $B 2"
pass() { d=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$d -o p -- $P > /dev/null 2>&1; }
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM
pass b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE
pass c SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM
export SGX_DET_HRB=0
pass a0 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM
pass b0 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE
cd $R
for d in a b c a0 b0; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f | grep -A9 "k_hrb\|k_fused_block2"; done > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
