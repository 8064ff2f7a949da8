# PMC passes over the detector plan launched step by step (tools/prof_det_ops.py B 1): one row per plan step (tools/pmc_det_ops.py).
# usage (GPU box): bash tools/pmc_irb.sh [tag]   -> gpurun_out/pmc_irb_<tag>.txt
R=$PWD; TAG=${1:-x}; O=$R/gpurun_out/pmc_irb_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B=${BATCH:-512}
python $R/tools/prof_det_ops.py $B 1 > $O/ops.txt 2>/dev/null
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/a -o p -- python $R/tools/prof_det_ops.py $B 1 > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/b -o p -- python $R/tools/prof_det_ops.py $B 1 > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/c -o p -- python $R/tools/prof_det_ops.py $B 1 > /dev/null 2>&1
cd $R
fa=$(find $O/a -name "*counter_collection.csv" | head -1); fb=$(find $O/b -name "*counter_collection.csv" | head -1); fc=$(find $O/c -name "*counter_collection.csv" | head -1)
python tools/pmc_det_ops.py $O/ops.txt 1 $fa $fb $fc > $R/gpurun_out/pmc_irb_$TAG.txt 2>&1
rm -rf $O/a $O/b $O/c
