R=$PWD; O=$R/gpurun_out/pmc_det; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B=${BATCH:-512}
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/a -o p -- python $R/tools/prof_det_ops.py $B 2 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $O/b -o p -- python $R/tools/prof_det_ops.py $B 2 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM --kernel-trace --output-format csv -d $O/c -o p -- python $R/tools/prof_det_ops.py $B 2 > /dev/null 2>&1
cd $R
for d in a b c; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f | grep -A12 "k_fused_block"; done
