set -u
R=$PWD; O=$R/gpurun_out/pmc_flow; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/a -o p -- python $R/tools/bench_flow.py --reps 3 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/b -o p -- python $R/tools/bench_flow.py --reps 3 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $O/c -o p -- python $R/tools/bench_flow.py --reps 3 > /dev/null 2>&1
cd $R
for d in a b c; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; [ -n "$f" ] && python tools/pmc_summary.py $f | grep -A12 "k_lk_track\|k_fm_ransac"; done
