R=$PWD; O=$R/gpurun_out/pmc_lk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export SGX_BENCH_TAPS_LIB=1   # the SGX_* switches exist in the tap build only (tests/taps/libsgx_taps.so); bench.py / the tools load it when this is set
for v in 1 2; do
SGX_LK_KPW=$v timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/a$v -o p -- python $R/tools/bench_flow.py --reps 3 > /dev/null 2>&1
SGX_LK_KPW=$v timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b$v -o p -- python $R/tools/bench_flow.py --reps 3 > /dev/null 2>&1
for d in a$v b$v; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -A9 "k_lk_track"; done
done
