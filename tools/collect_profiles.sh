#!/bin/bash
# Re-collects the measurement artefacts kept under profiles/ on a 1-GPU MI355X box (run from the repo root; ~6 minutes of GPU time):
#   tools/collect_profiles.sh <tag>        e.g.  /usr/local/graft/bin/gpurun --timeout 1500 -- 'tools/collect_profiles.sh r2_a'
# Writes into gpurun_out/<tag>/; copy what should be judged into profiles/ afterwards.  Counter passes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in
# separate passes, --kernel-trace only (no sys / runtime trace domains), each under a hard timeout.  The SQ / TCC passes run the chain without the detector
# (its ~100-launch graph is profiled per step by tools/prof_det_ops.py instead) on 4 steps (1 warm-up + 3 timed).
set -u
TAG=${1:-run}
S=${STREAMS:-512}          # frames per launch = bench.py's default --streams
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
PB="python $R/bench.py --streams $S --no-cpu-baseline --no-config2 --no-config4 --no-host-input --no-detector --steps 3 --warmup 1"
# 1. the default bench line (full chain, detector on, CPU baseline, config-2 secondary)
timeout 600 python bench.py --streams $S > $O/bench_default.json 2> $O/bench_default.err
# 2. kernel statistics of the same command (rocprofv3 wants a writable cwd / TMPDIR)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --streams $S --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 48 --warmup 4 > $O/bench_under_rocprof.json 2>/dev/null
# 3. SQ instruction counters (two passes) and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/sq_a -o p -- $PB > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq_b -o p -- $PB > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $PB > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $PB > /dev/null 2>&1
# 4. detector: per-step table, then the same SQ counters over its launches (batch $S, 2 repetitions, steps launched one by one)
timeout 200 python $R/tools/prof_det_ops.py $S 5 > $O/detector_ops.txt 2>/dev/null
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/det_sq -o p -- python $R/tools/prof_det_ops.py $S 2 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $O/det_mfma -o p -- python $R/tools/prof_det_ops.py $S 2 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/det_fetch -o p -- python $R/tools/prof_det_ops.py $S 2 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/det_write -o p -- python $R/tools/prof_det_ops.py $S 2 > /dev/null 2>&1
cd $R
f() { find $O/$1 -name "*counter_collection.csv" | head -1; }
python tools/pmc_traffic.py --det $(f det_fetch) $(f det_write) 3 $(f fetch) $(f write) $S 4 $O/traffic.json $O/bench_under_rocprof.json > /dev/null
python tools/pmc_insts.py $O/pmc_insts.json $S 4 $(f sq_a) $(f sq_b) --det $(f det_sq) $S $(python - <<PY
import csv
n = sum(1 for r in csv.DictReader(open("$(f det_sq)")) if r['Counter_Name'] == 'SQ_WAVES' and ('k_det_preprocess' in r['Kernel_Name'] or 'k_stem_pre' in r['Kernel_Name']))
print(max(n, 1))
PY
) > $O/pmc_insts.txt
for d in sq_a sq_b fetch write det_sq det_mfma det_fetch det_write; do python tools/pmc_summary.py $(f $d) > $O/pmc_$d.txt 2>/dev/null; done
# 5. standalone kernel times: the chain on ONE stream without the detector (no kernel shares the GPU with another), the detector on its own
timeout 300 python bench.py --streams $S --no-cpu-baseline --no-detector --no-config2 --no-config4 --no-host-input --no-pipeline --steps 48 --warmup 4 > $O/bench_serial.json 2>/dev/null
timeout 200 python tools/prof_det_output.py $S 10 > $O/det_standalone.txt 2>/dev/null
python - <<PY
import json, re
j = json.load(open("$O/bench_serial.json")); pk = j["roofline"]["per_kernel"]
res = {k: pk[k]["avg_ms_per_launch"] for k in pk}
for l in open("$O/det_standalone.txt"):
    m = re.match(r"(det_\w+)\s+([\d.]+) ms per launch", l)
    if m: res[m.group(1)] = float(m.group(2))
json.dump({"note": "average launch duration with nothing else on the GPU: bench.py --no-detector --no-pipeline (one stream) for the chain, tools/prof_det_output.py for the detector; $S frames per launch",
           "frames_per_launch": $S, "avg_ms_per_launch": res}, open("$O/standalone.json", "w"), indent=1)
PY
# 5b. the matrix-core inverted-residual block kernels of the detector: per-step SQ counter table; bundle adjustment (config 4): phases with both solvers + kernel statistics
bash $R/tools/pmc_irb.sh $TAG > /dev/null 2>&1; cp $R/gpurun_out/pmc_irb_$TAG.txt $O/pmc_detector_steps.txt 2>/dev/null
timeout 200 python tools/bench_ba_phases.py > $O/ba_phases.json 2>/dev/null
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba_stats -o b -- python $R/tools/bench_ba_phases.py > /dev/null 2>&1
cd $R
# 6. standalone stage benches
timeout 100 python tools/bench_flow.py > $O/flow.txt 2>/dev/null
timeout 100 python tools/bench_ba.py > $O/localba.json 2>/dev/null
timeout 100 python tools/bench_ba_big.py 500 12000 > $O/ba_500.json 2>/dev/null
timeout 100 python tools/bench_ba_big.py > $O/ba_2000.json 2>/dev/null
python tools/pmc_markdown.py $O $S > /dev/null
find $O -name "*.csv" -size +4M -delete      # raw traces stay on the box; the summaries above are what gets committed
# a summary that is a Python traceback (or empty) is not evidence: refuse to keep it (VERDICT r3 weak #3b: r3_pmc_detector_steps.txt was a 4-line AssertionError)
for f in $O/*.txt $O/*.json $O/*.md; do
  [ -f "$f" ] || continue
  if [ ! -s "$f" ] || head -5 "$f" | grep -q "^Traceback"; then echo "collect_profiles: $f is empty or a traceback - removed" >&2; mv "$f" "$f.FAILED"; fi
done
python - <<PY
import json
j = json.load(open("$O/bench_default.json"))
print("default bench:", round(j["value"]), "frames/s,", round(j["ms_per_step"], 3), "ms/step; dominant kernel", j["roofline"]["kernel"], "frac", j["roofline"]["frac"])
PY
