#!/bin/bash
# copies the summaries of a tools/collect_profiles.sh run (gpurun_out/<tag>/) into profiles/<tag>_* under the names bench.py's latest_profile() and DESIGN.md use
set -e
T=${1:?tag}; O=gpurun_out/$T; P=profiles
c() { [ -f "$O/$1" ] && cp "$O/$1" "$P/${T}_$2" && echo "$P/${T}_$2" || echo "missing $O/$1" >&2; }
c bench_default.json bench_default.json
c bench_under_rocprof.json bench_under_rocprof.json
c bench_serial.json bench_serial_no_detector.json
c stats/b_kernel_stats.csv bench_kernel_stats.csv
c standalone.json standalone.json
c traffic.json traffic.json
c pmc_insts.json pmc_insts.json
c pmc_kernels.md pmc_kernels.md
c pmc_detector.md pmc_detector.md
c pmc_detector_steps.txt pmc_detector_steps.txt
c detector_ops.txt detector_ops.txt
c ba_phases.json ba_2000kf_50klm_phases.json
c ba_500.json ba_500kf_12klm.json
c ba_stats/b_kernel_stats.csv ba_kernel_stats.csv
c flow.txt flow_standalone.json
c localba.json localba_bench.json
