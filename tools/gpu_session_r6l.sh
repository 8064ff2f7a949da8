#!/bin/bash
# round 6, GPU session L: the envelope solver with its tile products on the fp64 matrix cores and software-pipelined update pairs: BA tests, config-4 phases, campaign slices on the device
set -u
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests -q -p no:cacheprovider -m gpu -k "ba or bundle or solver or sim3" 2>&1 | tail -2 | tee $O/tests.txt
timeout 300 python tools/bench_ba_phases.py 2>/dev/null | tee $O/ba_phases.json | python -c "import json,sys; j=json.load(sys.stdin); [print(k, v['seconds'], v['kernel_ms']['ba_solve'], v['lm_iterations'], v['chi2'], v['erased']) for k,v in j['solvers'].items()]"
export SGX_CAMPAIGN_LIB=device
timeout 200 python tools/campaign_solvers.py 166 150 100000 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/campaign_solvers.txt
timeout 150 python tools/campaign_ba_large.py 167 100 100000 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/campaign_ba_large.txt
