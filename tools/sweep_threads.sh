# tuning sweep of workgroup sizes (env taps in sgx_orb.cpp / sgx_match.cpp); prints per-launch ms at 64 frames, single stream
export SGX_BENCH_TAPS_LIB=1   # the SGX_* switches exist in the tap build only (tests/taps/libsgx_taps.so); bench.py / the tools load it when this is set
run() { timeout 100 python bench.py --streams 64 --no-pipeline --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['roofline']['per_kernel']; print({n: k[n]['avg_ms_per_launch'] for n in ('pyramid_resize','fast_cells','octree','match_project_frame')}, j['config']['mean_inliers'])"; }
for t in 512 1024; do echo PYR=$t; SGX_TUNE_PYR_THREADS=$t run; done
for t in 256 512; do echo OCT=$t; SGX_TUNE_OCT_THREADS=$t run; done
