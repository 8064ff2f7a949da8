#!/bin/bash
# round 6, GPU session Q: the stand-alone reproducer of the packed-fp32 effect (tools/ubench/pk_f32_corun.hip; built by: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-fno-slp-vectorize])
set -u
O=gpurun_out/r6q; mkdir -p $O; U=tools/ubench; : > $O/pk_f32_corun.txt
{
echo "# the tracker's step arithmetic (mode 0): SLP-vectorised build (packed fp32) vs -fno-slp-vectorize (single fp32), co-runner none / fp32 FMAs / bf16 matrix products"
for c in 0 2 1; do timeout 60 $U/pk_f32_corun_slp $c 20 2048 4000 15; echo -n "   no-SLP build: "; timeout 60 $U/pk_f32_corun_noslp $c 20 2048 4000 15; done
echo "# which 16-lane rows run the chain (bit mask): the last row differs from the first active row whenever it runs together with another one"
for r in 12 10 8 3 4 2; do timeout 60 $U/pk_f32_corun_slp 1 20 2048 4000 $r; done
echo "# single instruction forms beside the bf16 co-runner (mode, s_nop operand between the links)"
for mw in "1 0" "1 7" "2 0" "3 0" "10 0" "4 0" "5 0" "6 0" "6 1" "6 3" "9 0" "7 0" "8 0"; do set -- $mw; timeout 60 $U/pk_f32_corun_slp 1 10 2048 4000 15 $1 $2; done
echo "# ... and the crossed form without a co-runner / beside fp32 FMAs"
timeout 60 $U/pk_f32_corun_slp 0 10 2048 4000 15 6 0; timeout 60 $U/pk_f32_corun_slp 2 10 2048 4000 15 6 0
} 2>&1 | tee -a $O/pk_f32_corun.txt
