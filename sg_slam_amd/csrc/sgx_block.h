// sgx_block.h — workgroup-level helpers shared by kernels (device + emulator variants with identical semantics)
#pragma once
#include "sgx_rt.h"

// Block-wide exclusive scan of a[0..n) (int32), total written to *total.  Called by all threads
// inside one SGX_THREADS region and followed by SGX_SYNC().
#ifdef SGX_EMU
SGX_DEV void sgx_block_exclusive_scan_i32(int *a, int n, int *total, int tid)
{
    if (tid != 0) return;
    int run = 0;
    for (int i = 0; i < n; i++) { const int v = a[i]; a[i] = run; run += v; }
    *total = run;
}
#else
SGX_DEV void sgx_block_exclusive_scan_i32(int *a, int n, int *total, int tid)
{
    // wave 0 scans the array in 64-wide chunks with shuffles (n <= 1280 -> <= 20 chunks)
    if (tid >= 64) return;
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + tid;
        const int v = i < n ? a[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (tid >= o) inc += t; }
        if (i < n) a[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (tid == 0) *total = carry;
}
#endif

