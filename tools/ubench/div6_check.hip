// Device-side exhaustive check of sgx_div_c2 (the product's function, included from the kernel header): for all 2^32 float operands u the result equals u / 6.0f bit for bit.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I include -I sg_slam_amd/csrc tools/ubench/div6_check.hip -o tools/ubench/div6_check && tools/ubench/div6_check
#include "sgx_block.h"
#include "sgx_det_kernels.h"
#include <stdio.h>
__global__ void k_check(unsigned long long *bad, unsigned *first, float c2)
{
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned ub = (unsigned)i; float u; memcpy(&u, &ub, 4);
        const float a = sgx_div_c2(u, c2), b = u / c2;
        unsigned ab, bb; memcpy(&ab, &a, 4); memcpy(&bb, &b, 4);
        const bool nan_both = (a != a) && (b != b);
        if (ab != bb && !nan_both) { local++; atomicMin(first, ub); }
    }
    if (local) atomicAdd(bad, local);
}
__global__ void k_check_clip(unsigned long long *bad, float lo, float hi)
{
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned ub = (unsigned)i; float u; memcpy(&u, &ub, 4);
        const float a = sgx_clipf(u, lo, hi), b = fminf(fmaxf(u, lo), hi);
        unsigned ab, bb; memcpy(&ab, &a, 4); memcpy(&bb, &b, 4);
        const bool snan = (ub & 0x7f800000u) == 0x7f800000u && (ub & 0x007fffffu) != 0 && !(ub & 0x00400000u);      // signalling NaN: never the result of an arithmetic instruction
        if (ab != bb && !snan) local++;
    }
    if (local) atomicAdd(bad, local);
}
int main()
{
    {
        unsigned long long *badc, hc = 0; hipMalloc(&badc, 8);
        const float bounds[2][2] = { { 0.f, 6.f }, { 0.f, INFINITY } };
        for (int b = 0; b < 2; b++) {
            hc = 0; hipMemcpy(badc, &hc, 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_check_clip, dim3(4096), dim3(256), 0, 0, badc, bounds[b][0], bounds[b][1]);
            hipDeviceSynchronize(); hipMemcpy(&hc, badc, 8, hipMemcpyDeviceToHost);
            printf("sgx_clipf(u, %g, %g) vs fminf(fmaxf(u, lo), hi) over all 2^32 operands but the signalling NaNs on the device: %llu differences\n", bounds[b][0], bounds[b][1], hc);
            if (hc) return 1;
        }
    }
    unsigned long long *bad, hb = 0; unsigned *first, hf = 0xffffffffu;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice); hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, bad, first, 6.0f);
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("sgx_div_c2(u, 6) vs u / 6.0f over all 2^32 operands on the device: %llu differences%s\n", hb, hb ? " (first operand printed below)" : "");
    if (hb) printf("first differing operand bits 0x%08x\n", hf);
    return hb != 0;
}
