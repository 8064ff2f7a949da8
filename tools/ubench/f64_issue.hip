// Micro-benchmark: cycles per fp64 / fp32 VALU instruction of one wave as a function of independent chains (ILP) and waves per SIMD (TLP).
// hipcc --offload-arch=gfx950 -O3 f64_issue.hip -o f64_issue && ./f64_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T, int ILP>
__global__ void k(T *out, unsigned long long *cyc, int iters, T a, T b)
{
    T x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = (T)threadIdx.x + (T)i;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) x[i] = __builtin_fma(x[i], a, b);
        }
    }
    const unsigned long long t1 = clock64();
    T s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename T, int ILP> void run(const char *name, int threads)
{
    T *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * sizeof(T) * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    hipLaunchKernelGGL((k<T, ILP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, (T)1.0000001, (T)1e-9);
    hipLaunchKernelGGL((k<T, ILP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, (T)1.0000001, (T)1e-9);
    hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s ilp=%d waves_per_simd=%d cycles_per_wave_instr=%.2f  (per SIMD: %.2f cycles per instr issued)\n", name, ILP, threads / 256, (double)h / (iters * 16.0 * ILP), (double)h / (iters * 16.0 * ILP * (threads / 256)));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<double, 1>("f64", 256); run<double, 2>("f64", 256); run<double, 4>("f64", 256); run<double, 8>("f64", 256);
    run<double, 1>("f64", 512); run<double, 2>("f64", 512); run<double, 4>("f64", 512); run<double, 1>("f64", 1024); run<double, 4>("f64", 1024);
    run<float, 1>("f32", 256); run<float, 2>("f32", 256); run<float, 4>("f32", 256); run<float, 8>("f32", 256);
    run<float, 1>("f32", 512); run<float, 4>("f32", 512); run<float, 1>("f32", 1024); run<float, 4>("f32", 1024);
    return 0;
}
