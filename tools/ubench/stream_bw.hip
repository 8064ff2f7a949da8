// Micro-benchmark: achievable HBM copy rate on MI355X for the access shapes the detector kernels use, on a 16-channel 150x150 fp32 activation x 256 images (368.6 MB in + 368.6 MB out):
//   lin4   linear copy, 16 B per lane                      (ideal streaming)
//   lin1   linear copy, 4 B per lane
//   rows   the pointwise-convolution shape: a wave reads 2 channel rows x 32 pixels per load (2 x 128 B), 16 rows, writes the same shape  (k_conv_pw2, PXB = 4)
//   rows4  same tile, 16 B per lane: lanes 0-31 read 128 consecutive pixels of channel k, lanes 32-63 of channel k+1
// hipcc --offload-arch=gfx950 -O3 stream_bw.hip -o stream_bw && ./stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 22500
#define C 16
#define B 256
__global__ void __launch_bounds__(256) lin4(const float4 *in, float4 *out, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = in[i]; }
__global__ void __launch_bounds__(256) lin1(const float *in, float *out, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = in[i]; }
__global__ void __launch_bounds__(256) lin4x4(const float4 *in, float4 *out, size_t n) { size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x); float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = in[min(i + (size_t)u * gridDim.x * 256, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) if (i + (size_t)u * gridDim.x * 256 < n) out[i + (size_t)u * gridDim.x * 256] = v[u]; }
template <int PXB>
__global__ void __launch_bounds__(256) rows(const float *in, float *out, int total)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int g0 = (blockIdx.x * 4 + wave) * 32 * PXB;
    float v[C / 2][PXB];
    unsigned off[PXB];
#pragma unroll
    for (int m = 0; m < PXB; m++) { const unsigned g = min(g0 + 32 * m + l31, total - 1), b = g / N, n = g - b * N; off[m] = b * (C * N) + n + half * N; }
#pragma unroll
    for (int k = 0; k < C / 2; k++)
#pragma unroll
        for (int m = 0; m < PXB; m++) v[k][m] = in[(size_t)(2 * k) * N + off[m]];
#pragma unroll
    for (int k = 0; k < C / 2; k++)
#pragma unroll
        for (int m = 0; m < PXB; m++) if (g0 + 32 * m + l31 < total) out[(size_t)(2 * k) * N + off[m]] = v[k][m];
}
__global__ void __launch_bounds__(256) rows4(const float *in, float *out, int total)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int g0 = (blockIdx.x * 4 + wave) * 128 + 4 * l31;       // 4 consecutive pixels per lane (N % 4 == 0: never straddles an image)
    const unsigned g = min(g0, total - 4), b = g / N, n = g - b * N;
    const size_t off = (size_t)b * (C * N) + n + (size_t)half * N;
    float4 v[C / 2];
#pragma unroll
    for (int k = 0; k < C / 2; k++) v[k] = *(const float4 *)(in + (size_t)(2 * k) * N + off);
#pragma unroll
    for (int k = 0; k < C / 2; k++) if (g0 < total) *(float4 *)(out + (size_t)(2 * k) * N + off) = v[k];
}
int main()
{
    const size_t n = (size_t)B * C * N; float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMemset(a, 1, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int total = B * N;
    for (int which = 0; which < 7; which++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(lin4, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n / 4);
            else if (which == 1) hipLaunchKernelGGL(lin1, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n);
            else if (which == 2) hipLaunchKernelGGL(lin4x4, dim3((n / 16 + 255) / 256), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n / 4);
            else if (which == 3) hipLaunchKernelGGL(rows<4>, dim3((total + 511) / 512), dim3(256), 0, 0, a, b, total);
            else if (which == 4) hipLaunchKernelGGL(rows<2>, dim3((total + 255) / 256), dim3(256), 0, 0, a, b, total);
            else if (which == 5) hipLaunchKernelGGL(rows<1>, dim3((total + 127) / 128), dim3(256), 0, 0, a, b, total);
            else hipLaunchKernelGGL(rows4, dim3((total + 511) / 512), dim3(256), 0, 0, a, b, total);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
        }
        static const char *nm[] = { "lin4", "lin1", "lin4x4", "rows<4>", "rows<2>", "rows<1>", "rows4" };
        printf("%-8s %.4f ms  %.2f TB/s (read + write)\n", nm[which], best, 2.0 * n * 4 / (best * 1e-3) / 1e12);
    }
    return 0;
}
