// sgx_rt.h — thin execution-model layer for the sgx kernels.
//
// The product is compiled by hipcc for gfx950 (default).  Every kernel is written in a
// "phase" style: per-thread work sits between SGX_THREADS_BEGIN/END, workgroup-uniform control
// flow sits outside and reads only LDS/global state, phases are separated by SGX_SYNC().
// That discipline lets the SAME kernel source also be compiled by g++ with -DSGX_EMU into
// tests/emu/libsgx_emu.so, where a workgroup is executed as a sequential loop over its
// threads.  The emulator exists only so the `-m "not gpu"` CI tier can check kernel LOGIC
// against the oracle in a container without a GPU; the product loader (sg_slam_amd/_lib.py)
// never loads it and there is no CPU fallback in the product.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

// Test / tuning taps (include/sgx_debug.h) and the SGX_* environment switches exist only in builds with -DSGX_DEBUG_TAPS (tests/taps/libsgx_taps.so, the emulator):
// in the product build a tap entry is a file-local function the linker drops and sgx_getenv() is a constant NULL, so every switch folds to its default.
#ifdef SGX_DEBUG_TAPS
#define SGX_TAP extern "C"
static inline const char *sgx_getenv(const char *name) { return getenv(name); }
#else
#define SGX_TAP [[maybe_unused]] static
static inline const char *sgx_getenv(const char *) { return nullptr; }
#endif

#ifndef SGX_EMU
// ------------------------------------------------------------------ device build (gfx950)
#include <hip/hip_runtime.h>
#define SGX_THREADS_BEGIN(tid) { const int tid = (int)threadIdx.x;
#define SGX_THREADS_END }
#define SGX_SYNC() __syncthreads()
#define SGX_LDS __shared__
#define SGX_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define SGX_LAUNCH_DYN(kern, grid, block, lds, stream, ...) hipLaunchKernelGGL((kern), grid, block, (lds), stream, __VA_ARGS__)
#define SGX_KERNEL(bounds) __global__ void __launch_bounds__(bounds)
#define SGX_KERNEL_OCC(bounds, waves_per_simd) __global__ void __launch_bounds__(bounds) __attribute__((amdgpu_waves_per_eu(waves_per_simd, waves_per_simd)))
#define SGX_DEV __device__ __forceinline__
#define SGX_CONST __constant__
#define sgx_atomic_add(p, v) atomicAdd((p), (v))
#define sgx_atomic_max(p, v) atomicMax((p), (v))
#define sgx_atomic_min_i32(p, v) atomicMin((p), (v))
#define sgx_atomic_or(p, v) atomicOr((p), (v))
#define SGX_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL((kern), grid, block, 0, stream, __VA_ARGS__)
#define SGX_POPCLL(x) __popcll(x)
#define SGX_WAVE_PRIORITY(p) __builtin_amdgcn_s_setprio(p)   /* issue priority of this wave against co-resident waves (0..3) */
#define SGX_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)      /* value known to be equal across the wave: keep it in a scalar register */
/* true if the predicate holds in any lane of the wave: a scalar branch around work whose result only those lanes select (the emulator always
   takes the branch: same results by construction) */
#define SGX_WAVE_ANY(pred) (__builtin_amdgcn_ballot_w64(pred) != 0)
static __device__ __forceinline__ double sgx_uniform_f64(double x)   /* the same for a double (two scalar registers) */
{ return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x))); }
// per-thread state that must survive SGX_SYNC(): registers on the device, a [threads][count] table in the emulator
#define SGX_PRIV_DECL(type, name, count, threads) type name[count]
#define SGX_PRIV_BIND(name, tid) ((void)0)
typedef hipStream_t sgx_stream_t;
#else
// ------------------------------------------------------------------ kernel-logic emulator (tests only)
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
struct sgx_dim3 { unsigned x, y, z; sgx_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef sgx_dim3 dim3;
extern thread_local sgx_dim3 blockIdx, blockDim, gridDim;
#define SGX_THREADS_BEGIN(tid) for (int tid = 0; tid < (int)blockDim.x; ++tid) {
#define SGX_THREADS_END }
#define SGX_SYNC() ((void)0)
#define SGX_LDS static thread_local
#define SGX_DYN_LDS(name) static thread_local unsigned char name[163840] __attribute__((aligned(16)))
#define SGX_LAUNCH_DYN(kern, grid, block, lds, stream, ...) SGX_LAUNCH(kern, grid, block, stream, __VA_ARGS__)
#define SGX_KERNEL(bounds) static void
#define SGX_KERNEL_OCC(bounds, waves_per_simd) static void
#define SGX_DEV static inline
#define SGX_CONST static
template <class T, class U> static inline T sgx_atomic_add(T *p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T sgx_atomic_max(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T sgx_atomic_or(T *p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
static inline int sgx_atomic_min_i32(int *p, int v) { int o = *p; if (v < o) *p = v; return o; }
#define SGX_LAUNCH(kern, grid, block, stream, ...)                                        \
    do { sgx_dim3 _g = (grid), _b = (block); gridDim = _g; blockDim = _b;                  \
         for (unsigned _z = 0; _z < _g.z; ++_z) for (unsigned _y = 0; _y < _g.y; ++_y)     \
         for (unsigned _x = 0; _x < _g.x; ++_x) { blockIdx = sgx_dim3(_x, _y, _z); (kern)(__VA_ARGS__); } } while (0)
#define SGX_POPCLL(x) __builtin_popcountll(x)
#define SGX_UNIFORM(x) (x)
#define SGX_WAVE_ANY(pred) ((void)(pred), true)
static inline double sgx_uniform_f64(double x) { return x; }
#define SGX_WAVE_PRIORITY(p) ((void)0)
#define SGX_PRIV_DECL(type, name, count, threads) static thread_local type name##_store[threads][count]
#define SGX_PRIV_BIND(name, tid) auto *name = name##_store[tid]
typedef void *sgx_stream_t;
using std::min; using std::max;
// minimal hip* memory API on host memory
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, sgx_stream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, sgx_stream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(sgx_stream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemcpyToSymbolEmu(void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
#endif

// round-half-to-even to int (OpenCV cvRound semantics)
SGX_DEV int sgx_cvround(float v) { return (int)rintf(v); }
