// Exhaustive check of sgx_div_c2 (sgx_det_kernels.h): for every float u, the guarded reciprocal form equals u / 6.0f bit for bit.  gcc -O2 -fopenmp -ffp-contract=off tools/check_div6.c -lm && ./a.out  (17 s on 4 cores)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float fl(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char **argv)
{
    const float c = 6.0f, r = 1.0f / 6.0f;     // RN(1/6)
    const float sc = ldexpf(1.0f, -3);
    unsigned long long bad = 0, guarded = 0, fast = 0;
    uint32_t lo_fail = 0xffffffffu, hi_fail = 0;
#pragma omp parallel for reduction(+:bad,guarded,fast)
    for (long long i = 0; i < (1ll << 32); i++) {
        const uint32_t ub = (uint32_t)i; const float u = fl(ub);
        const float t = u * sc;
        const int cls = fpclassify(t);
        const int slow = (cls == FP_SUBNORMAL) || (cls == FP_INFINITE) || (cls == FP_NAN);
        const float ref = u / c;
        if (slow) { guarded++; continue; }
        float q0 = u * r, e = fmaf(-q0, c, u), q = fmaf(e, r, q0);
        uint32_t qb = (bits(q) & 0x7fffffffu) | (ub & 0x80000000u);      // sign of u
        fast++;
        if (qb != bits(ref)) { bad++; if (bad < 20) printf("MISMATCH u=%08x fast=%08x ref=%08x\n", ub, qb, bits(ref)); }
    }
    printf("fast %llu guarded %llu bad %llu\n", fast, guarded, bad);
    return bad != 0;
}
