// Are the short forms of square root, reciprocal and division used by the vector helpers of
// vrt_trace.h bit-identical to the compiler's IEEE forms on the operand ranges their guards admit?
//
//   sqrt_short(x)   v_sqrt_f32 and the compiler's own one-ulp correction, without the scaling steps
//                   that only operands below 2^-96 need            checked for EVERY x in [2^-90, 2^120]
//   rcp_short(m)    v_rcp_f32 and one Newton step (what the hardware division sequence starts with)
//                   against 1.0f / m                                 checked for EVERY m in [2^-60, 2^60]
//   div_short(a, m) the hardware division sequence without v_div_scale / v_div_fixup: exact by
//                   construction while neither operand needs scaling; checked on random pairs
//                   (|a| in [2^-40, 2^40] or zero, m in +-[2^-40, 2^40], mantissas random and extreme)
//
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/exp/exact_math.hip -o tools/exp/exact_math
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "../../raytracing_amd/csrc/vrt_exact_math.h"

__global__ void k_sqrt(unsigned long long* bad, unsigned* ex)
{
    // x = 2^e * 1.m for e in [-90, 120]: bits from (37 << 23) to (247 << 23) + 0x7fffff
    // (below 2^-96 the compiler's form scales the argument, and the short form differs: 61 948 values of [2^-120, 2^-96))
    const uint32_t lo = 37u << 23, hi = (247u << 23) | 0x7fffffu;
    unsigned long long b = 0;
    for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)u);
        const float ref = sqrtf(x), got = vrt_sqrt_short(x);
        if (__float_as_uint(ref) != __float_as_uint(got)) { if (!b) atomicCAS(ex, 0u, (uint32_t)u); ++b; }
    }
    if (b) atomicAdd(bad, b);
}

__global__ void k_rcp(unsigned long long* bad, unsigned* ex)
{
    const uint32_t lo = 67u << 23, hi = (187u << 23) | 0x7fffffu;   // 2^-60 .. 2^60
    unsigned long long b = 0;
    for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
        for (int s = 0; s < 2; ++s) {
            const float m = __uint_as_float((uint32_t)u | (s ? 0x80000000u : 0u));
            const float ref = 1.0f / m, got = vrt_rcp_short(m);
            if (__float_as_uint(ref) != __float_as_uint(got)) { if (!b) atomicCAS(ex, 0u, (uint32_t)u); ++b; }
        }
    }
    if (b) atomicAdd(bad, b);
}

__device__ __forceinline__ uint32_t pcg(uint64_t& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t x = (uint32_t)(((s >> 18u) ^ s) >> 27u), r = (uint32_t)(s >> 59u);
    return (x >> r) | (x << ((32 - r) & 31));
}

// mantissa patterns that stress rounding: random, all ones, all zeros, one bit, alternating
__device__ __forceinline__ uint32_t mant(uint32_t r, uint32_t sel)
{
    switch (sel % 8u) {
    case 0: return 0u;
    case 1: return 0x7fffffu;
    case 2: return 1u << (r % 23u);
    case 3: return 0x7fffffu ^ (1u << (r % 23u));
    default: return r & 0x7fffffu;
    }
}

__global__ void k_div(uint64_t seed, int iters, unsigned long long* bad, float* ex)
{
    uint64_t s = seed + 977ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x);
    unsigned long long b = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t u = pcg(s), v = pcg(s), w = pcg(s);
        const int em = (int)(u % 81u) - 40, ea = (int)(v % 81u) - 40;
        const float m = __uint_as_float(((uint32_t)(em + 127) << 23) | mant(pcg(s), w) | (u & 0x80000000u));
        float a = __uint_as_float(((uint32_t)(ea + 127) << 23) | mant(pcg(s), w >> 3) | (v & 0x80000000u));
        if ((w >> 8) % 37u == 0) a = 0.f;
        if ((w >> 8) % 41u == 0) a = m;
        const float r = vrt_rcp_short(m);
        const float ref = a / m, got = vrt_div_short(a, m, r);
        if (ref != got && !(ref != ref && got != got)) {   // values (a zero numerator loses its sign in the short form)
            if (!b && ex[0] == 0.f) { ex[0] = a; ex[1] = m; ex[2] = ref; ex[3] = got; }
            ++b;
        }
    }
    if (b) atomicAdd(bad, b);
}

int main()
{
    unsigned long long* d; unsigned* ex; float* fex;
    (void)hipMalloc(&d, 8); (void)hipMalloc(&ex, 4); (void)hipMalloc(&fex, 16);
    unsigned long long h; unsigned hex; float hf[4];
    (void)hipMemset(d, 0, 8); (void)hipMemset(ex, 0, 4);
    hipLaunchKernelGGL(k_sqrt, dim3(4096), dim3(256), 0, 0, d, ex);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hex, ex, 4, hipMemcpyDeviceToHost);
    printf("sqrt_short vs sqrtf, every x in [2^-90, 2^120] (%.3g values): %llu mismatches (first at bits 0x%08x)\n",
           (double)((247u - 37u + 1u) * 8388608.0), h, hex);
    (void)hipMemset(d, 0, 8); (void)hipMemset(ex, 0, 4);
    hipLaunchKernelGGL(k_rcp, dim3(4096), dim3(256), 0, 0, d, ex);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hex, ex, 4, hipMemcpyDeviceToHost);
    printf("rcp_short vs 1.0f / m, every m in +-[2^-60, 2^60] (%.3g values): %llu mismatches (first at bits 0x%08x)\n",
           (double)(2.0 * 121 * 8388608.0), h, hex);
    (void)hipMemset(d, 0, 8); (void)hipMemset(fex, 0, 16);
    const int iters = 65536;
    for (int round = 0; round < 16; ++round)
        hipLaunchKernelGGL(k_div, dim3(4096), dim3(256), 0, 0, 0x9E3779B97F4A7C15ull * (round + 1), iters, d, fex);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, fex, 16, hipMemcpyDeviceToHost);
    printf("div_short vs a / m, %.3g random pairs with both operands in +-[2^-40, 2^40] or a = 0: %llu mismatches", 16.0 * 4096 * 256 * iters, h);
    if (h) printf(" (first: %a / %a = %a, got %a)", hf[0], hf[1], hf[2], hf[3]);
    printf("\n");
    return 0;
}
