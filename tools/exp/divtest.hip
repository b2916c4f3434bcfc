// Experiment: is the shared-reciprocal division sequence bit-identical to the compiler's IEEE
// a / b on gfx950 for the operand ranges the tracer's normalisations see?
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/exp/divtest.hip -o tools/exp/divtest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float rcp_refined(float m)
{
    float r = __builtin_amdgcn_rcpf(m);
    float e = __builtin_fmaf(-m, r, 1.f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_shared(float a, float m, float r)
{
    float q = a * r;
    float e = __builtin_fmaf(-m, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-m, q, a);
    return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ uint32_t pcg(uint64_t& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t x = (uint32_t)(((s >> 18u) ^ s) >> 27u), r = (uint32_t)(s >> 59u);
    return (x >> r) | (x << ((32 - r) & 31));
}

// mode 0: m in [1e-5, 1e4] log-uniform, |a| <= m (normalisation);  mode 1: a, m any normal floats
// with exponents in [-40, 40];  mode 2: m in [0.5, 2), a tiny (quotient near the denormal range)
__global__ void k(int mode, uint64_t seed, int iters, unsigned long long* mism, float* ex)
{
    uint64_t s = seed + 977ull * (blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t u = pcg(s), v = pcg(s);
        float a, m;
        if (mode == 0) {
            int e = (int)(u % 30u) - 17;                         // 2^-17 .. 2^12
            m = __uint_as_float(((uint32_t)(e + 127) << 23) | (v & 0x7fffffu));
            float f = (float)(pcg(s) >> 8) * (1.f / 16777216.f); // [0,1)
            a = m * f * ((u & 0x80000000u) ? -1.f : 1.f);
            if (pcg(s) % 64u == 0) a = m;                        // quotient exactly 1
        } else if (mode == 1) {
            int ea = (int)(u % 81u) - 40, em = (int)(v % 81u) - 40;
            a = __uint_as_float(((uint32_t)(ea + 127) << 23) | (pcg(s) & 0x7fffffu) | (u & 0x80000000u));
            m = __uint_as_float(((uint32_t)(em + 127) << 23) | (pcg(s) & 0x7fffffu) | (v & 0x80000000u));
        } else if (mode == 2) {
            m = __uint_as_float((126u << 23) | (v & 0xffffffu));
            int ea = -(int)(u % 30u) - 100;                      // 2^-100 .. 2^-129
            a = __uint_as_float(((uint32_t)max(ea + 127, 0) << 23) | (pcg(s) & 0x7fffffu));
        } else if (mode == 3) {                                  // small but normal quotients
            int em = (int)(v % 41u) - 20;
            m = __uint_as_float(((uint32_t)(em + 127) << 23) | (pcg(s) & 0x7fffffu));
            int eq = -(int)(u % 26u) - 100;                      // quotient exponent -100 .. -125
            int ea = eq + em + 1;
            if (ea < -126) ea = -126;
            a = __uint_as_float(((uint32_t)(ea + 127) << 23) | (pcg(s) & 0x7fffffu));
        } else {                                                 // extreme denominators, normal quotient
            int em = (v & 1u) ? 60 + (int)(u % 60u) : -60 - (int)(u % 60u);
            m = __uint_as_float(((uint32_t)(em + 127) << 23) | (pcg(s) & 0x7fffffu));
            int ea = em - (int)(pcg(s) % 20u);
            if (ea < -126) ea = -126; if (ea > 126) ea = 126;
            a = __uint_as_float(((uint32_t)(ea + 127) << 23) | (pcg(s) & 0x7fffffu));
        }
        float ref = a / m;
        float got = div_shared(a, m, rcp_refined(m));
        if (mode == 3 && fabsf(ref) < 0x1p-126f) continue;       // subnormal results are counted by mode 2
        if (__float_as_uint(ref) != __float_as_uint(got)) {
            if (bad == 0 && ex[0] == 0.f) { ex[0] = a; ex[1] = m; ex[2] = ref; ex[3] = got; }
            ++bad;
        }
    }
    if (bad) atomicAdd(mism, bad);
}

int main()
{
    unsigned long long* d; float* ex;
    hipMalloc(&d, 8); hipMalloc(&ex, 16);
    for (int mode = 0; mode < 5; ++mode) {
        hipMemset(d, 0, 8); hipMemset(ex, 0, 16);
        const int blocks = 4096, threads = 256, iters = 4096;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, 12345ull + mode, iters, d, ex);
        unsigned long long h = 0; float hx[4];
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(hx, ex, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches of %llu", mode, h, (unsigned long long)blocks * threads * iters);
        if (h) printf("  e.g. a=%a m=%a ref=%a got=%a", hx[0], hx[1], hx[2], hx[3]);
        printf("\n");
    }
    return 0;
}
