// Short forms of IEEE square root, reciprocal and division for operands that need no scaling.
//
// hipcc expands sqrtf(x) and a / b (float, correctly rounded) into 16 and 11 instructions.  A good
// part of both handles operands near the ends of the exponent range: the square root scales
// arguments below 2^-96 up and its result down, the division scales numerator and denominator
// (v_div_scale_f32), undoes that (v_div_fmas_f32) and patches infinities, zeros and NaNs
// (v_div_fixup_f32).  The functions below are those expansions without these steps, instruction for
// instruction, so on operands that need no scaling they return the same bits; the callers (vrt_trace.h)
// decide that with one test per wave and take the compiler's form otherwise.
// tools/exp/exact_math.hip checks them on the GPU: the square root for every float in
// [2^-90, 2^120], the reciprocal for every float in +-[2^-60, 2^60], the division on 1.1e12 random
// operand pairs in +-[2^-40, 2^40] with stress mantissas: no mismatch.
#pragma once
#include <hip/hip_runtime.h>

// sqrtf(x) for 2^-96 <= x < inf: v_sqrt_f32 is within one ulp; the two neighbours are tried with
// exact residuals (the correction the compiler's expansion makes).
__device__ __forceinline__ float vrt_sqrt_short(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = (0.f >= rm) ? sm : s;
    r = (0.f < rp) ? sp : r;
    return r;
}

// The refined reciprocal the division sequence starts with; equals 1.0f / m for every m in +-[2^-60, 2^60].
__device__ __forceinline__ float vrt_rcp_short(float m)
{
    const float r = __builtin_amdgcn_rcpf(m);
    const float e = __builtin_fmaf(-m, r, 1.f);
    return __builtin_fmaf(e, r, r);
}

// a / m given r = vrt_rcp_short(m): the quotient steps of the division sequence.  Same bits as a / m
// while |a| and |m| lie in [2^-40, 2^40] (no scaling, no special values); a = +-0 gives +0.
__device__ __forceinline__ float vrt_div_short(float a, float m, float r)
{
    float q = a * r;
    float e = __builtin_fmaf(-m, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-m, q, a);
    return __builtin_fmaf(e, r, q);
}
