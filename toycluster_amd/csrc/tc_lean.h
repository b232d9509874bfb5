/* tc_lean.h -- sqrt and divide for operands known to be in the normal range.
 *
 * The compiler's IEEE sequences wrap a Newton-Raphson core in range handling: v_div_scale / v_div_fixup for
 * quotients near the exponent limits, ldexp pre/post-scaling for roots of tiny or huge numbers.  The pair
 * arithmetic of the SPH kernels only ever sees separations of ~1e-7..1 box lengths (1e-4..1e5 kpc), so the core
 * alone returns the same bits; tools/ubench/lean_math_check.hip verifies that against the compiler's own
 * sequences on 2.7e8 random operands per function.  Zero inputs keep the IEEE result through an explicit
 * select (coincident particles: the reference divides by zero there and so must we).
 * Requires -ffp-contract=off (the FMAs below are explicit). */
#ifndef TC_LEAN_H
#define TC_LEAN_H

__device__ __forceinline__ double tc_sqrt_f64_lean(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0 ? x : g;
}

/* the same core for x > 0 (no zero select) */
__device__ __forceinline__ double tc_sqrt_f64_lean_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

__device__ __forceinline__ float tc_sqrt_f32_lean_pos(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float em = __builtin_fmaf(-sm, s, x);
    const float ep = __builtin_fmaf(-sp, s, x);
    float r = em <= 0.0f ? sm : s;
    return ep > 0.0f ? sp : r;
}

__device__ __forceinline__ double tc_rcp_f64_lean_nz(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double r = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(r, y, y);
}

__device__ __forceinline__ float tc_sqrt_f32_lean(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float em = __builtin_fmaf(-sm, s, x);
    const float ep = __builtin_fmaf(-sp, s, x);
    float r = em <= 0.0f ? sm : s;
    r = ep > 0.0f ? sp : r;
    return x == 0 ? x : r;
}

__device__ __forceinline__ float tc_div_f32_lean(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

__device__ __forceinline__ double tc_rcp_f64_lean(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double r = __builtin_fma(-b, y, 1.0);
    const double q = __builtin_fma(r, y, y);
    return b == 0 ? 1.0 / b : q;
}

/* a / b for |a|, |b| well inside the normal range (no quotient near the exponent limits) */
__device__ __forceinline__ double tc_div_f64_lean(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

#endif