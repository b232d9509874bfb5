/*
 * tc_math.h -- scalar arithmetic shared by all kernels of libtcgpu (gfx950).
 *
 * Everything here is a pure function of its arguments and is marked TC_HD so the
 * very same lines can be unit-checked on the host (tests/hostcheck) as well as run
 * inside the HIP kernels.  The library must be compiled with -ffp-contract=off:
 * the reference (gcc -std=c99, baseline x86-64) never fuses a multiply-add, and the
 * f32 ball-query predicate decides set membership.
 *
 * Reference lines whose arithmetic is reproduced are cited per function
 * (reference = jdonnert/Toycluster, paths relative to src/).
 */
#ifndef TC_MATH_H
#define TC_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define TC_HD __host__ __device__ __forceinline__
#else
#define TC_HD static inline
#endif

#ifdef TCGPU_SPH_CUBIC_SPLINE       /* the reference's -DSPH_CUBIC_SPLINE build, globals.h:40-52 */
#define TC_DESNNGB 50               /* globals.h:42 */
#else
#define TC_DESNNGB 295              /* globals.h:48 */
#endif
#define TC_NNGBDEV 0.05             /* globals.h:49 */
#define TC_NGBMAX (TC_DESNNGB * 8)  /* globals.h:50 */
#define TC_NUMITER 64               /* wvt_relax.c:7 */
#define TC_ERRDIFF_LIMIT 0.01       /* wvt_relax.c:8 */
#define TC_NTRIPLETS 42             /* peano.h:1 */
#define TC_MAXHALOS 4096            /* globals.h:56 */

#define TC_PI 3.14159265358979323846
#define TC_SQRT3 1.73205080756887719        /* globals.h:62 */
#define TC_FOURPITHIRD 4.18879032135009765  /* globals.h:63 */
#define TC_WC6_NORM (1365.0 / (64 * TC_PI)) /* sph.c:431 */

typedef unsigned __int128 tc_u128;

/* ------------------------------------------------------------------ Peano-Hilbert */

/* Integer coordinates of a particle: X = trunc(x * 2^63) with x = (double)pos / box
 * (peano.c:66-68,134-136).  Axis order of the transpose array is {y, z, x}. */
TC_HD void tc_scaled_coords(float px, float py, float pz, double box, uint64_t X[3])
{
    const double m = 9223372036854775808.0; /* 2^63 */
    double x = (double)px / box, y = (double)py / box, z = (double)pz / box;
    X[0] = (uint64_t)(y * m);
    X[1] = (uint64_t)(z * m);
    X[2] = (uint64_t)(x * m);
}

/* Skilling's transpose <-> Hilbert transform, "inverse undo" + Gray encode
 * (peano.c:140-177).  Works in place on the {y,z,x} transpose array.
 * The reference runs the undo loop over q = 2^63 ... 2.  The step at q only rewrites bits BELOW q, and the Gray
 * encode only mixes a bit with HIGHER bits, so bits 63..21 of the result -- all the key uses (peano.c:181-200:
 * 43 triplets from bit 63 down to bit 21) -- are final after the step q = 2^22: the loop stops there.  Bits
 * below 21 of the returned words are NOT the reference's (nobody reads them). */
TC_HD void tc_hilbert_transpose(uint64_t X[3])
{
    uint64_t X0 = X[0], X1 = X[1], X2 = X[2];
    for (uint64_t q = 1ULL << 63; q > (1ULL << 21); q >>= 1) {
        const uint64_t P = q - 1;
        if (X0 & q) X0 ^= P;
        if (X1 & q) {
            X0 ^= P;
        } else {
            uint64_t t = (X0 ^ X1) & P;
            X0 ^= t; X1 ^= t;
        }
        if (X2 & q) {
            X0 ^= P;
        } else {
            uint64_t t = (X0 ^ X2) & P;
            X0 ^= t; X2 ^= t;
        }
    }
    X1 ^= X0;
    X2 ^= X1;
    uint64_t t = X2;
    X2 ^= X2 >> 1; X2 ^= X2 >> 2; X2 ^= X2 >> 4; X2 ^= X2 >> 8; X2 ^= X2 >> 16; X2 ^= X2 >> 32;
    t ^= X2;
    X1 ^= t;
    X0 ^= t;
    X[0] = X0; X[1] = X1; X[2] = X2;
}

/* Spread the low 21 bits of v so that bit k lands at bit 3k. */
TC_HD uint64_t tc_spread3(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x001f00000000ffffULL;
    v = (v | (v << 16)) & 0x001f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}

/* 128-bit key of peano.c:181-200: triplets (X0,X1,X2) of bits 62..21, MSB first,
 * occupying key bits 127..2 (the bit-63 triplet is shifted out, low 2 bits are 0).
 * Written as two 21-triplet interleaves instead of the reference's 43-step loop. */
TC_HD void tc_key_from_transpose(const uint64_t X[3], uint64_t *hi, uint64_t *lo)
{
    /* triplets for bits 62..42 -> 63 bits "a"; bits 41..21 -> 63 bits "b" */
    uint64_t a = (tc_spread3(X[0] >> 42) << 2) | (tc_spread3(X[1] >> 42) << 1) | tc_spread3(X[2] >> 42);
    uint64_t b = (tc_spread3(X[0] >> 21) << 2) | (tc_spread3(X[1] >> 21) << 1) | tc_spread3(X[2] >> 21);
    /* key = (a << 65) | (b << 2) */
    *hi = (a << 1) | (b >> 62);
    *lo = b << 2;
}

TC_HD void tc_peano_key(float px, float py, float pz, double box, uint64_t *hi, uint64_t *lo)
{
    uint64_t X[3];
    tc_scaled_coords(px, py, pz, box, X);
    tc_hilbert_transpose(X);
    tc_key_from_transpose(X, hi, lo);
}

/* Table-driven Peano key (tools/gen_hilbert_lut.py): the same 128-bit key as tc_peano_key, two octree levels per
 * table look-up instead of one bit-serial step per bit.  `lut` = TC_HILBERT_LUT (tc_hilbert_lut.h; the kernels keep
 * a copy in LDS).  A coordinate equal to the box size (X = 2^63, bit 63 set) takes the bit-serial transform: its
 * first step is not the start state of the tables. */
TC_HD void tc_key_lut_from_scaled(uint64_t X[3], const unsigned short *lut, uint64_t *hi, uint64_t *lo)
{
    if (((X[0] | X[1] | X[2]) >> 63) != 0) {
        tc_hilbert_transpose(X);
        tc_key_from_transpose(X, hi, lo);
        return;
    }
    /* the 42 levels the key holds are bits 62..21: W = X >> 21, pairs of levels from the top */
    const uint32_t h0 = (uint32_t)(X[0] >> 53), h1 = (uint32_t)(X[1] >> 53), h2 = (uint32_t)(X[2] >> 53);   /* bits 62..53: 5 pairs */
    const uint32_t l0 = (uint32_t)(X[0] >> 21), l1 = (uint32_t)(X[1] >> 21), l2 = (uint32_t)(X[2] >> 21);   /* bits 52..21: 16 pairs */
    uint32_t st = 0;                         /* state * 64 */
    uint64_t oa = 0, ob = 0;                 /* output triplets of levels 1..21 (63 bits) and 22..42 (63 bits) */
#define TC_HSTEP(c0, c1, c2, sh, dst, pos)                                                                     \
    {                                                                                                          \
        const uint32_t d = (((c0) >> (sh)) & 3u) << 4 | (((c1) >> (sh)) & 3u) << 2 | (((c2) >> (sh)) & 3u);    \
        const uint32_t e = lut[st | d];                                                                        \
        st = e & 0xffc0u;                                                                                      \
        dst |= (uint64_t)(e & 63u) << (pos);                                                                   \
    }
    /* levels 1..10 */
    TC_HSTEP(h0, h1, h2, 8, oa, 57) TC_HSTEP(h0, h1, h2, 6, oa, 51) TC_HSTEP(h0, h1, h2, 4, oa, 45)
    TC_HSTEP(h0, h1, h2, 2, oa, 39) TC_HSTEP(h0, h1, h2, 0, oa, 33)
    /* levels 11..20 */
    TC_HSTEP(l0, l1, l2, 30, oa, 27) TC_HSTEP(l0, l1, l2, 28, oa, 21) TC_HSTEP(l0, l1, l2, 26, oa, 15)
    TC_HSTEP(l0, l1, l2, 24, oa, 9) TC_HSTEP(l0, l1, l2, 22, oa, 3)
    /* levels 21 | 22: the pair straddles the two halves of the key */
    {
        const uint32_t d = ((l0 >> 20) & 3u) << 4 | ((l1 >> 20) & 3u) << 2 | ((l2 >> 20) & 3u);
        const uint32_t e = lut[st | d];
        st = e & 0xffc0u;
        oa |= (uint64_t)((e >> 3) & 7u);
        ob |= (uint64_t)(e & 7u) << 60;
    }
    /* levels 23..42 */
    TC_HSTEP(l0, l1, l2, 18, ob, 54) TC_HSTEP(l0, l1, l2, 16, ob, 48) TC_HSTEP(l0, l1, l2, 14, ob, 42)
    TC_HSTEP(l0, l1, l2, 12, ob, 36) TC_HSTEP(l0, l1, l2, 10, ob, 30) TC_HSTEP(l0, l1, l2, 8, ob, 24)
    TC_HSTEP(l0, l1, l2, 6, ob, 18) TC_HSTEP(l0, l1, l2, 4, ob, 12) TC_HSTEP(l0, l1, l2, 2, ob, 6)
    TC_HSTEP(l0, l1, l2, 0, ob, 0)
#undef TC_HSTEP
    /* key = (oa << 65) | (ob << 2), as tc_key_from_transpose */
    *hi = (oa << 1) | (ob >> 62);
    *lo = ob << 2;
}

TC_HD void tc_peano_key_lut(float px, float py, float pz, double box, const unsigned short *lut, uint64_t *hi, uint64_t *lo)
{
    uint64_t X[3];
    tc_scaled_coords(px, py, pz, box, X);
    tc_key_lut_from_scaled(X, lut, hi, lo);
}

/* Number of leading Hilbert levels (triplets, level 1 = bits 62 of X) two keys share.
 * Keys are the 128-bit keys above; level l occupies key bits [128-3l, 128-3l+2]. */
TC_HD int tc_common_levels(uint64_t ahi, uint64_t alo, uint64_t bhi, uint64_t blo)
{
    uint64_t xh = ahi ^ bhi, xl = alo ^ blo;
    int lz;
    if (xh) lz = __builtin_clzll(xh);
    else if (xl) lz = 64 + __builtin_clzll(xl);
    else return TC_NTRIPLETS;
    return lz / 3;
}

/* ------------------------------------------------------------------ SPH kernels */

/* sph.c:426-432: Wendland C6 with 1/h^3; args are f32, u=r/h is an f32 divide,
 * the polynomial runs in f64 left to right, the result is rounded to f32.
 * `norm_h3` must be  TC_WC6_NORM / (double)(h*h*h)  with h*h*h evaluated in f32. */
/* sph.c:442-466: the cubic spline of the -DSPH_CUBIC_SPLINE build, statement for statement (u = r/h is an f32
 * quotient promoted to double; the result is rounded to f32) */
TC_HD float tc_m4(float r, float h)
{
    double wk = 0;
    const double u = r / h;
    if (u < 0.5) wk = (2.546479089470 + 15.278874536822 * (u - 1) * u * u);
    else wk = 5.092958178941 * (1.0 - u) * (1.0 - u) * (1.0 - u);
    return (float)(wk / (double)(h * h * h));
}

TC_HD float tc_dm4(float r, float h)
{
    double dwk = 0;
    const double u = r / h;
    if (u < 0.5) dwk = u * (45.836623610466 * u - 30.557749073644);
    else dwk = (-15.278874536822) * (1.0 - u) * (1.0 - u);
    return (float)(dwk / (double)(h * h * h * h));
}

TC_HD float tc_wc6(float r, float h, double norm_h3)
{
    const double u = r / h;
    const double t = 1 - u;
    return (float)(norm_h3 * t * t * t * t * t * t * t * t * (1 + 8 * u + 25 * u * u + 32 * u * u * u));
}

/* sph.c:434-440: derivative; u stays f32, t=(double)(1-u) with the subtraction in f32,
 * the trailing polynomial (16u^2+7u+1) is evaluated in f32.
 * `norm_h4m22` must be  TC_WC6_NORM / (double)(h*h*h*h) * -22.0 . */
TC_HD float tc_dwc6(float r, float h, double norm_h4m22)
{
    const float u = r / h;
    const double t = 1 - u;
    return (float)(norm_h4m22 * t * t * t * t * t * t * t * u * (16 * u * u + 7 * u + 1));
}

/* wvt_relax.c:275-281: un-normalised WC6 used by the WVT sweep (returns double) */
TC_HD double tc_wvt_wc6(float r, float h)
{
    const double u = r / h;
    const double t = 1 - u;
    return TC_WC6_NORM * t * t * t * t * t * t * t * t * (1 + 8 * u + 25 * u * u + 32 * u * u * u);
}

/* Correctly rounded f32 quotient a/b for many a and one b without a per-element divide:
 * q0 = a*y, rem = fma(-q0, b, a) (exact), q1 = fma(rem, y, q0) with y = RN(1/b) (Markstein).
 * q1 == RN(a/b) unless b's significand is all ones or an intermediate leaves the normal range;
 * those cases fall back to the divide. */
typedef struct {
    float b, y;
    int exact_div;
} tc_fdiv;

TC_HD tc_fdiv tc_fdiv_setup(float b)
{
    tc_fdiv d;
    d.b = b;
    d.y = 1.0f / b;
    union { float f; uint32_t u; } c;
    c.f = b;
    d.exact_div = ((c.u & 0x7fffffu) == 0x7fffffu) || !(b > 1e-30f && b < 1e30f);
    return d;
}

TC_HD float tc_fdiv_apply(tc_fdiv d, float a)
{
    if (d.exact_div) return a / d.b;          /* wave-uniform: one b per wave */
    float q0 = a * d.y;
    float rem = __builtin_fmaf(-q0, d.b, a);
    return __builtin_fmaf(rem, d.y, q0);
}

/* the common case of tc_fdiv_apply, for callers that have branched on exact_div themselves */
TC_HD float tc_fdiv_apply_fast(tc_fdiv d, float a)
{
    float q0 = a * d.y;
    float rem = __builtin_fmaf(-q0, d.b, a);
    return __builtin_fmaf(rem, d.y, q0);
}

/* ------------------------------------------------------------------ density model */

typedef struct {
    double cx, cy, cz;   /* D_CoM + boxhalf is NOT pre-added: see tc_density_model */
    double rho0, beta, rcore, rcut;
    double mass_gas;
    double rho0_cc, rc_cc;   /* -DDOUBLE_BETA_COOL_CORES component of a cuspy halo (tcgpu_halo); rho0_cc == 0: none */
} tc_halo_dev;

/* setup.c:598-615.  The reference's default build has no cool-core term (rho0_cc == 0); its
 * -DDOUBLE_BETA_COOL_CORES build adds  rho0_cc / (1 + (r/rc_cc)^2) / (1 + (r/rcut)^4)  for halos with Is_Cuspy */
TC_HD double tc_gas_density_profile(double r, double rho0, double beta, double rc, double rcut, double rho0_cc,
                                    double rc_cc)
{
    double rho = rho0 * pow(1 + (r / rc) * (r / rc), -3.0 / 2.0 * beta)
                 / (1 + ((r / rcut) * (r / rcut) * (r / rcut)) * (r / rcut));
    if (rho0_cc != 0)
        rho += rho0_cc / (1 + (r / rc_cc) * (r / rc_cc)) / (1 + ((r / rcut) * (r / rcut) * (r / rcut)) * (r / rcut));
    return rho;
}

/* wvt_relax.c:227-256 */
TC_HD float tc_density_model(float px, float py, float pz, double boxhalf, const tc_halo_dev *halo, int nhalos)
{
    const double x = px, y = py, z = pz;
    double rho = 0;
    for (int i = 0; i < nhalos; i++) {
        if (halo[i].mass_gas == 0) continue;
        double dx = x - halo[i].cx - boxhalf;
        double dy = y - halo[i].cy - boxhalf;
        double dz = z - halo[i].cz - boxhalf;
        double r2 = dx * dx + dy * dy + dz * dz;
        double rho_i = tc_gas_density_profile(sqrt(r2), halo[i].rho0, halo[i].beta, halo[i].rcore, halo[i].rcut,
                                              halo[i].rho0_cc, halo[i].rc_cc);
        rho = fmax(rho_i, rho);
    }
    return (float)rho;
}

/* ------------------------------------------------------------------ ball-query predicate */

/* tree.c:67-89: f32, |d| folded at box/2, strict <.  Returns r2 (f32). */
TC_HD float tc_ngb_r2(float xi, float yi, float zi, float xj, float yj, float zj, float boxhalf, float boxsize)
{
    float dx = fabsf(xi - xj), dy = fabsf(yi - yj), dz = fabsf(zi - zj);
    if (dx > boxhalf) dx -= boxsize;
    if (dy > boxhalf) dy -= boxsize;
    if (dz > boxhalf) dz -= boxsize;
    return dx * dx + dy * dy + dz * dz;
}

/* sph.c:111-138: f64 separation with +-box/2 folding, returns r = sqrt(r2) */
TC_HD double tc_pair_r(float xi, float yi, float zi, float xj, float yj, float zj, double boxhalf, double boxsize)
{
    double dx = (double)xi - (double)xj, dy = (double)yi - (double)yj, dz = (double)zi - (double)zj;
    if (dx > boxhalf) dx -= boxsize;
    if (dx < -boxhalf) dx += boxsize;
    if (dy > boxhalf) dy -= boxsize;
    if (dy < -boxhalf) dy += boxsize;
    if (dz > boxhalf) dz -= boxsize;
    if (dz < -boxhalf) dz += boxsize;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

#endif /* TC_MATH_H */
