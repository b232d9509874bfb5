/*
 * kernels_ngb.hip -- the neighbour-gather kernels (one 64-lane wavefront per particle):
 *   K5  hsml / density solve      src/sph.c:13-214   (+ ball query src/tree.c:25-111)
 *   K9  WVT displacement sweep    src/wvt_relax.c:126-171
 *   K11 SPH curl of A             src/sph.c:216-300
 *
 * Ball query.  The reference walks its octree and tests leaf particles with an all-f32
 * predicate (src/tree.c:67-89).  Here the query picks the cell-table level whose cell edge s
 * satisfies s < h <= 2s, enumerates the <= 6^3 cells overlapping [x-h, x+h]^3 (periodic), drops
 * cells farther than h from the particle, and streams the contiguous particle run of every
 * remaining cell through the same f32 predicate, 64 candidates per step (coalesced float4
 * loads).  The result set equals the reference's; its order is not ascending, which only
 * permutes f64 summation order (DESIGN.md "Numerics").
 *
 * The gather is latency-bound, so it is split into a producer that only touches the cell table
 * and writes candidate INDICES into a per-wave LDS list (small cells expanded lane-per-cell,
 * large cells cooperatively), and a consumer that walks that flat list 256 candidates at a
 * time with four independent float4 gathers in flight per lane.
 *
 * Hits are compacted with ballot + mbcnt into a per-wave list holding the f64 pair distance r
 * (first TC_RCAP entries in LDS, the rest -- only cold-start lists get that long -- in a per-wave
 * global spill area), on which the reference's Newton-Raphson / bisection control flow then
 * runs with 64-lane partial sums and butterfly reductions.  Kernels are persistent: a fixed
 * grid of waves strides over the particles, so the spill area is bounded.
 */
#include <type_traits>
#include "tc_ctx.h"
#include "tc_lean.h"
#include "tc_hilbert_lut.h"

/* Stage ablation for instruction-count profiling (tools/ablate_iter.py) is compiled in only with
 * -DTC_PROFILE_ABLATE (make ablate -> ../lib/libtcgpu_ablate.so); the product kernels carry no such branches. */
#ifdef TC_PROFILE_ABLATE
#define TC_ABLATE(k) ((k).ablate)
#else
#define TC_ABLATE(k) 0
#endif

/* Stage timers (tools/stage_times.py; make stages -> ../lib/libtcgpu_stages.so): every wave of k_iter accumulates the
 * shader cycles (s_memtime) it spends in each stage of the per-particle body.  The four waves of a SIMD interleave,
 * so a stage's share of a wave's life is its share of the launch -- issue slots and stalls alike.  Transitions name
 * both stages so that the accumulator index is a compile-time constant.  Not compiled into the product. */
#ifdef TC_PROFILE_STAGES
#define TC_NSTAGE 10
struct tc_prof { uint64_t last; uint32_t acc[TC_NSTAGE]; };
#define TC_PROF_PARAM , tc_prof *prof__ = nullptr
#define TC_PROF_PARAM_DEF , tc_prof *prof__
#define TC_PROF_PASS , prof__
#define TC_STAGE_SWITCH(from, to)                                                          \
    do {                                                                                   \
        if (prof__) {                                                                      \
            const uint64_t now__ = __builtin_amdgcn_s_memtime();                           \
            prof__->acc[from] += (uint32_t)(now__ - prof__->last);                         \
            prof__->last = now__;                                                          \
        }                                                                                  \
    } while (0)
#else
#define TC_PROF_PARAM
#define TC_PROF_PARAM_DEF
#define TC_PROF_PASS
#define TC_STAGE_SWITCH(from, to) do { } while (0)
#endif
enum { ST_PROLOGUE = 0, ST_PRODUCER, ST_WINDOW, ST_TEST, ST_CONVERT_D, ST_CONVERT_W, ST_SOLVE_PAIRS, ST_SOLVE_UNIFORM,
       ST_EPILOGUE, ST_QUEUE };

#define WPB TC_WAVES_PER_BLOCK
#define TBN (WPB * 64)

/* ------------------------------------------------------------------ wave helpers */

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

/* mark a value the program knows to be wave-uniform as such (keeps it in SGPRs, scalar branches) */
__device__ __forceinline__ int U(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t U(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float U(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

/* lane mask of a predicate; the builtin uses the compare's own SGPR pair (HIP's __ballot first turns the
 * predicate into 0/1 in a VGPR and compares again: two extra VALU instructions per call) */
__device__ __forceinline__ uint64_t tc_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ int mask_rank(uint64_t m) /* set bits below this lane */
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

/* A wave-uniform pointer parked in vector registers.  The kernel runs out of scalar registers and the compiler
 * spills them into VGPR lanes, paying two v_readlane per use of a 64-bit pointer in the hot loops; a pointer
 * that already lives in a VGPR pair feeds the address arithmetic directly. */
#define TC_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const TC_GLOBAL T *vgpr_ptr(const T *p)
{
    uint64_t u = (uint64_t)p;
    uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
    asm volatile("v_mov_b32 %0, %0" : "+v"(lo));
    asm volatile("v_mov_b32 %0, %0" : "+v"(hi));
    /* device memory: keep the global address space so that the loads are global_load, not flat_load */
    return (const TC_GLOBAL T *)(((uint64_t)hi << 32) | lo);
}

/* float4 loads through such a pointer (HIP's float4 class has no address-space-qualified copy) */
typedef float tc_f4n __attribute__((ext_vector_type(4)));
typedef const TC_GLOBAL tc_f4n *tc_gpos;
__device__ __forceinline__ tc_gpos vgpr_pos(const float4 *p) { return (tc_gpos)vgpr_ptr(reinterpret_cast<const tc_f4n *>(p)); }
__device__ __forceinline__ float4 ld4(tc_gpos p, uint32_t j)
{
    const tc_f4n v = p[j];
    return make_float4(v.x, v.y, v.z, v.w);
}

/* x, y, z only (12 of the 16 bytes of a slot): the candidate test of the fused kernel never looks at w, and the four
 * gathers in flight then hold 12 instead of 16 VGPRs */
typedef float tc_f3n __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float4 ld3(tc_gpos p, uint32_t j)
{
    const tc_f3n v = *reinterpret_cast<const TC_GLOBAL tc_f3n *>(p + j);
    return make_float4(v.x, v.y, v.z, 0.0f);
}

__device__ __forceinline__ uint32_t vgpr_u32(uint32_t v)
{
    asm volatile("v_mov_b32 %0, %0" : "+v"(v));
    return v;
}

/* min of two finite doubles in one instruction (fmin() adds a canonicalising v_max per operand) */
__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double max_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* wave-uniform copy of a double (lets the compiler keep dependent control flow scalar) */
__device__ __forceinline__ double bcast0(double v)
{
    uint64_t u = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

/* one DPP step of a 64-bit value over all rows: lanes without a source receive +0.0 (bound_ctrl) */
#define TC_DPP_F64(v, ctrl)                                                                                  \
    __builtin_bit_cast(double,                                                                               \
        ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(__builtin_bit_cast(uint64_t, v) >> 32), \
                                                         ctrl, 0xf, 0xf, true) << 32)                       \
        | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)__builtin_bit_cast(uint64_t, v), ctrl, 0xf, 0xf, true))

/* wave sum in registers (DPP row shifts + row broadcasts); the total lands in lane 63 and is returned
 * wave-uniform.  A reduction, not a scan: only lane 63's chain matters, ((row3 + row2) + (row1 + row0)) of the
 * per-row totals, so every step may run on all rows (no row masks, no zero-initialised temporaries).
 * Fixed summation tree => bitwise reproducible run to run. */
__device__ __forceinline__ double wsum(double v)
{
    v += TC_DPP_F64(v, 0x111);   /* row_shr:1 */
    v += TC_DPP_F64(v, 0x112);   /* row_shr:2 */
    v += TC_DPP_F64(v, 0x114);   /* row_shr:4 */
    v += TC_DPP_F64(v, 0x118);   /* row_shr:8  -> lane 15 of every row holds the row total */
    v += TC_DPP_F64(v, 0x142);   /* row_bcast:15 -> lane 63: rows 3+2, lane 31: rows 1+0 */
    v += TC_DPP_F64(v, 0x143);   /* row_bcast:31 -> lane 63: all four rows */
    uint64_t u = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

/* Two wave sums for the price of one: v_permlane32_swap (gfx950) puts a's two half-wave partials side by side in lanes
 * 0..31 and b's in lanes 32..63, one add folds them, and the row reduction of the single sum then finishes both at
 * once -- a's total in lane 31, b's in lane 63.  22 instructions instead of 40.  Fixed tree => reproducible. */
__device__ __forceinline__ void wsum2(double a, double b, double &sa, double &sb)
{
    const uint64_t ua = __builtin_bit_cast(uint64_t, a), ub = __builtin_bit_cast(uint64_t, b);
    /* upper half of the first operand <-> lower half of the second */
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)ua, (uint32_t)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32), false, false);
    const double x = __builtin_bit_cast(double, ((uint64_t)hi[0] << 32) | lo[0]);   /* a[l]      | b[l - 32] */
    const double y = __builtin_bit_cast(double, ((uint64_t)hi[1] << 32) | lo[1]);   /* a[l + 32] | b[l]      */
    double v = x + y;
    v += TC_DPP_F64(v, 0x111);   /* row_shr:1 */
    v += TC_DPP_F64(v, 0x112);   /* row_shr:2 */
    v += TC_DPP_F64(v, 0x114);   /* row_shr:4 */
    v += TC_DPP_F64(v, 0x118);   /* row_shr:8  -> lane 15 of every row holds the row total */
    v += TC_DPP_F64(v, 0x142);   /* row_bcast:15 -> lane 31: rows 1+0 (a), lane 63: rows 3+2 (b) */
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 31);
    const uint32_t ahi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 31);
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    sa = __builtin_bit_cast(double, ((uint64_t)ahi << 32) | alo);
    sb = __builtin_bit_cast(double, ((uint64_t)bhi << 32) | blo);
}

/* Inclusive prefix sums over the 64 lanes with fused DPP adds (x += x[lane - k]; masked rows keep x): six VALU
 * instructions per scan.  Written in assembly because the compiler expands the builtin form into
 * zero-init + v_mov_dpp + add per step.  A DPP read of a VGPR needs two wait states after the VALU write of
 * it; the hazard recogniser does not see inside inline assembly, so the single scan carries s_nop 1 and the
 * four-way version interleaves four independent scans (three instructions between dependent ones).
 * Requires all 64 lanes active (callers are in wave-uniform control flow). */
#define TC_SCAN_STEP(r, ctl) "v_add_u32_dpp " r ", " r ", " r " " ctl "\n"
#define TC_SHR1 "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define TC_SHR2 "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define TC_SHR4 "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define TC_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define TC_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define TC_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    asm volatile("s_nop 1\n"
                 TC_SCAN_STEP("%0", TC_SHR1) "s_nop 1\n"
                 TC_SCAN_STEP("%0", TC_SHR2) "s_nop 1\n"
                 TC_SCAN_STEP("%0", TC_SHR4) "s_nop 1\n"
                 TC_SCAN_STEP("%0", TC_SHR8) "s_nop 1\n"
                 TC_SCAN_STEP("%0", TC_BC15) "s_nop 1\n"
                 TC_SCAN_STEP("%0", TC_BC31)
                 : "+v"(v));
    return v;
}

__device__ __forceinline__ void wave_incl_scan4(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
#define TC_SCAN4(ctl) TC_SCAN_STEP("%0", ctl) TC_SCAN_STEP("%1", ctl) TC_SCAN_STEP("%2", ctl) TC_SCAN_STEP("%3", ctl)
    asm volatile("s_nop 1\n"
                 TC_SCAN4(TC_SHR1) TC_SCAN4(TC_SHR2) TC_SCAN4(TC_SHR4) TC_SCAN4(TC_SHR8) TC_SCAN4(TC_BC15) TC_SCAN4(TC_BC31)
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef TC_SCAN4
}

/* LDS written by some lanes is read by others of the same wave: keep the compiler from
 * moving DS operations across this point (the hardware executes a wave's DS ops in order). */
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ------------------------------------------------------------------ query geometry */

struct tc_query {
    int L, nL;            /* level, cells per dimension */
    int lo[3], nd[3];     /* first cell (unwrapped, may be negative) and cell count per dim */
    bool full[3];         /* the dimension covers the whole ring: no culling there */
    float sf, hpf;        /* cell edge and padded radius (f32 copies for the per-cell culling) */
    float inv_ny, inv_sf;
    uint32_t off;         /* table offset of the level's box */
    int ox, oy, oz, ny, nz;   /* ... its origin and the strides of its (x, y, z) layout (tc_level_desc) */
};

/* table level of a query of radius h (tc_query_level, tc_ctx.h), clamped to the levels this pass built */
__device__ __forceinline__ int query_level(const tc_dev_const &k, float h)
{
    return tc_query_level(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift, k.lmin_tab, k.lmax, h);
}

/* The constants as ONE particle's queries see them: on a sharded pass every query of the particle is held to the
 * table levels the marking kernel sized the boxes for (tc_particle_levels; h0 = the carried smoothing length the pass
 * started from, w = the WVT hsml), on a full set to the whole table. */
/* UNIFORM: the arguments are wave-uniform (one particle per wavefront) and the results are pinned to scalar registers;
 * otherwise every lane works on a particle of its own (k_prec) */
template <bool UNIFORM = true>
__device__ __forceinline__ tc_dev_const particle_view(const tc_dev_const &k, float h0, float w)
{
    tc_dev_const kk = k;
    if (k.margin_on) {
        const float rg = tc_margin_radius(h0, w, k.boxsize, k.margin_widen);
        int la, lb;
        tc_particle_levels(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift, k.lmax, h0, rg, &la, &lb);
        kk.lmin_tab = UNIFORM ? U(la) : la;
        kk.lmax = UNIFORM ? U(lb) : lb;
    }
    return kk;
}

__device__ __forceinline__ double query_cell_edge_at(const tc_dev_const &k, int L)
{
    return __builtin_ldexp(k.boxsize, -L);                                 /* box / 2^L, exact */
}

template <bool UNIFORM = true>
__device__ __forceinline__ void query_setup(const tc_dev_const &k, float xi, float yi, float zi, float h, tc_query &q)
{
#define UU(v) (UNIFORM ? U(v) : (v))
    const int L = query_level(k, h);
    q.L = L;
    q.nL = 1 << L;
    const double s = query_cell_edge_at(k, L);
    const double inv_s = (double)q.nL * k.boxinv;
    /* layout of the level's box (wave-uniform: scalar loads) */
    const tc_level_desc D = k.lvl[L];
    q.off = UU(D.off);
    q.ox = UU(D.ox); q.oy = UU(D.oy); q.oz = UU(D.oz); q.ny = UU(D.ny); q.nz = UU(D.nz);
    const int bo[3] = {q.ox, q.oy, q.oz}, bn[3] = {UU(D.nx), q.ny, q.nz};
#undef UU
    /* pad: the f32 predicate can accept pairs a few ulp beyond h, and the per-cell culling below
     * runs in f32 on coordinates of magnitude boxsize (absolute error < 3e-7 boxsize) */
    const double hp = (double)h * (1.0 + 1e-5) + k.boxsize * 2e-6;
    q.sf = (float)s;
    q.hpf = (float)hp;
    const float xs[3] = {xi, yi, zi};
    const bool huge = !(hp < k.boxsize);      /* ball wider than the box: take every cell once */
    for (int d = 0; d < 3; d++) {
        int lo = 0, nd = q.nL;
        if (!huge) {
            lo = (int)floor(((double)xs[d] - hp) * inv_s);     /* margins in hp absorb the rounding */
            int hi = (int)floor(((double)xs[d] + hp) * inv_s);
            /* a box narrower than the ring holds every cell a permitted query reaches (marking kernel, same
             * padding); clamping to it is then a no-op that keeps any other query inside the table */
            if (bn[d] < q.nL) {
                if (lo < bo[d]) lo = bo[d];
                if (hi > bo[d] + bn[d] - 1) hi = bo[d] + bn[d] - 1;
            }
            nd = hi - lo + 1;
            if (nd < 0) nd = 0;
        }
        q.full[d] = false;
        if (nd >= q.nL) { nd = q.nL; lo = 0; q.full[d] = true; }
        q.lo[d] = lo; q.nd[d] = nd;
    }
    q.inv_ny = 1.0f / (float)(q.nd[1] > 0 ? q.nd[1] : 1);
    q.inv_sf = (float)inv_s;
}

/* Row (a, b) of the query block = the cells sharing one (x, y) cell coordinate.  Returns the x/y part of
 * the table index, the first z cell (relative to q.lo[2]) and the number of consecutive z cells the ball
 * can reach in this row (0 if the row is farther than the padded radius).  Conservative by construction:
 * every cell with a point within the padded radius is inside the interval of its row. */
__device__ __forceinline__ void query_row(const tc_query &q, float xi, float yi, float zi, int rw, uint32_t &rowlin,
                                          int &c0, int &len)
{
    const int a = (int)(((float)rw + 0.5f) * q.inv_ny);        /* rw / nd[1] without an integer divide */
    const int b = rw - a * q.nd[1];
    const int off[2] = {a, b};
    const float xs[2] = {xi, yi};
    float g2 = 0;
    int cxy[2];
    for (int d = 0; d < 2; d++) {
        int u = q.lo[d] + off[d];
        if (!q.full[d]) {
            float clo = (float)u * q.sf, chi = clo + q.sf;
            float g = xs[d] < clo ? clo - xs[d] : (xs[d] > chi ? xs[d] - chi : 0.0f);
            g2 += g * g;
        }
        cxy[d] = u & (q.nL - 1);
    }
    /* table entry of the row's cell z is rowlin + (z mod 2^L): the box origin is folded in here */
    rowlin = q.off + (uint32_t)(((cxy[0] - q.ox) * q.ny + (cxy[1] - q.oy)) * q.nz - q.oz);
    c0 = 0;
    len = 0;
    const float rem = q.hpf * q.hpf - g2;
    if (rem < 0) return;
    if (q.full[2]) { len = q.nd[2]; return; }
    /* z reach of the ball in this row; +-1e-3 cell covers the f32 rounding of coordinate/cell-edge */
    /* the hardware root (1 ulp) is plenty under the 1e-5 padding; the interval only decides which cells are visited */
    const float dz = __builtin_amdgcn_sqrtf(rem) * 1.00001f;
    int zlo = (int)floorf((zi - dz) * q.inv_sf - 1e-3f) - q.lo[2];
    int zhi = (int)floorf((zi + dz) * q.inv_sf + 1e-3f) - q.lo[2];
    if (zlo < 0) zlo = 0;
    if (zhi > q.nd[2] - 1) zhi = q.nd[2] - 1;
    if (zhi < zlo) return;
    c0 = zlo;
    len = zhi - zlo + 1;
}

__device__ __forceinline__ bool is_orphan(const tc_dev_const &k, float4 p)
{
    return (double)p.x >= k.boxsize || (double)p.y >= k.boxsize || (double)p.z >= k.boxsize;
}

/* tc_ngb_r2 / tc_pair_r with the minimum-image folding under a wave-uniform flag: when every
 * candidate of a particle provably lies within box/2 per coordinate (no wrapped cell is visited),
 * none of the reference's `> boxhalf` tests can fire and the folding is skipped. */
__device__ __forceinline__ float ngb_r2_w(float xi, float yi, float zi, float xj, float yj, float zj, float boxhalf,
                                          float boxsize, bool wrap)
{
    float dx = fabsf(xi - xj), dy = fabsf(yi - yj), dz = fabsf(zi - zj);
    if (wrap) {
        if (dx > boxhalf) dx -= boxsize;
        if (dy > boxhalf) dy -= boxsize;
        if (dz > boxhalf) dz -= boxsize;
    }
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ double pair_r_w(float xi, float yi, float zi, float xj, float yj, float zj, double boxhalf,
                                           double boxsize, bool wrap)
{
    double dx = (double)xi - (double)xj, dy = (double)yi - (double)yj, dz = (double)zi - (double)zj;
    if (wrap) {
        if (dx > boxhalf) dx -= boxsize;
        if (dx < -boxhalf) dx += boxsize;
        if (dy > boxhalf) dy -= boxsize;
        if (dy < -boxhalf) dy += boxsize;
        if (dz > boxhalf) dz -= boxsize;
        if (dz < -boxhalf) dz += boxsize;
    }
    /* a sum of squares of f32 differences is 0 or >= 1e-90: far above the 2^-767 where the IEEE root starts
     * to rescale, so the unscaled core returns the same bits (tc_lean.h).  Zero (the particle itself) becomes
     * 1e-150 instead of a select on the result: its only uses are (float)r == 0 and r * dwk with dwk(u = 0) == -0,
     * which give the same bits as r == 0 (solve_hsml) */
    return tc_sqrt_f64_lean_pos(max_f64(dx * dx + dy * dy + dz * dz, 1e-300));
}

/* cell edge the query of radius h will use */
__device__ __forceinline__ double query_cell_edge(const tc_dev_const &k, float h)
{
    return query_cell_edge_at(k, query_level(k, h));
}

/* Consumer: walk the flat index list, four independent gathers in flight per lane. */
template <class Body>
__device__ __forceinline__ bool consume_candidates(const tc_dev_const &k, const uint32_t *idx, int fill, int norph,
                                                   Body &body)
{
    const int lane = lane_id();
    if (TC_ABLATE(k) == 1) return false;
    for (int c0 = 0; c0 < fill; c0 += 256) {
        uint32_t j[4];
        float4 p[4];
        bool act[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int t = c0 + 64 * u + lane;
            act[u] = t < fill;
            j[u] = act[u] ? idx[t] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = k.pos4[j[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (c0 + 64 * u < fill) {
                bool a = act[u];
                if (norph && a && is_orphan(k, p[u])) a = false;
                if (body((int)j[u], p[u], a)) return true;
            }
        }
    }
    return false;
}

/*
 * Stream every candidate of the ball (xi,h) through `body(j, p, active)`; body is called
 * wave-uniformly (all 64 lanes, `active` false for padding lanes) and returns true to stop.
 * `idx` is this wave's LDS index list (TC_IDXCAP entries).  Returns the number of candidates.
 */
template <class Body>
__device__ __forceinline__ uint32_t stream_candidates_q(const tc_dev_const &k, const tc_query &q, float xi, float yi, float zi,
                                                        uint32_t *idx, uint32_t idxcap, Body &&body);

template <class Body>
__device__ __forceinline__ uint32_t stream_candidates(const tc_dev_const &k, float xi, float yi, float zi, float h,
                                                      uint32_t *idx, uint32_t idxcap, Body &&body)
{
    tc_query q;
    query_setup(k, xi, yi, zi, h, q);
    return stream_candidates_q(k, q, xi, yi, zi, idx, idxcap, body);
}

/* ... with the query geometry already worked out (k_prec) */
template <class Body>
__device__ __forceinline__ uint32_t stream_candidates_q(const tc_dev_const &k, const tc_query &q, float xi, float yi, float zi,
                                                        uint32_t *idx, uint32_t idxcap, Body &&body)
{
    const int lane = lane_id();
    int norph = *k.norph;
    if (norph > TC_MAX_ORPHANS) norph = TC_MAX_ORPHANS;   /* overflow is flagged by k_cells */
    uint32_t ncand = 0;
    uint32_t fill = 0;

    /* Cells are enumerated row by row: one lane per (x, y) row computes the z interval the ball reaches
     * (no cell outside the ball's bounding cylinder slices is ever touched), a prefix sum over the rows
     * numbers the cells, and each lane of a 64-cell batch finds its row by bisection over that prefix
     * held in the lanes themselves (ds_bpermute). */
    const int nrow = q.nd[0] * q.nd[1];
    for (int rbase = 0; rbase < nrow; rbase += 64) {
        uint32_t rowlin = 0;
        int rc0 = 0, rlen = 0;
        if (rbase + lane < nrow) query_row(q, xi, yi, zi, rbase + lane, rowlin, rc0, rlen);
        const uint32_t rincl = wave_incl_scan((uint32_t)rlen);
        const uint32_t rexcl = rincl - (uint32_t)rlen;
        const int total = __builtin_amdgcn_readlane((int)rincl, 63);
        for (int base = 0; base < total; base += 64) {
            uint32_t st = 0, en = 0;
            {
                const uint32_t m = (uint32_t)(base + lane);
                int lo_r = 0, hi_r = 63;                 /* smallest lane r with rincl[r] > m */
    #pragma unroll
                for (int sstep = 0; sstep < 6; sstep++) {
                    const int mid = (lo_r + hi_r) >> 1;
                    const uint32_t v = __shfl(rincl, mid);
                    if (v > m) hi_r = mid; else lo_r = mid + 1;
                }
                const uint32_t rl = __shfl(rowlin, lo_r);
                const int zc = __shfl(rc0, lo_r) + (int)(m - __shfl(rexcl, lo_r));
                if (m < (uint32_t)total) {
                    const uint32_t lin = rl + (uint32_t)((q.lo[2] + zc) & (q.nL - 1));
                    const uint2 ce = k.cells[lin];                  /* {~first, last+1}, both 0 when empty */
                    const uint32_t s0 = ~ce.x, e0 = ce.y;
                    if (e0 > s0) { st = s0; en = e0; }
                }
            }
            const uint32_t cnt = en - st;

            /* large cells: the whole wave writes the run of consecutive indices */
            uint64_t big = tc_ballot(cnt > TC_SMALLCELL);
            while (big) {
                int l = __builtin_ctzll(big);
                big &= big - 1;
                uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)st, l);
                uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l);
                ncand += c0;
                uint32_t done = 0;
                while (done < c0) {
                    uint32_t take = min(c0 - done, idxcap - fill);
                    for (uint32_t t = lane; t < take; t += 64) idx[fill + t] = s0 + done + t;
                    fill += take;
                    done += take;
                    if (fill == idxcap) {
                        wave_lds_fence();
                        if (consume_candidates(k, idx, (int)fill, norph, body)) return ncand;
                        wave_lds_fence();
                        fill = 0;
                    }
                }
            }

            /* small cells: one lane expands one cell at its prefix-sum offset */
            uint32_t pend = (cnt <= TC_SMALLCELL) ? cnt : 0;
            while (tc_ballot(pend > 0)) {
                uint32_t incl = wave_incl_scan(pend);
                uint32_t excl = incl - pend;
                bool ok = pend > 0 && incl <= idxcap - fill;
                if (ok)
                    for (uint32_t t = 0; t < pend; t++) idx[fill + excl + t] = st + t;
                uint64_t okm = tc_ballot(ok);
                uint32_t emitted = 0;
                if (okm) emitted = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63 - __builtin_clzll(okm));
                fill += emitted;
                ncand += emitted;
                if (ok) pend = 0;
                if (tc_ballot(pend > 0)) {                       /* list full: drain it, then go on */
                    wave_lds_fence();
                    if (consume_candidates(k, idx, (int)fill, norph, body)) return ncand;
                    wave_lds_fence();
                    fill = 0;
                }
            }
        }
    }
    if (fill) {
        wave_lds_fence();
        if (consume_candidates(k, idx, (int)fill, norph, body)) return ncand;
        wave_lds_fence();
    }
    for (int o0 = 0; o0 < norph; o0 += 64) {        /* orphans: brute force */
        int o = o0 + lane;
        bool act = o < norph;
        uint32_t j = k.orphans[act ? o : 0];
        float4 p = k.pos4[j];
        if (body((int)j, p, act)) return ncand;
    }
    return ncand + norph;
}

/*
 * Row-run streaming over the row-major mirror (tc_dev_const::cum / mirror): the fast path of the fused
 * kernel for a ball that does not touch the periodic boundary (every cell coordinate it reaches lies in
 * [0, 2^L), so a row's z interval is one run of consecutive table entries) at a mirrored level.
 *
 * One lane per (x, y) row computes the z interval as query_row() does and reads the two ends of the run
 * from cum[]; a prefix sum over the run lengths numbers the candidates.  Flat candidate m of run r sits
 * in slot m + (a_r - excl_r).  That offset is piecewise constant in m, so it is recovered with one more
 * prefix sum: every run adds its jump (a_r - b_{r-1}) at its first flat position in a 256-entry LDS
 * window (`heads`, ds_add so that empty runs sharing a position accumulate), and the running sum of the
 * window is the offset -- about a dozen instructions per 64 candidates, no per-cell table reads, no
 * per-lane loops.  body(slot, p, active) as in stream_candidates, but `slot` indexes the mirror.
 */
template <bool XYZ_ONLY = false, class Body>
__device__ __forceinline__ uint32_t stream_rows_q(const tc_dev_const &k, const tc_query &q, float xi, float yi, float zi,
                                                  uint32_t *heads, Body &&body TC_PROF_PARAM);

template <bool XYZ_ONLY = false, class Body>
__device__ __forceinline__ uint32_t stream_rows(const tc_dev_const &k, float xi, float yi, float zi, float h,
                                                uint32_t *heads, Body &&body TC_PROF_PARAM)
{
    tc_query q;
    query_setup(k, xi, yi, zi, h, q);
    return stream_rows_q<XYZ_ONLY>(k, q, xi, yi, zi, heads, body TC_PROF_PASS);
}

template <bool XYZ_ONLY, class Body>
__device__ __forceinline__ uint32_t stream_rows_q(const tc_dev_const &k, const tc_query &q, float xi, float yi, float zi,
                                                  uint32_t *heads, Body &&body TC_PROF_PARAM_DEF)
{
    const int lane = lane_id();
    TC_STAGE_SWITCH(ST_PROLOGUE, ST_PRODUCER);
    const uint32_t *cum = k.cum;                      /* indexed with table entries (rowlin carries the level's offset) */
    const tc_gpos mirror = vgpr_pos(k.mirror);
    const uint32_t padslot = vgpr_u32(k.mirror_pad);
    const int nrow = q.nd[0] * q.nd[1];
    uint32_t ncand = 0;
    for (int rbase = 0; rbase < nrow; rbase += 64) {
        uint32_t rowlin = 0;
        int rc0 = 0, rlen = 0;
        if (rbase + lane < nrow) query_row(q, xi, yi, zi, rbase + lane, rowlin, rc0, rlen);
        uint32_t ra = 0, rb = 0;
        if (rlen > 0) {
            const uint32_t lin0 = rowlin + (uint32_t)(q.lo[2] + rc0);
            ra = cum[lin0];
            rb = cum[lin0 + (uint32_t)rlen];
        }
        const uint32_t cnt = rb - ra;
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t bprev = (uint32_t)__shfl_up((int)rb, 1);
        if (lane == 0) bprev = 0;
        const uint32_t jump = ra - bprev;                 /* mod 2^32; the running sum telescopes to a_r - excl_r */
        ncand += total;
        if (TC_ABLATE(k) == 1) continue;
        /* Mirror slots of the 256 flat candidates [base, base + 256): window of run jumps + running sum.  The slots
         * of window w + 1 are worked out while the gathers of window w are in flight (they only need LDS and DPP),
         * so a good part of the gather latency each window used to wait out is spent on work that has to be done anyway. */
        uint32_t carry = 0;
        auto slots = [&](uint32_t base, uint32_t (&j)[4]) {
            reinterpret_cast<uint4 *>(heads)[lane] = make_uint4(0, 0, 0, 0);
            wave_lds_fence();
            if (excl >= base && excl < base + 256 && excl < total) atomicAdd(&heads[excl - base], jump);
            wave_lds_fence();
            uint32_t hsum[4];
#pragma unroll
            for (int u = 0; u < 4; u++) hsum[u] = heads[64 * u + lane];
            wave_incl_scan4(hsum[0], hsum[1], hsum[2], hsum[3]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t sc = hsum[u] + carry;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)hsum[u], 63);
                const uint32_t m = base + 64 * u + lane;
                j[u] = m < total ? m + sc : padslot;            /* padding lanes: the slot at infinity */
            }
            wave_lds_fence();
        };
        uint32_t j[4], jn[4];
        TC_STAGE_SWITCH(ST_PRODUCER, ST_WINDOW);
        if (total > 0) slots(0, j);
        TC_STAGE_SWITCH(ST_WINDOW, ST_PRODUCER);
        for (uint32_t base = 0; base < total; base += 256) {
            TC_STAGE_SWITCH(ST_PRODUCER, ST_WINDOW);
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = XYZ_ONLY ? ld3(mirror, j[u]) : ld4(mirror, j[u]);
            if (base + 256 < total) slots(base + 256, jn);
            TC_STAGE_SWITCH(ST_WINDOW, ST_TEST);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (base + 64 * u < total) {
                    if (body(j[u], p[u], true)) return ncand;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) j[u] = jn[u];
            TC_STAGE_SWITCH(ST_TEST, ST_PRODUCER);
        }
    }
    TC_STAGE_SWITCH(ST_PRODUCER, ST_EPILOGUE);
    return ncand;
}

/* The same streaming with the rows replaced by a particle's ordered index runs (k_xruns: the cells of the ball in CURVE
 * order, each a contiguous run of the Peano-sorted positions themselves): the flat candidate number then ascends with
 * the particle index, so everything downstream -- hit lists, staged sweep neighbours -- comes out in ascending index,
 * the order the reference's lists have (src/tree.c:25-111).  No mirror, no per-row geometry in this kernel. */
template <class Body>
__device__ __forceinline__ uint32_t stream_runs(const tc_dev_const &k, const uint2 *__restrict__ pruns, int nruns,
                                               uint32_t *heads, Body &&body TC_PROF_PARAM)
{
    const int lane = lane_id();
    TC_STAGE_SWITCH(ST_PROLOGUE, ST_PRODUCER);
    const tc_gpos posv = vgpr_pos(k.pos4);
    const uint32_t padslot = vgpr_u32(k.pos_pad);
    uint32_t ncand = 0;
    for (int rbase = 0; rbase < nruns; rbase += 64) {
        uint32_t ra = 0, rb = 0;
        if (rbase + lane < nruns) { const uint2 r = pruns[rbase + lane]; ra = r.x; rb = r.y; }
        const uint32_t cnt = rb - ra;
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t excl = incl - cnt;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t bprev = (uint32_t)__shfl_up((int)rb, 1);
        if (lane == 0) bprev = 0;
        const uint32_t jump = ra - bprev;
        ncand += total;
        uint32_t carry = 0;
        auto slots = [&](uint32_t base, uint32_t (&j)[4]) {
            reinterpret_cast<uint4 *>(heads)[lane] = make_uint4(0, 0, 0, 0);
            wave_lds_fence();
            if (excl >= base && excl < base + 256 && excl < total) atomicAdd(&heads[excl - base], jump);
            wave_lds_fence();
            uint32_t hsum[4];
#pragma unroll
            for (int u = 0; u < 4; u++) hsum[u] = heads[64 * u + lane];
            wave_incl_scan4(hsum[0], hsum[1], hsum[2], hsum[3]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t sc = hsum[u] + carry;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)hsum[u], 63);
                const uint32_t m = base + 64 * u + lane;
                j[u] = m < total ? m + sc : padslot;
            }
            wave_lds_fence();
        };
        uint32_t j[4], jn[4];
        TC_STAGE_SWITCH(ST_PRODUCER, ST_WINDOW);
        if (total > 0) slots(0, j);
        TC_STAGE_SWITCH(ST_WINDOW, ST_PRODUCER);
        for (uint32_t base = 0; base < total; base += 256) {
            TC_STAGE_SWITCH(ST_PRODUCER, ST_WINDOW);
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = ld3(posv, j[u]);
            if (base + 256 < total) slots(base + 256, jn);
            TC_STAGE_SWITCH(ST_WINDOW, ST_TEST);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (base + 64 * u < total) {
                    if (body(j[u], p[u], true)) return ncand;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) j[u] = jn[u];
            TC_STAGE_SWITCH(ST_TEST, ST_PRODUCER);
        }
    }
    TC_STAGE_SWITCH(ST_PRODUCER, ST_EPILOGUE);
    return ncand;
}

/* Persistent-grid work assignment: a dynamic queue, XCD-aware.
 *
 * The instruction arbiter favours the oldest wave of a SIMD, so with a static assignment the four waves of a
 * SIMD finish one after the other (the first at 56 % of the launch time, measured) and the SIMD spends the last
 * 40 % of the launch under-occupied.  Here every wave keeps pulling small chunks of particles until the pool is
 * empty, so all waves end within one chunk of each other (k_iter: -19 %).
 * Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the group that shares an L2); each
 * group drains its own contiguous eighth of the Peano range first -- the candidates one L2 has to hold are then
 * one compact region of space instead of an eighth of everything -- and then helps the other eighths.
 * Which wave solves which particle affects speed only, never results.  The launcher zeroes the counters. */
#define TC_WORK_CHUNK 4         /* particles a wave takes from the queue at a time */

template <class F>
__device__ __forceinline__ void work_queue(const tc_dev_const &k, F &&f)
{
    const int lo = k.lo, hi = k.hi;
    const bool grouped = gridDim.x >= 16 && (gridDim.x & 7) == 0;
    const int ngroups = grouped ? 8 : 1;
    const int glen = grouped ? ((hi - lo + 7) >> 3) : (hi - lo);
    const int g0 = grouped ? (int)(blockIdx.x & 7) : 0;
    for (int gg = 0; gg < ngroups; gg++) {
        const int grp = (g0 + gg) & (ngroups - 1);
        const int gstart = lo + grp * glen;
        int gend = gstart + glen;
        if (gend > hi) gend = hi;
        if (gend <= gstart) continue;
        for (;;) {
            uint32_t got = 0;                       /* unsigned: the counter keeps growing after the range is empty */
            if ((threadIdx.x & 63) == 0)
                got = atomicAdd(reinterpret_cast<unsigned int *>(&k.work_ctr[16 * grp]), (unsigned int)TC_WORK_CHUNK);
            got = U(got);
            if (got >= (uint32_t)(gend - gstart)) break;
            const int base = gstart + (int)got;
            const int stop = base + TC_WORK_CHUNK < gend ? base + TC_WORK_CHUNK : gend;
            for (int t = base; t < stop; t++) f(k.own ? (int)U(k.own[t]) : t);
        }
    }
}

struct tc_density_args {
    tc_dev_const k;
    const float *hsml_in;      /* carried smoothing lengths (0 => use guess) */
    const float *guess;
    float *hsml_out, *rho_out, *vhf_out;
    double bias_const;         /* -0.0116 * pow(DESNNGB*0.01, -2.236), host libm */
    double *spill;             /* per wave: entries TC_RCAP..NGBMAX-1 of the hit list */
    int *flags;
    uint32_t *stats;           /* optional: 4 x n work counters */
    int stats_stride;
};

/* the hit list: r (f64) of every neighbour found by the last ball query */
/* Visit list entries two at a time per lane (two independent f64 chains per trip); a missing
 * second entry is replaced by `pad`.  One loop per storage segment so that every load has a single,
 * known address space (a per-entry LDS-or-global select compiles to flat loads). */
template <class F>
__device__ __forceinline__ void scan_segment(const double *p, int n, double pad, F &&f)
{
    /* `two`: the trip really has a second half-wave of entries (wave-uniform) -- the tail of a segment often has
     * not, and the second evaluation, which could only add zeros, is skipped */
    for (int base = 0; base < n; base += 128) {
        const int kk = base + lane_id(), k2 = kk + 64;
        const bool two = base + 64 < n;
        double ra = kk < n ? p[kk] : pad;
        double rb = pad;
        if (two && k2 < n) rb = p[k2];
        f(ra, rb, two);
    }
}

struct tc_rlist {
    double *lds;               /* `cap` entries */
    double *spill;             /* TC_NGBMAX entries (slot kk >= cap lives at spill[kk - cap]) */
    int cap;
    __device__ __forceinline__ void put(int kk, double v) const
    {
        if (kk < cap) lds[kk] = v;
        else spill[kk - cap] = v;
    }
    template <class F>
    __device__ __forceinline__ void scan(int cnt, double pad, F &&f) const
    {
        scan_segment(lds, cnt < cap ? cnt : cap, pad, f);
        if (cnt > cap) scan_segment(spill, cnt - cap, pad, f);
    }
};

/* two-part hit list of the fused kernel: neighbours inside the carried hsml ("inner") and the shell
 * out to 1.23 hsml ("outer"); as one list, inner entries come first */
struct tc_list2 {
    tc_rlist in, out;
    int cs;                    /* inner count */
    template <class F>
    __device__ __forceinline__ void scan(int cnt, double pad, F &&f) const
    {
        in.scan(cnt < cs ? cnt : cs, pad, f);
        if (cnt > cs) out.scan(cnt - cs, pad, f);
    }
};

/* ring of staged hit positions (and hsml_wvt for the sweep), TC_STAGE entries per wave */
struct tc_stage {
    float *x, *y, *z, *w;
};

/* src/sph.c:186-195, the rare bisection update (cold starts only); kept out of line so that the
 * f64 pow() is not speculated into every Newton iteration */
__device__ __attribute__((noinline)) double bisect_hsml(double lower, double upper)
{
    return pow(0.5 * ((lower * lower * lower) + (upper * upper * upper)), 1.0 / 3.0);
}

/* src/sph.c:80-214 on the hit list.  All lanes return identical values. */
template <class List>
__device__ __forceinline__ bool solve_hsml(const List &rl, int cnt, double mpart, double bias_const,
                                           float &hsml_io, float &rho_out, float &drho_io,
                                           uint32_t &iters, uint32_t &pairs TC_PROF_PARAM)
{
    double upper = (double)hsml_io * TC_SQRT3;
    double lower = 0;
    double hsml = (double)hsml_io;
    double rho = 0, dRhodHsml = 0;
    int it = 0;
    bool part_done = false;

    for (;;) {
        double wkNgb = 0;
        rho = dRhodHsml = 0;
        it++;
        iters++;
        pairs += cnt;

        const float hf = (float)hsml;
        /* positive operands far from the exponent limits: the unscaled divide returns the IEEE bits (tc_lean.h) */
        const double norm_h3 = tc_div_f64_lean(TC_WC6_NORM, (double)(hf * hf * hf));
        const double norm_h4 = tc_div_f64_lean(TC_WC6_NORM, (double)(hf * hf * hf * hf)) * -22.0;
        const double h3 = hsml * hsml * hsml;
        const double inv_h = tc_div_f64_lean(1.0, hsml);
        const double three_h = 3 * inv_h;
        const double nmpart = -mpart;
        const double fpt_h3 = TC_FOURPITHIRD * h3;
        /* tc_fdiv_setup (tc_math.h) with the reciprocal from the unscaled divide core: it is only used when h is
         * inside the range where that core returns the IEEE bits (exact_div covers the rest) */
        tc_fdiv fd;
        fd.b = hf;
        fd.y = tc_div_f32_lean(1.0f, hf);
        fd.exact_div = U((int)(((__builtin_bit_cast(uint32_t, hf) & 0x7fffffu) == 0x7fffffu) || !(hf > 1e-30f && hf < 1e30f)));

        /* Per-entry arithmetic (src/sph.c:133-153, :426-440), trimmed for the VALU: the f32 quotient
         * u = r/h is still correctly rounded (reciprocal + exact-residual correction); the f64
         * polynomials use t^8 = ((t^2)^2)^2 and Horner/FMA forms and r/h becomes r*(1/h), which
         * moves each f64 term by <= a few ulp (1e-16) -- the same size as the summation-order
         * difference already present -- before wk / dwk are rounded to f32 as in the reference. */
        /* Two running sums per lane instead of the reference's three (src/sph.c:149-153): S0 = sum wk and
         * S1 = sum r * dwk; then  wkNgb = 4pi/3 h^3 S0,  rho = m S0,  dRhodHsml = -m (3/h S0 + 1/h S1)  --
         * the scalar factors moved out of the neighbour sum, which moves each f64 total by <= a few ulp (the
         * same size as the summation-order difference already present) and saves 4 of 37 instructions per pair
         * and one wave reduction per solver iteration. */
        /* the inner constant of the W polynomial parked in a VGPR pair (the compiler re-materialises it with two
         * moves per entry otherwise) */
        double c25 = 25.0;
        asm volatile("; park %0" : "+v"(c25));
#ifdef TCGPU_SPH_CUBIC_SPLINE
        /* src/sph.c:140-146: the M4 kernel and its derivative as the reference writes them (tc_m4 / tc_dm4) */
        auto term = [&](auto exact, double r, double &s0, double &s1) {
            (void)exact; (void)c25; (void)norm_h3; (void)norm_h4;
            const float rf = (float)r;
            s0 += (double)tc_m4(rf, hf);
            s1 = fma(r, (double)tc_dm4(rf, hf), s1);
        };
#else
        auto term = [&](auto exact, double r, double &s0, double &s1) {
            const float rf = (float)r;
            const float u = decltype(exact)::value ? rf / fd.b : tc_fdiv_apply_fast(fd, rf);
            const double ud = (double)u;
            const double t = 1 - ud;
            const double t2 = t * t, t4 = t2 * t2, t8 = t4 * t4;
            const double poly = fma(ud, fma(ud, fma(ud, 32.0, c25), 8.0), 1.0);
            const double wk = (double)(float)(norm_h3 * t8 * poly);
            const double td = (double)(1 - u);
            const double d2 = td * td, d4 = d2 * d2, d7 = d4 * d2 * td;
            const float polyf = __builtin_fmaf(u, __builtin_fmaf(u, 16.0f, 7.0f), 1.0f);
            const double dwk = (double)(float)(norm_h4 * d7 * ud * (double)polyf);
            s0 += wk;
            s1 = fma(r, dwk, s1);
        };
#endif
        /* two independent entries per lane and trip: the f64 chains are latency-bound otherwise.
         * `r > hsml` entries are skipped: == (r2 > hsml^2) up to a zero-weight boundary;
         * a skipped entry is evaluated at r = hsml, where u = 1, t = 0 and both kernels are exactly 0.
         * The quotient's rare exact-division case (one h per wave) selects the loop, not every entry. */
        double s0 = 0, s1 = 0, s0B = 0, s1B = 0;
        TC_STAGE_SWITCH(ST_SOLVE_UNIFORM, ST_SOLVE_PAIRS);
        auto sweep_list = [&](auto exact) {
            rl.scan(cnt, hsml, [&](double ra, double rb, bool two) {
                ra = min_f64(ra, hsml);
                term(exact, ra, s0, s1);
                if (two) {
                    rb = min_f64(rb, hsml);
                    term(exact, rb, s0B, s1B);
                }
            });
        };
        if (fd.exact_div) sweep_list(std::true_type());
        else sweep_list(std::false_type());
        TC_STAGE_SWITCH(ST_SOLVE_PAIRS, ST_SOLVE_UNIFORM);
        wsum2(s0 + s0B, s1 + s1B, s0, s1);
        wkNgb = fpt_h3 * s0;
        rho = mpart * s0;
        dRhodHsml = nmpart * fma(three_h, s0, inv_h * s1);

        if (it > 128) break;

        double ngbDev = fabs(wkNgb - TC_DESNNGB);
        if (ngbDev < TC_NNGBDEV) { part_done = true; break; }

        if (fabs(upper - lower) < 1e-4) { hsml *= 1.26; break; }

        if (ngbDev < 0.5 * TC_DESNNGB) {
            /* src/sph.c:176-183; both quotients have operands in the normal range (rho, wkNgb > 0 here): the
             * unscaled divide returns the IEEE bits (tc_lean.h); single-instruction min / max as above */
            double omega = (1 + tc_div_f64_lean(dRhodHsml * hsml, 3 * rho));
            double fac = 1 - tc_div_f64_lean(wkNgb - TC_DESNNGB, 3 * wkNgb * omega);
            fac = min_f64(1.24, fac);
            fac = max_f64(1 / 1.24, fac);
            hsml *= fac;
        } else {
            if (wkNgb > TC_DESNNGB) upper = hsml;
            if (wkNgb < TC_DESNNGB) lower = hsml;
            hsml = bisect_hsml(lower, upper);
        }
    }

    hsml_io = (float)hsml;
    rho_out = (float)rho;
#ifndef TCGPU_SPH_CUBIC_SPLINE                   /* src/sph.c:201-212: neither with the cubic spline */
    if (part_done) {
        drho_io = (float)dRhodHsml;
        const float hf = (float)hsml;
        /* sph_kernel_WC6(0, hsml): u = 0, t = 1 */
        float w0 = (float)tc_div_f64_lean(TC_WC6_NORM, (double)(hf * hf * hf));   /* operands in the normal range: same bits */
        double bias_corr = bias_const * mpart * w0;
        rho_out = (float)((double)rho_out + bias_corr);
    }
#else
    (void)bias_const;
#endif
    return part_done;
}

/* per-particle state of the loop of src/sph.c:36-64 */
struct tc_dstate {
    float hsml, rho, dRhodHsml;
    uint32_t nq, nit, npair, ncand;
    bool ok;
};

/* src/sph.c:36-64 from the hsml in `d` on: ball query, raw-count guards, Find_hsml, until done */
__device__ __forceinline__ void density_loop(const tc_density_args &a, const tc_dev_const &k, int i, float xi, float yi,
                                             float zi, const tc_rlist &rl, uint32_t *idx, uint32_t idxcap,
                                             const tc_stage &st, tc_dstate &d, float rmax)
{
    const int lane = lane_id();
    float hsml = d.hsml;
    for (int guard = 0; guard < 4096; guard++) {
        /* sharded contexts: the ghost shell ends at rmax (tc_margin_radius); a wider query would miss neighbours.
         * Flag it -- the host repeats the pass on the full set -- and leave the particle alone. */
        if (hsml > rmax) {
            if (lane == 0) atomicOr(&a.flags[5], 1);
            d.ok = true;
            break;
        }
        /* ---- ball query (src/tree.c:25-111), f32 predicate, list capped at NGBMAX.
         * Hits (about a third of the candidates) are first compacted into a 128-entry LDS ring of
         * positions; the f64 pair distance is evaluated 64 hits at a time with every lane busy. */
        const float h2 = hsml * hsml;
        int cnt = 0, scnt = 0, head = 0;
        d.nq++;
        auto convert = [&](int nvalid) {
            wave_lds_fence();
            int sl = (head + lane) & (TC_STAGE - 1);
            float x = st.x[sl], y = st.y[sl], z = st.z[sl];
            if (lane < nvalid) {
                int slot = cnt + lane;
                if (slot < TC_NGBMAX) rl.put(slot, tc_pair_r(xi, yi, zi, x, y, z, k.boxhalf, k.boxsize));
            }
            cnt = U(cnt + nvalid);
            head = U((head + 64) & (TC_STAGE - 1));
            wave_lds_fence();
        };
        d.ncand += stream_candidates(k, xi, yi, zi, hsml, idx, idxcap, [&](int j, float4 p, bool act) -> bool {
            float r2 = tc_ngb_r2(xi, yi, zi, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f);
            bool hit = act && (r2 < h2);
            uint64_t m = tc_ballot(hit);
            if (TC_ABLATE(k) == 2) { cnt += __popcll(m); return false; }
            if (hit) {
                int sl = (head + scnt + mask_rank(m)) & (TC_STAGE - 1);
                st.x[sl] = p.x; st.y[sl] = p.y; st.z[sl] = p.z;
            }
            scnt = U(scnt + (int)__popcll(m));
            if (scnt >= 64) { convert(64); scnt = U(scnt - 64); }
            return cnt + scnt >= TC_NGBMAX;
        });
        if (cnt + scnt < TC_NGBMAX && scnt > 0) convert(scnt);
        else cnt += scnt;
        cnt = U(cnt);
        wave_lds_fence();
        if (TC_ABLATE(k)) {
            if (TC_ABLATE(k) != 3 || cnt >= TC_DESNNGB) { d.ok = true; break; }
            hsml = (float)((double)hsml * 1.23);
            continue;
        }
        if (cnt >= TC_NGBMAX) { hsml = (float)((double)hsml / 1.24); continue; }   /* src/sph.c:42-47 */
        if (cnt < TC_DESNNGB) { hsml = (float)((double)hsml * 1.23); continue; }   /* src/sph.c:49-54 */

        if (solve_hsml(rl, cnt, k.mpart, a.bias_const, hsml, d.rho, d.dRhodHsml, d.nit, d.npair)) { d.ok = true; break; }
        if (!isfinite(hsml)) break;
    }
    d.hsml = hsml;
}

/* src/sph.c:66-70 */
template <bool STATS = true>
__device__ __forceinline__ void density_store(const tc_density_args &a, int i, const tc_dstate &d)
{
    if (lane_id() != 0) return;
    if (!d.ok) atomicOr(&a.flags[2], 1);
    a.hsml_out[i] = d.hsml;
    a.rho_out[i] = d.rho;
    /* dRhodHsml; varHsmlFac = 1 / (1 + hsml / (3 rho) dRhodHsml) (src/sph.c:66) is formed by the kernel that scatters
     * the results to the global arrays, one lane per particle, instead of by a whole wavefront here */
    a.vhf_out[i] = d.dRhodHsml;
    if (STATS && a.stats) {
        a.stats[i] = d.nq;
        a.stats[i + (size_t)a.stats_stride] = d.nit;
        a.stats[i + 2 * (size_t)a.stats_stride] = d.npair;
        a.stats[i + 3 * (size_t)a.stats_stride] = d.ncand;
    }
}

/* src/sph.c:21-71 for particle i.  Returns false when hsml is not finite (src/sph.c:28). */
__device__ __forceinline__ bool density_init(const tc_density_args &a, int i, tc_dstate &d)
{
    d.rho = 0; d.dRhodHsml = 0; d.nq = d.nit = d.npair = d.ncand = 0; d.ok = false;
    d.hsml = a.hsml_in[i];
    if (d.hsml == 0) d.hsml = 2 * a.guess[i];                 /* src/sph.c:25-26 */
    if (!isfinite(d.hsml)) {                                  /* src/sph.c:28 */
        if (lane_id() == 0) atomicOr(&a.flags[0], 1);
        return false;
    }
    return true;
}

__device__ __forceinline__ void density_one(const tc_density_args &a, int i, const tc_rlist &rl, uint32_t *idx,
                                            const tc_stage &st)
{
    const float4 pi = a.k.pos4[i];
    tc_dstate d;
    if (!density_init(a, i, d)) return;
    const float rmax = a.k.margin_on ? tc_margin_radius(a.hsml_in[i], pi.w, a.k.boxsize, a.k.margin_widen) : HUGE_VALF;
    const tc_dev_const k = particle_view(a.k, a.hsml_in[i], pi.w);
    density_loop(a, k, i, pi.x, pi.y, pi.z, rl, idx, TC_IDXCAP, st, d, rmax);
    density_store(a, i, d);
}

#define TC_LDS_PER_WAVE_DENSITY (TC_RCAP * sizeof(double) + TC_IDXCAP * sizeof(uint32_t) + 3 * TC_STAGE * sizeof(float))
#define TC_LDS_PER_WAVE_IDX (TC_IDXCAP * sizeof(uint32_t))

__global__ __launch_bounds__(TBN) void k_density(tc_density_args a)
{
    __shared__ __align__(16) unsigned char lds_raw[WPB * TC_LDS_PER_WAVE_DENSITY];
    const int wave = threadIdx.x >> 6;
    unsigned char *mine = lds_raw + (size_t)wave * TC_LDS_PER_WAVE_DENSITY;
    const int gw = blockIdx.x * WPB + wave;
    tc_rlist rl;
    rl.lds = reinterpret_cast<double *>(mine);
    rl.spill = a.spill + (size_t)gw * (2 * TC_NGBMAX);
    rl.cap = TC_RCAP;
    uint32_t *idx = reinterpret_cast<uint32_t *>(mine + TC_RCAP * sizeof(double));
    tc_stage st;
    st.x = reinterpret_cast<float *>(idx + TC_IDXCAP);
    st.y = st.x + TC_STAGE;
    st.z = st.y + TC_STAGE;
    st.w = nullptr;
    work_queue(a.k, [&](int i) { density_one(a, i, rl, idx, st); });
}

void tc_fill_const(const tcgpu_ctx *c, tc_dev_const *k)
{
    k->boxsize = c->par.boxsize;
    k->boxhalf = 0.5 * c->par.boxsize;
    k->mpart = c->par.mpart_gas;
    k->boxinv = 1 / c->par.boxsize;
    int e2 = 0;
    k->box_mant = 2 * frexp(c->par.boxsize, &e2);       /* frexp: [0.5, 1) */
    k->box_exp = e2 - 1;
    k->boxsize_f = (float)c->par.boxsize;               /* src/tree.c:27 */
    k->boxhalf_f = (float)(c->par.boxsize * 0.5);       /* src/tree.c:28 */
    k->lmax = c->lmax;
    k->level_shift = c->level_shift;
    k->level_scale = tc_level_scale(c);
    k->cells = c->cells;
    k->lvl = c->d_lvl;
    k->orphans = c->orphans;
    k->norph = c->norph;
    k->pos4 = c->pos4;
    const bool rm = c->rows && c->mirror_valid && c->lmax_rm > 0;
    k->cum = rm ? tc_cum_base(c) : nullptr;
    k->mirror = rm ? c->mirror : nullptr;
    k->mirror_idx = rm ? c->mirror_idx : nullptr;
    k->lmax_rm = rm ? c->lmax_rm : 0;
    k->lmin_rm = rm ? c->lmin_rm : 1;
    k->mirror_pad = (uint32_t)c->mirror_alloc;
    k->pos_pad = (uint32_t)c->cap;
    k->n = (int)c->nloc;
    k->lo = 0;
    k->hi = (int)c->nown;
    k->own = c->nranks > 1 ? c->own_list : nullptr;
    k->lmin_tab = c->lmin_tab;
    k->margin_on = !c->local_full;
    k->margin_widen = c->margin_widen;
    k->work_ctr = c->work_ctr;
    k->ablate = c->ablate;
}

/* persistent grid: exactly the blocks that are co-resident (occupancy query), never more
 * blocks than work, so the static particle striding stays balanced */
template <class K>
static int grid_for(const tcgpu_ctx *c, int nloc, K kernel)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TBN, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    int need = (nloc + WPB - 1) / WPB;
    if (c->blocks_per_cu > 0 && c->blocks_per_cu < per_cu) per_cu = c->blocks_per_cu;   /* profiling: occupancy sweep */
    int cap = c->num_cu * per_cu;
    if (cap > TC_MAX_PERSISTENT_BLOCKS) cap = TC_MAX_PERSISTENT_BLOCKS;
    int g = need < cap ? need : cap;
    if (g >= 16) g &= ~7;                            /* whole round-robin turns over the 8 XCDs */
    return g;
}

int tc_launch_density(tcgpu_ctx *c)
{
    tc_density_args a;
    tc_fill_const(c, &a.k);
    a.hsml_in = c->hsml;
    a.guess = c->guess;
    a.hsml_out = c->hsml;
    a.rho_out = c->rho;
    a.vhf_out = c->vhf;
    a.bias_const = -0.0116 * pow(TC_DESNNGB * 0.01, -2.236);   /* src/sph.c:206 */
    a.spill = c->spill;
    a.flags = c->flags;
    if (c->want_stats && !c->stats) TC_HIP(c, hipMalloc(&c->stats, 4 * (size_t)c->cap * sizeof(uint32_t)));      /* on first use */
    a.stats = c->want_stats ? c->stats : nullptr;
    a.stats_stride = (int)c->cap;
    int nloc = a.k.hi - a.k.lo;
    if (nloc <= 0) return 0;
    tc_phase_begin(c, PH_DENSITY);
    TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 8 * 16 * sizeof(int), c->stream));
    k_density<<<grid_for(c, nloc, k_density), TBN, 0, c->stream>>>(a);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K9 WVT sweep */

struct tc_wvt_args {
    tc_dev_const k;
    double step;
    float *delta;       /* 3n xyz interleaved, G order */
    const uint32_t *lg; /* local slot -> G index */
    const float *hsml0; /* the carried smoothing lengths the pass started from (level ranges of a sharded pass) */
    int *flags;
};

/* accumulate one neighbour's contribution, src/wvt_relax.c:144-169 */
__device__ __forceinline__ void wvt_pair(const float4 pi, const float4 pj, double boxinv, double step_hi,
                                         double &d0, double &d1, double &d2, bool wrap = true)
{
    float dx = (float)((double)(pi.x - pj.x) * boxinv);
    float dy = (float)((double)(pi.y - pj.y) * boxinv);
    float dz = (float)((double)(pi.z - pj.z) * boxinv);
    if (wrap) {          /* wave-uniform: false when the particle's whole candidate region is inside the box */
        dx = (double)dx > 0.5 ? (float)((double)dx - 1) : dx;
        dy = (double)dy > 0.5 ? (float)((double)dy - 1) : dy;
        dz = (double)dz > 0.5 ? (float)((double)dz - 1) : dz;
        dx = (double)dx < -0.5 ? (float)((double)dx + 1) : dx;
        dy = (double)dy < -0.5 ? (float)((double)dy + 1) : dy;
        dz = (double)dz < -0.5 ? (float)((double)dz + 1) : dz;
    }
    float r2 = (dx * dx + dy * dy + dz * dz);
    float h = (float)(0.5 * (double)(pi.w + pj.w));
    if (r2 > h * h) return;
    /* (float)sqrt((double)r2) == sqrtf(r2): both are the correctly rounded f32 root.  Root, quotient and
     * reciprocal without the IEEE range handling where no operand needs it (tc_lean.h: same bits); the
     * branch is uniform over the lanes that got here */
    float r, q;
    double rinv;
    if (tc_ballot(r2 < 1e-24f)) {            /* tiny or zero (coincident particles: the reference divides by zero) */
        r = sqrtf(r2);
        q = r / h;
        rinv = 1.0 / (double)r;
    } else {
        r = tc_sqrt_f32_lean_pos(r2);
        q = tc_div_f32_lean(r, h);
        rinv = tc_rcp_f64_lean_nz((double)r);
    }
    /* src/wvt_relax.c:275-281 with t^8 by squaring and a Horner/FMA polynomial (terms move by a few
     * ulp of f64 before wk is rounded to f32, as in solve_hsml) */
    const double u = (double)q;
    const double t = 1 - u;
    const double t2 = t * t, t4 = t2 * t2, t8 = t4 * t4;
    float wk = (float)(TC_WC6_NORM * t8 * fma(u, fma(u, fma(u, 32.0, 25.0), 8.0), 1.0));
    /* step*hsml_i*wk*d/r for the three components with one reciprocal of r */
    double base = step_hi * (double)wk * rinv;
    d0 = fma(base, (double)dx, d0);
    d1 = fma(base, (double)dy, d1);
    d2 = fma(base, (double)dz, d2);
}

/* src/wvt_relax.c:128-171 for particle i: sum of step_hi * W6 * unit(r_ij) over the ball of radius hq */
__device__ __forceinline__ void wvt_sum(const tc_dev_const &k, int i, const float4 pi, double step_hi, int *flags,
                                        uint32_t *idx, uint32_t idxcap, const tc_stage &st, double &d0, double &d1,
                                        double &d2)
{
    const int lane = lane_id();
    const double boxinv = k.boxinv;
    const float hq = (float)((double)pi.w * k.boxsize);      /* src/wvt_relax.c:135 */
    const float hq2 = hq * hq;

    /* The reference truncates a list that overflows to its first NGBMAX hits in ascending index
     * (src/tree.c:91-92) without noticing.  Reproduced by re-gathering: pass 0 sums over all hits and counts
     * them; if there are NGBMAX or more, counting passes bisect the index threshold T with exactly NGBMAX hits
     * below it and a last pass sums the hits with j < T.  One stream call site for all passes (code size). */
    int limit = k.n, tlo = 0, thi = k.n, mode = 0;          /* 0: sum everything, 1: count j < limit, 2: sum j < limit */
    for (;;) {
        int cnt = 0, scnt = 0, head = 0;
        if (mode != 1) d0 = d1 = d2 = 0;
        /* hits are compacted into the LDS ring first so that the pair arithmetic runs on full waves */
        auto convert = [&](int nvalid) {
            wave_lds_fence();
            int sl = (head + lane) & (TC_STAGE - 1);
            float4 p = make_float4(st.x[sl], st.y[sl], st.z[sl], st.w[sl]);
            if (lane < nvalid) wvt_pair(pi, p, boxinv, step_hi, d0, d1, d2);
            head = U((head + 64) & (TC_STAGE - 1));
            wave_lds_fence();
        };
        stream_candidates(k, pi.x, pi.y, pi.z, hq, idx, idxcap, [&](int j, float4 p, bool act) -> bool {
            float r2 = tc_ngb_r2(pi.x, pi.y, pi.z, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f);
            bool hit = act && (r2 < hq2) && j < limit;
            cnt += __popcll(tc_ballot(hit));
            if (mode == 1) return false;
            bool use = hit && j != i;
            uint64_t m = tc_ballot(use);
            if (use) {
                int sl = (head + scnt + mask_rank(m)) & (TC_STAGE - 1);
                st.x[sl] = p.x; st.y[sl] = p.y; st.z[sl] = p.z; st.w[sl] = p.w;
            }
            scnt = U(scnt + (int)__popcll(m));
            if (scnt >= 64) { convert(64); scnt = U(scnt - 64); }
            return false;
        });
        if (mode != 1 && scnt > 0) convert(scnt);
        cnt = U(cnt);
        if (mode == 0) {
            if (cnt < TC_NGBMAX) break;
            if (lane == 0) atomicAdd(&flags[4], 1);
            mode = 1;                                        /* count(j < tlo) < NGBMAX <= count(j < thi) */
        } else if (mode == 1) {
            if (cnt >= TC_NGBMAX) thi = limit; else tlo = limit;
        } else {
            break;
        }
        if (thi - tlo > 1) limit = tlo + ((thi - tlo) >> 1);
        else { mode = 2; limit = thi; }
    }
    d0 = wsum(d0); d1 = wsum(d1); d2 = wsum(d2);
}

__device__ __forceinline__ void wvt_one(const tc_wvt_args &a, int i, uint32_t *idx, const tc_stage &st)
{
    const float4 pi = a.k.pos4[i];
    double d0, d1, d2;
    const tc_dev_const k = particle_view(a.k, a.hsml0[i], pi.w);
    wvt_sum(k, i, pi, a.step * (double)pi.w, a.flags, idx, TC_IDXCAP, st, d0, d1, d2);
    if (lane_id() == 0) {
        const size_t g = a.lg[i];
        a.delta[3 * g] = (float)d0;
        a.delta[3 * g + 1] = (float)d1;
        a.delta[3 * g + 2] = (float)d2;
    }
}

__global__ __launch_bounds__(TBN) void k_wvt(tc_wvt_args a)
{
    __shared__ __align__(16) uint32_t lds_idx[WPB * TC_IDXCAP];
    __shared__ __align__(16) float lds_stage[WPB * 4 * TC_STAGE];
    const int wave = threadIdx.x >> 6;
    uint32_t *idx = lds_idx + (size_t)wave * TC_IDXCAP;
    tc_stage st;
    st.x = lds_stage + (size_t)wave * 4 * TC_STAGE;
    st.y = st.x + TC_STAGE;
    st.z = st.y + TC_STAGE;
    st.w = st.z + TC_STAGE;
    work_queue(a.k, [&](int i) { wvt_one(a, i, idx, st); });
}

int tc_launch_wvt(tcgpu_ctx *c, double step)
{
    tc_wvt_args a;
    tc_fill_const(c, &a.k);
    a.step = step;
    a.delta = c->delta;
    a.lg = c->lg;
    a.hsml0 = c->hsml0;
    a.flags = c->flags;
    int nloc = a.k.hi - a.k.lo;
    if (nloc <= 0) return 0;
    tc_phase_begin(c, PH_WVT);
    TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 8 * 16 * sizeof(int), c->stream));
    k_wvt<<<grid_for(c, nloc, k_wvt), TBN, 0, c->stream>>>(a);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K9 exact: the sweep in the reference's order
 *
 * src/wvt_relax.c:167-169 adds every neighbour's term to a FLOAT accumulator (`delta[k][ipart] += ...`), in the order
 * Find_ngb_tree emits the list: ascending particle index (src/tree.c:25-111 walks the depth-first node array, whose
 * leaves tile [0, N) in ascending order).  A sum rounded after every addend depends on that order, so the displacement
 * can only be reproduced bit for bit by visiting the neighbours in ascending index and rounding where the reference
 * rounds.  round 2's sweep (k_wvt / the fused kernel: f64 partial sums over 64 lanes, one rounding) differs from it by
 * ~1e-6 |delta| per iteration -- the one systematic input difference between this library and the reference, and the
 * owner of the parity tail (tools/attribute_tail.py).
 *
 * One LANE per particle here (a sequentially rounded sum is a serial chain: a wavefront per particle would keep
 * three lanes busy), 64 Peano-consecutive particles per wavefront, three phases:
 *   A1  every lane walks the implicit octree of the cell table depth-first with the children of a node taken in
 *       CURVE order (TC_HILBERT_INV: the 24 orientations of the reference's transform), descending only into cells
 *       the ball overlaps, and notes the index runs of the overlapped cells of its query level -- they come out in
 *       ascending index, adjacent ones merged (per-wave scratch, [slot][lane]);
 *   A2  the lane walks its runs, tests every particle with the reference's all-f32 predicate (src/tree.c:67-89) and
 *       appends the hits to a per-lane LDS buffer, stopping like the reference at the NGBMAX-th hit (src/tree.c:91-92);
 *   B   whenever a buffer is full the lanes evaluate their buffered hits: src/wvt_relax.c:141-169 statement for
 *       statement -- f32 differences scaled in f64, f32 folding, f32 r2, (float)sqrt, f32 quotient, the f64 kernel
 *       polynomial multiplied out left to right, ((step * h_i) * wk) * d / r in f64, and the f32 accumulator rounded
 *       after each neighbour.  Divisions and roots use the unscaled cores of tc_lean.h where the operands are in the
 *       normal range (the IEEE bits, verified operand by operand) and the IEEE sequences otherwise.
 * No f64 value here depends on summation order, so the result does not depend on wave scheduling, grid size or on
 * how particles are split over ranks.
 */
#define TC_XHITS 48            /* buffered hits per lane (LDS: 4 B x 64 lanes x TC_XHITS per wave) */
#define TC_XRUNCAP 384         /* index runs per lane (after merging; <= 8^3 cells of the query level, about half inside the ball) */

struct tc_xwvt_args {
    tc_dev_const k;
    double step;
    float *delta;       /* 3n xyz interleaved, G order */
    const uint32_t *lg; /* local slot -> G index */
    const float *hsml0; /* the carried smoothing lengths the pass started from (level ranges of a sharded pass) */
    int *flags;
    uint2 *runs;        /* per wave: TC_XRUNCAP x 64 */
    int xshift;         /* added to the query level (tuning: coarser leaves = fewer cells to walk, more candidates to test) */
    int orphans_only;   /* k_wvt_exact: run only when the local set has orphans (k_wvt_exact4 did the launch otherwise) */
    int dbg;            /* profiling only (results invalid): 1 = stop after A1, 2 = no B */
    const uint32_t *xlist, *xlcnt;   /* k_iter's neighbour lists (WVT == 2): k_wvt_chain4 evaluates them, k_wvt_exact4 then only
                                      * takes the particles without a list ... */
    const uint32_t *wl;              /* ... which k_iter wrote down (local slots), *wl_cnt of them: the work items, 16 per wave so
                                      * that a handful of particles does not cost the latency of a 64-particle group */
    const int *wl_cnt;
    tc_pf pf;           /* cell starts in curve order (tc_launch_pfirst, tc_pf_first) */
};

/* periodic distance (one dimension) from x to the cell [c s, (c + 1) s) of a ring of circumference box; 0 inside */
__device__ __forceinline__ float cell_gap(float x, int c, float s, float box)
{
    const float clo = (float)c * s, chi = clo + s;
    float g = fmaxf(fmaxf(clo - x, x - chi), 0.0f);
    const float alt = fmaxf(box - s - g, 0.0f);            /* the other way round the ring */
    return fminf(g, alt);
}

/* groups of `chunk` (a power of two <= 64) consecutive work items, XCD-aware like work_queue(): f(first, stop) */
template <class F>
__device__ __forceinline__ void work_queue64(const tc_dev_const &k, F &&f, int chunk = 64)
{
    const int lo = k.lo, hi = k.hi;
    const bool grouped = gridDim.x >= 16 && (gridDim.x & 7) == 0;
    const int ngroups = grouped ? 8 : 1;
    int glen = grouped ? ((hi - lo + 7) >> 3) : (hi - lo);
    glen = (glen + chunk - 1) & ~(chunk - 1);
    const int g0 = grouped ? (int)(blockIdx.x & 7) : 0;
    for (int gg = 0; gg < ngroups; gg++) {
        const int grp = (g0 + gg) & (ngroups - 1);
        const int gstart = lo + grp * glen;
        int gend = gstart + glen;
        if (gend > hi) gend = hi;
        if (gend <= gstart) continue;
        for (;;) {
            uint32_t got = 0;
            if ((threadIdx.x & 63) == 0)
                got = atomicAdd(reinterpret_cast<unsigned int *>(&k.work_ctr[16 * grp]), (unsigned int)chunk);
            got = U(got);
            if (got >= (uint32_t)(gend - gstart)) break;
            const int base = gstart + (int)got;
            f(base, base + chunk < gend ? base + chunk : gend);
        }
    }
}

/* A1 of the exact sweep: the index runs of the level-Lq cells the ball (padded radius^2 hp2) overlaps, in ascending
 * index, adjacent ones merged; one lane per particle; entry s of this lane's list at runs[64 s].  Returns the count. */
__device__ __forceinline__ int ordered_runs(const tc_dev_const &k, bool valid, float xi, float yi, float zi, float hp2, int Lq,
                                            const tc_level_desc &D, const unsigned char *lds_inv, uint2 *runs, int *flags)
{
    const float boxf = k.boxsize_f;
    /* ---- A1: overlapped cells of level Lq in curve order -> index runs */
    int l = valid ? 1 : 0;
    int cx = 0, cy = 0, cz = 0;          /* the node (level l - 1) whose children are being enumerated */
    uint32_t dig = 0;                    /* 3 bits per level: next child (curve order) */
    uint64_t sts = 0;                    /* 5 bits per level: orientation of the node at that level (root: 0) */
    uint32_t cur_f = 0, cur_e = 0;
    int nruns = 0;
    while (tc_ballot(l > 0)) {
        if (l > 0) {
            const int s3 = 3 * (l - 1);
            const uint32_t d = (dig >> s3) & 7u;
            const uint32_t st = (uint32_t)(sts >> (5 * (l - 1))) & 31u;
            const uint32_t e = lds_inv[st * 8 + d];
            const int ccx = 2 * cx + (int)(e & 1u), ccy = 2 * cy + (int)((e >> 2) & 1u), ccz = 2 * cz + (int)((e >> 1) & 1u);
            const float s = __builtin_ldexpf(boxf, -l);
            const float gx = cell_gap(xi, ccx, s, boxf), gy = cell_gap(yi, ccy, s, boxf), gz = cell_gap(zi, ccz, s, boxf);
            const bool ov = gx * gx + gy * gy + gz * gz <= hp2;
            bool advance = true;
            if (ov) {
                if (l == Lq) {
                    const int ix = ccx - D.ox, iy = ccy - D.oy, iz = ccz - D.oz;
                    if (ix >= 0 && ix < D.nx && iy >= 0 && iy < D.ny && iz >= 0 && iz < D.nz) {
                        const uint2 ce = k.cells[(size_t)D.off + ((size_t)ix * D.ny + iy) * D.nz + iz];
                        const uint32_t f0 = ~ce.x, e0 = ce.y;
                        if (e0 > f0) {
                            if (cur_e > cur_f && f0 == cur_e) cur_e = e0;          /* adjacent in index: one run */
                            else {
                                if (cur_e > cur_f) {
                                    if (nruns < TC_XRUNCAP) runs[(size_t)nruns * 64] = make_uint2(cur_f, cur_e);
                                    nruns++;
                                }
                                cur_f = f0; cur_e = e0;
                            }
                        }
                    }
                } else {
                    cx = ccx; cy = ccy; cz = ccz;
                    sts = (sts & ~(31ull << (5 * l))) | ((uint64_t)(e >> 3) << (5 * l));
                    l++;
                    dig &= ~(7u << (3 * (l - 1)));
                    advance = false;
                }
            }
            if (advance) {
                for (;;) {
                    const int q3 = 3 * (l - 1);
                    if (((dig >> q3) & 7u) < 7u) { dig += 1u << q3; break; }
                    dig &= ~(7u << q3);
                    l--;
                    if (l == 0) break;
                    cx >>= 1; cy >>= 1; cz >>= 1;
                }
            }
        }
    }
    if (cur_e > cur_f) {
        if (nruns < TC_XRUNCAP) runs[(size_t)nruns * 64] = make_uint2(cur_f, cur_e);
        nruns++;
    }
    if (nruns > TC_XRUNCAP) { atomicOr(&flags[3], 1); nruns = TC_XRUNCAP; }
    return nruns;
}

__global__ __launch_bounds__(TBN) void k_wvt_exact(tc_xwvt_args a)
{
    __shared__ uint32_t lds_hits[WPB * TC_XHITS * 64];
    __shared__ unsigned char lds_inv[TC_HILBERT_NSTATES * 8];
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES * 8; t += TBN) lds_inv[t] = TC_HILBERT_INV[t];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t *hits = lds_hits + (size_t)wave * (TC_XHITS * 64) + lane;            /* entry s at hits[64 s] */
    uint2 *runs = a.runs + (size_t)(blockIdx.x * WPB + wave) * ((size_t)TC_XRUNCAP * 64) + lane;
    const tc_dev_const &k = a.k;
    const double boxinv = k.boxinv;
    int norph = *k.norph;
    if (norph > TC_MAX_ORPHANS) norph = TC_MAX_ORPHANS;
    if (a.orphans_only && norph == 0) return;

    work_queue64(k, [&](int base, int stop) {
        const int t = base + lane;
        const bool valid = t < stop;
        const int tt = valid ? t : base;
        const int i = k.own ? (int)k.own[tt] : tt;
        const float4 pi = k.pos4[i];
        const float xi = pi.x, yi = pi.y, zi = pi.z;
        const float hq = (float)((double)pi.w * k.boxsize);             /* src/wvt_relax.c:135 */
        const float hq2 = hq * hq;
        /* query level: the table levels this particle's queries may use (sharded passes: tc_particle_levels) */
        int lmin = k.lmin_tab, lmaxp = k.lmax;
        if (k.margin_on) {
            const float h0 = a.hsml0[i];
            const float rg = tc_margin_radius(h0, pi.w, k.boxsize, k.margin_widen);
            tc_particle_levels(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift, k.lmax, h0, rg, &lmin, &lmaxp);
        }
        const int Lq = tc_query_level(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift + a.xshift, lmin, lmaxp, hq);
        const tc_level_desc D = k.lvl[Lq];
        /* padded radius: the f32 predicate can accept pairs a few ulp beyond hq, and the cell bounds below are f32
         * products of magnitude boxsize */
        const float hp = (float)((double)hq * (1.0 + 1e-5) + k.boxsize * 4e-6);
        const float hp2 = hp * hp;

        const int nruns = ordered_runs(k, valid, xi, yi, zi, hp2, Lq, D, lds_inv, runs, a.flags);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        /* Orphans (a coordinate == boxsize: not in the cell table, k_cells) are tested one by one and merged into the
         * ascending order by index; `opend` = the smallest orphan index not yet considered.  None exist in practice. */
        auto next_orphan = [&](int after) -> uint32_t {
            uint32_t best = 0xffffffffu;
            for (int o = 0; o < norph; o++) {
                const uint32_t jo = k.orphans[o];
                if ((int)jo > after && jo < best) best = jo;
            }
            return best;
        };
        uint32_t opend = norph ? next_orphan(-1) : 0xffffffffu;

        /* ---- A2 / B */
        int ri = 0;
        uint32_t j = 0, jend = 0;
        int nlist = 0;                       /* length of the reference's list so far (self included) */
        bool cdone = !valid;
        float d0 = 0, d1 = 0, d2 = 0;
        const double step_hi = a.step * (double)pi.w;                   /* step * hsml[ipart], src/wvt_relax.c:167 */
        for (;;) {
            int cnt = 0;
            while (!tc_ballot(cnt >= TC_XHITS) && tc_ballot(!cdone)) {
                if (!cdone) {
                    if (j == jend) {
                        if (ri < nruns) {
                            const uint2 r = runs[(size_t)ri * 64];
                            ri++;
                            j = r.x; jend = r.y;
                        } else j = jend = 0xffffffffu;                  /* only orphans left, if any */
                    }
                    /* next item in ascending index: the run's next particle, or an orphan due before it */
                    uint32_t jc = 0;
                    bool take = false, from_run = false;
                    if (norph && opend < j) {
                        jc = opend;
                        opend = next_orphan((int)opend);
                        take = true;
                    } else if (j != 0xffffffffu) {
                        jc = j++;
                        take = from_run = true;
                    }
                    if (take) {
                        const float4 p = k.pos4[jc];
                        /* an orphan met inside a run's index range waits for its turn as an orphan */
                        const bool mine = !(norph && from_run && is_orphan(k, p));
                        const float r2 = tc_ngb_r2(xi, yi, zi, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f);
                        if (mine && r2 < hq2) {
                            hits[64 * cnt] = jc;
                            cnt++;
                            if (++nlist == TC_NGBMAX) { cdone = true; atomicAdd(&a.flags[4], 1); }   /* src/tree.c:91-92 */
                        }
                    } else cdone = true;
                }
            }
            for (int s = 0; tc_ballot(s < cnt); s++) {
                if (s < cnt) {
                    const uint32_t jj = hits[64 * s];
                    if ((int)jj != i) {                                 /* src/wvt_relax.c:141-142 */
                        const float4 pj = k.pos4[jj];
                        float dx = (float)((double)(xi - pj.x) * boxinv);
                        float dy = (float)((double)(yi - pj.y) * boxinv);
                        float dz = (float)((double)(zi - pj.z) * boxinv);
                        dx = (double)dx > 0.5 ? dx - 1.0f : dx;         /* src/wvt_relax.c:148-154 */
                        dy = (double)dy > 0.5 ? dy - 1.0f : dy;
                        dz = (double)dz > 0.5 ? dz - 1.0f : dz;
                        dx = (double)dx < -0.5 ? dx + 1.0f : dx;
                        dy = (double)dy < -0.5 ? dy + 1.0f : dy;
                        dz = (double)dz < -0.5 ? dz + 1.0f : dz;
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        const float h = (float)(0.5 * (double)(pi.w + pj.w));
                        if (!(r2 > h * h)) {
                            double e0, e1, e2;
                            if (r2 < 1e-24f) {                          /* coincident particles: the IEEE sequences (0 / 0 and all) */
                                const float r = sqrtf(r2);
                                const float wk = (float)tc_wvt_wc6(r, h);
                                e0 = step_hi * (double)wk * (double)dx / (double)r;
                                e1 = step_hi * (double)wk * (double)dy / (double)r;
                                e2 = step_hi * (double)wk * (double)dz / (double)r;
                            } else {
                                const float r = tc_sqrt_f32_lean_pos(r2);            /* == (float)sqrt((double)r2) */
                                const double u = (double)tc_div_f32_lean(r, h);      /* src/wvt_relax.c:277: f32 quotient */
                                const double tq = 1 - u;
                                const float wk = (float)(TC_WC6_NORM * tq * tq * tq * tq * tq * tq * tq * tq
                                                         * (1 + 8 * u + 25 * u * u + 32 * u * u * u));
                                const double b = step_hi * (double)wk;
                                const double rd = (double)r;
                                /* three quotients by the same r: one reciprocal refinement, then the residual
                                 * correction of tc_div_f64_lean per numerator */
                                double y = __builtin_amdgcn_rcp(rd);
                                double ee = __builtin_fma(-rd, y, 1.0);
                                y = __builtin_fma(y, ee, y);
                                ee = __builtin_fma(-rd, y, 1.0);
                                y = __builtin_fma(y, ee, y);
                                const double n0 = b * (double)dx, n1 = b * (double)dy, n2 = b * (double)dz;
                                const double q0 = n0 * y, q1 = n1 * y, q2 = n2 * y;
                                e0 = __builtin_fma(__builtin_fma(-rd, q0, n0), y, q0);
                                e1 = __builtin_fma(__builtin_fma(-rd, q1, n1), y, q1);
                                e2 = __builtin_fma(__builtin_fma(-rd, q2, n2), y, q2);
                            }
                            d0 = (float)((double)d0 + e0);              /* src/wvt_relax.c:167-169: rounded per neighbour */
                            d1 = (float)((double)d1 + e1);
                            d2 = (float)((double)d2 + e2);
                        }
                    }
                }
            }
            if (!tc_ballot(!cdone)) break;
        }
        if (valid) {
            const size_t g = a.lg[i];
            a.delta[3 * g] = d0;
            a.delta[3 * g + 1] = d1;
            a.delta[3 * g + 2] = d2;
        }
    });
}

/* A1 on the curve-ordered cell starts (pf, tc_launch_pfirst): no memory is touched while walking.  A node's eight
 * children are classified at once -- per dimension the periodic distance from the particle to the lower and to the upper
 * half, nearest and farthest point -- into "overlaps the ball" and "inside the ball" masks, brought into curve order by
 * a table look-up (perm[orientation][mask]).  Children that are inside the ball, or overlap it at the query level, are
 * final: their key ranges (level-Lq keys) are noted in curve order, adjacent ones merged; the others are descended into,
 * one at a time (the node's state waits in a per-lane LDS stack).  Every iteration of the wave's loop is one visit of
 * one node per lane.  At the end the key ranges become index runs: [pf[a], pf[b + 1]).  Returns the number of runs. */
/* `runs`: entry s of this lane's list at runs[s * stride], at most `cap` entries (a longer list is cut and its true length
 * returned; with `flags` the overflow is also reported as an error). */
__device__ __forceinline__ int ordered_runs_pf(const tc_dev_const &k, const tc_pf &pf, bool valid, float xi,
                                               float yi, float zi, float hq, int Lq, const uint64_t *inv64,
                                               const unsigned char *perm, uint32_t *stk, uint2 *runs, int *flags,
                                               const int stride = 64, const int cap = TC_XRUNCAP)
{
    const float boxf = k.boxsize_f, boxh = k.boxhalf_f;
    const float hp = (float)((double)hq * (1.0 + 1e-5) + k.boxsize * 4e-6);
    const float hp2 = hp * hp;
    const float hin = fmaxf((float)((double)hq * (1.0 - 1e-5) - k.boxsize * 4e-6), 0.0f);
    const float hin2 = hin * hin;
    int l = 0;                               /* level of the current node */
    int cx = 0, cy = 0, cz = 0;              /* its cell coordinates */
    uint32_t key = 0, st = 0;                /* its key prefix and orientation */
    uint32_t cov = 0, cfin = 0;              /* children still to do (curve order): overlapping / final among them */
    bool enter = true, active = valid;
    uint32_t ra = 0, rb = 0;
    bool have = false;
    int nout = 0;
    auto emit = [&](uint32_t A, uint32_t B) {
        if (have && A == rb + 1u) rb = B;
        else {
            if (have) { if (nout < cap) runs[(size_t)nout * stride] = make_uint2(ra, rb); nout++; }
            ra = A; rb = B; have = true;
        }
    };
    /* Every turn of the wave's loop EVALUATES one node per lane (the expensive, uniform part); what follows -- noting the
     * final children, opening the next partial one, or climbing back until there is one -- is cheap and of variable
     * length.  (First version: a turn per visit, entered or resumed: twice the turns, and lanes out of phase paid the
     * evaluation in every one of them.) */
    (void)enter;
    while (tc_ballot(active)) {
        if (active) {
            {
                const float s = __builtin_ldexpf(boxf, -(l + 1));      /* edge of the children */
                /* periodic distance per dimension from the particle to the midpoint of the lower / upper child */
                const float hs = 0.5f * s;
                float g[6], f[6];                                    /* nearest / farthest distance, [2 d + half] */
                /* midpoint of the node = (2 c + 1) s, from the integer coordinates: no drift over pushes and pops */
                const float xs[3] = {xi, yi, zi}, ms[3] = {(float)(2 * cx + 1) * s, (float)(2 * cy + 1) * s, (float)(2 * cz + 1) * s};
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const float u = xs[d] - ms[d];
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        float dd = fabsf(hf ? u - hs : u + hs);
                        dd = dd > boxh ? boxf - dd : dd;
                        g[2 * d + hf] = fmaxf(dd - hs, 0.0f);
                        f[2 * d + hf] = dd + hs;
                    }
                }
#pragma unroll
                for (int q = 0; q < 6; q++) { g[q] *= g[q]; f[q] *= f[q]; }
                uint32_t ov = 0, in = 0;                             /* octant o: bit 2 = y half, bit 1 = z half, bit 0 = x half */
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const float g2 = g[o & 1] + g[2 + ((o >> 2) & 1)] + g[4 + ((o >> 1) & 1)];
                    const float f2 = f[o & 1] + f[2 + ((o >> 2) & 1)] + f[4 + ((o >> 1) & 1)];
                    ov |= (g2 <= hp2 ? 1u : 0u) << o;
                    in |= (f2 <= hin2 ? 1u : 0u) << o;
                }
                cov = perm[st * 256 + ov];
                cfin = (l + 1 == Lq) ? cov : (uint32_t)perm[st * 256 + (in & ov)];
            }
            for (;;) {
                /* final children ahead of the first one that has to be opened: note their key ranges */
                const uint32_t part = cov & ~cfin;
                const uint32_t upto = part ? (1u << __builtin_ctz(part)) - 1u : 0xffu;
                uint32_t fn = cov & cfin & upto;
                cov &= ~fn;
                const int sh = 3 * (Lq - (l + 1));
                while (fn) {
                    const int a0 = __builtin_ctz(fn);
                    const int len = __builtin_ctz(~(fn >> a0));
                    fn &= ~(((1u << len) - 1u) << a0);
                    const uint32_t A = ((key << 3) + (uint32_t)a0) << sh;
                    const uint32_t B = ((((key << 3) + (uint32_t)(a0 + len - 1)) + 1u) << sh) - 1u;
                    emit(A, B);
                }
                if (part) {                                          /* open the next child: evaluated in the next turn */
                    const int kk = __builtin_ctz(part);
                    cov &= ~(1u << kk);
                    stk[64 * l] = cov | (cfin << 8) | (st << 16);
                    const uint32_t e = (uint32_t)(inv64[st] >> (8 * kk)) & 0xffu;
                    cx = 2 * cx + (int)(e & 1u);
                    cy = 2 * cy + (int)((e >> 2) & 1u);
                    cz = 2 * cz + (int)((e >> 1) & 1u);
                    key = (key << 3) + (uint32_t)kk;
                    st = e >> 3;
                    l++;
                    break;
                }
                if (l == 0) { active = false; break; }
                l--;                                                 /* node done: back to its parent, and look on there */
                const uint32_t w = stk[64 * l];
                key >>= 3;
                cx >>= 1; cy >>= 1; cz >>= 1;
                cov = w & 0xffu; cfin = (w >> 8) & 0xffu; st = w >> 16;
            }
        }
    }
    if (have) { if (nout < cap) runs[(size_t)nout * stride] = make_uint2(ra, rb); nout++; }
    const int ntrue = nout;
    if (nout > cap) { if (flags) atomicOr(&flags[3], 1); nout = cap; }
    /* key ranges -> index runs */
    const uint32_t offL = tc_pf_offset(pf.lmin, Lq), offC = tc_pf_offset(pf.lmin, pf.lc);
    int m = 0;
    for (int t = 0; t < nout; t++) {
        const uint2 kr = runs[(size_t)t * stride];
        const uint32_t f0 = tc_pf_first(pf, Lq, offL, offC, kr.x, (uint32_t)k.n);
        const uint32_t e0 = tc_pf_first(pf, Lq, offL, offC, kr.y + 1u, (uint32_t)k.n);
        if (e0 > f0) { runs[(size_t)m * stride] = make_uint2(f0, e0); m++; }
    }
    return ntrue > cap ? ntrue : m;
}

/* The same sweep with FOUR lanes per particle for the two phases that touch memory (the launch's normal kernel).
 * A1 runs one lane per particle on the curve-ordered cell starts (ordered_runs_pf); then, 16 particles at a
 * time, a quad walks its particle's runs four candidates per step -- one 64-byte read instead of four scattered ones --
 * keeps the hits' positions (in ascending index: rank inside the step from the quad's lane mask) in LDS, and evaluates
 * buffered hits four at a time; the f32 accumulator then takes the four terms in order, passed through the quad with
 * DPP (the order and every rounding of src/wvt_relax.c:167-169). */
#define TC_X4CAP 16            /* buffered hit positions per particle (12 with 6 blocks per CU instead of 5: measured the same, 8.3 ms) */

__device__ __forceinline__ uint32_t quad_bits(uint64_t ballot, int lane)
{
    return (uint32_t)(ballot >> (lane & 60)) & 15u;
}

__device__ __forceinline__ float quad_bcast(float v, int kk)
{
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (kk) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, false); break;
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xf, 0xf, false); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0xaa, 0xf, 0xf, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0xff, 0xf, 0xf, false); break;
    }
    return __builtin_bit_cast(float, r);
}

__global__ __launch_bounds__(TBN) void k_wvt_exact4(tc_xwvt_args a)
{
    __shared__ __align__(16) float4 lds_hits4[WPB * 16 * TC_X4CAP];       /* A2 / B: hit positions; A1: the per-lane node stack */
    __shared__ int lds_nruns[WPB * 64];
    __shared__ double lds_terms[WPB * 64 * 3];                             /* B: the terms of a step, [quad][hit][component] */
    __shared__ uint64_t lds_inv64[TC_HILBERT_NSTATES];                     /* the 8 children of an orientation, one per byte */
    __shared__ unsigned char lds_perm[TC_HILBERT_NSTATES * 256];           /* octant mask -> curve-order mask */
    static_assert(16 * TC_X4CAP * sizeof(float4) >= (TC_MAX_LEVEL + 1) * 64 * sizeof(uint32_t), "stack fits the hit buffer");
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES; t += TBN) {
        uint64_t r = 0;
        for (int kk = 0; kk < 8; kk++) r |= (uint64_t)TC_HILBERT_INV[t * 8 + kk] << (8 * kk);
        lds_inv64[t] = r;
    }
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES * 256; t += TBN) {
        const int stt = t >> 8, msk = t & 255;
        uint32_t cm = 0;
        for (int kk = 0; kk < 8; kk++) cm |= ((msk >> (TC_HILBERT_INV[stt * 8 + kk] & 7)) & 1u) << kk;
        lds_perm[t] = (unsigned char)cm;
    }
    __syncthreads();
    tc_dev_const kw = a.k;
    const int group = a.wl ? 16 : 64;                          /* work items per wave and turn */
    if (a.wl) { kw.lo = 0; kw.hi = *a.wl_cnt; kw.own = a.wl; }
    const tc_dev_const &k = kw;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t *stk = reinterpret_cast<uint32_t *>(lds_hits4 + (size_t)wave * 16 * TC_X4CAP);
    const int q = lane >> 2, lq = lane & 3;
    const uint32_t below = (1u << lq) - 1u;
    float4 *buf = lds_hits4 + ((size_t)wave * 16 + q) * TC_X4CAP;
    double *tq = lds_terms + ((size_t)wave * 16 + q) * 12;          /* this quad's 4 x 3 terms */
    const int lc = lq < 3 ? lq : 0;                                  /* the component this lane accumulates (lane 3: a copy of x) */
    int *nr = lds_nruns + wave * 64;
    uint2 *wruns = a.runs + (size_t)(blockIdx.x * WPB + wave) * ((size_t)TC_XRUNCAP * 64);
    const double boxinv = k.boxinv;

    work_queue64(k, [&](int base, int stop) {
        /* ---- A1, one lane per particle */
        {
            const int t = base + lane;
            const int tt = t < stop ? t : base;
            const int i = k.own ? (int)k.own[tt] : tt;
            const bool valid = t < stop;
            const float4 pi = k.pos4[i];
            const float hq = (float)((double)pi.w * k.boxsize);
            int lmin = k.lmin_tab, lmaxp = k.lmax;
            if (k.margin_on) {
                const float h0 = a.hsml0[i];
                const float rg = tc_margin_radius(h0, pi.w, k.boxsize, k.margin_widen);
                tc_particle_levels(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift, k.lmax, h0, rg, &lmin, &lmaxp);
            }
            const int Lq = tc_query_level(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift + a.xshift, lmin, lmaxp, hq);
            wave_lds_fence();                                         /* the stack shares its LDS with the hit buffers */
            nr[lane] = min(ordered_runs_pf(k, a.pf, valid, pi.x, pi.y, pi.z, hq, Lq, lds_inv64, lds_perm, stk + lane,
                                           wruns + lane, a.flags), TC_XRUNCAP);
        }
        /* the runs were written one lane per particle and are read by the particle's quad */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        wave_lds_fence();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        if (a.dbg & 1) return;
        /* ---- A2 / B, four lanes per particle, 16 particles per pass */
        for (int pass = 0; pass < 4; pass++) {
            if (base + 16 * pass >= stop) break;
            const int pl = 16 * pass + q;
            const int t = base + pl;
            const int tt = t < stop ? t : base;
            const int i = k.own ? (int)k.own[tt] : tt;
            const bool valid = t < stop;
            const float4 pi = k.pos4[i];
            const float xi = pi.x, yi = pi.y, zi = pi.z;
            const float hq = (float)((double)pi.w * k.boxsize);             /* src/wvt_relax.c:135 */
            const float hq2 = hq * hq;
            const int nruns = nr[pl];
            const uint2 *pruns = wruns + pl;
            const double step_hi = a.step * (double)pi.w;                   /* step * hsml[ipart], src/wvt_relax.c:167 */
            int nlist = 0, cnt = 0;
            float dacc = 0;                                              /* delta[lc] of this quad's particle */
            /* no neighbour of an interior particle lies beyond half a box in any coordinate: the folding of
             * src/wvt_relax.c:148-154 cannot fire for any quad of this wave */
            const double ext = (double)hq * (1.0 + 1e-5) + k.boxsize * 1e-5;
            const bool wrap = tc_ballot(valid && !((double)xi >= ext && (double)xi <= k.boxsize - ext && (double)yi >= ext
                                                   && (double)yi <= k.boxsize - ext && (double)zi >= ext && (double)zi <= k.boxsize - ext)) != 0;
            /* Candidate stream with its loads one step ahead: the four positions of the NEXT step and the run after
             * the next are requested before the current step's positions are looked at, so a step never waits for a
             * load it has just issued.  (j, jend): the current run; (nj, njend): the next one; ri: runs fetched. */
            bool cdone = !valid || nruns == 0;
            uint32_t j = 0, jend = 0, nj = 0, njend = 0;
            int ri = 0;
            if (!cdone) {
                const uint2 r0 = pruns[0];
                j = r0.x; jend = r0.y;
                ri = 1;
                if (nruns > 1) { const uint2 r1 = pruns[64]; nj = r1.x; njend = r1.y; ri = 2; }
            }
            float4 pc = k.pos4[(!cdone && j + (uint32_t)lq < jend) ? j + (uint32_t)lq : (uint32_t)i];
            for (;;) {
                while (!tc_ballot(cnt > TC_X4CAP - 4) && tc_ballot(!cdone)) {
                    const uint32_t jc = j + (uint32_t)lq;
                    const bool act = !cdone && jc < jend;
                    /* where the next step reads */
                    uint32_t j2 = j + 4u, jend2 = jend;
                    const bool adv = !cdone && j2 >= jend;
                    bool last = false;
                    if (adv) {
                        last = njend == 0u;                              /* no run left */
                        j2 = nj; jend2 = njend;
                    }
                    const uint32_t jn = j2 + (uint32_t)lq;
                    const float4 pn = k.pos4[(!cdone && !last && jn < jend2) ? jn : (uint32_t)i];
                    if (adv) {
                        nj = njend = 0u;
                        if (ri < nruns) { const uint2 r = pruns[(size_t)ri * 64]; ri++; nj = r.x; njend = r.y; }
                    }
                    /* the f32 predicate of src/tree.c:67-89; its folding cannot fire for an interior ball (wave-uniform) */
                    const float r2 = ngb_r2_w(xi, yi, zi, pc.x, pc.y, pc.z, k.boxhalf_f, k.boxsize_f, wrap);
                    const bool hit = act && r2 < hq2;
                    /* the particle itself is on the reference's list but not in the sum (src/wvt_relax.c:141-142); it is
                     * among this step's candidates when i is in [j, j + 4) of the run (the same for the quad's four lanes) */
                    bool keep = hit && jc != (uint32_t)i;
                    const bool selfhere = !cdone && (uint32_t)i - j < 4u && (uint32_t)i < jend;
                    if (tc_ballot(nlist + 4 >= TC_NGBMAX)) {
                        /* some list of this wave is about to end at its NGBMAX-th entry (src/tree.c:91-92): a hit stays
                         * only if its place on the list (hits before it in this step, the particle itself included) exists */
                        const uint32_t mh = quad_bits(tc_ballot(hit), lane);
                        keep = keep && nlist + (int)__popc(mh & below) < TC_NGBMAX;
                    }
                    const uint32_t mk = quad_bits(tc_ballot(keep), lane);
                    if (keep) buf[cnt + (int)__popc(mk & below)] = pc;
                    cnt += (int)__popc(mk);
                    nlist += (int)__popc(mk) + (selfhere ? 1 : 0);   /* length of the reference's list so far */
                    if (!cdone) {
                        j = j2; jend = jend2;
                        if (last) cdone = true;
                        if (nlist >= TC_NGBMAX) {
                            cdone = true;
                            if (lq == 0) atomicAdd(&a.flags[4], 1);
                        }
                    }
                    pc = pn;
                }
                wave_lds_fence();
                if (a.dbg & 2) cnt = 0;
                for (int s = 0; tc_ballot(s < cnt); s += 4) {
                    const bool ok = s + lq < cnt;
                    const float4 pj = buf[ok ? s + lq : 0];
                    float dx = (float)((double)(xi - pj.x) * boxinv);
                    float dy = (float)((double)(yi - pj.y) * boxinv);
                    float dz = (float)((double)(zi - pj.z) * boxinv);
                    if (wrap) {                                         /* wave-uniform */
                        dx = dx > 0.5f ? dx - 1.0f : dx;                /* src/wvt_relax.c:148-154 (0.5 is exact in f32) */
                        dy = dy > 0.5f ? dy - 1.0f : dy;
                        dz = dz > 0.5f ? dz - 1.0f : dz;
                        dx = dx < -0.5f ? dx + 1.0f : dx;
                        dy = dy < -0.5f ? dy + 1.0f : dy;
                        dz = dz < -0.5f ? dz + 1.0f : dz;
                    }
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    const float h = (float)(0.5 * (double)(pi.w + pj.w));
                    const bool in = ok && !(r2 > h * h);
                    double e0 = 0, e1 = 0, e2 = 0;
                    {
                        const float r = tc_sqrt_f32_lean_pos(r2);                /* == (float)sqrt((double)r2) */
                        const double u = (double)tc_div_f32_lean(r, h);          /* src/wvt_relax.c:277: f32 quotient */
                        const double tq = 1 - u;
                        const float wk = (float)(TC_WC6_NORM * tq * tq * tq * tq * tq * tq * tq * tq
                                                 * (1 + 8 * u + 25 * u * u + 32 * u * u * u));
                        const double b = step_hi * (double)wk;
                        const double rd = (double)r;
                        /* three quotients by the same r: one reciprocal refinement, then the residual correction
                         * of tc_div_f64_lean per numerator */
                        double y = __builtin_amdgcn_rcp(rd);
                        double ee = __builtin_fma(-rd, y, 1.0);
                        y = __builtin_fma(y, ee, y);
                        ee = __builtin_fma(-rd, y, 1.0);
                        y = __builtin_fma(y, ee, y);
                        const double n0 = b * (double)dx, n1 = b * (double)dy, n2 = b * (double)dz;
                        const double q0 = n0 * y, q1 = n1 * y, q2 = n2 * y;
                        if (in) {
                            e0 = __builtin_fma(__builtin_fma(-rd, q0, n0), y, q0);
                            e1 = __builtin_fma(__builtin_fma(-rd, q1, n1), y, q1);
                            e2 = __builtin_fma(__builtin_fma(-rd, q2, n2), y, q2);
                        }
                    }
                    if (tc_ballot(in && r2 < 1e-24f)) {                 /* coincident particles: the IEEE sequences (0 / 0 and all) */
                        if (in && r2 < 1e-24f) {
                            const float r = sqrtf(r2);
                            const float wk = (float)tc_wvt_wc6(r, h);
                            e0 = step_hi * (double)wk * (double)dx / (double)r;
                            e1 = step_hi * (double)wk * (double)dy / (double)r;
                            e2 = step_hi * (double)wk * (double)dz / (double)r;
                        }
                    }
                    /* src/wvt_relax.c:167-169: the f32 accumulators take the four terms in order, rounded after each (a
                     * lane without a term contributes 0.0, which changes nothing).  The quad's 4 x 3 terms are transposed
                     * through LDS: lane c then owns component c and adds the four hits' terms one after the other. */
                    tq[3 * lq] = e0; tq[3 * lq + 1] = e1; tq[3 * lq + 2] = e2;
                    wave_lds_fence();
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) dacc = (float)((double)dacc + tq[3 * kk + lc]);
                    wave_lds_fence();
                }
                cnt = 0;
                wave_lds_fence();
                if (!tc_ballot(!cdone)) break;
            }
            if (valid && lq < 3) a.delta[3 * (size_t)a.lg[i] + lq] = dacc;
        }
    }, group);
}

/* The sweep on k_iter's neighbour lists (WVT == 2): the lists are in ascending index already, so what is left of the
 * exact sweep is phase B alone -- four lanes per particle read four list entries and their positions per step, evaluate
 * src/wvt_relax.c:141-169 statement for statement and add the terms to the f32 accumulators in order (as k_wvt_exact4). */
__global__ __launch_bounds__(TBN) void k_wvt_chain4(tc_xwvt_args a)
{
    __shared__ double lds_terms[WPB * 64 * 3];
    const tc_dev_const &k = a.k;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int q = lane >> 2, lq = lane & 3;
    double *tq = lds_terms + ((size_t)wave * 16 + q) * 12;
    const int lc = lq < 3 ? lq : 0;
    const double boxinv = k.boxinv;
    work_queue64(k, [&](int base, int stop) {
        for (int pass = 0; pass < 4; pass++) {
            if (base + 16 * pass >= stop) break;
            const int t = base + 16 * pass + q;
            const int tt = t < stop ? t : base;
            const int i = k.own ? (int)k.own[tt] : tt;
            const uint32_t nl = a.xlcnt[i];
            const bool valid = t < stop && nl != TC_XNONE;
            const int cnt = valid ? (int)nl : 0;
            const float4 pi = k.pos4[i];
            const float xi = pi.x, yi = pi.y, zi = pi.z;
            const float hq = (float)((double)pi.w * k.boxsize);
            const double ext = (double)hq * (1.0 + 1e-5) + k.boxsize * 1e-5;
            const bool wrap = tc_ballot(valid && !((double)xi >= ext && (double)xi <= k.boxsize - ext && (double)yi >= ext
                                                   && (double)yi <= k.boxsize - ext && (double)zi >= ext && (double)zi <= k.boxsize - ext)) != 0;
            const double step_hi = a.step * (double)pi.w;                   /* step * hsml[ipart], src/wvt_relax.c:167 */
            const uint32_t *lst = a.xlist + (size_t)i * TC_XLCAP;
            float dacc = 0;
            /* the list entry and the position of the next step are requested before this step's are looked at */
            uint32_t jn = lq < cnt ? lst[lq] : (uint32_t)i;
            float4 pn = k.pos4[jn];
            for (int s = 0; tc_ballot(s < cnt); s += 4) {
                const bool ok = s + lq < cnt;
                const float4 pj = pn;
                const uint32_t j2 = s + 4 + lq < cnt ? lst[s + 4 + lq] : (uint32_t)i;
                pn = k.pos4[j2];
                float dx = (float)((double)(xi - pj.x) * boxinv);
                float dy = (float)((double)(yi - pj.y) * boxinv);
                float dz = (float)((double)(zi - pj.z) * boxinv);
                if (wrap) {                                             /* wave-uniform */
                    dx = dx > 0.5f ? dx - 1.0f : dx;                    /* src/wvt_relax.c:148-154 (0.5 is exact in f32) */
                    dy = dy > 0.5f ? dy - 1.0f : dy;
                    dz = dz > 0.5f ? dz - 1.0f : dz;
                    dx = dx < -0.5f ? dx + 1.0f : dx;
                    dy = dy < -0.5f ? dy + 1.0f : dy;
                    dz = dz < -0.5f ? dz + 1.0f : dz;
                }
                const float r2 = dx * dx + dy * dy + dz * dz;
                const float h = (float)(0.5 * (double)(pi.w + pj.w));
                const bool in = ok && !(r2 > h * h);
                double e0 = 0, e1 = 0, e2 = 0;
                {
                    const float r = tc_sqrt_f32_lean_pos(r2);                    /* == (float)sqrt((double)r2) */
                    const double u = (double)tc_div_f32_lean(r, h);              /* src/wvt_relax.c:277: f32 quotient */
                    const double tt8 = 1 - u;
                    const float wk = (float)(TC_WC6_NORM * tt8 * tt8 * tt8 * tt8 * tt8 * tt8 * tt8 * tt8
                                             * (1 + 8 * u + 25 * u * u + 32 * u * u * u));
                    const double b = step_hi * (double)wk;
                    const double rd = (double)r;
                    double y = __builtin_amdgcn_rcp(rd);
                    double ee = __builtin_fma(-rd, y, 1.0);
                    y = __builtin_fma(y, ee, y);
                    ee = __builtin_fma(-rd, y, 1.0);
                    y = __builtin_fma(y, ee, y);
                    const double n0 = b * (double)dx, n1 = b * (double)dy, n2 = b * (double)dz;
                    const double q0 = n0 * y, q1 = n1 * y, q2 = n2 * y;
                    if (in) {
                        e0 = __builtin_fma(__builtin_fma(-rd, q0, n0), y, q0);
                        e1 = __builtin_fma(__builtin_fma(-rd, q1, n1), y, q1);
                        e2 = __builtin_fma(__builtin_fma(-rd, q2, n2), y, q2);
                    }
                }
                if (tc_ballot(in && r2 < 1e-24f)) {                     /* coincident particles: the IEEE sequences (0 / 0 and all) */
                    if (in && r2 < 1e-24f) {
                        const float r = sqrtf(r2);
                        const float wk = (float)tc_wvt_wc6(r, h);
                        e0 = step_hi * (double)wk * (double)dx / (double)r;
                        e1 = step_hi * (double)wk * (double)dy / (double)r;
                        e2 = step_hi * (double)wk * (double)dz / (double)r;
                    }
                }
                tq[3 * lq] = e0; tq[3 * lq + 1] = e1; tq[3 * lq + 2] = e2;
                wave_lds_fence();
#pragma unroll
                for (int kk = 0; kk < 4; kk++) dacc = (float)((double)dacc + tq[3 * kk + lc]);   /* src/wvt_relax.c:167-169 */
                wave_lds_fence();
            }
            if (valid && lq < 3) a.delta[3 * (size_t)a.lg[i] + lq] = dacc;
        }
    });
}

/* The same sweep for the FEW particles k_iter could not list (more neighbours than a list holds, a cut list, no record):
 * one WAVEFRONT per particle.  Throughput is poor -- the f32 chain keeps three lanes busy for a thousand dependent
 * additions -- but latency is what counts for a handful of particles behind 2e6 others: a group of k_wvt_exact4 needs a
 * millisecond whatever its size, a wavefront here ~0.1 ms.  Lane 0 walks the cells in curve order (ordered_runs_pf), the wave
 * streams the runs (stream_runs), hits are compacted in index order 64 at a time, their terms evaluated lane-parallel
 * (src/wvt_relax.c:141-169, the arithmetic of k_wvt_exact4) and added in order by lanes 0..2 (x, y, z). */
__global__ __launch_bounds__(TBN) void k_wvt_exact_w(tc_xwvt_args a)
{
    __shared__ uint32_t lds_heads[WPB * 256];
    __shared__ __align__(16) float4 lds_ring[WPB * 128];
    __shared__ double lds_term[WPB * 64 * 3];
    __shared__ uint32_t lds_stk[WPB * (TC_MAX_LEVEL + 1) * 64];
    __shared__ uint2 lds_dense[WPB * TC_XRUNCAP];
    __shared__ uint64_t lds_inv64[TC_HILBERT_NSTATES];
    __shared__ unsigned char lds_perm[TC_HILBERT_NSTATES * 256];
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES; t += TBN) {
        uint64_t r = 0;
        for (int kk = 0; kk < 8; kk++) r |= (uint64_t)TC_HILBERT_INV[t * 8 + kk] << (8 * kk);
        lds_inv64[t] = r;
    }
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES * 256; t += TBN) {
        const int stt = t >> 8, msk = t & 255;
        uint32_t cm = 0;
        for (int kk = 0; kk < 8; kk++) cm |= ((msk >> (TC_HILBERT_INV[stt * 8 + kk] & 7)) & 1u) << kk;
        lds_perm[t] = (unsigned char)cm;
    }
    __syncthreads();
    tc_dev_const kw = a.k;
    if (a.wl) { kw.lo = 0; kw.hi = *a.wl_cnt; kw.own = a.wl; }
    const tc_dev_const &k = kw;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint2 *dense = lds_dense + (size_t)wave * TC_XRUNCAP;
    uint32_t *heads = lds_heads + wave * 256;
    float4 *ring = lds_ring + wave * 128;
    double *term = lds_term + wave * 192;
    uint32_t *stk = lds_stk + (size_t)wave * (TC_MAX_LEVEL + 1) * 64 + lane;
    uint2 *wruns = a.runs + (size_t)(blockIdx.x * WPB + wave) * ((size_t)TC_XRUNCAP * 64);
    const double boxinv = k.boxinv;
    work_queue(k, [&](int i) {
        const float4 pv = k.pos4[i];
        const float4 pi = make_float4(U(pv.x), U(pv.y), U(pv.z), U(pv.w));
        const float xi = pi.x, yi = pi.y, zi = pi.z;
        const float hq = (float)((double)pi.w * k.boxsize);             /* src/wvt_relax.c:135 */
        const float hq2 = hq * hq;
        int lmin = k.lmin_tab, lmaxp = k.lmax;
        if (k.margin_on) {
            const float h0 = a.hsml0[i];
            const float rg = tc_margin_radius(h0, pi.w, k.boxsize, k.margin_widen);
            tc_particle_levels(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift, k.lmax, h0, rg, &lmin, &lmaxp);
        }
        const int Lq = tc_query_level(k.boxsize, k.box_mant, k.box_exp, k.level_scale, k.level_shift + a.xshift, lmin, lmaxp, hq);
        /* lane 0 alone walks the cells: its runs land at wruns[64 s] */
        int nruns = ordered_runs_pf(k, a.pf, lane == 0, xi, yi, zi, hq, Lq, lds_inv64, lds_perm, stk, wruns, a.flags);
        nruns = min(U(nruns), TC_XRUNCAP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        /* ... and are repacked densely (LDS) for the streaming */
        wave_lds_fence();
        for (int q = lane; q < nruns; q += 64) dense[q] = wruns[(size_t)q * 64];
        wave_lds_fence();

        const double ext = (double)hq * (1.0 + 1e-5) + k.boxsize * 1e-5;
        const bool wrap = !((double)xi >= ext && (double)xi <= k.boxsize - ext && (double)yi >= ext
                            && (double)yi <= k.boxsize - ext && (double)zi >= ext && (double)zi <= k.boxsize - ext);
        const double step_hi = a.step * (double)pi.w;                   /* step * hsml[ipart], src/wvt_relax.c:167 */
        float dacc = 0;                                                 /* lane c < 3: delta[c] */
        int nlist = 0, scnt = 0, head = 0;
        /* 64 staged neighbours (ring, index order): their terms lane-parallel, then the ordered chain */
        auto convert = [&](int nvalid) {
            wave_lds_fence();
            const float4 pj = ring[(head + lane) & 127];
            float dx = (float)((double)(xi - pj.x) * boxinv);
            float dy = (float)((double)(yi - pj.y) * boxinv);
            float dz = (float)((double)(zi - pj.z) * boxinv);
            if (wrap) {
                dx = dx > 0.5f ? dx - 1.0f : dx;                        /* src/wvt_relax.c:148-154 (0.5 is exact in f32) */
                dy = dy > 0.5f ? dy - 1.0f : dy;
                dz = dz > 0.5f ? dz - 1.0f : dz;
                dx = dx < -0.5f ? dx + 1.0f : dx;
                dy = dy < -0.5f ? dy + 1.0f : dy;
                dz = dz < -0.5f ? dz + 1.0f : dz;
            }
            const float r2 = dx * dx + dy * dy + dz * dz;
            const float h = (float)(0.5 * (double)(pi.w + pj.w));
            const bool in = lane < nvalid && !(r2 > h * h);
            double e0 = 0, e1 = 0, e2 = 0;
            if (in) {
                if (r2 < 1e-24f) {                                      /* coincident particles: the IEEE sequences */
                    const float r = sqrtf(r2);
                    const float wk = (float)tc_wvt_wc6(r, h);
                    e0 = step_hi * (double)wk * (double)dx / (double)r;
                    e1 = step_hi * (double)wk * (double)dy / (double)r;
                    e2 = step_hi * (double)wk * (double)dz / (double)r;
                } else {
                    const float r = tc_sqrt_f32_lean_pos(r2);
                    const double u = (double)tc_div_f32_lean(r, h);
                    const double t1 = 1 - u;
                    const float wk = (float)(TC_WC6_NORM * t1 * t1 * t1 * t1 * t1 * t1 * t1 * t1
                                             * (1 + 8 * u + 25 * u * u + 32 * u * u * u));
                    const double b = step_hi * (double)wk;
                    e0 = tc_div_f64_lean(b * (double)dx, (double)r);
                    e1 = tc_div_f64_lean(b * (double)dy, (double)r);
                    e2 = tc_div_f64_lean(b * (double)dz, (double)r);
                }
            }
            term[3 * lane] = e0; term[3 * lane + 1] = e1; term[3 * lane + 2] = e2;
            wave_lds_fence();
            if (lane < 3)
                for (int q = 0; q < nvalid; q++) dacc = (float)((double)dacc + term[3 * q + lane]);   /* src/wvt_relax.c:167-169 */
            head = U((head + 64) & 127);
            wave_lds_fence();
        };
        bool cut = false;
        stream_runs(k, dense, nruns, heads, [&](uint32_t j, float4 p, bool act) -> bool {
            const float4 pf4 = k.pos4[j < (uint32_t)k.n ? j : (uint32_t)i];           /* w (hsml_j) too: the stream loads x, y, z */
            const float r2 = ngb_r2_w(xi, yi, zi, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f, wrap);
            const bool hit = act && r2 < hq2;
            const uint64_t mh = tc_ballot(hit);
            /* the reference's list ends at its NGBMAX-th entry (src/tree.c:91-92); the particle itself is on it, not in the sum */
            const bool keep = hit && nlist + mask_rank(mh) < TC_NGBMAX && j != (uint32_t)i;
            const uint64_t mk = tc_ballot(keep);
            if (keep) ring[(head + scnt + mask_rank(mk)) & 127] = pf4;
            scnt = U(scnt + (int)__popcll(mk));
            nlist = U(nlist + (int)__popcll(mh));
            if (scnt >= 64) { convert(64); scnt = U(scnt - 64); }
            if (nlist >= TC_NGBMAX) { cut = true; return true; }
            return false;
        });
        if (scnt > 0) convert(scnt);
        if (cut && lane == 0) atomicAdd(&a.flags[4], 1);
        if (lane < 3) a.delta[3 * (size_t)a.lg[i] + lane] = dacc;
    });
}

template <class K>
static int xgrid(const tcgpu_ctx *c, int nloc, K kernel)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TBN, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (c->blocks_per_cu > 0 && c->blocks_per_cu < per_cu) per_cu = c->blocks_per_cu;
    int need = (nloc + 64 * WPB - 1) / (64 * WPB);
    int cap = c->num_cu * per_cu;
    if (cap > TC_MAX_PERSISTENT_BLOCKS) cap = TC_MAX_PERSISTENT_BLOCKS;
    int g = need < cap ? need : cap;
    if (g >= 16) g &= ~7;
    return g;
}

int tc_launch_wvt_exact(tcgpu_ctx *c, double step)
{
    tc_xwvt_args a;
    tc_fill_const(c, &a.k);
    a.step = step;
    a.delta = c->delta;
    a.lg = c->lg;
    a.hsml0 = c->hsml0;
    a.flags = c->flags;
    a.xshift = c->xsweep_shift;
    a.dbg = c->ablate;
    const bool lists = c->xlist_valid && c->sweep_mode == 0 && !c->xsweep_kernel;
    a.xlist = lists ? c->xlist : nullptr;
    a.xlcnt = lists ? c->xlcnt : nullptr;
    a.wl = lists ? c->xun : nullptr;
    a.wl_cnt = lists ? (const int *)(c->xun + c->xr_cap) : nullptr;
    int nloc = a.k.hi - a.k.lo;
    if (nloc <= 0) return 0;
    const int g4 = xgrid(c, nloc, k_wvt_exact4), g1 = xgrid(c, nloc, k_wvt_exact);
    int gw = grid_for(c, nloc, k_wvt_exact_w);
    if (gw > 1024) gw = 1024;
    int gm = g4 > g1 ? g4 : g1;
    if (gw > gm) gm = gw;
    const size_t want = (size_t)gm * WPB * TC_XRUNCAP * 64 * sizeof(uint2);
    if (c->xruns_bytes < want) {
        if (c->xruns) TC_HIP(c, hipFree(c->xruns));
        c->xruns = nullptr; c->xruns_bytes = 0;
        TC_HIP(c, hipMalloc(&c->xruns, want));
        c->xruns_bytes = want;
    }
    a.runs = (uint2 *)c->xruns;
    a.pf.tab = c->pf; a.pf.lmin = c->pf_lmin; a.pf.lc = c->pf_lc;
    a.orphans_only = 0;
    tc_phase_begin(c, PH_WVT);
    if (lists) {
        /* k_iter listed the neighbours in index order: k_wvt_chain4 evaluates the lists.  The few particles without a list
         * (one wavefront each, a serial chain of up to 2 360 additions: 0.3 ms of latency, next to no throughput) go to a
         * side stream FIRST and run underneath: the list kernel's persistent blocks take the slots they leave and the
         * rest as they finish (dynamic work queue, no block waits for another).  They write disjoint particles. */
        TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 2 * 8 * 16 * sizeof(int), c->stream));
        TC_HIP(c, hipEventRecord(c->ev_fork, c->stream));
        TC_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        tc_xwvt_args a2 = a;
        a2.k.work_ctr = c->work_ctr + 8 * 16;
        k_wvt_exact_w<<<gw, TBN, 0, c->stream2>>>(a2);
        TC_HIP(c, hipEventRecord(c->ev_join, c->stream2));
        k_wvt_chain4<<<xgrid(c, nloc, k_wvt_chain4), TBN, 0, c->stream>>>(a);
        TC_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        tc_phase_end(c);
        TC_HIP(c, hipGetLastError());
        return 0;
    }
    TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 8 * 16 * sizeof(int), c->stream));
    /* option "xsweep_kernel" = 1 (tests): the one-lane-per-particle kernel on the (x, y, z) cell table -- an independent
     * second implementation of the same sums.  (A third layout -- one lane per particle with candidates served from LDS
     * tiles of the index space -- was tried in round 3 and dropped: the Peano runs of a ball are short and scattered, a
     * group of 64 particles touches hundreds of 128-particle tiles: 26 ms.) */
    if (c->xsweep_kernel == 1) k_wvt_exact<<<g1, TBN, 0, c->stream>>>(a);
    else k_wvt_exact4<<<g4, TBN, 0, c->stream>>>(a);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ fused K5 + K9: one gather per particle */

/*
 * The reference queries the ball of the carried hsml, very often finds fewer than 295 raw
 * neighbours, retries at 1.23 hsml (src/sph.c:49-54), and later queries a third, almost equal ball
 * for the WVT sweep (src/wvt_relax.c:135).  On warm iterations this kernel gathers ONCE at
 * R = max(1.23 hsml, hsml_wvt*box) and keeps for every hit its f32 r2, which is all the reference's
 * predicates look at: the list of the first query is {r2 < hsml^2} ("inner"), the list of the retry
 * is inner + {r2 < (1.23 hsml)^2} ("outer"), the sweep's list is {r2 < (hsml_wvt*box)^2}.  The
 * reference's control flow then runs on exactly those lists.  Anything unusual (cold start, NGBMAX
 * overflow, a third query) falls back to the plain per-query code above.
 * The sweep is accumulated for unit step: the step size is decided by the host from this very
 * pass's error sums (src/wvt_relax.c:89-101); k_move applies delta = step * U.
 */
/* What a wavefront needs to know about its particle before it can stream candidates -- the three radii, the table
 * level, the cell box of the query -- is ~150 instructions of mostly f64 arithmetic on wave-uniform values: executed
 * by 64 lanes for one particle it is as expensive as 150 instructions of the neighbour loops.  k_prec works it out
 * for 64 particles per wavefront instead (one lane each, the same functions) and leaves a 64-byte record that the
 * fused kernel reads with scalar loads. */
struct tc_prec {
    float h0, hb, hw, R;              /* carried hsml, 1.23 hsml, sweep radius, gather radius */
    float h0sq, hbsq, hwsq, smax;
    float sf, hpf, inv_ny, inv_sf;    /* tc_query */
    int lo[3];
    uint32_t pack;                    /* qL : 4 | flags : 7 | nd[0..2] : 7 each */
};
#define TC_PREC_WARM 1u
#define TC_PREC_WRAP 2u
#define TC_PREC_FAST 4u
#define TC_PREC_FULL0 8u              /* FULL0 << d: dimension d covers the whole ring (nd = 2^qL) */
#define TC_PREC_VALID 64u             /* the record describes the query (else: work it out in the kernel) */

/* compile-time flavours of the gather of k_iter: std::true_type = mirror slots of an interior ball (no folding, positions and
 * "self" through the mirror), std::false_type = particle indices with the folding decided at run time, tc_tag_ord = particle
 * indices of an interior ball (no folding), positions through the parked pointer */
struct tc_tag_ord { static constexpr bool value = true; };

struct tc_iter_args {
    tc_density_args d;
    double *ustep;             /* 3n, unit-step displacement sums (WVT == 1) */
    const tc_prec *prec;       /* one record per local slot of an own particle */
    /* WVT == 2: the candidates come in ascending index from per-particle run lists (k_xruns), and the sweep's
     * neighbours leave the kernel as index lists in that order for k_wvt_chain4 */
    const uint2 *xr;           /* [i * TC_XRCAP + s] */
    const uint32_t *xrn;       /* runs of particle i, TC_XNONE: none */
    uint32_t *xlist;           /* [i * TC_XLCAP + s] */
    uint32_t *xlcnt;           /* neighbours listed, TC_XNONE: not listed (k_wvt_exact4 takes the particle) */
    uint32_t *xun;             /* ... and the particles not listed, *xun_cnt of them */
    int *xun_cnt;
};

__global__ __launch_bounds__(256) void k_prec(tc_dev_const k0, const float *__restrict__ hsml_in, int do_wvt,
                                              int no_records, tc_prec *__restrict__ prec)
{
    const int t = k0.lo + blockIdx.x * 256 + threadIdx.x;
    if (t >= k0.hi) return;
    const int i = k0.own ? (int)k0.own[t] : t;
    const float4 pv = k0.pos4[i];
    const float h0 = hsml_in[i];
    tc_prec P;
    P.pack = 0;
    P.h0 = h0;
    P.hb = (float)((double)h0 * 1.23);
    P.hw = (float)((double)pv.w * k0.boxsize);                      /* src/wvt_relax.c:135 */
    P.h0sq = P.h0 * P.h0; P.hbsq = P.hb * P.hb; P.hwsq = P.hw * P.hw;
    P.R = (do_wvt && P.hw > P.hb) ? P.hw : P.hb;
    P.smax = (P.hwsq > P.hbsq && do_wvt) ? P.hwsq : P.hbsq;
    P.sf = P.hpf = P.inv_ny = P.inv_sf = 0;
    P.lo[0] = P.lo[1] = P.lo[2] = 0;
    if (isfinite(h0) && h0 != 0) {
        const tc_dev_const k = particle_view<false>(k0, h0, pv.w);
        tc_query q;
        query_setup<false>(k, pv.x, pv.y, pv.z, P.R, q);
        const int qL = q.L;
        const double ext = (double)P.R * (1.0 + 1e-5) + k.boxsize * 1.2e-5 + query_cell_edge_at(k, qL);
        const bool wrap = !((double)pv.x >= ext && (double)pv.x <= k.boxsize - ext && (double)pv.y >= ext
                            && (double)pv.y <= k.boxsize - ext && (double)pv.z >= ext && (double)pv.z <= k.boxsize - ext);
        const bool fast = !wrap && k.mirror != nullptr && qL <= k.lmax_rm && qL >= k.lmin_rm;
        uint32_t fl = TC_PREC_WARM | (wrap ? TC_PREC_WRAP : 0u) | (fast ? TC_PREC_FAST : 0u);
        bool fits = qL < 16;
        uint32_t nds = 0;
        for (int d = 0; d < 3; d++) {
            if (q.full[d]) fl |= TC_PREC_FULL0 << d;
            else if (q.nd[d] > 127) fits = false;
            else nds |= (uint32_t)q.nd[d] << (7 * d);
            P.lo[d] = q.lo[d];
        }
        if (fits && !no_records) fl |= TC_PREC_VALID;      /* no_records (tests): every particle down the plain path */
        P.sf = q.sf; P.hpf = q.hpf; P.inv_ny = q.inv_ny; P.inv_sf = q.inv_sf;
        P.pack = (uint32_t)qL | (fl << 4) | (nds << 11);
    }
    prec[i] = P;
}

/* Sized for 4 waves per SIMD (<= 128 VGPRs, 4 blocks x 38.9 KB LDS per CU): measured 4 % faster than 3 waves
 * with 512/384/512 (tools/try_libs.sh, same box).  Longer lists continue in the per-wave global spill.
 * Round 2 tried 5 waves per SIMD (one 640-entry region shared by the two lists growing towards each other = 32 KB per
 * block, 96 VGPRs): the occupancy sweep (tools/occupancy_sweep.py: 30.8 / 16.7 / 12.6 / 10.8 ms at 1 / 2 / 3 / 4 blocks
 * per CU) promised -10 %, the build lost 4 % (24 VGPRs spilled to scratch, more scalar spills) -- not taken. */
#ifdef TC_TRY_W5               /* experiment (make w5): 8 KB of LDS per wave and 96 VGPRs for a fifth wave per SIMD */
#define TC_ICAP 384
#define TC_OCAP 256
#define TC_ITER_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
#else
#define TC_ICAP 512            /* inner entries in LDS */
#define TC_OCAP 384            /* outer entries in LDS */
#define TC_ITER_ATTR
#endif
#define TC_ITER_IDXCAP 256
#define TC_ITER_MINWAVES 4
#define TC_LDS_PER_WAVE_ITER ((TC_ICAP + TC_OCAP) * sizeof(double) + TC_ITER_IDXCAP * sizeof(uint32_t) + 4 * TC_STAGE * sizeof(float))

#ifdef TC_PROFILE_ABLATE
/* profiling build only: begin / end time (100 MHz counter) of every wave of the last k_iter launch */
__device__ uint64_t g_wave_span[2 * TC_MAX_PERSISTENT_BLOCKS * WPB];
extern "C" int tcgpu_debug_wave_spans(uint64_t *out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_span), sizeof(uint64_t) * 2 * (size_t)nwaves) == hipSuccess ? 0 : -1;
}
#endif

/* STATS: keep the per-particle work counters (queries, solver iterations, pair evaluations, candidates) that
 * tcgpu_last_density_stats reports; without them the counters are dead code and cost no scalar registers */
template <bool STATS, int WVT>      /* WVT: 0 density only, 1 + round 2's f64 sweep sums, 2 + the sweep's neighbours listed in index order */
__device__ __forceinline__ void iter_one(const tc_iter_args &a, int i, unsigned char *mine, double *spill TC_PROF_PARAM)
{
    TC_STAGE_SWITCH(ST_QUEUE, ST_PROLOGUE);
    const tc_density_args &da = a.d;
    const int lane = lane_id();
    /* the particle's own data is wave-uniform: keep it in scalar registers */
    const float4 pv = da.k.pos4[i];
    const tc_prec P = a.prec[i];                       /* scalar loads: the wave-uniform prologue was done by k_prec */
    const uint32_t pflags = U(P.pack >> 4) & 0x7fu;
    const bool have_prec = (pflags & TC_PREC_VALID) != 0;
    /* the per-particle view of the constants (level range of a sharded pass) is only needed off the common path */
    const tc_dev_const &k = da.k;
    const float4 pi = make_float4(U(pv.x), U(pv.y), U(pv.z), U(pv.w));
    const float xi = pi.x, yi = pi.y, zi = pi.z;
    constexpr bool do_wvt = WVT != 0;
    constexpr bool ord = WVT == 2;
    /* WVT == 2: this particle's ordered runs, if it has them */
    const uint32_t nxr = ord ? U(a.xrn[i]) : TC_XNONE;
    int cwout = 0;                                     /* sweep neighbours written to the list */

    double *lds_lists = reinterpret_cast<double *>(mine);
    uint32_t *idx = reinterpret_cast<uint32_t *>(lds_lists + TC_ICAP + TC_OCAP);
    /* two staging rings of particle indices: density hits (j, r2) and sweep hits (j); positions are
     * re-read at conversion time (they are L1/L2-hot), which keeps the rings small enough for the
     * hit lists to stay in LDS at 4 waves per SIMD */
    uint32_t *dj = idx + TC_ITER_IDXCAP;
    float *dr2 = reinterpret_cast<float *>(dj + TC_STAGE);
    uint32_t *wj = reinterpret_cast<uint32_t *>(dr2 + TC_STAGE);
    /* without round 2's sweep (WVT != 1) there is no second ring, and the area holds (j, x, y, z) per staged hit: the
     * conversion then needs no second read of the positions -- a dependent gather whose latency four waves per SIMD
     * did not cover (stage timers: convert_d 21 % of the waves' lives); r2 is recomputed there by the same function */
    constexpr bool spos = WVT != 1;
    float4 *dq = reinterpret_cast<float4 *>(dj);
    /* the plain fallback code stages positions: it reuses the whole ring area (4 x TC_STAGE floats) */
    tc_stage st;
    st.x = reinterpret_cast<float *>(dj);
    st.y = st.x + TC_STAGE; st.z = st.y + TC_STAGE; st.w = st.z + TC_STAGE;
    const tc_stage sw = st;
    const uint32_t idxcap = TC_ITER_IDXCAP;
    tc_rlist plain;                                   /* the fallback path sees one list over both LDS parts */
    plain.lds = lds_lists; plain.spill = spill; plain.cap = TC_ICAP + TC_OCAP;

    tc_dstate d;
    const bool finite = density_init(da, i, d);
    const bool warm = finite && da.hsml_in[i] != 0;
    double u0 = 0, u1 = 0, u2 = 0;
    bool wvt_done = false;

    if (warm && have_prec) {
        const float hb = U(P.hb), h0sq = U(P.h0sq), hbsq = U(P.hbsq), hwsq = U(P.hwsq);
        const double boxinv = k.boxinv;
        const double step_hi = (double)pi.w;                          /* unit step */
        /* wrap: some candidate may lie beyond box/2 in a coordinate; fast: interior ball at a mirrored level --
         * candidates come as contiguous runs of the row-major mirror (stream_rows), `j` is then a mirror slot.  No
         * orphan (coordinate == boxsize) can be within R of an interior particle, so skipping them there changes nothing. */
        const bool wrap = (pflags & TC_PREC_WRAP) != 0;
        const bool fast = (pflags & TC_PREC_FAST) != 0;
        tc_query q;
        {
            const uint32_t pk = U(P.pack);
            q.L = (int)(pk & 15u);
            q.nL = 1 << q.L;
            const tc_level_desc D = k.lvl[q.L];
            q.off = U(D.off);
            q.ox = U(D.ox); q.oy = U(D.oy); q.oz = U(D.oz); q.ny = U(D.ny); q.nz = U(D.nz);
#pragma unroll
            for (int dd = 0; dd < 3; dd++) {
                q.full[dd] = (pflags & (TC_PREC_FULL0 << dd)) != 0;
                q.nd[dd] = q.full[dd] ? q.nL : (int)((pk >> (11 + 7 * dd)) & 127u);
                q.lo[dd] = U(P.lo[dd]);
            }
            q.sf = U(P.sf); q.hpf = U(P.hpf); q.inv_ny = U(P.inv_ny); q.inv_sf = U(P.inv_sf);
        }
        /* the gather below is compiled twice (tag F): on the row-run path nothing wraps, positions come
         * from the mirror and "self" is the slot whose Peano index is i -- all compile-time there */

        const tc_gpos vmirror = vgpr_pos(ord ? k.pos4 : k.mirror);   /* WVT == 2 never touches the mirror */
        tc_list2 L;
        L.in.lds = lds_lists;            L.in.spill = spill;               L.in.cap = TC_ICAP;
        L.out.lds = lds_lists + TC_ICAP; L.out.spill = spill + TC_NGBMAX;  L.out.cap = TC_OCAP;
        int cs = 0, co = 0, cw = 0;
        int dcnt = 0, dhead = 0, wcnt = 0, whead = 0;

        /* The sweep's ball is nearly always the smaller one (hsml_wvt * box = 0.96 ... 1.0 hsml on a relaxing state,
         * 1.23 hsml for the density ball), so candidates are tested ONCE against the larger of the two squares and
         * staged in one ring; which lists a staged hit belongs to is decided 64 hits at a time in convert_d, on
         * full waves, from the f32 r2 staged with it -- the reference's predicates exactly (src/tree.c:67-89). */
        float smax = P.smax;
        asm volatile("v_mov_b32 %0, %0" : "+v"(smax));          /* parked in a VGPR: the scalar file is full (a spilled
                                                                 * SGPR costs a v_readlane per candidate batch) */
        bool stopped = false;

        /* 64 staged sweep hits -> pair terms */
        auto convert_w = [&](auto ftag, int nvalid) {
            constexpr bool F = decltype(ftag)::value;
            constexpr bool MIR = std::is_same<std::decay_t<decltype(ftag)>, std::true_type>::value;
            const bool wr = F ? false : wrap;
            TC_STAGE_SWITCH(ST_CONVERT_D, ST_CONVERT_W);
            wave_lds_fence();
            int sl = (whead + lane) & (TC_STAGE - 1);
            if (ord) {
                /* the neighbours reach this point in ascending index: 64 more entries of the particle's list */
                if (lane < nvalid && cwout + lane < TC_XLCAP)      /* streamed out: not to displace the positions in L2 */
                    __builtin_nontemporal_store(wj[sl], &a.xlist[(size_t)i * TC_XLCAP + cwout + lane]);
                cwout = U(cwout + nvalid);
            } else {
            const float4 p = MIR ? ld4(vmirror, lane < nvalid ? wj[sl] : 0u) : k.pos4[lane < nvalid ? wj[sl] : (uint32_t)i];
            if (lane < nvalid && TC_ABLATE(k) != 3) wvt_pair(pi, p, boxinv, step_hi, u0, u1, u2, wr);
            }
            whead = U((whead + 64) & (TC_STAGE - 1));
            wave_lds_fence();
            TC_STAGE_SWITCH(ST_CONVERT_W, ST_CONVERT_D);
        };
        /* 64 staged hits -> f64 separations -> inner / outer list; sweep hits -> second ring (indices).
         * Always a full ring segment: the tail pads it with hits at r2 = infinity (pad_stage), so there is no
         * per-lane validity here, and the lane masks are ballots of the bare f32 compares (a ballot of a combined
         * predicate costs a select and a second compare): inner = r2 < h0^2 (h0 < hb), outer = in hb, not in h0.
         * Returns true when the density lists come close to NGBMAX (the caller then replays the particle with the
         * plain code, which is exact whenever it runs). */
        auto convert_d = [&](auto ftag) -> bool {
            constexpr bool F = decltype(ftag)::value;
            constexpr bool MIR = std::is_same<std::decay_t<decltype(ftag)>, std::true_type>::value;
            constexpr bool PARK = MIR || std::is_same<std::decay_t<decltype(ftag)>, tc_tag_ord>::value;
            const bool wr = F ? false : wrap;
            TC_STAGE_SWITCH(ST_TEST, ST_CONVERT_D);
            wave_lds_fence();
            int sl = (dhead + lane) & (TC_STAGE - 1);
            uint32_t jj;
            float r2;
            float4 pj;
            if (spos) {
                const float4 e = dq[sl];
                jj = __float_as_uint(e.x);
                pj = make_float4(e.y, e.z, e.w, 0.0f);
                r2 = ngb_r2_w(xi, yi, zi, pj.x, pj.y, pj.z, k.boxhalf_f, k.boxsize_f, wr);      /* as in gather(): same bits */
            } else {
                jj = dj[sl];
                r2 = dr2[sl];
                /* the gather of the 64 positions is issued first; the sweep's part of the work -- which only needs the
                 * staged index and r2, and every other time runs a whole batch of pair terms -- goes on underneath it */
                pj = PARK ? ld3(vmirror, jj) : k.pos4[jj];     /* 12 bytes: no register of the gather is free for reuse */
            }
            dhead = U((dhead + 64) & (TC_STAGE - 1));
            if (do_wvt) {
                const bool hwv = r2 < hwsq;
                uint64_t mw = tc_ballot(hwv);
                cw = U(cw + (int)__popcll(mw));
                /* the particle itself (the only hit at distance zero, bar coincident particles) is not a sweep
                 * neighbour: found once per particle, so the test sits behind a wave-uniform branch */
                bool use = hwv;
                if (tc_ballot(r2 == 0.0f)) {
                    const bool self = r2 == 0.0f && (MIR ? k.mirror_idx[jj] == (uint32_t)i : jj == (uint32_t)i);
                    use = hwv && !self;
                    mw = tc_ballot(use);
                }
                if (ord) {
                    /* the staged hits are in ascending index: this step's sweep neighbours go straight to the next places of
                     * the particle's list (streamed out: not to displace the positions in L2) */
                    const int at = cwout + mask_rank(mw);
                    if (use && at < TC_XLCAP) __builtin_nontemporal_store(jj, &a.xlist[(size_t)i * TC_XLCAP + at]);
                    cwout = U(cwout + (int)__popcll(mw));
                } else {
                if (use) {
                    int sl2 = (whead + wcnt + mask_rank(mw)) & (TC_STAGE - 1);
                    wj[sl2] = jj;
                }
                wcnt = U(wcnt + (int)__popcll(mw));
                if (wcnt >= 64) { convert_w(ftag, 64); wcnt = U(wcnt - 64); }
                }
            }
            const float x = pj.x, y = pj.y, z = pj.z;
            const bool b_hb = r2 < hbsq, b_h0 = r2 < h0sq;
            double r = 0;
            if (TC_ABLATE(k) != 3) r = pair_r_w(xi, yi, zi, x, y, z, k.boxhalf, k.boxsize, wr);
            const uint64_t m_hb = tc_ballot(b_hb), m_in = tc_ballot(b_h0);
            const uint64_t m_out = m_hb ^ m_in;                       /* h0 < hb: the inner hits are among m_hb */
            const int slot_in = cs + mask_rank(m_in), slot_out = co + mask_rank(m_out);
            if (cs + 64 <= TC_ICAP && co + 64 <= TC_OCAP) {        /* wave-uniform: everything lands in LDS */
                double *dst = b_h0 ? L.in.lds + slot_in : L.out.lds + slot_out;    /* one store per staged hit */
                if (b_hb) *dst = r;
            } else {
                if (b_h0) { if (slot_in < TC_NGBMAX) L.in.put(slot_in, r); }
                else if (b_hb) { if (slot_out < TC_NGBMAX) L.out.put(slot_out, r); }
            }
            cs = U(cs + (int)__popcll(m_in));
            co = U(co + (int)__popcll(m_out));
            wave_lds_fence();
            TC_STAGE_SWITCH(ST_CONVERT_D, ST_TEST);
            return cs + co + TC_STAGE >= TC_NGBMAX;
        };
        /* pad the ring segment [dcnt, 64) behind the staged hits with entries no predicate accepts */
        auto pad_stage = [&](uint32_t padj, int nvalid) {
            if (lane >= nvalid) {
                int sl = (dhead + lane) & (TC_STAGE - 1);
                if (spos) dq[sl] = make_float4(__uint_as_float(padj), HUGE_VALF, 0.0f, 0.0f);      /* r2 = infinity */
                else { dj[sl] = padj; dr2[sl] = HUGE_VALF; }
            }
        };
        auto gather = [&](auto ftag, uint32_t j, float4 p, bool act) -> bool {
            constexpr bool F = decltype(ftag)::value;
            const bool wr = F ? false : wrap;
            float r2 = ngb_r2_w(xi, yi, zi, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f, wr);
            const bool hd = act && (r2 < smax);
            const uint64_t md = tc_ballot(hd);
            if (TC_ABLATE(k) == 2) { cs += (int)__popcll(md); return false; }      /* profiling only */
            if (hd) {
                int sl = (dhead + dcnt + mask_rank(md)) & (TC_STAGE - 1);
                if (spos) dq[sl] = make_float4(__uint_as_float((uint32_t)j), p.x, p.y, p.z);
                else { dj[sl] = (uint32_t)j; dr2[sl] = r2; }
            }
            dcnt = U(dcnt + (int)__popcll(md));
            if (dcnt >= 64) {
                dcnt = U(dcnt - 64);
                if (convert_d(ftag)) { stopped = true; return true; }
            }
            return false;
        };
        bool overflow;
        if (ord && nxr != TC_XNONE) {
            /* candidates in ascending index from the particle's run list; an interior ball needs no folding */
            auto run_it = [&](auto F) {
                d.ncand += stream_runs(k, a.xr + (size_t)i * TC_XRCAP, (int)nxr, idx,
                                       [&](uint32_t j, float4 p, bool act) { return gather(F, j, p, act); } TC_PROF_PASS);
                overflow = stopped || cs + co + dcnt >= TC_NGBMAX;
                TC_STAGE_SWITCH(ST_EPILOGUE, ST_TEST);
                if (!overflow) {
                    if (dcnt > 0) { pad_stage((uint32_t)i, dcnt); convert_d(F); }
                    if (wcnt > 0) convert_w(F, wcnt);
                }
                TC_STAGE_SWITCH(ST_TEST, ST_EPILOGUE);
            };
            if (wrap) run_it(std::false_type());
            else run_it(tc_tag_ord());
        } else if (!ord && fast) {
            const std::true_type F;
            d.ncand += stream_rows_q<true>(k, q, xi, yi, zi, idx,
                                           [&](uint32_t j, float4 p, bool act) { return gather(F, j, p, act); } TC_PROF_PASS);
            overflow = stopped || cs + co + dcnt >= TC_NGBMAX;
            TC_STAGE_SWITCH(ST_EPILOGUE, ST_TEST);
            if (!overflow) {
                if (dcnt > 0) { pad_stage(0u, dcnt); convert_d(F); }
                TC_STAGE_SWITCH(ST_TEST, ST_CONVERT_D);
                if (do_wvt && wcnt > 0) convert_w(F, wcnt);
                TC_STAGE_SWITCH(ST_CONVERT_D, ST_TEST);
            }
            TC_STAGE_SWITCH(ST_TEST, ST_EPILOGUE);
        } else {
            const std::false_type F;
            d.ncand += stream_candidates_q(k, q, xi, yi, zi, idx, idxcap,
                                           [&](uint32_t j, float4 p, bool act) { return gather(F, j, p, act); });
            overflow = stopped || cs + co + dcnt >= TC_NGBMAX;
            if (!overflow) {
                if (dcnt > 0) { pad_stage((uint32_t)i, dcnt); convert_d(F); }
                if (do_wvt && wcnt > 0) convert_w(F, wcnt);
            }
        }
        wave_lds_fence();
        const int ca = cs + co;

        if (!overflow) {
            if (ord) {
                /* a list in index order exists only if the runs fed the gather, the reference's list would not have
                 * been cut (src/tree.c:91-92) and everything fitted */
                wvt_done = nxr != TC_XNONE && cw < TC_NGBMAX && cwout <= TC_XLCAP;
            } else if (do_wvt && cw < TC_NGBMAX) {
                wsum2(u0, u1, u0, u1); u2 = wsum(u2);
                wvt_done = true;
            }
            /* src/sph.c:36-64 on the two virtual queries */
            bool solved = false;
            if (TC_ABLATE(k)) {                                            /* profiling only: no solve */
                solved = true; d.rho = 1; wvt_done = true;
            } else {
                int cnt_use = -1;                                      /* one call site: the solver is large */
                if (cs >= TC_DESNNGB) {                                /* first query already has >= 295 */
                    d.nq += 1;
                    cnt_use = cs;
                } else {                                               /* hsml *= 1.23, second query */
                    d.nq += 2;
                    d.hsml = hb;
                    if (ca >= TC_DESNNGB) cnt_use = ca;
                    else d.hsml = (float)((double)hb * 1.23);          /* still too few: third query, plain path */
                }
                if (cnt_use >= 0) {
                    L.cs = cs;
                    TC_STAGE_SWITCH(ST_EPILOGUE, ST_SOLVE_UNIFORM);
                    solved = solve_hsml(L, cnt_use, k.mpart, da.bias_const, d.hsml, d.rho, d.dRhodHsml, d.nit, d.npair
                                        TC_PROF_PASS);
                    TC_STAGE_SWITCH(ST_SOLVE_UNIFORM, ST_EPILOGUE);
                }
            }
            if (solved) d.ok = true;
        }
        /* overflow: the first or second query of the reference would have filled its NGBMAX list;
         * replay it exactly from the carried hsml with the plain code */
    }

    if (finite && !d.ok && isfinite(d.hsml)) {
        const float rmax = k.margin_on ? tc_margin_radius(da.hsml_in[i], pi.w, k.boxsize, k.margin_widen) : HUGE_VALF;
        const tc_dev_const kv = particle_view(da.k, da.hsml_in[i], pi.w);
        density_loop(da, kv, i, xi, yi, zi, plain, idx, TC_ITER_IDXCAP, st, d, rmax);
    }
    if (finite) density_store<STATS>(da, i, d);

    if (ord) {
        if (lane == 0) {
            a.xlcnt[i] = wvt_done ? (uint32_t)cwout : TC_XNONE;
            if (!wvt_done) a.xun[atomicAdd(a.xun_cnt, 1)] = (uint32_t)i;       /* not listed: k_wvt_exact4 takes the particle */
        }
    } else if (do_wvt) {
        if (!wvt_done) {
            const tc_dev_const kv = particle_view(da.k, da.hsml_in[i], pi.w);
            wvt_sum(kv, i, pi, (double)pi.w, da.flags, idx, TC_ITER_IDXCAP, sw, u0, u1, u2);
        }
        if (lane == 0) {
            a.ustep[3 * (size_t)i] = u0;
            a.ustep[3 * (size_t)i + 1] = u1;
            a.ustep[3 * (size_t)i + 2] = u2;
        }
    }
    TC_STAGE_SWITCH(ST_EPILOGUE, ST_QUEUE);
}

#ifdef TC_PROFILE_STAGES
/* profiling build only: per-stage cycles of every wave of the last k_iter launch */
__device__ uint32_t g_stage_cycles[TC_NSTAGE * TC_MAX_PERSISTENT_BLOCKS * WPB];
extern "C" int tcgpu_debug_stage_cycles(uint32_t *out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stage_cycles), sizeof(uint32_t) * TC_NSTAGE * (size_t)nwaves) == hipSuccess ? 0 : -1;
}
#endif

template <bool STATS, int WVT>
__global__ __launch_bounds__(TBN, TC_ITER_MINWAVES) TC_ITER_ATTR void k_iter(tc_iter_args a)
{
    __shared__ __align__(16) unsigned char lds_raw[WPB * TC_LDS_PER_WAVE_ITER];
    const int wave = threadIdx.x >> 6;
    unsigned char *mine = lds_raw + (size_t)wave * TC_LDS_PER_WAVE_ITER;
    const int gw = blockIdx.x * WPB + wave;
    double *spill = a.d.spill + (size_t)gw * (2 * TC_NGBMAX);
#ifdef TC_PROFILE_ABLATE
    const uint64_t t_begin = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef TC_PROFILE_STAGES
    tc_prof prof, *prof__ = &prof;
    for (int s = 0; s < TC_NSTAGE; s++) prof.acc[s] = 0;
    prof.last = __builtin_amdgcn_s_memtime();
#endif
    work_queue(a.d.k, [&](int i) { iter_one<STATS, WVT>(a, i, mine, spill TC_PROF_PASS); });
#ifdef TC_PROFILE_STAGES
    TC_STAGE_SWITCH(ST_QUEUE, ST_QUEUE);
    if ((threadIdx.x & 63) == 0 && gw < TC_MAX_PERSISTENT_BLOCKS * WPB)
        for (int s = 0; s < TC_NSTAGE; s++) g_stage_cycles[TC_NSTAGE * gw + s] = prof.acc[s];
#endif
#ifdef TC_PROFILE_ABLATE
    if ((threadIdx.x & 63) == 0 && gw < TC_MAX_PERSISTENT_BLOCKS * WPB) {       /* profiling build: wave life span */
        g_wave_span[2 * gw] = t_begin;
        g_wave_span[2 * gw + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

/* Prepass of the ordered gather (WVT == 2): the index runs of the cells each particle's gather ball overlaps, in curve
 * order (ordered_runs_pf, one lane per particle), transposed into the particle's own list xr[i * TC_XRCAP ...] so that
 * the wavefront that later solves the particle reads them with one coalesced load.  Radius and level are the query
 * record's (k_prec): the ball k_iter gathers, R = max(1.23 hsml, hsml_wvt box). */
__global__ __launch_bounds__(TBN) void k_xruns(tc_dev_const k, const tc_prec *__restrict__ prec, const tc_pf pf,
                                               uint2 *__restrict__ scratch, uint2 *__restrict__ xr,
                                               uint32_t *__restrict__ xrn, int *__restrict__ flags)
{
    __shared__ uint32_t lds_stk[WPB * (TC_MAX_LEVEL + 1) * 64];
    __shared__ uint64_t lds_inv64[TC_HILBERT_NSTATES];
    __shared__ unsigned char lds_perm[TC_HILBERT_NSTATES * 256];
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES; t += TBN) {
        uint64_t r = 0;
        for (int kk = 0; kk < 8; kk++) r |= (uint64_t)TC_HILBERT_INV[t * 8 + kk] << (8 * kk);
        lds_inv64[t] = r;
    }
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES * 256; t += TBN) {
        const int stt = t >> 8, msk = t & 255;
        uint32_t cm = 0;
        for (int kk = 0; kk < 8; kk++) cm |= ((msk >> (TC_HILBERT_INV[stt * 8 + kk] & 7)) & 1u) << kk;
        lds_perm[t] = (unsigned char)cm;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t *stk = lds_stk + (size_t)wave * (TC_MAX_LEVEL + 1) * 64 + lane;
    uint2 *wruns = scratch + (size_t)(blockIdx.x * WPB + wave) * ((size_t)TC_XRUNCAP * 64);
    work_queue64(k, [&](int base, int stop) {
        const int t = base + lane;
        const bool valid = t < stop;
        const int tt = valid ? t : base;
        const int i = k.own ? (int)k.own[tt] : tt;
        const float4 pi = k.pos4[i];
        const tc_prec P = prec[i];
        const uint32_t pfl = (P.pack >> 4) & 0x7fu;
        const bool use = valid && (pfl & TC_PREC_VALID) && (pfl & TC_PREC_WARM);
        const int Lq = (int)(P.pack & 15u);
        /* (writing straight into the particles' own lists -- 8-byte stores at a stride of TC_XRCAP entries per lane -- was
         * slower than the coalesced copy below: 1.61 vs 1.46 ms) */
        int nr = ordered_runs_pf(k, pf, use, pi.x, pi.y, pi.z, P.R, Lq > 0 ? Lq : 1, lds_inv64, lds_perm, stk,
                                 wruns + lane, nullptr);
        if (valid) xrn[i] = (use && nr <= TC_XRCAP) ? (uint32_t)nr : TC_XNONE;
        /* the runs were written one lane per particle ([slot][lane]); each particle's list is now copied out by the
         * whole wave, coalesced */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int pl = 0; pl < 64 && base + pl < stop; pl++) {
            const int np = __shfl(nr, pl);
            const bool up = __shfl((int)use, pl) != 0;
            if (!up || np > TC_XRCAP) continue;
            const int ip = __shfl(i, pl);
            for (int q = lane; q < np; q += 64) xr[(size_t)ip * TC_XRCAP + q] = wruns[(size_t)q * 64 + pl];
        }
    });
}

/* the 64-byte query records (tc_prec / tc_cprec share the slots): one per LOCAL slot of the pass, with head room; a sharded
 * rank that has shrunk from a full-set pass to its steady-state local set gives the difference back */
int tc_ensure_prec(tcgpu_ctx *c)
{
    static_assert(sizeof(tc_prec) == 64, "64-byte record slots");
    size_t want = (size_t)c->nloc + (size_t)c->nloc / 8 + 1024;
    if (want > (size_t)c->cap) want = (size_t)c->cap;
    if (c->prec && c->prec_cap >= (size_t)c->nloc && !(c->nranks > 1 && c->prec_cap > 2 * want)) return 0;
    if (c->prec) { TC_HIP(c, hipStreamSynchronize(c->stream)); hipFree(c->prec); c->prec = nullptr; c->prec_cap = 0; }
    TC_HIP(c, hipMalloc(&c->prec, want * sizeof(tc_prec)));
    c->prec_cap = want;
    return 0;
}

/* run lists and neighbour lists of every local slot (own particles use theirs), sized by the local set of this pass (with
 * head room), not by the particle capacity: on a sharded rank the local set is a fraction of everything, and the lists are
 * what takes the memory (xlists_fit, api.hip).  Returns non-zero when the memory is not to be had: the caller then runs
 * the pass without lists (stand-alone sweep) -- a rank must not fail alone over a size only it knows. */
int tc_ensure_xlists(tcgpu_ctx *c)
{
    if (c->xr_cap >= (size_t)c->nloc) return 0;
    size_t cap = (size_t)c->nloc + (size_t)c->nloc / 8 + 1024;
    if (cap > (size_t)c->cap) cap = (size_t)c->cap;
    hipFree(c->xr); hipFree(c->xrn); hipFree(c->xlist); hipFree(c->xlcnt); hipFree(c->xun);
    c->xr = nullptr; c->xrn = nullptr; c->xlist = nullptr; c->xlcnt = nullptr; c->xun = nullptr; c->xr_cap = 0;
    bool ok = hipMalloc(&c->xr, cap * TC_XRCAP * sizeof(uint2)) == hipSuccess;
    ok = ok && hipMalloc(&c->xrn, cap * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&c->xlist, cap * TC_XLCAP * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&c->xlcnt, cap * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&c->xun, (cap + 4) * sizeof(uint32_t)) == hipSuccess;           /* the count lives behind the list */
    if (!ok) {
        (void)hipGetLastError();                                                         /* not an error of the pass */
        hipFree(c->xr); hipFree(c->xrn); hipFree(c->xlist); hipFree(c->xlcnt); hipFree(c->xun);
        c->xr = nullptr; c->xrn = nullptr; c->xlist = nullptr; c->xlcnt = nullptr; c->xun = nullptr;
        return -1;
    }
    c->xr_cap = cap;
    return 0;
}

/* with_wvt: 0 density only; 1 + round 2's f64 sweep sums (ustep); 2 + the sweep's neighbours listed in index order
 * (xlist / xlcnt, for k_wvt_chain4) with the gather fed from ordered runs */
int tc_launch_iter(tcgpu_ctx *c, int with_wvt)
{
    tc_iter_args a;
    tc_fill_const(c, &a.d.k);
    a.d.hsml_in = c->hsml;
    a.d.guess = c->guess;
    a.d.hsml_out = c->hsml;
    a.d.rho_out = c->rho;
    a.d.vhf_out = c->vhf;
    a.d.bias_const = -0.0116 * pow(TC_DESNNGB * 0.01, -2.236);   /* src/sph.c:206 */
    a.d.spill = c->spill;
    a.d.flags = c->flags;
    if (c->want_stats && !c->stats) TC_HIP(c, hipMalloc(&c->stats, 4 * (size_t)c->cap * sizeof(uint32_t)));      /* on first use */
    a.d.stats = c->want_stats ? c->stats : nullptr;
    a.d.stats_stride = (int)c->cap;
    if (with_wvt == 1 && !c->ustep) TC_HIP(c, hipMalloc(&c->ustep, 3 * (size_t)c->cap * sizeof(double)));          /* option sweep = 1 only */
    a.ustep = with_wvt == 1 ? c->ustep : nullptr;
    a.xr = nullptr; a.xrn = nullptr; a.xlist = nullptr; a.xlcnt = nullptr; a.xun = nullptr; a.xun_cnt = nullptr;
    int nloc = a.d.k.hi - a.d.k.lo;
    if (nloc <= 0) return 0;
    { int rp = tc_ensure_prec(c); if (rp) return rp; }
    a.prec = (const tc_prec *)c->prec;
    tc_phase_begin(c, PH_PREC);
    k_prec<<<(nloc + 255) / 256, 256, 0, c->stream>>>(a.d.k, a.d.hsml_in, with_wvt != 0, c->no_records, (tc_prec *)c->prec);
    if (with_wvt == 2) {
        /* run lists and neighbour lists of every local slot (own particles use theirs) */
        /* sized by the local set of this pass (with head room), not by the particle capacity: on a sharded rank the local
         * set is a fraction of everything, and the lists are what takes the memory (xlists_fit, api.hip) */
        if (tc_ensure_xlists(c)) TC_FAIL(c, TCGPU_ERR_NOMEM, "per-particle lists of the ordered gather: allocation failed");
        const int gx = xgrid(c, nloc, k_xruns);
        const size_t want = (size_t)gx * WPB * TC_XRUNCAP * 64 * sizeof(uint2);
        if (c->xruns_bytes < want) {
            if (c->xruns) TC_HIP(c, hipFree(c->xruns));
            c->xruns = nullptr; c->xruns_bytes = 0;
            TC_HIP(c, hipMalloc(&c->xruns, want));
            c->xruns_bytes = want;
        }
        TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 8 * 16 * sizeof(int), c->stream));
        k_xruns<<<gx, TBN, 0, c->stream>>>(a.d.k, a.prec, tc_pf{c->pf, c->pf_lmin, c->pf_lc}, (uint2 *)c->xruns, (uint2 *)c->xr, c->xrn, c->flags);
        a.xr = (const uint2 *)c->xr; a.xrn = c->xrn; a.xlist = c->xlist; a.xlcnt = c->xlcnt;
        a.xun = c->xun; a.xun_cnt = (int *)(c->xun + c->xr_cap);
        TC_HIP(c, hipMemsetAsync(a.xun_cnt, 0, sizeof(int), c->stream));
    }
    tc_phase_end(c);
    TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 8 * 16 * sizeof(int), c->stream));
    tc_phase_begin(c, PH_DENSITY);
#define TC_LAUNCH_ITER(S, W) k_iter<S, W><<<grid_for(c, nloc, k_iter<S, W>), TBN, 0, c->stream>>>(a)
    if (a.d.stats) { if (with_wvt == 2) TC_LAUNCH_ITER(true, 2); else if (with_wvt) TC_LAUNCH_ITER(true, 1); else TC_LAUNCH_ITER(true, 0); }
    else { if (with_wvt == 2) TC_LAUNCH_ITER(false, 2); else if (with_wvt) TC_LAUNCH_ITER(false, 1); else TC_LAUNCH_ITER(false, 0); }
#undef TC_LAUNCH_ITER
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K11 curl(A) */

struct tc_curl_args {
    tc_dev_const k;
    const float *hsml, *rho, *vhf;
    const float *apot;     /* 3n */
    float *bfld;           /* 3n */
    uint32_t *ovf_list;    /* particles whose ball holds >= NGBMAX neighbours: solved by k_curl_slow afterwards */
    int *ovf_count;
    int ovf_cap;
    int *flags;
    const struct tc_cprec *prec;   /* per-particle records of k_curl (k_cprec); shares the fused kernel's record buffer */
};

/* the wave-uniform prologue of curl_one, one lane per particle (see tc_prec): 64 bytes */
struct tc_cprec {
    double norm_h4, wfac;             /* WC6' normalisation (src/sph.c:439), -m / rho_i * varHsmlFac (src/sph.c:282) */
    float hq, hq2, fdy;               /* smoothing length, its square, 1 / hq of the f32 quotient (tc_fdiv) */
    float sf, hpf, inv_ny, inv_sf;    /* tc_query */
    int lo[3];
    uint32_t pack;                    /* qL : 4 | flags : 7 | nd[0..2] : 7 each; flag bit 0 = exact division here */
    uint32_t pad;
};
static_assert(sizeof(tc_cprec) == sizeof(tc_prec), "both records live in the same 64-byte slots");

/* src/sph.c:224-295 for particle i, pair by pair under the candidate predicate: the fall-back for a ball that
 * overflows NGBMAX (the reference then truncates its list, src/tree.c:91-92) */
__device__ __forceinline__ void curl_one_slow(const tc_curl_args &a, int i, uint32_t *idx)
{
    const uint32_t idxcap = TC_IDXCAP;
    const int lane = lane_id();
    const float4 pi = a.k.pos4[i];
    const float hq = a.hsml[i];
    const tc_dev_const k = particle_view(a.k, hq, 0.0f);
    const float hq2 = hq * hq;
    const double hsml = hq, rho_i = a.rho[i], vhf = a.vhf[i];
    const double ax = a.apot[3 * (size_t)i], ay = a.apot[3 * (size_t)i + 1], az = a.apot[3 * (size_t)i + 2];
    const double norm_h4 = TC_WC6_NORM / (double)(hq * hq * hq * hq) * -22.0;
    (void)norm_h4;                                           /* the cubic-spline build has its own normalisation */
    const double nm_rho = -k.mpart / rho_i;
    double b0 = 0, b1 = 0, b2 = 0;
    int cnt = 0;
    int thi = k.n;

    for (int pass = 0; pass < 2; pass++) {
        cnt = 0; b0 = b1 = b2 = 0;
        stream_candidates(k, pi.x, pi.y, pi.z, hq, idx, idxcap, [&](int j, float4 p, bool act) -> bool {
            float r2f = tc_ngb_r2(pi.x, pi.y, pi.z, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f);
            bool hit = act && (r2f < hq2) && j < thi;
            cnt += __popcll(tc_ballot(hit));
            if (hit && j != i) {
                double dx = (double)pi.x - (double)p.x, dy = (double)pi.y - (double)p.y, dz = (double)pi.z - (double)p.z;
                if (dx > k.boxhalf) dx -= k.boxsize;
                if (dx < -k.boxhalf) dx += k.boxsize;
                if (dy > k.boxhalf) dy -= k.boxsize;
                if (dy < -k.boxhalf) dy += k.boxsize;
                if (dz > k.boxhalf) dz -= k.boxsize;
                if (dz < -k.boxhalf) dz += k.boxsize;
                double r2 = dx * dx + dy * dy + dz * dz;
                if (!(r2 > hsml * hsml)) {
                    double r = sqrt(r2);
#ifdef TCGPU_SPH_CUBIC_SPLINE
                    double dwk = tc_dm4((float)r, hq);                      /* src/sph.c:374-378 */
#else
                    double dwk = tc_dwc6((float)r, hq, norm_h4);
#endif
                    double weight = nm_rho * dwk / r * vhf;
                    double dAx = ax - (double)a.apot[3 * (size_t)j];
                    double dAy = ay - (double)a.apot[3 * (size_t)j + 1];
                    double dAz = az - (double)a.apot[3 * (size_t)j + 2];
                    b0 += weight * (dz * dAy - dy * dAz);
                    b1 += weight * (dx * dAz - dz * dAx);
                    b2 += weight * (dy * dAx - dx * dAy);
                }
            }
            return false;
        });
        if (cnt < TC_NGBMAX || pass == 1) break;
        /* list truncation as in wvt_one: find the ascending-index threshold, then redo */
        int tlo = 0;
        thi = k.n;
        while (thi - tlo > 1) {
            int mid = tlo + ((thi - tlo) >> 1);
            int cm = 0;
            stream_candidates(k, pi.x, pi.y, pi.z, hq, idx, idxcap, [&](int j, float4 p, bool act) -> bool {
                float r2f = tc_ngb_r2(pi.x, pi.y, pi.z, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f);
                cm += __popcll(tc_ballot(act && (r2f < hq2) && j < mid));
                return false;
            });
            if (cm >= TC_NGBMAX) thi = mid; else tlo = mid;
        }
    }
    b0 = wsum(b0); b1 = wsum(b1); b2 = wsum(b2);
    if (lane == 0) {
        a.bfld[3 * (size_t)i] = (float)b0;
        a.bfld[3 * (size_t)i + 1] = (float)b1;
        a.bfld[3 * (size_t)i + 2] = (float)b2;
    }
}


/* src/sph.c:224-295 for particle i.  Same structure as the sweep of the fused kernel: candidates come as runs of
 * the row-major mirror where the ball is interior (stream_rows) or cell by cell, the f32 predicate of the
 * reference's ball query (src/tree.c:67-89) picks the hits, hits are compacted into an LDS ring of indices, and
 * the f64 pair term -- separation, W', Price (2010) eq. 79 -- is evaluated 64 hits at a time on full waves. */
/* one lane per particle: everything curl_one needs to know about its particle before it streams candidates */
__global__ __launch_bounds__(256) void k_cprec(tc_curl_args a, int no_records, tc_cprec *__restrict__ prec)
{
    const tc_dev_const &k0 = a.k;
    const int t = k0.lo + blockIdx.x * 256 + threadIdx.x;
    if (t >= k0.hi) return;
    const int i = k0.own ? (int)k0.own[t] : t;
    const float4 pv = k0.pos4[i];
    const float hq = a.hsml[i];
    const tc_dev_const k = particle_view<false>(k0, hq, 0.0f);   /* the curl's local set is marked from hsml alone (api.hip) */
    tc_cprec P;
    P.hq = hq;
    P.hq2 = hq * hq;
    P.norm_h4 = TC_WC6_NORM / (double)(hq * hq * hq * hq) * -22.0;
    P.wfac = -k.mpart / (double)a.rho[i] * (double)a.vhf[i];
    const tc_fdiv fd = tc_fdiv_setup(hq);
    P.fdy = fd.y;
    tc_query q;
    query_setup<false>(k, pv.x, pv.y, pv.z, hq, q);
    const int qL = q.L;
    const double ext = (double)hq * (1.0 + 1e-5) + k.boxsize * 1.2e-5 + query_cell_edge_at(k, qL);
    const bool wrap = !((double)pv.x >= ext && (double)pv.x <= k.boxsize - ext && (double)pv.y >= ext
                        && (double)pv.y <= k.boxsize - ext && (double)pv.z >= ext && (double)pv.z <= k.boxsize - ext);
    const bool fast = !wrap && k.mirror != nullptr && qL <= k.lmax_rm && qL >= k.lmin_rm;
    uint32_t fl = (fd.exact_div ? TC_PREC_WARM : 0u) | (wrap ? TC_PREC_WRAP : 0u) | (fast ? TC_PREC_FAST : 0u);
    bool fits = qL < 16 && isfinite(hq);
    uint32_t nds = 0;
    for (int d = 0; d < 3; d++) {
        if (q.full[d]) fl |= TC_PREC_FULL0 << d;
        else if (q.nd[d] > 127) fits = false;
        else nds |= (uint32_t)q.nd[d] << (7 * d);
        P.lo[d] = q.lo[d];
    }
    if (fits && !no_records) fl |= TC_PREC_VALID;
    P.sf = q.sf; P.hpf = q.hpf; P.inv_ny = q.inv_ny; P.inv_sf = q.inv_sf;
    P.pack = (uint32_t)qL | (fl << 4) | (nds << 11);
    P.pad = 0;
    prec[i] = P;
}

template <bool AW>     /* AW: the three components of A are equal and ride in the w lane of the positions (magnetic_field.c:63-65) */
__device__ __forceinline__ void curl_one(const tc_curl_args &a, int i, uint32_t *idx, uint32_t *ring, const tc_stage &st)
{
    const int lane = lane_id();
    const float4 pv = a.k.pos4[i];
    const float xi = U(pv.x), yi = U(pv.y), zi = U(pv.z);
    const tc_dev_const &k = a.k;
    const tc_cprec P = a.prec[i];                             /* scalar loads (k_cprec) */
    const uint32_t pk = U(P.pack);
    const uint32_t pflags = (pk >> 4) & 0x7fu;
    if (!(pflags & TC_PREC_VALID)) {                          /* a query the record cannot describe: the literal path */
        if (lane == 0) {
            const int slot = atomicAdd(a.ovf_count, 1);
            if (slot < a.ovf_cap) a.ovf_list[slot] = (uint32_t)i;
            else atomicOr(&a.flags[3], 1);
        }
        return;
    }
    const float hq = U(P.hq), hq2 = U(P.hq2);
    const double hsml = hq;
    const double ax = a.apot[3 * (size_t)i], ay = a.apot[3 * (size_t)i + 1], az = a.apot[3 * (size_t)i + 2];
    const double norm_h4 = P.norm_h4;
    (void)norm_h4;                                           /* the cubic-spline build has its own normalisation */
    const double wfac = P.wfac;                              /* -m/rho_i * varHsmlFac (src/sph.c:282), dwk/r per pair */
    tc_fdiv fd;
    fd.b = hq; fd.y = U(P.fdy); fd.exact_div = (pflags & TC_PREC_WARM) != 0;
    const bool wrap = (pflags & TC_PREC_WRAP) != 0;
    const bool fast = (pflags & TC_PREC_FAST) != 0;
    tc_query q;
    {
        q.L = (int)(pk & 15u);
        q.nL = 1 << q.L;
        const tc_level_desc D = k.lvl[q.L];
        q.off = U(D.off);
        q.ox = U(D.ox); q.oy = U(D.oy); q.oz = U(D.oz); q.ny = U(D.ny); q.nz = U(D.nz);
#pragma unroll
        for (int dd = 0; dd < 3; dd++) {
            q.full[dd] = (pflags & (TC_PREC_FULL0 << dd)) != 0;
            q.nd[dd] = q.full[dd] ? q.nL : (int)((pk >> (11 + 7 * dd)) & 127u);
            q.lo[dd] = U(P.lo[dd]);
        }
        q.sf = U(P.sf); q.hpf = U(P.hpf); q.inv_ny = U(P.inv_ny); q.inv_sf = U(P.inv_sf);
    }
    const tc_gpos vmirror = vgpr_pos(k.mirror);

    double b0 = 0, b1 = 0, b2 = 0;
    int cnt = 0, scnt = 0, head = 0;
    auto convert = [&](auto ftag, int nvalid) {
        constexpr bool F = decltype(ftag)::value;
        wave_lds_fence();
        const int sl = (head + lane) & (TC_STAGE - 1);
        const bool valid = lane < nvalid;
        float4 pj;
        uint32_t jp = 0;
        if (AW) {                                                          /* staged positions, A_j in the w lane: no loads here */
            pj = make_float4(st.x[sl], st.y[sl], st.z[sl], st.w[sl]);
        } else {
            const uint32_t j = valid ? ring[sl] : (F ? 0u : (uint32_t)i);
            pj = F ? ld4(vmirror, j) : k.pos4[j];
            jp = F ? k.mirror_idx[valid ? j : 0u] : j;                     /* particle index: A lives in particle order */
        }
        if (valid) {
            double dx = (double)xi - (double)pj.x, dy = (double)yi - (double)pj.y, dz = (double)zi - (double)pj.z;
            if (!F && wrap) {
                if (dx > k.boxhalf) dx -= k.boxsize;
                if (dx < -k.boxhalf) dx += k.boxsize;
                if (dy > k.boxhalf) dy -= k.boxsize;
                if (dy < -k.boxhalf) dy += k.boxsize;
                if (dz > k.boxhalf) dz -= k.boxsize;
                if (dz < -k.boxhalf) dz += k.boxsize;
            }
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (!(r2 > hsml * hsml)) {          /* src/sph.c:271-272 (a coincident particle gives 0/0 = NaN there and here) */
                double r, rinv;
                if (tc_ballot(r2 == 0)) {       /* wave-uniform: the IEEE sequences only where somebody divides by zero */
                    r = tc_sqrt_f64_lean(r2);
                    rinv = tc_rcp_f64_lean(r);
                } else {
                    r = tc_sqrt_f64_lean_pos(r2);
                    rinv = tc_rcp_f64_lean_nz(r);
                }
                /* sph_kernel_derivative_WC6 (src/sph.c:434-440): u = r/h in f32, t = (double)(1-u), trailing polynomial in f32 */
#ifdef TCGPU_SPH_CUBIC_SPLINE
                const double dwk = (double)tc_dm4((float)r, hq);           /* src/sph.c:374-378 */
#else
                const float u = tc_fdiv_apply(fd, (float)r);
                const double t = (double)(1 - u);
                const double t2 = t * t, t4 = t2 * t2, t7 = t4 * t2 * t;
                const float polyf = __builtin_fmaf(u, __builtin_fmaf(u, 16.0f, 7.0f), 1.0f);
                const double dwk = (double)(float)(norm_h4 * t7 * (double)u * (double)polyf);
#endif
                const double weight = wfac * dwk * rinv;
                if (AW) {
                    const double wdA = weight * (ax - (double)pj.w);         /* dAx = dAy = dAz */
                    b0 = fma(wdA, dz - dy, b0);
                    b1 = fma(wdA, dx - dz, b1);
                    b2 = fma(wdA, dy - dx, b2);
                } else {
                    const double dAx = ax - (double)a.apot[3 * (size_t)jp];
                    const double dAy = ay - (double)a.apot[3 * (size_t)jp + 1];
                    const double dAz = az - (double)a.apot[3 * (size_t)jp + 2];
                    b0 = fma(weight, dz * dAy - dy * dAz, b0);
                    b1 = fma(weight, dx * dAz - dz * dAx, b1);
                    b2 = fma(weight, dy * dAx - dx * dAy, b2);
                }
            }
        }
        head = U((head + 64) & (TC_STAGE - 1));
        wave_lds_fence();
    };
    auto gather = [&](auto ftag, uint32_t j, float4 p, bool act) -> bool {
        constexpr bool F = decltype(ftag)::value;
        const float r2 = ngb_r2_w(xi, yi, zi, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f, F ? false : wrap);
        const bool hit = act && (r2 < hq2);
        uint64_t m = tc_ballot(hit);
        cnt = U(cnt + (int)__popcll(m));
        bool use = hit;
        if (tc_ballot(r2 == 0.0f)) {                           /* the particle itself is not a neighbour (src/sph.c:243-244) */
            const bool self = hit && r2 == 0.0f && (F ? k.mirror_idx[j] == (uint32_t)i : j == (uint32_t)i);
            use = hit && !self;
            m = tc_ballot(use);
        }
        if (use) {
            const int sl = (head + scnt + mask_rank(m)) & (TC_STAGE - 1);
            if (AW) { st.x[sl] = p.x; st.y[sl] = p.y; st.z[sl] = p.z; st.w[sl] = p.w; }
            else ring[sl] = j;
        }
        scnt = U(scnt + (int)__popcll(m));
        if (scnt >= 64) { convert(ftag, 64); scnt = U(scnt - 64); }
        return cnt >= TC_NGBMAX;
    };
    if (fast) {
        const std::true_type F;
        stream_rows_q<false>(k, q, xi, yi, zi, idx, [&](uint32_t j, float4 p, bool act) { return gather(F, j, p, act); });
        if (cnt < TC_NGBMAX && scnt > 0) convert(F, scnt);
    } else {
        const std::false_type F;
        stream_candidates_q(k, q, xi, yi, zi, idx, TC_IDXCAP, [&](uint32_t j, float4 p, bool act) { return gather(F, j, p, act); });
        if (cnt < TC_NGBMAX && scnt > 0) convert(F, scnt);
    }
    if (cnt >= TC_NGBMAX) {                                    /* list truncation semantics: left to k_curl_slow */
        if (lane == 0) {
            const int slot = atomicAdd(a.ovf_count, 1);
            if (slot < a.ovf_cap) a.ovf_list[slot] = (uint32_t)i;
            else atomicOr(&a.flags[3], 1);
        }
        return;
    }
    b0 = wsum(b0); b1 = wsum(b1); b2 = wsum(b2);
    if (lane == 0) {
        a.bfld[3 * (size_t)i] = (float)b0;
        a.bfld[3 * (size_t)i + 1] = (float)b1;
        a.bfld[3 * (size_t)i + 2] = (float)b2;
    }
}

template <bool AW>
__global__ __launch_bounds__(TBN, 5) void k_curl(tc_curl_args a)
{
    __shared__ __align__(16) uint32_t lds_idx[WPB * TC_IDXCAP];
    __shared__ __align__(16) float lds_ring[WPB * 4 * TC_STAGE];
    const int wave = threadIdx.x >> 6;
    uint32_t *idx = lds_idx + (size_t)wave * TC_IDXCAP;
    tc_stage st;
    st.x = lds_ring + (size_t)wave * 4 * TC_STAGE;
    st.y = st.x + TC_STAGE; st.z = st.y + TC_STAGE; st.w = st.z + TC_STAGE;
    uint32_t *ring = reinterpret_cast<uint32_t *>(st.x);
    work_queue(a.k, [&](int i) { curl_one<AW>(a, i, idx, ring, st); });
}

/* the literal path for the particles k_curl set aside (normally none); the list length is read on the device */
__global__ __launch_bounds__(TBN) void k_curl_slow(tc_curl_args a)
{
    __shared__ __align__(16) uint32_t lds_idx[WPB * TC_IDXCAP];
    const int wave = threadIdx.x >> 6;
    uint32_t *idx = lds_idx + (size_t)wave * TC_IDXCAP;
    int n = *a.ovf_count;
    if (n > a.ovf_cap) n = a.ovf_cap;
    for (int t = blockIdx.x * WPB + wave; t < n; t += gridDim.x * WPB) curl_one_slow(a, (int)a.ovf_list[t], idx);
}

int tc_launch_curl(tcgpu_ctx *c, float *l_bfld, int a_in_w)
{
    tc_curl_args a;
    tc_fill_const(c, &a.k);
    a.hsml = c->hsml;
    a.rho = c->rho;
    a.vhf = c->vhf;
    a.apot = c->l_apot;
    a.bfld = l_bfld;
    a.ovf_list = reinterpret_cast<uint32_t *>(c->ngb_buf);
    a.ovf_count = c->d_count + 3;
    a.ovf_cap = (int)c->cap;
    a.flags = c->flags;
    int nloc = a.k.hi - a.k.lo;
    if (nloc <= 0) return 0;
    { int rp = tc_ensure_prec(c); if (rp) return rp; }
    a.prec = (const tc_cprec *)c->prec;
    tc_phase_begin(c, PH_CURL);
    TC_HIP(c, hipMemsetAsync(c->work_ctr, 0, 8 * 16 * sizeof(int), c->stream));
    TC_HIP(c, hipMemsetAsync(c->d_count + 3, 0, sizeof(int), c->stream));
    if (!c->curl_literal) k_cprec<<<(nloc + 255) / 256, 256, 0, c->stream>>>(a, c->no_records, (tc_cprec *)c->prec);
    if (c->curl_literal) {                       /* option "curl_literal" (tests): every particle through the literal path */
        const int cnt = nloc;
        if (c->nranks > 1) TC_HIP(c, hipMemcpyAsync(a.ovf_list, c->own_list, (size_t)nloc * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        else if (tc_launch_iota(c, a.ovf_list, (size_t)nloc, 0)) return TCGPU_ERR_HIP;
        TC_HIP(c, hipMemcpyAsync(c->d_count + 3, &cnt, sizeof(int), hipMemcpyHostToDevice, c->stream));
    } else if (a_in_w) k_curl<true><<<grid_for(c, nloc, k_curl<true>), TBN, 0, c->stream>>>(a);
    else k_curl<false><<<grid_for(c, nloc, k_curl<false>), TBN, 0, c->stream>>>(a);
    k_curl_slow<<<256, TBN, 0, c->stream>>>(a);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ single ball query (API / tests) */

__global__ __launch_bounds__(64) void k_find_ngb(tc_dev_const k, int i, float hsml, int32_t *out, int *count)
{
    __shared__ __align__(16) uint32_t idx[TC_IDXCAP];
    const uint32_t idxcap = TC_IDXCAP;
    const float4 pi = k.pos4[i];
    const float h2 = hsml * hsml;
    int cnt = 0;
    stream_candidates(k, pi.x, pi.y, pi.z, hsml, idx, idxcap, [&](int j, float4 p, bool act) -> bool {
        float r2 = tc_ngb_r2(pi.x, pi.y, pi.z, p.x, p.y, p.z, k.boxhalf_f, k.boxsize_f);
        bool hit = act && (r2 < h2);
        uint64_t m = tc_ballot(hit);
        if (hit) out[cnt + mask_rank(m)] = j;
        cnt += __popcll(m);
        return false;
    });
    if (lane_id() == 0) *count = cnt;
}

int tc_launch_find_ngb(tcgpu_ctx *c, int ipart, float hsml)
{
    tc_dev_const k;
    tc_fill_const(c, &k);
    k_find_ngb<<<1, 64, 0, c->stream>>>(k, ipart, hsml, c->ngb_buf, c->ngb_cnt);
    TC_HIP(c, hipGetLastError());
    return 0;
}
