/*
 * kernels_stream.hip -- the streaming (HBM-bound) kernels of the path:
 *   K1  Peano keys            src/peano.c:63-71
 *   K3  permutation gather    src/peano.c:85-126
 *   K4' neighbour index       replaces the serial octree build of src/tree.c:124-271 by
 *                             dense per-level cell tables over the Peano-sorted particles
 *   K4g first-pass hsml guess src/tree.c:113-121 evaluated without materialising the tree
 *   K6  density error sums    src/wvt_relax.c:73-85
 *   K7/K8 model hsml          src/wvt_relax.c:108-124
 *   K10 move + wrap           src/wvt_relax.c:177-214
 * One thread per particle, 16-byte coalesced accesses on float4 positions.
 */
#include <cstring>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include <iterator>
#include "tc_ctx.h"
#include "tc_hilbert_lut.h"

#define TB 256

__device__ const unsigned short TC_HILBERT_LUT_DEV[TC_HILBERT_NSTATES * 64] = {
#define TC_HILBERT_LUT_ROWS_ONLY
#include "tc_hilbert_lut.h"
#undef TC_HILBERT_LUT_ROWS_ONLY
};

/* ------------------------------------------------------------------ interest mask + local set

 * Sharded contexts hold every POSITION (all-gathered after each move) but build their search structures only
 * over the particles their own queries can reach: the own index range plus a ghost shell.  The shell comes from an
 * "interest pyramid": one bit per octree cell of levels 1..lp_max (dense (x, y, z) layout of tc_level_offset).
 * Every own particle marks the <= 3^3 cells its ball of radius tc_margin_radius() overlaps at the level whose
 * cell edge is >= that radius; a particle belongs to the local set when the cell holding it is marked at any
 * level.  Both sides pad like the cell-table query does (query_setup in kernels_ngb.hip), so every particle a
 * permitted query can accept is in the set; queries beyond the margin radius are refused by the solver kernels
 * and the pass is repeated (api.hip). */
__device__ __forceinline__ size_t tc_sum_bit(size_t x, size_t y, size_t z)      /* bit of a level-TC_LS cell */
{
    return (((x << TC_LS) + y) << TC_LS) + z;
}

/* Marks are global atomicOr's on a few thousand words that tens of thousands of waves want to set.  A plain
 * "test first" load does not help: a CU's L1 is never refreshed by other CUs' atomics, so every wave keeps seeing the
 * stale zero and fires again, and same-address atomics serialise (measured: 1 ms for 2e6 particles).  Each block
 * therefore walks a CONTIGUOUS chunk of the Peano-ordered own range -- its waves mark nearly the same cells -- and
 * remembers in LDS which bits of which word it has already set: one 64-bit entry {word index, bits known set}. */
#define TC_MARK_CACHE 2048
__device__ __forceinline__ void tc_mark_bit(uint32_t *__restrict__ bits, unsigned long long *cache, uint32_t tag, size_t bit)
{
    const uint32_t w = (uint32_t)(bit >> 5) | tag;             /* tag separates pyramid and summary words */
    const uint32_t m = 1u << (bit & 31);
    unsigned long long *e = &cache[(w * 2654435761u) >> (32 - 11)];
    const unsigned long long v = *e;
    if ((uint32_t)(v >> 32) == w && ((uint32_t)v & m)) return;
    atomicOr(&bits[bit >> 5], m);
    *e = (uint32_t)(v >> 32) == w ? (v | m) : (((unsigned long long)w << 32) | m);   /* one 64-bit LDS store: never torn */
}

/* All 64 lanes mark the cells [c0, c0 + n) (unwrapped coordinates, periodic) of level L -- in the pyramid, or in the
 * coarse summary when `bits` is the summary and off = 0.  Arguments are wave-uniform. */
__device__ __forceinline__ void tc_mark_box(uint32_t *__restrict__ bits, unsigned long long *cache, uint32_t tag, int L,
                                            size_t off, const int c0[3], const int n[3])
{
    const int nL = 1 << L;
    const int ncell = n[0] * n[1] * n[2];
    for (int t = threadIdx.x & 63; t < ncell; t += 64) {
        const int iz = t % n[2], iy = (t / n[2]) % n[1], ix = t / (n[2] * n[1]);
        const size_t x = (size_t)((c0[0] + ix) & (nL - 1)), y = (size_t)((c0[1] + iy) & (nL - 1)),
                     z = (size_t)((c0[2] + iz) & (nL - 1));
        tc_mark_bit(bits, cache, tag, off + ((((x << L) + y) << L) + z));
    }
}

/* One wavefront per 64 consecutive own particles; a block works through a contiguous chunk of the own range.
 *  - Coarse balls (pyramid level <= TC_LCOARSE: the few particles of the outskirts, whose cells are a sizeable part
 *    of the box, so that bounding boxes would claim far too much): every lane sets the <= 125 cells of its own ball in
 *    a block-wide LDS copy of the coarse pyramid levels (ds_or); the block flushes the non-zero words once.
 *  - Fine balls (where nearly all particles are): the 64 particles are neighbours along the Peano curve and their
 *    balls overlap almost completely -- the wave marks the cells of the common bounding box of the balls once, with
 *    the largest radius: about one mark per particle instead of up to 125.  A wave of mixed fine levels or far-flung
 *    particles goes ball by ball, the 64 lanes sharing the cells of one ball.
 *  - The level-TC_LS summary (a fast-reject filter: over-inclusion is harmless) gets the bounding box of all the
 *    wave's balls, in LDS, flushed with the coarse levels.
 * Per table level the block also accumulates (LDS) the bounding box of the cells its queries can touch and the range
 * of levels; one set of global atomics per block at the end. */
#define TC_LCOARSE 5
#define TC_COARSE_WORDS ((8 + 64 + 512 + 4096 + 32768) / 32 + 1)       /* cells of levels 1..5, tc_level_offset layout */
#define TC_SUM_WORDS ((1 << (3 * TC_LS)) / 32)
__global__ __launch_bounds__(TB) void k_mark_interest(const float4 *__restrict__ gpos4, const float *__restrict__ ghsml,
                                                      int lo, int hi, int chunk, double box, double box_mant, int box_exp,
                                                      double level_scale, int level_shift, int lmax, int lp_max, int widen,
                                                      int ignore_w, uint32_t *__restrict__ imask, uint32_t *__restrict__ isum,
                                                      int *__restrict__ lvl_range, int *__restrict__ bbox)
{
    __shared__ unsigned long long cache[TC_MARK_CACHE];
    __shared__ uint32_t s_coarse[TC_COARSE_WORDS];
    __shared__ uint32_t s_sum[TC_SUM_WORDS];
    __shared__ int s_bb[6 * (TC_MAX_LEVEL + 1)];
    __shared__ int s_lv[2];
    for (int t = threadIdx.x; t < TC_MARK_CACHE; t += TB) cache[t] = ~0ull;
    for (int t = threadIdx.x; t < TC_COARSE_WORDS; t += TB) s_coarse[t] = 0;
    for (int t = threadIdx.x; t < TC_SUM_WORDS; t += TB) s_sum[t] = 0;
    for (int t = threadIdx.x; t < 6 * (TC_MAX_LEVEL + 1); t += TB) s_bb[t] = (t % 6) < 3 ? (1 << 30) : -1;
    if (threadIdx.x == 0) { s_lv[0] = TC_MAX_LEVEL + 1; s_lv[1] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int cbeg = lo + blockIdx.x * chunk, cend = min(hi, cbeg + chunk);

    /* marking level of a radius: cell edge s with rp/2 <= s < rp, i.e. a ball spans <= 5 cells per dimension
     * (clamped to the pyramid); rp is the radius padded like the cell-table query (query_setup, kernels_ngb.hip) */
    auto level_of = [&](double rp) {
        int L = 1;
        if (rp < box) {
            const int er = __builtin_amdgcn_frexp_exp(rp) - 1;
            const double mr = 2 * __builtin_amdgcn_frexp_mant(rp);
            L = box_exp - er - (box_mant < mr ? 1 : 0) + 1;      /* floor(log2(box / rp)) + 1 */
        }
        if (L < 1) L = 1;
        if (L > lp_max) L = lp_max;
        return L;
    };
    /* cells of level L overlapped by the ball (q, rp): first cell and count per dimension (whole ring if it wraps round) */
    auto ball_cells = [&](float qx, float qy, float qz, double rp, int L, int c0[3], int n[3]) {
        const int nL = 1 << L;
        const double inv_s = (double)nL / box;
        const float xs[3] = {qx, qy, qz};
        for (int d = 0; d < 3; d++) {
            const int a = (int)floor(((double)xs[d] - rp) * inv_s), b = (int)floor(((double)xs[d] + rp) * inv_s);
            c0[d] = a; n[d] = b - a + 1;
            if (n[d] >= nL) { n[d] = nL; c0[d] = 0; }
        }
    };

    for (int base = cbeg + (threadIdx.x & ~63); base < cend; base += TB) {      /* wave-uniform: every wave its 64 particles */
        const int g = base + lane;
        const bool act = g < cend;
        float4 p = make_float4(0, 0, 0, 0);
        float h0 = 0, rg = 0;
        int la = TC_MAX_LEVEL + 1, lb = 0;
        if (act) {
            p = gpos4[g];
            h0 = ghsml[g];
            rg = tc_margin_radius(h0, ignore_w ? 0.0f : p.w, box, widen);
            /* table levels the own queries of this pass can ask for (tc_particle_levels) */
            tc_particle_levels(box, box_mant, box_exp, level_scale, level_shift, lmax, h0, rg, &la, &lb);
        }
        const int la_own = la, lb_own = lb;
        for (int o = 32; o > 0; o >>= 1) {
            la = min(la, __shfl_xor(la, o));
            lb = max(lb, __shfl_xor(lb, o));
        }
        if (la > TC_MAX_LEVEL) continue;                        /* a wave without own particles */
        if (lane == 0) { atomicMin(&s_lv[0], la); atomicMax(&s_lv[1], lb); }
        /* per table level: bounding box of the cells the wave's queries can touch (the query's own padding,
         * query_setup); a range that leaves [0, 2^L) wraps round the box: the whole ring in that dimension */
        const double rp_own = (double)rg * (1.0 + 1e-5) + box * 2e-6;
        {
            const float xs3[3] = {p.x, p.y, p.z};
            for (int L = la; L <= lb; L++) {
                const int nL = 1 << L;
                const double inv_s = (double)nL / box;
                const bool in = act && L >= la_own && L <= lb_own;
                for (int d = 0; d < 3; d++) {
                    int a = in ? (int)floor(((double)xs3[d] - rp_own) * inv_s) : (1 << 30);
                    int b = in ? (int)floor(((double)xs3[d] + rp_own) * inv_s) : -(1 << 30);
                    if (in && (a < 0 || b >= nL)) { a = 0; b = nL - 1; }
                    for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); b = max(b, __shfl_xor(b, o)); }
                    if (lane == 0 && a <= b) { atomicMin(&s_bb[6 * L + d], a); atomicMax(&s_bb[6 * L + 3 + d], b); }
                }
            }
        }
        const int L_own = act ? level_of(rp_own) : 0;
        /* summary: the bounding box of all the wave's balls at level TC_LS, in LDS */
        {
            const int nS = 1 << TC_LS;
            const double inv_s = (double)nS / box;
            const float xs3[3] = {p.x, p.y, p.z};
            int s0[3], sn[3];
            for (int d = 0; d < 3; d++) {
                int a = act ? (int)floor(((double)xs3[d] - rp_own) * inv_s) : (1 << 30);
                int b = act ? (int)floor(((double)xs3[d] + rp_own) * inv_s) : -(1 << 30);
                for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); b = max(b, __shfl_xor(b, o)); }
                s0[d] = a; sn[d] = b - a + 1;
                if (sn[d] >= nS) { sn[d] = nS; s0[d] = 0; }
            }
            const int nc = sn[0] * sn[1] * sn[2];
            for (int t = lane; t < nc; t += 64) {
                const int iz = t % sn[2], iy = (t / sn[2]) % sn[1], ix = t / (sn[2] * sn[1]);
                const size_t sb = tc_sum_bit((size_t)((s0[0] + ix) & (nS - 1)), (size_t)((s0[1] + iy) & (nS - 1)),
                                             (size_t)((s0[2] + iz) & (nS - 1)));
                atomicOr(&s_sum[sb >> 5], 1u << (sb & 31));
            }
        }
        /* coarse balls: lane by lane into the block's LDS copy of the coarse levels */
        const bool coarse = act && L_own <= TC_LCOARSE;
        if (coarse) {
            int c0[3], n[3];
            ball_cells(p.x, p.y, p.z, rp_own, L_own, c0, n);
            const int nL = 1 << L_own;
            const size_t off = tc_level_offset(L_own);
            for (int ix = 0; ix < n[0]; ix++)
                for (int iy = 0; iy < n[1]; iy++)
                    for (int iz = 0; iz < n[2]; iz++) {
                        const size_t x = (size_t)((c0[0] + ix) & (nL - 1)), y = (size_t)((c0[1] + iy) & (nL - 1)),
                                     z = (size_t)((c0[2] + iz) & (nL - 1));
                        const size_t bit = off + ((((x << L_own) + y) << L_own) + z);
                        atomicOr(&s_coarse[bit >> 5], 1u << (bit & 31));
                    }
        }
        /* fine balls */
        const bool fine = act && !coarse;
        float rmax = fine ? rg : 0.0f;
        int Lmin = fine ? L_own : 99, Lmax = fine ? L_own : 0;
        for (int o = 32; o > 0; o >>= 1) {
            rmax = fmaxf(rmax, __shfl_xor(rmax, o));
            Lmin = min(Lmin, __shfl_xor(Lmin, o));
            Lmax = max(Lmax, __shfl_xor(Lmax, o));
        }
        if (Lmax == 0) continue;                                 /* no fine ball in this wave */
        bool done = false;
        if (Lmax - Lmin <= 1) {
            const double rp = (double)rmax * (1.0 + 1e-5) + box * 2e-6;
            const int L = level_of(rp), nL = 1 << L;
            const double inv_s = (double)nL / box;
            const float xs[3] = {p.x, p.y, p.z};
            int w0[3], n[3];
            bool compact = true;
            for (int d = 0; d < 3; d++) {
                int a = fine ? (int)floor(((double)xs[d] - rp) * inv_s) : (1 << 30);
                int b = fine ? (int)floor(((double)xs[d] + rp) * inv_s) : -(1 << 30);
                for (int o = 32; o > 0; o >>= 1) { a = min(a, __shfl_xor(a, o)); b = max(b, __shfl_xor(b, o)); }
                w0[d] = a; n[d] = b - a + 1;
                if (n[d] > 8) compact = false;               /* a ball spans <= 5 cells; a compact wave adds one or two */
            }
            if (compact) {
                tc_mark_box(imask, cache, 0u, L, tc_level_offset(L), w0, n);
                done = true;
            }
        }
        if (!done) {                                             /* ball by ball */
            uint64_t todo = __ballot(fine);
            while (todo) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const float qx = __shfl(p.x, l), qy = __shfl(p.y, l), qz = __shfl(p.z, l);
                const double rp = __shfl(rp_own, l);
                const int L = __shfl(L_own, l);
                int c0[3], n[3];
                ball_cells(qx, qy, qz, rp, L, c0, n);
                tc_mark_box(imask, cache, 0u, L, tc_level_offset(L), c0, n);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TC_COARSE_WORDS; t += TB)
        if (s_coarse[t]) atomicOr(&imask[t], s_coarse[t]);
    for (int t = threadIdx.x; t < TC_SUM_WORDS; t += TB)
        if (s_sum[t]) atomicOr(&isum[t], s_sum[t]);
    for (int t = threadIdx.x; t < 6 * (TC_MAX_LEVEL + 1); t += TB) {
        const int v = s_bb[t];
        if ((t % 6) < 3) { if (v < (1 << 30)) atomicMin(&bbox[t], v); }
        else if (v >= 0) atomicMax(&bbox[t], v);
    }
    if (threadIdx.x == 0 && s_lv[0] <= TC_MAX_LEVEL) { atomicMin(&lvl_range[0], s_lv[0]); atomicMax(&lvl_range[1], s_lv[1]); }
}

/* is particle p inside the interest mask?  (orphans -- a coordinate == boxsize, see k_cells -- always are: every
 * query that touches the periodic boundary tests them by brute force) */
struct tc_in_mask {
    const float4 *gpos4;
    const uint32_t *imask, *isum;
    double box;
    int lp_max;
    __device__ bool operator()(uint32_t g) const
    {
        const float4 p = gpos4[g];
        uint64_t X[3];
        tc_scaled_coords(p.x, p.y, p.z, box, X);
        if (((X[0] | X[1] | X[2]) >> 63) != 0) return true;
        {                                                     /* fast reject: nothing marked in or above this coarse cell */
            const int sh = 63 - TC_LS;
            const size_t sb = tc_sum_bit((size_t)(X[2] >> sh), (size_t)(X[0] >> sh), (size_t)(X[1] >> sh));
            if (!(isum[sb >> 5] & (1u << (sb & 31)))) return false;
        }
        for (int L = lp_max; L >= 1; L--) {
            const int sh = 63 - L;
            const size_t nL = (size_t)1 << L;
            const size_t bit = tc_level_offset(L) + (((size_t)(X[2] >> sh) * nL) + (size_t)(X[0] >> sh)) * nL + (size_t)(X[1] >> sh);
            if (imask[bit >> 5] & (1u << (bit & 31))) return true;
        }
        return false;
    }
};

struct tc_in_range {
    const uint32_t *lg;
    uint32_t lo, hi;
    __device__ bool operator()(uint32_t i) const { const uint32_t g = lg[i]; return g >= lo && g < hi; }
};

int tc_select_temp_bytes(size_t n, size_t *bytes)
{
    size_t b = 0;
    auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), tc_in_range{nullptr, 0, 0});
    hipError_t e = rocprim::select(nullptr, b, rocprim::counting_iterator<uint32_t>(0), flags, (uint32_t *)nullptr,
                                   (int *)nullptr, n);
    size_t b2 = 0;
    if (e == hipSuccess)
        e = rocprim::select(nullptr, b2, rocprim::counting_iterator<uint32_t>(0), (unsigned char *)nullptr, (uint32_t *)nullptr,
                            (int *)nullptr, n);
    if (b2 > b) b = b2;
    size_t b3 = 0;                                  /* the ghost exchange's scan over [dest][block] counts */
    if (e == hipSuccess)
        e = rocprim::inclusive_scan(nullptr, b3, (int *)nullptr, (int *)nullptr, (size_t)TC_GHOST_MAXR * (n / TB + 2),
                                    rocprim::plus<int>());
    if (b3 > b) b = b3;
    *bytes = b;
    return e == hipSuccess ? 0 : -1;
}

int tc_launch_mark_interest(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    const size_t nbits = tc_level_offset(c->lp_max + 1);
    int e2 = 0;
    const double mant = 2 * frexp(c->par.boxsize, &e2);
    const int init[8] = {TC_MAX_LEVEL + 1, 0, 0, 0, 0, 0, 0, 0};
    int binit[6 * (TC_MAX_LEVEL + 1)];
    for (int L = 0; L <= TC_MAX_LEVEL; L++)
        for (int d = 0; d < 3; d++) { binit[6 * L + d] = 1 << 30; binit[6 * L + 3 + d] = -1; }
    TC_HIP(c, hipMemcpyAsync(c->d_bbox, binit, sizeof(binit), hipMemcpyHostToDevice, c->stream));
    TC_HIP(c, hipMemsetAsync(c->imask, 0, (nbits / 32 + 1) * sizeof(uint32_t), c->stream));
    TC_HIP(c, hipMemsetAsync(c->isum, 0, ((size_t)1 << (3 * TC_LS)) / 8, c->stream));
    TC_HIP(c, hipMemcpyAsync(c->lvl_range, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    if (hi > lo) {
        /* contiguous chunks of the own range, a multiple of the block size, 4 blocks per CU (measured best of 2 / 4 / 8 / 16: the per-block LDS set-up and flush against parallelism) */
        int64_t nblocks = (int64_t)c->num_cu * 4;
        int64_t chunk = ((hi - lo + nblocks - 1) / nblocks + TB - 1) / TB * TB;
        nblocks = (hi - lo + chunk - 1) / chunk;
        k_mark_interest<<<(unsigned)nblocks, TB, 0, c->stream>>>(
            c->g_pos4[c->gcur], c->g_hsml[c->gcur], (int)lo, (int)hi, (int)chunk, c->par.boxsize, mant, e2 - 1, tc_level_scale(c),
            c->level_shift, c->lmax, c->lp_max, c->margin_widen, c->mark_ignore_w, c->imask, c->isum, c->lvl_range,
            c->d_bbox);
    }
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* one byte per particle: inside the interest mask?  (a plain streaming kernel: the test inside the library's
 * partition kernel ran at a third of this rate) */
__global__ __launch_bounds__(TB) void k_flag_interest(tc_in_mask pred, int n, unsigned char *__restrict__ flag)
{
    int g = blockIdx.x * TB + threadIdx.x;
    if (g < n) flag[g] = pred((uint32_t)g) ? 1 : 0;
}

/* lsel = ascending global indices of the particles inside the mask; *nloc their number (synchronises) */
int tc_select_local(tcgpu_ctx *c, int64_t *nloc)
{
    tc_in_mask pred{c->g_pos4[c->gcur], c->imask, c->isum, c->par.boxsize, c->lp_max};
    unsigned char *flag = reinterpret_cast<unsigned char *>(c->idx);          /* n bytes of scratch (the sort's index array) */
    k_flag_interest<<<(unsigned)((c->n + TB - 1) / TB), TB, 0, c->stream>>>(pred, (int)c->n, flag);
    size_t b = c->sel_tmp_bytes;
    hipError_t e = rocprim::select(c->sel_tmp, b, rocprim::counting_iterator<uint32_t>(0), flag, c->lsel, c->d_count,
                                   (size_t)c->n, c->stream);
    TC_HIP(c, e);
    int h[3] = {0, 0, 0};
    tc_phase_end(c);                                          /* PH_LOCAL, begun by the caller: mark + select */
    TC_HIP(c, hipMemcpyAsync(&h[0], c->d_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipMemcpyAsync(&h[1], c->lvl_range, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    *nloc = h[0];
    c->lmin_tab = h[1] < 1 ? 1 : (h[1] > c->lmax ? c->lmax : h[1]);
    int bb[6 * (TC_MAX_LEVEL + 1)];
    TC_HIP(c, hipMemcpy(bb, c->d_bbox, sizeof(bb), hipMemcpyDeviceToHost));
    return tc_layout_table(c, bb);
}

/* ------------------------------------------------------------------ ghost exchange (sharded contexts) */

/* tc_in_mask for coordinates already scaled (one particle is tested against several pyramids) */
__device__ __forceinline__ bool tc_in_pyramid(const uint64_t X[3], const uint32_t *__restrict__ imask,
                                              const uint32_t *__restrict__ isum, int lp_max)
{
    {
        const int sh = 63 - TC_LS;
        const size_t sb = tc_sum_bit((size_t)(X[2] >> sh), (size_t)(X[0] >> sh), (size_t)(X[1] >> sh));
        if (!(isum[sb >> 5] & (1u << (sb & 31)))) return false;
    }
    for (int L = lp_max; L >= 1; L--) {
        const int sh = 63 - L;
        const size_t nL = (size_t)1 << L;
        const size_t bit = tc_level_offset(L) + (((size_t)(X[2] >> sh) * nL) + (size_t)(X[0] >> sh)) * nL + (size_t)(X[1] >> sh);
        if (imask[bit >> 5] & (1u << (bit & 31))) return true;
    }
    return false;
}

/* Sender side of the ghost exchange: which other ranks' pyramids is each OWN particle inside?  (The same test, on
 * the same bits, that the receiver's flag pass ran over all positions before -- so the ghost sets are the same.)
 * One bit per destination rank in mask[]; per block and destination the number of set bits. */
__global__ __launch_bounds__(TB) void k_ghost_count(const float4 *__restrict__ gpos4, int lo, int hi,
                                                    const uint32_t *__restrict__ pyr, size_t chunk, size_t imask_words,
                                                    int rank, int R, double box, int lp_max, uint32_t *__restrict__ mask,
                                                    int *__restrict__ blk_cnt, int nblk)
{
    __shared__ int s_cnt[TC_GHOST_MAXR];
    if (threadIdx.x < TC_GHOST_MAXR) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int g = lo + blockIdx.x * TB + threadIdx.x;
    uint32_t m = 0;
    if (g < hi) {
        const float4 p = gpos4[g];
        uint64_t X[3];
        tc_scaled_coords(p.x, p.y, p.z, box, X);
        const bool orphan = ((X[0] | X[1] | X[2]) >> 63) != 0;       /* coordinate == boxsize: everybody tests these */
        for (int q = 0; q < R; q++) {
            if (q == rank) continue;
            const uint32_t *im = pyr + (size_t)q * chunk;
            if (orphan || tc_in_pyramid(X, im, im + imask_words, lp_max)) m |= 1u << q;
        }
        mask[g - lo] = m;
    }
    for (int q = 0; q < R; q++) {
        const uint64_t b = __ballot((m >> q) & 1u);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&s_cnt[q], (int)__popcll(b));
    }
    __syncthreads();
    if (threadIdx.x < R) blk_cnt[(size_t)threadIdx.x * nblk + blockIdx.x] = s_cnt[threadIdx.x];
}

/* this rank's row of the send-count matrix from the inclusive scan of the [dest][block] counts */
__global__ void k_ghost_counts_row(const int *__restrict__ incl, int nblk, int R, int *__restrict__ row)
{
    const int q = threadIdx.x;
    if (q < R) row[q] = incl[(size_t)(q + 1) * nblk - 1] - (q > 0 ? incl[(size_t)q * nblk - 1] : 0);
}

/* write the send buffers: grouped by destination, ascending own index inside a group (block offsets from the scan,
 * ranks inside a block from ballots) -- deterministic */
__global__ __launch_bounds__(TB) void k_ghost_fill(const float4 *__restrict__ gpos4, int lo, int hi, int R,
                                                   const uint32_t *__restrict__ mask, const int *__restrict__ blk_cnt,
                                                   const int *__restrict__ incl, int nblk, uint32_t *__restrict__ send_idx,
                                                   float4 *__restrict__ send_pos)
{
    __shared__ int s_w[TC_GHOST_MAXR][TB / 64];
    __shared__ uint64_t s_b[TC_GHOST_MAXR][TB / 64];           /* the wave's lane mask per destination */
    const int g = lo + blockIdx.x * TB + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t m = g < hi ? mask[g - lo] : 0u;
    /* every wave intrinsic sits in this loop, which all 64 lanes run together; below, lanes leave and skip at will and
     * only read the masks back (a ballot behind a divergent return relies on reconvergence the language
     * does not promise) */
    for (int q = 0; q < R; q++) {
        const uint64_t b = __ballot((m >> q) & 1u);
        if (lane == 0) { s_w[q][wave] = (int)__popcll(b); s_b[q][wave] = b; }
    }
    __syncthreads();
    if (!m) return;
    const float4 p = gpos4[g];
    for (int q = 0; q < R; q++) {
        if (!((m >> q) & 1u)) continue;
        const uint64_t b = s_b[q][wave];
        const size_t e = (size_t)q * nblk + blockIdx.x;
        int off = incl[e] - blk_cnt[e];
        for (int w = 0; w < wave; w++) off += s_w[q][w];
        off += (int)__popcll(b & ((1ull << lane) - 1));
        send_idx[off] = (uint32_t)g;
        send_pos[off] = p;
    }
}

/* received ghosts: positions into g_pos4, and the unsorted local set in ascending global index = ghosts of the
 * lower ranks (as received: by sender, ascending inside), the own range, ghosts of the higher ranks */
__global__ __launch_bounds__(TB) void k_ghost_scatter(int nghost, int nghost_lo, int own_lo, int nown,
                                                      const uint32_t *__restrict__ ghost_idx, const float4 *__restrict__ ghost_pos,
                                                      float4 *__restrict__ gpos4, uint32_t *__restrict__ lsel)
{
    const int t = blockIdx.x * TB + threadIdx.x;
    if (t >= nghost + nown) return;
    if (t < nghost_lo) {
        const uint32_t g = ghost_idx[t];
        gpos4[g] = ghost_pos[t];
        lsel[t] = g;
    } else if (t < nghost_lo + nown) {
        lsel[t] = (uint32_t)(own_lo + (t - nghost_lo));
    } else {
        const uint32_t g = ghost_idx[t - nown];
        gpos4[g] = ghost_pos[t - nown];
        lsel[t] = g;
    }
}

int tc_launch_ghost_count(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    const int R = c->nranks, nblk = c->ghost_nblk;
    if (hi > lo)
        k_ghost_count<<<(unsigned)((hi - lo + TB - 1) / TB), TB, 0, c->stream>>>(
            c->g_pos4[c->gcur], (int)lo, (int)hi, c->pyr_all, c->pyr_chunk, c->pyr_imask_words, c->rank, R, c->par.boxsize,
            c->lp_max, c->ghost_mask, c->ghost_blk_cnt, nblk);
    const int nb = (int)((hi - lo + TB - 1) / TB);
    if (nb < nblk)        /* a short last shard: the blocks it does not launch count nothing */
        for (int q = 0; q < R; q++)
            TC_HIP(c, hipMemsetAsync(c->ghost_blk_cnt + (size_t)q * nblk + nb, 0, (size_t)(nblk - nb) * sizeof(int), c->stream));
    size_t b = c->sel_tmp_bytes;
    TC_HIP(c, rocprim::inclusive_scan(c->sel_tmp, b, c->ghost_blk_cnt, c->ghost_blk_incl, (size_t)R * nblk,
                                      rocprim::plus<int>(), c->stream));
    k_ghost_counts_row<<<1, 64, 0, c->stream>>>(c->ghost_blk_incl, nblk, R, c->ghost_cnt_mat + (size_t)c->rank * R);
    TC_HIP(c, hipGetLastError());
    return 0;
}

int tc_launch_ghost_fill(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    if (hi > lo)
        k_ghost_fill<<<(unsigned)((hi - lo + TB - 1) / TB), TB, 0, c->stream>>>(
            c->g_pos4[c->gcur], (int)lo, (int)hi, c->nranks, c->ghost_mask, c->ghost_blk_cnt, c->ghost_blk_incl, c->ghost_nblk,
            c->send_idx, c->send_pos);
    TC_HIP(c, hipGetLastError());
    return 0;
}

int tc_launch_ghost_scatter(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    const int64_t tot = c->nghost + (hi - lo);
    if (tot > 0)
        k_ghost_scatter<<<(unsigned)((tot + TB - 1) / TB), TB, 0, c->stream>>>(
            (int)c->nghost, (int)c->nghost_lo, (int)lo, (int)(hi - lo), c->ghost_idx, c->ghost_pos, c->g_pos4[c->gcur], c->lsel);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* the marking kernel's level range and bounding boxes -> this pass's table layout (synchronises) */
int tc_finish_local_layout(tcgpu_ctx *c)
{
    int h[2] = {0, 0};
    TC_HIP(c, hipMemcpyAsync(h, c->lvl_range, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    c->lmin_tab = h[0] < 1 ? 1 : (h[0] > c->lmax ? c->lmax : h[0]);
    int bb[6 * (TC_MAX_LEVEL + 1)];
    TC_HIP(c, hipMemcpy(bb, c->d_bbox, sizeof(bb), hipMemcpyDeviceToHost));
    return tc_layout_table(c, bb);
}

/* Lay out this pass's cell table: level by level (lmin_tab..lmax), each the dense array of its bounding box
 * (bb = {min x, y, z, max x, y, z} per level; NULL = the whole level).  Uploads the descriptors. */
int tc_layout_table(tcgpu_ctx *c, const int *bb)
{
    size_t off = 0;
    memset(c->h_lvl, 0, sizeof(c->h_lvl));
    for (int L = c->lmin_tab; L <= c->lmax; L++) {
        tc_level_desc &D = c->h_lvl[L];
        const int nL = 1 << L;
        D.ox = D.oy = D.oz = 0;
        D.nx = D.ny = D.nz = nL;
        if (bb) {
            const int *b = bb + 6 * L;
            if (b[0] > b[3]) { D.nx = D.ny = D.nz = 0; }            /* no own query uses this level */
            else {
                D.ox = b[0]; D.oy = b[1]; D.oz = b[2];
                D.nx = b[3] - b[0] + 1; D.ny = b[4] - b[1] + 1; D.nz = b[5] - b[2] + 1;
            }
        }
        if (off + (size_t)D.nx * D.ny * D.nz >= ((size_t)1 << 32)) TC_FAIL(c, TCGPU_ERR_NOMEM, "cell table exceeds 2^32 cells");
        D.off = (uint32_t)off;
        off += (size_t)D.nx * D.ny * D.nz;
    }
    c->h_lvl[c->lmax + 1].off = (uint32_t)off;
    c->ncells_used = off;
    if (off > c->ncells_alloc) TC_FAIL(c, TCGPU_ERR_NOMEM, "cell table layout larger than its allocation");
    TC_HIP(c, hipMemcpyAsync(c->d_lvl, c->h_lvl, sizeof(tc_level_desc) * (TC_MAX_LEVEL + 1), hipMemcpyHostToDevice, c->stream));
    return 0;
}

/* ------------------------------------------------------------------ K1 keys (of the local set) */

/* the orientation table of the Hilbert curve (tc_hilbert_lut.h, 3 KB) in LDS: per-lane indexed reads */
__device__ __forceinline__ void load_hilbert_lut(unsigned short *lds)
{
    for (int t = threadIdx.x; t < TC_HILBERT_NSTATES * 64 / 2; t += TB)
        reinterpret_cast<uint32_t *>(lds)[t] = reinterpret_cast<const uint32_t *>(TC_HILBERT_LUT_DEV)[t];
    __syncthreads();
}

__global__ __launch_bounds__(TB) void k_keys_local(const float4 *__restrict__ gpos4, const uint32_t *__restrict__ lsel,
                                                   int nloc, double box, uint32_t own_lo, uint32_t own_hi,
                                                   tc_u128 *__restrict__ key, uint32_t *__restrict__ idx,
                                                   tc_u128 *__restrict__ gkey, int *__restrict__ flags)
{
    __shared__ __align__(16) unsigned short lut[TC_HILBERT_NSTATES * 64];
    load_hilbert_lut(lut);
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= nloc) return;
    const uint32_t g = lsel ? lsel[i] : (uint32_t)i;
    float4 p = gpos4[g];
    double x = (double)p.x / box, y = (double)p.y / box, z = (double)p.z / box;
    /* src/peano.c:130-132 Assert: coordinates must be inside [0,1] (NaN fails too) */
    if (!(x >= 0 && x <= 1 && y >= 0 && y <= 1 && z >= 0 && z <= 1)) {
        atomicOr(&flags[1], 1);
        p.x = p.y = p.z = 0;
    }
    uint64_t hi, lo;
    tc_peano_key_lut(p.x, p.y, p.z, box, lut, &hi, &lo);       /* src/peano.c:128-203, two levels per table look-up */
    const tc_u128 k = ((tc_u128)hi << 64) | lo;
    key[i] = k;
    idx[i] = (uint32_t)i;
    if (g >= own_lo && g < own_hi) gkey[g] = k;          /* P[].Key of the own range (src/peano.c:70), for the presentation */
}

__global__ __launch_bounds__(TB) void k_keys_xyz(const double *__restrict__ xyz, int64_t n,
                                                 uint64_t *__restrict__ khi, uint64_t *__restrict__ klo)
{
    __shared__ __align__(16) unsigned short lut[TC_HILBERT_NSTATES * 64];
    load_hilbert_lut(lut);
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const double m = 9223372036854775808.0;
    uint64_t X[3];
    X[0] = (uint64_t)(xyz[3 * i + 1] * m);
    X[1] = (uint64_t)(xyz[3 * i + 2] * m);
    X[2] = (uint64_t)(xyz[3 * i + 0] * m);
    uint64_t hi, lo;
    tc_key_lut_from_scaled(X, lut, &hi, &lo);                  /* the product's key path, on the known answers too */
    khi[i] = hi; klo[i] = lo;
}

int tc_launch_keys_local(tcgpu_ctx *c)
{
    const int n = (int)c->nloc;
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    tc_phase_begin(c, PH_KEYS);
    k_keys_local<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->g_pos4[c->gcur], c->local_full ? nullptr : c->lsel, n,
                                                         c->par.boxsize, (uint32_t)lo, (uint32_t)hi, c->key, c->idx,
                                                         c->g_key, c->flags);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

int tc_launch_keys_xyz(tcgpu_ctx *c, int64_t n, const double *d_xyz, uint64_t *d_hi, uint64_t *d_lo)
{
    k_keys_xyz<<<(unsigned)((n + TB - 1) / TB), TB, 0, c->stream>>>(d_xyz, n, d_hi, d_lo);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K3 gather into sorted local order
 *
 * The reference permutes P and SphP in place every pass (src/peano.c:85-126).  Here the global arrays keep
 * their order G between presentations; a pass gathers what its kernels read -- position, WVT hsml and the
 * carried smoothing length -- into the Peano-sorted local arrays and remembers the way back (lg). */
__global__ __launch_bounds__(TB) void k_gather_local(int nloc, const uint32_t *__restrict__ perm,
                                                     const uint32_t *__restrict__ lsel,
                                                     const float4 *__restrict__ gpos4, const float *__restrict__ ghsml,
                                                     uint32_t *__restrict__ lg, float4 *__restrict__ pos4,
                                                     float *__restrict__ hsml, float *__restrict__ hsml0)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= nloc) return;
    const uint32_t s = perm[i];
    const uint32_t g = lsel ? lsel[s] : s;
    lg[i] = g;
    pos4[i] = gpos4[g];
    const float h = ghsml[g];
    hsml[i] = h;
    hsml0[i] = h;                      /* stays as gathered: the pass's level ranges derive from it (particle_view) */
}

int tc_launch_gather_local(tcgpu_ctx *c)
{
    const int n = (int)c->nloc;
    tc_phase_begin(c, PH_PERMUTE);
    k_gather_local<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->idx_sorted, c->local_full ? nullptr : c->lsel,
                                                           c->g_pos4[c->gcur], c->g_hsml[c->gcur], c->lg, c->pos4, c->hsml,
                                                           c->hsml0);
    hipError_t e = hipSuccess;
    if (c->nranks > 1) {              /* the own particles' slots, ascending (their number is known: the own range) */
        int64_t lo, hi;
        tc_own_range(c, &lo, &hi);
        auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0),
                                                      tc_in_range{c->lg, (uint32_t)lo, (uint32_t)hi});
        size_t b = c->sel_tmp_bytes;
        e = rocprim::select(c->sel_tmp, b, rocprim::counting_iterator<uint32_t>(0), flags, c->own_list, c->d_count + 1,
                            (size_t)n, c->stream);
        c->nown = hi - lo;
    } else {
        c->nown = c->nloc;
    }
    tc_phase_end(c);
    TC_HIP(c, e);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* results of the own particles back into G order */
__global__ __launch_bounds__(TB) void k_scatter_results(int nown, const uint32_t *__restrict__ own,
                                                        const uint32_t *__restrict__ lg, const float *__restrict__ hsml,
                                                        const float *__restrict__ rho, const float *__restrict__ vhf,
                                                        float *__restrict__ ghsml, float *__restrict__ grho,
                                                        float *__restrict__ gvhf)
{
    int t = blockIdx.x * TB + threadIdx.x;
    if (t >= nown) return;
    const uint32_t i = own ? own[t] : (uint32_t)t;
    const uint32_t g = lg[i];
    const float h = hsml[i], r = rho[i], drho = vhf[i];      /* the solver kernels leave dRhodHsml in the vhf slot */
    ghsml[g] = h;
    grho[g] = r;
    gvhf[g] = (float)(1.0 / (double)(1 + h / (3 * r) * drho));   /* src/sph.c:66 */
}

int tc_launch_scatter_results(tcgpu_ctx *c)
{
    const int n = (int)c->nown;
    if (n <= 0) return 0;
    k_scatter_results<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->nranks > 1 ? c->own_list : nullptr, c->lg, c->hsml,
                                                              c->rho, c->vhf, c->g_hsml[c->gcur], c->g_rho[c->gcur],
                                                              c->g_vhf[c->gcur]);
    TC_HIP(c, hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(TB) void k_scatter_rho(int nown, const uint32_t *__restrict__ own, const uint32_t *__restrict__ lg,
                                                    const float *__restrict__ rho, float *__restrict__ grho)
{
    int t = blockIdx.x * TB + threadIdx.x;
    if (t >= nown) return;
    const uint32_t i = own ? own[t] : (uint32_t)t;
    grho[lg[i]] = rho[i];
}

int tc_launch_scatter_rho(tcgpu_ctx *c)
{
    const int n = (int)c->nown;
    if (n <= 0) return 0;
    k_scatter_rho<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->nranks > 1 ? c->own_list : nullptr, c->lg, c->rho,
                                                          c->g_rho[c->gcur]);
    TC_HIP(c, hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(TB) void k_refresh_w(int n, const uint32_t *__restrict__ lg, const float4 *__restrict__ gpos4,
                                                  float4 *__restrict__ pos4)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i < n) pos4[i].w = gpos4[lg[i]].w;
}

int tc_launch_refresh_w(tcgpu_ctx *c)
{
    const int n = (int)c->nloc;
    k_refresh_w<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->lg, c->g_pos4[c->gcur], c->pos4);
    TC_HIP(c, hipGetLastError());
    c->mirror_valid = 0;
    return 0;
}

/* ------------------------------------------------------------------ presentation: G <- Peano order */

__global__ __launch_bounds__(TB) void k_iota(uint32_t *__restrict__ dst, size_t n, uint32_t first)
{
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i < n) dst[i] = first + (uint32_t)i;
}

int tc_launch_iota(tcgpu_ctx *c, uint32_t *dst, size_t n, uint32_t first)
{
    if (n == 0) return 0;
    k_iota<<<(unsigned)((n + TB - 1) / TB), TB, 0, c->stream>>>(dst, n, first);
    TC_HIP(c, hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(TB) void k_present(int n, const uint32_t *__restrict__ perm,
                                                const float4 *__restrict__ p_in, float4 *__restrict__ p_out,
                                                const int32_t *__restrict__ id_in, int32_t *__restrict__ id_out,
                                                const float *__restrict__ h_in, float *__restrict__ h_out,
                                                const float *__restrict__ r_in, float *__restrict__ r_out,
                                                const float *__restrict__ v_in, float *__restrict__ v_out,
                                                const float *__restrict__ m_in, float *__restrict__ m_out)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    uint32_t s = perm[i];
    p_out[i] = p_in[s];
    id_out[i] = id_in[s];
    h_out[i] = h_in[s];
    r_out[i] = r_in[s];
    v_out[i] = v_in[s];
    m_out[i] = m_in[s];
}

/* side arrays without a second copy (hwvt, rhom_next: width 1; delta: width 3) go through a scratch buffer */
__global__ __launch_bounds__(TB) void k_permute_rows(int n, int width, const uint32_t *__restrict__ perm,
                                                     const float *__restrict__ in, float *__restrict__ out)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    for (int q = 0; q < width; q++) out[(size_t)width * i + q] = in[(size_t)width * s + q];
}

int tc_launch_present_permute(tcgpu_ctx *c, const uint32_t *perm)
{
    const int n = (int)c->n, a = c->gcur, b = 1 - c->gcur;
    const unsigned g = (n + TB - 1) / TB;
    k_present<<<g, TB, 0, c->stream>>>(n, perm, c->g_pos4[a], c->g_pos4[b], c->g_id[a], c->g_id[b], c->g_hsml[a],
                                       c->g_hsml[b], c->g_rho[a], c->g_rho[b], c->g_vhf[a], c->g_vhf[b], c->g_rhom[a],
                                       c->g_rhom[b]);
    float *tmp = c->guess;                                   /* 3*cap floats of scratch (see ensure_capacity) */
    float *side[3] = {c->hwvt, c->rhom_next, c->delta};
    const int width[3] = {1, 1, 3};
    for (int q = 0; q < 3; q++) {
        k_permute_rows<<<g, TB, 0, c->stream>>>(n, width[q], perm, side[q], tmp);
        TC_HIP(c, hipMemcpyAsync(side[q], tmp, (size_t)n * width[q] * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    TC_HIP(c, hipGetLastError());
    c->gcur = b;
    return 0;
}

/* ------------------------------------------------------------------ K4' cell tables */

/* Level-L cell of a particle = top L bits of each scaled coordinate, i.e. the octree cell
 * whose Hilbert prefix is the top 3L key bits.  In Peano order every cell is one contiguous
 * index run, so (first, last+1) per cell is a complete neighbour index.  Runs are recorded
 * with atomics (min of first, max of last+1) so that a run split by an "orphan" (below) still
 * yields one range; `first` is stored complemented so that an all-zero table means "empty".
 *
 * Orphans: a coordinate exactly equal to boxsize scales to X = 2^63 (src/peano.c:134-136);
 * the reference's transform then keys the particle away from its spatial neighbours.  Such
 * particles are kept out of the tables and tested by brute force in every query. */
__device__ __forceinline__ void cell_coords(float4 p, double box, int lmax, uint32_t c[3], bool *orphan)
{
    uint64_t X[3];
    tc_scaled_coords(p.x, p.y, p.z, box, X);
    *orphan = ((X[0] | X[1] | X[2]) >> 63) != 0;
    const int sh = 63 - lmax;
    const uint32_t mask = (1u << lmax) - 1;
    c[0] = (uint32_t)(X[2] >> sh) & mask; /* x */
    c[1] = (uint32_t)(X[0] >> sh) & mask; /* y */
    c[2] = (uint32_t)(X[1] >> sh) & mask; /* z */
}

/* coarsest level (1..lmax) at which two lmax-level cells differ; lmax+1 if identical */
__device__ __forceinline__ int first_diff_level(const uint32_t a[3], const uint32_t b[3], int lmax)
{
    uint32_t x = (a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]);
    if (!x) return lmax + 1;
    int top = 31 - __clz(x);          /* highest differing bit, 0..lmax-1 */
    return lmax - top;                /* bit lmax-1 <-> level 1 */
}

/* table slot of the level-L cell (x, y, z), or false if the cell lies outside the level's box (a ghost far from
 * every own query: nobody will ask for it) */
__device__ __forceinline__ bool tc_cell_slot(const tc_level_desc &D, uint32_t x, uint32_t y, uint32_t z, size_t *o)
{
    const int ix = (int)x - D.ox, iy = (int)y - D.oy, iz = (int)z - D.oz;
    if (ix < 0 || ix >= D.nx || iy < 0 || iy >= D.ny || iz < 0 || iz >= D.nz) return false;
    *o = (size_t)D.off + ((size_t)ix * D.ny + iy) * D.nz + iz;
    return true;
}

__global__ __launch_bounds__(TB) void k_cells(const float4 *__restrict__ pos4, int n, double box, int lmax, int lmin_tab,
                                              const tc_level_desc *__restrict__ lvl, uint2 *__restrict__ cells,
                                              uint32_t *__restrict__ orphans, int *__restrict__ norph,
                                              int *__restrict__ flags)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    uint32_t ci[3], cp[3], cn[3];
    bool oi, op = true, on = true;
    cell_coords(pos4[i], box, lmax, ci, &oi);
    if (oi) {
        int k = atomicAdd(norph, 1);
        if (k < TC_MAX_ORPHANS) orphans[k] = (uint32_t)i;
        else atomicOr(&flags[3], 1);
        return;
    }
    if (i > 0) cell_coords(pos4[i - 1], box, lmax, cp, &op);
    if (i < n - 1) cell_coords(pos4[i + 1], box, lmax, cn, &on);
    int dprev = op ? 1 : first_diff_level(ci, cp, lmax);   /* head at every level >= dprev */
    int dnext = on ? 1 : first_diff_level(ci, cn, lmax);   /* tail at every level >= dnext */
    int dmin = dprev < dnext ? dprev : dnext;
    if (dmin < lmin_tab) dmin = lmin_tab;                  /* levels no query of this pass can ask for are not built */
    for (int L = lmax; L >= dmin; L--) {
        int sh = lmax - L;
        size_t o;
        if (!tc_cell_slot(lvl[L], ci[0] >> sh, ci[1] >> sh, ci[2] >> sh, &o)) continue;
        if (L >= dprev) atomicMax(&cells[o].x, ~(uint32_t)i);          /* max(~i) = ~min(i): zero-initialised */
        if (L >= dnext) atomicMax(&cells[o].y, (uint32_t)(i + 1));
    }
}

int tc_launch_cells(tcgpu_ctx *c)
{
    int n = (int)c->nloc;
    tc_phase_begin(c, PH_CELLS);
    TC_HIP(c, hipMemsetAsync(c->cells, 0, c->ncells_used * sizeof(uint2), c->stream));
    TC_HIP(c, hipMemsetAsync(c->norph, 0, sizeof(int), c->stream));
    k_cells<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4, n, c->par.boxsize, c->lmax, c->lmin_tab, c->d_lvl, c->cells,
                                                     c->orphans, c->norph, c->flags);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    c->index_valid = 1;
    c->mirror_valid = 0;
    return 0;
}

/* ------------------------------------------------------------------ K4p cell starts in CURVE order

 * pf[L][p] = index of the first particle whose Peano key has a level-L prefix >= p, p = 0 .. 8^L (pf[L][8^L] = n): the
 * cells of a level taken in curve order are consecutive index ranges, so the particles of the cells p .. q are
 * [pf[L][p], pf[L][q + 1]) -- what the ordered traversals (k_xruns, k_wvt_exact4, k_wvt_exact_w) turn their key ranges
 * into.  Keys, not positions, decide membership: a particle with a coordinate == boxsize ("orphan", k_cells) sits in
 * the cell its key names and is found there.
 * Levels lmin .. lc (lc = lmax - 3) are dense tables built from the sorted keys by one pass: heads of runs are marked,
 * empty cells take the head of the next occupied one (one reverse running minimum over all of them; level t's entries
 * carry the bias t (n + 1) so that the minimum never crosses from one level into the previous one).
 * The three deepest levels hold 8/7 x 8^lmax entries -- 153 M at lmax = 9, where a sharded rank's local set (3.8e6
 * particles of 1.6e7) occupies a small part of the key space but paid the memset and the scan of all of it (0.5 ms per
 * iteration).  They are filled block by block instead (k_pf_deep): one wavefront per level-lc cell; an EMPTY cell is left
 * alone (tc_pf_first answers from the level-lc entries, which say so), an occupied one gets its 8 + 64 + 512 entries
 * from its own particles, the running minimum starting from the head of the next cell (= the cell's own end). */
__global__ __launch_bounds__(TB) void k_pf_mark(const tc_u128 *__restrict__ key, int n, int lmin, int lmax, uint32_t *__restrict__ pf)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i > n) return;
    if (i == n) {                                             /* the end marker of every level */
        uint32_t off = 0;
        for (int L = lmin; L <= lmax; L++) {
            const uint32_t cells = 1u << (3 * L);
            pf[off + cells] = (uint32_t)n + (uint32_t)(L - lmin) * (uint32_t)(n + 1);
            off += cells + 1;
        }
        return;
    }
    const tc_u128 k = key[i];
    const tc_u128 kp = i > 0 ? key[i - 1] : 0;
    uint32_t off = 0;
    for (int L = lmin; L <= lmax; L++) {
        const uint32_t cells = 1u << (3 * L);
        const uint32_t p = (uint32_t)(k >> (128 - 3 * L)), pp = (uint32_t)(kp >> (128 - 3 * L));
        if (i == 0 || p != pp) pf[off + p] = (uint32_t)i + (uint32_t)(L - lmin) * (uint32_t)(n + 1);
        off += cells + 1;
    }
}

/* the levels lc + 1 .. lmax (at most three) under one level-lc cell per wavefront */
__global__ __launch_bounds__(256) void k_pf_deep(const tc_u128 *__restrict__ key, int n, int lmin, int lc, int lmax,
                                                 uint32_t *__restrict__ pf)
{
    __shared__ uint32_t lds[4][8 + 64 + 512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t ncell = 1u << (3 * lc);
    if (c >= ncell) return;
    uint32_t offC = 0;
    for (int L = lmin; L < lc; L++) offC += (1u << (3 * L)) + 1u;
    const uint32_t biasC = (uint32_t)(lc - lmin) * (uint32_t)(n + 1);
    const uint32_t lo = pf[offC + c] - biasC, hi = pf[offC + c + 1u] - biasC;
    if (lo == hi) return;                                     /* empty: nobody reads below it (tc_pf_first) */
    uint32_t *t = lds[wave];
    const int nd = lmax - lc;                                 /* 1 .. 3 deep levels */
    const int tot = nd == 3 ? 584 : nd == 2 ? 72 : 8;
    for (int q = lane; q < tot; q += 64) t[q] = 0xffffffffu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    /* heads: the first particle of every deep cell (the keys are sorted: a particle is a head when its prefix differs
     * from its predecessor's) */
    for (uint32_t i = lo + (uint32_t)lane; i < hi; i += 64u) {
        const tc_u128 k = key[i];
        const tc_u128 kp = i > lo ? key[i - 1] : 0;
        int base = 0;
        for (int d = 1; d <= nd; d++) {
            const int L = lc + d;
            const uint32_t m = (1u << (3 * d)) - 1u;
            const uint32_t p = (uint32_t)(k >> (128 - 3 * L)) & m, pp = (uint32_t)(kp >> (128 - 3 * L)) & m;
            if (i == lo || p != pp) t[base + (int)p] = i;
            base += 1 << (3 * d);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    /* running minimum from the back, starting at the head of whatever follows this cell (hi), level by level; a lane
     * takes a contiguous piece, the pieces are chained through a wave scan */
    uint32_t offL = offC + ncell + 1u;                        /* level lc + 1 starts behind level lc */
    int base = 0;
    for (int d = 1; d <= nd; d++) {
        const int E = 1 << (3 * d);
        const int per = (E + 63) / 64;                        /* 1, 1, 8 entries per lane */
        const int a0 = lane * per, a1 = a0 + per < E ? a0 + per : E;
        uint32_t mine = 0xffffffffu;
        for (int q = a0; q < a1 && q < E; q++) mine = min(mine, t[base + q]);
        /* suffix minimum over the lanes behind this one */
        uint32_t suf = mine;
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = (uint32_t)__shfl_down((int)suf, sft);
            if (lane + sft < 64) suf = min(suf, o);
        }
        uint32_t run = (uint32_t)__shfl_down((int)suf, 1);   /* minimum of everything behind this lane's piece */
        if (lane == 63) run = 0xffffffffu;
        run = min(run, hi);
        for (int q = a1 - 1; q >= a0; q--) {
            if (q < E) { run = min(run, t[base + q]); pf[offL + c * (uint32_t)E + (uint32_t)q] = run; }
        }
        offL += (1u << (3 * (lc + d))) + 1u;
        base += E;
    }
}

size_t tc_pf_entries(int lmin, int lmax)
{
    size_t t = 0;
    for (int L = lmin; L <= lmax; L++) t += ((size_t)1 << (3 * L)) + 1;
    return t;
}

int tc_pf_temp_bytes(size_t nent, size_t *bytes)
{
    size_t b = 0;
    auto it = std::make_reverse_iterator((uint32_t *)nullptr + nent);
    hipError_t e = rocprim::inclusive_scan(nullptr, b, it, it, nent, rocprim::minimum<uint32_t>());
    *bytes = b;
    return e == hipSuccess ? 0 : -1;
}

int tc_launch_pfirst(tcgpu_ctx *c)
{
    const int n = (int)c->nloc;
    const int lmin = c->lmin_tab, lmax = c->lmax;
    /* dense by one scan up to level lc, block by block below it -- when the local set is thin in key space (a sharded
     * rank: 3.8e6 particles under Lmax = 9 chosen for 1.6e7); a full set occupies nearly every block, and there the
     * single scan over everything is the cheaper way (measured at 2e6: 0.22 against 0.30 ms, and 0.1 ms more in k_xruns
     * for the look-ups' second load) */
    const bool thin = c->pf_mode == 1 ? false : c->pf_mode == 2 ? true : (double)n < 0.06 * pow(8.0, (double)lmax);   /* option "pf_mode": tests */
    const int lc = !thin ? lmax : lmax - 3 > lmin ? lmax - 3 : lmin;
    c->pf_valid = 0;
    if ((uint64_t)(lc - lmin + 1) * (uint64_t)(n + 1) >= 0xffffffffull) TC_FAIL(c, TCGPU_ERR_ARG, "cell-start table: too many particles for the level bias");
    const size_t nent = tc_pf_entries(lmin, lmax);
    const size_t nentC = tc_pf_entries(lmin, lc);
    if (nent > c->pf_alloc) {
        hipFree(c->pf); hipFree(c->pf_tmp);
        c->pf = nullptr; c->pf_tmp = nullptr; c->pf_alloc = 0;
        TC_HIP(c, hipMalloc(&c->pf, nent * sizeof(uint32_t)));
        if (tc_pf_temp_bytes(nent, &c->pf_tmp_bytes)) TC_FAIL(c, TCGPU_ERR_HIP, "scan temp query failed");
        TC_HIP(c, hipMalloc(&c->pf_tmp, c->pf_tmp_bytes ? c->pf_tmp_bytes : 16));
        c->pf_alloc = nent;
    }
    tc_phase_begin(c, PH_CELLS);
    TC_HIP(c, hipMemsetAsync(c->pf, 0xff, nentC * sizeof(uint32_t), c->stream));
    k_pf_mark<<<(n + 1 + TB - 1) / TB, TB, 0, c->stream>>>(c->key_sorted, n, lmin, lc, c->pf);
    auto it = std::make_reverse_iterator(c->pf + nentC);
    size_t b = c->pf_tmp_bytes;
    hipError_t e = rocprim::inclusive_scan(c->pf_tmp, b, it, it, nentC, rocprim::minimum<uint32_t>(), c->stream);
    if (e == hipSuccess && lmax > lc) {
        const size_t ncell = (size_t)1 << (3 * lc);
        k_pf_deep<<<(unsigned)((ncell + 3) / 4), 256, 0, c->stream>>>(c->key_sorted, n, lmin, lc, lmax, c->pf);
    }
    tc_phase_end(c);
    TC_HIP(c, e);
    TC_HIP(c, hipGetLastError());
    c->pf_lmin = lmin;
    c->pf_lc = lc;
    c->pf_valid = 1;
    return 0;
}

/* ------------------------------------------------------------------ K4m row-major mirror

 * The cell table is dense in (x, y, z) with z fastest, so the cells of one (x, y) row that a ball
 * reaches are consecutive table entries -- but their particles are scattered over the Peano order.
 * The mirror stores a second copy of the positions in TABLE order (level by level, cell by cell):
 * cum[] is the exclusive prefix sum of the cell populations over the whole table, cell o owns the
 * slots [cum[o], cum[o+1]).  A ball query then needs two cum[] reads per row instead of one table
 * read per cell, and its candidates are contiguous runs of `mirror`. */
struct tc_cell_count {
    __device__ uint32_t operator()(const uint2 &ce) const
    {
        const uint32_t s0 = ~ce.x, e0 = ce.y;
        return e0 > s0 ? e0 - s0 : 0u;
    }
};

int tc_scan_temp_bytes(size_t ncell, size_t *bytes)
{
    size_t b = 0;
    auto in = rocprim::make_transform_iterator((const uint2 *)nullptr, tc_cell_count());
    hipError_t e = rocprim::exclusive_scan(nullptr, b, in, (uint32_t *)nullptr, 0u, ncell, rocprim::plus<uint32_t>());
    *bytes = b;
    return e == hipSuccess ? 0 : -1;
}

__global__ __launch_bounds__(TB) void k_mirror(const float4 *__restrict__ pos4, int n, double box, int lmax, int lmin_rm, int lmax_rm,
                                               const tc_level_desc *__restrict__ lvl, const uint2 *__restrict__ cells,
                                               const uint32_t *__restrict__ cum, float4 *__restrict__ mirror,
                                               uint32_t *__restrict__ mirror_idx)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos4[i];
    uint32_t ci[3];
    bool orphan;
    cell_coords(p, box, lmax, ci, &orphan);
    if (orphan) return;                                   /* not in the table (see k_cells) */
    for (int L = lmin_rm; L <= lmax_rm; L++) {
        const int sh = lmax - L;
        size_t o;
        if (!tc_cell_slot(lvl[L], ci[0] >> sh, ci[1] >> sh, ci[2] >> sh, &o)) continue;
        const uint32_t slot = cum[o] + ((uint32_t)i - ~cells[o].x);
        mirror[slot] = p;
        mirror_idx[slot] = (uint32_t)i;
    }
}

int tc_launch_mirror(tcgpu_ctx *c)
{
    c->mirror_valid = 0;
    if (!c->rows || c->lmax_rm <= 0 || !c->index_valid) return 0;
    const int n = (int)c->nloc;
    {   /* slots: one per particle of THIS pass's local set and mirrored level (with head room; at most the capacity) */
        size_t per = (size_t)c->nloc + (size_t)c->nloc / 8 + 1024;
        if (per > (size_t)c->cap) per = (size_t)c->cap;
        const size_t need = (size_t)(c->lmax_rm - c->lmin_rm + 1) * (size_t)c->nloc;
        if (need > c->mirror_alloc) {
            const size_t nslot = (size_t)(c->lmax_rm0 - c->lmin_rm0 + 1) * per;
            hipFree(c->mirror); hipFree(c->mirror_idx);
            c->mirror = nullptr; c->mirror_idx = nullptr; c->mirror_alloc = 0;
            TC_HIP(c, hipMalloc(&c->mirror, (nslot + 1) * sizeof(float4)));
            TC_HIP(c, hipMalloc(&c->mirror_idx, (nslot + 1) * sizeof(uint32_t)));
            c->mirror_alloc = nslot;
            /* one extra slot infinitely far away: padding lanes of a candidate batch load it and fail every
             * distance test by themselves (no per-lane "active" flag in the predicate) */
            const float inf = HUGE_VALF;
            const float4 far = make_float4(inf, inf, inf, 0.0f);
            const uint32_t none = 0xffffffffu;
            TC_HIP(c, hipMemcpy(c->mirror + nslot, &far, sizeof(far), hipMemcpyHostToDevice));
            TC_HIP(c, hipMemcpy(c->mirror_idx + nslot, &none, sizeof(none), hipMemcpyHostToDevice));
        }
    }
    /* scan only the mirrored levels (contiguous in the table); `cum` is addressed with table offsets through a pointer
     * shifted back by the offset of the first mirrored level (tc_cum_base) */
    const size_t first = c->h_lvl[c->lmin_rm].off;
    const tc_level_desc &T = c->h_lvl[c->lmax_rm];
    const size_t ncell = (size_t)T.off + (size_t)T.nx * T.ny * T.nz - first + 1;     /* +1: the end of the last cell */
    tc_phase_begin(c, PH_MIRROR);
    auto in = rocprim::make_transform_iterator((const uint2 *)c->cells + first, tc_cell_count());
    size_t b = c->scan_tmp_bytes;
    hipError_t e = rocprim::exclusive_scan(c->scan_tmp, b, in, c->cum, 0u, ncell, rocprim::plus<uint32_t>(), c->stream);
    if (e == hipSuccess)
        k_mirror<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4, n, c->par.boxsize, c->lmax, c->lmin_rm, c->lmax_rm,
                                                         c->d_lvl, c->cells, tc_cum_base(c), c->mirror, c->mirror_idx);
    tc_phase_end(c);
    TC_HIP(c, e);
    TC_HIP(c, hipGetLastError());
    c->mirror_valid = 1;
    return 0;
}

/* ------------------------------------------------------------------ K4g hsml guess */

/* The reference's first-pass guess (src/tree.c:113-121) reads Npart and Size of the tree node
 * P[i].Tree_Parent.  That node is determined by the shared-prefix structure of the sorted keys
 * and by the "collapse leaves of <= 8 particles" rule of src/tree.c:201-226; both are evaluated
 * here per particle from a +-8 window of sorted keys, without building the tree:
 *   lcp(k) = Hilbert levels shared by particles k-1 and k   (lcp(0) = 0)
 *   raw parent level of j = max(lcp(j), lcp(j+1))           (src/tree.c:163-171,231)
 *   a node X (level l, particles [first,last], count c <= 8) is collapsed when particle
 *   last+1 arrives and starts a new branch (lcp(last) > lcp(last+1)) and either X is the
 *   highest completed node (l == lcp(last+1)+1) or, that one being larger than 8, X is the
 *   previous particle's parent (l == lcp(last)).  The outermost collapsed ancestor wins.
 * Not reproduced: particle 0 being keyed from f32 coordinates (src/tree.c:137-141) and the
 * 42-level depth limit -- both need particles closer than 2^-21 of the box. */
__device__ __forceinline__ int lcp_at(const tc_u128 *__restrict__ key, int k, int n)
{
    if (k <= 0) return 0;
    if (k >= n) return -1;
    tc_u128 a = key[k - 1], b = key[k];
    return tc_common_levels((uint64_t)(a >> 64), (uint64_t)a, (uint64_t)(b >> 64), (uint64_t)b);
}

__global__ __launch_bounds__(TB) void k_guess(const tc_u128 *__restrict__ key, int n, double box,
                                              float *__restrict__ guess)
{
    int j = blockIdx.x * TB + threadIdx.x;
    if (j >= n) return;

    int lc[18];                         /* lc[t] = lcp(j - 8 + t), t = 0..17 */
    for (int t = 0; t < 18; t++) lc[t] = lcp_at(key, j - 8 + t, n);
#define LCP(k) lc[(k) - j + 8]

    int lj = LCP(j), lj1 = LCP(j + 1);
    int maxlev = lj > lj1 ? lj : lj1;
    if (n == 1) maxlev = 0;

    int plevel = -1, pcount = 0;
    for (int l = 1; l <= maxlev && l < TC_NTRIPLETS; l++) {
        int first = j, last = j;
        while (first > 0 && j - first < 8 && LCP(first) >= l) first--;
        if (first > 0 && LCP(first) >= l) continue;            /* more than 8 to the left */
        while (last < n - 1 && last - j < 8 && LCP(last + 1) >= l) last++;
        if (last < n - 1 && LCP(last + 1) >= l) continue;
        int cnt = last - first + 1;
        if (cnt > 8) continue;
        if (last == n - 1) continue;                           /* final branch: no later particle */
        int l_last = LCP(last), l_next = LCP(last + 1);
        if (!(l_last > l_next)) continue;                      /* src/tree.c:175: not a new branch */
        if (l == l_next + 1 || l == l_last) { plevel = l; pcount = cnt; break; }
    }
#undef LCP
    if (plevel < 0) {
        plevel = maxlev;
        if (plevel == 0) {
            pcount = n;
        } else {
            /* count particles sharing the top 3*plevel key bits with j */
            int sh = 128 - 3 * plevel;
            tc_u128 pre = key[j] >> sh;
            int lo = 0, hi = j;                                 /* first index with prefix >= pre */
            while (lo < hi) { int mid = (lo + hi) >> 1; if ((key[mid] >> sh) < pre) lo = mid + 1; else hi = mid; }
            int first = lo;
            lo = j; hi = n;                                     /* first index with prefix > pre */
            while (lo < hi) { int mid = (lo + hi) >> 1; if ((key[mid] >> sh) <= pre) lo = mid + 1; else hi = mid; }
            pcount = lo - first;
        }
    }
    /* src/tree.c:304 Size, :117-120 */
    float size = (float)(box / (double)(1 << plevel));
    float numDens = (float)pcount / (size * size * size);
    float sz = (float)pow(TC_FOURPITHIRD / (double)numDens, 1. / 3.);
    guess[j] = 2 * sz;
}

int tc_launch_guess(tcgpu_ctx *c)
{
    int n = (int)c->nloc;                        /* cold passes run on the full set: the tree the guess reads is global */
    tc_phase_begin(c, PH_GUESS);
    k_guess<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->key_sorted, n, c->par.boxsize, c->guess);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ exact block reductions
 *
 * Sums that decide control flow (the mean density error, the normalisation of the model hsml) are accumulated
 * as 128-bit fixed-point integers: integer addition is associative, so the result does not depend on how the
 * particles are split over threads, blocks or GPUs.  A block leaves {sum lo, sum hi, count, max} in four 64-bit
 * slots; k_final_exact adds the blocks and writes the total as three 40-bit limbs in doubles (exact, and still
 * exact after an all-reduce over up to 2^13 ranks) plus count and max. */
typedef unsigned __int128 tc_u128s;

__device__ __forceinline__ tc_u128s wave_sum_u128(tc_u128s v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t lo = __shfl_xor((unsigned long long)(uint64_t)v, o);
        const uint64_t hi = __shfl_xor((unsigned long long)(uint64_t)(v >> 64), o);
        v += ((tc_u128s)hi << 64) | lo;
    }
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void block_reduce_exact(tc_u128s s, double cnt, double mx, uint64_t *out)
{
    __shared__ uint64_t sh[4][TB / 64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    s = wave_sum_u128(s); cnt = wave_sum(cnt); mx = wave_max(mx);          /* counts are integers: exact in f64 */
    if (l == 0) { sh[0][w] = (uint64_t)s; sh[1][w] = (uint64_t)(s >> 64); sh[2][w] = __double_as_longlong(cnt); sh[3][w] = __double_as_longlong(mx); }
    __syncthreads();
    if (threadIdx.x == 0) {
        tc_u128s a = 0;
        double b = 0, m = 0;
        for (int k = 0; k < TB / 64; k++) {
            a += ((tc_u128s)sh[1][k] << 64) | sh[0][k];
            b += __longlong_as_double(sh[2][k]);
            m = fmax(m, __longlong_as_double(sh[3][k]));
        }
        out[0] = (uint64_t)a; out[1] = (uint64_t)(a >> 64); out[2] = __double_as_longlong(b); out[3] = __double_as_longlong(m);
    }
}

/* fin[0..2] = 40-bit limbs of the total, fin[3] = count, fin[4] = max */
__global__ __launch_bounds__(TB) void k_final_exact(const uint64_t *__restrict__ part, int nblocks, double *__restrict__ fin)
{
    tc_u128s s = 0;
    double cnt = 0, mx = 0;
    for (int b = threadIdx.x; b < nblocks; b += TB) {
        s += ((tc_u128s)part[4 * b + 1] << 64) | part[4 * b];
        cnt += __longlong_as_double(part[4 * b + 2]);
        mx = fmax(mx, __longlong_as_double(part[4 * b + 3]));
    }
    __shared__ uint64_t tot[4];
    block_reduce_exact(s, cnt, mx, tot);
    if (threadIdx.x == 0) {
        const tc_u128s t = ((tc_u128s)tot[1] << 64) | tot[0];
        const uint64_t m40 = ((uint64_t)1 << 40) - 1;
        fin[0] = (double)((uint64_t)t & m40);
        fin[1] = (double)((uint64_t)(t >> 40) & m40);
        fin[2] = (double)(uint64_t)(t >> 80);
        fin[3] = __longlong_as_double(tot[2]);
        fin[4] = __longlong_as_double(tot[3]);
    }
}

/* the error flags as 0/1 doubles, so that they can ride in the pass's all-reduce (api.hip) */
__global__ void k_flags_to_f64(const int *__restrict__ flags, double *__restrict__ out)
{
    if (threadIdx.x < 4) out[threadIdx.x] = flags[threadIdx.x] != 0 ? 1.0 : 0.0;
    if (threadIdx.x == 4) out[4] = flags[5] != 0 ? 1.0 : 0.0;
}

int tc_launch_flags_to_f64(tcgpu_ctx *c, double *d_out)
{
    k_flags_to_f64<<<1, 64, 0, c->stream>>>(c->flags, d_out);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ density model (API) */

__global__ __launch_bounds__(TB) void k_model(const float4 *__restrict__ pos4, int n, double boxhalf,
                                              const tc_halo_dev *__restrict__ halo, int nhalos,
                                              float *__restrict__ out)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    float4 p = pos4[i];
    out[i] = tc_density_model(p.x, p.y, p.z, boxhalf, halo, nhalos);
}

int tc_launch_model(tcgpu_ctx *c, float *d_out)
{
    int n = (int)c->n;
    k_model<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->g_pos4[c->gcur], n, c->par.boxsize * 0.5, c->d_halo,
                                                     c->par.nhalos, d_out);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K6 error sums */

/* src/wvt_relax.c:73-85 over the own range [lo,hi) of G.  err is an f32; it enters the sum as the integer
 * trunc(err * 2^36) (exact for every err >= 2^-12, 1.5e-11 absolute below; errors above 1.3e8 saturate). */
__global__ __launch_bounds__(TB) void k_error(const float4 *__restrict__ pos4, const float *__restrict__ rho_sph,
                                              int lo, int hi, double boxhalf,
                                              const tc_halo_dev *__restrict__ halo, int nhalos,
                                              uint64_t *__restrict__ part)
{
    tc_u128s s = 0;
    double cnt = 0, mx = 0;
    for (int i = lo + blockIdx.x * TB + threadIdx.x; i < hi; i += gridDim.x * TB) {
        float4 p = pos4[i];
        float rho = tc_density_model(p.x, p.y, p.z, boxhalf, halo, nhalos);
        float err = (float)(fabs((double)(rho_sph[i] - rho)) / (double)rho);
        mx = fmax((double)err, mx);
        const double e36 = (double)err * TC_ERR_SCALE;
        s += (tc_u128s)(e36 < 9.0e18 ? (uint64_t)e36 : (uint64_t)9.0e18);    /* NaN -> saturated too (flags report it) */
        cnt += 1;
    }
    block_reduce_exact(s, cnt, mx, part + 4 * blockIdx.x);
}

int tc_launch_error(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    int nb = TC_RED_BLOCKS;
    tc_phase_begin(c, PH_ERROR);
    k_error<<<nb, TB, 0, c->stream>>>(c->g_pos4[c->gcur], c->g_rho[c->gcur], (int)lo, (int)hi, c->par.boxsize * 0.5,
                                      c->d_halo, c->par.nhalos, (uint64_t *)c->red);
    k_final_exact<<<1, TB, 0, c->stream>>>((const uint64_t *)c->red, nb, tc_pass_scalars(c));
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K7/K8 model hsml */

/* src/wvt_relax.c:108-118 over the own range.  h^3 (f32) enters the sum as the integer trunc(h^3 / unit), `unit` a
 * power of two 2^30 below the smallest h^3 the model allows (tcgpu_set_model: h3_unit): exact for density contrasts up to 2^33. */
__global__ __launch_bounds__(TB) void k_model_hsml(const float4 *__restrict__ pos4, int lo, int hi, double boxhalf,
                                                   double mpart, double inv_unit, const tc_halo_dev *__restrict__ halo, int nhalos,
                                                   float *__restrict__ rhom, float *__restrict__ hw,
                                                   uint64_t *__restrict__ part)
{
    tc_u128s s = 0;
    for (int i = lo + blockIdx.x * TB + threadIdx.x; i < hi; i += gridDim.x * TB) {
        float4 p = pos4[i];
        float rho = tc_density_model(p.x, p.y, p.z, boxhalf, halo, nhalos);
        rhom[i] = rho;
        float h = (float)pow(TC_DESNNGB * mpart / (double)rho / TC_FOURPITHIRD, 1. / 3.);
        hw[i] = h;
        const double h3 = (double)(h * h * h) * inv_unit;
        s += (tc_u128s)(h3 < 9.0e18 ? (uint64_t)h3 : (uint64_t)9.0e18);
    }
    block_reduce_exact(s, 0, 0, part + 4 * blockIdx.x);
}

/* src/wvt_relax.c:120-124; the normalised value also goes into g_pos4.w for the sweep's gathers */
__global__ __launch_bounds__(TB) void k_scale_hsml(float4 *__restrict__ pos4, int lo, int hi, const double *__restrict__ fin,
                                                   double unit, float *__restrict__ hw)
{
    int i = lo + blockIdx.x * TB + threadIdx.x;
    if (i >= hi) return;
    const double sum_h3 = tc_limbs_to_double(fin[0], fin[1], fin[2], unit);
    float norm = (float)pow(TC_DESNNGB / sum_h3 / TC_FOURPITHIRD, 1.0 / 3.0);
    float h = hw[i] * norm;
    hw[i] = h;
    pos4[i].w = h;
}

int tc_launch_model_hsml_own(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    int nb = TC_RED_BLOCKS;
    tc_phase_begin(c, PH_MODEL_HSML);
    /* Rho_Model goes to a side buffer: the reference stores it when the sweep runs (wvt_relax.c:113),
     * so it is committed by tc_launch_commit_rhom(), not on iterations that stop before the sweep */
    k_model_hsml<<<nb, TB, 0, c->stream>>>(c->g_pos4[c->gcur], (int)lo, (int)hi, c->par.boxsize * 0.5, c->par.mpart_gas,
                                           1.0 / c->h3_unit, c->d_halo, c->par.nhalos, c->rhom_next, c->hwvt,
                                           (uint64_t *)c->red);
    k_final_exact<<<1, TB, 0, c->stream>>>((const uint64_t *)c->red, nb, tc_pass_scalars(c) + TC_PS_H3);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

int tc_launch_scale_hsml_own(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    tc_phase_begin(c, PH_MODEL_HSML);
    if (hi > lo)
        k_scale_hsml<<<(unsigned)((hi - lo + TB - 1) / TB), TB, 0, c->stream>>>(c->g_pos4[c->gcur], (int)lo, (int)hi,
                                                                                tc_pass_scalars(c) + TC_PS_H3, c->h3_unit,
                                                                                c->hwvt);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

int tc_launch_commit_rhom(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    if (hi > lo)
        TC_HIP(c, hipMemcpyAsync(c->g_rhom[c->gcur] + lo, c->rhom_next + lo, (size_t)(hi - lo) * sizeof(float),
                                 hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

/* delta (G order) = step * U for the fused kernel's unit-step sums in local order (src/wvt_relax.c:167-169 with
 * the scalar step factored out of the neighbour sum) */
__global__ __launch_bounds__(TB) void k_apply_step(int nown, const uint32_t *__restrict__ own, const uint32_t *__restrict__ lg,
                                                   const double *__restrict__ ustep, float *__restrict__ delta, double step)
{
    int t = blockIdx.x * TB + threadIdx.x;
    if (t >= nown) return;
    const uint32_t i = own ? own[t] : (uint32_t)t;
    const uint32_t g = lg[i];
    for (int q = 0; q < 3; q++) delta[3 * (size_t)g + q] = (float)(step * ustep[3 * (size_t)i + q]);
}

int tc_launch_apply_step(tcgpu_ctx *c, double step)
{
    const int n = (int)c->nown;
    tc_phase_begin(c, PH_WVT);
    if (n > 0)
        k_apply_step<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->nranks > 1 ? c->own_list : nullptr, c->lg, c->ustep,
                                                             c->delta, step);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K10 move + wrap */

/* src/wvt_relax.c:193-213 */
__device__ __forceinline__ float move1(float p, float d, double box)
{
    p += (float)((double)d * box);
    int guard = 0;
    while (p < 0 && guard++ < 64) p = (float)((double)p + box);
    while ((double)p > box && guard++ < 64) p = (float)((double)p - box);
    return p;
}

__global__ __launch_bounds__(TB) void k_move(float4 *__restrict__ pos4, const float *__restrict__ delta,
                                             int lo, int hi, double box)
{
    int i = lo + blockIdx.x * TB + threadIdx.x;
    if (i >= hi) return;
    float4 p = pos4[i];
    p.x = move1(p.x, delta[3 * (size_t)i], box);
    p.y = move1(p.y, delta[3 * (size_t)i + 1], box);
    p.z = move1(p.z, delta[3 * (size_t)i + 2], box);
    pos4[i] = p;
}

int tc_launch_move(tcgpu_ctx *c)
{
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    tc_phase_begin(c, PH_MOVE);
    if (hi > lo)
        k_move<<<(unsigned)((hi - lo + TB - 1) / TB), TB, 0, c->stream>>>(c->g_pos4[c->gcur], c->delta, (int)lo, (int)hi,
                                                                          c->par.boxsize);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    c->index_valid = 0; c->mirror_valid = 0;
    c->ustep_valid = 0;
    c->w_valid = 0;
    c->pos_all_valid = 0;                         /* sharded contexts: the other ranks' copies of the own range are stale */
    return 0;
}

/* ------------------------------------------------------------------ curl: A into local order, B back */

__global__ __launch_bounds__(TB) void k_gather_rows3(int n, const uint32_t *__restrict__ lg, const float *__restrict__ in,
                                                     float *__restrict__ out)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const size_t g = lg[i];
    for (int q = 0; q < 3; q++) out[3 * (size_t)i + q] = in[3 * g + q];
}

__global__ __launch_bounds__(TB) void k_scatter_rows3(int nown, const uint32_t *__restrict__ own, const uint32_t *__restrict__ lg,
                                                      const float *__restrict__ in, float *__restrict__ out)
{
    int t = blockIdx.x * TB + threadIdx.x;
    if (t >= nown) return;
    const uint32_t i = own ? own[t] : (uint32_t)t;
    const size_t g = lg[i];
    for (int q = 0; q < 3; q++) out[3 * g + q] = in[3 * (size_t)i + q];
}

__global__ __launch_bounds__(TB) void k_gather_rho_vhf(int n, const uint32_t *__restrict__ lg, const float *__restrict__ grho,
                                                       const float *__restrict__ gvhf, float *__restrict__ rho,
                                                       float *__restrict__ vhf)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = lg[i];
    rho[i] = grho[g];
    vhf[i] = gvhf[g];
}

/* rho and varHsmlFac of the local particles (the curl reads those of the particle it solves) */
int tc_launch_gather_rho_vhf(tcgpu_ctx *c)
{
    const int n = (int)c->nloc;
    k_gather_rho_vhf<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->lg, c->g_rho[c->gcur], c->g_vhf[c->gcur], c->rho, c->vhf);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* A in local order; its first component also into the w lane of the local positions, with a flag raised when
 * the three components differ anywhere (then the curl reads l_apot instead) */
__global__ __launch_bounds__(TB) void k_gather_apot(int n, const uint32_t *__restrict__ lg, const float *__restrict__ in,
                                                    float *__restrict__ out, float4 *__restrict__ pos4, int *__restrict__ differ)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const size_t g = lg[i];
    const float a0 = in[3 * g], a1 = in[3 * g + 1], a2 = in[3 * g + 2];
    out[3 * (size_t)i] = a0; out[3 * (size_t)i + 1] = a1; out[3 * (size_t)i + 2] = a2;
    pos4[i].w = a0;
    if (!(a0 == a1 && a1 == a2)) atomicOr(differ, 1);
}

int tc_launch_gather_apot(tcgpu_ctx *c, int *equal_components)
{
    const int n = (int)c->nloc;
    TC_HIP(c, hipMemsetAsync(c->d_count + 2, 0, sizeof(int), c->stream));
    k_gather_apot<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->lg, c->apot, c->l_apot, c->pos4, c->d_count + 2);
    TC_HIP(c, hipGetLastError());
    int differ = 0;
    TC_HIP(c, hipMemcpyAsync(&differ, c->d_count + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    *equal_components = !differ;
    c->local_w_valid = 0;                     /* the w lane now holds A, not the WVT hsml */
    c->mirror_valid = 0;
    return 0;
}

int tc_launch_scatter_bfld(tcgpu_ctx *c, const float *l_bfld)
{
    const int n = (int)c->nown;
    if (n > 0)
        k_scatter_rows3<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->nranks > 1 ? c->own_list : nullptr, c->lg, l_bfld,
                                                                c->bfld);
    TC_HIP(c, hipGetLastError());
    return 0;
}
