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
#include <rocprim/rocprim.hpp>
#include "tc_ctx.h"

#define TB 256

/* ------------------------------------------------------------------ K1 keys */

__global__ __launch_bounds__(TB) void k_keys(const float4 *__restrict__ pos4, int n, double box,
                                             tc_u128 *__restrict__ key, uint32_t *__restrict__ idx,
                                             int *__restrict__ flags)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    float4 p = pos4[i];
    double x = (double)p.x / box, y = (double)p.y / box, z = (double)p.z / box;
    /* src/peano.c:130-132 Assert: coordinates must be inside [0,1] (NaN fails too) */
    if (!(x >= 0 && x <= 1 && y >= 0 && y <= 1 && z >= 0 && z <= 1)) {
        atomicOr(&flags[1], 1);
        x = y = z = 0;
        p.x = p.y = p.z = 0;
    }
    uint64_t hi, lo;
    tc_peano_key(p.x, p.y, p.z, box, &hi, &lo);
    key[i] = ((tc_u128)hi << 64) | lo;
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(TB) void k_keys_xyz(const double *__restrict__ xyz, int64_t n,
                                                 uint64_t *__restrict__ khi, uint64_t *__restrict__ klo)
{
    int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const double m = 9223372036854775808.0;
    uint64_t X[3];
    X[0] = (uint64_t)(xyz[3 * i + 1] * m);
    X[1] = (uint64_t)(xyz[3 * i + 2] * m);
    X[2] = (uint64_t)(xyz[3 * i + 0] * m);
    tc_hilbert_transpose(X);
    uint64_t hi, lo;
    tc_key_from_transpose(X, &hi, &lo);
    khi[i] = hi; klo[i] = lo;
}

int tc_launch_keys(tcgpu_ctx *c)
{
    int n = (int)c->n;
    tc_phase_begin(c, PH_KEYS);
    k_keys<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4[c->cur], n, c->par.boxsize, c->key, c->idx, c->flags);
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

/* ------------------------------------------------------------------ K3 permutation */

__global__ __launch_bounds__(TB) void k_permute(int n, const uint32_t *__restrict__ perm,
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

int tc_launch_permute(tcgpu_ctx *c)
{
    int n = (int)c->n, a = c->cur, b = 1 - c->cur;
    tc_phase_begin(c, PH_PERMUTE);
    k_permute<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(n, c->idx_sorted, c->pos4[a], c->pos4[b], c->id[a], c->id[b],
                                                       c->hsml[a], c->hsml[b], c->rho[a], c->rho[b],
                                                       c->vhf[a], c->vhf[b], c->rhom[a], c->rhom[b]);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    c->cur = b;
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

__global__ __launch_bounds__(TB) void k_cells(const float4 *__restrict__ pos4, int n, double box, int lmax,
                                              uint2 *__restrict__ cells,
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
    for (int L = lmax; L >= dmin; L--) {
        int sh = lmax - L;
        size_t nL = (size_t)1 << L;
        size_t lin = (((size_t)(ci[0] >> sh) * nL) + (ci[1] >> sh)) * nL + (ci[2] >> sh);
        size_t o = tc_level_offset(L) + lin;
        if (L >= dprev) atomicMax(&cells[o].x, ~(uint32_t)i);          /* max(~i) = ~min(i): zero-initialised */
        if (L >= dnext) atomicMax(&cells[o].y, (uint32_t)(i + 1));
    }
}

int tc_launch_cells(tcgpu_ctx *c)
{
    int n = (int)c->n;
    size_t ncell = tc_level_offset(c->lmax + 1);
    tc_phase_begin(c, PH_CELLS);
    TC_HIP(c, hipMemsetAsync(c->cells, 0, ncell * sizeof(uint2), c->stream));
    TC_HIP(c, hipMemsetAsync(c->norph, 0, sizeof(int), c->stream));
    k_cells<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4[c->cur], n, c->par.boxsize, c->lmax, c->cells,
                                                     c->orphans, c->norph, c->flags);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    c->index_valid = 1;
    c->mirror_valid = 0;
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
                                               const uint2 *__restrict__ cells, const uint32_t *__restrict__ cum,
                                               float4 *__restrict__ mirror, uint32_t *__restrict__ mirror_idx)
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
        const size_t nL = (size_t)1 << L;
        const size_t o = tc_level_offset(L) + (((size_t)(ci[0] >> sh) * nL) + (ci[1] >> sh)) * nL + (ci[2] >> sh);
        const uint32_t slot = cum[o] + ((uint32_t)i - ~cells[o].x);
        mirror[slot] = p;
        mirror_idx[slot] = (uint32_t)i;
    }
}

int tc_launch_mirror(tcgpu_ctx *c)
{
    c->mirror_valid = 0;
    if (!c->rows || c->lmax_rm <= 0 || !c->index_valid) return 0;
    const int n = (int)c->n;
    /* scan only the mirrored levels; `cum` is addressed with whole-table cell offsets through a pointer
     * shifted back by the offset of the first mirrored level (tc_cum_base) */
    const size_t first = tc_level_offset(c->lmin_rm);
    const size_t ncell = tc_level_offset(c->lmax_rm + 1) - first + 1;     /* +1: the end of the last cell */
    tc_phase_begin(c, PH_MIRROR);
    auto in = rocprim::make_transform_iterator((const uint2 *)c->cells + first, tc_cell_count());
    size_t b = c->scan_tmp_bytes;
    hipError_t e = rocprim::exclusive_scan(c->scan_tmp, b, in, c->cum, 0u, ncell, rocprim::plus<uint32_t>(), c->stream);
    if (e == hipSuccess)
        k_mirror<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4[c->cur], n, c->par.boxsize, c->lmax, c->lmin_rm, c->lmax_rm,
                                                         c->cells, tc_cum_base(c), c->mirror, c->mirror_idx);
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
    int n = (int)c->n;
    tc_phase_begin(c, PH_GUESS);
    k_guess<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->key_sorted, n, c->par.boxsize, c->guess);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ block reductions */

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

/* reduce up to 3 sums + 1 max per block into out[block*4 .. +3] (fixed order => deterministic) */
__device__ __forceinline__ void block_reduce4(double s0, double s1, double s2, double mx, double *out)
{
    __shared__ double sh[4][TB / 64];
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); mx = wave_max(mx);
    if (l == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, cc = 0, m = 0;
        for (int k = 0; k < TB / 64; k++) { a += sh[0][k]; b += sh[1][k]; cc += sh[2][k]; m = fmax(m, sh[3][k]); }
        out[0] = a; out[1] = b; out[2] = cc; out[3] = m;
    }
}

__global__ __launch_bounds__(TB) void k_final4(const double *__restrict__ part, int nblocks, double *__restrict__ fin)
{
    double s0 = 0, s1 = 0, s2 = 0, mx = 0;
    for (int b = threadIdx.x; b < nblocks; b += TB) {
        s0 += part[4 * b]; s1 += part[4 * b + 1]; s2 += part[4 * b + 2]; mx = fmax(mx, part[4 * b + 3]);
    }
    block_reduce4(s0, s1, s2, mx, fin);
}

/* the four error flags as 0/1 doubles, so that they can ride in the pass's all-reduce (api.hip) */
__global__ void k_flags_to_f64(const int *__restrict__ flags, double *__restrict__ out)
{
    if (threadIdx.x < 4) out[threadIdx.x] = flags[threadIdx.x] != 0 ? 1.0 : 0.0;
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
    k_model<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4[c->cur], n, c->par.boxsize * 0.5, c->d_halo,
                                                     c->par.nhalos, d_out);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K6 error sums */

/* src/wvt_relax.c:73-85 over [lo,hi) */
__global__ __launch_bounds__(TB) void k_error(const float4 *__restrict__ pos4, const float *__restrict__ rho_sph,
                                              int lo, int hi, double boxhalf,
                                              const tc_halo_dev *__restrict__ halo, int nhalos,
                                              double *__restrict__ part)
{
    double s = 0, cnt = 0, mx = 0;
    for (int i = lo + blockIdx.x * TB + threadIdx.x; i < hi; i += gridDim.x * TB) {
        float4 p = pos4[i];
        float rho = tc_density_model(p.x, p.y, p.z, boxhalf, halo, nhalos);
        float err = (float)(fabs((double)(rho_sph[i] - rho)) / (double)rho);
        mx = fmax((double)err, mx);
        s += (double)err;
        cnt += 1;
    }
    block_reduce4(s, cnt, 0, mx, part + 4 * blockIdx.x);
}

int tc_launch_error(tcgpu_ctx *c)
{
    int lo = (int)(c->rank * c->shard_len), hi = (int)((c->rank + 1) * c->shard_len);
    if (hi > c->n) hi = (int)c->n;
    if (lo > hi) lo = hi;
    int nb = TC_RED_BLOCKS;
    tc_phase_begin(c, PH_ERROR);
    k_error<<<nb, TB, 0, c->stream>>>(c->pos4[c->cur], c->rho[c->cur], lo, hi, c->par.boxsize * 0.5, c->d_halo,
                                      c->par.nhalos, c->red);
    k_final4<<<1, TB, 0, c->stream>>>(c->red, nb, c->red + 4 * TC_RED_BLOCKS);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

/* ------------------------------------------------------------------ K7/K8 model hsml */

/* src/wvt_relax.c:108-118 */
__global__ __launch_bounds__(TB) void k_model_hsml(const float4 *__restrict__ pos4, int n, double boxhalf,
                                                   double mpart, const tc_halo_dev *__restrict__ halo, int nhalos,
                                                   float *__restrict__ rhom, float *__restrict__ hw,
                                                   double *__restrict__ part)
{
    double s = 0;
    for (int i = blockIdx.x * TB + threadIdx.x; i < n; i += gridDim.x * TB) {
        float4 p = pos4[i];
        float rho = tc_density_model(p.x, p.y, p.z, boxhalf, halo, nhalos);
        rhom[i] = rho;
        float h = (float)pow(TC_DESNNGB * mpart / (double)rho / TC_FOURPITHIRD, 1. / 3.);
        hw[i] = h;
        s += (double)(h * h * h);
    }
    block_reduce4(s, 0, 0, 0, part + 4 * blockIdx.x);
}

/* src/wvt_relax.c:120-124; the normalised value also goes into pos4.w for the sweep's gathers */
__global__ __launch_bounds__(TB) void k_scale_hsml(float4 *__restrict__ pos4, int n, const double *__restrict__ fin,
                                                   float *__restrict__ hw)
{
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    float norm = (float)pow(TC_DESNNGB / fin[0] / TC_FOURPITHIRD, 1.0 / 3.0);
    float h = hw[i] * norm;
    hw[i] = h;
    pos4[i].w = h;
}

int tc_launch_model_hsml(tcgpu_ctx *c)
{
    int n = (int)c->n, nb = TC_RED_BLOCKS;
    double *fin = c->red + 4 * TC_RED_BLOCKS + 16;        /* [0..7] error pass, [8..15] flag agreement, [16..19] here */
    tc_phase_begin(c, PH_MODEL_HSML);
    /* Rho_Model goes to a side buffer: the reference stores it when the sweep runs (wvt_relax.c:113),
     * so it is committed by tc_launch_commit_rhom(), not on iterations that stop before the sweep */
    k_model_hsml<<<nb, TB, 0, c->stream>>>(c->pos4[c->cur], n, c->par.boxsize * 0.5, c->par.mpart_gas, c->d_halo,
                                           c->par.nhalos, c->rhom_next, c->hwvt, c->red);
    k_final4<<<1, TB, 0, c->stream>>>(c->red, nb, fin);
    k_scale_hsml<<<(n + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4[c->cur], n, fin, c->hwvt);
    c->mirror_valid = 0;                      /* pos4.w changed */
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    return 0;
}

int tc_launch_commit_rhom(tcgpu_ctx *c)
{
    TC_HIP(c, hipMemcpyAsync(c->rhom[c->cur], c->rhom_next, (size_t)c->n * sizeof(float), hipMemcpyDeviceToDevice,
                             c->stream));
    return 0;
}

/* delta = step * U for the fused kernel's unit-step sums (src/wvt_relax.c:167-169 with the scalar
 * step factored out of the neighbour sum) */
__global__ __launch_bounds__(TB) void k_apply_step(const double *__restrict__ ustep, float *__restrict__ delta,
                                                   int lo, int hi, double step)
{
    int i = lo + blockIdx.x * TB + threadIdx.x;
    if (i >= hi) return;
    for (int c = 0; c < 3; c++) delta[3 * (size_t)i + c] = (float)(step * ustep[3 * (size_t)i + c]);
}

int tc_launch_apply_step(tcgpu_ctx *c, double step)
{
    int lo = (int)(c->rank * c->shard_len), hi = (int)((c->rank + 1) * c->shard_len);
    if (hi > c->n) hi = (int)c->n;
    if (lo > hi) lo = hi;
    tc_phase_begin(c, PH_WVT);
    if (hi > lo)
        k_apply_step<<<(hi - lo + TB - 1) / TB, TB, 0, c->stream>>>(c->ustep, c->delta, lo, hi, step);
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
    int lo = (int)(c->rank * c->shard_len), hi = (int)((c->rank + 1) * c->shard_len);
    if (hi > c->n) hi = (int)c->n;
    if (lo > hi) lo = hi;
    tc_phase_begin(c, PH_MOVE);
    if (hi > lo)
        k_move<<<(hi - lo + TB - 1) / TB, TB, 0, c->stream>>>(c->pos4[c->cur], c->delta, lo, hi, c->par.boxsize);
    tc_phase_end(c);
    TC_HIP(c, hipGetLastError());
    c->keys_valid = 0;
    c->index_valid = 0; c->mirror_valid = 0;
    c->ustep_valid = 0;
    return 0;
}
