/* tc_ctx.h -- internal context of libtcgpu (not part of the ABI). */
#ifndef TC_CTX_H
#define TC_CTX_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/tcgpu.h"
#include "tc_math.h"

#define TC_MAX_LEVEL 10          /* deepest cell-table level (8^10 cells x 8 B = 8.6 GB) */
#define TC_MAX_ORPHANS 4096
#define TC_MAX_HALOS_DEV TC_MAXHALOS
#define TC_WAVES_PER_BLOCK 4
#define TC_RED_BLOCKS 1024       /* partial-sum slots of the streaming reductions */
#define TC_RCAP 768              /* hit-list entries kept in LDS (f64 each); the rest spill */
#define TC_IDXCAP 1024           /* candidate-index list per wave in LDS (u32 each) */
#define TC_STAGE 128             /* ring of staged hit positions per wave (power of two, >= 128) */
#define TC_SMALLCELL 64          /* cells up to this size are expanded one lane per cell */
#define TC_MAX_PERSISTENT_BLOCKS 2048
#define TC_XRCAP 96              /* ordered index runs per particle handed to k_iter (k_xruns; mean 29, max 78 at 2e6); more: the plain paths */
#define TC_XLCAP 448             /* sweep neighbours per particle handed to k_wvt_chain4; more: k_wvt_exact4 does that particle */
#define TC_XNONE 0xffffffffu     /* xlcnt / xrn: no list for this particle */

enum tc_phase {
    PH_KEYS = 0, PH_SORT, PH_PERMUTE, PH_CELLS, PH_GUESS, PH_DENSITY, PH_ERROR, PH_MODEL_HSML,
    PH_WVT, PH_MOVE, PH_CURL, PH_COMM, PH_MIRROR, PH_LOCAL, PH_PRESENT, PH_PREC, PH_COUNT
};

/* One level of the cell table: the dense (x, y, z) array, z fastest, of the cells [o, o + n) per dimension -- the
 * whole level on a full local set, on a sharded rank the bounding box of every cell a permitted query of this pass
 * can touch at that level (marking kernel).  Cell (x, y, z) lives at off + ((x - ox) ny + (y - oy)) nz + (z - oz);
 * a dimension whose box is the whole ring (n == 2^L) is addressed modulo 2^L (periodic queries). */
struct tc_level_desc {
    uint32_t off;                 /* first cell of the level in `cells` (and, minus the first mirrored level's, in `cum`) */
    int ox, oy, oz;
    int nx, ny, nz;
    int pad;
};

/* Constants every neighbour kernel needs; passed by value. */
struct tc_dev_const {
    double boxsize, boxhalf, mpart;
    double boxinv;                /* 1 / boxsize */
    double box_mant;              /* boxsize = box_mant * 2^box_exp, box_mant in [1, 2) */
    int box_exp;
    float boxsize_f, boxhalf_f;
    int lmax;                     /* deepest table level in use */
    int level_shift;              /* added to floor(log2(box/h))+1 when choosing the query level */
    double level_scale;           /* h is multiplied by this before the level is chosen (tuning) */
    const uint2 *cells;           /* {~first, last+1} per cell; level L laid out as lvl[L] says */
    const tc_level_desc *lvl;     /* [TC_MAX_LEVEL + 1] */
    const uint32_t *orphans;      /* particles with a coordinate == boxsize (X has bit 63) */
    const int *norph;
    const float4 *pos4;           /* x,y,z,(w = hsml_wvt) in Peano order */
    /* row-major mirror of the particles (levels 1..lmax_rm; NULL / 0 when not built): every cell of the
     * dense (x, y, z) table owns the slots [cum[o], cum[o+1]) of `mirror`, cells taken in table order, so a
     * run of consecutive z cells of one (x, y) row is ONE contiguous slot range */
    const uint32_t *cum;          /* exclusive prefix sum of the cell populations over the whole table (+1 entry) */
    const float4 *mirror;         /* positions (w = hsml_wvt) in slot order */
    const uint32_t *mirror_idx;   /* slot -> Peano index */
    int lmax_rm, lmin_rm;         /* mirrored levels: lmin_rm..lmax_rm */
    uint32_t mirror_pad;          /* slot holding a position at infinity (padding lanes load it) */
    uint32_t pos_pad;             /* ... and the like slot of pos4 (index cap) */
    int n;                        /* particles of the local set (neighbour candidates) */
    int lo, hi;                   /* work items [lo,hi): particle own[t] (own != NULL) or t itself */
    const uint32_t *own;          /* sharded contexts: local indices of the particles this GPU solves, ascending */
    int lmin_tab;                 /* coarsest level the cell table holds (queries are clamped to [lmin_tab, lmax]) */
    int margin_on;                /* sharded contexts: a query beyond tc_margin_radius() must not run (ghosts end there) */
    int margin_widen;             /* ... with this many extra x1.23 retries allowed (pass being repeated) */
    int *work_ctr;                /* dynamic work queue: per XCD group (stride 16 ints) the next unassigned particle */
    int ablate;                   /* profiling only: 1 producer only, 2 +predicate, 3 no solver (results invalid) */
};

TC_HD size_t tc_level_offset(int L) /* cells of levels 1..L-1 */
{
    size_t o = 0, c = 8;
    for (int l = 1; l < L; l++) { o += c; c *= 8; }
    return o;
}

#if defined(__HIPCC__)
/* table level of a query of radius h: floor(log2(box/(h*level_scale))) + 1 + level_shift, clamped to
 * [lmin, lmax]; floor(log2) from the exponents and mantissas of the two numbers (no division).  Any level is
 * CORRECT for any query (the cell enumeration adapts); the choice only decides how many cells and candidates a
 * query touches. */
__device__ __forceinline__ int tc_query_level(double boxsize, double box_mant, int box_exp, double level_scale,
                                              int level_shift, int lmin, int lmax, float h)
{
    const double hd = (double)h * level_scale;
    int L = 1;
    if (hd <= boxsize) {
        const int eh = __builtin_amdgcn_frexp_exp(hd) - 1;                 /* hd = mh * 2^eh, mh in [1, 2) */
        const double mh = 2 * __builtin_amdgcn_frexp_mant(hd);
        L = box_exp - eh - (box_mant < mh ? 1 : 0) + 1;
    }
    L += level_shift;
    if (L < lmin) L = lmin;
    if (L > lmax) L = lmax;
    return L;
}
#endif

#if defined(__HIPCC__)
/* Table levels the queries of one particle may use in a sharded pass: from the level of the margin radius (the widest
 * permitted query) to one below the level of the carried smoothing length (a query shrunk by the NGBMAX guard,
 * src/sph.c:42-47, may want finer cells; it gets these -- any level is correct).  The marking kernel sizes the
 * per-level boxes of the cell table from exactly these ranges, and the solver kernels clamp every query of the
 * particle into them, so no query can index outside a box. */
__device__ __forceinline__ void tc_particle_levels(double boxsize, double box_mant, int box_exp, double level_scale,
                                                   int level_shift, int lmax, float h0, float rg, int *la, int *lb)
{
    *la = tc_query_level(boxsize, box_mant, box_exp, level_scale, level_shift, 1, lmax, rg);
    int b = tc_query_level(boxsize, box_mant, box_exp, level_scale, level_shift, 1, lmax, h0) + 1;
    *lb = b > lmax ? lmax : b;
}
#endif

struct tc_event_rec { int phase; hipEvent_t a, b; };

struct tcgpu_ctx {
    int device;
    hipStream_t stream;
    char err[512];

    /* model */
    tcgpu_params par;
    int have_model;
    tc_halo_dev *d_halo;

    /* GLOBAL arrays, all n particles in the order G (upload order, then the Peano order of the last
     * "presentation", tc present() in api.hip).  Positions are complete on every rank; the per-particle state
     * (hsml, rho, ...) is valid for the rank's own index range between presentations.  Two copies for the
     * out-of-place permutation of a presentation, `gcur` is live. */
    int64_t n, cap;
    int gcur;
    float4 *g_pos4[2];            /* x, y, z, w = WVT hsml (box units) */
    int32_t *g_id[2];
    float *g_hsml[2], *g_rho[2], *g_vhf[2], *g_rhom[2];
    tc_u128 *g_key;               /* key of every particle at the last density pass (own range written per pass) */
    tc_u128 *g_key_sorted;        /* ... in presented order (tcgpu_download_keys) */
    int order_dirty;              /* a density pass has run since the last presentation */
    int g_compact;                /* G is a Peano order: index ranges are spatially compact shards */
    int w_valid;                  /* g_pos4.w / rhom_next / hwvt hold the model hsml of the current positions */
    float *apot, *bfld;           /* 3*cap each, allocated on demand (G order) */
    float *l_apot;                /* 3*cap, local order */

    /* LOCAL set: the particles this rank's queries can reach (own range + ghosts; everything on a single rank
     * and on cold passes), Peano-sorted every density pass */
    int64_t nloc;
    int local_full;               /* the local set is all n particles (then local order == presented order) */
    uint32_t *lsel;               /* unsorted local set: global indices (ascending) */
    uint32_t *lg;                 /* local sorted slot -> global index */
    uint32_t *own_list;           /* local sorted slots of the own range, ascending (sharded, !local_full) */
    int64_t nown;
    float4 *pos4;                 /* local sorted positions (w = WVT hsml) */
    float *hsml, *rho, *vhf;      /* local sorted: carried hsml in, results out */
    float *hsml0;                 /* local sorted: the carried hsml as gathered (kept through the pass) */
    int mark_ignore_w;            /* the marking of this local set used hsml only (curl) */
    uint32_t *imask;              /* interest pyramid: one bit per cell of levels 1..lp_max (tc_level_offset layout) */
    int lp_max;
    int *lvl_range;               /* device: [0] = min, [1] = max table level the own queries of this pass can use */
    int lmin_tab;                 /* coarsest level built this pass */
    tc_level_desc h_lvl[TC_MAX_LEVEL + 2];   /* layout of this pass's cell table (host copy) */
    tc_level_desc *d_lvl;
    int *d_bbox;                  /* device: per level {min x, y, z, max x, y, z} of the cells the own queries can touch */
    size_t ncells_used;           /* cells of this pass's table (cleared every pass) */
    void *sel_tmp;
    size_t sel_tmp_bytes;
    int *d_count;                 /* device: selected count */
    int margin_retry;             /* passes repeated because a query left the ghost margin (diagnostic) */
    int margin_widen;             /* extra x1.23 retries the current pass allows */
    uint32_t *isum;               /* 8^TC_LS bits: coarse summary of the interest pyramid (fast reject) */
    int local_w_valid;            /* the w lane of the local positions is the current model hsml */
    int lmax_rm0, lmin_rm0;       /* mirrored level range of a full local set (per pass: clipped to lmin_tab) */
    double comm_bytes;            /* bytes received in collectives since the last reset */
    /* ghost exchange (sharded contexts, <= TC_GHOST_MAXR ranks): every rank's pyramid side by side, then only the
     * particles inside a receiver's pyramid travel */
    uint32_t *pyr_all;            /* nranks x pyr_chunk words: rank q's pyramid (imask | isum) at q * pyr_chunk */
    size_t pyr_chunk, pyr_imask_words;
    int pos_all_valid;            /* g_pos4 holds the current position of EVERY particle (else: own range + last ghosts) */
    uint32_t *ghost_mask;         /* own range: destination ranks of each own particle (bit q) */
    int *ghost_blk_cnt, *ghost_blk_incl;   /* [dest][block] counts and their inclusive scan */
    int ghost_nblk;
    int *ghost_cnt_mat, *h_cnt_mat;        /* nranks x nranks send counts (row = sender), device and host */
    uint32_t *send_idx, *ghost_idx;        /* what this rank sends (grouped by destination) / receives (by sender) */
    float4 *send_pos, *ghost_pos;
    size_t send_cap, ghost_cap;
    int64_t nghost, nghost_lo;    /* ghosts received this pass; those from lower ranks (they precede the own range in lsel) */
    int ghost_mode;               /* option "ghost_exchange": 0 = position all-gather every pass, 1 = whichever moves fewer
                                   * bytes (default), 2 = always the ghost exchange */
    int ghost_pause;              /* passes left on the all-gather path before the ghost exchange is tried again */
    double h3_unit;               /* fixed-point unit (a power of two) of the exact sum of h^3, from the model (set_model) */

    /* sort (local set; also the scratch of a presentation) */
    tc_u128 *key, *key_sorted;
    uint32_t *idx, *idx_sorted;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int keys_valid;               /* g_key_sorted matches the presented order */

    /* neighbour index */
    int lmax, lmax_alloc;
    uint2 *cells;
    size_t ncells_alloc;
    double *spill;                /* TC_MAX_PERSISTENT_BLOCKS*WPB x 2*NGBMAX */
    double *ustep;                /* 3*cap: unit-step WVT displacement sums of the fused kernel (local order) */
    int no_records;               /* option "no_records" (tests): mark every per-particle record unusable */
    void *prec;                   /* 64 B per local slot: per-particle query records of the fused kernel (k_prec) / the curl, on demand */
    size_t prec_cap;              /* ... slots allocated (sized by the local set of the pass that needs them) */
    float *rhom_next;             /* cap: model density at the current positions, committed by the sweep (G order) */
    int ustep_valid;              /* ustep belongs to the current local order and positions */
    int fuse;                     /* option: use the fused kernel (default 1) */
    int sweep_mode;               /* option "sweep": 0 = the reference's f32 accumulation in ascending index, the neighbours listed
                                   * by k_iter on the way (default); 2 = the same sums by the stand-alone k_wvt_exact4;
                                   * 1 = round 2's f64 sums rounded once (fused into k_iter, or k_wvt) */
    int xsweep_shift;             /* option "xsweep_shift": added to the query level of k_wvt_exact (tuning) */
    int xsweep_kernel;            /* option "xsweep_kernel" (tests): 1 = the one-lane-per-particle kernel for every launch */
    uint32_t *pf;                 /* cell starts in curve order, levels pf_lmin..lmax (tc_launch_pfirst); entries carry a per-level bias */
    void *pf_tmp;
    size_t pf_alloc, pf_tmp_bytes;
    int lists_unfit;              /* the per-particle lists of the ordered gather did not fit in the last pass that wanted them */
    int pf_mode;                  /* option (tests): 0 automatic, 1 every level by the one scan, 2 the deep levels block by block */
    int pf_lmin, pf_lc, pf_valid; /* ... built for the current local order; levels <= pf_lc by one scan, deeper ones block by block */
    void *xr; uint32_t *xrn;      /* per-particle ordered run lists of the gather (k_xruns) and their lengths; on demand */
    uint32_t *xlist, *xlcnt;      /* per-particle sweep neighbours in index order, written by k_iter (WVT == 2) */
    uint32_t *xun;                /* particles k_iter could not list (+ their count at [xr_cap]) */
    size_t xr_cap;
    int xlist_valid;              /* ... belong to the current local order, positions and model hsml */
    void *xruns;                  /* per-wave index runs of k_wvt_exact, on demand */
    size_t xruns_bytes;
    int num_cu;
    int blocks_per_cu;            /* profiling only: cap on the co-resident blocks per CU of the persistent kernels (0 = all) */
    uint32_t *orphans;
    int *norph;
    int *work_ctr;                /* 8 x 16 ints: per-XCD-group particle counters of the dynamic work queue (+ a second set for the side stream) */
    hipStream_t stream2;          /* side stream: the few unlisted particles of the sweep run under the list evaluation */
    hipEvent_t ev_fork, ev_join;
    int index_valid;
    uint32_t *cum;                /* cells of levels lmin_rm..lmax_rm, +1 */
    float4 *mirror;               /* (lmax_rm - lmin_rm + 1) x cap slots, +1 pad */
    uint32_t *mirror_idx;
    void *scan_tmp;
    size_t scan_tmp_bytes, cum_alloc, mirror_alloc;
    int lmax_rm;                  /* deepest mirrored level (0: none) */
    int lmin_rm;                  /* coarsest mirrored level */
    int mirror_valid;
    int rows;                     /* option: use the row-run fast path (default 1) */
    int level_shift;
    double level_scale;           /* option; 0 = automatic (tc_level_scale) */
    int lmax_override;
    int ablate;
    int curl_literal;             /* option (tests): the curl's literal per-pair path for every particle */

    /* scratch */
    float *guess;
    float *hwvt, *delta;          /* cap, 3*cap (G order; own range) */
    double *red;                  /* TC_RED_BLOCKS*4 partials + 32 finals */
    double *h_red;                /* pinned, 32 doubles */
    int *flags;                   /* device: [0]=nonfinite [1]=coord range [2]=no convergence [3]=overflow [4]=wvt ngbmax hits
                                   * [5]=a query left the ghost margin (sharded contexts: the pass is repeated) */
    int *h_flags;                 /* pinned */
    uint32_t *stats;              /* 4 x cap, optional */
    int want_stats;
    tcgpu_density_stats last_stats;
    int32_t *ngb_buf;             /* cap, for tcgpu_find_ngb */
    int *ngb_cnt;

    /* sharding */
    int rank, nranks;
    void *comm;                   /* ncclComm_t */
    int force_comm;
    int debug_fail_rank;          /* option (tests): rank (value - 1) reports a failure of its own before the ghost exchange */
    struct tc_loop_comm *loop;    /* testing: in-process loopback communicator (threads + D2D copies) */
    int need_guess;               /* some hsml == 0: the first-pass guess is required */
    int64_t shard_len;

    /* phase timing */
    tc_event_rec *recs;
    int nrecs, caprecs;
    double ph_sec[PH_COUNT];
    int64_t ph_launch[PH_COUNT];
    int timing;
};

/* Cell size of a query relative to its radius (speed only, never results).  On the mirror's row-run path cells of
 * h/3.4..h/1.7 (2^(1/4)) were measured best (tools/shift_probe.py); the default sweep's ordered cell walk (k_xruns) pays
 * per cell, not per row, and likes them coarser: 1.5 gives 13.65 against 13.88 ms per iteration at 2e6, flat up to 1.7
 * (tools/shift_probe_ord.py).  One value per pass: records, interest marking and table ranges all use it.  When the
 * per-particle lists of that walk do not fit (1e8 particles on one GPU) the passes run on the mirror and the stand-alone
 * sweep, and the finer cells are the better ones again (measured at 1e8: 813 against 872 ms per iteration). */
static inline double tc_level_scale(const tcgpu_ctx *c)
{
    if (c->level_scale > 0) return c->level_scale;
    return (c->sweep_mode == 0 && !c->xsweep_kernel && !c->lists_unfit) ? 1.5 : 1.189207115002721;
}

/* Cell starts in curve order (tc_launch_pfirst): first local index whose level-L key prefix is >= p.  Levels lmin..lc are
 * dense tables made by one scan (entries carry the bias (L - lmin) (n + 1)); a deeper level is filled block by block --
 * only the 8^(L - lc) entries under OCCUPIED level-lc cells exist, an empty cell answers from its own level-lc entry. */
struct tc_pf {
    const uint32_t *tab;
    int lmin, lc;
};
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t tc_pf_offset(int lmin, int L)
{
    uint32_t off = 0;
    for (int t = lmin; t < L; t++) off += (1u << (3 * t)) + 1u;
    return off;
}
/* offL = tc_pf_offset(lmin, L), offC = tc_pf_offset(lmin, lc): hoisted by the caller */
__device__ __forceinline__ uint32_t tc_pf_first(const tc_pf &P, int L, uint32_t offL, uint32_t offC, uint32_t p, uint32_t n)
{
    if (L <= P.lc) return P.tab[offL + p] - (uint32_t)(L - P.lmin) * (n + 1u);
    if (p >= (1u << (3 * L))) return n;
    const uint32_t c = p >> (3 * (L - P.lc));
    const uint32_t biasC = (uint32_t)(P.lc - P.lmin) * (n + 1u);
    const uint32_t lo = P.tab[offC + c] - biasC, hi = P.tab[offC + c + 1u] - biasC;
    if (lo == hi) return lo;                       /* nothing under this level-lc cell: the next occupied one starts at lo */
    return P.tab[offL + p];
}
#endif

#define TC_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
                     hipGetErrorString(e_));                                                    \
            return TCGPU_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

#define TC_FAIL(ctx, code, ...)                                  \
    do {                                                         \
        snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__);   \
        return (code);                                           \
    } while (0)

/* `cum` holds the prefix sum of the mirrored levels only; kernels index it with whole-table cell offsets */
static inline const uint32_t *tc_cum_base(const tcgpu_ctx *c) { return c->cum - c->h_lvl[c->lmin_rm].off; }

/* Scalars of one pass: device doubles at tc_pass_scalars(c); api.hip all-reduces them over the ranks.
 *   [0..2]  sum of the density errors as three 40-bit limbs of an exact fixed-point sum (2^-40 units)
 *   [3]     particles counted                                                   -> [0..3] are SUMMED
 *   [4]     largest density error
 *   [5..9]  error flags as 0/1: nonfinite, coordinate range, no convergence, overflow, ghost margin
 *                                                                               -> [4..9] are MAXIMISED
 * and at +16: [0..2] limbs of the exact sum of h^3 of the model hsml pass (2^-20 units)  -> SUMMED
 * Exact integer sums make every all-reduced value independent of how the particles are split over ranks,
 * blocks and threads: a sharded run takes bit for bit the decisions of the single-GPU run. */
#define TC_PS_NSUM 4
#define TC_PS_NMAX 6
#define TC_PS_FLAGS 5
#define TC_PS_H3 16
static inline double *tc_pass_scalars(const tcgpu_ctx *c) { return c->red + 4 * TC_RED_BLOCKS; }

/* total of three (all-reduced) 40-bit limbs -> f64; exact integer recombination, one rounding */
TC_HD double tc_limbs_to_double(double l0, double l1, double l2, double unit)
{
    unsigned __int128 t = ((unsigned __int128)(uint64_t)l2 << 80) + ((unsigned __int128)(uint64_t)l1 << 40)
                          + (unsigned __int128)(uint64_t)l0;
    return ((double)(uint64_t)(t >> 64) * 18446744073709551616.0 + (double)(uint64_t)t) * unit;
}

#define TC_ERR_SCALE 68719476736.0        /* 2^36: fixed-point unit of the exact density-error sum */

/* deepest level of the interest pyramid (8^8 cells = 2 MB of bits) and the level of its coarse summary */
#define TC_LP_MAX 8
#define TC_LS 4

/* Largest radius the kernels may query for a particle with carried smoothing length h0 and WVT hsml w (box
 * units) before a sharded pass must be repeated with more ghosts: the reference's first query (h0), its retry at
 * 1.23 h0 (src/sph.c:49-54), one more retry, and the sweep's ball w*box (src/wvt_relax.c:135).  The marking
 * kernel and the solver kernels evaluate this very function, so the ghost set provably covers every query that
 * is allowed to run.  `widen` more retries are allowed when a pass had to be repeated (api.hip). */
TC_HD float tc_margin_radius(float h0, float w, double boxsize, int widen)
{
    const float hb = (float)((double)h0 * 1.23);
    float h3 = (float)((double)hb * 1.23);
    for (int q = 0; q < widen; q++) h3 = (float)((double)h3 * 1.23);       /* a repeated pass allows more retries */
    const float hw = (float)((double)w * boxsize);
    const float r = h3 > hw ? h3 : hw;
    return r * 1.000001f;
}

/* ---- launchers implemented in the kernel translation units ---- */
int tc_sort_temp_bytes(size_t n, size_t *bytes);
int tc_sort_pairs_u128(void *tmp, size_t tmp_bytes, const tc_u128 *kin, tc_u128 *kout,
                       const uint32_t *vin, uint32_t *vout, size_t n, int sort_bits, hipStream_t s);

int tc_launch_keys_xyz(tcgpu_ctx *c, int64_t n, const double *d_xyz, uint64_t *d_hi, uint64_t *d_lo);
/* local set */
int tc_select_temp_bytes(size_t n, size_t *bytes);
int tc_launch_mark_interest(tcgpu_ctx *c);      /* imask, lvl_range from the own range (g_hsml, g_pos4.w) */
int tc_select_local(tcgpu_ctx *c, int64_t *nloc);  /* lsel = particles inside the interest mask; synchronises */
#define TC_GHOST_MAXR 32
int tc_launch_ghost_count(tcgpu_ctx *c);        /* ghost_mask, per-destination counts -> ghost_cnt_mat row of this rank */
int tc_launch_ghost_fill(tcgpu_ctx *c);         /* send_idx / send_pos, grouped by destination, ascending index inside */
int tc_launch_ghost_scatter(tcgpu_ctx *c);      /* received positions -> g_pos4; lsel = lower ghosts, own range, upper ghosts */
int tc_finish_local_layout(tcgpu_ctx *c);       /* level range + bounding boxes of the marking -> cell table layout; synchronises */
int tc_layout_table(tcgpu_ctx *c, const int *bbox);   /* h_lvl / d_lvl for this pass (NULL: whole levels) */
int tc_launch_keys_local(tcgpu_ctx *c);         /* key, idx of the local set; g_key of the own range */
int tc_launch_gather_local(tcgpu_ctx *c);       /* lg, pos4, hsml in sorted local order; own_list */
int tc_launch_cells(tcgpu_ctx *c);
int tc_scan_temp_bytes(size_t ncell, size_t *bytes);
int tc_launch_mirror(tcgpu_ctx *c);             /* cum, mirror, mirror_idx from cells + pos4 */
int tc_launch_pfirst(tcgpu_ctx *c);             /* pf from the sorted keys of the local set */
int tc_launch_guess(tcgpu_ctx *c);
int tc_launch_scatter_results(tcgpu_ctx *c);    /* hsml, rho, vhf of the own particles -> G order */
/* global arrays */
int tc_launch_iota(tcgpu_ctx *c, uint32_t *dst, size_t n, uint32_t first);   /* dst[i] = first + i */
int tc_launch_scatter_rho(tcgpu_ctx *c);        /* rho of the own particles -> G order (the error sums read it) */
int tc_launch_refresh_w(tcgpu_ctx *c);          /* local pos4.w <- g_pos4.w */
int tc_launch_present_permute(tcgpu_ctx *c, const uint32_t *perm);   /* G <- G[perm]; flips gcur */
int tc_launch_model(tcgpu_ctx *c, float *d_out);
int tc_launch_error(tcgpu_ctx *c);              /* pass scalars [0..4] over the own range */
int tc_launch_model_hsml_own(tcgpu_ctx *c);     /* rhom_next, hwvt (not yet normalised), pass scalars +16 */
int tc_launch_scale_hsml_own(tcgpu_ctx *c);     /* normalise with the (all-reduced) sum: hwvt, g_pos4.w */
int tc_launch_move(tcgpu_ctx *c);
int tc_launch_flags_to_f64(tcgpu_ctx *c, double *d_out);   /* flags[0..3], flags[5] != 0 as five doubles */
int tc_launch_commit_rhom(tcgpu_ctx *c);
int tc_launch_gather_rho_vhf(tcgpu_ctx *c);     /* rho, vhf (local) = G values */
int tc_launch_gather_apot(tcgpu_ctx *c, int *equal_components);   /* l_apot[i] = apot[lg[i]]; pos4.w = A if Ax == Ay == Az everywhere */
int tc_launch_scatter_bfld(tcgpu_ctx *c, const float *l_bfld);
/* neighbour kernels */
int tc_launch_density(tcgpu_ctx *c);
int tc_launch_iter(tcgpu_ctx *c, int with_wvt);   /* fused density (+ unit-step WVT sums) */
int tc_ensure_prec(tcgpu_ctx *c);                 /* query records for this pass's local set */
int tc_ensure_xlists(tcgpu_ctx *c);               /* per-particle lists of the ordered gather for this pass's local set; != 0: no memory */
int tc_launch_apply_step(tcgpu_ctx *c, double step);  /* delta (G order) = step * ustep (local order) */
int tc_launch_wvt(tcgpu_ctx *c, double step);
int tc_launch_wvt_exact(tcgpu_ctx *c, double step);   /* delta in the reference's summation order and roundings */
int tc_launch_curl(tcgpu_ctx *c, float *l_bfld, int a_in_w);   /* a_in_w: pos4.w / mirror.w hold A (equal components) */
int tc_launch_find_ngb(tcgpu_ctx *c, int ipart, float hsml);
void tc_fill_const(const tcgpu_ctx *c, tc_dev_const *k);
/* the rank's own index range of G */
static inline void tc_own_range(const tcgpu_ctx *c, int64_t *lo, int64_t *hi)
{
    int64_t a = c->rank * c->shard_len, b = (c->rank + 1) * c->shard_len;
    if (b > c->n) b = c->n;
    if (a > b) a = b;
    *lo = a; *hi = b;
}

void tc_phase_begin(tcgpu_ctx *c, int phase);
void tc_phase_end(tcgpu_ctx *c);
void tc_phase_collect(tcgpu_ctx *c);

#endif
