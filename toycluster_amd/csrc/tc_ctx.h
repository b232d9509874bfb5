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

enum tc_phase {
    PH_KEYS = 0, PH_SORT, PH_PERMUTE, PH_CELLS, PH_GUESS, PH_DENSITY, PH_ERROR, PH_MODEL_HSML,
    PH_WVT, PH_MOVE, PH_CURL, PH_COMM, PH_MIRROR, PH_COUNT
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
    const uint2 *cells;           /* {~first, last+1} per cell; level L at offset tc_level_offset(L) */
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
    int n;                        /* all particles (neighbour candidates) */
    int lo, hi;                   /* [lo,hi): the particles this GPU solves for */
    int *work_ctr;                /* dynamic work queue: per XCD group (stride 16 ints) the next unassigned particle */
    int ablate;                   /* profiling only: 1 producer only, 2 +predicate, 3 no solver (results invalid) */
};

TC_HD size_t tc_level_offset(int L) /* cells of levels 1..L-1 */
{
    size_t o = 0, c = 8;
    for (int l = 1; l < L; l++) { o += c; c *= 8; }
    return o;
}

struct tc_event_rec { int phase; hipEvent_t a, b; };

struct tcgpu_ctx {
    int device;
    hipStream_t stream;
    char err[512];

    /* model */
    tcgpu_params par;
    int have_model;
    tc_halo_dev *d_halo;

    /* particles: two copies for the out-of-place permutation, `cur` is live */
    int64_t n, cap;
    int cur;
    float4 *pos4[2];
    int32_t *id[2];
    float *hsml[2], *rho[2], *vhf[2], *rhom[2];
    float *apot, *bfld;           /* 3*cap each, allocated on demand */

    /* sort */
    tc_u128 *key, *key_sorted;
    uint32_t *idx, *idx_sorted;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int keys_valid;               /* key_sorted matches the current order */

    /* neighbour index */
    int lmax, lmax_alloc;
    uint2 *cells;
    size_t ncells_alloc;
    double *spill;                /* TC_MAX_PERSISTENT_BLOCKS*WPB x 2*NGBMAX */
    double *ustep;                /* 3*cap: unit-step WVT displacement sums of the fused kernel */
    float *rhom_next;             /* cap: model density at the current positions, committed by the sweep */
    int ustep_valid;              /* ustep/rhom_next/hwvt belong to the current order and positions */
    int fuse;                     /* option: use the fused kernel (default 1) */
    int num_cu;
    uint32_t *orphans;
    int *norph;
    int *work_ctr;                /* 8 x 16 ints: per-XCD-group particle counters of the dynamic work queue */
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
    double level_scale;
    int lmax_override;
    int ablate;

    /* scratch */
    float *guess;
    float *hwvt, *delta;          /* cap, 3*cap */
    double *red;                  /* TC_RED_BLOCKS*4 partials + 32 finals */
    double *h_red;                /* pinned, 32 doubles */
    int *flags;                   /* device: [0]=nonfinite [1]=coord range [2]=no convergence [3]=overflow [4]=wvt ngbmax hits */
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

#define TC_HIP(ctx, call)                                                                       \
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
static inline const uint32_t *tc_cum_base(const tcgpu_ctx *c) { return c->cum - tc_level_offset(c->lmin_rm); }

/* ---- launchers implemented in the kernel translation units ---- */
int tc_sort_temp_bytes(size_t n, size_t *bytes);
int tc_sort_pairs_u128(void *tmp, size_t tmp_bytes, const tc_u128 *kin, tc_u128 *kout,
                       const uint32_t *vin, uint32_t *vout, size_t n, int sort_bits, hipStream_t s);

int tc_launch_keys(tcgpu_ctx *c);
int tc_launch_keys_xyz(tcgpu_ctx *c, int64_t n, const double *d_xyz, uint64_t *d_hi, uint64_t *d_lo);
int tc_launch_permute(tcgpu_ctx *c);
int tc_launch_cells(tcgpu_ctx *c);
int tc_scan_temp_bytes(size_t ncell, size_t *bytes);
int tc_launch_mirror(tcgpu_ctx *c);             /* cum, mirror, mirror_idx from cells + pos4 (after pos4.w is final) */
int tc_launch_guess(tcgpu_ctx *c);
int tc_launch_model(tcgpu_ctx *c, float *d_out);
int tc_launch_error(tcgpu_ctx *c);              /* -> red[0..2] = sum err, count, max err (over the shard) */
int tc_launch_model_hsml(tcgpu_ctx *c);         /* rhom, hwvt (normalised, also into pos4.w) */
int tc_launch_move(tcgpu_ctx *c);
int tc_launch_flags_to_f64(tcgpu_ctx *c, double *d_out);   /* flags[0..3] != 0 as doubles */
int tc_launch_commit_rhom(tcgpu_ctx *c);
int tc_launch_density(tcgpu_ctx *c);
int tc_launch_iter(tcgpu_ctx *c, int with_wvt);   /* fused density (+ unit-step WVT sums) */
int tc_launch_apply_step(tcgpu_ctx *c, double step);  /* delta = step * ustep, rhom <- rhom_next */
int tc_launch_wvt(tcgpu_ctx *c, double step);
int tc_launch_curl(tcgpu_ctx *c);
int tc_launch_find_ngb(tcgpu_ctx *c, int ipart, float hsml);
void tc_fill_const(const tcgpu_ctx *c, tc_dev_const *k);

void tc_phase_begin(tcgpu_ctx *c, int phase);
void tc_phase_end(tcgpu_ctx *c);
void tc_phase_collect(tcgpu_ctx *c);

#endif
