/*
 * api.hip -- C ABI of libtcgpu (include/tcgpu.h): context, buffers, and the host-side
 * control flow of the path (the loop of src/wvt_relax.c:61-218 and the driver part of
 * src/sph.c:13-17).  All device work is enqueued on the context's own HIP stream.
 *
 * Data model.  GLOBAL arrays hold all n particles in a fixed order G; positions are
 * complete on every rank (all-gathered after each move), per-particle state is valid for the rank's own index
 * range.  Every density pass builds a LOCAL set -- own range + the ghost shell its queries can reach; everything
 * on a single rank and on cold passes -- sorts it along the Peano curve and builds the cell table and mirror
 * over it; results go back to G by index.  G is re-sorted into Peano order only when somebody looks
 * ("presentation": downloads, the per-particle API calls), which is also what makes the shards compact.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <algorithm>
#include <rccl/rccl.h>
#include "tc_ctx.h"

static const char *PHASE_NAMES[PH_COUNT] = {"peano_keys", "radix_sort", "permute", "cell_index", "hsml_guess",
                                            "density", "error_sums", "model_hsml", "wvt_sweep", "move",
                                            "curl", "comm", "mirror", "local_set", "presentation", "query_records"};

struct rccl_api {
    void *h;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    ncclResult_t (*CommDestroy)(ncclComm_t);
};
static rccl_api g_rccl;

/* Test-only communicator: R contexts driven by R host threads of ONE process (any devices, also all
 * on the same GPU).  Collectives = pthread barrier + device-to-device copies, with exactly the
 * offsets and shard arithmetic of the RCCL path, so the sharded control flow can be verified on a
 * single-GPU box.  Never used by the product path (tcgpu_comm_init installs RCCL). */
struct tc_loop_comm {
    int nranks;
    int exclusive;             /* profiling: one rank computes at a time (token below), so that per-rank HIP-event
                                * times on the one shared GPU are those of a rank alone on its own GPU */
    pthread_mutex_t token;
    pthread_barrier_t bar;
    void *bufs[16];
    double red[16][16];
    const uint32_t *sidx[16];  /* ghost exchange: every rank's send buffers and the offsets of its destination groups */
    const float4 *spos[16];
    int soff[16][17];
};


/* exclusive loopback (profiling): a rank holds the token whenever it is not waiting in a collective */
struct tc_loop_guard {
    tcgpu_ctx *c;
    explicit tc_loop_guard(tcgpu_ctx *ctx);
    ~tc_loop_guard();
};
static void loop_wait(tcgpu_ctx *c);

/* ------------------------------------------------------------------ phase timing */

void tc_phase_begin(tcgpu_ctx *c, int phase)
{
    c->ph_launch[phase]++;
    if (!c->timing) return;
    if (c->nrecs == c->caprecs) {
        int nc = c->caprecs ? 2 * c->caprecs : 256;
        c->recs = (tc_event_rec *)realloc(c->recs, nc * sizeof(tc_event_rec));
        for (int i = c->caprecs; i < nc; i++) {
            hipEventCreate(&c->recs[i].a);
            hipEventCreate(&c->recs[i].b);
        }
        c->caprecs = nc;
    }
    c->recs[c->nrecs].phase = phase;
    hipEventRecord(c->recs[c->nrecs].a, c->stream);
}

void tc_phase_end(tcgpu_ctx *c)
{
    if (!c->timing) return;
    hipEventRecord(c->recs[c->nrecs].b, c->stream);
    c->nrecs++;
}

/* call only after the stream has been synchronised */
void tc_phase_collect(tcgpu_ctx *c)
{
    for (int i = 0; i < c->nrecs; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->recs[i].a, c->recs[i].b) == hipSuccess) c->ph_sec[c->recs[i].phase] += ms * 1e-3;
    }
    c->nrecs = 0;
}

/* ------------------------------------------------------------------ life cycle */

#ifdef TCGPU_SPH_CUBIC_SPLINE
extern "C" const char *tcgpu_version(void) { return "tcgpu 0.2 (gfx950, -DSPH_CUBIC_SPLINE variant)"; }
#else
extern "C" const char *tcgpu_version(void) { return "tcgpu 0.2 (gfx950)"; }
#endif

extern "C" const char *tcgpu_last_error(const tcgpu_ctx *ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int tcgpu_create(tcgpu_ctx **out, int device)
{
    if (!out) return TCGPU_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TCGPU_ERR_HIP;
    if (device < 0 || device >= ndev) return TCGPU_ERR_ARG;
    tcgpu_ctx *c = (tcgpu_ctx *)calloc(1, sizeof(tcgpu_ctx));
    if (!c) return TCGPU_ERR_NOMEM;
    c->device = device;
    c->nranks = 1;
    c->timing = 0;                               /* option "timing": HIP events per phase, off in the product path */
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        free(c);
        return TCGPU_ERR_HIP;
    }
    bool ok = hipMalloc(&c->d_halo, sizeof(tc_halo_dev) * TC_MAX_HALOS_DEV) == hipSuccess;
    ok = ok && hipMalloc(&c->red, sizeof(double) * (4 * TC_RED_BLOCKS + 48)) == hipSuccess;
    ok = ok && hipHostMalloc(&c->h_red, sizeof(double) * 48) == hipSuccess;
    ok = ok && hipMalloc(&c->flags, sizeof(int) * 8) == hipSuccess;
    ok = ok && hipHostMalloc(&c->h_flags, sizeof(int) * 8) == hipSuccess;
    ok = ok && hipMalloc(&c->orphans, sizeof(uint32_t) * TC_MAX_ORPHANS) == hipSuccess;
    ok = ok && hipMalloc(&c->norph, sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->work_ctr, 2 * 8 * 16 * sizeof(int)) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&c->ngb_cnt, sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->lvl_range, 8 * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->d_lvl, sizeof(tc_level_desc) * (TC_MAX_LEVEL + 2)) == hipSuccess;
    ok = ok && hipMalloc(&c->d_bbox, sizeof(int) * 6 * (TC_MAX_LEVEL + 1)) == hipSuccess;
    ok = ok && hipMalloc(&c->d_count, 4 * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->spill, sizeof(double) * (size_t)TC_MAX_PERSISTENT_BLOCKS * TC_WAVES_PER_BLOCK
                                        * (2 * TC_NGBMAX)) == hipSuccess;
    c->fuse = 1;
    c->rows = 1;
    c->ghost_mode = 1;
    c->level_shift = 1;                          /* cells of h/4..h/2: fewest candidates per query (tools/fuse_stats.py) */
    c->level_scale = 0;                          /* automatic: tc_level_scale() */
    ok = ok && hipDeviceGetAttribute(&c->num_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess;
    ok = ok && hipMemset(c->flags, 0, sizeof(int) * 8) == hipSuccess;
    ok = ok && hipMemset(c->norph, 0, sizeof(int)) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (!ok) { tcgpu_destroy(c); return TCGPU_ERR_NOMEM; }
    *out = c;
    return TCGPU_OK;
}

#define TC_FREE(p) do { hipFree(p); (p) = nullptr; } while (0)

static void free_particles(tcgpu_ctx *c)
{
    for (int b = 0; b < 2; b++) {
        TC_FREE(c->g_pos4[b]); TC_FREE(c->g_id[b]); TC_FREE(c->g_hsml[b]); TC_FREE(c->g_rho[b]);
        TC_FREE(c->g_vhf[b]); TC_FREE(c->g_rhom[b]);
    }
    TC_FREE(c->g_key); TC_FREE(c->g_key_sorted);
    TC_FREE(c->lsel); TC_FREE(c->lg); TC_FREE(c->own_list); TC_FREE(c->pos4); TC_FREE(c->hsml); TC_FREE(c->hsml0); TC_FREE(c->rho);
    TC_FREE(c->vhf); TC_FREE(c->sel_tmp); TC_FREE(c->ghost_mask); TC_FREE(c->ghost_blk_cnt); TC_FREE(c->ghost_blk_incl);
    TC_FREE(c->apot); TC_FREE(c->bfld); TC_FREE(c->l_apot);
    TC_FREE(c->key); TC_FREE(c->key_sorted); TC_FREE(c->idx); TC_FREE(c->idx_sorted); TC_FREE(c->sort_tmp);
    TC_FREE(c->cells); TC_FREE(c->guess); TC_FREE(c->hwvt); TC_FREE(c->delta); TC_FREE(c->stats); TC_FREE(c->ngb_buf);
    TC_FREE(c->ustep); TC_FREE(c->rhom_next); TC_FREE(c->prec); c->prec_cap = 0; TC_FREE(c->xruns); c->xruns_bytes = 0;
    TC_FREE(c->xr); TC_FREE(c->xrn); TC_FREE(c->xlist); TC_FREE(c->xlcnt); TC_FREE(c->xun); c->xr_cap = 0; c->xlist_valid = 0;
    TC_FREE(c->cum); TC_FREE(c->scan_tmp); TC_FREE(c->mirror); TC_FREE(c->mirror_idx);
    TC_FREE(c->pf); TC_FREE(c->pf_tmp); c->pf_alloc = 0; c->pf_valid = 0;
    c->cum_alloc = c->mirror_alloc = 0; c->mirror_valid = 0;
    c->cap = 0; c->n = 0; c->nloc = 0; c->nown = 0; c->ncells_alloc = 0;
}

extern "C" void tcgpu_destroy(tcgpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_particles(c);
    hipFree(c->d_halo); hipFree(c->red); hipHostFree(c->h_red); hipFree(c->flags); hipHostFree(c->h_flags);
    hipFree(c->orphans); hipFree(c->norph); hipFree(c->work_ctr); hipFree(c->ngb_cnt); hipFree(c->spill);
    hipFree(c->lvl_range); hipFree(c->d_lvl); hipFree(c->d_bbox); hipFree(c->d_count); hipFree(c->pyr_all); hipFree(c->ghost_cnt_mat); free(c->h_cnt_mat);
    hipFree(c->send_idx); hipFree(c->send_pos); hipFree(c->ghost_idx); hipFree(c->ghost_pos);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->comm);
    for (int i = 0; i < c->caprecs; i++) { hipEventDestroy(c->recs[i].a); hipEventDestroy(c->recs[i].b); }
    free(c->recs);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream) hipStreamDestroy(c->stream);
    free(c);
}

extern "C" void *tcgpu_stream(tcgpu_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int64_t tcgpu_num_particles(const tcgpu_ctx *c) { return c ? c->n : 0; }

/* ------------------------------------------------------------------ model + particles */

extern "C" int tcgpu_set_model(tcgpu_ctx *c, const tcgpu_params *par, const tcgpu_halo *halos)
{
    if (!c || !par || (par->nhalos > 0 && !halos)) return TCGPU_ERR_ARG;
    if (par->nhalos < 0 || par->nhalos > TC_MAX_HALOS_DEV) TC_FAIL(c, TCGPU_ERR_ARG, "nhalos %d out of range", par->nhalos);
    if (!(par->boxsize > 0) || !(par->mpart_gas > 0)) TC_FAIL(c, TCGPU_ERR_ARG, "boxsize and mpart_gas must be > 0");
    TC_HIP(c, hipSetDevice(c->device));
    c->par = *par;
    tc_halo_dev *h = (tc_halo_dev *)malloc(sizeof(tc_halo_dev) * (par->nhalos ? par->nhalos : 1));
    for (int i = 0; i < par->nhalos; i++) {
        h[i].cx = halos[i].d_com[0]; h[i].cy = halos[i].d_com[1]; h[i].cz = halos[i].d_com[2];
        h[i].rho0 = halos[i].rho0; h[i].beta = halos[i].beta; h[i].rcore = halos[i].rcore; h[i].rcut = halos[i].rcut;
        h[i].mass_gas = halos[i].mass_gas;
        h[i].rho0_cc = halos[i].rho0_cc; h[i].rc_cc = halos[i].rc_cc;
        if (h[i].rho0_cc != 0 && !(h[i].rc_cc > 0)) { free(h); TC_FAIL(c, TCGPU_ERR_ARG, "halo %d: rho0_cc set but rc_cc <= 0", i); }
    }
    hipError_t e = hipMemcpy(c->d_halo, h, sizeof(tc_halo_dev) * par->nhalos, hipMemcpyHostToDevice);
    free(h);
    TC_HIP(c, e);
    c->have_model = 1;
    c->w_valid = 0;
    /* unit of the exact sum of h^3 (k_model_hsml): the model density is nowhere above the sum of the central
     * densities -- the cool-core component of a cuspy halo included (rho0 + rho0_cc at r = 0, setup.c:604-612) --
     * so h^3 = 295 m / rho / (4 pi / 3) is nowhere below h3_min; 2^-30 of that keeps 30 bits under
     * the smallest term and 33 bits of head room above it in a 64-bit term. */
    double rho_up = 0;
    for (int i = 0; i < par->nhalos; i++)
        if (halos[i].mass_gas != 0 && halos[i].rho0 > 0) rho_up += halos[i].rho0 + (halos[i].rho0_cc > 0 ? halos[i].rho0_cc : 0.0);
    int e2 = 0;
    if (rho_up > 0) (void)frexp(TC_DESNNGB * par->mpart_gas / rho_up / TC_FOURPITHIRD, &e2);
    c->h3_unit = ldexp(1.0, e2 - 1 - 30);
    return TCGPU_OK;
}

static int pick_lmax(int64_t n)
{
    /* Deepest table level: one below the mean inter-particle level, round(log8 n) + 1 (8 at N = 2e6, 9 at
     * 1.6e7, 10 at 1e8).  The densest few per cent of the particles would like one level more; clamped to
     * this one they see about twice the candidates, which costs less than clearing, scanning and mirroring a
     * table eight times larger every iteration (measured at N = 2e6: -0.33 ms per iteration, tools/shift_probe.py).
     * Chosen from the TOTAL particle number on every rank: the cell decomposition -- hence the order in which a
     * particle's neighbours are summed -- is then the same on one GPU and on eight. */
    int l = (int)floor(log((double)(n > 1 ? n : 1)) / log(8.0) + 0.5) + 1;
    if (l < 3) l = 3;
    if (l > TC_MAX_LEVEL) l = TC_MAX_LEVEL;
    return l;
}

static int ensure_capacity(tcgpu_ctx *c, int64_t n)
{
    int64_t need = n;
    if (c->comm || c->loop) {                       /* all-gather needs nranks equal shards */
        int64_t s = (n + c->nranks - 1) / c->nranks;
        need = s * c->nranks;
    }
    if (need > c->cap) {
        free_particles(c);
        size_t cap = (size_t)need;
        for (int b = 0; b < 2; b++) {
            TC_HIP(c, hipMalloc(&c->g_pos4[b], cap * sizeof(float4)));
            TC_HIP(c, hipMalloc(&c->g_id[b], cap * sizeof(int32_t)));
            TC_HIP(c, hipMalloc(&c->g_hsml[b], cap * sizeof(float)));
            TC_HIP(c, hipMalloc(&c->g_rho[b], cap * sizeof(float)));
            TC_HIP(c, hipMalloc(&c->g_vhf[b], cap * sizeof(float)));
            TC_HIP(c, hipMalloc(&c->g_rhom[b], cap * sizeof(float)));
        }
        TC_HIP(c, hipMalloc(&c->g_key, cap * sizeof(tc_u128)));
        TC_HIP(c, hipMalloc(&c->g_key_sorted, cap * sizeof(tc_u128)));
        /* local set: sized for the full set (cold passes and single-rank contexts work on all particles) */
        TC_HIP(c, hipMalloc(&c->lsel, cap * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->lg, cap * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->own_list, cap * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->pos4, (cap + 1) * sizeof(float4)));
        {   /* one extra slot infinitely far away (index cap): padding lanes of the ordered gather load it */
            const float inf = HUGE_VALF;
            const float4 far = make_float4(inf, inf, inf, 0.0f);
            TC_HIP(c, hipMemcpy(c->pos4 + cap, &far, sizeof(far), hipMemcpyHostToDevice));
        }
        TC_HIP(c, hipMalloc(&c->hsml, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->hsml0, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->rho, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->vhf, cap * sizeof(float)));
        if (tc_select_temp_bytes(cap, &c->sel_tmp_bytes)) TC_FAIL(c, TCGPU_ERR_HIP, "select temp query failed");
        TC_HIP(c, hipMalloc(&c->sel_tmp, c->sel_tmp_bytes ? c->sel_tmp_bytes : 16));
        TC_HIP(c, hipMalloc(&c->key, cap * sizeof(tc_u128)));
        TC_HIP(c, hipMalloc(&c->key_sorted, cap * sizeof(tc_u128)));
        TC_HIP(c, hipMalloc(&c->idx, cap * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->idx_sorted, cap * sizeof(uint32_t)));
        if (tc_sort_temp_bytes(cap, &c->sort_tmp_bytes)) TC_FAIL(c, TCGPU_ERR_HIP, "radix sort temp query failed");
        TC_HIP(c, hipMalloc(&c->sort_tmp, c->sort_tmp_bytes ? c->sort_tmp_bytes : 16));
        TC_HIP(c, hipMalloc(&c->guess, 3 * cap * sizeof(float)));     /* also 3 floats/particle of scratch (presentation) */
        TC_HIP(c, hipMalloc(&c->hwvt, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->delta, 3 * cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->rhom_next, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->ngb_buf, cap * sizeof(int32_t)));
        TC_HIP(c, hipMemset(c->hwvt, 0, cap * sizeof(float)));
        TC_HIP(c, hipMemset(c->delta, 0, 3 * cap * sizeof(float)));
        TC_HIP(c, hipMemset(c->rhom_next, 0, cap * sizeof(float)));
        c->cap = need;
    }
    if (!c->pyr_all) {
        /* one pyramid (+ summary) per rank, side by side: a rank marks its own chunk, the ghost exchange all-gathers */
        c->pyr_imask_words = ((tc_level_offset(TC_LP_MAX + 1) / 32 + 1) + 3) & ~(size_t)3;
        c->pyr_chunk = c->pyr_imask_words + ((size_t)1 << (3 * TC_LS)) / 32;
        TC_HIP(c, hipMalloc(&c->pyr_all, (size_t)c->nranks * c->pyr_chunk * sizeof(uint32_t)));
        c->imask = c->pyr_all + (size_t)c->rank * c->pyr_chunk;
        c->isum = c->imask + c->pyr_imask_words;
        TC_HIP(c, hipMalloc(&c->ghost_cnt_mat, (size_t)c->nranks * c->nranks * sizeof(int)));
        c->h_cnt_mat = (int *)calloc((size_t)c->nranks * c->nranks, sizeof(int));
        if (!c->h_cnt_mat) return TCGPU_ERR_NOMEM;
    }
    if ((c->comm || c->loop) && c->nranks <= TC_GHOST_MAXR && !c->ghost_mask) {
        const size_t slen = (size_t)(need / c->nranks);
        c->ghost_nblk = (int)((slen + 255) / 256) + 1;
        TC_HIP(c, hipMalloc(&c->ghost_mask, (slen + 1) * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->ghost_blk_cnt, (size_t)c->nranks * c->ghost_nblk * sizeof(int)));
        TC_HIP(c, hipMalloc(&c->ghost_blk_incl, (size_t)c->nranks * c->ghost_nblk * sizeof(int)));
    }
    int lmax = c->lmax_override > 0 ? c->lmax_override : pick_lmax(n);
    if (lmax > TC_MAX_LEVEL) lmax = TC_MAX_LEVEL;
    size_t ncell = tc_level_offset(lmax + 1);
    if (ncell > c->ncells_alloc) {
        hipFree(c->cells);
        c->cells = nullptr;
        TC_HIP(c, hipMalloc(&c->cells, ncell * sizeof(uint2)));
        c->ncells_alloc = ncell;
    }
    c->lmax = lmax;
    c->lp_max = lmax < TC_LP_MAX ? lmax : TC_LP_MAX;
    /* row-major mirror: every level whose scan (12 B per cell) is cheap next to the per-particle saving;
     * queries at a deeper level use the cell-by-cell path */
    int lmax_rm = lmax;
    if ((double)n < 0.03 * pow(8.0, (double)lmax)) lmax_rm = lmax - 1;
    if (!c->rows) lmax_rm = 0;
    /* the coarsest levels are queried by next to no particle (balls wider than 1/16 of the box): not worth a
     * copy of every particle per level; those few queries take the cell-by-cell path */
    int lmin_rm = lmax_rm > 5 ? lmax_rm - 4 : 1;
    if (lmax_rm > 0 && (double)(lmax_rm - lmin_rm + 1) * (double)(c->cap + 1) >= 4.0e9) {   /* slots are 32-bit */
        fprintf(stderr, "tcgpu: notice: %lld particles x %d mirrored levels exceed the 32-bit slot range; "
                        "row-run fast path disabled (cell-by-cell path only)\n", (long long)n, lmax_rm - lmin_rm + 1);
        lmax_rm = 0;
    }
    c->lmax_rm0 = lmax_rm;
    c->lmin_rm0 = lmin_rm;
    c->lmax_rm = lmax_rm;
    c->lmin_rm = lmin_rm;
    c->mirror_valid = 0;
    if (lmax_rm > 0) {
        /* only the mirrored levels lmin_rm..lmax_rm are scanned and own slots */
        size_t ncum = tc_level_offset(lmax_rm + 1) - tc_level_offset(lmin_rm) + 1;
        size_t nslot = (size_t)(lmax_rm - lmin_rm + 1) * (size_t)c->cap;
        if (ncum > c->cum_alloc) {
            hipFree(c->cum); hipFree(c->scan_tmp);
            c->cum = nullptr; c->scan_tmp = nullptr; c->cum_alloc = 0;
            TC_HIP(c, hipMalloc(&c->cum, ncum * sizeof(uint32_t)));
            if (tc_scan_temp_bytes(ncum, &c->scan_tmp_bytes)) TC_FAIL(c, TCGPU_ERR_HIP, "scan temp query failed");
            TC_HIP(c, hipMalloc(&c->scan_tmp, c->scan_tmp_bytes ? c->scan_tmp_bytes : 16));
            c->cum_alloc = ncum;
        }
        (void)nslot;         /* the mirror itself is allocated by the pass that builds it, for its local set (tc_launch_mirror) */
    }
    return 0;
}

static void set_shard(tcgpu_ctx *c)
{
    c->shard_len = (c->n + c->nranks - 1) / c->nranks;
    if (c->shard_len < 1) c->shard_len = 1;
}

extern "C" int tcgpu_upload_particles(tcgpu_ctx *c, int64_t n, const float *pos, const int32_t *id, const float *hsml)
{
    tc_loop_guard guard_(c);
    if (!c || !pos || n <= 0) return TCGPU_ERR_ARG;
    if (n >= (1LL << 31) - 64) TC_FAIL(c, TCGPU_ERR_ARG, "n=%lld exceeds the 32-bit index range of the path", (long long)n);
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    int rc = ensure_capacity(c, n);
    if (rc) return rc;
    c->n = n;
    c->gcur = 0;
    set_shard(c);
    size_t cap = (size_t)c->cap;
    /* pack xyz -> float4 on the host (w = 0) */
    float4 *tmp = (float4 *)malloc(cap * sizeof(float4));
    int32_t *tid = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!tmp || !tid) { free(tmp); free(tid); return TCGPU_ERR_NOMEM; }
    for (size_t i = 0; i < (size_t)n; i++) {
        tmp[i] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 0.f);
        tid[i] = id ? id[i] : (int32_t)(i + 1);
    }
    for (size_t i = n; i < cap; i++) { tmp[i] = make_float4(0, 0, 0, 0); tid[i] = 0; }
    hipError_t e1 = hipMemcpyAsync(c->g_pos4[0], tmp, cap * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(c->g_id[0], tid, cap * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    free(tmp); free(tid);
    TC_HIP(c, e1); TC_HIP(c, e2); TC_HIP(c, e3);
    TC_HIP(c, hipMemsetAsync(c->g_hsml[0], 0, cap * sizeof(float), c->stream));
    if (hsml) TC_HIP(c, hipMemcpyAsync(c->g_hsml[0], hsml, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TC_HIP(c, hipMemsetAsync(c->g_rho[0], 0, cap * sizeof(float), c->stream));
    TC_HIP(c, hipMemsetAsync(c->g_vhf[0], 0, cap * sizeof(float), c->stream));
    TC_HIP(c, hipMemsetAsync(c->g_rhom[0], 0, cap * sizeof(float), c->stream));
    TC_HIP(c, hipMemsetAsync(c->g_key, 0, cap * sizeof(tc_u128), c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    c->keys_valid = 0;
    c->index_valid = 0; c->mirror_valid = 0;
    c->ustep_valid = 0;
    c->order_dirty = 0;
    c->g_compact = 0;
    c->w_valid = 0;
    c->pos_all_valid = 1;                         /* every rank was given every position */
    c->nloc = 0; c->nown = 0; c->local_full = 0;
    c->lists_unfit = 0;
    c->need_guess = 1;
    if (hsml) {                                   /* warm start: the guess is only read where hsml == 0 */
        c->need_guess = 0;
        for (int64_t i = 0; i < n; i++)
            if (hsml[i] == 0) { c->need_guess = 1; break; }
    }
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ flags */

#define TC_RETRY_FULL 1000      /* internal: a query left the ghost margin, repeat the pass on the full set */

/* `reduced`: the five flags already maximised over the ranks (multi-rank contexts: every rank must take
 * the same exit, or the ranks that carry on would wait forever in the next collective); NULL = this rank's own. */
static int check_flags(tcgpu_ctx *c, const double *reduced = nullptr)
{
    TC_HIP(c, hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 8, hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    int f[8];
    memcpy(f, c->h_flags, sizeof(f));
    if (f[0] || f[1] || f[2] || f[3] || f[5]) {
        TC_HIP(c, hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
        TC_HIP(c, hipMemsetAsync(c->flags + 5, 0, sizeof(int), c->stream));
    }
    int margin = f[5] != 0;
    if (reduced) {
        for (int q = 0; q < 4; q++) f[q] = reduced[q] != 0;
        margin = reduced[4] != 0;
    }
    if (f[1]) TC_FAIL(c, TCGPU_ERR_COORD_RANGE, "coordinate outside [0,boxsize] (reference: peano.c:130-132 Assert)");
    if (f[0]) TC_FAIL(c, TCGPU_ERR_NONFINITE, "hsml not finite (reference: sph.c:28 Assert)");
    if (f[3]) TC_FAIL(c, TCGPU_ERR_OVERFLOW, "more than %d particles sit exactly on the upper box face", TC_MAX_ORPHANS);
    if (margin) return TC_RETRY_FULL;
    if (f[2]) TC_FAIL(c, TCGPU_ERR_NO_CONVERGENCE, "hsml iteration did not terminate for some particle");
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ RCCL (loaded on demand) */

static int load_rccl(void)
{
    if (g_rccl.h) return 0;
    /* Prefer an RCCL that is already mapped into the process (e.g. the copy PyTorch links): two RCCL
     * builds side by side each keep their own device state.  Otherwise load ROCm's. */
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (int i = 0; i < 2 && !h; i++) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 1; i >= 0 && !h; i--) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    g_rccl.GetUniqueId = (ncclResult_t(*)(ncclUniqueId *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (ncclResult_t(*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (ncclResult_t(*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (ncclResult_t(*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.Send = (ncclResult_t(*)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclSend");
    g_rccl.Recv = (ncclResult_t(*)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclRecv");
    g_rccl.GroupStart = (ncclResult_t(*)(void))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (ncclResult_t(*)(void))dlsym(h, "ncclGroupEnd");
    g_rccl.CommDestroy = (ncclResult_t(*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.GroupStart
        || !g_rccl.GroupEnd || !g_rccl.Send || !g_rccl.Recv) {
        dlclose(h);
        return -1;
    }
    g_rccl.h = h;
    return 0;
}

extern "C" int tcgpu_comm_unique_id(uint8_t id[128])
{
    if (!id || load_rccl()) return TCGPU_ERR_COMM;
    ncclUniqueId u;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return TCGPU_ERR_COMM;
    memcpy(id, &u, 128);
    return TCGPU_OK;
}

extern "C" int tcgpu_comm_init(tcgpu_ctx *c, int rank, int nranks, const uint8_t id[128])
{
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return TCGPU_ERR_ARG;
    if (c->n > 0) TC_FAIL(c, TCGPU_ERR_ARG, "tcgpu_comm_init must precede tcgpu_upload_particles");
    if (nranks == 1 && !c->force_comm) { c->rank = 0; c->nranks = 1; return TCGPU_OK; }
    if (load_rccl()) TC_FAIL(c, TCGPU_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    TC_HIP(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm;
    if (g_rccl.CommInitRank(&comm, nranks, u, rank) != ncclSuccess) TC_FAIL(c, TCGPU_ERR_COMM, "ncclCommInitRank failed");
    c->comm = comm;
    c->rank = rank;
    c->nranks = nranks;
    return TCGPU_OK;
}

/* In-place all-gather of one shard-partitioned array of G (elements of `esize` bytes). */
/* loopback barrier; in exclusive mode the token is handed on while waiting */
static void loop_wait(tcgpu_ctx *c)
{
    tc_loop_comm *L = c->loop;
    if (L->exclusive) pthread_mutex_unlock(&L->token);
    pthread_barrier_wait(&L->bar);
    if (L->exclusive) pthread_mutex_lock(&L->token);
}

static int allgather_chunks(tcgpu_ctx *c, void *base, size_t bytes);

static int allgather_inplace(tcgpu_ctx *c, void *base, size_t esize)
{
    return allgather_chunks(c, base, (size_t)c->shard_len * esize);
}

/* in-place all-gather of `bytes` per rank: rank q's chunk lives at base + q * bytes on every rank */
static int allgather_chunks(tcgpu_ctx *c, void *base, size_t bytes)
{
    c->comm_bytes += (double)bytes * (c->nranks - 1);                 /* received per rank */
    if (c->loop) {
        tc_loop_comm *L = c->loop;
        TC_HIP(c, hipStreamSynchronize(c->stream));
        L->bufs[c->rank] = base;
        loop_wait(c);
        for (int p = 0; p < L->nranks; p++)
            if (p != c->rank)
                TC_HIP(c, hipMemcpyAsync((char *)L->bufs[p] + (size_t)c->rank * bytes,
                                         (const char *)base + (size_t)c->rank * bytes, bytes,
                                         hipMemcpyDeviceToDevice, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        loop_wait(c);
        return 0;
    }
    ncclResult_t r = g_rccl.AllGather((const char *)base + (size_t)c->rank * bytes, base, bytes, ncclInt8,
                                      (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess) TC_FAIL(c, TCGPU_ERR_COMM, "ncclAllGather failed (%d)", (int)r);
    return 0;
}

/* all-reduce, in place on the device: buf[0..nsum) summed, buf[nsum..nsum+nmax) maximised.  All summed values
 * are integers below 2^53 held in doubles (exact limbs, counts), so the result is exact and order-independent. */
static int allreduce_scalars(tcgpu_ctx *c, double *buf, int nsum, int nmax)
{
    const int nt = nsum + nmax;
    if (c->loop) {
        tc_loop_comm *L = c->loop;
        double h[16];
        TC_HIP(c, hipMemcpyAsync(h, buf, nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(L->red[c->rank], h, nt * sizeof(double));
        loop_wait(c);
        double o[16] = {0};
        for (int p = 0; p < L->nranks; p++) {
            for (int q = 0; q < nsum; q++) o[q] += L->red[p][q];
            for (int q = nsum; q < nt; q++) o[q] = fmax(o[q], L->red[p][q]);
        }
        loop_wait(c);
        TC_HIP(c, hipMemcpyAsync(buf, o, nt * sizeof(double), hipMemcpyHostToDevice, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    g_rccl.GroupStart();
    ncclResult_t r1 = ncclSuccess, r2 = ncclSuccess;
    if (nsum > 0) r1 = g_rccl.AllReduce(buf, buf, nsum, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream);
    if (nmax > 0) r2 = g_rccl.AllReduce(buf + nsum, buf + nsum, nmax, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream);
    ncclResult_t r3 = g_rccl.GroupEnd();
    if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess) TC_FAIL(c, TCGPU_ERR_COMM, "ncclAllReduce failed");
    return 0;
}

static inline bool multi(const tcgpu_ctx *c) { return c->comm || c->loop; }
static int ensure_pos_all(tcgpu_ctx *c);

tc_loop_guard::tc_loop_guard(tcgpu_ctx *ctx) : c(ctx)
{
    if (c && c->loop && c->loop->exclusive) pthread_mutex_lock(&c->loop->token);
}
tc_loop_guard::~tc_loop_guard()
{
    if (c && c->loop && c->loop->exclusive) {
        hipStreamSynchronize(c->stream);
        pthread_mutex_unlock(&c->loop->token);
    }
}

/* multi-rank contexts: agree on the error flags (maximum over the ranks), then check them; every rank
 * returns the same status.  Single rank: the plain check. */
static int check_flags_collective(tcgpu_ctx *c)
{
    if (!multi(c)) return check_flags(c);
    double *buf = tc_pass_scalars(c) + 32;
    int rc = tc_launch_flags_to_f64(c, buf);
    if (rc) return rc;
    tc_phase_begin(c, PH_COMM);
    rc = allreduce_scalars(c, buf, 0, TC_PS_FLAGS);
    tc_phase_end(c);
    if (rc) return rc;
    TC_HIP(c, hipMemcpyAsync(c->h_red + 32, buf, TC_PS_FLAGS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    return check_flags(c, c->h_red + 32);
}

/* testing: every RCCL entry point the sharded path uses, through the context's communicator (tests give it a 1-rank
 * one, option "force_comm"): all-gather and all-reduce in place, and a grouped ncclSend / ncclRecv pair to the own rank
 * -- the ghost exchange's call pattern -- with the payload checked.  Not part of the public header. */
extern "C" int tcgpu_debug_comm_selftest(tcgpu_ctx *c)
{
    if (!c || !c->comm) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    const int n = 4096;
    uint32_t *d = nullptr;
    TC_HIP(c, hipMalloc(&d, 2 * n * sizeof(uint32_t)));
    uint32_t *h = (uint32_t *)malloc(2 * n * sizeof(uint32_t));
    if (!h) { hipFree(d); return TCGPU_ERR_NOMEM; }
    for (int i = 0; i < n; i++) { h[i] = 0x9e3779b9u * (uint32_t)(i + 1); h[n + i] = 0; }
    hipError_t e = hipMemcpy(d, h, 2 * n * sizeof(uint32_t), hipMemcpyHostToDevice);
    bool bad = e != hipSuccess;
    bad |= g_rccl.GroupStart() != ncclSuccess;
    bad |= g_rccl.Send(d, n * sizeof(uint32_t), ncclInt8, c->rank, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= g_rccl.Recv(d + n, n * sizeof(uint32_t), ncclInt8, c->rank, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= g_rccl.GroupEnd() != ncclSuccess;
    bad |= g_rccl.AllGather((const char *)d + (size_t)c->rank * n * sizeof(uint32_t) / c->nranks, d,
                            n * sizeof(uint32_t) / c->nranks, ncclInt8, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= hipStreamSynchronize(c->stream) != hipSuccess;
    bad |= hipMemcpy(h, d, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess;
    for (int i = 0; i < n && !bad; i++) bad = h[n + i] != 0x9e3779b9u * (uint32_t)(i + 1) || h[i] != h[n + i];
    /* the shapes the ghost exchange produces: a group in which a peer sends nothing (its sends and receives are simply
     * not posted: an EMPTY group must come back), and two messages of different, odd sizes per peer in one group (the
     * index and the position part of an uneven tail shard) */
    bad |= g_rccl.GroupStart() != ncclSuccess;
    bad |= g_rccl.GroupEnd() != ncclSuccess;
    const size_t n1 = 1237, n2 = 3 * 1237 + 5;                      /* bytes: neither a multiple of 4 nor of each other */
    bad |= hipMemsetAsync(d + n, 0, n * sizeof(uint32_t), c->stream) != hipSuccess;
    bad |= g_rccl.GroupStart() != ncclSuccess;
    bad |= g_rccl.Send(d, n1, ncclInt8, c->rank, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= g_rccl.Send((const char *)d + 2048, n2, ncclInt8, c->rank, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= g_rccl.Recv(d + n, n1, ncclInt8, c->rank, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= g_rccl.Recv((char *)(d + n) + 2048, n2, ncclInt8, c->rank, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
    bad |= g_rccl.GroupEnd() != ncclSuccess;
    bad |= hipStreamSynchronize(c->stream) != hipSuccess;
    bad |= hipMemcpy(h, d, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess;
    {
        const unsigned char *a = (const unsigned char *)h, *b = (const unsigned char *)(h + n);
        for (size_t i = 0; i < n1 && !bad; i++) bad = a[i] != b[i];
        for (size_t i = 0; i < n2 && !bad; i++) bad = a[2048 + i] != b[2048 + i];
        for (size_t i = n1; i < 2048 && !bad; i++) bad = b[i] != 0;     /* nothing beyond the message lengths */
    }
    free(h);
    hipFree(d);
    if (bad) TC_FAIL(c, TCGPU_ERR_COMM, "RCCL self-test failed");
    return TCGPU_OK;
}

/* testing: tie `nranks` contexts of this process into a loopback communicator (one thread each) */
extern "C" int tcgpu_comm_init_loopback(tcgpu_ctx **ctxs, int nranks)
{
    if (!ctxs || nranks < 1 || nranks > 16) return TCGPU_ERR_ARG;
    tc_loop_comm *L = (tc_loop_comm *)calloc(1, sizeof(*L));
    if (!L) return TCGPU_ERR_NOMEM;
    L->nranks = nranks;
    L->exclusive = getenv("TCGPU_LOOPBACK_EXCLUSIVE") != nullptr;
    pthread_mutex_init(&L->token, nullptr);
    pthread_barrier_init(&L->bar, nullptr, (unsigned)nranks);
    for (int r = 0; r < nranks; r++) {
        if (!ctxs[r] || ctxs[r]->n > 0) { free(L); return TCGPU_ERR_ARG; }
        ctxs[r]->loop = L;              /* shared; intentionally leaked with the last context (test only) */
        ctxs[r]->rank = r;
        ctxs[r]->nranks = nranks;
    }
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ presentation: G <- Peano order */

/* Bring the global arrays into the Peano order of the last density pass -- the order the reference leaves P and
 * SphP in (src/peano.c:85-126) -- completing the per-particle state on every rank first.  Called lazily by
 * everything that shows or takes per-particle arrays; inside the relaxation loop nothing needs it.  It is also
 * what makes the index ranges of G spatially compact shards (g_compact). */
static int present(tcgpu_ctx *c)
{
    if (!c->order_dirty) return 0;
    int rc = 0;
    const int b = c->gcur;
    if ((rc = ensure_pos_all(c))) return rc;      /* the permute moves every particle's position */
    tc_phase_begin(c, PH_PRESENT);
    if (multi(c)) {
        if (c->comm) g_rccl.GroupStart();
        rc = allgather_inplace(c, c->g_hsml[b], sizeof(float));
        if (!rc) rc = allgather_inplace(c, c->g_rho[b], sizeof(float));
        if (!rc) rc = allgather_inplace(c, c->g_vhf[b], sizeof(float));
        if (!rc) rc = allgather_inplace(c, c->g_rhom[b], sizeof(float));
        if (!rc) rc = allgather_inplace(c, c->g_key, sizeof(tc_u128));
        if (c->comm && g_rccl.GroupEnd() != ncclSuccess && !rc) {
            snprintf(c->err, sizeof(c->err), "ncclGroupEnd failed");
            rc = TCGPU_ERR_COMM;
        }
        if (rc) { tc_phase_end(c); return rc; }
    }
    if ((rc = tc_launch_iota(c, c->idx, (size_t)c->n, 0))) { tc_phase_end(c); return rc; }
    int sort_levels = c->lmax + 5;
    if (sort_levels > 21) sort_levels = 21;
    if (tc_sort_pairs_u128(c->sort_tmp, c->sort_tmp_bytes, c->g_key, c->g_key_sorted, c->idx, c->idx_sorted, (size_t)c->n,
                           3 * sort_levels + 1, c->stream)) {
        tc_phase_end(c);
        TC_FAIL(c, TCGPU_ERR_HIP, "radix sort failed");
    }
    if ((rc = tc_launch_present_permute(c, c->idx_sorted))) { tc_phase_end(c); return rc; }
    TC_HIP(c, hipMemcpyAsync(c->g_key, c->g_key_sorted, (size_t)c->n * sizeof(tc_u128), hipMemcpyDeviceToDevice, c->stream));
    c->order_dirty = 0;
    c->keys_valid = 1;
    c->g_compact = 1;
    if (c->local_full && c->index_valid) {
        /* a full local set is sorted by the same keys with the same tie rule: local slot i IS presented index i */
        int64_t lo, hi;
        tc_own_range(c, &lo, &hi);
        if ((rc = tc_launch_iota(c, c->lg, (size_t)c->n, 0))) return rc;
        if (multi(c) && (rc = tc_launch_iota(c, c->own_list, (size_t)(hi - lo), (uint32_t)lo))) return rc;
        c->nown = hi - lo;
    } else {
        c->index_valid = 0; c->mirror_valid = 0; c->ustep_valid = 0;
    }
    if (multi(c)) c->w_valid = 0;                 /* hwvt / rhom_next are own-range arrays: the ranges just changed */
    tc_phase_end(c);
    return 0;
}

/* ------------------------------------------------------------------ model hsml of the current positions */

/* src/wvt_relax.c:108-124 for the own range, normalised with the sum over ALL particles (exact, all-reduced);
 * the result rides in g_pos4.w, so sharded contexts complete positions and w with ONE all-gather. */
static int ensure_w(tcgpu_ctx *c)
{
    if (c->w_valid) return 0;
    int rc;
    if ((rc = tc_launch_model_hsml_own(c))) return rc;
    if (multi(c)) {
        tc_phase_begin(c, PH_COMM);
        rc = allreduce_scalars(c, tc_pass_scalars(c) + TC_PS_H3, 3, 0);
        tc_phase_end(c);
        if (rc) return rc;
    }
    if ((rc = tc_launch_scale_hsml_own(c))) return rc;
    c->w_valid = 1;
    c->local_w_valid = 0;                         /* the local copy of the w lane (if any) predates it */
    c->mirror_valid = 0;
    if (multi(c)) c->pos_all_valid = 0;           /* the other ranks' copies of the own w lane are stale */
    return 0;
}

/* Sharded contexts: make g_pos4 complete -- every rank's own positions and model hsml, ONE in-place all-gather of
 * 16 B per particle.  Needed by passes over the whole set (cold start, repeated pass, presentation); the steady
 * state exchanges ghosts only (exchange_ghosts). */
static int ensure_pos_all(tcgpu_ctx *c)
{
    if (!multi(c) || c->pos_all_valid) return 0;
    tc_phase_begin(c, PH_COMM);
    int rc = allgather_inplace(c, c->g_pos4[c->gcur], sizeof(float4));
    tc_phase_end(c);
    if (rc) return rc;
    c->pos_all_valid = 1;
    return 0;
}

/* Ghost exchange of one sharded pass (after the marking kernel has written this rank's pyramid):
 *   1. all-gather the pyramids (2.4 MB per rank);
 *   2. every rank tests ITS OWN particles against the other ranks' pyramids (the test the receivers used to run
 *      over all positions) and counts per destination; the count matrix is all-gathered (R x R ints) -- the one
 *      host synchronisation of the pass;
 *   3. grouped ncclSend / ncclRecv of (global index, position + model hsml) = 20 B per ghost;
 *   4. received positions land in g_pos4 at their global index; lsel = ascending union of own range and ghosts.
 * The local set that results is the one the flag pass over all positions selected (same bits, same test), so
 * everything downstream -- and the bit-identity with the single-rank run -- is unchanged. */
/* Sharded passes: a failure that only ONE rank sees (an allocation sized by its own ghost counts, its table layout, a
 * launch) must not let that rank return while the others post their sends and receives to it -- they would wait for
 * ever.  The ranks therefore agree on the status before any point-to-point traffic: one scalar all-reduce (max) and a
 * stream synchronisation, ~20 us per pass.  Every rank returns an error, or none does.   */
static int agree_status(tcgpu_ctx *c, int rc_local, const char *what)
{
    if (!multi(c)) return rc_local;
    double *buf = tc_pass_scalars(c) + 44;
    const double mine = rc_local ? 1.0 : 0.0;
    double any = 1.0;
    if (hipMemcpyAsync(buf, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return rc_local ? rc_local : TCGPU_ERR_HIP;
    if (allreduce_scalars(c, buf, 0, 1)) return rc_local ? rc_local : TCGPU_ERR_COMM;
    if (hipMemcpyAsync(&any, buf, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess
        || hipStreamSynchronize(c->stream) != hipSuccess)
        return rc_local ? rc_local : TCGPU_ERR_HIP;
    if (rc_local) return rc_local;
    if (any != 0.0) TC_FAIL(c, TCGPU_ERR_COMM, "another rank failed in %s: pass abandoned on every rank", what);
    return 0;
}

/* rc_in: a failure of this rank on the way here (the marking launch); it is reported through the agreed status */
static int exchange_ghosts(tcgpu_ctx *c, int64_t *nloc, int rc_in)
{
    const int R = c->nranks, me = c->rank;
    int rc;
    int64_t lo, hi;
    tc_own_range(c, &lo, &hi);
    /* A failure only this rank sees (a launch, a copy) must not make it leave while the others wait in the next collective:
     * it is remembered, the collectives below are still entered, and the status is agreed before anybody sends. */
    int rc_early = rc_in;
    tc_phase_begin(c, PH_COMM);
    rc = allgather_chunks(c, c->pyr_all, c->pyr_chunk * sizeof(uint32_t));
    tc_phase_end(c);
    if (rc && !rc_early) rc_early = rc;
    tc_phase_begin(c, PH_LOCAL);
    rc = tc_launch_ghost_count(c);
    tc_phase_end(c);
    if (rc && !rc_early) rc_early = rc;
    tc_phase_begin(c, PH_COMM);
    rc = allgather_chunks(c, c->ghost_cnt_mat, (size_t)R * sizeof(int));
    tc_phase_end(c);
    if (rc && !rc_early) rc_early = rc;
    int soff[TC_GHOST_MAXR + 1], roff[TC_GHOST_MAXR + 1];
    size_t nsend = 0, nrecv = 0;
    auto prepare = [&]() -> int {
        int r;
        TC_HIP(c, hipMemcpyAsync(c->h_cnt_mat, c->ghost_cnt_mat, (size_t)R * R * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if ((r = tc_finish_local_layout(c))) return r;                /* synchronises: counts and table layout */
        soff[0] = roff[0] = 0;
        for (int q = 0; q < R; q++) {
            soff[q + 1] = soff[q] + c->h_cnt_mat[(size_t)me * R + q];      /* row me: what I send to q (0 for q == me) */
            roff[q + 1] = roff[q] + c->h_cnt_mat[(size_t)q * R + me];      /* column me: what q sends to me */
        }
        nsend = (size_t)soff[R]; nrecv = (size_t)roff[R];
        {
            /* Is this cheaper than sending every position to every rank?  Not when nearly everything is somebody's
             * ghost (few particles per rank: thick shells) -- then the next passes all-gather instead and the question
             * is asked again later.  Decided from the whole count matrix, so every rank decides the same. */
            double ghosts = 0;
            for (int q = 0; q < R * R; q++) ghosts += c->h_cnt_mat[q];
            const double cost_ghost = ghosts * (sizeof(uint32_t) + sizeof(float4))
                                      + (double)R * (R - 1) * (double)(c->pyr_chunk * sizeof(uint32_t));
            const double cost_all = (double)R * (double)(c->n - c->shard_len) * sizeof(float4);
            if (c->ghost_mode == 1 && c->margin_widen == 0 && cost_ghost > cost_all) c->ghost_pause = 7;   /* (a repeated pass
                                                                                           * has a wider shell: not typical) */
        }
        if ((int64_t)nrecv + (hi - lo) > c->cap) TC_FAIL(c, TCGPU_ERR_NOMEM, "local set larger than the particle capacity");
        if (nsend > c->send_cap) {
            hipFree(c->send_idx); hipFree(c->send_pos);
            c->send_idx = nullptr; c->send_pos = nullptr; c->send_cap = 0;
            const size_t cap = nsend + nsend / 4 + 1024;
            TC_HIP(c, hipMalloc(&c->send_idx, cap * sizeof(uint32_t)));
            TC_HIP(c, hipMalloc(&c->send_pos, cap * sizeof(float4)));
            c->send_cap = cap;
        }
        if (nrecv > c->ghost_cap) {
            hipFree(c->ghost_idx); hipFree(c->ghost_pos);
            c->ghost_idx = nullptr; c->ghost_pos = nullptr; c->ghost_cap = 0;
            const size_t cap = nrecv + nrecv / 4 + 1024;
            TC_HIP(c, hipMalloc(&c->ghost_idx, cap * sizeof(uint32_t)));
            TC_HIP(c, hipMalloc(&c->ghost_pos, cap * sizeof(float4)));
            c->ghost_cap = cap;
        }
        tc_phase_begin(c, PH_LOCAL);
        r = tc_launch_ghost_fill(c);
        tc_phase_end(c);
        return r;
    };
    rc = rc_early ? rc_early : prepare();
    if (c->debug_fail_rank == me + 1 && !rc) {                        /* tests: a failure only this rank sees */
        snprintf(c->err, sizeof(c->err), "injected failure on rank %d", me);
        rc = TCGPU_ERR_NOMEM;
    }
    if ((rc = agree_status(c, rc, "the ghost exchange's preparation"))) return rc;
    tc_phase_begin(c, PH_COMM);
    c->comm_bytes += (double)nrecv * (sizeof(uint32_t) + sizeof(float4));
    if (c->loop) {
        tc_loop_comm *L = c->loop;
        TC_HIP(c, hipStreamSynchronize(c->stream));
        L->sidx[me] = c->send_idx; L->spos[me] = c->send_pos;
        memcpy(L->soff[me], soff, sizeof(int) * (R + 1));
        loop_wait(c);
        for (int q = 0; q < R; q++) {
            const int cnt = roff[q + 1] - roff[q];
            if (q == me || cnt == 0) continue;
            TC_HIP(c, hipMemcpyAsync(c->ghost_idx + roff[q], L->sidx[q] + L->soff[q][me], (size_t)cnt * sizeof(uint32_t),
                                     hipMemcpyDeviceToDevice, c->stream));
            TC_HIP(c, hipMemcpyAsync(c->ghost_pos + roff[q], L->spos[q] + L->soff[q][me], (size_t)cnt * sizeof(float4),
                                     hipMemcpyDeviceToDevice, c->stream));
        }
        TC_HIP(c, hipStreamSynchronize(c->stream));
        loop_wait(c);
    } else {
        bool bad = g_rccl.GroupStart() != ncclSuccess;
        for (int q = 0; q < R; q++) {
            if (q == me) continue;
            const size_t ns = (size_t)(soff[q + 1] - soff[q]), nr = (size_t)(roff[q + 1] - roff[q]);
            if (ns) {
                bad |= g_rccl.Send(c->send_idx + soff[q], ns * sizeof(uint32_t), ncclInt8, q, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
                bad |= g_rccl.Send(c->send_pos + soff[q], ns * sizeof(float4), ncclInt8, q, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
            }
            if (nr) {
                bad |= g_rccl.Recv(c->ghost_idx + roff[q], nr * sizeof(uint32_t), ncclInt8, q, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
                bad |= g_rccl.Recv(c->ghost_pos + roff[q], nr * sizeof(float4), ncclInt8, q, (ncclComm_t)c->comm, c->stream) != ncclSuccess;
            }
        }
        bad |= g_rccl.GroupEnd() != ncclSuccess;                       /* the group is closed on every path */
        if (bad) { tc_phase_end(c); TC_FAIL(c, TCGPU_ERR_COMM, "ghost exchange (ncclSend / ncclRecv) failed"); }
    }
    tc_phase_end(c);
    c->nghost = (int64_t)nrecv;
    c->nghost_lo = roff[me];
    tc_phase_begin(c, PH_LOCAL);
    rc = tc_launch_ghost_scatter(c);
    tc_phase_end(c);
    if (rc) return rc;
    *nloc = (int64_t)nrecv + (hi - lo);
    return 0;
}

/* ------------------------------------------------------------------ local set + neighbour index */

/* K1-K4 of one pass: choose the local set (everything if `full`), key and sort it (src/peano.c:46-126), gather
 * it into sorted order, build the cell table (replaces src/tree.c:124-271). */
static int build_local(tcgpu_ctx *c, int full, int with_cells, int mark_dirty)
{
    int rc;
    c->index_valid = 0; c->mirror_valid = 0; c->ustep_valid = 0; c->pf_valid = 0; c->xlist_valid = 0;
    if (!multi(c)) full = 1;
    if (full) {
        if ((rc = ensure_pos_all(c))) return rc;
        c->local_full = 1;
        c->nloc = c->n;
        c->lmin_tab = 1;
        if ((rc = tc_layout_table(c, nullptr))) return rc;
    } else if (c->nranks <= TC_GHOST_MAXR && c->ghost_mode != 0 && c->ghost_pause == 0) {
        int64_t nloc = 0;
        tc_phase_begin(c, PH_LOCAL);
        rc = tc_launch_mark_interest(c);
        tc_phase_end(c);
        rc = exchange_ghosts(c, &nloc, rc);                          /* synchronises: the launch sizes below need nloc */
        if (rc) return rc;
        c->local_full = 0;
        c->nloc = nloc;
    } else {                                                          /* every position to every rank, then select */
        int64_t nloc = 0;
        if (c->ghost_pause > 0) c->ghost_pause--;
        if ((rc = ensure_pos_all(c))) return rc;
        tc_phase_begin(c, PH_LOCAL);
        rc = tc_launch_mark_interest(c);
        if (!rc) rc = tc_select_local(c, &nloc);                      /* synchronises: the launch sizes below need nloc */
        tc_phase_end(c);
        if (rc) return rc;
        c->local_full = 0;
        c->nloc = nloc;
    }
    c->lmax_rm = c->lmax_rm0;
    c->lmin_rm = c->lmin_rm0 > c->lmin_tab ? c->lmin_rm0 : c->lmin_tab;
    if (c->lmin_rm > c->lmax_rm) c->lmax_rm = 0;
    if ((rc = tc_launch_keys_local(c))) return rc;
    tc_phase_begin(c, PH_SORT);
    /* large sets: radix passes only over the Hilbert levels that separate particles in practice, five below
     * the deepest cell-table level (3 bits per level); the fix-up orders whatever still ties, exactly */
    int sort_levels = c->lmax + 5;
    if (sort_levels > 21) sort_levels = 21;
    int s = tc_sort_pairs_u128(c->sort_tmp, c->sort_tmp_bytes, c->key, c->key_sorted, c->idx, c->idx_sorted,
                               (size_t)c->nloc, 3 * sort_levels + 1, c->stream);
    tc_phase_end(c);
    if (s) TC_FAIL(c, TCGPU_ERR_HIP, "radix sort failed");
    if ((rc = tc_launch_gather_local(c))) return rc;
    c->local_w_valid = c->w_valid;
    if (mark_dirty) { c->order_dirty = 1; c->keys_valid = 0; }
    if (with_cells && (rc = tc_launch_cells(c))) return rc;
    return 0;
}

extern "C" int tcgpu_sort_particles_by_peano_key(tcgpu_ctx *c)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = present(c))) return rc;                           /* state of earlier passes first */
    if ((rc = build_local(c, 1, 0, 1))) return rc;
    c->index_valid = 0;
    return present(c);
}

extern "C" int tcgpu_download_particles(tcgpu_ctx *c, float *pos, int32_t *id, float *hsml, float *rho, float *vhf,
                                        float *rhom)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = present(c);
    if (rc) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    size_t n = (size_t)c->n;
    int b = c->gcur;
    if (pos) {
        float4 *tmp = (float4 *)malloc(n * sizeof(float4));
        if (!tmp) return TCGPU_ERR_NOMEM;
        hipError_t e = hipMemcpy(tmp, c->g_pos4[b], n * sizeof(float4), hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < n; i++) { pos[3 * i] = tmp[i].x; pos[3 * i + 1] = tmp[i].y; pos[3 * i + 2] = tmp[i].z; }
        free(tmp);
        TC_HIP(c, e);
    }
    if (id) TC_HIP(c, hipMemcpy(id, c->g_id[b], n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (hsml) TC_HIP(c, hipMemcpy(hsml, c->g_hsml[b], n * sizeof(float), hipMemcpyDeviceToHost));
    if (rho) TC_HIP(c, hipMemcpy(rho, c->g_rho[b], n * sizeof(float), hipMemcpyDeviceToHost));
    if (vhf) TC_HIP(c, hipMemcpy(vhf, c->g_vhf[b], n * sizeof(float), hipMemcpyDeviceToHost));
    if (rhom) TC_HIP(c, hipMemcpy(rhom, c->g_rhom[b], n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

extern "C" int tcgpu_download_keys(tcgpu_ctx *c, uint64_t *hi, uint64_t *lo)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = present(c);
    if (rc) return rc;
    if (!c->keys_valid) TC_FAIL(c, TCGPU_ERR_ARG, "no keys yet (call tcgpu_sort_particles_by_peano_key or a density pass first)");
    TC_HIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)c->n;
    tc_u128 *t = (tc_u128 *)malloc(n * sizeof(tc_u128));
    if (!t) return TCGPU_ERR_NOMEM;
    hipError_t e = hipMemcpy(t, c->g_key_sorted, n * sizeof(tc_u128), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        for (size_t i = 0; i < n; i++) {
            if (hi) hi[i] = (uint64_t)(t[i] >> 64);
            if (lo) lo[i] = (uint64_t)t[i];
        }
    free(t);
    TC_HIP(c, e);
    rc = check_flags(c);
    return rc == TC_RETRY_FULL ? TCGPU_OK : rc;
}

extern "C" int tcgpu_peano_keys(tcgpu_ctx *c, int64_t n, const double *xyz, uint64_t *hi, uint64_t *lo)
{
    if (!c || n <= 0 || !xyz || !hi || !lo) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    double *d_xyz = nullptr;
    uint64_t *d_hi = nullptr, *d_lo = nullptr;
    TC_HIP(c, hipMalloc(&d_xyz, 3 * n * sizeof(double)));
    TC_HIP(c, hipMalloc(&d_hi, n * sizeof(uint64_t)));
    TC_HIP(c, hipMalloc(&d_lo, n * sizeof(uint64_t)));
    int rc = TCGPU_OK;
    if (hipMemcpy(d_xyz, xyz, 3 * n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = TCGPU_ERR_HIP;
    if (!rc) rc = tc_launch_keys_xyz(c, n, d_xyz, d_hi, d_lo);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = TCGPU_ERR_HIP;
    if (!rc && hipMemcpy(hi, d_hi, n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = TCGPU_ERR_HIP;
    if (!rc && hipMemcpy(lo, d_lo, n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = TCGPU_ERR_HIP;
    hipFree(d_xyz); hipFree(d_hi); hipFree(d_lo);
    return rc;
}

/* the whole set, sorted and indexed, in presented order: what tcgpu_find_ngb / tcgpu_guess_hsml address */
static int full_index(tcgpu_ctx *c)
{
    int rc;
    if ((rc = present(c))) return rc;
    if (c->index_valid && c->local_full) return 0;
    if ((rc = build_local(c, 1, 1, 1))) return rc;
    return present(c);
}

extern "C" int tcgpu_build_neighbour_index(tcgpu_ctx *c)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = full_index(c);
    if (rc) return rc;
    rc = check_flags_collective(c);
    return rc == TC_RETRY_FULL ? TCGPU_OK : rc;
}

/* ------------------------------------------------------------------ density pass */

static int density_stats(tcgpu_ctx *c)
{
    size_t cap = (size_t)c->cap, nown = (size_t)c->nown;
    uint32_t *h = (uint32_t *)malloc(4 * cap * sizeof(uint32_t));
    uint32_t *own = (uint32_t *)malloc((nown ? nown : 1) * sizeof(uint32_t));
    if (!h || !own) { free(h); free(own); return TCGPU_ERR_NOMEM; }
    if (!c->stats) { free(h); free(own); TC_FAIL(c, TCGPU_ERR_ARG, "no work counters: option stats was off during the pass"); }
    hipError_t e = hipMemcpy(h, c->stats, 4 * cap * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && c->nranks > 1) e = hipMemcpy(own, c->own_list, nown * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        double s[4] = {0, 0, 0, 0};
        for (int q = 0; q < 4; q++)
            for (size_t t = 0; t < nown; t++) s[q] += h[q * cap + (c->nranks > 1 ? own[t] : t)];
        double m = (double)(nown ? nown : 1);
        c->last_stats.queries_per_particle = s[0] / m;
        c->last_stats.solver_iters_per_particle = s[1] / m;
        c->last_stats.pair_evals_per_particle = s[2] / m;
        c->last_stats.candidates_per_particle = s[3] / m;
    }
    free(h); free(own);
    TC_HIP(c, e);
    return 0;
}

/* The ordered gather keeps a run list (TC_XRCAP x 8 B) and a neighbour list (TC_XLCAP x 4 B) per local slot: 2.8 KB per
 * particle, 5.7 GB at 2e6, 45 GB at 1.6e7 -- and 283 GB at 1e8, which one GPU does not have next to everything else.
 * Lists are used when they take at most 40 % of the device's memory and fit into what is free; otherwise the sweep runs
 * stand-alone (k_wvt_exact4: same sums, 4 ms more per 2e6 particles).  They are sized by the LOCAL SET of the pass: a
 * sharded rank in steady state (own range + ghost shell, 1.4-1.9 x N / R) has room for them where the full set has not. */
static bool xlists_fit(tcgpu_ctx *c)
{
    if (c->xr_cap >= (size_t)c->nloc) return true;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    const double per = TC_XRCAP * 8.0 + TC_XLCAP * 4.0 + 12.0;
    const double need = ((double)c->nloc * 1.125 + 1024) * per;
    const double have = (double)c->xr_cap * per;                   /* freed when the lists grow */
    return need <= 0.4 * (double)tot && need + 4e9 <= (double)fr + have;
}

/* One density pass (src/sph.c:13-72) up to, not including, the write-back of the results: local set, sort, index,
 * first-pass guess, solve.  `full` forces the whole set as local set. */
static int density_pass_launch(tcgpu_ctx *c, int need_guess, int with_wvt, int full)
{
    int rc;
    /* first warm pass of a sharded context: G is still the upload order -- make the shards compact first */
    if (multi(c) && !c->g_compact && c->order_dirty && !need_guess && (rc = present(c))) return rc;
    if (with_wvt && (rc = ensure_w(c))) return rc;
    if (need_guess || !c->g_compact) full = 1;
    if ((rc = build_local(c, full, 1, 1))) return rc;
    if (need_guess && (rc = tc_launch_guess(c))) return rc;
    if (c->fuse) {
        /* one gather per particle serves the density solve and (with_wvt) the WVT sweep that
         * follows on the same positions */
        /* the sweep rides along only in round 2's mode (f64 sums, one rounding); the default sweep reproduces the
         * reference's order and roundings in a kernel of its own (k_wvt_exact), after the step is known */
        int ride = !with_wvt ? 0 : c->sweep_mode == 1 ? 1 : (c->sweep_mode == 0 && !c->xsweep_kernel) ? 2 : 0;
        /* per-particle lists: not when they would not fit, and not for the few full-set passes of a sharded context (cold
         * start, repeated pass) -- sized for everything they would be R times what the rank needs afterwards */
        if (ride == 2 && multi(c) && (c->local_full || c->margin_widen > 0)) ride = 0;    /* (a repeated pass has a wider shell) */
        if (ride == 2) { c->lists_unfit = !xlists_fit(c); if (c->lists_unfit) ride = 0; }
        /* nor on a cold pass: without a carried hsml no particle gets a list, and the wave-per-particle kernel that serves
         * the unlisted few would serve everybody (41 ms at 2e6 against 8 ms for the stand-alone sweep) */
        if (ride == 2 && need_guess) ride = 0;
        if (ride == 2 && tc_ensure_xlists(c)) { ride = 0; c->lists_unfit = 1; }       /* memory not to be had: no lists this pass */
        if (ride == 2) {                                                            /* the ordered runs come from pf; */
            if (!c->pf_valid && (rc = tc_launch_pfirst(c))) return rc;              /* nobody needs the mirror */
            if (multi(c) && c->mirror) {
                /* a sharded rank in steady state: the mirror of the full-set passes (cold start, presentation) is
                 * 20 B x 5 levels x N that nobody reads any more -- given back; a later mirror pass allocates for ITS set */
                TC_HIP(c, hipStreamSynchronize(c->stream));
                TC_FREE(c->mirror); TC_FREE(c->mirror_idx);
                c->mirror_alloc = 0; c->mirror_valid = 0;
            }
        } else if ((rc = tc_launch_mirror(c))) return rc;
        if ((rc = tc_launch_iter(c, ride))) return rc;
        c->ustep_valid = ride == 1;
        c->xlist_valid = ride == 2;
    } else if ((rc = tc_launch_density(c))) return rc;
    return 0;
}

extern "C" int tcgpu_find_sph_quantities(tcgpu_ctx *c)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc, full = 0;
    c->margin_widen = 0;
    for (;;) {
        if ((rc = density_pass_launch(c, c->need_guess, 0, full))) return rc;
        rc = check_flags_collective(c);
        if (rc != TC_RETRY_FULL) break;
        /* a query left the ghost margin: the same pass again with a wider shell, at last on the full set */
        c->margin_retry++;
        c->margin_widen += 2;
        if (c->margin_widen > 4) full = 1;
    }
    c->margin_widen = 0;
    if (rc) return rc;
    if ((rc = tc_launch_scatter_results(c))) return rc;
    c->need_guess = 0;                            /* every hsml is > 0 after a successful pass */
    if (c->want_stats) return density_stats(c);
    return TCGPU_OK;
}

extern "C" int tcgpu_last_density_stats(tcgpu_ctx *c, tcgpu_density_stats *out)
{
    if (!c || !out) return TCGPU_ERR_ARG;
    *out = c->last_stats;
    return TCGPU_OK;
}

extern "C" int tcgpu_global_density_model(tcgpu_ctx *c, float *out)
{
    tc_loop_guard guard_(c);
    if (!c || !out || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = present(c);
    if (rc) return rc;
    rc = tc_launch_model(c, c->guess);           /* guess doubles as f32 scratch */
    if (rc) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    TC_HIP(c, hipMemcpy(out, c->guess, (size_t)c->n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

extern "C" int tcgpu_guess_hsml(tcgpu_ctx *c, float *out)
{
    tc_loop_guard guard_(c);
    if (!c || !out || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = present(c))) return rc;
    if ((rc = build_local(c, 1, 0, 1))) return rc;            /* keys of the whole set, sorted: the guess reads the tree */
    if ((rc = present(c))) return rc;
    if ((rc = tc_launch_guess(c))) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    TC_HIP(c, hipMemcpy(out, c->guess, (size_t)c->n * sizeof(float), hipMemcpyDeviceToHost));
    rc = check_flags(c);
    return rc == TC_RETRY_FULL ? TCGPU_OK : rc;
}

extern "C" int tcgpu_find_ngb(tcgpu_ctx *c, int64_t ipart, float hsml, int32_t *list, int32_t *count)
{
    tc_loop_guard guard_(c);
    if (!c || !list || !count || ipart < 0 || ipart >= c->n) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = present(c);
    if (rc) return rc;
    if (!c->index_valid || !c->local_full)
        TC_FAIL(c, TCGPU_ERR_ARG, "neighbour index over the whole set not built (call tcgpu_build_neighbour_index)");
    rc = tc_launch_find_ngb(c, (int)ipart, hsml);
    if (rc) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    int cnt = 0;
    TC_HIP(c, hipMemcpy(&cnt, c->ngb_cnt, sizeof(int), hipMemcpyDeviceToHost));
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (cnt > 0 ? cnt : 1));
    if (!tmp) return TCGPU_ERR_NOMEM;
    hipError_t e = hipMemcpy(tmp, c->ngb_buf, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        std::sort(tmp, tmp + cnt);                                  /* tree.c emits ascending indices */
        if (cnt > TCGPU_NGBMAX) cnt = TCGPU_NGBMAX;                 /* tree.c:91-92 */
        memcpy(list, tmp, sizeof(int32_t) * cnt);
        *count = cnt;
    }
    free(tmp);
    TC_HIP(c, e);
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ WVT */

static int wvt_step_nocheck(tcgpu_ctx *c, double step, int move)
{
    int rc;
    if (c->ustep_valid) {                         /* the fused pass already summed the sweep for unit step */
        if ((rc = tc_launch_commit_rhom(c))) return rc;
        if ((rc = tc_launch_apply_step(c, step))) return rc;
    } else {
        if ((rc = ensure_w(c))) return rc;
        if (!c->index_valid) {                    /* e.g. a presentation dropped a sharded local set: rebuild it */
            if ((rc = build_local(c, !c->g_compact, 1, 0))) return rc;
        } else if (!c->local_w_valid) {           /* the model hsml was computed after the local gather */
            if ((rc = ensure_pos_all(c))) return rc;     /* ... and after the ghosts' w lane travelled */
            if ((rc = tc_launch_refresh_w(c))) return rc;
            c->local_w_valid = 1;
        }
        if ((rc = tc_launch_commit_rhom(c))) return rc;
        if (c->sweep_mode != 1) {
            /* the exact sweep walks the cells in curve order: their index ranges come from the sorted keys of the
             * local set (key_sorted stays valid as long as the local order does) */
            if (!c->pf_valid && (rc = tc_launch_pfirst(c))) return rc;
            if ((rc = tc_launch_wvt_exact(c, step))) return rc;
        } else if ((rc = tc_launch_wvt(c, step))) return rc;
    }
    if (move) {
        if ((rc = tc_launch_move(c))) return rc;
        /* sharded contexts work out the model hsml of the new positions right away (it rides in the w lane of the
         * positions, which the next pass's ghost exchange -- or a full all-gather -- carries to the other ranks) */
        if (multi(c) && (rc = ensure_w(c))) return rc;
    }
    return 0;
}

extern "C" int tcgpu_wvt_step(tcgpu_ctx *c, double step, float *hsml_wvt, float *delta, int move)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    if (c->need_guess) TC_FAIL(c, TCGPU_ERR_ARG, "no density pass yet (call tcgpu_find_sph_quantities first)");
    if (multi(c) && (hsml_wvt || delta))
        TC_FAIL(c, TCGPU_ERR_ARG, "hsml_wvt / delta outputs are own-range arrays: not available on sharded contexts");
    TC_HIP(c, hipSetDevice(c->device));
    int rc = wvt_step_nocheck(c, step, 0);
    if (rc) return rc;
    if (hsml_wvt || delta) {
        if ((rc = present(c))) return rc;         /* outputs in the order the caller sees */
        TC_HIP(c, hipStreamSynchronize(c->stream));
        size_t n = (size_t)c->n;
        if (hsml_wvt) TC_HIP(c, hipMemcpy(hsml_wvt, c->hwvt, n * sizeof(float), hipMemcpyDeviceToHost));
        if (delta) TC_HIP(c, hipMemcpy(delta, c->delta, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (move) {
        if ((rc = tc_launch_move(c))) return rc;
        if (multi(c) && (rc = ensure_w(c))) return rc;
    }
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    return TCGPU_OK;
}

/* First half of one loop body of src/wvt_relax.c:61-92: density pass (sort, index, K5) and the
 * error sums K6.  Synchronises the stream (the caller needs errMean to decide the step). */
static int density_error_sync(tcgpu_ctx *c, int need_guess, double *err_mean, double *err_max)
{
    int rc, full = 0;
    double *ps = tc_pass_scalars(c);
    c->margin_widen = 0;
    for (;;) {
        if ((rc = density_pass_launch(c, need_guess, 1, full))) return rc;
        /* the error sums need this pass's densities in G order; the carried hsml is written back only once the
         * pass is known to be good (a repeated pass must start from the same hsml) */
        if ((rc = tc_launch_scatter_rho(c))) return rc;
        if ((rc = tc_launch_error(c))) return rc;
        /* the error flags ride along (maximised): every rank sees the same flags and takes the same exit -- a
         * rank that returned alone would leave the others waiting in the next collective */
        if ((rc = tc_launch_flags_to_f64(c, ps + TC_PS_NSUM + 1))) return rc;
        if (multi(c)) {
            tc_phase_begin(c, PH_COMM);
            rc = allreduce_scalars(c, ps, TC_PS_NSUM, TC_PS_NMAX);
            tc_phase_end(c);
            if (rc) return rc;
        }
        TC_HIP(c, hipMemcpyAsync(c->h_red, ps, (TC_PS_NSUM + TC_PS_NMAX) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        rc = check_flags(c, multi(c) ? c->h_red + TC_PS_NSUM + 1 : nullptr);      /* synchronises the stream */
        if (rc != TC_RETRY_FULL) break;
        c->margin_retry++;                        /* repeat with a wider ghost shell, at last on the full set */
        c->margin_widen += 2;
        if (c->margin_widen > 4) full = 1;
    }
    c->margin_widen = 0;
    if (rc) return rc;
    if ((rc = tc_launch_scatter_results(c))) return rc;
    /* exact sum -> f64 once, the same on every rank and for every split of the particles (wvt_relax.c:87) */
    const double sum = tc_limbs_to_double(c->h_red[0], c->h_red[1], c->h_red[2], 1.0 / TC_ERR_SCALE);
    *err_mean = sum / c->h_red[3];
    *err_max = c->h_red[4];
    return 0;
}

extern "C" int tcgpu_density_error(tcgpu_ctx *c, double *err_mean, double *err_max)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0 || !c->have_model || !err_mean || !err_max) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = density_error_sync(c, c->need_guess, err_mean, err_max);
    if (!rc) c->need_guess = 0;
    if (!rc && c->want_stats) return density_stats(c);
    return rc;
}

/* src/wvt_relax.c:25-225: the loop, its step control and stop rules run on the host;
 * one small device->host copy (error sums + flags) per iteration. */
extern "C" int tcgpu_regularise_sph_particles(tcgpu_ctx *c, int max_iter, tcgpu_iterlog *log, int32_t *nlog_out)
{
    tc_loop_guard guard_(c);
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int it = -1, nlog = 0;
#ifdef TCGPU_SPH_CUBIC_SPLINE
    double step = 0.035;                                      /* wvt_relax.c:48-49 */
#else
    double step = 0.0085;                                     /* wvt_relax.c:51 */
    if (c->par.mtotal < 1e5) step /= 2;                       /* wvt_relax.c:53-54 */
#endif
    double errLast = DBL_MAX, errDiff = DBL_MAX, errDiffLast = DBL_MAX;
    const int numiter = max_iter >= 0 ? max_iter : TCGPU_NUMITER;
    int rc;

    for (;;) {
        if (it++ >= numiter) break;                           /* wvt_relax.c:63-64 */

        double errMean = 0, errMax = 0;
        if ((rc = density_error_sync(c, c->need_guess, &errMean, &errMax))) return rc;
        c->need_guess = 0;
        errDiff = (errLast - errMean) / errMean;              /* wvt_relax.c:89 */

        if (log && nlog < TCGPU_MAXLOG) {
            log[nlog].it = it; log[nlog].err_max = errMax; log[nlog].err_mean = errMean;
            log[nlog].err_diff = errDiff; log[nlog].step = step;
        }
        nlog++;

        if (errDiff < TC_ERRDIFF_LIMIT && it > 25) break;                       /* wvt_relax.c:94-95 */
        if ((errDiff < 0) && (errDiffLast < 0) && (it > 10)) break;             /* wvt_relax.c:97-98 */
        if (errDiff < 0.01 && (it > 1)) step *= 0.8;                            /* wvt_relax.c:100-101 */
        errLast = errMean;
        errDiffLast = errDiff;

        if ((rc = wvt_step_nocheck(c, step, 1))) return rc;
    }
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    if (nlog_out) *nlog_out = nlog;
    if (c->want_stats) return density_stats(c);
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ curl */

extern "C" int tcgpu_bfld_from_rotA_sph(tcgpu_ctx *c, const float *apot, float *bfld)
{
    tc_loop_guard guard_(c);
    if (!c || !apot || !bfld || c->n <= 0) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = present(c);                          /* `apot` is in the order the caller sees */
    if (rc) return rc;
    size_t cap = (size_t)c->cap, n = (size_t)c->n;
    if (!c->apot) {
        TC_HIP(c, hipMalloc(&c->apot, 3 * cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->bfld, 3 * cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->l_apot, 6 * cap * sizeof(float)));          /* A and B in local order */
    }
    float *l_bfld = c->l_apot + 3 * cap;
    TC_HIP(c, hipMemcpyAsync(c->apot, apot, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    /* the reference uses the tree of the preceding Find_sph_quantities() (src/main.c:54-56); a sharded local set
     * does not survive the presentation, so it is rebuilt here from the converged smoothing lengths */
    if (!c->index_valid || !c->local_full) {      /* marked from the smoothing lengths alone: the curl's query is that ball */
        c->mark_ignore_w = 1;
        rc = build_local(c, !c->g_compact, 1, 0);
        c->mark_ignore_w = 0;
        if (rc) return rc;
    }
    if ((rc = tc_launch_gather_rho_vhf(c))) return rc;
    /* Make_magnetic_field sets Ax = Ay = Az (magnetic_field.c:63-65): then A rides in the w lane of the positions
     * and of the mirror, and the kernel reads no side array per neighbour; any other A takes the general path */
    int a_in_w = 0;
    if ((rc = tc_launch_gather_apot(c, &a_in_w))) return rc;
    if (multi(c)) {                               /* every rank must run the same kernel: agree on the flag */
        double *buf = tc_pass_scalars(c) + 40;
        const double mine = a_in_w ? 0.0 : 1.0;
        TC_HIP(c, hipMemcpyAsync(buf, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
        if ((rc = allreduce_scalars(c, buf, 0, 1))) return rc;
        double any = 0;
        TC_HIP(c, hipMemcpyAsync(&any, buf, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        a_in_w = any == 0.0;
    }
    if ((rc = tc_launch_mirror(c))) return rc;
    if ((rc = tc_launch_curl(c, l_bfld, a_in_w))) return rc;
    if ((rc = tc_launch_scatter_bfld(c, l_bfld))) return rc;
    if (multi(c)) {
        tc_phase_begin(c, PH_COMM);
        rc = allgather_inplace(c, c->bfld, 3 * sizeof(float));
        tc_phase_end(c);
        if (rc) return rc;
    }
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    if (!c->local_full) c->index_valid = 0;       /* that local set was marked for the curl's balls only */
    TC_HIP(c, hipMemcpy(bfld, c->bfld, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

/* diagnostics (tools/): how the ordered gather went in the last WVT == 2 pass -- out[0] own particles, [1] with a run list,
 * [2] with a neighbour list, [3] mean runs, [4] mean listed neighbours, [5] max runs, [6] max listed */
extern "C" int tcgpu_debug_xlist_stats(tcgpu_ctx *c, double *out)
{
    if (!c || !out || !c->xrn || !c->xlcnt || !c->xlist_valid) return TCGPU_ERR_ARG;     /* no lists from the last pass */
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->nloc;
    uint32_t *a = (uint32_t *)malloc(n * 4), *b = (uint32_t *)malloc(n * 4), *own = (uint32_t *)malloc(n * 4);
    hipError_t e = hipMemcpy(a, c->xrn, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(b, c->xlcnt, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && c->nranks > 1) e = hipMemcpy(own, c->own_list, (size_t)c->nown * 4, hipMemcpyDeviceToHost);
    for (int q = 0; q < 7; q++) out[q] = 0;
    if (e == hipSuccess)
        for (int64_t t = 0; t < c->nown; t++) {
            const size_t i = c->nranks > 1 ? own[t] : (size_t)t;
            out[0] += 1;
            if (a[i] != TC_XNONE) { out[1] += 1; out[3] += a[i]; if (a[i] > out[5]) out[5] = a[i]; }
            if (b[i] != TC_XNONE) { out[2] += 1; out[4] += b[i]; if (b[i] > out[6]) out[6] = b[i]; }
        }
    if (out[1] > 0) out[3] /= out[1];
    if (out[2] > 0) out[4] /= out[2];
    free(a); free(b); free(own);
    TC_HIP(c, e);
    return TCGPU_OK;
}

/* profiling: the neighbour-list lengths of the last list pass, one per local slot (TC_XNONE: not listed) */
extern "C" int tcgpu_debug_xlcnt(tcgpu_ctx *c, uint32_t *out, int64_t n)
{
    if (!c || !out || !c->xlcnt || !c->xlist_valid || n > c->nloc) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    TC_HIP(c, hipMemcpy(out, c->xlcnt, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ options / timing */

extern "C" int tcgpu_set_option(tcgpu_ctx *c, const char *name, double value)
{
    if (!c || !name) return TCGPU_ERR_ARG;
    if (!strcmp(name, "stats")) c->want_stats = value != 0;
    else if (!strcmp(name, "timing")) c->timing = value != 0;
    else if (!strcmp(name, "level_shift")) c->level_shift = (int)value;
    else if (!strcmp(name, "level_scale")) c->level_scale = value > 0 ? value : 0.0;     /* 0: automatic */
    else if (!strcmp(name, "ablate")) c->ablate = (int)value;
    else if (!strcmp(name, "blocks_per_cu")) c->blocks_per_cu = (int)value;   /* profiling: cap the persistent grid */
    else if (!strcmp(name, "fuse")) c->fuse = value != 0;
    else if (!strcmp(name, "sweep")) c->sweep_mode = (int)value;         /* 0: the reference's order and roundings, lists from k_iter (default); 2: the same, stand-alone; 1: f64 sums, rounded once */
    else if (!strcmp(name, "pf_mode")) { c->pf_mode = (int)value; c->pf_valid = 0; }
    else if (!strcmp(name, "xsweep_shift")) c->xsweep_shift = (int)value;
    else if (!strcmp(name, "xsweep_kernel")) c->xsweep_kernel = (int)value;
    else if (!strcmp(name, "curl_literal")) c->curl_literal = value != 0;
    else if (!strcmp(name, "no_records")) c->no_records = value != 0;    /* tests: the fall-back of k_prec / k_cprec */
    else if (!strcmp(name, "rows")) { c->rows = value != 0; c->mirror_valid = 0; if (!c->rows) c->lmax_rm = c->lmax_rm0 = 0; }
    else if (!strcmp(name, "force_comm")) c->force_comm = value != 0;   /* tests: 1-rank RCCL communicator */
    else if (!strcmp(name, "debug_fail_rank")) c->debug_fail_rank = (int)value;   /* tests: rank value - 1 fails before the ghost exchange */
    else if (!strcmp(name, "ghost_exchange")) c->ghost_mode = (int)value; /* 0: position all-gather every pass, 1: whichever is cheaper (default), 2: always ghosts */
    else if (!strcmp(name, "lmax")) {
        if (c->n > 0) TC_FAIL(c, TCGPU_ERR_ARG, "lmax must be set before tcgpu_upload_particles");
        c->lmax_override = (int)value;
    } else TC_FAIL(c, TCGPU_ERR_ARG, "unknown option %s", name);
    return TCGPU_OK;
}

extern "C" int tcgpu_phase_times(tcgpu_ctx *c, const char **names, double *seconds, int64_t *launches, int *n, int reset)
{
    if (!c || !n) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    int m = *n < PH_COUNT ? *n : PH_COUNT;
    for (int i = 0; i < m; i++) {
        if (names) names[i] = PHASE_NAMES[i];
        if (seconds) seconds[i] = c->ph_sec[i];
        if (launches) launches[i] = c->ph_launch[i];
    }
    *n = m;
    if (reset) { memset(c->ph_sec, 0, sizeof(c->ph_sec)); memset(c->ph_launch, 0, sizeof(c->ph_launch)); }
    return TCGPU_OK;
}

/* bytes this rank has received in collectives since the last reset (sharded contexts; 0 on a single rank) */
extern "C" double tcgpu_comm_bytes(tcgpu_ctx *c, int reset)
{
    if (!c) return 0;
    const double b = c->comm_bytes;
    if (reset) c->comm_bytes = 0;
    return b;
}

/* size of the last local set, of the own range, and passes repeated on the full set so far */
extern "C" int tcgpu_local_set_info(tcgpu_ctx *c, int64_t *nloc, int64_t *nown, int32_t *retries)
{
    if (!c) return TCGPU_ERR_ARG;
    if (nloc) *nloc = c->nloc;
    if (nown) *nown = c->nown;
    if (retries) *retries = c->margin_retry;
    return TCGPU_OK;
}
