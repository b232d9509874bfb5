/*
 * api.hip -- C ABI of libtcgpu (include/tcgpu.h): context, buffers, and the host-side
 * control flow of the path (the loop of src/wvt_relax.c:61-218 and the driver part of
 * src/sph.c:13-17).  All device work is enqueued on the context's own HIP stream.
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
                                            "curl", "comm", "mirror"};

struct rccl_api {
    void *h;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
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
    pthread_barrier_t bar;
    void *bufs[16];
    double red[16][8];
};


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

extern "C" const char *tcgpu_version(void) { return "tcgpu 0.1 (gfx950)"; }

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
    ok = ok && hipMalloc(&c->red, sizeof(double) * (4 * TC_RED_BLOCKS + 32)) == hipSuccess;
    ok = ok && hipHostMalloc(&c->h_red, sizeof(double) * 32) == hipSuccess;
    ok = ok && hipMalloc(&c->flags, sizeof(int) * 8) == hipSuccess;
    ok = ok && hipHostMalloc(&c->h_flags, sizeof(int) * 8) == hipSuccess;
    ok = ok && hipMalloc(&c->orphans, sizeof(uint32_t) * TC_MAX_ORPHANS) == hipSuccess;
    ok = ok && hipMalloc(&c->norph, sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->work_ctr, 8 * 16 * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->ngb_cnt, sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc(&c->spill, sizeof(double) * (size_t)TC_MAX_PERSISTENT_BLOCKS * TC_WAVES_PER_BLOCK
                                        * (2 * TC_NGBMAX)) == hipSuccess;
    c->fuse = 1;
    c->rows = 1;
    c->level_shift = 1;                          /* cells of h/4..h/2: fewest candidates per query (tools/fuse_stats.py) */
    c->level_scale = 1.189207115002721;           /* 2^(1/4): cells of h/3.4..h/1.7, measured best (tools/shift_probe.py) */
    ok = ok && hipDeviceGetAttribute(&c->num_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess;
    ok = ok && hipMemset(c->flags, 0, sizeof(int) * 8) == hipSuccess;
    ok = ok && hipMemset(c->norph, 0, sizeof(int)) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (!ok) { tcgpu_destroy(c); return TCGPU_ERR_NOMEM; }
    *out = c;
    return TCGPU_OK;
}

static void free_particles(tcgpu_ctx *c)
{
    for (int b = 0; b < 2; b++) {
        hipFree(c->pos4[b]); hipFree(c->id[b]); hipFree(c->hsml[b]); hipFree(c->rho[b]);
        hipFree(c->vhf[b]); hipFree(c->rhom[b]);
        c->pos4[b] = nullptr; c->id[b] = nullptr; c->hsml[b] = c->rho[b] = c->vhf[b] = c->rhom[b] = nullptr;
    }
    hipFree(c->apot); hipFree(c->bfld); hipFree(c->key); hipFree(c->key_sorted); hipFree(c->idx);
    hipFree(c->idx_sorted); hipFree(c->sort_tmp); hipFree(c->cells); hipFree(c->guess);
    hipFree(c->hwvt); hipFree(c->delta); hipFree(c->stats); hipFree(c->ngb_buf); hipFree(c->ustep); hipFree(c->rhom_next);
    c->ustep = nullptr; c->rhom_next = nullptr;
    c->apot = c->bfld = nullptr; c->key = c->key_sorted = nullptr; c->idx = c->idx_sorted = nullptr;
    c->sort_tmp = nullptr; c->cells = nullptr; c->guess = c->hwvt = c->delta = nullptr;
    c->stats = nullptr; c->ngb_buf = nullptr;
    hipFree(c->cum); hipFree(c->scan_tmp); hipFree(c->mirror); hipFree(c->mirror_idx);
    c->cum = nullptr; c->scan_tmp = nullptr; c->mirror = nullptr; c->mirror_idx = nullptr;
    c->cum_alloc = c->mirror_alloc = 0; c->mirror_valid = 0;
    c->cap = 0; c->n = 0; c->ncells_alloc = 0;
}

extern "C" void tcgpu_destroy(tcgpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_particles(c);
    hipFree(c->d_halo); hipFree(c->red); hipHostFree(c->h_red); hipFree(c->flags); hipHostFree(c->h_flags);
    hipFree(c->orphans); hipFree(c->norph); hipFree(c->work_ctr); hipFree(c->ngb_cnt); hipFree(c->spill);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->comm);
    for (int i = 0; i < c->caprecs; i++) { hipEventDestroy(c->recs[i].a); hipEventDestroy(c->recs[i].b); }
    free(c->recs);
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
    }
    hipError_t e = hipMemcpy(c->d_halo, h, sizeof(tc_halo_dev) * par->nhalos, hipMemcpyHostToDevice);
    free(h);
    TC_HIP(c, e);
    c->have_model = 1;
    return TCGPU_OK;
}

static int pick_lmax(int64_t n)
{
    /* Deepest table level: one below the mean inter-particle level, round(log8 n) + 1 (8 at N = 2e6, 9 at
     * 1.6e7, 10 at 1e8).  The densest few per cent of the particles would like one level more; clamped to
     * this one they see about twice the candidates, which costs less than clearing, scanning and mirroring a
     * table eight times larger every iteration (measured at N = 2e6: -0.33 ms per iteration, tools/shift_probe.py). */
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
            TC_HIP(c, hipMalloc(&c->pos4[b], cap * sizeof(float4)));
            TC_HIP(c, hipMalloc(&c->id[b], cap * sizeof(int32_t)));
            TC_HIP(c, hipMalloc(&c->hsml[b], cap * sizeof(float)));
            TC_HIP(c, hipMalloc(&c->rho[b], cap * sizeof(float)));
            TC_HIP(c, hipMalloc(&c->vhf[b], cap * sizeof(float)));
            TC_HIP(c, hipMalloc(&c->rhom[b], cap * sizeof(float)));
        }
        TC_HIP(c, hipMalloc(&c->key, cap * sizeof(tc_u128)));
        TC_HIP(c, hipMalloc(&c->key_sorted, cap * sizeof(tc_u128)));
        TC_HIP(c, hipMalloc(&c->idx, cap * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->idx_sorted, cap * sizeof(uint32_t)));
        if (tc_sort_temp_bytes(cap, &c->sort_tmp_bytes)) TC_FAIL(c, TCGPU_ERR_HIP, "radix sort temp query failed");
        TC_HIP(c, hipMalloc(&c->sort_tmp, c->sort_tmp_bytes ? c->sort_tmp_bytes : 16));
        TC_HIP(c, hipMalloc(&c->guess, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->hwvt, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->delta, 3 * cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->ustep, 3 * cap * sizeof(double)));
        TC_HIP(c, hipMalloc(&c->rhom_next, cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->stats, 4 * cap * sizeof(uint32_t)));
        TC_HIP(c, hipMalloc(&c->ngb_buf, cap * sizeof(int32_t)));
        c->cap = need;
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
        if (nslot > c->mirror_alloc) {
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
    return 0;
}

static void set_shard(tcgpu_ctx *c)
{
    c->shard_len = (c->n + c->nranks - 1) / c->nranks;
    if (c->shard_len < 1) c->shard_len = 1;
}

extern "C" int tcgpu_upload_particles(tcgpu_ctx *c, int64_t n, const float *pos, const int32_t *id, const float *hsml)
{
    if (!c || !pos || n <= 0) return TCGPU_ERR_ARG;
    if (n >= (1LL << 31) - 64) TC_FAIL(c, TCGPU_ERR_ARG, "n=%lld exceeds the 32-bit index range of the path", (long long)n);
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    int rc = ensure_capacity(c, n);
    if (rc) return rc;
    c->n = n;
    c->cur = 0;
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
    hipError_t e1 = hipMemcpyAsync(c->pos4[0], tmp, cap * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(c->id[0], tid, cap * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    hipError_t e3 = hipStreamSynchronize(c->stream);
    free(tmp); free(tid);
    TC_HIP(c, e1); TC_HIP(c, e2); TC_HIP(c, e3);
    TC_HIP(c, hipMemsetAsync(c->hsml[0], 0, cap * sizeof(float), c->stream));
    if (hsml) TC_HIP(c, hipMemcpyAsync(c->hsml[0], hsml, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TC_HIP(c, hipMemsetAsync(c->rho[0], 0, cap * sizeof(float), c->stream));
    TC_HIP(c, hipMemsetAsync(c->vhf[0], 0, cap * sizeof(float), c->stream));
    TC_HIP(c, hipMemsetAsync(c->rhom[0], 0, cap * sizeof(float), c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    c->keys_valid = 0;
    c->index_valid = 0; c->mirror_valid = 0;
    c->ustep_valid = 0;
    c->need_guess = 1;
    if (hsml) {                                   /* warm start: the guess is only read where hsml == 0 */
        c->need_guess = 0;
        for (int64_t i = 0; i < n; i++)
            if (hsml[i] == 0) { c->need_guess = 1; break; }
    }
    return TCGPU_OK;
}

extern "C" int tcgpu_download_particles(tcgpu_ctx *c, float *pos, int32_t *id, float *hsml, float *rho, float *vhf,
                                        float *rhom)
{
    if (!c || c->n <= 0) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)c->n;
    int b = c->cur;
    if (pos) {
        float4 *tmp = (float4 *)malloc(n * sizeof(float4));
        if (!tmp) return TCGPU_ERR_NOMEM;
        hipError_t e = hipMemcpy(tmp, c->pos4[b], n * sizeof(float4), hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < n; i++) { pos[3 * i] = tmp[i].x; pos[3 * i + 1] = tmp[i].y; pos[3 * i + 2] = tmp[i].z; }
        free(tmp);
        TC_HIP(c, e);
    }
    if (id) TC_HIP(c, hipMemcpy(id, c->id[b], n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (hsml) TC_HIP(c, hipMemcpy(hsml, c->hsml[b], n * sizeof(float), hipMemcpyDeviceToHost));
    if (rho) TC_HIP(c, hipMemcpy(rho, c->rho[b], n * sizeof(float), hipMemcpyDeviceToHost));
    if (vhf) TC_HIP(c, hipMemcpy(vhf, c->vhf[b], n * sizeof(float), hipMemcpyDeviceToHost));
    if (rhom) TC_HIP(c, hipMemcpy(rhom, c->rhom[b], n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ flags */

/* `reduced`: the four error flags already maximised over the ranks (multi-rank contexts: every rank must take
 * the same exit, or the ranks that carry on would wait forever in the next collective); NULL = this rank's own. */
static int check_flags(tcgpu_ctx *c, const double *reduced = nullptr)
{
    TC_HIP(c, hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 8, hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    int f[8];
    memcpy(f, c->h_flags, sizeof(f));
    if (f[0] || f[1] || f[2] || f[3]) TC_HIP(c, hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
    if (reduced)
        for (int q = 0; q < 4; q++) f[q] = reduced[q] != 0;
    if (f[1]) TC_FAIL(c, TCGPU_ERR_COORD_RANGE, "coordinate outside [0,boxsize] (reference: peano.c:130-132 Assert)");
    if (f[0]) TC_FAIL(c, TCGPU_ERR_NONFINITE, "hsml not finite (reference: sph.c:28 Assert)");
    if (f[3]) TC_FAIL(c, TCGPU_ERR_OVERFLOW, "more than %d particles sit exactly on the upper box face", TC_MAX_ORPHANS);
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
    g_rccl.GroupStart = (ncclResult_t(*)(void))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (ncclResult_t(*)(void))dlsym(h, "ncclGroupEnd");
    g_rccl.CommDestroy = (ncclResult_t(*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.GroupStart
        || !g_rccl.GroupEnd) {
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

/* In-place all-gather of one shard-partitioned array (elements of `esize` bytes). */
static int allgather_inplace(tcgpu_ctx *c, void *base, size_t esize)
{
    size_t bytes = (size_t)c->shard_len * esize;
    if (c->loop) {
        tc_loop_comm *L = c->loop;
        TC_HIP(c, hipStreamSynchronize(c->stream));
        L->bufs[c->rank] = base;
        pthread_barrier_wait(&L->bar);
        for (int p = 0; p < L->nranks; p++)
            if (p != c->rank)
                TC_HIP(c, hipMemcpyAsync((char *)L->bufs[p] + (size_t)c->rank * bytes,
                                         (const char *)base + (size_t)c->rank * bytes, bytes,
                                         hipMemcpyDeviceToDevice, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        pthread_barrier_wait(&L->bar);
        return 0;
    }
    ncclResult_t r = g_rccl.AllGather((const char *)base + (size_t)c->rank * bytes, base, bytes, ncclInt8,
                                      (ncclComm_t)c->comm, c->stream);
    if (r != ncclSuccess) TC_FAIL(c, TCGPU_ERR_COMM, "ncclAllGather failed (%d)", (int)r);
    return 0;
}

/* all-reduce of the scalars of one pass, in place on the device: buf[0..2] summed (error sum, particle count,
 * spare), buf[3..7] maximised (largest error and the four error flags as 0/1, see check_flags) */
static int allreduce_pass_scalars(tcgpu_ctx *c, double *buf)
{
    if (c->loop) {
        tc_loop_comm *L = c->loop;
        double h[8];
        TC_HIP(c, hipMemcpyAsync(h, buf, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(L->red[c->rank], h, sizeof(h));
        pthread_barrier_wait(&L->bar);
        double o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < L->nranks; p++) {
            for (int q = 0; q < 3; q++) o[q] += L->red[p][q];
            for (int q = 3; q < 8; q++) o[q] = fmax(o[q], L->red[p][q]);
        }
        pthread_barrier_wait(&L->bar);
        TC_HIP(c, hipMemcpyAsync(buf, o, sizeof(o), hipMemcpyHostToDevice, c->stream));
        TC_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    g_rccl.GroupStart();
    ncclResult_t r1 = g_rccl.AllReduce(buf, buf, 3, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream);
    ncclResult_t r2 = g_rccl.AllReduce(buf + 3, buf + 3, 5, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream);
    ncclResult_t r3 = g_rccl.GroupEnd();
    if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess) TC_FAIL(c, TCGPU_ERR_COMM, "ncclAllReduce failed");
    return 0;
}

/* multi-rank contexts: agree on the error flags (maximum over the ranks), then check them; every rank
 * returns the same status.  Single rank: the plain check. */
static int check_flags_collective(tcgpu_ctx *c)
{
    if (!(c->comm || c->loop)) return check_flags(c);
    double *buf = c->red + 4 * TC_RED_BLOCKS + 8;
    TC_HIP(c, hipMemsetAsync(buf, 0, 8 * sizeof(double), c->stream));
    int rc = tc_launch_flags_to_f64(c, buf + 4);
    if (rc) return rc;
    tc_phase_begin(c, PH_COMM);
    rc = allreduce_pass_scalars(c, buf);
    tc_phase_end(c);
    if (rc) return rc;
    TC_HIP(c, hipMemcpyAsync(c->h_red + 8, buf, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    return check_flags(c, c->h_red + 8 + 4);
}

/* testing: tie `nranks` contexts of this process into a loopback communicator (one thread each) */
extern "C" int tcgpu_comm_init_loopback(tcgpu_ctx **ctxs, int nranks)
{
    if (!ctxs || nranks < 1 || nranks > 16) return TCGPU_ERR_ARG;
    tc_loop_comm *L = (tc_loop_comm *)calloc(1, sizeof(*L));
    if (!L) return TCGPU_ERR_NOMEM;
    L->nranks = nranks;
    pthread_barrier_init(&L->bar, nullptr, (unsigned)nranks);
    for (int r = 0; r < nranks; r++) {
        if (!ctxs[r] || ctxs[r]->n > 0) { free(L); return TCGPU_ERR_ARG; }
        ctxs[r]->loop = L;              /* shared; intentionally leaked with the last context (test only) */
        ctxs[r]->rank = r;
        ctxs[r]->nranks = nranks;
    }
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ sort / index */

extern "C" int tcgpu_sort_particles_by_peano_key(tcgpu_ctx *c)
{
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = tc_launch_keys(c))) return rc;
    tc_phase_begin(c, PH_SORT);
    /* large sets: radix passes only over the Hilbert levels that separate particles in practice, five below
     * the deepest cell-table level (3 bits per level); the fix-up orders whatever still ties, exactly */
    int sort_levels = c->lmax + 5;
    if (sort_levels > 21) sort_levels = 21;
    int s = tc_sort_pairs_u128(c->sort_tmp, c->sort_tmp_bytes, c->key, c->key_sorted, c->idx, c->idx_sorted,
                               (size_t)c->n, 3 * sort_levels + 1, c->stream);
    tc_phase_end(c);
    if (s) TC_FAIL(c, TCGPU_ERR_HIP, "radix sort failed");
    if ((rc = tc_launch_permute(c))) return rc;
    c->keys_valid = 1;
    c->index_valid = 0; c->mirror_valid = 0;
    c->ustep_valid = 0;
    return TCGPU_OK;
}

extern "C" int tcgpu_download_keys(tcgpu_ctx *c, uint64_t *hi, uint64_t *lo)
{
    if (!c || c->n <= 0 || !c->keys_valid) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)c->n;
    tc_u128 *t = (tc_u128 *)malloc(n * sizeof(tc_u128));
    if (!t) return TCGPU_ERR_NOMEM;
    hipError_t e = hipMemcpy(t, c->key_sorted, n * sizeof(tc_u128), hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        for (size_t i = 0; i < n; i++) {
            if (hi) hi[i] = (uint64_t)(t[i] >> 64);
            if (lo) lo[i] = (uint64_t)t[i];
        }
    free(t);
    TC_HIP(c, e);
    return check_flags(c);
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

extern "C" int tcgpu_build_neighbour_index(tcgpu_ctx *c)
{
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc;
    if (!c->keys_valid && (rc = tcgpu_sort_particles_by_peano_key(c))) return rc;
    if ((rc = tc_launch_cells(c))) return rc;
    return check_flags(c);
}

/* ------------------------------------------------------------------ density pass */

static int density_stats(tcgpu_ctx *c)
{
    size_t n = (size_t)c->n, cap = (size_t)c->cap;
    uint32_t *h = (uint32_t *)malloc(4 * cap * sizeof(uint32_t));
    if (!h) return TCGPU_ERR_NOMEM;
    hipError_t e = hipMemcpy(h, c->stats, 4 * cap * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        int64_t lo = c->rank * c->shard_len, hi = std::min<int64_t>((c->rank + 1) * c->shard_len, (int64_t)n);
        double s[4] = {0, 0, 0, 0};
        for (int q = 0; q < 4; q++)
            for (int64_t i = lo; i < hi; i++) s[q] += h[q * cap + i];
        double m = (double)std::max<int64_t>(1, hi - lo);
        c->last_stats.queries_per_particle = s[0] / m;
        c->last_stats.solver_iters_per_particle = s[1] / m;
        c->last_stats.pair_evals_per_particle = s[2] / m;
        c->last_stats.candidates_per_particle = s[3] / m;
    }
    free(h);
    TC_HIP(c, e);
    return 0;
}

/* gather_all: also all-gather rho and varHsmlFac (needed once results are read back; inside the
 * relaxation loop other ranks only need the carried hsml for their next warm start) */
static int gather_sph(tcgpu_ctx *c, int gather_all)
{
    if (!(c->comm || c->loop)) return 0;
    tc_phase_begin(c, PH_COMM);
    if (c->comm) g_rccl.GroupStart();
    int r1 = allgather_inplace(c, c->hsml[c->cur], sizeof(float)), r2 = 0, r3 = 0;
    if (gather_all && !r1) {
        r2 = allgather_inplace(c, c->rho[c->cur], sizeof(float));
        if (!r2) r3 = allgather_inplace(c, c->vhf[c->cur], sizeof(float));
    }
    /* the group is closed whatever happened in it */
    if (c->comm && g_rccl.GroupEnd() != ncclSuccess && !(r1 || r2 || r3)) {
        snprintf(c->err, sizeof(c->err), "ncclGroupEnd failed");
        r1 = TCGPU_ERR_COMM;
    }
    tc_phase_end(c);
    return (r1 || r2 || r3) ? TCGPU_ERR_COMM : 0;
}

static int find_sph_quantities_nocheck(tcgpu_ctx *c, int need_guess, int with_wvt, int gather_all)
{
    int rc;
    if ((rc = tcgpu_sort_particles_by_peano_key(c))) return rc;
    if ((rc = tc_launch_cells(c))) return rc;
    if (need_guess && (rc = tc_launch_guess(c))) return rc;
    if (c->fuse) {
        /* one gather per particle serves the density solve and (with_wvt) the WVT sweep that
         * follows on the same positions; the sweep needs the model hsml up front */
        if (with_wvt && (rc = tc_launch_model_hsml(c))) return rc;
        if ((rc = tc_launch_mirror(c))) return rc;
        if ((rc = tc_launch_iter(c, with_wvt))) return rc;
        c->ustep_valid = with_wvt;
    } else if ((rc = tc_launch_density(c))) return rc;
    return gather_sph(c, gather_all);
}

extern "C" int tcgpu_find_sph_quantities(tcgpu_ctx *c)
{
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = find_sph_quantities_nocheck(c, c->need_guess, 0, 1);
    if (rc) return rc;
    rc = check_flags_collective(c);
    if (rc) return rc;
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
    if (!c || !out || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc = tc_launch_model(c, c->guess);       /* guess doubles as f32 scratch */
    if (rc) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    TC_HIP(c, hipMemcpy(out, c->guess, (size_t)c->n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

extern "C" int tcgpu_guess_hsml(tcgpu_ctx *c, float *out)
{
    if (!c || !out || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int rc;
    if (!c->keys_valid && (rc = tcgpu_sort_particles_by_peano_key(c))) return rc;
    if ((rc = tc_launch_guess(c))) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    TC_HIP(c, hipMemcpy(out, c->guess, (size_t)c->n * sizeof(float), hipMemcpyDeviceToHost));
    return check_flags(c);
}

extern "C" int tcgpu_find_ngb(tcgpu_ctx *c, int64_t ipart, float hsml, int32_t *list, int32_t *count)
{
    if (!c || !list || !count || ipart < 0 || ipart >= c->n) return TCGPU_ERR_ARG;
    if (!c->index_valid) TC_FAIL(c, TCGPU_ERR_ARG, "neighbour index not built");
    TC_HIP(c, hipSetDevice(c->device));
    int rc = tc_launch_find_ngb(c, (int)ipart, hsml);
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
        if ((rc = tc_launch_model_hsml(c))) return rc;
        if ((rc = tc_launch_commit_rhom(c))) return rc;
        if ((rc = tc_launch_wvt(c, step))) return rc;
    }
    if (move) {
        if ((rc = tc_launch_move(c))) return rc;
        if (c->comm || c->loop) {
            tc_phase_begin(c, PH_COMM);
            rc = allgather_inplace(c, c->pos4[c->cur], sizeof(float4));
            tc_phase_end(c);
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int tcgpu_wvt_step(tcgpu_ctx *c, double step, float *hsml_wvt, float *delta, int move)
{
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    if (!c->index_valid) TC_FAIL(c, TCGPU_ERR_ARG, "neighbour index not built (call tcgpu_find_sph_quantities first)");
    TC_HIP(c, hipSetDevice(c->device));
    int rc = wvt_step_nocheck(c, step, move);
    if (rc) return rc;
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    size_t n = (size_t)c->n;
    if (hsml_wvt) TC_HIP(c, hipMemcpy(hsml_wvt, c->hwvt, n * sizeof(float), hipMemcpyDeviceToHost));
    if (delta) TC_HIP(c, hipMemcpy(delta, c->delta, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

/* First half of one loop body of src/wvt_relax.c:61-92: density pass (sort, index, K5) and the
 * error sums K6.  Synchronises the stream (the caller needs errMean to decide the step). */
static int density_error_sync(tcgpu_ctx *c, int need_guess, double *err_mean, double *err_max)
{
    int rc;
    if ((rc = find_sph_quantities_nocheck(c, need_guess, 1, 0))) return rc;
    if ((rc = tc_launch_error(c))) return rc;
    double *fin = c->red + 4 * TC_RED_BLOCKS;                 /* {sum err, count, 0, max err} */
    const bool multi = c->comm || c->loop;
    if (multi) {
        /* the error flags ride along (fin[4..7], maximised): every rank sees the same flags and takes the
         * same exit -- a rank that returned alone would leave the others waiting in the next collective */
        if ((rc = tc_launch_flags_to_f64(c, fin + 4))) return rc;
        tc_phase_begin(c, PH_COMM);
        rc = allreduce_pass_scalars(c, fin);
        tc_phase_end(c);
        if (rc) return rc;
    }
    TC_HIP(c, hipMemcpyAsync(c->h_red, fin, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if ((rc = check_flags(c, multi ? c->h_red + 4 : nullptr))) return rc;      /* synchronises the stream */
    *err_mean = c->h_red[0] / c->h_red[1];                    /* wvt_relax.c:87 */
    *err_max = c->h_red[3];
    return 0;
}

extern "C" int tcgpu_density_error(tcgpu_ctx *c, double *err_mean, double *err_max)
{
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
    if (!c || c->n <= 0 || !c->have_model) return TCGPU_ERR_ARG;
    TC_HIP(c, hipSetDevice(c->device));
    int it = -1, nlog = 0;
    double step = 0.0085;                                     /* wvt_relax.c:51 */
    if (c->par.mtotal < 1e5) step /= 2;                       /* wvt_relax.c:53-54 */
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
    if ((rc = gather_sph(c, 1))) return rc;                   /* rho / varHsmlFac of the last pass, all ranks */
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    if (nlog_out) *nlog_out = nlog;
    if (c->want_stats) return density_stats(c);
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ curl */

extern "C" int tcgpu_bfld_from_rotA_sph(tcgpu_ctx *c, const float *apot, float *bfld)
{
    if (!c || !apot || !bfld || c->n <= 0) return TCGPU_ERR_ARG;
    if (!c->index_valid) TC_FAIL(c, TCGPU_ERR_ARG, "neighbour index not built (call tcgpu_find_sph_quantities first)");
    TC_HIP(c, hipSetDevice(c->device));
    size_t cap = (size_t)c->cap, n = (size_t)c->n;
    if (!c->apot) {
        TC_HIP(c, hipMalloc(&c->apot, 3 * cap * sizeof(float)));
        TC_HIP(c, hipMalloc(&c->bfld, 3 * cap * sizeof(float)));
    }
    TC_HIP(c, hipMemcpyAsync(c->apot, apot, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TC_HIP(c, hipStreamSynchronize(c->stream));
    int rc = tc_launch_curl(c);
    if (rc) return rc;
    if (c->comm || c->loop) {
        tc_phase_begin(c, PH_COMM);
        rc = allgather_inplace(c, c->bfld, 3 * sizeof(float));
        tc_phase_end(c);
        if (rc) return rc;
    }
    TC_HIP(c, hipStreamSynchronize(c->stream));
    tc_phase_collect(c);
    TC_HIP(c, hipMemcpy(bfld, c->bfld, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return TCGPU_OK;
}

/* ------------------------------------------------------------------ options / timing */

extern "C" int tcgpu_set_option(tcgpu_ctx *c, const char *name, double value)
{
    if (!c || !name) return TCGPU_ERR_ARG;
    if (!strcmp(name, "stats")) c->want_stats = value != 0;
    else if (!strcmp(name, "timing")) c->timing = value != 0;
    else if (!strcmp(name, "level_shift")) c->level_shift = (int)value;
    else if (!strcmp(name, "level_scale")) c->level_scale = value > 0 ? value : 1.0;
    else if (!strcmp(name, "ablate")) c->ablate = (int)value;
    else if (!strcmp(name, "fuse")) c->fuse = value != 0;
    else if (!strcmp(name, "rows")) { c->rows = value != 0; c->mirror_valid = 0; if (!c->rows) c->lmax_rm = 0; }
    else if (!strcmp(name, "force_comm")) c->force_comm = value != 0;   /* tests: 1-rank RCCL communicator */
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
