/*
 * tcgpu.h -- C ABI of libtcgpu: the MI355X (gfx950) implementation of Toycluster's
 * SPH-density / WVT-relaxation path.
 *
 * Drop-in boundary.  The reference exposes this path as
 * argument-less C functions working on process globals (P, SphP, Param, Halo):
 *
 *   void Regularise_sph_particles(void)   src/proto.h:25, src/wvt_relax.c:25
 *   void Find_sph_quantities(void)        src/proto.h:17, src/sph.c:13
 *   void Bfld_from_rotA_SPH(void)         src/proto.h:23, src/sph.h:2, src/sph.c:216
 *   float Global_density_model(int)       src/proto.h:46, src/wvt_relax.c:227
 *   void Sort_Particles_By_Peano_Key()    src/peano.h:5,  src/peano.c:46
 *   peanoKey Peano_Key(double,double,double)  src/peano.h:6, src/peano.c:128
 *   int  Find_ngb_tree(ipart, hsml, list) src/tree.h:2,   src/tree.c:25
 *   float Guess_hsml(ipart, DesNumNgb)    src/tree.h:5,   src/tree.c:113
 *
 * libtcgpu provides the same operations on an explicit context that owns the device
 * buffers; host arrays stay caller-owned.  toycluster_amd/host/tc_shim.c maps the
 * reference's symbol names and globals onto these entry points.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * tcgpu_status; no function calls exit(); tcgpu_last_error() gives the message.
 * Not re-entrant per context (same as the reference); one context per GPU.
 */
#ifndef TCGPU_H
#define TCGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* -DTCGPU_SPH_CUBIC_SPLINE: the library build that restates the reference's -DSPH_CUBIC_SPLINE build (Makefile:25:
 * M4 kernel, 50 neighbours; libtcgpu_m4.so, `make -C toycluster_amd/csrc m4`).  A compile-time choice there and here. */
#ifdef TCGPU_SPH_CUBIC_SPLINE
#define TCGPU_DESNNGB 50       /* src/globals.h:42 */
#define TCGPU_NGBMAX  400      /* src/globals.h:44 */
#else
#define TCGPU_DESNNGB 295      /* src/globals.h:48 */
#define TCGPU_NGBMAX  2360     /* src/globals.h:50 */
#endif
#define TCGPU_NUMITER 64       /* src/wvt_relax.c:7 */
#define TCGPU_MAXLOG  (TCGPU_NUMITER + 2)

typedef enum {
    TCGPU_OK = 0,
    TCGPU_ERR_HIP = -1,          /* a HIP runtime call failed */
    TCGPU_ERR_ARG = -2,          /* bad argument / call order */
    TCGPU_ERR_NOMEM = -3,
    TCGPU_ERR_NONFINITE = -4,    /* reference: Assert(isfinite(hsml)), src/sph.c:28 */
    TCGPU_ERR_COORD_RANGE = -5,  /* reference: Assert x in [0,1], src/peano.c:130-132 */
    TCGPU_ERR_NO_CONVERGENCE = -6, /* hsml loop guard tripped (the reference would spin forever) */
    TCGPU_ERR_COMM = -7,
    TCGPU_ERR_OVERFLOW = -8      /* internal table overflow (orphan list) */
} tcgpu_status;

/* Scalars of the reference's `Param` read by this path (src/globals.h:94-121). */
typedef struct {
    double boxsize;      /* Param.Boxsize */
    double mpart_gas;    /* Param.Mpart[0] */
    double mtotal;       /* Param.Mtotal (only `< 1e5` is tested, src/wvt_relax.c:53) */
    double bfld_eta;     /* Param.Bfld_Eta (unused by the path itself) */
    int32_t nhalos;      /* Param.Nhalos */
    int32_t reserved;
} tcgpu_params;

/* Scalars of the reference's `Halo[i]` read by this path (src/globals.h:132-159). */
typedef struct {
    double mass_gas;     /* Halo[i].Mass[0]; 0 => skipped (src/wvt_relax.c:237) */
    double d_com[3];     /* Halo[i].D_CoM */
    double rho0;         /* Halo[i].Rho0 */
    double beta;         /* Halo[i].Beta */
    double rcore;        /* Halo[i].Rcore */
    double rcut;         /* Halo[i].Rcut */
    int32_t have_cuspy;  /* Halo[i].Have_Cuspy */
    int32_t reserved;
    /* The reference's -DDOUBLE_BETA_COOL_CORES build (src/setup.c:604-612) adds a second, narrower beta = 2/3 component
     * to the profile of halos with Have_Cuspy: rho0_cc = Rho0 * Param.Rho0_Fac, rc_cc = Rcore / Param.Rc_Fac.
     * rho0_cc == 0 (every halo of the reference's default build) switches the term off. */
    double rho0_cc;
    double rc_cc;
} tcgpu_halo;

/* One line of the reference's convergence log (src/wvt_relax.c:91-92). */
typedef struct {
    int32_t it;
    int32_t reserved;
    double err_max, err_mean, err_diff, step;
} tcgpu_iterlog;

/* Work counters of the last density pass (for roofline accounting). */
typedef struct {
    double queries_per_particle;      /* ball queries (src/sph.c:40) */
    double solver_iters_per_particle; /* Find_hsml iterations (src/sph.c:96) */
    double pair_evals_per_particle;   /* list entries visited by the solver */
    double candidates_per_particle;   /* candidate particles tested by the ball queries */
} tcgpu_density_stats;

typedef struct tcgpu_ctx tcgpu_ctx;

/* ---- life cycle ---------------------------------------------------------------- */
int  tcgpu_create(tcgpu_ctx **ctx, int device);
void tcgpu_destroy(tcgpu_ctx *ctx);
const char *tcgpu_last_error(const tcgpu_ctx *ctx);
const char *tcgpu_version(void);

/* ---- model + particles (replaces the globals Param/Halo/P/SphP) ------------------ */
int tcgpu_set_model(tcgpu_ctx *ctx, const tcgpu_params *par, const tcgpu_halo *halos);
/* pos: xyz interleaved f32[3n] in [0,boxsize]; id may be NULL (=> 1..n); hsml may be
 * NULL (=> 0, the reference's zero-initialised SphP, src/setup.c:248-250). */
int tcgpu_upload_particles(tcgpu_ctx *ctx, int64_t n, const float *pos, const int32_t *id,
                           const float *hsml);
/* Any output pointer may be NULL.  Particles come back in Peano order (src/peano.c:85-126). */
int tcgpu_download_particles(tcgpu_ctx *ctx, float *pos, int32_t *id, float *hsml, float *rho,
                             float *varhsmlfac, float *rho_model);
int64_t tcgpu_num_particles(const tcgpu_ctx *ctx);

/* ---- the path -------------------------------------------------------------------- */
/* src/peano.c:46-126: key, sort, permute all per-particle arrays. */
int tcgpu_sort_particles_by_peano_key(tcgpu_ctx *ctx);
/* Sorted 128-bit keys as (hi,lo) halves, after a sort.  src/peano.c:70 (P[].Key). */
int tcgpu_download_keys(tcgpu_ctx *ctx, uint64_t *key_hi, uint64_t *key_lo);
/* Peano_Key for arbitrary points in [0,1]^3 evaluated ON THE DEVICE (known-answer tests).
 * xyz: f64[3n].  src/peano.c:128-203. */
int tcgpu_peano_keys(tcgpu_ctx *ctx, int64_t n, const double *xyz, uint64_t *key_hi, uint64_t *key_lo);
/* src/sph.c:13-75: sort + neighbour structure + hsml/rho/varHsmlFac for every particle. */
int tcgpu_find_sph_quantities(tcgpu_ctx *ctx);
int tcgpu_last_density_stats(tcgpu_ctx *ctx, tcgpu_density_stats *out);
/* src/wvt_relax.c:227-256 for every particle (current order). */
int tcgpu_global_density_model(tcgpu_ctx *ctx, float *rho_model_out);
/* src/tree.c:25-111 for one particle of the CURRENT (sorted) order; requires a prior
 * tcgpu_find_sph_quantities() or tcgpu_build_neighbour_index().  list has TCGPU_NGBMAX
 * slots and is returned in ascending index order; *count <= TCGPU_NGBMAX. */
int tcgpu_find_ngb(tcgpu_ctx *ctx, int64_t ipart, float hsml, int32_t *list, int32_t *count);
int tcgpu_build_neighbour_index(tcgpu_ctx *ctx);
/* src/tree.c:113-121 evaluated for every particle (the first-pass hsml guess, before doubling). */
int tcgpu_guess_hsml(tcgpu_ctx *ctx, float *guess_out);
/* One WVT sweep (src/wvt_relax.c:106-214) at the given step on the current order.
 * hsml_wvt[n] / delta[3n] may be NULL (and must be on sharded contexts: own-range arrays); move!=0 applies
 * src/wvt_relax.c:193-213. */
int tcgpu_wvt_step(tcgpu_ctx *ctx, double step, float *hsml_wvt, float *delta, int move);
/* First half of one loop body (src/wvt_relax.c:66-87): Find_sph_quantities() + the error sums.
 * One WVT "step" of the benchmark = tcgpu_density_error() + tcgpu_wvt_step(move=1). */
int tcgpu_density_error(tcgpu_ctx *ctx, double *err_mean, double *err_max);
/* Sharded contexts: the passes leave results in the rank's own index range; every call that returns or takes
 * per-particle arrays (downloads, find_ngb, bfld, ...) first completes them on all ranks and brings the particles
 * into Peano order ("presentation"), so it is COLLECTIVE: all ranks must make it, in the same order. */
/* src/wvt_relax.c:25-225.  max_iter < 0 => NUMITER.  log has TCGPU_MAXLOG slots. */
int tcgpu_regularise_sph_particles(tcgpu_ctx *ctx, int max_iter, tcgpu_iterlog *log, int32_t *nlog);
/* src/sph.c:216-300.  apot: f32[3n] in the CURRENT order; bfld out f32[3n]. */
int tcgpu_bfld_from_rotA_sph(tcgpu_ctx *ctx, const float *apot, float *bfld);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) ------------------------------ */
/* 128-byte RCCL unique id, created on rank 0 and distributed by the host program. */
int tcgpu_comm_unique_id(uint8_t id[128]);
int tcgpu_comm_init(tcgpu_ctx *ctx, int rank, int nranks, const uint8_t id[128]);

/* Testing only: tie `nranks` contexts of ONE process (one host thread each, any devices -- also all on
 * the same GPU) into a loopback communicator whose collectives are barriers + device-to-device
 * copies with the shard arithmetic of the RCCL path.  Lets a single-GPU box verify the sharded
 * control flow.  Must precede tcgpu_upload_particles on every context. */
int tcgpu_comm_init_loopback(tcgpu_ctx **ctxs, int nranks);

/* ---- tuning / introspection ------------------------------------------------------- */
/* Options (defaults in brackets).  None of them changes results beyond the f64 summation order.
 *   "stats" [0]       keep per-particle work counters for tcgpu_last_density_stats
 *   "timing" [0]      record HIP events per phase for tcgpu_phase_times
 *   "fuse" [1]        fused density + sweep kernel; 0 = one plain kernel per reference loop
 *   "sweep" [0]       the WVT displacement (wvt_relax.c:126-171): 0 = the reference's summation order and roundings -- f32
 *                     accumulator, one rounding per neighbour, neighbours in ascending index -- bit for bit (the fused
 *                     kernel lists each particle's neighbours in that order while it solves the densities, a second
 *                     kernel evaluates the lists); 2 = the same sums by a stand-alone kernel (also taken automatically
 *                     when the per-particle lists would not fit the device); 1 = f64 sums over 64 lanes rounded once
 *                     (~1e-6 |delta| off per iteration; fused into the density kernel, the fastest)
 *   "xsweep_kernel" [0] tests: 1 = sweep = 0/2 through the one-lane-per-particle kernel on the (x, y, z) cell table (an
 *                     independent implementation of the same sums)
 *   "xsweep_shift" [0] tuning: added to the cell level of the stand-alone exact sweep
 *   "debug_fail_rank" [0] tests: rank (value - 1) reports a failure of its own before the ghost exchange
 *   "rows" [1]        row-run candidate streaming over the row-major mirror (set before upload)
 *   "level_shift" [1] cell level finer than the smoothing length by this many octree levels
 *   "level_scale" [auto] the radius is multiplied by this before its level is chosen (speed only; auto: 1.5 with the default
 *                     sweep, whose ordered cell walk likes coarser leaves, 2^(1/4) with "sweep" = 1 / 2)
 *   "pf_mode" [0]     tests: how the curve-ordered cell starts are built -- 0 automatic, 1 every level by one scan, 2 the three
 *                     deepest levels block by block (what a sharded rank's thin local set gets); same table either way
 *   "lmax" [auto]     deepest cell-table level (set before upload)
 *   "force_comm" [0]  tests: run the RCCL calls with a 1-rank communicator
 *   "curl_literal" [0] tests: the curl's literal pair-by-pair path (the NGBMAX-overflow fall-back) for every particle
 *   "no_records" [0]  tests: treat every per-particle query record as unusable (the plain per-query code / the curl's
 *                     literal path then solve every particle)
 *   "ghost_exchange" [1] sharded contexts: 1 = a rank receives only its ghost particles whenever that moves fewer bytes
 *                     than every position to every rank, 2 = always, 0 = never (position all-gather); same results
 *   "blocks_per_cu" [0] profiling: cap on the co-resident blocks per CU of the persistent kernels (0 = all)
 *   "ablate" [0]      only honoured by the profiling build libtcgpu_ablate.so (results invalid) */
int tcgpu_set_option(tcgpu_ctx *ctx, const char *name, double value);
/* Seconds spent on the device in each phase since the last reset (HIP events on the
 * library's own stream): names/values arrays of length *n (in: capacity, out: used). */
int tcgpu_phase_times(tcgpu_ctx *ctx, const char **names, double *seconds, int64_t *launches, int *n, int reset);
void *tcgpu_stream(tcgpu_ctx *ctx);
/* Sharded contexts: bytes this rank has received in collectives since the last reset (0 on a single rank). */
double tcgpu_comm_bytes(tcgpu_ctx *ctx, int reset);
/* Particles in the last pass's local set (own range + ghost shell; all n on a single rank), in the own range, and
 * the number of passes that had to be repeated on the full set because a query left the ghost margin. */
int tcgpu_local_set_info(tcgpu_ctx *ctx, int64_t *nloc, int64_t *nown, int32_t *retries);

#ifdef __cplusplus
}
#endif
#endif /* TCGPU_H */
