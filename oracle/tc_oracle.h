/*
 * tc_oracle.h -- CPU restatement of Toycluster's SPH-density / WVT-relaxation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (toycluster_amd/, include/,
 * the HIP library, the C host) may include, link, or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as a checker
 * or as the timed CPU baseline -- never as the thing shipped.
 *
 * PARITY STATUS: the reference path cannot be compiled in this image
 * (src/globals.h:21-24 includes <gsl/...>, libgsl is absent, and stand-ins are
 * not permitted), and the reference ships no tests or golden vectors.  The
 * oracle is therefore pinned ONLY by the Peano-key known answers recorded in
 * SURVEY.md section 8c (tests/golden/peano_kat.json).  Everything else is
 * "parity unpinned": a line-by-line arithmetic restatement, each function
 * citing the reference file:line it follows.
 *
 * Third-party algorithm restated: GSL gsl_heapsort_index (sort/sortind.c of
 * GNU GSL, version unpinned by the reference Makefile:84), called from
 * reference src/sort.c:192.
 */
#ifndef TC_ORACLE_H
#define TC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

/* -DORC_SPH_CUBIC_SPLINE = the reference's -DSPH_CUBIC_SPLINE build (Makefile:25; globals.h:40-52): M4 kernel with 50
 * neighbours in the density solve and the curl, no bias correction, WVT step 0.035.  Built as libtcoracle_m4.so. */
#ifdef ORC_SPH_CUBIC_SPLINE
#define ORC_DESNNGB 50           /* globals.h:42 */
#else
#define ORC_DESNNGB 295          /* globals.h:48 */
#endif
#define ORC_NNGBDEV 0.05         /* globals.h:49 */
#define ORC_NGBMAX (ORC_DESNNGB * 8)   /* globals.h:50 */
#define ORC_NUMITER 64           /* wvt_relax.c:7 */
#define ORC_ERRDIFF_LIMIT 0.01   /* wvt_relax.c:8 */
#define ORC_MAXLOG (ORC_NUMITER + 2)

typedef struct {
    double mass_gas;     /* Halo[i].Mass[0]; 0 => skipped (wvt_relax.c:237) */
    double d_com[3];     /* Halo[i].D_CoM */
    double rho0;         /* Halo[i].Rho0 */
    double beta;         /* Halo[i].Beta */
    double rcore;        /* Halo[i].Rcore */
    double rcut;         /* Halo[i].Rcut */
    int    have_cuspy;   /* Halo[i].Have_Cuspy (read only by the DOUBLE_BETA_COOL_CORES variant, orc_set_double_beta) */
    int    pad_;
} orc_halo;

/* The reference's -DDOUBLE_BETA_COOL_CORES build as a run-time switch of the oracle: Param.Rho0_Fac, Param.Rc_Fac
 * (globals.h:117-120); both 0 = the default build.  Process-wide, like the reference's Param. */
void orc_set_double_beta(double rho0_fac, double rc_fac);

typedef struct {
    int    it;
    double err_max, err_mean, err_diff, step;
} orc_iterlog;

typedef struct orc_state orc_state;

/* life cycle */
orc_state *orc_create(int npart, double boxsize, double mpart_gas, double mtotal,
                      int nhalos, const orc_halo *halos, int nthreads);
void orc_destroy(orc_state *s);

/* particle state (gas only).  pos is xyz-interleaved f32[3n] in [0,box]. */
void orc_set_particles(orc_state *s, const float *pos, const int32_t *id, const float *hsml);
void orc_get_particles(const orc_state *s, float *pos, int32_t *id, float *hsml, float *rho,
                       float *varhsmlfac, float *rho_model);
void orc_set_apot(orc_state *s, const float *apot3);
void orc_get_bfld(const orc_state *s, float *bfld3);
void orc_get_apot(const orc_state *s, float *apot3);

/* peano.c:128-203, 211-284 : keys as (hi,lo) 64-bit halves of the 128-bit key */
void orc_peano_key(double x, double y, double z, uint64_t *hi, uint64_t *lo);
void orc_reversed_peano_key(double x, double y, double z, uint64_t *hi, uint64_t *lo);

/* peano.c:46-126 : key, index heapsort, permutation of all per-particle arrays.
 * Optional outputs (may be NULL): sorted keys (hi,lo)[n], permutation perm[n]
 * (new index i holds old particle perm[i]). */
void orc_sort_by_peano_key(orc_state *s, uint64_t *key_hi, uint64_t *key_lo, int64_t *perm);

/* tree.c:124-271.  Requires sorted particles. */
int  orc_build_tree(orc_state *s);     /* returns NNodes, <0 on overflow */
int  orc_tree_nodes(const orc_state *s, uint32_t *bitfield, int32_t *dnext, float *pos3,
                    int32_t *npart, float *size, int32_t *tree_parent);
/* tree.c:25-111 */
int  orc_find_ngb_tree(const orc_state *s, int ipart, float hsml, int32_t *ngblist);
/* wvt_relax.c:296-340 */
int  orc_find_ngb_simple(const orc_state *s, int ipart, float hsml, int32_t *ngblist);
/* tree.c:113-121 */
float orc_guess_hsml(const orc_state *s, int ipart);

/* sph.c:13-75 (includes sort + tree build).  Returns 0, or <0 on failure. */
int  orc_find_sph_quantities(orc_state *s);
/* wvt_relax.c:227-256 for every particle */
void orc_global_density_model(const orc_state *s, float *rho_model_out);
/* wvt_relax.c:25-225.  max_iter<0 => reference behaviour (NUMITER).  Returns number of log lines. */
int  orc_regularise(orc_state *s, orc_iterlog *log, int max_iter);
/* one WVT sweep body only (wvt_relax.c:106-214) at the given step, for unit parity;
 * writes hsml_wvt[n] and delta[3n] (xyz interleaved) if non-NULL. Does NOT call the density pass. */
void orc_wvt_step(orc_state *s, double step, float *hsml_wvt, float *delta3, int move);
/* sph.c:216-300 */
void orc_bfld_from_rotA(orc_state *s);

/* sph.c:426-440, wvt_relax.c:275-281 (scalar kernels, for unit checks) */
float orc_wc6(float r, float h);
float orc_dwc6(float r, float h);
double orc_wvt_wc6(float r, float h);

/* behind the path: positions.c:264-445 (gas halo reassignment + index heapsort by halo id) */
int orc_reassign_to_halos(int n, const float *pos, double boxsize, int nhalos, const orc_halo *halos,
                          const double *r_sample, int32_t *halo_id, int64_t *perm, int64_t *npart);

/* Attribution experiment (tools/attribute_tail.py; NOT reference behaviour, never set by the parity tests): run the
 * oracle with one of the GPU library's documented arithmetic deviations injected.  0 = the faithful restatement. */
#define ORC_DEV_SWEEP_ROUND_ONCE 1   /* WVT displacement: f64 sum at unit step, rounded once (round 2's GPU sweep) */
#define ORC_DEV_TREE_SUMS        2   /* Find_hsml's three f64 sums as 64 partial sums + a reduction tree */
#define ORC_DEV_POW_ULP          4   /* every pow() result moved by -1/0/+1 f64 ulp */
#define ORC_DEV_KERNEL_ULP       8   /* W and W' moved by -1/0/+1 f64 ulp before their f32 rounding */
#define ORC_DEV_EXACT_BALL      16   /* every ball query answered as Find_ngb_simple (wvt_relax.c:296-340) would: the reference's
                                      * tree search misses neighbours stored under a mis-placed node (tc_oracle.c, find_ngb_grid) */
void orc_set_deviation(int mask);
/* number of ball queries that filled the NGBMAX list (tree.c:91-92) since the last reset */
void orc_truncations(orc_state *s, long *density, long *sweep, int reset);

/* stats of the last orc_find_sph_quantities call */
void orc_last_stats(const orc_state *s, double *queries_per_part, double *solver_iters_per_part,
                    double *pair_evals_per_part);

/* magnetic_field.c:33-131 */
void orc_set_vector_potential(int n, const float *pos, double boxsize, int nhalos, const orc_halo *halos, double eta,
                              float *apot);
void orc_normalise_magnetic_field(int n, const float *pos, float *bfld, double boxsize, int nhalos, const orc_halo *halos,
                                  const double *r_sample_gas, const double *r_sample_dm, int sub_first, double bfld_norm,
                                  double *norm_out, int *cnt_out);

#endif
