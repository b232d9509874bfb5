/*
 * tc_host.h -- C host side of the MI355X Toycluster path: the reference's run-time interface
 * (parameter file in, Gadget-2 "format 2" snapshot out) around libtcgpu.
 *
 *   parameter file : reference src/io.c:298-507 (Read_param_file) -- `tag value` lines, `%` comments,
 *                    unknown tags ignored, every known tag mandatory
 *   snapshot       : reference src/io.c:13-287 + src/io.h:1-41 -- HEAD, POS, VEL, ID, U, RHO, HSML,
 *                    BFLD, RHOM blocks, each preceded by an 8-byte label record, all wrapped in
 *                    4-byte Fortran record markers
 *   state file     : this repo's exchange format for what the hot path consumes (model scalars +
 *                    gas positions + ids; SURVEY.md Appendix A "hot-path state file")
 */
#ifndef TC_HOST_H
#define TC_HOST_H

#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include "../../include/tcgpu.h"

#define TC_CHARBUF 512        /* reference CHARBUFSIZE, src/globals.h:37 */

/* ---- parameter file (the tags of the default build: Makefile:3-25 without GIVEPARAMS etc.) ---- */
typedef struct {
    char   output_file[TC_CHARBUF];   /* Output_file */
    long long ntotal;                 /* Ntotal   (read with atoi, src/io.c:476-478) */
    double mtot200;                   /* Mtotal -> Param.Mtot200 (src/io.c:319-321) */
    double redshift;                  /* Redshift */
    double mass_ratio;                /* Mass_Ratio */
    double impact_param;              /* ImpactParam */
    double zero_e_orbit_frac;         /* ZeroEOrbitFrac */
    int    cuspy;                     /* Cuspy */
    double bfld_norm;                 /* Bfld_Norm */
    double bfld_eta;                  /* Bfld_Eta */
    double baryon_fraction;           /* bf */
    double unit_length, unit_mass, unit_vel;   /* UnitLength_in_cm, UnitMass_in_g, UnitVelocity_in_cm_per_s */
    /* the reference's -DDOUBLE_BETA_COOL_CORES build (Makefile:11) as a run-time switch: TC_DOUBLE_BETA=1 in the
     * environment makes the two tags mandatory (src/io.c:435-443) and switches the cool-core component on */
    int    double_beta, pad_;
    double rho0_fac, rc_fac;          /* Rho0_Fac, Rc_Fac */
} tc_parfile;

/* 0 on success; 1 = file not found, 2 = a tag is missing (both fatal in the reference: exit(1)).
 * err receives the reference's message. */
int tc_read_param_file(const char *filename, tc_parfile *out, char *err, size_t errlen);

/* ---- Gadget-2 format-2 snapshot ---- */
typedef struct {                      /* src/io.h:1-19, 256 bytes */
    int32_t  npart[6];
    double   mass[6];
    double   time;
    double   redshift;
    int32_t  flag_sfr;
    int32_t  flag_feedback;
    uint32_t npartTotal[6];
    int32_t  flag_cooling;
    int32_t  num_files;
    double   BoxSize;
    double   Omega0;
    double   OmegaLambda;
    double   HubbleParam;
    int32_t  flag_stellarage;
    int32_t  flag_metals;
    uint32_t npartTotalHighWord[6];
    char     fill[64];
} tc_gadget_header;

typedef struct {
    long long npart[6];               /* Param.Npart */
    double mpart[6];                  /* Param.Mpart */
    double boxsize;
    double hubble_param;              /* Cosmo.h_100 (0.7, src/cosmo.c:11) */
    /* all particles, gas first: */
    const float *pos;                 /* 3*ntot */
    const float *vel;                 /* 3*ntot */
    const int32_t *id;                /* ntot */
    /* gas only (npart[0]): */
    const float *u, *rho, *hsml, *bfld /* 3*ngas */, *rho_model;
} tc_snapshot;

/* 0 on success, 5/6 on I/O error (the reference's exit codes, src/io.c:256,280) */
int tc_write_snapshot(const char *filename, const tc_snapshot *s);

/* ---- stages in front of the hot path: units, cosmology, halo set-up, sampling ---- */
#define TC_SETUP_MAXHALOS 72           /* 2 clusters + at most 70 subhalos (src/substructure.c:129) */

typedef struct {                      /* the fields of the reference's HaloProperties that Setup() fills */
    double mtotal200, mass200[2];     /* [0] gas, [1] dark matter */
    double c_nfw, r200, rs, a_hernq;
    double rho0, beta, rcore, rcut;
    double r_sample[2], mass[2], mtotal, mass_corr_fac;
    double d_com[3];
    long long npart[2];
    int have_cuspy, is_stripped;
    double rho0_cc, rc_cc;            /* cool-core component (src/setup.c:604-612), 0 without -DDOUBLE_BETA_COOL_CORES */
} tc_halo_setup;

/* src/setup.c:598-615: the beta model with its r^4 cut-off, plus the -DDOUBLE_BETA_COOL_CORES component when
 * rho0_cc != 0 (a halo with Is_Cuspy in that build) */
static inline double tc_host_gas_profile(double r, double rho0, double beta, double rcore, double rcut, double rho0_cc,
                                         double rc_cc)
{
    const double a = r / rcore, b = r / rcut;
    double rho = rho0 * pow(1 + a * a, -3.0 / 2.0 * beta) / (1 + (b * b * b) * b);
    if (rho0_cc != 0) {
        const double c = r / rc_cc;
        rho += rho0_cc / (1 + c * c) / (1 + (b * b * b) * b);
    }
    return rho;
}

typedef struct {
    tc_parfile par;
    double unit_length, unit_mass, unit_vel, unit_time;
    double h_100, omega_m, omega_l, h0_cgs, rho_crit, delta;   /* rho_crit at the cluster redshift, Delta_vir */
    int nhalos, pad_;
    tc_halo_setup halo[TC_SETUP_MAXHALOS];
    double boxsize, mtotal, mpart[2];
    long long npart[2];
    /* -DSUBSTRUCTURE state (struct SubhaloData, src/globals.h): filled by tc_setup_substructure */
    int sub_first, sub_nhalos, subhost, pad2_;
    double sub_mtotal, sub_mass_fraction, grav_softening;
    long long sub_npart[2];
} tc_setup;

/* Set_units + Set_cosmology + Setup (src/unit.c, src/cosmo.c, src/setup.c:21-344), default build options */
int  tc_setup_system(const tc_parfile *par, tc_setup *out);
void tc_setup_to_model(const tc_setup *s, tcgpu_params *par, tcgpu_halo *halos /* [s->nhalos] */);
/* Setup_Substructure (src/substructure.c, build option -DSUBSTRUCTURE -DSUBHOST=<subhost>): appends up to 70
 * subhalos to s->halo[], takes their particles from the host cluster.  `seed` is the calling thread's
 * erand48 state (the reference draws from thread 0's stream, src/main.c:20-21) and is advanced. */
int  tc_setup_substructure(tc_setup *s, int subhost, unsigned short seed[3]);
/* Make_positions (gas), Make_IDs, Shift_Origin: pos f32[3*npart[0]] in [0,boxsize], id i32[npart[0]] */
int  tc_sample_gas(const tc_setup *s, int nthreads, float *pos, int32_t *id);
void tc_thread_seed(int tid, unsigned short seed[3]);
int  tc_sample_gas_seeded(const tc_setup *s, int nthreads, const unsigned short seed0[3], float *pos, int32_t *id);

/* ---- state file ---- */
typedef struct {
    tcgpu_params par;
    tcgpu_halo *halos;                /* par.nhalos entries */
    int64_t ngas;
    float *pos;                       /* 3*ngas, in [0, boxsize] */
    int32_t *id;                      /* ngas */
    double *r_sample;                 /* Halo[i].R_Sample[0] per halo, or NULL (optional trailer of the file) */
    double *r_sample_dm;              /* Halo[i].R_Sample[1]; in memory only (native set-up), NULL from a state file */
    int sub_first;                    /* Sub.First (src/aux.c:10: 2 unless Setup_Substructure ran) */
} tc_state;

int  tc_read_state(const char *filename, tc_state *st, char *err, size_t errlen);
int  tc_write_state(const char *filename, const tc_state *st);
void tc_free_state(tc_state *st);

/* ---- the step behind the hot path: Reassign_particles_to_halos, src/positions.c:264-445 ---- */
int  tc_halo_containing_gas(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample,
                            float x, float y, float z);
void tc_heapsort_index_i32(size_t *p, const int32_t *key, size_t n);
int  tc_reassign_particles_to_halos(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample,
                                    size_t n, const float *pos, int32_t *halo_id, size_t *perm, long long *npart);
int  tc_permute_rows(void *data, size_t n, size_t width, const size_t *perm);

/* ---- the wrapper around the curl: src/magnetic_field.c:33-131 ---- */
void tc_set_magnetic_vector_potential(const tcgpu_params *par, const tcgpu_halo *halos, double bfld_eta, size_t n,
                                      const float *pos, float *apot);
int  tc_halo_containing_dm(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample_dm, int sub_first,
                           float x, float y, float z);
int  tc_normalise_magnetic_field(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample_gas,
                                 const double *r_sample_dm, int sub_first, double bfld_norm, size_t n, const float *pos,
                                 float *bfld, double *norm_out, long long *nlimited);

#endif
