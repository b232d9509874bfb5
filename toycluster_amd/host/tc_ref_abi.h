/*
 * tc_ref_abi.h -- binary layout of the reference's global state that the hot path reads and
 * writes, restated so that libtcshim can be linked into an unmodified Toycluster build
 * (default Makefile options: no ADD_THIRD_SUBHALO, no DOUBLE_BETA_COOL_CORES).
 *
 * Source of truth: reference src/globals.h:94-121 (Param), :132-159 (Halo), :161-168 (P),
 * :170-180 (SphP), definitions in src/aux.c:3-13.  Field order and types must stay in step
 * with that header; the static asserts below pin the sizes the survey recorded (64 B / 60 B).
 */
#ifndef TC_REF_ABI_H
#define TC_REF_ABI_H

#include <stdint.h>

#define REF_CHARBUFSIZE 512
#define REF_MAXHALOS 4096

typedef __uint128_t peanoKey;                 /* src/peano.h:3 */

struct Parameters {                            /* src/globals.h:94-121 */
    char Output_File[REF_CHARBUFSIZE];
    long long Ntotal;
    long long Npart[6];
    double Mtotal;
    double Mtot200;
    double Mass_Ratio;
    double Impact_Param;
    double Mpart[6];
    double Redshift;
    int Cuspy;
    double Bfld_Norm;
    double Bfld_Eta;
    double Boxsize;
    double VelMerger[2];
    int Nhalos;
    double GravSofteningLength;
    double Zero_Energy_Orbit_Fraction;
};

struct ParticleData {                          /* src/globals.h:161-168, 64 bytes */
    float Pos[3];
    float Vel[3];
    int32_t ID;
    int Type;
    peanoKey Key;
    int Tree_Parent;
};

struct GasParticleData {                       /* src/globals.h:170-180, 60 bytes */
    float U;
    float Rho;
    float Hsml;
    float VarHsmlFac;
    float Bfld[3];
    float Apot[3];
    float ID;
    float Rho_Model;
    float Rs[3];
};

struct HaloProperties {                        /* src/globals.h:132-159 */
    long long Ntotal;
    long long Npart[6];
    int Have_Cuspy;
    int Is_Stripped;
    double Mtotal;
    double Mass[6];
    double Mtotal200;
    double Mass200[6];
    double MassCorrFac;
    double C_nfw;
    double Rs;
    double R200;
    double R500;
    double A_hernq;
    double Rho0;
    double Beta;
    double Rcore;
    double Bf_eff;
    double D_CoM[3];
    double BulkVel[3];
    double R_Sample[2];
    double Rcut;
    double TempOffset;
    struct ParticleData *DM;
    struct ParticleData *Gas;
    struct GasParticleData *SphP;
};

_Static_assert(sizeof(struct ParticleData) == 64, "ParticleData must be 64 bytes (SURVEY.md 8a1)");
_Static_assert(sizeof(struct GasParticleData) == 60, "GasParticleData must be 60 bytes (SURVEY.md 8a2)");

/* defined by the reference (src/aux.c:3-6) */
extern struct Parameters Param;
extern struct HaloProperties Halo[REF_MAXHALOS];
extern struct ParticleData *P;
extern struct GasParticleData *SphP;

#endif
