/* Test harness standing in for the rest of Toycluster: defines the reference's globals
 * (as src/aux.c:3-6 does), fills the gas particles from a state file, and calls the reference's
 * entry points, which resolve to libtcshim.so.  Writes what a caller would observe afterwards. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../toycluster_amd/host/tc_host.h"
#include "../toycluster_amd/host/tc_ref_abi.h"

struct Parameters Param;
struct HaloProperties Halo[REF_MAXHALOS];
struct ParticleData *P;
struct GasParticleData *SphP;

void Regularise_sph_particles(void);
void Find_sph_quantities(void);
void Bfld_from_rotA_SPH(void);
float Global_density_model(const int ipart);

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    tc_state st;
    char err[512];
    if (tc_read_state(argv[1], &st, err, sizeof(err))) { fprintf(stderr, "%s\n", err); return 3; }
    const size_t n = (size_t)st.ngas;
    memset(&Param, 0, sizeof(Param));
    Param.Npart[0] = st.ngas; Param.Ntotal = st.ngas;
    Param.Boxsize = st.par.boxsize; Param.Mpart[0] = st.par.mpart_gas; Param.Mtotal = st.par.mtotal;
    Param.Nhalos = st.par.nhalos; Param.Bfld_Eta = 0.5;
    for (int i = 0; i < st.par.nhalos; i++) {
        Halo[i].Mass[0] = st.halos[i].mass_gas;
        for (int c = 0; c < 3; c++) Halo[i].D_CoM[c] = st.halos[i].d_com[c];
        Halo[i].Rho0 = st.halos[i].rho0; Halo[i].Beta = st.halos[i].beta;
        Halo[i].Rcore = st.halos[i].rcore; Halo[i].Rcut = st.halos[i].rcut;
    }
    P = calloc(n, sizeof(*P));
    SphP = calloc(n, sizeof(*SphP));
    for (size_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) P[i].Pos[c] = st.pos[3 * i + c];
        P[i].ID = st.id[i];
        P[i].Vel[0] = (float)st.id[i] * 0.5f;        /* payload that must travel with the particle */
        SphP[i].U = (float)st.id[i] + 0.25f;
    }
    int iters = atoi(argv[3]);
    (void)iters;
    Regularise_sph_particles();
    Find_sph_quantities();
    for (size_t i = 0; i < n; i++) SphP[i].Apot[0] = SphP[i].Apot[1] = SphP[i].Apot[2] = SphP[i].Rho_Model;
    Bfld_from_rotA_SPH();

    FILE *fp = fopen(argv[2], "wb");
    for (size_t i = 0; i < n; i++) {
        float rec[12] = {P[i].Pos[0], P[i].Pos[1], P[i].Pos[2], (float)P[i].ID, P[i].Vel[0], SphP[i].U,
                         SphP[i].Hsml, SphP[i].Rho, SphP[i].VarHsmlFac, SphP[i].Rho_Model, SphP[i].Bfld[0],
                         Global_density_model((int)i)};
        fwrite(rec, sizeof(float), 12, fp);
    }
    fclose(fp);
    return 0;
}
