/*
 * tc_shim.c -- the reference's entry points for the hot path, implemented on libtcgpu.
 *
 * Link libtcshim.so into Toycluster in place of sph.o wvt_relax.o tree.o peano.o sort.o
 * (reference Makefile:71-73).  Each function gathers the gas part of the
 * reference's globals P / SphP / Param / Halo into the C ABI's flat arrays, runs the GPU
 * path, and scatters the results back IN PEANO ORDER, exactly as the reference leaves them
 * (src/peano.c:85-126 permutes P and SphP in place).  Errors follow the reference's
 * convention: message on stderr and exit(EXIT_FAILURE) (src/aux.c:57-83).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/tcgpu.h"
#include "tc_ref_abi.h"
#include "../csrc/tc_math.h"

static tcgpu_ctx *g_ctx;
static int64_t g_uploaded_n;

static void fail(const char *func, const char *msg)
{
    fprintf(stderr, "\nERROR : In libtcshim, function %s() :\n\n\t%s\n\n", func, msg);
    fflush(stderr);
    exit(EXIT_FAILURE);
}

#define CK(call) do { if ((call)) fail(__func__, tcgpu_last_error(g_ctx)); } while (0)

static void ensure_ctx(void)
{
    if (g_ctx) return;
    const char *dev = getenv("TCGPU_DEVICE");
    if (tcgpu_create(&g_ctx, dev ? atoi(dev) : 0)) fail(__func__, "no usable gfx950 device (tcgpu_create failed)");
}

/* Param / Halo -> model scalars */
static void push_model(void)
{
    tcgpu_params par;
    memset(&par, 0, sizeof(par));
    par.boxsize = Param.Boxsize;
    par.mpart_gas = Param.Mpart[0];
    par.mtotal = Param.Mtotal;
    par.bfld_eta = Param.Bfld_Eta;
    par.nhalos = Param.Nhalos;
    tcgpu_halo *h = calloc(Param.Nhalos > 0 ? Param.Nhalos : 1, sizeof(*h));
    for (int i = 0; i < Param.Nhalos; i++) {
        h[i].mass_gas = Halo[i].Mass[0];
        for (int c = 0; c < 3; c++) h[i].d_com[c] = Halo[i].D_CoM[c];
        h[i].rho0 = Halo[i].Rho0; h[i].beta = Halo[i].Beta; h[i].rcore = Halo[i].Rcore; h[i].rcut = Halo[i].Rcut;
        h[i].have_cuspy = Halo[i].Have_Cuspy;
    }
    int rc = tcgpu_set_model(g_ctx, &par, h);
    free(h);
    if (rc) fail(__func__, tcgpu_last_error(g_ctx));
}

/* P/SphP (gas) -> device */
static void push_particles(void)
{
    const size_t n = (size_t)Param.Npart[0];
    float *pos = malloc(3 * n * sizeof(float)), *hsml = malloc(n * sizeof(float));
    int32_t *id = malloc(n * sizeof(int32_t));
    if (!pos || !hsml || !id) fail(__func__, "out of memory");
    for (size_t i = 0; i < n; i++) {
        pos[3 * i] = P[i].Pos[0]; pos[3 * i + 1] = P[i].Pos[1]; pos[3 * i + 2] = P[i].Pos[2];
        id[i] = P[i].ID;
        hsml[i] = SphP[i].Hsml;
    }
    int rc = tcgpu_upload_particles(g_ctx, (int64_t)n, pos, id, hsml);
    free(pos); free(hsml); free(id);
    if (rc) fail(__func__, tcgpu_last_error(g_ctx));
    g_uploaded_n = (int64_t)n;
}

static int cmp_idpair(const void *a, const void *b)
{
    const int64_t *x = a, *y = b;
    return (x[0] > y[0]) - (x[0] < y[0]);
}

/* device -> P/SphP: whole structs are moved to the new (Peano) order, joined on the particle ID */
static void pull_particles(int with_keys, int with_rho_model)
{
    const size_t n = (size_t)Param.Npart[0];
    float *pos = malloc(3 * n * sizeof(float)), *hsml = malloc(n * sizeof(float)), *rho = malloc(n * sizeof(float));
    float *vhf = malloc(n * sizeof(float)), *rhom = malloc(n * sizeof(float));
    int32_t *id = malloc(n * sizeof(int32_t));
    int64_t *old = malloc(2 * n * sizeof(int64_t));
    struct ParticleData *Pn = malloc(n * sizeof(*Pn));
    struct GasParticleData *Sn = malloc(n * sizeof(*Sn));
    uint64_t *khi = NULL, *klo = NULL;
    if (!pos || !hsml || !rho || !vhf || !rhom || !id || !old || !Pn || !Sn) fail(__func__, "out of memory");
    CK(tcgpu_download_particles(g_ctx, pos, id, hsml, rho, vhf, rhom));
    if (with_keys) {
        khi = malloc(n * sizeof(uint64_t)); klo = malloc(n * sizeof(uint64_t));
        if (!khi || !klo || tcgpu_download_keys(g_ctx, khi, klo)) { free(khi); free(klo); khi = klo = NULL; }
    }
    for (size_t i = 0; i < n; i++) { old[2 * i] = P[i].ID; old[2 * i + 1] = (int64_t)i; }
    qsort(old, n, 2 * sizeof(int64_t), cmp_idpair);
    for (size_t i = 0; i < n; i++) {
        int64_t key[2] = {id[i], 0};
        int64_t *hit = bsearch(key, old, n, 2 * sizeof(int64_t), cmp_idpair);
        if (!hit) fail(__func__, "particle id lost between upload and download");
        size_t src = (size_t)hit[1];
        Pn[i] = P[src];
        Sn[i] = SphP[src];
        Pn[i].Pos[0] = pos[3 * i]; Pn[i].Pos[1] = pos[3 * i + 1]; Pn[i].Pos[2] = pos[3 * i + 2];
        if (khi) Pn[i].Key = ((peanoKey)khi[i] << 64) | klo[i];
        Pn[i].Tree_Parent = 0;                    /* no tree is built; only the replaced files read it */
        Sn[i].Hsml = hsml[i]; Sn[i].Rho = rho[i]; Sn[i].VarHsmlFac = vhf[i];
        if (with_rho_model) Sn[i].Rho_Model = rhom[i];   /* only the WVT loop writes it (wvt_relax.c:113) */
    }
    memcpy(P, Pn, n * sizeof(*Pn));
    memcpy(SphP, Sn, n * sizeof(*Sn));
    free(pos); free(hsml); free(rho); free(vhf); free(rhom); free(id); free(old); free(Pn); free(Sn); free(khi); free(klo);
}

/* ---- src/peano.h:5 ---- */
void Sort_Particles_By_Peano_Key(void)
{
    ensure_ctx(); push_model(); push_particles();
    CK(tcgpu_sort_particles_by_peano_key(g_ctx));
    pull_particles(1, 0);
}

/* ---- src/proto.h:17 ---- */
void Find_sph_quantities(void)
{
    ensure_ctx(); push_model(); push_particles();
    CK(tcgpu_find_sph_quantities(g_ctx));
    pull_particles(1, 0);
}

/* ---- src/proto.h:25 ---- */
void Regularise_sph_particles(void)
{
    ensure_ctx(); push_model(); push_particles();
    printf("Starting iterative SPH regularisation \n"
           "   max %d iterations, tree update every %d iterations\n"
           "   stop at  errmax < %g%%   \n\n", TCGPU_NUMITER, 1, 0.01 * 100);
    fflush(stdout);
    tcgpu_iterlog log[TCGPU_MAXLOG];
    int32_t nlog = 0;
    CK(tcgpu_regularise_sph_particles(g_ctx, -1, log, &nlog));
    for (int i = 0; i < nlog && i < TCGPU_MAXLOG; i++)
        printf("   #%02d: Err max=%3g mean=%03g diff=%03g step=%g\n", log[i].it, log[i].err_max, log[i].err_mean,
               log[i].err_diff, log[i].step);
    printf("\ndone\n\n");
    fflush(stdout);
    pull_particles(0, 1);
}

/* ---- src/proto.h:23, src/sph.h:2 : uses the neighbour index of the preceding Find_sph_quantities() ---- */
void Bfld_from_rotA_SPH(void)
{
    const size_t n = (size_t)Param.Npart[0];
    if (!g_ctx || g_uploaded_n != (int64_t)n) Find_sph_quantities();
    printf("Constructing B from rot(A)"); fflush(stdout);
    float *apot = malloc(3 * n * sizeof(float)), *bfld = malloc(3 * n * sizeof(float));
    if (!apot || !bfld) fail(__func__, "out of memory");
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) apot[3 * i + c] = SphP[i].Apot[c];
    CK(tcgpu_bfld_from_rotA_sph(g_ctx, apot, bfld));
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) SphP[i].Bfld[c] = bfld[3 * i + c];
    free(apot); free(bfld);
    printf(" done \n\n"); fflush(stdout);
}

/* ---- src/proto.h:46 : scalar, one particle; same arithmetic as src/wvt_relax.c:227-256 ---- */
float Global_density_model(const int ipart)
{
    const double boxhalf = Param.Boxsize * 0.5;
    const double x = P[ipart].Pos[0], y = P[ipart].Pos[1], z = P[ipart].Pos[2];
    double rho = 0;
    for (int i = 0; i < Param.Nhalos; i++) {
        if (Halo[i].Mass[0] == 0) continue;
        double dx = x - Halo[i].D_CoM[0] - boxhalf, dy = y - Halo[i].D_CoM[1] - boxhalf, dz = z - Halo[i].D_CoM[2] - boxhalf;
        double r = sqrt(dx * dx + dy * dy + dz * dz);
        double a = r / Halo[i].Rcore, b = r / Halo[i].Rcut;
        double rho_i = Halo[i].Rho0 * pow(1 + a * a, -3.0 / 2.0 * Halo[i].Beta) / (1 + (b * b * b) * b);
        rho = fmax(rho_i, rho);
    }
    return rho;
}

/* ---- src/sort.h:7-8, src/sort.c:185-195.  The live part of the reference's Qsort_Index is one call of GSL's
 * gsl_heapsort_index (everything behind the early return is dead code); its only caller outside the replaced
 * files is sort_particles() (src/positions.c:409: int halo ids, comparator compare_int, massive ties).  The
 * published index-heapsort scheme, restated for an arbitrary comparator: p[] starts as the identity, a max-heap
 * is built on p[0..last] with the children of slot k taken at 2k and 2k+1, from k = last/2 down to 0; then the
 * root is swapped with the last entry and sifted down.  Unstable; the order of ties is a property of this
 * procedure alone, which is why the gas block's file order needs exactly it (toycluster_amd/host/tc_reassign.c
 * holds the same scheme specialised for int keys; tests/test_reassign.py compares the two). ---- */
static void qi_sift_down(size_t *p, const char *data, size_t size, size_t last, size_t k,
                         int (*cmp)(const void *, const void *))
{
    const size_t pk = p[k];
    while (k <= last / 2) {
        size_t j = 2 * k;
        if (j < last && cmp(data + size * p[j], data + size * p[j + 1]) < 0) j++;
        if (!(cmp(data + size * pk, data + size * p[j]) < 0)) break;
        p[k] = p[j];
        k = j;
    }
    p[k] = pk;
}

void Qsort_Index(const int nThreads, size_t *perm, void *const data, const int nData, const size_t datasize,
                 int (*cmp)(const void *, const void *))
{
    (void)nThreads;                                   /* the reference sorts under `omp single` */
    if (nData <= 0) return;
    const size_t n = (size_t)nData;
    for (size_t i = 0; i < n; i++) perm[i] = i;
    size_t last = n - 1, k = last / 2 + 1;
    do {
        k--;
        qi_sift_down(perm, data, datasize, last, k, cmp);
    } while (k > 0);
    while (last > 0) {
        const size_t t = perm[0]; perm[0] = perm[last]; perm[last] = t;
        last--;
        qi_sift_down(perm, data, datasize, last, 0, cmp);
    }
}

/* ---- src/peano.h:6, src/peano.c:128-203: host arithmetic shared with the device kernels (csrc/tc_math.h);
 * no caller outside the replaced files, exported for completeness of peano.h ---- */
peanoKey Peano_Key(const double x, const double y, const double z)
{
    if (!(x >= 0 && x <= 1 && y >= 0 && y <= 1 && z >= 0 && z <= 1)) fail(__func__, "coordinate outside [0,1]");
    const double m = 9223372036854775808.0;           /* 2^63, src/peano.c:134-136 */
    uint64_t X[3] = {(uint64_t)(y * m), (uint64_t)(z * m), (uint64_t)(x * m)}, hi, lo;
    tc_hilbert_transpose(X);
    tc_key_from_transpose(X, &hi, &lo);
    return ((peanoKey)hi << 64) | lo;
}
