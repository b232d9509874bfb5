/*
 * toycluster_hip -- C host program for the MI355X SPH/WVT path.
 *
 *   usage: ./toycluster_hip <parameterfile> [statefile|-] [device]
 *
 * Without a state file the set-up stages run natively (tc_setup.c: units, cosmology, halo scalars,
 * gas sampling, ids, origin shift -- the gas half of src/main.c:32-46); with one, their result is
 * read from it (e.g. a state dumped from a real Toycluster run).
 *
 * It is the gas branch of the reference's main() (src/main.c:50-69) with the hot path running on
 * the GPU through libtcgpu's C ABI:
 *     Regularise_sph_particles();   src/main.c:52  -> tcgpu_regularise_sph_particles()
 *     Find_sph_quantities();        src/main.c:54  -> tcgpu_find_sph_quantities()
 *     Make_magnetic_field();        src/main.c:56  -> A from the model (src/magnetic_field.c:33-69),
 *                                                     tcgpu_bfld_from_rotA_sph(), normalisation
 *                                                     (src/magnetic_field.c:71-131, race-free max)
 *     Reassign_particles_to_halos(); src/main.c:58 -> tc_reassign_particles_to_halos() (tc_reassign.c)
 *     Write_output();               src/main.c:69  -> tc_write_snapshot()
 * Temperatures and velocities (out of scope) are written as zeros.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "tc_host.h"

#define BMAX 18e-6   /* src/magnetic_field.c:4 */

static void die(int code, const char *what, const char *msg)
{
    fprintf(stderr, "\nERROR : %s: %s\n\n", what, msg);
    exit(code);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage : ./toycluster_hip $parameterfile [$statefile|-] [device]\n");
        return EXIT_FAILURE;
    }
    char err[1024];
    tc_parfile par;
    if (tc_read_param_file(argv[1], &par, err, sizeof(err))) { fprintf(stderr, "%s\n", err); return 1; }   /* exit(1) */
    printf("\nReading Parameter file : %s \n\n", argv[1]);

    tc_state st;
    if (argc > 2 && strcmp(argv[2], "-")) {
        if (tc_read_state(argv[2], &st, err, sizeof(err))) die(EXIT_FAILURE, "state file", err);
    } else {                                  /* Set_units .. Shift_Origin, natively */
        tc_setup S;
        tc_setup_system(&par, &S);
        if (S.npart[0] <= 0) die(EXIT_FAILURE, "setup", "no gas particles (bf = 0): nothing for the SPH path to do");
        printf("System Setup : \n   Boxsize         = %g kpc\n   Total Mass      = %g 10^10 Msol\n"
               "   Sph Part Mass   = %g 10^10 Msol\n   Npart Total     = %8lld, %8lld\n\n",
               S.boxsize, S.mtotal, S.mpart[0], S.npart[0], S.npart[1]);
        for (int i = 0; i < S.nhalos; i++)
            printf("Halo Setup : <%d>\n   R200 = %g kpc  c_nfw = %g  rho0_gas = %g [gadget]  beta = %g  rc = %g kpc  Rcut = %g kpc\n",
                   i, S.halo[i].r200, S.halo[i].c_nfw, S.halo[i].rho0, S.halo[i].beta, S.halo[i].rcore, S.halo[i].rcut);
        /* -DSUBSTRUCTURE -DSUBHOST=n are compile-time options of the reference (Makefile:17-18); here they
         * are run-time: TC_SUBSTRUCTURE=1 [TC_SUBHOST=n] in the environment */
        unsigned short seed0[3];
        tc_thread_seed(0, seed0);
        const char *want_sub = getenv("TC_SUBSTRUCTURE");
        if (want_sub && atoi(want_sub)) {
            const char *sh = getenv("TC_SUBHOST");
            const int subhost = sh ? atoi(sh) : 0;
            printf("\nSubhalos hosted by cluster <%d> \n\n", subhost);              /* src/setup.c:339-341 */
            int src = tc_setup_substructure(&S, subhost, seed0);
            if (src) { snprintf(err, sizeof(err), "Setup_Substructure failed (%d)", src); die(EXIT_FAILURE, "setup", err); }
            printf("\nSubhalo Setup : \n   Total Mass DM   = %g \n   Mass Fraction   = %4.2g\n   Target Fraction = %g \n"
                   "   Total Number    = %d / %d \n   Total Ngas      = %lld \n   Total NDM       = %lld \n",
                   S.sub_mtotal, S.sub_mtotal / S.halo[S.subhost].mtotal200, S.sub_mass_fraction, S.sub_nhalos,
                   S.nhalos, S.sub_npart[0], S.sub_npart[1]);
        }
        memset(&st, 0, sizeof(st));
        st.ngas = S.npart[0];
        st.halos = calloc((size_t)S.nhalos, sizeof(tcgpu_halo));
        st.pos = malloc(3 * (size_t)st.ngas * sizeof(float));
        st.id = malloc((size_t)st.ngas * sizeof(int32_t));
        if (!st.halos || !st.pos || !st.id) die(EXIT_FAILURE, "malloc", "out of memory");
        tc_setup_to_model(&S, &st.par, st.halos);
        st.r_sample = malloc((size_t)S.nhalos * sizeof(double));
        if (!st.r_sample) die(EXIT_FAILURE, "malloc", "out of memory");
        st.r_sample_dm = malloc((size_t)S.nhalos * sizeof(double));
        if (!st.r_sample_dm) die(EXIT_FAILURE, "malloc", "out of memory");
        for (int i = 0; i < S.nhalos; i++) { st.r_sample[i] = S.halo[i].r_sample[0]; st.r_sample_dm[i] = S.halo[i].r_sample[1]; }
        st.sub_first = S.sub_nhalos > 0 ? S.sub_first : 2;                          /* src/aux.c:10, substructure.c:33-36 */
        const char *nt = getenv("OMP_NUM_THREADS");
        printf("Sampling positions "); fflush(stdout);
        tc_sample_gas_seeded(&S, nt ? atoi(nt) : 1, seed0, st.pos, st.id);
        printf(" done\n");
    }
    st.par.bfld_eta = par.bfld_eta;
    const size_t n = (size_t)st.ngas;

    tcgpu_ctx *ctx = NULL;
    int rc = tcgpu_create(&ctx, argc > 3 ? atoi(argv[3]) : 0);
    if (rc) die(EXIT_FAILURE, "tcgpu_create", "no usable gfx950 device");
#define CK(call) do { if ((rc = (call))) die(EXIT_FAILURE, #call, tcgpu_last_error(ctx)); } while (0)
    CK(tcgpu_set_model(ctx, &st.par, st.halos));
    CK(tcgpu_upload_particles(ctx, st.ngas, st.pos, st.id, NULL));

    /* ---- Regularise_sph_particles(): banner and log lines of src/wvt_relax.c:31-34,91-92,222 */
    printf("Starting iterative SPH regularisation \n"
           "   max %d iterations, tree update every %d iterations\n"
           "   stop at  errmax < %g%%   \n\n", TCGPU_NUMITER, 1, 0.01 * 100);
    fflush(stdout);
    tcgpu_iterlog log[TCGPU_MAXLOG];
    int32_t nlog = 0;
    CK(tcgpu_regularise_sph_particles(ctx, -1, log, &nlog));
    for (int i = 0; i < nlog && i < TCGPU_MAXLOG; i++)
        printf("   #%02d: Err max=%3g mean=%03g diff=%03g step=%g\n", log[i].it, log[i].err_max, log[i].err_mean,
               log[i].err_diff, log[i].step);
    printf("\ndone\n\n");
    fflush(stdout);

    /* ---- Find_sph_quantities() */
    CK(tcgpu_find_sph_quantities(ctx));

    float *pos = malloc(3 * n * sizeof(float)), *hsml = malloc(n * sizeof(float)), *rho = malloc(n * sizeof(float));
    float *rhom = malloc(n * sizeof(float)), *apot = malloc(3 * n * sizeof(float)), *bfld = malloc(3 * n * sizeof(float));
    float *zeros = calloc(3 * n, sizeof(float));
    int32_t *id = malloc(n * sizeof(int32_t));
    if (!pos || !hsml || !rho || !rhom || !apot || !bfld || !zeros || !id) die(EXIT_FAILURE, "malloc", "out of memory");
    CK(tcgpu_download_particles(ctx, pos, id, hsml, rho, NULL, rhom));

    /* ---- Make_magnetic_field() */
    printf("Magnetic field: \n   B0              = %g G\n   eta             = %g \n\n", par.bfld_norm, par.bfld_eta);
    tc_set_magnetic_vector_potential(&st.par, st.halos, par.bfld_eta, n, pos, apot);   /* src/magnetic_field.c:33-69 */
    printf("Constructing B from rot(A)"); fflush(stdout);
    CK(tcgpu_bfld_from_rotA_sph(ctx, apot, bfld));
    printf(" done \n\n");
    /* src/magnetic_field.c:71-131 (tc_bfield.c): normalisation, 18e-6 G limit, 2e-6 G for "subhalo" particles as the
     * reference's Halo_containing(ipart, ...) call classifies them */
    double norm = 0;
    long long cnt = 0;
    tc_normalise_magnetic_field(&st.par, st.halos, st.r_sample, st.r_sample_dm, st.sub_first, par.bfld_norm, n, pos, bfld,
                                &norm, &cnt);
    printf("Bfld Norm = %g \n", norm);
    printf("Bfld of %lld particles limited to %g G\n", cnt, BMAX);

    /* ---- Reassign_particles_to_halos(): gas block reordered by halo (src/positions.c:264-331) */
    if (st.r_sample) {
        int32_t *halo_id = malloc(n * sizeof(int32_t));
        size_t *perm = malloc(n * sizeof(size_t));
        long long *np_halo = calloc((size_t)(st.par.nhalos > 0 ? st.par.nhalos : 1), sizeof(long long));
        if (!halo_id || !perm || !np_halo) die(EXIT_FAILURE, "malloc", "out of memory");
        rc = tc_reassign_particles_to_halos(&st.par, st.halos, st.r_sample, n, pos, halo_id, perm, np_halo);
        if (rc) die(EXIT_FAILURE, "Reassign_particles_to_halos", "particle outside the box");
        rc = tc_permute_rows(pos, n, 3 * sizeof(float), perm) || tc_permute_rows(id, n, sizeof(int32_t), perm)
             || tc_permute_rows(hsml, n, sizeof(float), perm) || tc_permute_rows(rho, n, sizeof(float), perm)
             || tc_permute_rows(rhom, n, sizeof(float), perm) || tc_permute_rows(bfld, n, 3 * sizeof(float), perm);
        if (rc) die(EXIT_FAILURE, "malloc", "out of memory");
        printf("Particle Distribution after Relaxation :\n   Main     %8lld   %8lld   %8lld  \n",
               np_halo[0], np_halo[0], 0LL);                              /* gas only: no dark matter here */
        if (st.par.nhalos > 1) printf("   Bullet   %8lld   %8lld   %8lld  \n", np_halo[1], np_halo[1], 0LL);
        free(halo_id); free(perm); free(np_halo);
    } else {
        printf("State file carries no sampling radii: gas block left in Peano order\n");
    }

    /* ---- Write_output() */
    tc_snapshot s;
    memset(&s, 0, sizeof(s));
    s.npart[0] = st.ngas;
    s.mpart[0] = st.par.mpart_gas;
    s.boxsize = st.par.boxsize;
    s.hubble_param = 0.7;                                               /* src/cosmo.c:11 */
    s.pos = pos; s.vel = zeros; s.id = id; s.u = zeros; s.rho = rho; s.hsml = hsml; s.bfld = bfld; s.rho_model = rhom;
    printf("Output : \n   File Name = %s\n", par.output_file);
    rc = tc_write_snapshot(par.output_file, &s);
    if (rc) { fprintf(stderr, "I/O error (fwrite) "); return rc; }
    printf("done\n");

    tcgpu_destroy(ctx);
    tc_free_state(&st);
    free(pos); free(hsml); free(rho); free(rhom); free(apot); free(bfld); free(zeros); free(id);
    return EXIT_SUCCESS;
}
