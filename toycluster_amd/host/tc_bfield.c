/* tc_bfield.c -- the wrapper around the SPH curl: vector potential before it, normalisation and
 * limiter behind it.
 *
 *   set_magnetic_vector_potential   src/magnetic_field.c:33-69
 *   normalise_magnetic_field        src/magnetic_field.c:71-131
 *   Halo_containing (DM branch)     src/positions.c:333-362   <- what the limiter really calls, see below
 *
 * The limiter clamps |B| to BMAX = 18e-6 G, and to 2e-6 G for particles "in subhaloes":
 *     int i = Halo_containing(ipart, x, y, z);   if (i > 1) bmax = 2e-6;            magnetic_field.c:109-114
 * Halo_containing's first parameter is the particle TYPE (positions.c:333); the call passes the particle INDEX.
 * So particle 0 of the (Peano-ordered) gas block is classified by the gas branch (densest halo) and every other
 * particle by the DARK-MATTER branch: inside Halo[1]'s DM sampling radius with x > 0 -> 1, then the first
 * subhalo j >= Sub.First whose DM sampling radius contains it.  That is what the reference's output holds, so it
 * is what is reproduced here, quirk included (`sic`).  In the default build (Sub.First = 2 = Nhalos at most,
 * aux.c:10) no index above 1 can come out and only the 18e-6 limit exists.
 * Arithmetic as in the reference: B2 from f32 products summed in f32 (p2() on float operands), the scale
 * factors applied as (float)((double)b * factor).  The maximum is taken without the reference's data race
 * (magnetic_field.c:75-85 updates max_B2 from all threads without a reduction). */
#include <math.h>
#include <stdlib.h>
#include "tc_host.h"

#define BMAX 18e-6          /* src/magnetic_field.c:4 */
#define BMAX_SUBHALO 2e-6   /* src/magnetic_field.c:114 */

static double gas_profile(double r, const tcgpu_halo *h)       /* src/setup.c:598-615 */
{
    return tc_host_gas_profile(r, h->rho0, h->beta, h->rcore, h->rcut, h->rho0_cc, h->rc_cc);
}

/* src/magnetic_field.c:33-69: A = max over gas halos of (rho_i(r) / rho0_i)^eta, same value on all three components */
void tc_set_magnetic_vector_potential(const tcgpu_params *par, const tcgpu_halo *halos, double bfld_eta, size_t n,
                                      const float *pos, float *apot)
{
    const float boxhalf = 0.5 * par->boxsize;
    for (size_t i = 0; i < n; i++) {
        double a_max = 0;
        for (int k = 0; k < par->nhalos; k++) {
            const tcgpu_halo *h = &halos[k];
            if (h->mass_gas == 0) continue;
            float dx = pos[3 * i] - h->d_com[0] - boxhalf, dy = pos[3 * i + 1] - h->d_com[1] - boxhalf,
                  dz = pos[3 * i + 2] - h->d_com[2] - boxhalf;
            double r2 = dx * dx + dy * dy + dz * dz;
            double a = pow(gas_profile(sqrt(r2), h) / h->rho0, bfld_eta);
            if (a > a_max) a_max = a;
        }
        apot[3 * i] = apot[3 * i + 1] = apot[3 * i + 2] = (float)a_max;
    }
}

/* DM branch of Halo_containing, src/positions.c:343-362; x, y, z relative to the box centre */
int tc_halo_containing_dm(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample_dm, int sub_first,
                          float x, float y, float z)
{
    if (x > par->boxsize || y > par->boxsize || z > par->boxsize) return -1;
    int i = 0;
    if (par->nhalos > 1) {          /* the reference reads Halo[1] unconditionally; an unused slot is all zero there */
        const double dx = x - halos[1].d_com[0], dy = y - halos[1].d_com[1], dz = z - halos[1].d_com[2];
        const float r = sqrt(dx * dx + dy * dy + dz * dz);
        if (r < r_sample_dm[1] && x > 0) i = 1;
    }
    for (int j = sub_first; j < par->nhalos; j++) {
        const double dx = x - halos[j].d_com[0], dy = y - halos[j].d_com[1], dz = z - halos[j].d_com[2];
        const float r = sqrt(dx * dx + dy * dy + dz * dz);
        if (r < r_sample_dm[j]) { i = j; break; }
    }
    return i;
}

/* src/magnetic_field.c:71-131.  pos / bfld in the order the hot path left them (Peano order): the index of a
 * particle decides which branch classifies it (sic, see the head of this file).  r_sample_dm may be NULL (a state
 * file carries no dark-matter radii): then no particle counts as subhalo member, as in the default build.
 * Returns the norm and the number of limited particles through the pointers. */
int tc_normalise_magnetic_field(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample_gas,
                                const double *r_sample_dm, int sub_first, double bfld_norm, size_t n, const float *pos,
                                float *bfld, double *norm_out, long long *nlimited)
{
    const float boxhalf = 0.5 * par->boxsize;
    double max_b2 = 0;
    for (size_t i = 0; i < n; i++) {
        const float *b = bfld + 3 * i;
        double b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];            /* f32 products, f32 sums (p2 on floats) */
        max_b2 = fmax(max_b2, b2);
    }
    const double norm = bfld_norm / sqrt(max_b2) / sqrt(3);
    long long cnt = 0;
    for (size_t i = 0; i < n; i++) {
        float *b = bfld + 3 * i;
        b[0] *= norm; b[1] *= norm; b[2] *= norm;
        double B2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        const float x = pos[3 * i] - boxhalf, y = pos[3 * i + 1] - boxhalf, z = pos[3 * i + 2] - boxhalf;
        int h = 0;
        if (i == 0) h = r_sample_gas ? tc_halo_containing_gas(par, halos, r_sample_gas, x, y, z) : 0;   /* type 0 (sic) */
        else if (r_sample_dm) h = tc_halo_containing_dm(par, halos, r_sample_dm, sub_first, x, y, z);     /* type > 0 (sic) */
        double bmax = BMAX;
        if (h > 1) bmax = BMAX_SUBHALO;
        if (B2 > bmax * bmax) {
            const double B = sqrt(B2);
            b[0] *= bmax / B; b[1] *= bmax / B; b[2] *= bmax / B;
            cnt++;
        }
    }
    if (norm_out) *norm_out = norm;
    if (nlimited) *nlimited = cnt;
    return 0;
}
