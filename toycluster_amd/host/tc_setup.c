/*
 * tc_setup.c -- the stages in front of the hot path, natively in C: units, cosmology,
 * halo scalars, gas mass profile, gas position sampling, ids, origin shift.  This is what produces the
 * hot path's input (Param / Halo scalars + gas positions) from the reference's parameter file.
 *
 * Behaviour restated from the reference (default Makefile options: NFWC_DUFFY08, BETA=0.54, COMET,
 * NO_RCUT_IN_T; no SUBSTRUCTURE, no GIVEPARAMS):
 *   Set_units        src/unit.c:3-20        Set_cosmology     src/cosmo.c:8-35
 *   Critical_Density src/cosmo.c:42-45      Overdensity_Parameter  src/cosmo.c:71-90
 *   Setup            src/setup.c:21-344     Concentration_parameter src/setup.c:503-552
 *   Gas_core_radius  src/setup.c:555-592    Setup_Mass_Profile src/setup.c:643-701
 *   sample_Gas_particles src/positions.c:90-133, Halo_containing src/positions.c:333-388
 *   Make_IDs         src/ids.c:8-44         Shift_Origin      src/setup.c:427-500
 *
 * Numerics that the reference takes from GSL (absent here) are this file's own: the mass integral uses
 * an adaptive 15-point Gauss-Legendre rule to 1e-10 instead of gsl_integration_qag(GAUSS41, 1e-6), the
 * tables use a natural cubic spline like gsl_interp_cspline.  Halo.Rho0 / Mpart therefore agree with a
 * GSL build only to the reference integrator's own 1e-6 ("parity unpinned at the GSL boundary",
 * SURVEY.md 8c); the closed-form scalars are pinned by the survey's probe values (tests/test_host_setup.py).
 * Physical constants are GSL's CGSM values (gsl_const_cgsm.h), restated.
 */
#define _XOPEN_SOURCE 600
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "tc_host.h"

#define PI 3.14159265358979323846
#define FOURPITHIRD 4.18879032135009765          /* src/globals.h:63 */
#define GRAV_CGS 6.673e-8                        /* GSL_CONST_CGSM_GRAVITATIONAL_CONSTANT */
#define MSOL2CGS 1.98892e33                      /* src/globals.h:73 */
#define KPC2CGS 3.08568025e21                    /* src/globals.h:74 */
#define BETA_DEFAULT 0.54                        /* Makefile:6 */
#define R200_TO_RMAX_RATIO 3.75                  /* src/globals.h:55 */
#define NTABLE 1024                              /* src/setup.c:624 */

/* ---------------------------------------------------------------- quadrature + spline */

static double profile(double r, const tc_halo_setup *h)      /* src/setup.c:598-615 */
{
    /* the cool-core component scales with rho0 (rho0_cc = rho0 * Rho0_Fac), and rho0 is still being normalised when
     * the mass profile is first built: the factor is carried in rho0_cc / rho0 form by the callers (set_cool_core) */
    return tc_host_gas_profile(r, h->rho0, h->beta, h->rcore, h->rcut, h->rho0_cc, h->rc_cc);
}

/* src/setup.c:604-612: rho0_cc = rho0 * Param.Rho0_Fac, rc_cc = rc / Param.Rc_Fac for a cuspy halo of the
 * -DDOUBLE_BETA_COOL_CORES build; to be called whenever rho0 or rcore of the halo changes */
static void set_cool_core(const tc_parfile *par, tc_halo_setup *h)
{
    h->rho0_cc = h->rc_cc = 0;
    if (par->double_beta && h->have_cuspy) {
        h->rho0_cc = h->rho0 * par->rho0_fac;
        h->rc_cc = h->rcore / par->rc_fac;
    }
}

static double m_integrand(double r, const tc_halo_setup *h) { return 4 * PI * r * r * profile(r, h); }

/* 15-point Gauss-Legendre nodes/weights on [-1,1] (positive half + centre) */
static const double GLX[8] = {0.0, 0.2011940939974345, 0.3941513470775634, 0.5709721726085388, 0.7244177313601701,
                              0.8482065834104272, 0.9372733924007060, 0.9879925180204854};
static const double GLW[8] = {0.2025782419255613, 0.1984314853271116, 0.1861610000155622, 0.1662692058169939,
                              0.1395706779261543, 0.1071592204671719, 0.0703660474881081, 0.0307532419961173};

static double gl15(double a, double b, const tc_halo_setup *h)
{
    double c = 0.5 * (a + b), d = 0.5 * (b - a), s = GLW[0] * m_integrand(c, h);
    for (int i = 1; i < 8; i++) s += GLW[i] * (m_integrand(c - d * GLX[i], h) + m_integrand(c + d * GLX[i], h));
    return s * d;
}

static double integrate(double a, double b, const tc_halo_setup *h, int depth)
{
    double whole = gl15(a, b, h), m = 0.5 * (a + b);
    double halves = gl15(a, m, h) + gl15(m, b, h);
    if (depth > 30 || fabs(halves - whole) <= 1e-10 * fabs(halves)) return halves;
    return integrate(a, m, h, depth + 1) + integrate(m, b, h, depth + 1);
}

typedef struct { int n; double *x, *y, *y2; } spline_t;

/* natural cubic spline (second derivative zero at both ends), as gsl_interp_cspline */
static void spline_init(spline_t *s, const double *x, const double *y, int n)
{
    s->n = n;
    s->x = malloc(n * sizeof(double)); s->y = malloc(n * sizeof(double)); s->y2 = calloc(n, sizeof(double));
    memcpy(s->x, x, n * sizeof(double)); memcpy(s->y, y, n * sizeof(double));
    double *u = calloc(n, sizeof(double));
    for (int i = 1; i < n - 1; i++) {
        double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
        double p = sig * s->y2[i - 1] + 2.0;
        s->y2[i] = (sig - 1.0) / p;
        double dd = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
        u[i] = (6.0 * dd / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
    }
    for (int k = n - 2; k >= 0; k--) s->y2[k] = s->y2[k] * s->y2[k + 1] + u[k];
    free(u);
}

static double spline_eval(const spline_t *s, double x)
{
    int lo = 0, hi = s->n - 1;
    while (hi - lo > 1) { int m = (lo + hi) >> 1; if (s->x[m] > x) hi = m; else lo = m; }
    double h = s->x[hi] - s->x[lo];
    if (h == 0) return s->y[lo];
    double a = (s->x[hi] - x) / h, b = (x - s->x[lo]) / h;
    return a * s->y[lo] + b * s->y[hi] + ((a * a * a - a) * s->y2[lo] + (b * b * b - b) * s->y2[hi]) * (h * h) / 6.0;
}

static void spline_free(spline_t *s) { free(s->x); free(s->y); free(s->y2); memset(s, 0, sizeof(*s)); }

typedef struct { spline_t m_of_r, r_of_m; } mass_profile_t;

/* src/setup.c:643-701 */
static void setup_mass_profile(const tc_halo_setup *h, mass_profile_t *mp)
{
    static double m_table[NTABLE], r_table[NTABLE];
    const double rmin = 0.1, rmax = h->r_sample[0] * 1.1;
    const double log_dr = log10(rmax / rmin) / (NTABLE - 1);
    m_table[0] = r_table[0] = 0;
    for (int i = 1; i < NTABLE; i++) {
        r_table[i] = rmin * pow(10, log_dr * i);
        /* cumulative: previous value + the new shell (same integral, cheaper than restarting at 0) */
        m_table[i] = m_table[i - 1] + integrate(r_table[i - 1], r_table[i], h, 0);
        if (m_table[i] < m_table[i - 1]) m_table[i] = m_table[i - 1];
    }
    spline_init(&mp->m_of_r, r_table, m_table, NTABLE);
    spline_init(&mp->r_of_m, m_table, r_table, NTABLE);
}

static double mass_profile(const mass_profile_t *mp, const tc_halo_setup *h, double r)      /* src/setup.c:703-708 */
{
    return spline_eval(&mp->m_of_r, fmin(r, h->r_sample[0]));
}

/* ---------------------------------------------------------------- cosmology */

static const double cij[5][5] = {   /* Pierpaoli+ 01 Table 1, src/cosmo.c:71-77 */
    {546.67, -137.82, 94.083, -204.68, 111.51},
    {-1745.6, 627.22, -1175.2, 2445.7, -1341.7},
    {3928.8, -1519.3, 4015.8, -8415.3, 4642.1},
    {-4384.8, 1748.7, -5362.1, 11257., -6218.2},
    {1842.3, -765.53, 2507.7, -5210.7, 2867.5}};

int tc_setup_system(const tc_parfile *par, tc_setup *S)
{
    memset(S, 0, sizeof(*S));
    S->par = *par;
    /* Set_units */
    S->unit_length = par->unit_length; S->unit_mass = par->unit_mass; S->unit_vel = par->unit_vel;
    S->unit_time = S->unit_length / S->unit_vel;
    /* Set_cosmology */
    S->h_100 = 0.7; S->omega_m = 0.3; S->omega_l = 0.7;
    const double omega_0 = S->omega_m + S->omega_l;
    S->h0_cgs = 100 * S->h_100 * 1e5 / 1000 / KPC2CGS;
    const double z = par->redshift;
    const double ez = sqrt(S->omega_l + (1 - omega_0) * (1 + z) * (1 + z) + S->omega_m * (1 + z) * (1 + z) * (1 + z));
    const double hz = S->h0_cgs * ez;
    S->rho_crit = 3 * hz * hz / (8 * PI * GRAV_CGS);                              /* Critical_Density(z) */
    {
        const double x = S->omega_m - 0.2, y = S->omega_l;
        double r = 0;
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) r += cij[i][j] * pow(x, i) * pow(y, j);
        S->delta = S->omega_m * r;                                               /* Overdensity_Parameter() */
    }

    /* Setup(), src/setup.c:21-344 */
    const double bf = par->baryon_fraction, Xm = par->mass_ratio;
    tc_halo_setup *H = S->halo;
    H[0].mtotal200 = par->mtot200 / (1 + Xm);
    H[1].mtotal200 = par->mtot200 - H[0].mtotal200;
    S->nhalos = (Xm == 0) ? 1 : 2;
    for (int i = 0; i < S->nhalos; i++) {
        H[i].beta = BETA_DEFAULT;
        H[i].mass200[1] = H[i].mtotal200 / (1 + bf);
        H[i].mass200[0] = H[i].mtotal200 - H[i].mass200[1];
        const double mass = H[i].mtotal200 * S->unit_mass / MSOL2CGS;             /* Duffy+08, src/setup.c:512-521 */
        H[i].c_nfw = 5.74 * pow(mass / (2e12 / S->h_100), -0.097) * pow(1 + z, -0.47);
        H[i].r200 = pow(H[i].mtotal200 * S->unit_mass / (S->delta * S->rho_crit * FOURPITHIRD), 1. / 3.) / S->unit_length;
        H[i].rs = H[i].r200 / H[i].c_nfw;
        const double c = H[i].c_nfw;
        H[i].a_hernq = H[i].rs * sqrt(2 * (log(1 + c) - c / (1 + c)));
    }
    S->boxsize = floor(2 * R200_TO_RMAX_RATIO * H[0].r200);

    double mtot[2] = {0, 0};
    for (int i = 0; i < S->nhalos; i++) {
        H[i].r_sample[0] = H[i].r_sample[1] = H[i].r200 * 1.8;
        H[i].rcut = 1.4 * H[i].r200;
        if (i == 0) {
            H[i].r_sample[1] = S->boxsize / 2;
            H[i].r_sample[0] = sqrt(3) * S->boxsize / 2;
        }
        if (par->cuspy & (1 << i)) {                                                      /* src/setup.c:567-589 */
            H[i].rcore = H[i].rs / (par->double_beta ? 3 : 9);                            /* :583-587 */
            H[i].have_cuspy = 1;
        } else { H[i].rcore = H[i].rs / 3; H[i].have_cuspy = 0; }

        mass_profile_t mp;
        H[i].rho0 = 1;
        set_cool_core(par, &H[i]);
        setup_mass_profile(&H[i], &mp);
        H[i].rho0 = H[i].mass200[0] / mass_profile(&mp, &H[i], H[i].r200);
        set_cool_core(par, &H[i]);
        spline_free(&mp.m_of_r); spline_free(&mp.r_of_m);
        setup_mass_profile(&H[i], &mp);
        H[i].mass[0] = mass_profile(&mp, &H[i], H[i].r_sample[0]);
        spline_free(&mp.m_of_r); spline_free(&mp.r_of_m);

        const double a = H[i].a_hernq, rs_dm = H[i].r_sample[1], r200 = H[i].r200;
        H[i].mass_corr_fac = 1 / (1 + 2 * a / rs_dm + (a / rs_dm) * (a / rs_dm));
        H[i].mass[1] = H[i].mass200[1] * (1 + 2 * a / r200 + (a / r200) * (a / r200)) * H[i].mass_corr_fac;
        H[i].mtotal = H[i].mass[0] + H[i].mass[1];
        if (!bf) { H[i].mass[1] += H[i].mass[0]; H[i].mass[0] = 0; }
        S->mtotal += H[i].mtotal;
        mtot[0] += H[i].mass[0];
        mtot[1] += H[i].mass[1];
    }
    const int nDM = 0.5 * par->ntotal, nGas = 0.5 * par->ntotal;                 /* src/setup.c:189-193 */
    S->mpart[0] = bf ? mtot[0] / nGas : 0;
    S->mpart[1] = bf ? mtot[1] / nDM : S->mtotal / par->ntotal;
    for (int i = 0; i < S->nhalos; i++) {
        H[i].npart[0] = bf ? (long long)round(H[i].mass[0] / S->mpart[0]) : 0;
        H[i].npart[1] = (long long)round((bf ? H[i].mass[1] : H[i].mtotal) / S->mpart[1]);
        S->npart[0] += H[i].npart[0];
        S->npart[1] += H[i].npart[1];
    }
    S->grav_softening = pow(pow(H[0].r_sample[1], 3) / par->ntotal, 1 / 3.) / 7;  /* src/setup.c:266-268 */
    S->sub_first = S->nhalos;                                                    /* src/io.c:500-503 */
    if (Xm) {                                                                    /* src/setup.c:271-287 */
        const double d = 0.9 * (H[0].r200 + H[1].r200);
        H[0].d_com[0] = -1 * H[1].mtotal200 * d / par->mtot200;
        H[1].d_com[0] = d + H[0].d_com[0];
        H[0].d_com[1] = -1 * H[1].mtotal200 * par->impact_param / par->mtot200;
        H[1].d_com[1] = par->impact_param + H[0].d_com[1];
    }
    return 0;
}

/* Param / Halo scalars the hot path reads */
void tc_setup_to_model(const tc_setup *S, tcgpu_params *par, tcgpu_halo *halos)
{
    memset(par, 0, sizeof(*par));
    par->boxsize = S->boxsize; par->mpart_gas = S->mpart[0]; par->mtotal = S->mtotal;
    par->bfld_eta = S->par.bfld_eta; par->nhalos = S->nhalos;
    for (int i = 0; i < S->nhalos; i++) {
        memset(&halos[i], 0, sizeof(halos[i]));
        halos[i].mass_gas = S->halo[i].mass[0];
        for (int c = 0; c < 3; c++) halos[i].d_com[c] = S->halo[i].d_com[c];
        halos[i].rho0 = S->halo[i].rho0; halos[i].beta = S->halo[i].beta;
        halos[i].rcore = S->halo[i].rcore; halos[i].rcut = S->halo[i].rcut;
        halos[i].have_cuspy = S->halo[i].have_cuspy;
        halos[i].rho0_cc = S->halo[i].rho0_cc; halos[i].rc_cc = S->halo[i].rc_cc;
    }
}

/* ---------------------------------------------------------------- substructure
 *
 * Setup_Substructure, src/substructure.c:31-114 with its helpers (:116-558), build options
 * -DSUBSTRUCTURE -DSUBHOST=<subhost>, without SLOW_SUBSTRUCTURE / ADD_THIRD_SUBHALO / THIRD_HALO_ONLY /
 * REPORTSUBHALOS.  Subhalo bulk velocities (set_subhalo_bulkvel, :560-610) belong to the velocity stage,
 * which is out of scope; their four erand48 draws per subhalo are still consumed so that the stream
 * stays aligned with the reference's.
 */
#define DESNNGB_HOST 295                         /* src/globals.h:48 */
#define MIN_DENSITY_CONTRAST 3                   /* src/substructure.c:8 */

static double hernquist_density(double m, double a, double r)                  /* src/setup.c:715-718 */
{
    return m / (2 * PI) * a / (r * ((r + a) * (r + a) * (r + a)));             /* r * p3(r + a) */
}

static double sub_mass_function(const tc_setup *S, double m)                    /* src/substructure.c:471-482 */
{
    const double cc = 1, Am = 9.33e-4, alpha = -0.9, beta = 12.2715;
    const double z = S->par.redshift;
    const double mSub = m * S->unit_mass / MSOL2CGS;
    const double mHost = S->halo[S->subhost].mass200[1] * S->unit_mass / MSOL2CGS;
    const double x = mSub / mHost;
    return mHost * sqrt(1 + z) * cc * Am * pow(mSub, alpha) * exp(-beta * (x * x * x));      /* p3(x) is one factor */
}

static double sub_number_density(const tc_setup *S, double r)                   /* src/substructure.c:495-500 */
{
    const double ac = 0.244 * S->halo[S->subhost].c_nfw, alpha = 2, beta = 2.75;
    return (1 + ac) * pow(r, beta) / (1 + ac * pow(r, alpha));
}

static double sub_inverted_number_density(const tc_setup *S, double q)          /* src/substructure.c:502-519 */
{
    double left = 0, right = S->halo[S->subhost].r200, r = 0, delta = 1e300;
    while (fabs(delta) > 1e-3) {
        r = left + 0.5 * (right - left);
        delta = sub_number_density(S, r) - q;
        if (delta > 0) right = r; else left = r;
    }
    return r;
}

static double nfw_mass(const tc_setup *S, double c_nfw, double rs, double r)    /* src/substructure.c:542-553 */
{
    const double delta_s = S->delta / 3 * (c_nfw * c_nfw * c_nfw) / (log(1 + c_nfw) - c_nfw / (1 + c_nfw));   /* p3() groups */
    const double rho_crit0 = 3. / 8. / PI / GRAV_CGS * S->h0_cgs * S->h0_cgs;  /* src/cosmo.c:20 */
    const double unit_density = S->unit_mass / (S->unit_length * S->unit_length * S->unit_length);
    const double rho_s = delta_s * rho_crit0 / unit_density;
    return 4 * PI * rho_s * (rs * rs * rs) * (log((rs + r) / rs) - r / (rs + r));
}

static double nfw_scale_radius(const tc_setup *S, double c_nfw, double M_t, double r)   /* src/substructure.c:521-540 */
{
    double left = 0, right = 10 * S->halo[S->subhost].r_sample[0], rs = 0, delta = 1e300;
    int guard = 0;
    while (fabs(delta) > 1e-3) {
        rs = left + 0.5 * (right - left);
        delta = nfw_mass(S, c_nfw, rs, r) - M_t;
        if (delta > 0) right = rs; else left = rs;
        if (++guard > 4000) break;                       /* the reference has no guard */
    }
    return rs;
}

static double sub_sampling_radius(const tc_setup *S, int i, double d)           /* src/substructure.c:434-456 */
{
    const double rho_host = hernquist_density(S->halo[0].mass[1], S->halo[0].a_hernq, d);
    const double m = S->halo[i].mass[1], a = S->halo[i].a_hernq;
    /* With a == 0 (see set_subhalo_properties) the subhalo density is 0 everywhere, delta stays -1 and r
     * halves until it underflows to 0, where 0/0 = NaN ends the loop: the reference then returns 0. */
    double left = 0, right = 10 * S->halo[0].r200, r = 0, delta = 1e300;
    int guard = 0;
    while (fabs(delta) > 1e-3) {
        r = left + 0.5 * (right - left);
        delta = (hernquist_density(m, a, r) - rho_host) / rho_host;
        if (delta < 0) right = r; else left = r;
        if (++guard > 4000) break;
    }
    return r;
}

static double sub_tidal_radius(const tc_setup *S, int i, double r)              /* src/substructure.c:459-468 */
{
    const double m_sub = S->halo[i].mass[1], m_host = S->halo[S->subhost].mass200[1], a = S->halo[S->subhost].a_hernq;
    const double fac = 2 * r * r / ((a + r) * (a + r)) * (1 - a * r * r / ((r + a) * (r + a) * (r + a)));
    return r * pow(m_sub / (m_host * fac), 1.0 / 3.0);
}

static double sub_concentration(const tc_setup *S, int i)                       /* src/setup.c:529-549 */
{
    const double mass_sub = S->halo[i].mass[1] * S->unit_mass / MSOL2CGS;
    const double aR = 0.237, c1 = 232.15, c2 = -181.74, a1 = 0.0146, a2 = 0.008;
    const tc_halo_setup *host = &S->halo[S->subhost], *h = &S->halo[i];
    const double dx = host->d_com[0] - h->d_com[0], dy = host->d_com[1] - h->d_com[1], dz = host->d_com[2] - h->d_com[2];
    const double d_vir = sqrt(dx * dx + dy * dy + dz * dz) / S->halo[0].r200;
    double c = pow(d_vir, -aR) * (c1 * pow(mass_sub, -a1) + c2 * pow(mass_sub, -a2));
    return c / (1 + S->par.redshift);
}

static void set_subhalo_masses(tc_setup *S, unsigned short seed[3])             /* src/substructure.c:116-183 */
{
    const tc_halo_setup *host = &S->halo[S->subhost];
    const double min_mass = 10 * DESNNGB_HOST * (S->mpart[0] + S->mpart[1]);    /* MIN_SUBHALO_MASS, :7 */
    const double mass_limit = host->mass200[1] * S->sub_mass_fraction;
    const double qmax = sub_mass_function(S, min_mass) / min_mass;
    const double max_subhalo_mass = S->sub_mass_fraction * host->mass[1] / 10;
    int i = S->sub_first;
    while (S->sub_mtotal < mass_limit && i < 70) {
        double mDM = 0;
        int j;
        for (j = 0; j < 10000; j++) {                                           /* rejection sampling */
            mDM = min_mass + erand48(seed) * (host->mass200[1] - min_mass);
            const double q = sub_mass_function(S, mDM) / mDM;
            const double lower_bound = qmax * erand48(seed);
            if (mass_limit - S->sub_mtotal < min_mass) { mDM = min_mass; break; }
            if (S->sub_mtotal + mDM > 1.05 * mass_limit) continue;
            if (mDM > max_subhalo_mass) continue;
            if (q >= lower_bound) break;
        }
        /* sic: when all 10 000 tries are rejected the loop leaves j == 10000, this test does not fire and the
         * LAST draw is kept whatever it was (about one subhalo in three at config-4 resolution, where a try
         * is accepted with p ~ 1e-4) -- the reference's behaviour, kept */
        if (j == 9999) mDM = min_mass;
        S->halo[i].mass[1] = mDM;
        S->sub_mtotal += mDM;
        S->sub_nhalos++;
        i++;
    }
    /* sic (src/substructure.c:180): with two clusters this is 2 + nsub; with a single cluster (nhalos == 1,
     * first subhalo at index 1) it is nsub, i.e. the last sampled subhalo is never set up or populated */
    S->nhalos += i - 2;
}

static void set_subhalo_position(tc_setup *S, int i, unsigned short seed[3])    /* src/substructure.c:189-220 */
{
    const tc_halo_setup *host = &S->halo[S->subhost];
    const double q = erand48(seed);
    const double r = host->r200 * sub_inverted_number_density(S, q);
    const float theta = acos(2 * erand48(seed) - 1);
    const float phi = 2 * PI * erand48(seed);
    const double x = r * sin(theta) * cos(phi), y = r * sin(theta) * sin(phi), z = r * cos(theta);
    S->halo[i].d_com[0] = (float)(x + host->d_com[0]);
    S->halo[i].d_com[1] = (float)(y + host->d_com[1]);
    S->halo[i].d_com[2] = (float)(z + host->d_com[2]);
}

static void set_subhalo_properties(tc_setup *S, int i)                          /* src/substructure.c:278-375 */
{
    tc_halo_setup *h = &S->halo[i];
    const tc_halo_setup *host = &S->halo[S->subhost];
    const double dx = host->d_com[0] - h->d_com[0], dy = host->d_com[1] - h->d_com[1], dz = host->d_com[2] - h->d_com[2];
    const double r_i = sqrt(dx * dx + dy * dy + dz * dz);
    double a = host->a_hernq / 10, r200 = host->r200, c_nfw = 0, rsample = 0;
    int cnt = 0;
    for (;;) {
        const double last_a = a;
        /* sampling_radius() reads Halo[i].A_hernq, which the reference only assigns after this loop: the
         * first subhalo set-up sees 0 (calloc), a resampled one its previous value.  Reproduced as is. */
        rsample = fmax(sub_sampling_radius(S, i, r_i), sub_tidal_radius(S, i, r_i));
        rsample = fmin(rsample, r200 * 0.5);
        /* Concentration_parameter() evaluates the Duffy fit on Halo[i].Mtotal200 first; its value is
         * overwritten by the subhalo fit (src/setup.c:529-549) */
        c_nfw = sub_concentration(S, i);
        h->rs = nfw_scale_radius(S, c_nfw, h->mass[1], rsample);
        a = h->rs * sqrt(2 * (log(1 + c_nfw) - c_nfw / (1 + c_nfw)));
        r200 = h->rs * c_nfw;
        if (fabs((last_a - a) / a) < 1e-4) break;
        if (cnt++ > 100) break;
    }
    h->r_sample[0] = h->r_sample[1] = rsample;
    h->a_hernq = a;
    h->r200 = r200;
    h->c_nfw = c_nfw;
    const double r_strip = 0;
    h->rcut = 0.6 * h->r_sample[0];
    h->mass200[1] = nfw_mass(S, c_nfw, h->rs, r200);
    if (r_i > r_strip) h->mass200[0] = h->mass200[1] / (1 / S->par.baryon_fraction - 1);
    h->mtotal200 = h->mass200[0] + h->mass200[1];
    h->mass_corr_fac = 1 / (1 + 2 * a / r200 + (a / r200) * (a / r200));
    h->beta = 2.0 / 3.0;
    if (i < 31 && (S->par.cuspy & (1 << i))) {                                              /* src/setup.c:567-589 */
        h->rcore = h->rs / (S->par.double_beta ? 3 : 9);
        h->have_cuspy = 1;
    } else { h->rcore = h->rs / 3; h->have_cuspy = 0; }
    const double rc = h->rcore;
    h->rho0 = h->mass200[0] / (4 * PI * (rc * rc * rc)) / (r200 / rc - atan(r200 / rc));
    set_cool_core(&S->par, h);
    h->mass[0] = 0;
    h->is_stripped = 1;
    if (r_i > r_strip) {
        mass_profile_t mp;
        setup_mass_profile(h, &mp);
        h->is_stripped = 0;
        h->mass[0] = mass_profile(&mp, h, h->r_sample[0]);
        spline_free(&mp.m_of_r); spline_free(&mp.r_of_m);
    }
    h->mtotal = h->mass[0] + h->mass[1];
}

static int reject_subhalo(const tc_setup *S, int i)                             /* src/substructure.c:228-270 */
{
    int resample = 0;
    const tc_halo_setup *h = &S->halo[i], *host = &S->halo[S->subhost];
    for (int j = S->sub_first; j < i; j++) {
        const double d0 = h->d_com[0] - S->halo[j].d_com[0], d1 = h->d_com[1] - S->halo[j].d_com[1],
                     d2 = h->d_com[2] - S->halo[j].d_com[2];
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2, size = h->r_sample[0] + S->halo[j].r_sample[0];
        if (r2 < size * size) resample = 1;
    }
    const double dx = h->d_com[0] - host->d_com[0], dy = h->d_com[1] - host->d_com[1], dz = h->d_com[2] - host->d_com[2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double rho_host = hernquist_density(S->halo[0].mass[1], S->halo[0].a_hernq, r);
    const double rho_sub = hernquist_density(h->mass[1], h->a_hernq, 3 * S->grav_softening);
    if (rho_sub < rho_host * MIN_DENSITY_CONTRAST) resample = 1;
    if (r > host->r200) resample = 1;
    return resample;
}

int tc_setup_substructure(tc_setup *S, int subhost, unsigned short seed[3])
{
    if (subhost < 0 || subhost >= S->nhalos) return 1;
    if (!(S->par.baryon_fraction > 0)) return 2;
    S->subhost = subhost;
    S->sub_first = (S->par.mass_ratio != 0) ? 2 : 1;
    S->sub_mass_fraction = 0.22 * sqrt(1 + S->par.redshift);                    /* src/substructure.c:485-492 */
    S->sub_mtotal = 0; S->sub_nhalos = 0; S->sub_npart[0] = S->sub_npart[1] = 0;
    set_subhalo_masses(S, seed);
    if (S->nhalos > TC_SETUP_MAXHALOS) return 3;
    for (int i = S->sub_first; i < S->nhalos; i++) {
        int tries = 0;
        do {
            set_subhalo_position(S, i, seed);
            set_subhalo_properties(S, i);
            if (++tries > 100000) return 4;
        } while (reject_subhalo(S, i));
        for (int k = 0; k < 4; k++) erand48(seed);      /* set_subhalo_bulkvel's draws, src/substructure.c:572-577 */
    }
    /* set_subhalo_particle_numbers, src/substructure.c:378-408 */
    long long sub_ntotal = 0;
    for (int i = S->sub_first; i < S->nhalos; i++) {
        tc_halo_setup *h = &S->halo[i];
        long long nDM = (long long)round(h->mass[1] / S->mpart[1]);
        long long nGas = S->mpart[0] == 0 ? 0 : (long long)round(h->mass[0] / S->mpart[0]);
        h->npart[0] = nGas; h->npart[1] = nDM;
        sub_ntotal += nGas + nDM;
        S->sub_npart[0] += nGas; S->sub_npart[1] += nDM;
    }
    S->halo[subhost].npart[0] -= S->sub_npart[0];
    S->halo[subhost].npart[1] -= S->sub_npart[1];
    (void)sub_ntotal;
    return (S->halo[subhost].npart[0] < 0 || S->halo[subhost].npart[1] < 0) ? 5 : 0;
}

/* ---------------------------------------------------------------- sampling */

/* src/positions.c:333-388, gas branch (type 0); arguments arrive as float like in the reference */
static int halo_containing_gas(const tc_setup *S, float x, float y, float z)
{
    if (x > S->boxsize || y > S->boxsize || z > S->boxsize) return -1;
    int best = 0;
    double rho_max = 0;
    for (int j = 0; j < S->nhalos; j++) {
        const tc_halo_setup *h = &S->halo[j];
        if (h->is_stripped) continue;                                          /* src/positions.c:369-370 */
        float r = sqrt((x - h->d_com[0]) * (x - h->d_com[0]) + (y - h->d_com[1]) * (y - h->d_com[1])
                       + (z - h->d_com[2]) * (z - h->d_com[2]));
        double rho = profile(r, h);
        if (rho > rho_max && r < h->r_sample[0]) { best = j; rho_max = rho; }
    }
    return best;
}

/* Gas positions in [0, boxsize] + ids: Make_positions (gas part), Make_IDs, Shift_Origin.
 * `nthreads` reproduces the reference's per-thread erand48 streams (src/main.c:15-26: seed
 * {0,0,14041981*(tid+1)} truncated to 16 bit, one burn-in draw) and its static loop partition. */
int tc_sample_gas(const tc_setup *S, int nthreads, float *pos, int32_t *id)
{
    return tc_sample_gas_seeded(S, nthreads, NULL, pos, id);
}

/* src/main.c:20-21 for thread `tid` */
void tc_thread_seed(int tid, unsigned short seed[3])
{
    seed[0] = seed[1] = 0;
    seed[2] = (unsigned short)(14041981 * (tid + 1));
    erand48(seed);
}

/* seed0: state of thread 0's stream if something (the substructure set-up) has drawn from it already */
int tc_sample_gas_seeded(const tc_setup *S, int nthreads, const unsigned short seed0[3], float *pos, int32_t *id)
{
    if (nthreads < 1) nthreads = 1;
    const double boxhalf = S->boxsize / 2;
    unsigned short (*seed)[3] = malloc(sizeof(unsigned short[3]) * nthreads);
    for (int t = 0; t < nthreads; t++) tc_thread_seed(t, seed[t]);
    if (seed0) memcpy(seed[0], seed0, sizeof(unsigned short[3]));
    long long base = 0;
    for (int i = 0; i < S->nhalos; i++) {
        const tc_halo_setup *h = &S->halo[i];
        mass_profile_t mp;
        setup_mass_profile(h, &mp);
        const long long n = h->npart[0];
        for (int t = 0; t < nthreads; t++) {                        /* static schedule of `omp parallel for` */
            long long q = n / nthreads, r = n % nthreads;
            long long lo = t * q + (t < r ? t : r), hi = lo + q + (t < r ? 1 : 0);
            for (long long ip = lo; ip < hi; ip++) {
                for (;;) {
                    double theta = acos(2 * erand48(seed[t]) - 1);
                    double phi = 2 * PI * erand48(seed[t]);
                    double m = erand48(seed[t]) * h->mass[0];
                    double r_ = spline_eval(&mp.r_of_m, m);
                    double x = r_ * sin(theta) * cos(phi), y = r_ * sin(theta) * sin(phi), z = r_ * cos(theta);
                    if (i != halo_containing_gas(S, x + h->d_com[0], y + h->d_com[1], z + h->d_com[2])) continue;
                    if (x < -boxhalf || x > boxhalf || y < -boxhalf || y > boxhalf || z < -boxhalf || z > boxhalf) continue;
                    float *p = pos + 3 * (base + ip);
                    p[0] = (float)x; p[1] = (float)y; p[2] = (float)z;
                    break;
                }
            }
        }
        spline_free(&mp.m_of_r); spline_free(&mp.r_of_m);
        /* Shift_Origin, first part: src/setup.c:437-470 */
        const float dx = h->d_com[0], dy = h->d_com[1], dz = h->d_com[2];
        for (long long ip = 0; ip < n; ip++) {
            float *p = pos + 3 * (base + ip);
            p[0] += dx; p[1] += dy; p[2] += dz;
        }
        base += n;
    }
    free(seed);
    const long long ngas = S->npart[0];
    const float boxsize = S->boxsize, boxHalf = boxsize / 2;               /* src/setup.c:429-430,472-497 */
    for (long long ip = 0; ip < ngas; ip++)
        for (int c = 0; c < 3; c++) {
            float *p = &pos[3 * ip + c];
            *p += boxHalf;
            while (*p > boxsize) *p -= boxsize;
            while (*p < 0) *p += boxsize;
        }
    /* Make_IDs, src/ids.c:16-39 */
    size_t delta = 127;
    for (;;) if ((ngas % ++delta) == 0) break;
    int idv = 1 - (int)delta, start = 1;
    for (long long ip = 0; ip < ngas; ip++) {
        idv += (int)delta;
        if (idv > ngas) { start++; idv = start; }
        id[ip] = idv;
    }
    return 0;
}
