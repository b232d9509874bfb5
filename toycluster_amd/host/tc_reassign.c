/* tc_reassign.c -- the step right behind the SPH path: after relaxation every gas
 * particle is attributed to the halo whose model density is highest at its position and the gas block
 * is reordered by halo, so that Halo[i].Gas / Halo[i].SphP are contiguous again.
 *
 *   Reassign_particles_to_halos   src/positions.c:264-331
 *   Halo_containing (gas branch)  src/positions.c:333-388
 *   sort_particles                src/positions.c:399-445  (index sort by halo id + permutation)
 *   Qsort_Index                   src/sort.c:185-195 -> gsl_heapsort_index (GNU GSL sort/sortind.c)
 *
 * The halo ids have massive ties and the reference's index heapsort is not stable, so the order of the
 * particles inside one halo is whatever that sift-down scheme produces; the same scheme is used here so
 * that the gas block comes out in the reference's file order, not just the same partition.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "tc_host.h"

static double gas_profile(double r, const tcgpu_halo *h)       /* src/setup.c:598-615 */
{
    return tc_host_gas_profile(r, h->rho0, h->beta, h->rcore, h->rcut, h->rho0_cc, h->rc_cc);
}

/* Gas branch of Halo_containing: x, y, z are relative to the box centre (src/positions.c:273-275).
 * Returns -1 for a coordinate beyond the box like the reference. */
int tc_halo_containing_gas(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample,
                           float x, float y, float z)
{
    if (x > par->boxsize || y > par->boxsize || z > par->boxsize) return -1;
    int best = 0;
    double rho_max = 0;
    for (int j = 0; j < par->nhalos; j++) {
        const tcgpu_halo *h = &halos[j];
        const double dx = x - h->d_com[0], dy = y - h->d_com[1], dz = z - h->d_com[2];
        const float r = sqrt(dx * dx + dy * dy + dz * dz);
        const double rho = gas_profile(r, h);
        if (rho > rho_max && r < r_sample[j]) { best = j; rho_max = rho; }
    }
    return best;
}

/* Index heapsort over int keys: p[] becomes a permutation that sorts key[] ascending.  The published
 * scheme, literally: max-heap on p[0..last], the children of slot k taken at 2k and 2k+1 (so slot 0 is
 * compared with itself and slot 1), built from k = last/2 down to 0, then the root is swapped with the
 * last entry and sifted down.  Ties are ordered by this procedure and nothing else. */
static void sift_down(size_t *p, const int32_t *key, size_t last, size_t k)
{
    const size_t pk = p[k];
    while (k <= last / 2) {
        size_t j = 2 * k;
        if (j < last && key[p[j]] < key[p[j + 1]]) j++;
        if (!(key[pk] < key[p[j]])) break;
        p[k] = p[j];
        k = j;
    }
    p[k] = pk;
}

void tc_heapsort_index_i32(size_t *p, const int32_t *key, size_t n)
{
    if (n == 0) return;
    for (size_t i = 0; i < n; i++) p[i] = i;
    size_t last = n - 1;
    size_t k = last / 2 + 1;
    do {
        k--;
        sift_down(p, key, last, k);
    } while (k > 0);
    while (last > 0) {
        const size_t t = p[0]; p[0] = p[last]; p[last] = t;
        last--;
        sift_down(p, key, last, 0);
    }
}

/* pos: f32[3n] in [0, boxsize].  Outputs: halo_id[n] (before sorting), perm[n] (new slot i takes old
 * particle perm[i]), npart[nhalos].  Returns 0, or 3 if a particle lies outside the box (the reference
 * would index npart[-1] there). */
int tc_reassign_particles_to_halos(const tcgpu_params *par, const tcgpu_halo *halos, const double *r_sample,
                                   size_t n, const float *pos, int32_t *halo_id, size_t *perm, long long *npart)
{
    const float boxhalf = 0.5 * par->boxsize;
    memset(npart, 0, sizeof(*npart) * (size_t)(par->nhalos > 0 ? par->nhalos : 1));
    for (size_t i = 0; i < n; i++) {
        const float x = pos[3 * i] - boxhalf, y = pos[3 * i + 1] - boxhalf, z = pos[3 * i + 2] - boxhalf;
        const int h = tc_halo_containing_gas(par, halos, r_sample, x, y, z);
        if (h < 0) return 3;
        halo_id[i] = h;
        npart[h]++;
    }
    tc_heapsort_index_i32(perm, halo_id, n);
    return 0;
}

/* out[i] = in[perm[i]] for `width` bytes per particle (what the reference's in-place cycle walk of
 * src/positions.c:405-443 leaves in P[] and SphP[]) */
int tc_permute_rows(void *data, size_t n, size_t width, const size_t *perm)
{
    const size_t bytes = n * width;
    unsigned char *tmp = malloc(bytes ? bytes : 1);
    if (!tmp) return 4;
    const unsigned char *in = data;
    for (size_t i = 0; i < n; i++) memcpy(tmp + i * width, in + perm[i] * width, width);
    memcpy(data, tmp, n * width);
    free(tmp);
    return 0;
}
