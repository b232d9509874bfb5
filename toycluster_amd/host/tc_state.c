/* tc_state.c -- the hot-path state file: everything Regularise_sph_particles() reads besides the
 * per-particle SPH fields.  Little-endian, fixed layout:
 *   char magic[8] = "TCSTATE2" (2: tcgpu_halo carries the cool-core component); int64 ngas; tcgpu_params par; tcgpu_halo halos[par.nhalos];
 *   float pos[3*ngas]; int32 id[ngas]
 *   optional trailer: double r_sample[par.nhalos]  (Halo[i].R_Sample[0]; only the halo reassignment
 *   behind the path reads it, src/positions.c:378)
 */
#include <stdlib.h>
#include <string.h>
#include "tc_host.h"

static const char MAGIC[8] = {'T', 'C', 'S', 'T', 'A', 'T', 'E', '2'};

void tc_free_state(tc_state *st)
{
    free(st->halos); free(st->pos); free(st->id); free(st->r_sample); free(st->r_sample_dm);
    memset(st, 0, sizeof(*st));
}

int tc_read_state(const char *filename, tc_state *st, char *err, size_t errlen)
{
    memset(st, 0, sizeof(*st));
    FILE *fp = fopen(filename, "rb");
    if (!fp) { snprintf(err, errlen, "state file %s not found", filename); return 1; }
    char magic[8];
    int ok = fread(magic, 1, 8, fp) == 8 && !memcmp(magic, MAGIC, 8);
    ok = ok && fread(&st->ngas, sizeof(int64_t), 1, fp) == 1 && fread(&st->par, sizeof(st->par), 1, fp) == 1;
    ok = ok && st->ngas > 0 && st->par.nhalos >= 0 && st->par.nhalos <= 4096;
    if (ok) {
        size_t n = (size_t)st->ngas, nh = (size_t)st->par.nhalos;
        st->halos = malloc(sizeof(tcgpu_halo) * (nh ? nh : 1));
        st->pos = malloc(3 * n * sizeof(float));
        st->id = malloc(n * sizeof(int32_t));
        ok = st->halos && st->pos && st->id;
        ok = ok && fread(st->halos, sizeof(tcgpu_halo), nh, fp) == nh;
        ok = ok && fread(st->pos, sizeof(float), 3 * n, fp) == 3 * n;
        ok = ok && fread(st->id, sizeof(int32_t), n, fp) == n;
        if (ok && nh) {                                   /* optional trailer */
            st->r_sample = malloc(nh * sizeof(double));
            if (!st->r_sample || fread(st->r_sample, sizeof(double), nh, fp) != nh) {
                free(st->r_sample);
                st->r_sample = NULL;
            }
        }
    }
    fclose(fp);
    if (!ok) { snprintf(err, errlen, "state file %s is malformed", filename); tc_free_state(st); return 2; }
    return 0;
}

int tc_write_state(const char *filename, const tc_state *st)
{
    FILE *fp = fopen(filename, "wb");
    if (!fp) return 5;
    size_t n = (size_t)st->ngas, nh = (size_t)st->par.nhalos;
    int ok = fwrite(MAGIC, 1, 8, fp) == 8 && fwrite(&st->ngas, sizeof(int64_t), 1, fp) == 1;
    ok = ok && fwrite(&st->par, sizeof(st->par), 1, fp) == 1 && fwrite(st->halos, sizeof(tcgpu_halo), nh, fp) == nh;
    ok = ok && fwrite(st->pos, sizeof(float), 3 * n, fp) == 3 * n && fwrite(st->id, sizeof(int32_t), n, fp) == n;
    if (ok && st->r_sample) ok = fwrite(st->r_sample, sizeof(double), nh, fp) == nh;
    return (fclose(fp) == 0 && ok) ? 0 : 6;
}
