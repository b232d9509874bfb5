/*
 * tc_oracle.c -- CPU restatement of Toycluster's SPH / WVT hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see tc_oracle.h).  "Parity unpinned" except for the
 * Peano-key known answers of SURVEY.md 8c.
 *
 * Build: gcc -std=c99 -O2 -fopenmp (same language mode as the reference
 * Makefile:82, hence -ffp-contract=off and no FMA on baseline x86-64), so that
 * the mixed f32/f64 arithmetic below rounds exactly where the reference's does.
 * Each routine names the reference lines whose arithmetic it restates; the data
 * layout (structure-of-arrays, explicit state handle) is this repo's own.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <stdbool.h>
#include <omp.h>
#include "tc_oracle.h"

typedef unsigned __int128 u128;

/* globals.h:60-63 */
#define ORC_PI 3.14159265358979323846
#define ORC_SQRT3 1.73205080756887719
#define ORC_FOURPITHIRD 4.18879032135009765
#define ORC_NTRIPLETS 42 /* peano.h:1 : 128/3 */

/* ---------------------------------------------------------------- attribution experiment (NOT reference behaviour)
 * tools/attribute_tail.py runs this oracle against itself with ONE of the GPU library's documented arithmetic
 * deviations (DESIGN.md "Numerics") injected at a time, to see which of them -- if any -- owns the parity tail of the
 * GPU-vs-oracle fuzz.  orc_dev == 0 (the default, and the only mode the parity tests use) is the faithful restatement. */
static int orc_dev = 0;
void orc_set_deviation(int mask) { orc_dev = mask; }

static inline uint64_t dev_hash(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
/* value moved by -1, 0 or +1 f64 ulp, decided by a hash of its bits (deterministic, input-dependent) */
static inline double dev_ulp(double v)
{
    uint64_t b; memcpy(&b, &v, 8);
    uint64_t h = dev_hash(b) % 3;
    if (h == 0 || !isfinite(v) || v == 0) return v;
    return nextafter(v, h == 1 ? INFINITY : -INFINITY);
}
static inline double dev_pow(double x, double y)
{
    double r = pow(x, y);
    return (orc_dev & ORC_DEV_POW_ULP) ? dev_ulp(r) : r;
}

struct orc_node {          /* tree.c:5-11 */
    uint32_t bitfield;
    int32_t dnext;
    float pos[3];
    int32_t npart;
    float size;
};

struct orc_state {
    int n;
    double box, mpart, mtotal;
    int nhalos;
    orc_halo *halo;
    int nthreads;
    /* per particle */
    float *pos;            /* 3n */
    int32_t *id;
    float *hsml, *rho, *vhf, *rhom;
    float *apot, *bfld;    /* 3n each */
    int32_t *tparent;
    u128 *key;
    /* tree */
    struct orc_node *tree;
    int nnodes, maxnodes;
    /* stats */
    double st_queries, st_iters, st_pairs;
    long st_trunc_density, st_trunc_sweep;   /* queries that filled the NGBMAX list (tree.c:91-92) */
    /* ORC_DEV_EXACT_BALL only: a uniform grid over the positions the tree was built on */
    int gdim;
    int32_t *gstart, *gitem;
};

/* ---------------------------------------------------------------- life cycle */

orc_state *orc_create(int npart, double boxsize, double mpart_gas, double mtotal,
                      int nhalos, const orc_halo *halos, int nthreads)
{
    orc_state *s = calloc(1, sizeof(*s));
    s->n = npart; s->box = boxsize; s->mpart = mpart_gas; s->mtotal = mtotal;
    s->nhalos = nhalos;
    s->halo = malloc(sizeof(orc_halo) * (nhalos > 0 ? nhalos : 1));
    memcpy(s->halo, halos, sizeof(orc_halo) * nhalos);
    s->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
    size_t n = npart;
    s->pos = calloc(3 * n, sizeof(float));
    s->id = calloc(n, sizeof(int32_t));
    s->hsml = calloc(n, sizeof(float));
    s->rho = calloc(n, sizeof(float));
    s->vhf = calloc(n, sizeof(float));
    s->rhom = calloc(n, sizeof(float));
    s->apot = calloc(3 * n, sizeof(float));
    s->bfld = calloc(3 * n, sizeof(float));
    s->tparent = calloc(n, sizeof(int32_t));
    s->key = calloc(n, sizeof(u128));
    s->maxnodes = (int)(npart * 0.7);          /* tree.c:3,341 */
    if (s->maxnodes < 64) s->maxnodes = 64;     /* tiny fixtures only */
    s->tree = calloc(s->maxnodes, sizeof(struct orc_node));
    return s;
}

void orc_destroy(orc_state *s)
{
    if (!s) return;
    free(s->halo); free(s->pos); free(s->id); free(s->hsml); free(s->rho); free(s->vhf);
    free(s->rhom); free(s->apot); free(s->bfld); free(s->tparent); free(s->key); free(s->tree);
    free(s->gstart); free(s->gitem);
    free(s);
}

void orc_set_particles(orc_state *s, const float *pos, const int32_t *id, const float *hsml)
{
    size_t n = s->n;
    memcpy(s->pos, pos, 3 * n * sizeof(float));
    if (id) memcpy(s->id, id, n * sizeof(int32_t));
    else for (size_t i = 0; i < n; i++) s->id[i] = (int32_t)(i + 1);
    if (hsml) memcpy(s->hsml, hsml, n * sizeof(float));
    else memset(s->hsml, 0, n * sizeof(float));   /* setup.c:248-250: SphP zeroed */
    memset(s->rho, 0, n * sizeof(float));
    memset(s->vhf, 0, n * sizeof(float));
    memset(s->rhom, 0, n * sizeof(float));
}

void orc_get_particles(const orc_state *s, float *pos, int32_t *id, float *hsml, float *rho,
                       float *varhsmlfac, float *rho_model)
{
    size_t n = s->n;
    if (pos) memcpy(pos, s->pos, 3 * n * sizeof(float));
    if (id) memcpy(id, s->id, n * sizeof(int32_t));
    if (hsml) memcpy(hsml, s->hsml, n * sizeof(float));
    if (rho) memcpy(rho, s->rho, n * sizeof(float));
    if (varhsmlfac) memcpy(varhsmlfac, s->vhf, n * sizeof(float));
    if (rho_model) memcpy(rho_model, s->rhom, n * sizeof(float));
}

void orc_set_apot(orc_state *s, const float *a) { memcpy(s->apot, a, 3 * (size_t)s->n * sizeof(float)); }
void orc_get_apot(const orc_state *s, float *a) { memcpy(a, s->apot, 3 * (size_t)s->n * sizeof(float)); }
void orc_get_bfld(const orc_state *s, float *b) { memcpy(b, s->bfld, 3 * (size_t)s->n * sizeof(float)); }

/* ---------------------------------------------------------------- Peano keys */

/* Shared first half of peano.c:134-177 and :217-260: scale to 2^63, Skilling
 * "inverse undo" over q = 2^63 .. 2, then Gray encode.  Axis order {y,z,x}. */
static void hilbert_transpose(double x, double y, double z, uint64_t X[3])
{
    const uint64_t m = 1UL << 63;
    X[0] = y * m; X[1] = z * m; X[2] = x * m;

    for (uint64_t q = m; q > 1; q >>= 1) {
        uint64_t P = q - 1;
        if (X[0] & q) X[0] ^= P;
        for (int i = 1; i < 3; i++) {
            if (X[i] & q) {
                X[0] ^= P;
            } else {
                uint64_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
    uint64_t t = X[2];
    for (int i = 1; i < 64; i <<= 1) X[2] ^= X[2] >> i;
    t ^= X[2];
    for (int i = 1; i >= 0; i--) X[i] ^= t;
}

/* peano.c:181-200 */
static u128 peano_key128(double x, double y, double z)
{
    uint64_t X[3];
    hilbert_transpose(x, y, z, X);
    u128 key = 0;
    X[1] >>= 1; X[2] >>= 2;
    for (int i = 0; i < ORC_NTRIPLETS + 1; i++) {
        uint64_t col = ((X[0] & 0x8000000000000000UL) | (X[1] & 0x4000000000000000UL)
                        | (X[2] & 0x2000000000000000UL)) >> 61;
        key <<= 3;
        X[0] <<= 1; X[1] <<= 1; X[2] <<= 1;
        key |= col;
    }
    key <<= 2;
    return key;
}

/* peano.c:264-283 */
static u128 reversed_peano_key128(double x, double y, double z)
{
    uint64_t X[3];
    hilbert_transpose(x, y, z, X);
    u128 key = 0;
    X[0] >>= 18; X[1] >>= 19; X[2] >>= 20;
    for (int i = 0; i < ORC_NTRIPLETS + 1; i++) {
        uint64_t col = ((X[0] & 0x4) | (X[1] & 0x2) | (X[2] & 0x1));
        key <<= 3;
        key |= col;
        X[0] >>= 1; X[1] >>= 1; X[2] >>= 1;
    }
    key <<= 3;
    return key;
}

void orc_peano_key(double x, double y, double z, uint64_t *hi, uint64_t *lo)
{
    u128 k = peano_key128(x, y, z);
    *hi = (uint64_t)(k >> 64); *lo = (uint64_t)k;
}

void orc_reversed_peano_key(double x, double y, double z, uint64_t *hi, uint64_t *lo)
{
    u128 k = reversed_peano_key128(x, y, z);
    *hi = (uint64_t)(k >> 64); *lo = (uint64_t)k;
}

/* ---------------------------------------------------------------- index heapsort */

/* GSL gsl_heapsort_index (published algorithm, sort/sortind.c), specialised to
 * 128-bit keys with the comparator of peano.c:33-39.  Reached from sort.c:192. */
static inline int cmp_key(const u128 *d, size_t a, size_t b)
{
    return (int)(d[a] > d[b]) - (int)(d[a] < d[b]);
}

static void downheap(size_t *p, const u128 *d, size_t N, size_t k)
{
    const size_t pki = p[k];
    while (k <= N / 2) {
        size_t j = 2 * k;
        if (j < N && cmp_key(d, p[j], p[j + 1]) < 0) j++;
        if (cmp_key(d, pki, p[j]) >= 0) break;
        p[k] = p[j];
        k = j;
    }
    p[k] = pki;
}

static void heapsort_index(size_t *p, const u128 *d, size_t count)
{
    if (count == 0) return;
    for (size_t i = 0; i < count; i++) p[i] = i;
    size_t N = count - 1;
    size_t k = N / 2;
    k++;
    do {
        k--;
        downheap(p, d, N, k);
    } while (k > 0);
    while (N > 0) {
        size_t tmp = p[0]; p[0] = p[N]; p[N] = tmp;
        N--;
        downheap(p, d, N, 0);
    }
}

/* gather of 4-byte words (f32 or i32 payload alike): new[i] = old[perm[i]] */
static void permute_w32(void *arr, const size_t *perm, size_t n, int w, uint32_t *tmp)
{
    uint32_t *a = (uint32_t *)arr;
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < w; c++) tmp[i * w + c] = a[perm[i] * w + c];
    memcpy(a, tmp, n * w * sizeof(uint32_t));
}

/* peano.c:46-126.  The reference permutes the AoS in place by cycle following;
 * the result is new[i] = old[Idx[i]], which is what the gathers below produce. */
void orc_sort_by_peano_key(orc_state *s, uint64_t *key_hi, uint64_t *key_lo, int64_t *perm_out)
{
    const size_t n = s->n;
    const double box = s->box;

    #pragma omp parallel for num_threads(s->nthreads)
    for (size_t i = 0; i < n; i++) {
        double px = s->pos[3 * i + 0] / box;   /* peano.c:66-68: f32 widened, f64 divide */
        double py = s->pos[3 * i + 1] / box;
        double pz = s->pos[3 * i + 2] / box;
        s->key[i] = peano_key128(px, py, pz);
    }

    size_t *perm = malloc(n * sizeof(size_t));
    heapsort_index(perm, s->key, n);

    uint32_t *tmp = malloc(3 * n * sizeof(uint32_t));
    permute_w32(s->pos, perm, n, 3, tmp);
    permute_w32(s->hsml, perm, n, 1, tmp);
    permute_w32(s->rho, perm, n, 1, tmp);
    permute_w32(s->vhf, perm, n, 1, tmp);
    permute_w32(s->rhom, perm, n, 1, tmp);
    permute_w32(s->apot, perm, n, 3, tmp);
    permute_w32(s->bfld, perm, n, 3, tmp);
    permute_w32(s->id, perm, n, 1, tmp);
    permute_w32(s->tparent, perm, n, 1, tmp);
    free(tmp);

    u128 *ktmp = malloc(n * sizeof(u128));
    for (size_t i = 0; i < n; i++) ktmp[i] = s->key[perm[i]];
    memcpy(s->key, ktmp, n * sizeof(u128));
    free(ktmp);

    for (size_t i = 0; i < n; i++) {
        if (key_hi) key_hi[i] = (uint64_t)(s->key[i] >> 64);
        if (key_lo) key_lo[i] = (uint64_t)s->key[i];
        if (perm_out) perm_out[i] = (int64_t)perm[i];
    }
    free(perm);
}

/* ---------------------------------------------------------------- tree build */

static inline int node_level(const struct orc_node *t) { return t->bitfield & 0x3FUL; }          /* tree.c:333-336 */
static inline int node_triplet(const struct orc_node *t) { return (t->bitfield & (7UL << 6)) >> 6; } /* tree.c:326-331 */

/* tree.c:282-317 */
static int new_node(orc_state *s, int ipart, int parent, u128 key, int lvl)
{
    const int node = s->nnodes++;
    if (s->nnodes >= s->maxnodes) return -1;            /* tree.c:287-292 ("Too many nodes") */
    struct orc_node *t = &s->tree[node];
    const struct orc_node *p = &s->tree[parent];
    t->dnext = -ipart - 1;
    t->bitfield = lvl | ((int)(key & 0x7) << 6);
    float size = s->box / (1 << lvl);
    t->size = size;
    for (int c = 0; c < 3; c++) {
        int sign = -1 + 2 * (s->pos[3 * ipart + c] > p->pos[c]);
        t->pos[c] = p->pos[c] + sign * size * 0.5;      /* float*0.5 -> double, sum in double, round */
    }
    s->tparent[ipart] = parent;
    t->npart++;
    return node;
}

/* ORC_DEV_EXACT_BALL (attribution experiment, NOT reference behaviour): the ball query as the reference's own brute-force
 * twin Find_ngb_simple (wvt_relax.c:296-340) answers it -- every particle passing the f32 predicate of tree.c:67-89, in
 * ascending index, the list cut at NGBMAX -- found through a uniform grid instead of an O(N) scan.  The reference's
 * tree search is NOT always that set: Build_Tree now and then creates a node under a parent of the wrong level
 * (tree.c:201-226 collapses leaves, the next particle's walk then steps into stale or unrelated nodes), the node's
 * centre (tree.c:297-306) is then cells away from its particles and ball queries that should reach them pass it by. */
static void build_grid(orc_state *s)
{
    const int n = s->n;
    int g = (int)cbrt(n / 4.0);
    if (g < 1) g = 1;
    if (g > 128) g = 128;
    s->gdim = g;
    free(s->gstart); free(s->gitem);
    s->gstart = calloc((size_t)g * g * g + 1, sizeof(int32_t));
    s->gitem = malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    int32_t *cell = malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    for (int i = 0; i < n; i++) {
        int c[3];
        for (int d = 0; d < 3; d++) {
            c[d] = (int)((double)s->pos[3 * i + d] / s->box * g);
            if (c[d] >= g) c[d] = g - 1;
            if (c[d] < 0) c[d] = 0;
        }
        cell[i] = (c[0] * g + c[1]) * g + c[2];
        s->gstart[cell[i] + 1]++;
    }
    for (int q = 0; q < g * g * g; q++) s->gstart[q + 1] += s->gstart[q];
    int32_t *fill = calloc((size_t)g * g * g, sizeof(int32_t));
    for (int i = 0; i < n; i++) s->gitem[s->gstart[cell[i]] + fill[cell[i]]++] = i;   /* ascending inside a cell */
    free(fill); free(cell);
}

static int cmp_i32(const void *a, const void *b)
{
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

static int find_ngb_grid(const orc_state *s, int ipart, float hsml, int32_t *ngblist)
{
    const float boxsize = s->box, boxhalf = s->box * 0.5;
    const float xi = s->pos[3 * ipart], yi = s->pos[3 * ipart + 1], zi = s->pos[3 * ipart + 2];
    const int g = s->gdim;
    const double cs = s->box / g;
    const double hp = (double)hsml * (1 + 1e-5) + s->box * 1e-5;
    int lo[3], cnt[3];
    const float xs[3] = {xi, yi, zi};
    for (int d = 0; d < 3; d++) {
        lo[d] = (int)floor((xs[d] - hp) / cs);
        int hi = (int)floor((xs[d] + hp) / cs);
        cnt[d] = hi - lo[d] + 1;
        if (cnt[d] >= g) { lo[d] = 0; cnt[d] = g; }
    }
    int cap = 4096, m = 0;
    int32_t *tmp = malloc((size_t)cap * sizeof(int32_t));
    for (int a = 0; a < cnt[0]; a++)
        for (int b = 0; b < cnt[1]; b++)
            for (int c = 0; c < cnt[2]; c++) {
                const int cx = ((lo[0] + a) % g + g) % g, cy = ((lo[1] + b) % g + g) % g, cz = ((lo[2] + c) % g + g) % g;
                const int q = (cx * g + cy) * g + cz;
                for (int t = s->gstart[q]; t < s->gstart[q + 1]; t++) {
                    const int j = s->gitem[t];
                    float dx = fabs(xi - s->pos[3 * j]), dy = fabs(yi - s->pos[3 * j + 1]), dz = fabs(zi - s->pos[3 * j + 2]);
                    if (dx > boxhalf) dx -= boxsize;
                    if (dy > boxhalf) dy -= boxsize;
                    if (dz > boxhalf) dz -= boxsize;
                    if (dx * dx + dy * dy + dz * dz < hsml * hsml) {
                        if (m == cap) { cap *= 2; tmp = realloc(tmp, (size_t)cap * sizeof(int32_t)); }
                        tmp[m++] = j;
                    }
                }
            }
    qsort(tmp, m, sizeof(int32_t), cmp_i32);
    if (m > ORC_NGBMAX) m = ORC_NGBMAX;
    memcpy(ngblist, tmp, (size_t)m * sizeof(int32_t));
    free(tmp);
    return m;
}

/* tree.c:124-271 */
int orc_build_tree(orc_state *s)
{
    const int n = s->n;
    const double boxsize = s->box, boxhalf = s->box / 2;
    struct orc_node *T = s->tree;

    s->maxnodes = (int)(n * 0.7);
    if (s->maxnodes < 64) s->maxnodes = 64;
    memset(T, 0, (size_t)s->maxnodes * sizeof(*T));
    s->nnodes = 0;

    if (new_node(s, 0, 0, 0, 0) < 0) return -1;
    T[0].pos[0] = T[0].pos[1] = T[0].pos[2] = boxhalf;

    int last_parent = 0;

    /* tree.c:137-141: particle 0 is keyed from FLOAT coordinates */
    float px0 = s->pos[0] / boxsize, py0 = s->pos[1] / boxsize, pz0 = s->pos[2] / boxsize;
    u128 last_key = reversed_peano_key128(px0, py0, pz0);
    last_key >>= 3;

    for (int ipart = 1; ipart < n; ipart++) {
        double px = s->pos[3 * ipart + 0] / boxsize;
        double py = s->pos[3 * ipart + 1] / boxsize;
        double pz = s->pos[3 * ipart + 2] / boxsize;
        u128 key = reversed_peano_key128(px, py, pz);

        int node = 0, lvl = 0, parent = 0;
        bool new_branch = true;

        while (lvl < ORC_NTRIPLETS) {
            if ((int)(key & 0x7) == node_triplet(&T[node])) {           /* open node */
                if (T[node].npart == 1) {                               /* refine */
                    T[node].dnext = 0;
                    if (new_node(s, ipart - 1, node, last_key, lvl + 1) < 0) return -1;
                    last_key >>= 3;
                }
                T[node].npart++;
                new_branch &= (node != last_parent);
                parent = node;
                node++;
                lvl++;
                key >>= 3;
            } else {                                                    /* skip node */
                if (T[node].dnext == 0 || node == s->nnodes - 1) break;
                node += fmax(1, T[node].dnext);
            }
        }

        if (lvl > ORC_NTRIPLETS - 1) {                                  /* tree.c:194-199 */
            s->tparent[ipart] = parent;
            continue;
        }

        if (new_branch) {                                               /* tree.c:201-226 */
            int c = 0;
            if (T[node].npart <= 8) c = node;
            else if (T[last_parent].npart <= 8) c = last_parent;
            if (c != 0) {
                T[c].dnext = -ipart + T[c].npart - 1;
                int nzero = s->nnodes - c - 1;
                s->nnodes = c + 1;
                memset(&T[s->nnodes], 0, (size_t)nzero * sizeof(*T));
                int first = -(T[c].dnext + 1);
                int last = first + T[c].npart;
                for (int j = first; j < last; j++) s->tparent[j] = c;
            }
        }

        if (T[node].dnext == 0) T[node].dnext = s->nnodes - node;       /* tree.c:228-229 */

        if (new_node(s, ipart, parent, key, lvl) < 0) return -1;        /* sibling */
        last_key = key >> 3;
        last_parent = parent;
    }

    /* tree.c:238-266 */
    T[0].dnext = 0;
    int stack[ORC_NTRIPLETS + 1] = { 0 };
    int lowest = 0;
    for (int i = 1; i < s->nnodes; i++) {
        int lvl = node_level(&T[i]);
        while (lvl <= lowest) {
            int node = stack[lowest];
            if (node > 0) T[node].dnext = i - node;
            stack[lowest] = 0;
            lowest--;
        }
        if (T[i].dnext == 0) {
            stack[lvl] = i;
            lowest = lvl;
        }
    }
    if (orc_dev & ORC_DEV_EXACT_BALL) build_grid(s);
    return s->nnodes;
}

int orc_tree_nodes(const orc_state *s, uint32_t *bitfield, int32_t *dnext, float *pos3,
                   int32_t *npart, float *size, int32_t *tree_parent)
{
    for (int i = 0; i < s->nnodes; i++) {
        if (bitfield) bitfield[i] = s->tree[i].bitfield;
        if (dnext) dnext[i] = s->tree[i].dnext;
        if (pos3) { pos3[3*i] = s->tree[i].pos[0]; pos3[3*i+1] = s->tree[i].pos[1]; pos3[3*i+2] = s->tree[i].pos[2]; }
        if (npart) npart[i] = s->tree[i].npart;
        if (size) size[i] = s->tree[i].size;
    }
    if (tree_parent) memcpy(tree_parent, s->tparent, (size_t)s->n * sizeof(int32_t));
    return s->nnodes;
}

/* ---------------------------------------------------------------- neighbour search */

/* tree.c:25-111 -- all arithmetic in f32 as in the reference */
int orc_find_ngb_tree(const orc_state *s, int ipart, float hsml, int32_t *ngblist)
{
    if ((orc_dev & ORC_DEV_EXACT_BALL) && s->gstart) return find_ngb_grid(s, ipart, hsml, ngblist);
    const float boxsize = s->box;
    const float boxhalf = s->box * 0.5;
    const float xi = s->pos[3 * ipart], yi = s->pos[3 * ipart + 1], zi = s->pos[3 * ipart + 2];
    const struct orc_node *T = s->tree;
    const int nnodes = s->nnodes;
    int node = 1, cnt = 0;

    if (nnodes < 2) {   /* degenerate single-node tree: not reachable in the reference for n>1 */
        return 0;
    }

    for (;;) {
        float dx = fabs(xi - T[node].pos[0]);
        float dy = fabs(yi - T[node].pos[1]);
        float dz = fabs(zi - T[node].pos[2]);
        if (dx > boxhalf) dx -= boxsize;
        if (dy > boxhalf) dy -= boxsize;
        if (dz > boxhalf) dz -= boxsize;

        float dl = 0.5 * ORC_SQRT3 * T[node].size + hsml;

        if (dx * dx + dy * dy + dz * dz < dl * dl) {
            if (T[node].dnext < 0) {
                int first = -(T[node].dnext + 1);
                int last = first + T[node].npart;
                for (int j = first; j < last; j++) {
                    float ex = fabs(xi - s->pos[3 * j]);
                    float ey = fabs(yi - s->pos[3 * j + 1]);
                    float ez = fabs(zi - s->pos[3 * j + 2]);
                    if (ex > boxhalf) ex -= boxsize;
                    if (ey > boxhalf) ey -= boxsize;
                    if (ez > boxhalf) ez -= boxsize;
                    if (ex * ex + ey * ey + ez * ez < hsml * hsml) ngblist[cnt++] = j;
                    if (cnt == ORC_NGBMAX) return cnt;
                }
            }
            node++;
            if (node >= nnodes) break;
            continue;
        }
        node += (1 > T[node].dnext) ? 1 : T[node].dnext;
        if (node >= nnodes) break;
    }
    return cnt;
}

/* wvt_relax.c:296-340 (brute force; used to cross-check the tree walk) */
int orc_find_ngb_simple(const orc_state *s, int ipart, float hsml, int32_t *ngblist)
{
    const float boxhalf = s->box / 2;
    const float boxsize = s->box;
    int cnt = 0;
    for (int j = 0; j < s->n; j++) {
        float dx = s->pos[3 * ipart] - s->pos[3 * j];
        float dy = s->pos[3 * ipart + 1] - s->pos[3 * j + 1];
        float dz = s->pos[3 * ipart + 2] - s->pos[3 * j + 2];
        if (dx > boxhalf) dx -= boxsize;
        if (dy > boxhalf) dy -= boxsize;
        if (dz > boxhalf) dz -= boxsize;
        if (dx < -boxhalf) dx += boxsize;
        if (dy < -boxhalf) dy += boxsize;
        if (dz < -boxhalf) dz += boxsize;
        float r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < hsml * hsml) ngblist[cnt++] = j;
        if (cnt == ORC_NGBMAX) break;
    }
    return cnt;
}

/* tree.c:113-121 */
float orc_guess_hsml(const orc_state *s, int ipart)
{
    const struct orc_node *t = &s->tree[s->tparent[ipart]];
    float numDens = t->npart / (t->size * t->size * t->size);
    float size = pow(ORC_FOURPITHIRD / numDens, 1. / 3.);
    return 2 * size;
}

/* ---------------------------------------------------------------- SPH kernels (sph.c:426-440) */

static inline float wc6(const float r, const float h)
{
    const double u = r / h;
    const double t = 1 - u;
    if (orc_dev & ORC_DEV_KERNEL_ULP)
        return dev_ulp(1365.0 / (64 * ORC_PI) / (h * h * h) * t * t * t * t * t * t * t * t
                       * (1 + 8 * u + 25 * u * u + 32 * u * u * u));
    return 1365.0 / (64 * ORC_PI) / (h * h * h) * t * t * t * t * t * t * t * t
           * (1 + 8 * u + 25 * u * u + 32 * u * u * u);
}

static inline float dwc6(const float r, const float h)
{
    const float u = r / h;
    const double t = 1 - u;
    if (orc_dev & ORC_DEV_KERNEL_ULP)
        return dev_ulp(1365.0 / (64 * ORC_PI) / (h * h * h * h) * -22.0 * t * t * t * t * t * t * t * u
                       * (16 * u * u + 7 * u + 1));
    return 1365.0 / (64 * ORC_PI) / (h * h * h * h) * -22.0 * t * t * t * t * t * t * t * u
           * (16 * u * u + 7 * u + 1);
}

/* sph.c:442-466 (cubic spline, -DSPH_CUBIC_SPLINE) */
static inline float sph_kernel_M4(const float r, const float h)
{
    double wk = 0;
    double u = r / h;
    if (u < 0.5)
        wk = (2.546479089470 + 15.278874536822 * (u - 1) * u * u);
    else
        wk = 5.092958178941 * (1.0 - u) * (1.0 - u) * (1.0 - u);
    return wk / (h * h * h);
}

static inline float sph_kernel_derivative_M4(const float r, const float h)
{
    double dwk = 0;
    double u = r / h;
    if (u < 0.5)
        dwk = u * (45.836623610466 * u - 30.557749073644);
    else
        dwk = (-15.278874536822) * (1.0 - u) * (1.0 - u);
    return dwk / (h * h * h * h);
}

#ifdef ORC_SPH_CUBIC_SPLINE                       /* sph.c:140-146, :374-378 */
#define sph_w(r, h) sph_kernel_M4(r, h)
#define sph_dw(r, h) sph_kernel_derivative_M4(r, h)
#else
#define sph_w(r, h) wc6(r, h)
#define sph_dw(r, h) dwc6(r, h)
#endif

float orc_wc6(float r, float h) { return wc6(r, h); }
float orc_dwc6(float r, float h) { return dwc6(r, h); }
float orc_m4(float r, float h) { return sph_kernel_M4(r, h); }
float orc_dm4(float r, float h) { return sph_kernel_derivative_M4(r, h); }

/* sph.c:80-214 */
static bool find_hsml(const orc_state *s, int ipart, const int32_t *ngblist, int ngbcnt,
                      float *dRhodHsml_out, float *hsml_out, float *rho_out,
                      long *iters, long *pairs)
{
    const double boxhalf = 0.5 * s->box;
    const double boxsize = s->box;
    const double mpart = s->mpart;

    double upper = *hsml_out * ORC_SQRT3;
    double lower = 0;
    double hsml = *hsml_out;
    double rho = 0, dRhodHsml = 0;
    int it = 0;
    bool part_done = 0;

    for (;;) {
        const double pi0 = s->pos[3 * ipart], pi1 = s->pos[3 * ipart + 1], pi2 = s->pos[3 * ipart + 2];
        double wkNgb = 0;
        rho = dRhodHsml = 0;
        it++;
        (*iters)++;
        (*pairs) += ngbcnt;

        double pw[64], pr[64], pd[64];                 /* ORC_DEV_TREE_SUMS only */
        if (orc_dev & ORC_DEV_TREE_SUMS)
            for (int l = 0; l < 64; l++) pw[l] = pr[l] = pd[l] = 0;

        for (int i = 0; i < ngbcnt; i++) {
            int j = ngblist[i];
            double dx = pi0 - s->pos[3 * j];
            double dy = pi1 - s->pos[3 * j + 1];
            double dz = pi2 - s->pos[3 * j + 2];
            if (dx > boxhalf) dx -= boxsize;
            if (dx < -boxhalf) dx += boxsize;
            if (dy > boxhalf) dy -= boxsize;
            if (dy < -boxhalf) dy += boxsize;
            if (dz > boxhalf) dz -= boxsize;
            if (dz < -boxhalf) dz += boxsize;
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 > hsml * hsml) continue;
            double r = sqrt(r2);
            double wk = sph_w(r, hsml);
            double dwk = sph_dw(r, hsml);
            if (orc_dev & ORC_DEV_TREE_SUMS) {         /* the GPU's order: 64 lane partials, then a reduction tree */
                pw[i & 63] += ORC_FOURPITHIRD * wk * (hsml * hsml * hsml);
                pr[i & 63] += mpart * wk;
                pd[i & 63] += -mpart * (3 / hsml * wk + r / hsml * dwk);
                continue;
            }
            wkNgb += ORC_FOURPITHIRD * wk * (hsml * hsml * hsml);
            rho += mpart * wk;
            dRhodHsml += -mpart * (3 / hsml * wk + r / hsml * dwk);
        }
        if (orc_dev & ORC_DEV_TREE_SUMS) {
            for (int w = 32; w >= 1; w >>= 1)
                for (int l = 0; l < w; l++) { pw[l] += pw[l + w]; pr[l] += pr[l + w]; pd[l] += pd[l + w]; }
            wkNgb = pw[0]; rho = pr[0]; dRhodHsml = pd[0];
        }

        if (it > 128) break;

        double ngbDev = fabs(wkNgb - ORC_DESNNGB);
        if (ngbDev < ORC_NNGBDEV) { part_done = true; break; }

        if (fabs(upper - lower) < 1e-4) { hsml *= 1.26; break; }

        if (ngbDev < 0.5 * ORC_DESNNGB) {
            double omega = (1 + dRhodHsml * hsml / (3 * rho));
            double fac = 1 - (wkNgb - ORC_DESNNGB) / (3 * wkNgb * omega);
            fac = fmin(1.24, fac);
            fac = fmax(1 / 1.24, fac);
            hsml *= fac;
        } else {
            if (wkNgb > ORC_DESNNGB) upper = hsml;
            if (wkNgb < ORC_DESNNGB) lower = hsml;
            hsml = dev_pow(0.5 * ((lower * lower * lower) + (upper * upper * upper)), 1.0 / 3.0);
        }
    }

    *hsml_out = (float)hsml;
    *rho_out = (float)rho;

#ifndef ORC_SPH_CUBIC_SPLINE                      /* sph.c:201-212 */
    if (part_done) {
        *dRhodHsml_out = (float)dRhodHsml;
        double bias_corr = -0.0116 * pow(ORC_DESNNGB * 0.01, -2.236) * mpart * wc6(0, hsml);
        *rho_out += bias_corr;
    }
#endif
    return part_done;
}

/* sph.c:13-75 */
int orc_find_sph_quantities(orc_state *s)
{
    orc_sort_by_peano_key(s, NULL, NULL, NULL);
    if (orc_build_tree(s) < 0) return -1;

    const int n = s->n;
    long tq = 0, ti = 0, tp = 0, ttr = 0;
    int bad = 0;
    int chunk = n / s->nthreads / 64;
    if (chunk < 1) chunk = 1;    /* the reference hangs with chunk 0 */

    #pragma omp parallel for schedule(dynamic, chunk) num_threads(s->nthreads) \
            reduction(+:tq,ti,tp,ttr) reduction(|:bad)
    for (int ipart = 0; ipart < n; ipart++) {
        float hsml = s->hsml[ipart];
        if (hsml == 0) hsml = 2 * orc_guess_hsml(s, ipart);
        if (!isfinite(hsml)) { bad = 1; continue; }        /* sph.c:28 Assert */

        float dRhodHsml = 0, rho = 0;
        int32_t ngblist[ORC_NGBMAX];
        long guard = 0;

        for (;;) {
            if (++guard > 100000) { bad = 1; break; }      /* not in the reference: bounded for tests */
            int ngbcnt = orc_find_ngb_tree(s, ipart, hsml, ngblist);
            tq++;
            if (ngbcnt == ORC_NGBMAX) { hsml /= 1.24; ttr++; continue; }
            if (ngbcnt < ORC_DESNNGB) { hsml *= 1.23; continue; }
            bool done = find_hsml(s, ipart, ngblist, ngbcnt, &dRhodHsml, &hsml, &rho, &ti, &tp);
            if (done) break;
        }
        float varHsmlFac = 1.0 / (1 + hsml / (3 * rho) * dRhodHsml);
        s->hsml[ipart] = hsml;
        s->rho[ipart] = rho;
        s->vhf[ipart] = varHsmlFac;
    }
    s->st_queries = (double)tq / n; s->st_iters = (double)ti / n; s->st_pairs = (double)tp / n;
    s->st_trunc_density += ttr;
    return bad ? -2 : 0;
}

void orc_truncations(orc_state *s, long *density, long *sweep, int reset)
{
    if (density) *density = s->st_trunc_density;
    if (sweep) *sweep = s->st_trunc_sweep;
    if (reset) s->st_trunc_density = s->st_trunc_sweep = 0;
}

void orc_last_stats(const orc_state *s, double *q, double *it, double *p)
{
    if (q) *q = s->st_queries;
    if (it) *it = s->st_iters;
    if (p) *p = s->st_pairs;
}

/* ---------------------------------------------------------------- density model */

/* Param.Rho0_Fac / Param.Rc_Fac (globals.h:117-120); `double_beta` stands for -DDOUBLE_BETA_COOL_CORES */
static double Param_Rho0_Fac = 0, Param_Rc_Fac = 0;
static int double_beta = 0;

void orc_set_double_beta(double rho0_fac, double rc_fac)
{
    Param_Rho0_Fac = rho0_fac;
    Param_Rc_Fac = rc_fac;
    double_beta = rho0_fac != 0 && rc_fac != 0;
}

/* setup.c:598-615 */
static inline double gas_density_profile_c(double r, double rho0, double beta, double rc, double rcut, int Is_Cuspy)
{
    double rho = rho0 * dev_pow(1 + (r / rc) * (r / rc), -3.0 / 2.0 * beta)
                 / (1 + ((r / rcut) * (r / rcut) * (r / rcut)) * (r / rcut));
    if (double_beta) {                                   /* #ifdef DOUBLE_BETA_COOL_CORES, setup.c:604-612 */
        double rho0_cc = rho0 * Param_Rho0_Fac;
        double rc_cc = rc / Param_Rc_Fac;
        if (Is_Cuspy)
            rho += rho0_cc / (1 + (r / rc_cc) * (r / rc_cc)) / (1 + ((r / rcut) * (r / rcut) * (r / rcut)) * (r / rcut));
    }
    return rho;
}

/* wvt_relax.c:227-256 */
static inline float density_model(const orc_state *s, int ipart)
{
    const double boxhalf = s->box * 0.5;
    const double x = s->pos[3 * ipart], y = s->pos[3 * ipart + 1], z = s->pos[3 * ipart + 2];
    double rho = 0;
    for (int i = 0; i < s->nhalos; i++) {
        const orc_halo *h = &s->halo[i];
        if (h->mass_gas == 0) continue;
        double dx = x - h->d_com[0] - boxhalf;
        double dy = y - h->d_com[1] - boxhalf;
        double dz = z - h->d_com[2] - boxhalf;
        double r2 = dx * dx + dy * dy + dz * dz;
        double rho_i = gas_density_profile_c(sqrt(r2), h->rho0, h->beta, h->rcore, h->rcut, h->have_cuspy);
        rho = fmax(rho_i, rho);
    }
    return rho;
}

void orc_global_density_model(const orc_state *s, float *out)
{
    #pragma omp parallel for num_threads(s->nthreads)
    for (int i = 0; i < s->n; i++) out[i] = density_model(s, i);
}

/* ---------------------------------------------------------------- WVT relaxation */

/* wvt_relax.c:275-281 (no 1/h^3) */
static inline double wvt_wc6(const float r, const float h)
{
    const double u = r / h;
    const double t = 1 - u;
    return 1365.0 / (64 * ORC_PI) * t * t * t * t * t * t * t * t * (1 + 8 * u + 25 * u * u + 32 * u * u * u);
}

double orc_wvt_wc6(float r, float h) { return wvt_wc6(r, h); }

/* wvt_relax.c:106-214 */
void orc_wvt_step(orc_state *s, double step, float *hsml_out, float *delta_out, int move)
{
    const int n = s->n;
    const double boxsize = s->box, boxinv = 1 / boxsize;
    float *hsml = malloc((size_t)n * sizeof(float));
    float *delta = malloc(3 * (size_t)n * sizeof(float));

    double vSphSum = 0;
    #pragma omp parallel for reduction(+:vSphSum) num_threads(s->nthreads)
    for (int i = 0; i < n; i++) {
        float rho = density_model(s, i);
        s->rhom[i] = rho;
        hsml[i] = dev_pow(ORC_DESNNGB * s->mpart / rho / ORC_FOURPITHIRD, 1. / 3.);
        vSphSum += hsml[i] * hsml[i] * hsml[i];
    }
    float norm_hsml = dev_pow(ORC_DESNNGB / vSphSum / ORC_FOURPITHIRD, 1.0 / 3.0);

    #pragma omp parallel for num_threads(s->nthreads)
    for (int i = 0; i < n; i++) hsml[i] *= norm_hsml;

    int chunk = n / s->nthreads / 256;
    if (chunk < 1) chunk = 1;

    long ttr = 0;
    #pragma omp parallel for schedule(dynamic, chunk) num_threads(s->nthreads) reduction(+:ttr)
    for (int ipart = 0; ipart < n; ipart++) {
        float d0 = 0, d1 = 0, d2 = 0;
        double u0 = 0, u1 = 0, u2 = 0;                     /* ORC_DEV_SWEEP_ROUND_ONCE only */
        int32_t ngblist[ORC_NGBMAX];
        int ngbcnt = orc_find_ngb_tree(s, ipart, hsml[ipart] * boxsize, ngblist);
        if (ngbcnt == ORC_NGBMAX) ttr++;

        for (int i = 0; i < ngbcnt; i++) {
            int j = ngblist[i];
            if (ipart == j) continue;
            float dx = (s->pos[3 * ipart] - s->pos[3 * j]) * boxinv;
            float dy = (s->pos[3 * ipart + 1] - s->pos[3 * j + 1]) * boxinv;
            float dz = (s->pos[3 * ipart + 2] - s->pos[3 * j + 2]) * boxinv;
            dx = dx > 0.5 ? dx - 1 : dx;
            dy = dy > 0.5 ? dy - 1 : dy;
            dz = dz > 0.5 ? dz - 1 : dz;
            dx = dx < -0.5 ? dx + 1 : dx;
            dy = dy < -0.5 ? dy + 1 : dy;
            dz = dz < -0.5 ? dz + 1 : dz;
            float r2 = (dx * dx + dy * dy + dz * dz);
            float h = 0.5 * (hsml[ipart] + hsml[j]);
            if (r2 > h * h) continue;
            float r = sqrt(r2);
            float wk = wvt_wc6(r, h);
            if (orc_dev & ORC_DEV_SWEEP_ROUND_ONCE) {      /* round 2's GPU sweep: f64 sum for unit step, one rounding */
                u0 += hsml[ipart] * (double)wk * dx / r;
                u1 += hsml[ipart] * (double)wk * dy / r;
                u2 += hsml[ipart] * (double)wk * dz / r;
                continue;
            }
            d0 += step * hsml[ipart] * wk * dx / r;
            d1 += step * hsml[ipart] * wk * dy / r;
            d2 += step * hsml[ipart] * wk * dz / r;
        }
        if (orc_dev & ORC_DEV_SWEEP_ROUND_ONCE) { d0 = (float)(step * u0); d1 = (float)(step * u1); d2 = (float)(step * u2); }
        delta[3 * ipart] = d0; delta[3 * ipart + 1] = d1; delta[3 * ipart + 2] = d2;
    }
    s->st_trunc_sweep += ttr;

    if (hsml_out) memcpy(hsml_out, hsml, (size_t)n * sizeof(float));
    if (delta_out) memcpy(delta_out, delta, 3 * (size_t)n * sizeof(float));

    if (move) {
        #pragma omp parallel for num_threads(s->nthreads)
        for (int i = 0; i < n; i++) {
            for (int c = 0; c < 3; c++) {
                float p = s->pos[3 * i + c];
                p += (float)(delta[3 * i + c] * boxsize);
                while (p < 0) p += boxsize;
                while (p > boxsize) p -= boxsize;
                s->pos[3 * i + c] = p;
            }
        }
    }
    free(hsml); free(delta);
}

/* wvt_relax.c:25-225 */
int orc_regularise(orc_state *s, orc_iterlog *log, int max_iter)
{
    const int n = s->n;
    int it = -1, nlog = 0;
#ifdef ORC_SPH_CUBIC_SPLINE                       /* wvt_relax.c:48-56 */
    double step = 0.035;
#else
    double step = 0.0085;
    if (s->mtotal < 1e5) step /= 2;
#endif
    double errLast = DBL_MAX, errDiff = DBL_MAX, errDiffLast = DBL_MAX;
    const int numiter = max_iter >= 0 ? max_iter : ORC_NUMITER;

    for (;;) {
        if (it++ >= numiter) break;
        if (orc_find_sph_quantities(s) < 0) return -1;

        int nIn = 0;
        double errMax = 0, errMean = 0;
        #pragma omp parallel for reduction(+:errMean,nIn) reduction(max:errMax) num_threads(s->nthreads)
        for (int i = 0; i < n; i++) {
            float rho = density_model(s, i);
            float err = fabs(s->rho[i] - rho) / rho;
            errMax = fmax(err, errMax);
            errMean += err;
            nIn++;
        }
        errMean /= nIn;
        errDiff = (errLast - errMean) / errMean;

        if (log && nlog < ORC_MAXLOG) {
            log[nlog].it = it; log[nlog].err_max = errMax; log[nlog].err_mean = errMean;
            log[nlog].err_diff = errDiff; log[nlog].step = step;
        }
        nlog++;

        if (errDiff < ORC_ERRDIFF_LIMIT && it > 25) break;
        if ((errDiff < 0) && (errDiffLast < 0) && (it > 10)) break;
        if (errDiff < 0.01 && (it > 1)) step *= 0.8;
        errLast = errMean;
        errDiffLast = errDiff;

        orc_wvt_step(s, step, NULL, NULL, 1);
    }
    return nlog;
}

/* ---------------------------------------------------------------- curl(A)  sph.c:216-300 */
void orc_bfld_from_rotA(orc_state *s)
{
    const int n = s->n;
    const double mpart = s->mpart, boxhalf = s->box / 2, boxsize = s->box;
    int chunk = n / s->nthreads / 64;
    if (chunk < 1) chunk = 1;

    #pragma omp parallel for schedule(dynamic, chunk) num_threads(s->nthreads)
    for (int ipart = 0; ipart < n; ipart++) {
        int32_t ngblist[ORC_NGBMAX];
        int ngbcnt = orc_find_ngb_tree(s, ipart, s->hsml[ipart], ngblist);
        double varHsmlFac = s->vhf[ipart];
        double hsml = s->hsml[ipart];
        double rho_i = s->rho[ipart];
        double pi0 = s->pos[3 * ipart], pi1 = s->pos[3 * ipart + 1], pi2 = s->pos[3 * ipart + 2];
        double a0 = s->apot[3 * ipart], a1 = s->apot[3 * ipart + 1], a2 = s->apot[3 * ipart + 2];
        double b0 = 0, b1 = 0, b2 = 0;

        for (int i = 0; i < ngbcnt; i++) {
            int j = ngblist[i];
            if (j == ipart) continue;
            double dx = pi0 - s->pos[3 * j];
            double dy = pi1 - s->pos[3 * j + 1];
            double dz = pi2 - s->pos[3 * j + 2];
            if (dx > boxhalf) dx -= boxsize;
            if (dx < -boxhalf) dx += boxsize;
            if (dy > boxhalf) dy -= boxsize;
            if (dy < -boxhalf) dy += boxsize;
            if (dz > boxhalf) dz -= boxsize;
            if (dz < -boxhalf) dz += boxsize;
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 > hsml * hsml) continue;
            double r = sqrt(r2);
            double dwk = sph_dw(r, hsml);
            double weight = -mpart / rho_i * dwk / r * varHsmlFac;
            double dAx = a0 - s->apot[3 * j];
            double dAy = a1 - s->apot[3 * j + 1];
            double dAz = a2 - s->apot[3 * j + 2];
            b0 += weight * (dz * dAy - dy * dAz);
            b1 += weight * (dx * dAz - dz * dAx);
            b2 += weight * (dy * dAx - dx * dAy);
        }
        s->bfld[3 * ipart] = (float)b0;
        s->bfld[3 * ipart + 1] = (float)b1;
        s->bfld[3 * ipart + 2] = (float)b2;
    }
}

/* ---------------------------------------------------------------- behind the path: halo reassignment */

/* gsl_heapsort_index with its generic signature (sort.c:192 passes compare_int of positions.c:390-396
 * for the halo ids). */
typedef int (*orc_cmp_fn)(const void *, const void *);

static void downheap_generic(size_t *p, const char *data, size_t size, size_t N, size_t k, orc_cmp_fn cmp)
{
    const size_t pki = p[k];
    while (k <= N / 2) {
        size_t j = 2 * k;
        if (j < N && cmp(data + size * p[j], data + size * p[j + 1]) < 0) j++;
        if (cmp(data + size * pki, data + size * p[j]) >= 0) break;
        p[k] = p[j];
        k = j;
    }
    p[k] = pki;
}

static void heapsort_index_generic(size_t *p, const void *data, size_t count, size_t size, orc_cmp_fn cmp)
{
    if (count == 0) return;
    for (size_t i = 0; i < count; i++) p[i] = i;
    size_t N = count - 1;
    size_t k = N / 2;
    k++;
    do {
        k--;
        downheap_generic(p, (const char *)data, size, N, k, cmp);
    } while (k > 0);
    while (N > 0) {
        size_t tmp = p[0]; p[0] = p[N]; p[N] = tmp;
        N--;
        downheap_generic(p, (const char *)data, size, N, 0, cmp);
    }
}

static int compare_int(const void *a, const void *b)      /* positions.c:390-396 */
{
    const int *x = (const int *)a, *y = (const int *)b;
    return (*x > *y) - (*x < *y);
}

/* positions.c:264-331 (gas), Halo_containing positions.c:333-388 (SPH branch), sort_particles
 * positions.c:399-445.  halo_id[n] as assigned, perm[n] with new[i] = old[perm[i]], npart[nhalos]. */
int orc_reassign_to_halos(int n, const float *pos, double boxsize, int nhalos, const orc_halo *halos,
                          const double *r_sample, int32_t *halo_id, int64_t *perm, int64_t *npart)
{
    const float boxhalf = 0.5 * boxsize;
    for (int j = 0; j < nhalos; j++) npart[j] = 0;
    for (int ip = 0; ip < n; ip++) {
        float x = pos[3 * ip] - boxhalf, y = pos[3 * ip + 1] - boxhalf, z = pos[3 * ip + 2] - boxhalf;
        if (x > boxsize || y > boxsize || z > boxsize) return -1;
        int i = 0;
        double rho_max = 0;
        for (int j = 0; j < nhalos; j++) {
            float r = sqrt((x - halos[j].d_com[0]) * (x - halos[j].d_com[0])
                           + (y - halos[j].d_com[1]) * (y - halos[j].d_com[1])
                           + (z - halos[j].d_com[2]) * (z - halos[j].d_com[2]));
            double rho_gas = gas_density_profile_c(r, halos[j].rho0, halos[j].beta, halos[j].rcore, halos[j].rcut, halos[j].have_cuspy);
            if (rho_gas > rho_max && r < r_sample[j]) { i = j; rho_max = rho_gas; }
        }
        halo_id[ip] = i;
        npart[i]++;
    }
    size_t *idx = malloc(sizeof(size_t) * (n > 0 ? n : 1));
    heapsort_index_generic(idx, halo_id, (size_t)n, sizeof(*halo_id), compare_int);
    for (int i = 0; i < n; i++) perm[i] = (int64_t)idx[i];
    free(idx);
    return 0;
}

/* ---- the wrapper around the curl: magnetic_field.c:33-131 (test infrastructure, like the rest of this file) ---- */

#define ORC_P2(a) ((a) * (a))                               /* macro.h p2() */

/* Halo_containing, positions.c:333-388, both branches; r_sample_gas = R_Sample[0], r_sample_dm = R_Sample[1] */
static int orc_halo_containing(int type, float x, float y, float z, double boxsize, int nhalos, const orc_halo *Halo,
                               const double *r_sample_gas, const double *r_sample_dm, int sub_first)
{
    if ((x > boxsize || y > boxsize || z > boxsize)) return -1;
    int i = 0;
    if (type > 0) {                                         /* DM, positions.c:343-362 */
        if (nhalos > 1) {
            float r = sqrt(ORC_P2(x - Halo[1].d_com[0]) + ORC_P2(y - Halo[1].d_com[1]) + ORC_P2(z - Halo[1].d_com[2]));
            if ((r < r_sample_dm[1]) && (x > 0)) i = 1;
        }
        for (int j = sub_first; j < nhalos; j++) {
            float r = sqrt(ORC_P2(x - Halo[j].d_com[0]) + ORC_P2(y - Halo[j].d_com[1]) + ORC_P2(z - Halo[j].d_com[2]));
            if (r < r_sample_dm[j]) {
                i = j;
                break;
            }
        }
    } else {                                                /* SPH, positions.c:364-385 */
        double rho_max = 0;
        for (int j = 0; j < nhalos; j++) {
            float r = sqrt(ORC_P2(x - Halo[j].d_com[0]) + ORC_P2(y - Halo[j].d_com[1]) + ORC_P2(z - Halo[j].d_com[2]));
            double rho_gas = gas_density_profile_c(r, Halo[j].rho0, Halo[j].beta, Halo[j].rcore, Halo[j].rcut, Halo[j].have_cuspy);
            if ((rho_gas > rho_max) && (r < r_sample_gas[j])) {
                i = j;
                rho_max = rho_gas;
            }
        }
    }
    return i;
}

/* set_magnetic_vector_potential, magnetic_field.c:33-69 */
void orc_set_vector_potential(int n, const float *pos, double boxsize, int nhalos, const orc_halo *Halo, double eta,
                              float *apot)
{
    const float boxhalf = 0.5 * boxsize;
    for (int ipart = 0; ipart < n; ipart++) {
        double A_max = 0;
        for (int i = 0; i < nhalos; i++) {
            if (Halo[i].mass_gas == 0) continue;
            float dx = pos[3 * ipart] - Halo[i].d_com[0] - boxhalf, dy = pos[3 * ipart + 1] - Halo[i].d_com[1] - boxhalf,
                  dz = pos[3 * ipart + 2] - Halo[i].d_com[2] - boxhalf;
            double r2 = dx * dx + dy * dy + dz * dz;
            double rho_i = gas_density_profile_c(sqrt(r2), Halo[i].rho0, Halo[i].beta, Halo[i].rcore, Halo[i].rcut, Halo[i].have_cuspy);
            double A = pow(rho_i / Halo[i].rho0, eta);
            if (A > A_max) A_max = A;
        }
        apot[3 * ipart] = (float)A_max;
        apot[3 * ipart + 1] = (float)A_max;
        apot[3 * ipart + 2] = (float)A_max;
    }
}

/* normalise_magnetic_field, magnetic_field.c:71-131, serial (the reference's max_B2 update races, SURVEY.md section 5;
 * serial = the race-free value).  Halo_containing is called with the particle INDEX as `type`, as the reference does
 * (magnetic_field.c:109). */
void orc_normalise_magnetic_field(int n, const float *pos, float *bfld, double boxsize, int nhalos, const orc_halo *Halo,
                                  const double *r_sample_gas, const double *r_sample_dm, int sub_first, double bfld_norm,
                                  double *norm_out, int *cnt_out)
{
    const float boxhalf = 0.5 * boxsize;
    double max_B2 = 0;
    for (int ipart = 0; ipart < n; ipart++) {
        double bfld2 = ORC_P2(bfld[3 * ipart]) + ORC_P2(bfld[3 * ipart + 1]) + ORC_P2(bfld[3 * ipart + 2]);
        max_B2 = fmax(max_B2, bfld2);
    }
    double max_bfld = sqrt(max_B2);
    double norm = bfld_norm / max_bfld / sqrt(3);
    int cnt = 0;
    for (int ipart = 0; ipart < n; ipart++) {
        bfld[3 * ipart] *= norm;
        bfld[3 * ipart + 1] *= norm;
        bfld[3 * ipart + 2] *= norm;
        double B2 = ORC_P2(bfld[3 * ipart]) + ORC_P2(bfld[3 * ipart + 1]) + ORC_P2(bfld[3 * ipart + 2]);
        float x = pos[3 * ipart] - boxhalf, y = pos[3 * ipart + 1] - boxhalf, z = pos[3 * ipart + 2] - boxhalf;
        int i = orc_halo_containing(ipart, x, y, z, boxsize, nhalos, Halo, r_sample_gas, r_sample_dm, sub_first);
        double bmax = 18e-6;                                /* BMAX, magnetic_field.c:4 */
        if (i > 1) bmax = 2e-6;
        if ((B2 > ORC_P2(bmax))) {
            double B = sqrt(B2);
            bfld[3 * ipart] *= bmax / B;
            bfld[3 * ipart + 1] *= bmax / B;
            bfld[3 * ipart + 2] *= bmax / B;
            cnt++;
        }
    }
    *norm_out = norm;
    *cnt_out = cnt;
}
