/* tc_gadget.c -- Gadget-2 "format 2" writer with the block set and order of the reference
 * (src/io.h:31-41 enum order; record structure src/io.c:70-80,117-128). */
#include <stdlib.h>
#include <string.h>
#include "tc_host.h"

static int put(FILE *fp, const void *data, size_t bytes)
{
    if (bytes == 0) return 0;
    return fwrite(data, 1, bytes, fp) == bytes ? 0 : 6;       /* src/io.c:274-281: exit(6) */
}

/* [8]["LABL"][payload+8][8] then [payload][...][payload] */
static int put_block(FILE *fp, const char label[4], const void *payload, size_t bytes)
{
    int32_t eight = 8, next = (int32_t)(bytes + 2 * sizeof(int32_t)), len = (int32_t)bytes;
    int rc = 0;
    rc |= put(fp, &eight, 4);
    rc |= put(fp, label, 4);
    rc |= put(fp, &next, 4);
    rc |= put(fp, &eight, 4);
    rc |= put(fp, &len, 4);
    rc |= put(fp, payload, bytes);
    rc |= put(fp, &len, 4);
    return rc;
}

int tc_write_snapshot(const char *filename, const tc_snapshot *s)
{
    FILE *fp = fopen(filename, "w");
    if (!fp) return 5;

    tc_gadget_header h;
    memset(&h, 0, sizeof(h));
    long long ntot = 0;
    for (int i = 0; i < 6; i++) {                            /* src/io.c:48-67 */
        h.npart[i] = (int32_t)s->npart[i];
        h.mass[i] = s->mpart[i];
        h.npartTotal[i] = (uint32_t)h.npart[i];
        ntot += s->npart[i];
    }
    h.num_files = 1;
    h.BoxSize = s->boxsize;
    h.Omega0 = 1;
    h.OmegaLambda = 0.7;
    h.HubbleParam = s->hubble_param;
    const size_t ngas = (size_t)s->npart[0], n = (size_t)ntot;

    int rc = put_block(fp, "HEAD", &h, sizeof(h));
    rc |= put_block(fp, "POS ", s->pos, 3 * n * sizeof(float));
    rc |= put_block(fp, "VEL ", s->vel, 3 * n * sizeof(float));
    rc |= put_block(fp, "ID  ", s->id, n * sizeof(uint32_t));
    rc |= put_block(fp, "U   ", s->u, ngas * sizeof(float));
    rc |= put_block(fp, "RHO ", s->rho, ngas * sizeof(float));
    rc |= put_block(fp, "HSML", s->hsml, ngas * sizeof(float));
    rc |= put_block(fp, "BFLD", s->bfld, 3 * ngas * sizeof(float));
    rc |= put_block(fp, "RHOM", s->rho_model, ngas * sizeof(float));
    if (fclose(fp)) rc |= 6;
    return rc;
}
