/* tc_parfile.c -- reader for the reference's parameter file (behaviour of src/io.c:298-507). */
#include <stdlib.h>
#include <string.h>
#include "tc_host.h"

enum { T_REAL, T_STRING, T_INT };

int tc_read_param_file(const char *filename, tc_parfile *p, char *err, size_t errlen)
{
    struct { const char *tag; int type; void *addr; int done; } tab[] = {
        {"Output_file", T_STRING, p->output_file, 0},
        {"Ntotal", T_INT, &p->ntotal, 0},
        {"Mtotal", T_REAL, &p->mtot200, 0},
        {"Redshift", T_REAL, &p->redshift, 0},
        {"Mass_Ratio", T_REAL, &p->mass_ratio, 0},
        {"ImpactParam", T_REAL, &p->impact_param, 0},
        {"ZeroEOrbitFrac", T_REAL, &p->zero_e_orbit_frac, 0},
        {"Cuspy", T_INT, &p->cuspy, 0},
        {"Bfld_Norm", T_REAL, &p->bfld_norm, 0},
        {"Bfld_Eta", T_REAL, &p->bfld_eta, 0},
        {"bf", T_REAL, &p->baryon_fraction, 0},
        {"UnitLength_in_cm", T_REAL, &p->unit_length, 0},
        {"UnitMass_in_g", T_REAL, &p->unit_mass, 0},
        {"UnitVelocity_in_cm_per_s", T_REAL, &p->unit_vel, 0},
        {"Rho0_Fac", T_REAL, &p->rho0_fac, 0},          /* -DDOUBLE_BETA_COOL_CORES only (src/io.c:435-443) */
        {"Rc_Fac", T_REAL, &p->rc_fac, 0},
    };
    const int nt_all = (int)(sizeof(tab) / sizeof(tab[0]));
    memset(p, 0, sizeof(*p));
    const char *db = getenv("TC_DOUBLE_BETA");
    p->double_beta = db && atoi(db) != 0;
    const int nt = p->double_beta ? nt_all : nt_all - 2;   /* the default build does not know the two tags */

    FILE *fd = fopen(filename, "r");
    if (!fd) {
        snprintf(err, errlen, "Parameter file %s not found.", filename);
        return 1;
    }
    char line[TC_CHARBUF], key[TC_CHARBUF], val[TC_CHARBUF], rest[2 * TC_CHARBUF];
    while (fgets(line, TC_CHARBUF, fd)) {
        if (sscanf(line, "%s%s%s", key, val, rest) < 2) continue;      /* needs tag and value */
        if (key[0] == '%') continue;                                    /* comment */
        for (int i = 0; i < nt; i++) {
            if (strcmp(key, tab[i].tag) || tab[i].done) continue;       /* first occurrence wins */
            tab[i].done = 1;
            if (tab[i].type == T_REAL) *(double *)tab[i].addr = atof(val);
            else if (tab[i].type == T_STRING) strcpy((char *)tab[i].addr, val);
            else if (tab[i].addr == (void *)&p->ntotal) p->ntotal = atoi(val);   /* 32-bit, as the reference */
            else *(int *)tab[i].addr = atoi(val);
            break;
        }
    }
    fclose(fd);
    for (int i = 0; i < nt; i++)
        if (!tab[i].done) {
            snprintf(err, errlen, "Value for tag '%s' missing in parameter file '%s'.", tab[i].tag, filename);
            return 2;
        }
    return 0;
}
