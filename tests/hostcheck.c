/* Host build of toycluster_amd/csrc/tc_math.h (the very lines the HIP kernels compile) so that the
 * CPU test-suite can check them against the oracle.  Test infrastructure only: never loaded by the product. */
#include "../toycluster_amd/csrc/tc_math.h"

void hc_peano_key(float px, float py, float pz, double box, uint64_t *hi, uint64_t *lo)
{
    tc_peano_key(px, py, pz, box, hi, lo);
}
int hc_common_levels(uint64_t ahi, uint64_t alo, uint64_t bhi, uint64_t blo)
{
    return tc_common_levels(ahi, alo, bhi, blo);
}
float hc_fdiv(float a, float b)
{
    tc_fdiv d = tc_fdiv_setup(b);
    return tc_fdiv_apply(d, a);
}
float hc_wc6(float r, float h) { return tc_wc6(r, h, TC_WC6_NORM / (double)(h * h * h)); }
float hc_dwc6(float r, float h) { return tc_dwc6(r, h, TC_WC6_NORM / (double)(h * h * h * h) * -22.0); }
double hc_wvt_wc6(float r, float h) { return tc_wvt_wc6(r, h); }
float hc_ngb_r2(float xi, float yi, float zi, float xj, float yj, float zj, double box)
{
    return tc_ngb_r2(xi, yi, zi, xj, yj, zj, (float)(box * 0.5), (float)box);
}
double hc_pair_r(float xi, float yi, float zi, float xj, float yj, float zj, double box)
{
    return tc_pair_r(xi, yi, zi, xj, yj, zj, 0.5 * box, box);
}
float hc_density_model(float px, float py, float pz, double boxhalf, const tc_halo_dev *halo, int nhalos)
{
    return tc_density_model(px, py, pz, boxhalf, halo, nhalos);
}

/* the table-driven key the kernels use (csrc/tc_math.h::tc_peano_key_lut, table from tools/gen_hilbert_lut.py) */
#include "../toycluster_amd/csrc/tc_hilbert_lut.h"
void hc_peano_key_lut(float x, float y, float z, double box, uint64_t *hi, uint64_t *lo)
{
    tc_peano_key_lut(x, y, z, box, TC_HILBERT_LUT, hi, lo);
}
