// Profiling helper (not part of the product): issue rate of a few VALU instruction kinds on gfx950,
// relative to v_fma_f32.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 4096

template <int KIND>
__global__ void k(float *out, float a, double b)
{
    float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    double d0 = x0 + b, d1 = x1 + b, d2 = x2 + b, d3 = x3 + b;
    for (int i = 0; i < REP; i++) {
        if (KIND == 0) {   // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        } else if (KIND == 1) {   // v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b));
        } else if (KIND == 2) {   // v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b));
        } else if (KIND == 3) {   // v_cvt_f32_f64
            asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        } else if (KIND == 4) {   // v_cvt_f64_f32
            asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        } else if (KIND == 5) {   // v_add_f64
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b));
        } else if (KIND == 6) {   // v_rcp_f64
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (KIND == 7) {   // v_add_u32 with DPP
            asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (KIND == 8) {   // v_readlane_b32 (to SGPR)
            int s;
            asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 3\n v_readlane_b32 %0, %3, 3\n v_readlane_b32 %0, %4, 3\n"
                         : "=s"(s) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        } else if (KIND == 9) {   // v_min_f64
            asm volatile("v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b));
        } else if (KIND == 10) {  // v_sqrt_f32
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (KIND == 11) {  // v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(d0 + d1 + d2 + d3);
}

template <int KIND>
static double run(const char *name, float *d_out, double base)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, threads = 256;      // 8 blocks x 4 waves per CU = 8 waves per SIMD
    k<KIND><<<blocks, threads>>>(d_out, 1.0f, 1.0);
    hipEventRecord(a);
    k<KIND><<<blocks, threads>>>(d_out, 1.0f, 1.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double inst = (double)blocks * (threads / 64) * REP * 4;      // wave-level instructions
    double per_simd_cycle = inst / (ms * 1e-3) / (1024 * 2.4e9);
    printf("%-16s %8.3f ms  %.3f wave-instr/cycle/SIMD  -> %.2f cycles each%s\n", name, ms, per_simd_cycle,
           1 / per_simd_cycle, base > 0 ? "" : "");
    return ms;
}

int main()
{
    float *d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", d_out, 0);
    run<1>("v_fma_f64", d_out, 0);
    run<2>("v_mul_f64", d_out, 0);
    run<5>("v_add_f64", d_out, 0);
    run<9>("v_min_f64", d_out, 0);
    run<3>("v_cvt_f32_f64", d_out, 0);
    run<4>("v_cvt_f64_f32", d_out, 0);
    run<6>("v_rcp_f64", d_out, 0);
    run<10>("v_sqrt_f32", d_out, 0);
    run<7>("v_add_u32_dpp", d_out, 0);
    run<8>("v_readlane_b32", d_out, 0);
    run<11>("v_pk_mul_f32", d_out, 0);
    return 0;
}
