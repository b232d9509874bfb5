// Profiling helper (not part of the product): what one wave-level VALU instruction of each class costs a
// gfx950 SIMD, in SHADER CYCLES, and the clock the chip holds while doing it.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost [waves_per_simd=4]
//
// Method (MI355X_MICROARCH.md, "DVFS give-back" item 6): every wave stamps s_memtime (one tick = one shader cycle)
// and s_memrealtime (100 MHz) around a loop of REP x 8 independent instructions of one kind; W waves share each
// SIMD (grid = 256 CUs x W blocks of 256 threads), so
//     issue cost  = median over waves of  d(s_memtime) / (REP * 8 * W)      [cycles per wave-instruction per SIMD]
//     clock       = median d(s_memtime) / d(s_memrealtime) * 100 MHz
// No clock is assumed anywhere.  Output: one JSON object (classes follow rocprofv3's SQ_INSTS_VALU_* counters,
// which is what tools/roofline_valu.py combines these costs with).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 2048

enum {
    K_FMA_F32, K_ADD_F32, K_MUL_F32, K_FMA_F64, K_ADD_F64, K_MUL_F64, K_MIN_F64, K_RCP_F64, K_RSQ_F64, K_SQRT_F32, K_RCP_F32,
    K_CVT_F32_F64, K_CVT_F64_F32, K_ADD_U32, K_MUL_LO_U32, K_LSHL_B64, K_ADD_U32_DPP, K_MOV_DPP, K_MOV_B32, K_CNDMASK,
    K_CMP_F32, K_CMP_F64, K_READLANE, K_MBCNT, K_PK_MUL_F32, K_PK_FMA_F32, K_COUNT
};
static const char *NAMES[K_COUNT] = {
    "v_fma_f32", "v_add_f32", "v_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_min_f64", "v_rcp_f64", "v_rsq_f64",
    "v_sqrt_f32", "v_rcp_f32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_add_u32", "v_mul_lo_u32", "v_lshlrev_b64",
    "v_add_u32_dpp", "v_mov_b32_dpp", "v_mov_b32", "v_cndmask_b32", "v_cmp_lt_f32", "v_cmp_lt_f64", "v_readlane_b32",
    "v_mbcnt_lo_u32_b32", "v_pk_mul_f32", "v_pk_fma_f32"};

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t *stamps, float *sink, float a, double b)
{
    float x[8];
    double d[8];
    uint32_t u[8];
    for (int q = 0; q < 8; q++) { x[q] = threadIdx.x * 1e-3f + a + q; d[q] = x[q] + b; u[q] = threadIdx.x + q; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < REP; i++) {
#define X(q) "%" #q
        if (KIND == K_FMA_F32) asm volatile(
#define S(q) "v_fma_f32 %" #q ", %" #q ", %8, %" #q "\n"
            R8(S)
#undef S
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));
        else if (KIND == K_ADD_F32) asm volatile(
#define S(q) "v_add_f32 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));
        else if (KIND == K_MUL_F32) asm volatile(
#define S(q) "v_mul_f32 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));
        else if (KIND == K_FMA_F64) asm volatile(
#define S(q) "v_fma_f64 %" #q ", %" #q ", %8, %" #q "\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(b));
        else if (KIND == K_ADD_F64) asm volatile(
#define S(q) "v_add_f64 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(b));
        else if (KIND == K_MUL_F64) asm volatile(
#define S(q) "v_mul_f64 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(b));
        else if (KIND == K_MIN_F64) asm volatile(
#define S(q) "v_min_f64 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(b));
        else if (KIND == K_RCP_F64) asm volatile(
#define S(q) "v_rcp_f64 %" #q ", %" #q "\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
        else if (KIND == K_RSQ_F64) asm volatile(
#define S(q) "v_rsq_f64 %" #q ", %" #q "\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
        else if (KIND == K_SQRT_F32) asm volatile(
#define S(q) "v_sqrt_f32 %" #q ", %" #q "\n"
            R8(S)
#undef S
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        else if (KIND == K_RCP_F32) asm volatile(
#define S(q) "v_rcp_f32 %" #q ", %" #q "\n"
            R8(S)
#undef S
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        else if (KIND == K_CVT_F32_F64) asm volatile(
            "v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
            "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
            : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
        else if (KIND == K_CVT_F64_F32) asm volatile(
            "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
            "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
        else if (KIND == K_ADD_U32) asm volatile(
#define S(q) "v_add_u32 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(i));
        else if (KIND == K_MUL_LO_U32) asm volatile(
#define S(q) "v_mul_lo_u32 %" #q ", %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(i | 1));
        else if (KIND == K_LSHL_B64) asm volatile(
#define S(q) "v_lshlrev_b64 %" #q ", 1, %" #q "\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
        else if (KIND == K_ADD_U32_DPP) asm volatile(
#define S(q) "v_add_u32_dpp %" #q ", %" #q ", %" #q " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
        else if (KIND == K_MOV_DPP) asm volatile(
#define S(q) "v_mov_b32_dpp %" #q ", %" #q " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
        else if (KIND == K_MOV_B32) asm volatile(
#define S(q) "v_mov_b32 %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(i));
        else if (KIND == K_CNDMASK) asm volatile(
#define S(q) "v_cndmask_b32 %" #q ", %" #q ", %8, vcc\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(i) : "vcc");
        else if (KIND == K_CMP_F32) asm volatile(
#define S(q) "v_cmp_lt_f32 vcc, %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a) : "vcc");
        else if (KIND == K_CMP_F64) asm volatile(
#define S(q) "v_cmp_lt_f64 vcc, %" #q ", %8\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(b) : "vcc");
        else if (KIND == K_READLANE) {
            int s;
            asm volatile(
#define S(q) "v_readlane_b32 %8, %" #q ", 3\n"
                R8(S)
#undef S
                : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "=s"(s));
        } else if (KIND == K_MBCNT) asm volatile(
#define S(q) "v_mbcnt_lo_u32_b32 %" #q ", -1, %" #q "\n"
            R8(S)
#undef S
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
        else if (KIND == K_PK_MUL_F32) asm volatile(
#define S(q) "v_pk_mul_f32 %" #q ", %" #q ", %" #q "\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
        else if (KIND == K_PK_FMA_F32) asm volatile(
#define S(q) "v_pk_fma_f32 %" #q ", %" #q ", %" #q ", %" #q "\n"
            R8(S)
#undef S
            : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int q = 0; q < 8; q++) acc += x[q] + (float)d[q] + (float)u[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <int KIND>
static void run(int W, int ncu, uint64_t *d_st, float *d_sink, bool last)
{
    const int blocks = ncu * W, nw = blocks * 4;
    for (int rep = 0; rep < 3; rep++) k<KIND><<<blocks, 256>>>(d_st, d_sink, 1.0f, 1.0);     // last launch is read
    hipDeviceSynchronize();
    std::vector<uint64_t> h(2 * nw);
    hipMemcpy(h.data(), d_st, sizeof(uint64_t) * 2 * nw, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), clk(nw);
    for (int w = 0; w < nw; w++) {
        cyc[w] = (double)h[2 * w] / ((double)REP * 8 * W);
        clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 100e6;
    }
    printf("  \"%s\": {\"cycles_per_wave_instr_per_simd\": %.3f, \"clock_ghz\": %.3f}%s\n", NAMES[KIND], median(cyc),
           median(clk) * 1e-9, last ? "" : ",");
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 4;
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    uint64_t *d_st;
    float *d_sink;
    hipMalloc(&d_st, sizeof(uint64_t) * 2 * ncu * W * 4);
    hipMalloc(&d_sink, sizeof(float) * ncu * W * 256);
    printf("{\"waves_per_simd\": %d, \"cus\": %d, \"rep\": %d, \"instr\": {\n", W, ncu, REP * 8);
    run<K_FMA_F32>(W, ncu, d_st, d_sink, false);
    run<K_ADD_F32>(W, ncu, d_st, d_sink, false);
    run<K_MUL_F32>(W, ncu, d_st, d_sink, false);
    run<K_FMA_F64>(W, ncu, d_st, d_sink, false);
    run<K_ADD_F64>(W, ncu, d_st, d_sink, false);
    run<K_MUL_F64>(W, ncu, d_st, d_sink, false);
    run<K_MIN_F64>(W, ncu, d_st, d_sink, false);
    run<K_RCP_F64>(W, ncu, d_st, d_sink, false);
    run<K_RSQ_F64>(W, ncu, d_st, d_sink, false);
    run<K_SQRT_F32>(W, ncu, d_st, d_sink, false);
    run<K_RCP_F32>(W, ncu, d_st, d_sink, false);
    run<K_CVT_F32_F64>(W, ncu, d_st, d_sink, false);
    run<K_CVT_F64_F32>(W, ncu, d_st, d_sink, false);
    run<K_ADD_U32>(W, ncu, d_st, d_sink, false);
    run<K_MUL_LO_U32>(W, ncu, d_st, d_sink, false);
    run<K_LSHL_B64>(W, ncu, d_st, d_sink, false);
    run<K_ADD_U32_DPP>(W, ncu, d_st, d_sink, false);
    run<K_MOV_DPP>(W, ncu, d_st, d_sink, false);
    run<K_MOV_B32>(W, ncu, d_st, d_sink, false);
    run<K_CNDMASK>(W, ncu, d_st, d_sink, false);
    run<K_CMP_F32>(W, ncu, d_st, d_sink, false);
    run<K_CMP_F64>(W, ncu, d_st, d_sink, false);
    run<K_READLANE>(W, ncu, d_st, d_sink, false);
    run<K_MBCNT>(W, ncu, d_st, d_sink, false);
    run<K_PK_MUL_F32>(W, ncu, d_st, d_sink, false);
    run<K_PK_FMA_F32>(W, ncu, d_st, d_sink, true);
    printf("}}\n");
    return 0;
}
