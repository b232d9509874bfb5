// Profiling helper (not part of the product): do scalar instructions, s_nop and LDS reads of one wave take issue
// slots away from the VALU instructions of the OTHER waves of a gfx950 SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix [waves_per_simd=4]
//
// Every wave runs REP x (a group of 8 VALU instructions interleaved with 0 or 8 instructions of another kind); W waves
// share each SIMD.  Reported: shader cycles (s_memtime) per group per SIMD = median d(s_memtime) / (REP * W).  If the
// other kind is free (issued in parallel from another wave) a mixed group costs what the 8 VALU instructions cost alone.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 4096
enum { M_F32, M_F32_SALU, M_F32_NOP0, M_F32_NOP1, M_SALU, M_F64, M_F64_SALU, M_F32_LDS, M_F32_SALUDEP, M_F32_VCCHOP, M_F32_WAIT,
       M_F32_BR, M_COUNT };
static const char *NAMES[M_COUNT] = {"8 v_fma_f32", "8 v_fma_f32 + 8 s_add_u32", "8 v_fma_f32 + 8 s_nop 0", "8 v_fma_f32 + 8 s_nop 1",
    "8 s_add_u32", "8 v_fma_f64", "8 v_fma_f64 + 8 s_add_u32", "8 v_fma_f32 + 4 ds_read_b32 (+wait)",
    "8 v_fma_f32 + 8 dependent s_add_u32", "8 x (v_cmp -> s_and_b64 vcc -> v_cndmask) hop", "8 v_fma_f32 + 8 s_waitcnt lgkmcnt(0)",
    "8 v_fma_f32 + 8 untaken s_cbranch_scc1"};

template <int M>
__global__ __launch_bounds__(256) void k(uint64_t *stamps, float *sink, float a, double b)
{
    __shared__ float lds[1024];
    float x[8];
    double d[8];
    for (int q = 0; q < 8; q++) { x[q] = threadIdx.x * 1e-3f + a + q; d[q] = x[q] + b; }
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = a; lds[threadIdx.x + 512] = a; lds[threadIdx.x + 768] = a;
    __syncthreads();
    uint32_t la = threadIdx.x * 4;
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
#define V32(q) "v_fma_f32 %" #q ", %" #q ", %8, %" #q "\n"
#define V64(q) "v_fma_f64 %" #q ", %" #q ", %8, %" #q "\n"
#define OUT32 "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
#define OUT64 "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
        if (M == M_F32) asm volatile(V32(0) V32(1) V32(2) V32(3) V32(4) V32(5) V32(6) V32(7) : OUT32 : "v"(a));
        else if (M == M_F32_SALU) asm volatile(
            V32(0) "s_add_u32 s20, s20, 1\n" V32(1) "s_add_u32 s21, s21, 1\n" V32(2) "s_add_u32 s22, s22, 1\n" V32(3) "s_add_u32 s23, s23, 1\n"
            V32(4) "s_add_u32 s24, s24, 1\n" V32(5) "s_add_u32 s25, s25, 1\n" V32(6) "s_add_u32 s26, s26, 1\n" V32(7) "s_add_u32 s27, s27, 1\n"
            : OUT32 : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        else if (M == M_F32_SALUDEP) asm volatile(
            V32(0) "s_add_u32 s20, s20, 1\n" V32(1) "s_add_u32 s20, s20, 1\n" V32(2) "s_add_u32 s20, s20, 1\n" V32(3) "s_add_u32 s20, s20, 1\n"
            V32(4) "s_add_u32 s20, s20, 1\n" V32(5) "s_add_u32 s20, s20, 1\n" V32(6) "s_add_u32 s20, s20, 1\n" V32(7) "s_add_u32 s20, s20, 1\n"
            : OUT32 : "v"(a) : "s20", "scc");
        else if (M == M_F32_NOP0) asm volatile(
            V32(0) "s_nop 0\n" V32(1) "s_nop 0\n" V32(2) "s_nop 0\n" V32(3) "s_nop 0\n" V32(4) "s_nop 0\n" V32(5) "s_nop 0\n" V32(6) "s_nop 0\n" V32(7) "s_nop 0\n"
            : OUT32 : "v"(a));
        else if (M == M_F32_NOP1) asm volatile(
            V32(0) "s_nop 1\n" V32(1) "s_nop 1\n" V32(2) "s_nop 1\n" V32(3) "s_nop 1\n" V32(4) "s_nop 1\n" V32(5) "s_nop 1\n" V32(6) "s_nop 1\n" V32(7) "s_nop 1\n"
            : OUT32 : "v"(a));
        else if (M == M_F32_WAIT) asm volatile(
            V32(0) "s_waitcnt lgkmcnt(0)\n" V32(1) "s_waitcnt lgkmcnt(0)\n" V32(2) "s_waitcnt lgkmcnt(0)\n" V32(3) "s_waitcnt lgkmcnt(0)\n"
            V32(4) "s_waitcnt lgkmcnt(0)\n" V32(5) "s_waitcnt lgkmcnt(0)\n" V32(6) "s_waitcnt lgkmcnt(0)\n" V32(7) "s_waitcnt lgkmcnt(0)\n"
            : OUT32 : "v"(a));
        else if (M == M_F32_BR) asm volatile(
            "s_cmp_eq_u32 s20, s20\n s_cmp_lg_u32 s20, s20\n"
            V32(0) "s_cbranch_scc1 1f\n" V32(1) "s_cbranch_scc1 1f\n" V32(2) "s_cbranch_scc1 1f\n" V32(3) "s_cbranch_scc1 1f\n"
            V32(4) "s_cbranch_scc1 1f\n" V32(5) "s_cbranch_scc1 1f\n" V32(6) "s_cbranch_scc1 1f\n" V32(7) "s_cbranch_scc1 1f\n 1:\n"
            : OUT32 : "v"(a) : "s20", "scc");
        else if (M == M_SALU) asm volatile(
            "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
            "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
            ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        else if (M == M_F64) asm volatile(V64(0) V64(1) V64(2) V64(3) V64(4) V64(5) V64(6) V64(7) : OUT64 : "v"(b));
        else if (M == M_F64_SALU) asm volatile(
            V64(0) "s_add_u32 s20, s20, 1\n" V64(1) "s_add_u32 s21, s21, 1\n" V64(2) "s_add_u32 s22, s22, 1\n" V64(3) "s_add_u32 s23, s23, 1\n"
            V64(4) "s_add_u32 s24, s24, 1\n" V64(5) "s_add_u32 s25, s25, 1\n" V64(6) "s_add_u32 s26, s26, 1\n" V64(7) "s_add_u32 s27, s27, 1\n"
            : OUT64 : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        else if (M == M_F32_LDS) asm volatile(
            "ds_read_b32 %9, %13\n" V32(0) V32(1) "ds_read_b32 %10, %13 offset:1024\n" V32(2) V32(3)
            "ds_read_b32 %11, %13 offset:2048\n" V32(4) V32(5) "ds_read_b32 %12, %13 offset:3072\n" V32(6) V32(7) "s_waitcnt lgkmcnt(0)\n"
            : OUT32, "+v"(a), "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3) : "v"(la));
        else if (M == M_F32_VCCHOP) asm volatile(
#define HOP(q) "v_cmp_lt_f32 vcc, %" #q ", %8\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %" #q ", %" #q ", %8, vcc\n"
            HOP(0) HOP(1) HOP(2) HOP(3) HOP(4) HOP(5) HOP(6) HOP(7)
            : OUT32 : "v"(a) : "vcc", "scc");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float acc = l0 + l1 + l2 + l3;
    for (int q = 0; q < 8; q++) acc += x[q] + (float)d[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M>
static void run(int W, int ncu, uint64_t *d_st, float *d_sink)
{
    const int blocks = ncu * W, nw = blocks * 4;
    for (int rep = 0; rep < 3; rep++) k<M><<<blocks, 256>>>(d_st, d_sink, 1.0f, 1.0);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nw);
    hipMemcpy(h.data(), d_st, sizeof(uint64_t) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-52s %7.2f cycles per group per SIMD\n", NAMES[M], (double)h[nw / 2] / ((double)REP * W));
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    uint64_t *d_st; float *d_sink;
    hipMalloc(&d_st, sizeof(uint64_t) * ncu * W * 4);
    hipMalloc(&d_sink, sizeof(float) * ncu * W * 256);
    printf("%s, %d CUs, %d waves per SIMD\n", pr.gcnArchName, ncu, W);
    run<M_F32>(W, ncu, d_st, d_sink); run<M_F32_SALU>(W, ncu, d_st, d_sink); run<M_F32_SALUDEP>(W, ncu, d_st, d_sink);
    run<M_F32_NOP0>(W, ncu, d_st, d_sink); run<M_F32_NOP1>(W, ncu, d_st, d_sink); run<M_F32_WAIT>(W, ncu, d_st, d_sink);
    run<M_F32_BR>(W, ncu, d_st, d_sink); run<M_SALU>(W, ncu, d_st, d_sink); run<M_F64>(W, ncu, d_st, d_sink);
    run<M_F64_SALU>(W, ncu, d_st, d_sink); run<M_F32_LDS>(W, ncu, d_st, d_sink); run<M_F32_VCCHOP>(W, ncu, d_st, d_sink);
    return 0;
}
