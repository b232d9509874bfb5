// Verification helper (not part of the product): the range-restricted sqrt / divide sequences of
// toycluster_amd/csrc/tc_lean.h against the compiler's IEEE ones, bit for bit, on random inputs drawn from the
// ranges the kernels feed them.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I toycluster_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "tc_lean.h"

__device__ uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ double u01(uint64_t &s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
__device__ double logu(uint64_t &s, double lo, double hi) { return exp(log(lo) + (log(hi) - log(lo)) * u01(s)); }

__global__ void check(unsigned long long *bad, int per_thread)
{
    uint64_t s = 0x1234567ull + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 7919;
    unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
    for (int i = 0; i < per_thread; i++) {
        // f64 sqrt of a sum of squares of kpc-scale separations
        double x = logu(s, 1e-8, 1e10);
        if (__builtin_bit_cast(uint64_t, tc_sqrt_f64_lean(x)) != __builtin_bit_cast(uint64_t, sqrt(x))) b0++;
        // f32 sqrt of r2 in box units
        float r2 = (float)logu(s, 1e-14, 2.0);
        float r = sqrtf(r2);
        if (__builtin_bit_cast(uint32_t, tc_sqrt_f32_lean(r2)) != __builtin_bit_cast(uint32_t, r)) b1++;
        // f32 quotient r / h, r <= h
        float h = (float)logu(s, 1e-5, 1.0);
        float rr = h * (float)u01(s);
        if (__builtin_bit_cast(uint32_t, tc_div_f32_lean(rr, h)) != __builtin_bit_cast(uint32_t, rr / h)) b2++;
        // f64 reciprocal of an f32 distance
        double rd = (double)r;
        if (rd > 0 && __builtin_bit_cast(uint64_t, tc_rcp_f64_lean(rd)) != __builtin_bit_cast(uint64_t, 1.0 / rd)) b3++;
        // f64 quotient of two numbers anywhere in 1e-60 .. 1e60, either sign
        double a = logu(s, 1e-60, 1e60) * (u01(s) < 0.5 ? -1 : 1), b = logu(s, 1e-60, 1e60) * (u01(s) < 0.5 ? -1 : 1);
        if (__builtin_bit_cast(uint64_t, tc_div_f64_lean(a, b)) != __builtin_bit_cast(uint64_t, a / b)) b4++;
    }
    atomicAdd(&bad[0], b0); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2); atomicAdd(&bad[3], b3); atomicAdd(&bad[4], b4);
}

__global__ void specials(double *o)
{
    o[0] = tc_sqrt_f64_lean(0.0); o[1] = sqrt(0.0);
    o[2] = tc_sqrt_f32_lean(0.0f); o[3] = sqrtf(0.0f);
    o[4] = tc_div_f32_lean(0.0f, 0.25f); o[5] = 0.0f / 0.25f;
    o[6] = tc_div_f32_lean(0.25f, 0.25f); o[7] = 1.0;
    o[8] = tc_rcp_f64_lean(0.0); o[9] = 1.0 / 0.0 * (o[1] + 1);
}

int main()
{
    unsigned long long *bad, h[5];
    hipMalloc(&bad, 5 * sizeof(*bad));
    hipMemset(bad, 0, 5 * sizeof(*bad));
    const int blocks = 4096, threads = 256, per = 256;
    check<<<blocks, threads>>>(bad, per);
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("samples per function: %.3g\n", (double)blocks * threads * per);
    printf("mismatches: sqrt_f64 %llu  sqrt_f32 %llu  div_f32 %llu  rcp_f64 %llu  div_f64 %llu\n", h[0], h[1], h[2], h[3], h[4]);
    double *o, ho[10];
    hipMalloc(&o, sizeof(ho));
    specials<<<1, 1>>>(o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; i += 2) printf("special %d: lean %g  ieee %g\n", i / 2, ho[i], ho[i + 1]);
    return (h[0] | h[1] | h[2] | h[3] | h[4]) ? 1 : 0;
}
