/*
 * sort.hip -- device sort of (128-bit Peano key, particle index) pairs.
 *
 * Replaces the reference's serial index heapsort (src/sort.c:185-195 -> gsl_heapsort_index,
 * comparator src/peano.c:33-39).  LSD radix sort (rocPRIM device primitive) in two stable
 * 64-bit passes: first by the low key half, then by the high half.  The sort is stable, so
 * particles with identical keys keep their previous relative order (the reference's heapsort
 * leaves their order unspecified).
 */
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "tc_ctx.h"

#define TBS 256

__global__ __launch_bounds__(TBS) void k_split_lo(const tc_u128 *__restrict__ key, uint64_t *__restrict__ lo, size_t n)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i < n) lo[i] = (uint64_t)key[i];
}

__global__ __launch_bounds__(TBS) void k_gather_hi(const tc_u128 *__restrict__ key, const uint32_t *__restrict__ idx,
                                                   uint64_t *__restrict__ hi, size_t n)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i < n) hi[i] = (uint64_t)(key[idx[i]] >> 64);
}

__global__ __launch_bounds__(TBS) void k_gather_key(const tc_u128 *__restrict__ key, const uint32_t *__restrict__ idx,
                                                    tc_u128 *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i < n) out[i] = key[idx[i]];
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

/* temp layout: [rocprim temp][k0: n u64][k1: n u64][v1: n u32] */
int tc_sort_temp_bytes(size_t n, size_t *bytes)
{
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64);
    if (e != hipSuccess) return -1;
    *bytes = align256(b) + 2 * align256(n * sizeof(uint64_t)) + align256(n * sizeof(uint32_t));
    return 0;
}

int tc_sort_pairs_u128(void *tmp, size_t tmp_bytes, const tc_u128 *kin, tc_u128 *kout,
                       const uint32_t *vin, uint32_t *vout, size_t n, hipStream_t s)
{
    size_t b = 0;
    if (rocprim::radix_sort_pairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64) != hipSuccess)
        return -1;
    char *p = (char *)tmp;
    void *rp_tmp = p;                 p += align256(b);
    uint64_t *k0 = (uint64_t *)p;     p += align256(n * sizeof(uint64_t));
    uint64_t *k1 = (uint64_t *)p;     p += align256(n * sizeof(uint64_t));
    uint32_t *v1 = (uint32_t *)p;     p += align256(n * sizeof(uint32_t));
    if ((size_t)(p - (char *)tmp) > tmp_bytes) return -1;
    unsigned g = (unsigned)((n + TBS - 1) / TBS);

    /* pass 1: by the low half (bits 2..63 carry information, src/peano.c:200) */
    k_split_lo<<<g, TBS, 0, s>>>(kin, k0, n);
    if (rocprim::radix_sort_pairs(rp_tmp, b, k0, k1, vin, v1, n, 2, 64, s) != hipSuccess) return -1;
    /* pass 2: stable, by the high half */
    k_gather_hi<<<g, TBS, 0, s>>>(kin, v1, k0, n);
    if (rocprim::radix_sort_pairs(rp_tmp, b, k0, k1, v1, vout, n, 0, 64, s) != hipSuccess) return -1;
    k_gather_key<<<g, TBS, 0, s>>>(kin, vout, kout, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
