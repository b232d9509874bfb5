/*
 * sort.hip -- device sort of (128-bit Peano key, particle index) pairs.
 *
 * Replaces the reference's serial index heapsort (src/sort.c:185-195 -> gsl_heapsort_index,
 * comparator src/peano.c:33-39).  rocPRIM's LSD radix sort over key bits [2,128) (the two
 * low key bits are always zero, src/peano.c:200).  The radix sort is stable, so particles
 * with identical keys keep their previous relative order (the reference's heapsort leaves
 * their order unspecified).
 */
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "tc_ctx.h"

int tc_sort_temp_bytes(size_t n, size_t *bytes)
{
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (const tc_u128 *)nullptr, (tc_u128 *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 2, 128);
    if (e != hipSuccess) return -1;
    *bytes = b;
    return 0;
}

int tc_sort_pairs_u128(void *tmp, size_t tmp_bytes, const tc_u128 *kin, tc_u128 *kout,
                       const uint32_t *vin, uint32_t *vout, size_t n, hipStream_t s)
{
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 2, 128, s);
    return e == hipSuccess ? 0 : -1;
}
