/*
 * sort.hip -- device sort of (128-bit Peano key, particle index) pairs.
 *
 * Replaces the reference's serial index heapsort (src/sort.c:185-195 -> gsl_heapsort_index,
 * comparator src/peano.c:33-39).  One stable LSD radix sort (rocPRIM device primitive) on the high
 * 64 key bits -- 21 Hilbert levels, i.e. cells of 2^-21 of the box -- followed by an exact fix-up of
 * the (practically only coincident-particle) runs whose high halves tie, ordering them by the low
 * half and then by previous position.  The result is the order of a stable sort on the full
 * 128-bit key: particles with identical keys keep their previous relative order (the reference's
 * heapsort leaves their order unspecified).
 */
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "tc_ctx.h"

#define TBS 256

__global__ __launch_bounds__(TBS) void k_split_hi(const tc_u128 *__restrict__ key, uint64_t *__restrict__ hi, size_t n)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i < n) hi[i] = (uint64_t)(key[i] >> 64);
}

/* strict order of two entries of a run with equal high halves: low half, then previous index */
__device__ __forceinline__ bool tie_less(const tc_u128 *__restrict__ key, uint32_t a, uint32_t b)
{
    uint64_t la = (uint64_t)key[a], lb = (uint64_t)key[b];
    return la < lb || (la == lb && a < b);
}

/* One thread per run head.  Runs are 2-3 entries long when they exist at all; long runs (many
 * particles inside one 2^-21 cell: degenerate input) are heap-sorted by the same thread so that the
 * cost stays O(L log L). */
__global__ __launch_bounds__(TBS) void k_fix_ties(const uint64_t *__restrict__ hi_sorted, uint32_t *__restrict__ idx,
                                                  const tc_u128 *__restrict__ key, size_t n)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i + 1 >= n) return;
    const uint64_t h = hi_sorted[i];
    if (hi_sorted[i + 1] != h || (i > 0 && hi_sorted[i - 1] == h)) return;      /* not a run head */
    size_t e = i + 2;
    while (e < n && hi_sorted[e] == h) e++;
    uint32_t *a = idx + i;
    const size_t L = e - i;
    if (L <= 16) {                                   /* insertion sort */
        for (size_t p = 1; p < L; p++) {
            uint32_t v = a[p];
            size_t q = p;
            while (q > 0 && tie_less(key, v, a[q - 1])) { a[q] = a[q - 1]; q--; }
            a[q] = v;
        }
        return;
    }
    /* heapsort on the total order (lo, previous index) */
    for (size_t start = L / 2; start-- > 0;) {
        size_t root = start;
        for (;;) {
            size_t child = 2 * root + 1;
            if (child >= L) break;
            if (child + 1 < L && tie_less(key, a[child], a[child + 1])) child++;
            if (!tie_less(key, a[root], a[child])) break;
            uint32_t t = a[root]; a[root] = a[child]; a[child] = t;
            root = child;
        }
    }
    for (size_t end = L - 1; end > 0; end--) {
        uint32_t t = a[0]; a[0] = a[end]; a[end] = t;
        size_t root = 0;
        for (;;) {
            size_t child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && tie_less(key, a[child], a[child + 1])) child++;
            if (!tie_less(key, a[root], a[child])) break;
            uint32_t t2 = a[root]; a[root] = a[child]; a[child] = t2;
            root = child;
        }
    }
}

__global__ __launch_bounds__(TBS) void k_gather_key(const tc_u128 *__restrict__ key, const uint32_t *__restrict__ idx,
                                                    tc_u128 *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i < n) out[i] = key[idx[i]];
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

/* temp layout: [rocprim temp][k0: n u64][k1: n u64] */
int tc_sort_temp_bytes(size_t n, size_t *bytes)
{
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64);
    if (e != hipSuccess) return -1;
    *bytes = align256(b) + 2 * align256(n * sizeof(uint64_t));
    return 0;
}

/* vin must be the identity permutation (k_keys writes it): the tie order relies on it */
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
    if ((size_t)(p - (char *)tmp) > tmp_bytes) return -1;
    unsigned g = (unsigned)((n + TBS - 1) / TBS);

    k_split_hi<<<g, TBS, 0, s>>>(kin, k0, n);
    if (rocprim::radix_sort_pairs(rp_tmp, b, k0, k1, vin, vout, n, 0, 64, s) != hipSuccess) return -1;
    k_fix_ties<<<g, TBS, 0, s>>>(k1, vout, kin, n);
    k_gather_key<<<g, TBS, 0, s>>>(kin, vout, kout, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
