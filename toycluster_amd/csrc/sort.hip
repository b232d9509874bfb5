/*
 * sort.hip -- device sort of (128-bit Peano key, particle index) pairs.
 *
 * Replaces the reference's serial index heapsort (src/sort.c:185-195 -> gsl_heapsort_index,
 * comparator src/peano.c:33-39).  One stable LSD radix sort (rocPRIM device primitive) on the leading
 * bits of the high key half -- all 64 (21 Hilbert levels) for up to 2^20 particles, fewer for large
 * sets, where a handful of levels below the deepest cell-table level already separates all but
 * coincident or extremely close particles and every radix pass saved is a full sweep over the data --
 * followed by an exact fix-up of the runs whose sorted bits tie, ordering them by the whole 128-bit key
 * and then by previous position.  The result is the order of a stable sort on the full 128-bit key
 * whatever the number of bits: particles with identical keys keep their previous relative order (the
 * reference's heapsort leaves their order unspecified).
 *
 * Bit ranges below 64 are only used above rocPRIM's merge-sort limit (2^20 items): its merge-sort
 * comparator builds its mask with (T(1) << (begin_bit + bits)) - 1, which is undefined for a range that
 * ends at bit 64 of a 64-bit key and mis-sorts; the Onesweep path used for larger inputs extracts digits
 * and is fine.
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

/* strict order of two entries of a run with equal sorted bits: full key, then previous index */
__device__ __forceinline__ bool tie_less(const tc_u128 *__restrict__ key, uint32_t a, uint32_t b)
{
    const tc_u128 ka = key[a], kb = key[b];
    return ka < kb || (ka == kb && a < b);
}

/* One thread per run head.  Runs are 2-3 entries long when they exist at all; long runs (many
 * particles inside one 2^-21 cell: degenerate input) are heap-sorted by the same thread so that the
 * cost stays O(L log L). */
__global__ __launch_bounds__(TBS) void k_fix_ties(const uint64_t *__restrict__ hi_sorted, uint32_t *__restrict__ idx,
                                                  const tc_u128 *__restrict__ key, size_t n, int shift)
{
    size_t i = (size_t)blockIdx.x * TBS + threadIdx.x;
    if (i + 1 >= n) return;
    const uint64_t h = hi_sorted[i] >> shift;
    if ((hi_sorted[i + 1] >> shift) != h || (i > 0 && (hi_sorted[i - 1] >> shift) == h)) return;   /* not a run head */
    size_t e = i + 2;
    while (e < n && (hi_sorted[e] >> shift) == h) e++;
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

#define TC_SORT_FULL_BITS_LIMIT ((size_t)1 << 20)     /* rocPRIM's default merge_sort_limit */

static hipError_t radix_temp_query(size_t n, int begin_bit, size_t *b)
{
    *b = 0;
    return rocprim::radix_sort_pairs(nullptr, *b, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                     (const uint32_t *)nullptr, (uint32_t *)nullptr, n, begin_bit, 64);
}

/* temp layout: [rocprim temp][k0: n u64][k1: n u64].  rocPRIM does not check the scratch size once a pointer
 * is passed and its need may depend on the bit range, so the allocation covers every range
 * tc_sort_pairs_u128 may ask for and the sort re-checks before launching anything. */
int tc_sort_temp_bytes(size_t n, size_t *bytes)
{
    size_t worst = 0;
    for (int begin = 0; begin < 64; begin++) {
        size_t b = 0;
        if (radix_temp_query(n, begin, &b) != hipSuccess) return -1;
        if (b > worst) worst = b;
    }
    *bytes = align256(worst) + 2 * align256(n * sizeof(uint64_t));
    return 0;
}

/* vin must be the identity permutation (k_keys writes it): the tie order relies on it.
 * sort_bits: leading key bits the radix sort looks at (1..64; forced to 64 up to 2^20 items, see above). */
int tc_sort_pairs_u128(void *tmp, size_t tmp_bytes, const tc_u128 *kin, tc_u128 *kout,
                       const uint32_t *vin, uint32_t *vout, size_t n, int sort_bits, hipStream_t s)
{
    if (sort_bits < 1 || sort_bits > 64 || n <= TC_SORT_FULL_BITS_LIMIT) sort_bits = 64;
    const int shift = 64 - sort_bits;
    size_t b = 0;
    if (radix_temp_query(n, shift, &b) != hipSuccess) return -1;
    char *p = (char *)tmp;
    void *rp_tmp = p;                 p += align256(b);
    uint64_t *k0 = (uint64_t *)p;     p += align256(n * sizeof(uint64_t));
    uint64_t *k1 = (uint64_t *)p;     p += align256(n * sizeof(uint64_t));
    if ((size_t)(p - (char *)tmp) > tmp_bytes) return -1;
    unsigned g = (unsigned)((n + TBS - 1) / TBS);

    k_split_hi<<<g, TBS, 0, s>>>(kin, k0, n);
    if (rocprim::radix_sort_pairs(rp_tmp, b, k0, k1, vin, vout, n, shift, 64, s) != hipSuccess) return -1;
    k_fix_ties<<<g, TBS, 0, s>>>(k1, vout, kin, n, shift);
    k_gather_key<<<g, TBS, 0, s>>>(kin, vout, kout, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
