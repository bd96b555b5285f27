// sampling.cuh -- counter-based RNG and the reference's BPR sampling rule, shared by the MF and SLIM-BPR epochs.
#pragma once

#include <hip/hip_runtime.h>

namespace mi355rec {

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// draw number `d` of sample `sid`: stateless, so every lane of the wavefront computes the same value
__device__ __forceinline__ unsigned draw32(unsigned long long seed, unsigned long long sid, unsigned d) {
    return (unsigned)(mix64(seed ^ mix64(sid * 0xD1B54A32D192ED03ull + d)) >> 32);
}
__device__ __forceinline__ int bounded(unsigned r, int n) { return (int)(((unsigned long long)r * (unsigned)n) >> 32); }

// is `item` absent from the sorted profile [row, row + n)?  (the reference scans linearly, .pyx:975-983)
__device__ __forceinline__ bool profile_lacks(const int *row, int n, int item) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (row[mid] < item) lo = mid + 1; else hi = mid;
    }
    return lo == n || row[lo] != item;
}

// sampleBPR_Cython (MatrixFactorization_Cython_Epoch.pyx:940-985 == SLIM_BPR_Cython_Epoch.pyx:439-483): a user with
// at least one and fewer than n_items interactions, one of its items, and an item outside its profile by rejection.
__device__ __forceinline__ void sample_bpr(unsigned long long seed, unsigned long long sid, int n_users, int n_items,
                                           const int *indptr, const int *indices, int &u, int &i, int &j) {
    unsigned d = 0;
    int start = 0, n_seen = 0;
    do {
        u = bounded(draw32(seed, sid, d++), n_users);
        start = indptr[u];
        n_seen = indptr[u + 1] - start;
    } while (n_seen == 0 || n_seen == n_items);
    const int *row = indices + start;
    i = row[bounded(draw32(seed, sid, d++), n_seen)];
    do { j = bounded(draw32(seed, sid, d++), n_items); } while (!profile_lacks(row, n_seen, j));
}

}  // namespace mi355rec
