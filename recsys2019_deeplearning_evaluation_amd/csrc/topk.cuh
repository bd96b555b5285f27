// topk.cuh -- block-wide top-K selection in LDS, shared by the similarity build (per-column top-K,
// Compute_Similarity_Cython.pyx:523-562) and SLIM-BPR's get_S (per-row top-K, SLIM_BPR_Cython_Epoch.pyx:343-391).
// 4-pass radix select on an order-preserving uint32 key with bank-replicated histograms, then a bitonic sort of
// the survivors.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi355rec {

constexpr uint32_t ZERO_KEY = 0x80000000u;
constexpr int AUX_WORDS = 8192;  // 32 KiB: 256 bins x 32 bank replicas, later re-used as the candidate buffer
constexpr int MAX_TOPK = 4096;   // AUX_WORDS * 4 B / 8 B per candidate

// Order-preserving map float -> uint32 (larger float <=> larger key); +0.0 maps to ZERO_KEY.
__device__ __forceinline__ uint32_t float_key(float v) {
    uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

struct SelectScratch {
    uint32_t wave_tot[4];
    uint32_t digit, want, bin_count;
};

// Block-wide radix select: key of the `want`-th largest element (1-based) of
//   { kf(j).key : j < n, kf(j).active }  U  { virt_key repeated virt_cnt times }.
// On return every thread holds T (that key), need_eq (how many elements equal to T belong to the top `want`)
// and eq_total (how many elements equal T, virtual ones included).
template <int THREADS, class KeyFn>
__device__ void block_select(KeyFn kf, int n, uint32_t want, uint32_t virt_key, uint32_t virt_cnt, uint32_t *hist,
                             SelectScratch &sc, uint32_t &T, uint32_t &need_eq, uint32_t &eq_total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t prefix = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int w = tid; w < AUX_WORDS; w += THREADS) hist[w] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += THREADS) {
            uint32_t key;
            if (kf(j, key) && (pass == 0 || (key >> (shift + 8)) == prefix))
                atomicAdd(&hist[((key >> shift) & 255u) * 32 + (lane & 31)], 1u);
        }
        __syncthreads();
        uint32_t cnt = 0, suffix = 0;
        if (tid < 256) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) cnt += hist[tid * 32 + ((r + tid) & 31)];
            if (virt_cnt && (pass == 0 || (virt_key >> (shift + 8)) == prefix) && ((virt_key >> shift) & 255u) == (uint32_t)tid)
                cnt += virt_cnt;
            suffix = cnt;  // inclusive suffix sum inside the wave (towards higher bins)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t t = __shfl_down(suffix, off);
                if (lane + off < 64) suffix += t;
            }
            if (lane == 0) sc.wave_tot[wave] = suffix;
        }
        __syncthreads();
        if (tid < 256) {
            for (int w = wave + 1; w < 4; ++w) suffix += sc.wave_tot[w];
            const uint32_t above = suffix - cnt;
            if (suffix >= want && above < want) {
                sc.digit = tid;
                sc.want = want - above;
                sc.bin_count = cnt;
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | sc.digit;
        want = sc.want;
        eq_total = sc.bin_count;
        __syncthreads();
    }
    T = prefix;
    need_eq = want;
}

template <int THREADS>
__device__ void bitonic_sort_desc(uint64_t *a, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P; t += THREADS) {
                int ixj = t ^ j;
                if (ixj > t) {
                    uint64_t x = a[t], y = a[ixj];
                    bool desc = (t & k) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        a[t] = y;
                        a[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}


// Which cells take part in a top-K.
//   TOPK_ZEROS_COMPETE : the K largest cells of the whole array are taken, zeros included, and zeros are then dropped
//                        (Compute_Similarity_Cython.pyx:523-555, Triangular_Matrix.get_scipy_csr :1384-1404)
//   TOPK_NONZERO       : only non-zero cells compete (similarityMatrixTopK, Base/Recommender_utils.py:100-104)
//   TOPK_FINITE        : every cell except -inf competes and nothing is dropped (BaseRecommender.recommend :182-205,
//                        where -inf marks excluded items)
enum { TOPK_ZEROS_COMPETE = 0, TOPK_NONZERO = 1, TOPK_FINITE = 2 };

// Top-K of the n floats in LDS array `acc` (K = topK <= sortP, sortP a power of two, sortP * 8 B <= AUX_WORDS * 4 B).
//   npos / nneg     : TOPK_ZEROS_COMPETE / TOPK_NONZERO: number of strictly positive / negative cells;
//                     TOPK_FINITE: npos = number of cells > -inf, nneg = 0          (block-uniform, counted by the caller)
//   *ncand          : shared counter, must be 0 on entry
// Output: out_idx / out_val [topK] (out_val may be null), value-descending, ties broken towards the lower index,
// (-1, 0) padded.
//   idx_offset      : added to every emitted index (tiled callers: base of the tile)
//   idx_map         : if non-null the emitted index is idx_map[local index] (merge of per-tile candidates)
//   zero_count      : TOPK_ZEROS_COMPETE only: number of competing zeros; < 0 = n - npos - nneg
template <int THREADS>
__device__ void block_topk_emit(const float *acc, int n, int topK, int sortP, uint32_t npos, uint32_t nneg, int mode,
                                uint32_t *aux, SelectScratch &sc, uint32_t *ncand_shared, int *out_idx, float *out_val,
                                int idx_offset = 0, const int *idx_map = nullptr, long long zero_count = -1) {
    const int tid = threadIdx.x;
    const bool zeros_compete = mode == TOPK_ZEROS_COMPETE;
    const uint32_t nzero = zeros_compete ? (zero_count >= 0 ? (uint32_t)zero_count : (uint32_t)n - npos - nneg) : 0u;
    uint32_t K = (uint32_t)topK;
    if (!zeros_compete) K = min(K, npos + nneg);
    uint32_t T = ZERO_KEY, need_eq = 0, eq_total = 0;
    // all positives fit and no negative can displace a zero: nothing to select
    bool take_all = zeros_compete ? (npos <= K && (nneg == 0 || npos + nzero >= K)) : (npos + nneg <= K);
    if (!zeros_compete && take_all) T = 0u;                      // every candidate key is > 0
    auto candidate = [&](float v) { return mode == TOPK_FINITE ? v > -INFINITY : v != 0.f; };
    auto value_key = [&](int j, uint32_t &key) {
        const float v = acc[j];
        key = float_key(v);
        return candidate(v);
    };
    if (!take_all) {
        block_select<THREADS>(value_key, n, K, ZERO_KEY, nzero, aux, sc, T, need_eq, eq_total);
        if (zeros_compete && T == ZERO_KEY) need_eq = 0;          // zeros are never emitted
    }
    uint32_t T2 = 0;  // tie-break on the index when more cells equal T than fit: lowest index wins
    const bool partial_ties = need_eq > 0 && need_eq < eq_total;
    if (partial_ties) {
        uint32_t dummy_need, dummy_tot;
        auto index_key = [&](int j, uint32_t &key) {
            const float v = acc[j];
            key = ~(uint32_t)j;
            return candidate(v) && float_key(v) == T;
        };
        block_select<THREADS>(index_key, n, need_eq, 0u, 0u, aux, sc, T2, dummy_need, dummy_tot);
    }
    __syncthreads();
    uint64_t *cand = reinterpret_cast<uint64_t *>(aux);
    for (int j = tid; j < n; j += THREADS) {
        const float v = acc[j];
        if (!candidate(v)) continue;
        const uint32_t key = float_key(v);
        const bool take = key > T || (need_eq > 0 && key == T && (!partial_ties || ~(uint32_t)j >= T2));
        if (take) {
            const uint32_t slot_c = atomicAdd(ncand_shared, 1u);
            if (slot_c < (uint32_t)sortP) cand[slot_c] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)j);
        }
    }
    __syncthreads();
    const int ncand = min((int)*ncand_shared, topK);
    for (int t = ncand + tid; t < sortP; t += THREADS) cand[t] = 0ull;
    __syncthreads();
    bitonic_sort_desc<THREADS>(cand, sortP);
    for (int t = tid; t < topK; t += THREADS) {
        int idx = -1;
        float val = 0.f;
        if (t < ncand) {
            const uint64_t e = cand[t];
            idx = (int)(~(uint32_t)(e & 0xFFFFFFFFull));
            idx = idx_map ? idx_map[idx] : idx + idx_offset;
            val = key_float((uint32_t)(e >> 32));
        }
        out_idx[t] = idx;
        if (out_val) out_val[t] = val;
    }
}

}  // namespace mi355rec
