// topk.cuh -- block-wide top-K selection in LDS, shared by the similarity build (per-column top-K,
// Compute_Similarity_Cython.pyx:523-562) and SLIM-BPR's get_S (per-row top-K, SLIM_BPR_Cython_Epoch.pyx:343-391).
// Radix select on an order-preserving uint32 key (bank-replicated histograms, digit windows placed where the keys
// differ, early exit to a small candidate superset), then a counting rank of the survivors.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi355rec {

constexpr uint32_t ZERO_KEY = 0x80000000u;
constexpr int AUX_WORDS = 8192;  // 32 KiB: 256 bins x 32 bank replicas, later re-used as the candidate buffer
constexpr int MAX_TOPK = 4096;   // AUX_WORDS * 4 B / 8 B per candidate

// The selection routines below work on LDS arrays, but block_topk_emit is big enough to stay an out-of-line function, and inside an
// out-of-line function a `float *` is a generic pointer: every access to the accumulator, the histogram and the scratch words was a
// flat_load / flat_store / flat_atomic (117 of them, found in round 6) -- the vector-memory path to LDS, several times the latency of
// ds_read / ds_write and tied to the wavefront's outstanding global loads.  The low 32 bits of a generic address inside the shared
// aperture are the LDS offset: the routines take LDS-typed pointers, the public entry points cast once.
#define MI355REC_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ MI355REC_LDS T *lds_of(T *q) { return (MI355REC_LDS T *)(uintptr_t)(unsigned)(unsigned long long)q; }
typedef float lds_f4 __attribute__((ext_vector_type(4)));      // (HIP's float4 class has no constructor from an address-space-qualified lvalue)
__device__ __forceinline__ uint32_t lds_add(MI355REC_LDS uint32_t *q, uint32_t v) {
    return __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

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
    uint32_t digit, want, bin_count, out_count;
};

// Block-wide radix select: key of the `want`-th largest element (1-based) of
//   { kf(j).key : j < n, kf(j).active }  U  { virt_key repeated virt_cnt times }.
// [key_lo, key_hi] must contain every key that can belong to the top `want`, and every active key outside it must lie
// BELOW key_lo (such keys are simply not counted).  The bits the two bounds share are skipped, so the 8-bit digit
// windows start where the keys actually differ (similarity values of one column share sign and leading exponent bits:
// a window over bits 31..24 would put nearly every cell in two or three bins).
// Early exit: as soon as (elements above the current bin) + (elements in it) <= cap, the function returns true with
// T = the lowest key of that bin: { key >= T } is a superset of the answer of at most cap elements, which the caller
// sorts.  Otherwise all bits are resolved and it returns false with T = the exact key, need_eq = how many elements
// equal to T belong to the top `want`, eq_total = how many elements equal T (virtual ones included).
// `vals` (LDS, readable up to the next multiple of 4) is scanned four cells per thread and step; kf(j, v, key) maps
// cell j with value v to its key and says whether it takes part.
template <int THREADS, class KeyFn>
__device__ bool block_select(KeyFn kf, MI355REC_LDS const float *vals, int n, uint32_t want, uint32_t virt_key, uint32_t virt_cnt,
                             MI355REC_LDS uint32_t *hist, MI355REC_LDS SelectScratch &sc, uint32_t key_lo, uint32_t key_hi, uint32_t cap, uint32_t &T,
                             uint32_t &need_eq, uint32_t &eq_total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t want0 = want;
    int remaining = max(1, 32 - (int)__clz(key_lo ^ key_hi));          // __clz(0) == 32
    uint32_t prefix = (uint32_t)((uint64_t)key_hi >> remaining);
    for (int w = tid; w < AUX_WORDS; w += THREADS) hist[w] = 0;
    __syncthreads();
    while (remaining > 0) {
        const int width = min(8, remaining), shift = remaining - width;
        const uint32_t mask = (1u << width) - 1u;
        for (int w = tid; w < (n + 3) / 4; w += THREADS) {
            const lds_f4 q = ((MI355REC_LDS const lds_f4 *)vals)[w];
            const float vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * w + e;
                uint32_t key;
                if (j < n && kf(j, vv[e], key) && (uint32_t)((uint64_t)key >> remaining) == prefix)
                    lds_add(&hist[((key >> shift) & mask) * 32 + (lane & 31)], 1u);
            }
        }
        __syncthreads();
        uint32_t cnt = 0, suffix = 0;
        if (tid < 256) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const int at = tid * 32 + ((r + tid) & 31);
                cnt += hist[at];
                hist[at] = 0;                                           // ready for the next pass
            }
            if (virt_cnt && (uint32_t)((uint64_t)virt_key >> remaining) == prefix && ((virt_key >> shift) & mask) == (uint32_t)tid)
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
        prefix = (prefix << width) | sc.digit;
        want = sc.want;
        eq_total = sc.bin_count;
        remaining = shift;
        __syncthreads();
        if (remaining > 0 && (want0 - want) + eq_total <= cap) {
            T = prefix << remaining;
            return true;
        }
    }
    T = prefix;
    need_eq = want;
    return false;
}

template <int THREADS>
__device__ void bitonic_sort_desc(MI355REC_LDS uint64_t *a, int P) {
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


// Ranks the `ncand` candidates (key << 32 | ~index) in LDS and writes those that fall inside the top K of [positives, the
// `nzero` competing zeros, negatives] straight to their output slot (value-descending, ties towards the lower index), then pads
// the tail with (-1, 0).  Counting rank (every candidate counts the candidates above it: keys carry the index, so ranks are a
// permutation) up to 1024 candidates, a bitonic sort above that.  sc.out_count must be 0 on entry (barrier in between).
template <int THREADS>
__device__ void block_rank_emit_lds(MI355REC_LDS uint64_t *cand, int ncand, int topK, uint32_t K, uint32_t nzero, MI355REC_LDS SelectScratch &sc, int *out_idx,
                                    float *out_val, int idx_offset = 0, const int *idx_map = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    // rank r (0 = largest) is emitted when it falls inside the top K of [positives, the competing zeros, negatives]
    auto emit = [&](uint64_t e, int r) {
        const uint32_t key = (uint32_t)(e >> 32);
        const bool ok = (uint32_t)r + (key < ZERO_KEY ? nzero : 0u) < K;
        if (ok) {
            int idx = (int)(~(uint32_t)(e & 0xFFFFFFFFull));
            idx = idx_map ? idx_map[idx] : idx + idx_offset;
            out_idx[r] = idx;
            if (out_val) out_val[r] = key_float(key);
        }
        return ok;
    };
    uint32_t emitted = 0;
    if (ncand <= 1024) {
        // counting rank: G lanes share one candidate (G a power of two <= 64, G * ncand <= THREADS when possible)
        int G = 1;
        while (G < 64 && 2 * G * ncand <= THREADS) G <<= 1;
        const int per_round = THREADS / G, part = tid & (G - 1);
        for (int c0 = 0; c0 < ncand; c0 += per_round) {
            const int c = c0 + tid / G;
            const bool live = c < ncand;
            const uint64_t mine = live ? cand[c] : 0ull;
            int r = 0;
            if (live) {
#pragma unroll 8
                for (int i = part; i < ncand; i += G) r += cand[i] > mine;
            }
            for (int off = 1; off < G; off <<= 1) r += __shfl_xor(r, off);
            if (live && part == 0) emitted += emit(mine, r);
        }
    } else {
        int P = 2048;
        while (P < ncand) P <<= 1;
        for (int t = ncand + tid; t < P; t += THREADS) cand[t] = 0ull;
        __syncthreads();
        bitonic_sort_desc<THREADS>(cand, P);
        for (int t = tid; t < min(ncand, topK); t += THREADS) emitted += emit(cand[t], t);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) emitted += __shfl_down(emitted, off);
    if (lane == 0 && emitted) lds_add(&sc.out_count, emitted);
    __syncthreads();
    for (int t = (int)sc.out_count + tid; t < topK; t += THREADS) {
        out_idx[t] = -1;
        if (out_val) out_val[t] = 0.f;
    }
    __syncthreads();   // sc and aux may be re-used by the caller's next column
}
template <int THREADS>
__device__ __forceinline__ void block_rank_emit(uint64_t *cand, int ncand, int topK, uint32_t K, uint32_t nzero, SelectScratch &sc, int *out_idx,
                                                float *out_val, int idx_offset = 0, const int *idx_map = nullptr) {
    block_rank_emit_lds<THREADS>(lds_of(cand), ncand, topK, K, nzero, *lds_of(&sc), out_idx, out_val, idx_offset, idx_map);
}

// The 16 leading bits of the K-th largest (1-based) of the THREADS keys the threads of the block hold, one each: returns P such that
// at least K keys are >= P << 16 and fewer than K are >= (P + 1) << 16.  Two 8-bit passes over a 256-bin histogram `hist` (LDS,
// all zero on entry and on exit).  Keys of one column's maxima share sign and most exponent bits, so the first pass would send
// nearly every lane of a wavefront to the same bin: the lanes of up to three distinct digits are counted with one atomic per
// digit (ballot), whoever is left adds for itself.
// KEYS = 2: every thread holds two keys (the K-th largest of the 2 x THREADS keys is looked for).
template <int THREADS, int KEYS = 1>
__device__ uint32_t block_kth_largest_prefix16(uint32_t key, uint32_t K, uint32_t *hist, SelectScratch &sc, uint32_t key_b = 0u) {
    static_assert(THREADS >= 256, "the bins are scanned by the first 256 threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t prefix = 0, want = K;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int which = 0; which < KEYS; ++which) {
            const uint32_t k = which ? key_b : key;
            bool pending = pass == 0 || (k >> 24) == prefix;
            const uint32_t digit = (k >> shift) & 255u;
            for (int it = 0; it < 3; ++it) {
                const unsigned long long todo = __ballot(pending);
                if (!todo) break;
                const int first = __ffsll((long long)todo) - 1;
                const uint32_t d = (uint32_t)__shfl((int)digit, first);
                const unsigned long long same = __ballot(pending && digit == d);
                if (lane == first) atomicAdd(&hist[d], (uint32_t)__popcll(same));
                pending = pending && digit != d;
            }
            if (pending) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        uint32_t cnt = 0, suffix = 0;
        if (tid < 256) {
            cnt = hist[tid];
            hist[tid] = 0;                                              // ready for the next pass / the next column
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
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | sc.digit;
        want = sc.want;
    }
    __syncthreads();     // sc is rewritten by the caller's next selection
    return prefix;
}

// Suffix sums inside a wavefront without LDS traffic: lane l gets v[l] + v[l + 1] + ... + v[63].  Four DPP additions inside the rows
// of 16 lanes (row_shl:n reads lane l + n of the same row, lanes past the row's end contribute 0), then the totals of the rows above
// through v_readlane.  (__shfl_down goes through the LDS crossbar: ~100 cycles per step, six steps.)
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t row_suffix_sum(uint32_t v) {
    v += dpp_u32<0x101>(v);   // row_shl:1
    v += dpp_u32<0x102>(v);   // row_shl:2
    v += dpp_u32<0x104>(v);   // row_shl:4
    v += dpp_u32<0x108>(v);   // row_shl:8
    return v;
}
__device__ __forceinline__ uint32_t wave_suffix_sum(uint32_t v) {
    const uint32_t s = row_suffix_sum(v);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)s, 16), t2 = (uint32_t)__builtin_amdgcn_readlane((int)s, 32),
                   t3 = (uint32_t)__builtin_amdgcn_readlane((int)s, 48);
    const int row = (threadIdx.x & 63) >> 4;
    return s + (row == 0 ? t1 + t2 + t3 : (row == 1 ? t2 + t3 : (row == 2 ? t3 : 0u)));
}

// A lower bound on the K-th largest (1-based) of the THREADS keys the threads of the block hold, one each, all >= ZERO_KEY: the
// return value P (a 16-bit prefix like block_kth_largest_prefix16's) is the lower edge of the 12-bit bin -- 8 exponent bits and the
// 4 leading mantissa bits of a non-negative float -- that holds the K-th largest key, so at least K keys are >= P << 16; P ==
// ZERO_KEY >> 16 when fewer than K keys are above the zero bin.  ONE pass over a 4 096-bin histogram `hist` (LDS, 4 112 words, the first 4 096 all zero
// on entry; left dirty) and three barriers: every wavefront sums its 4096 / WAVES bins (suffix sums by DPP) and publishes its total;
// from the totals every wavefront finds -- for itself, no further exchange -- the wavefront whose bins hold the K-th key, re-reads
// those bins and locates the bin.  block_kth_largest_prefix16 spends 6 000 cycles per call on its seven barriers, ballot rounds and
// LDS-crossbar shuffles (scripts/micro/kth_select.hip); the bound here is up to 1/16 below the exact key instead of 2^-7 -- a caller
// that filters cells with it lets a few more through (114 instead of 102 per column of the ML-20M shape at K = 100).
template <int THREADS>
__device__ uint32_t block_kth_largest_bin12(uint32_t key, uint32_t K, uint32_t *hist, SelectScratch &sc) {
    static_assert(THREADS >= 256 && THREADS <= 1024 && 4096 % THREADS == 0, "one lane sums 4096 / THREADS bins");
    constexpr int WAVES = THREADS / 64, BPL = 4096 / THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t bin = (key >> 19) & 0xFFFu;
    const unsigned long long zeros = __ballot(bin == 0u);           // threads without a positive cell: one atomic for all of them
    if (bin != 0u) atomicAdd(&hist[bin], 1u);
    else if (lane == __ffsll((long long)zeros) - 1) atomicAdd(&hist[0], (uint32_t)__popcll(zeros));
    __syncthreads();
    auto lane_bins = [&](int w, uint32_t (&c)[BPL]) {          // the BPL bins of this lane in wavefront w's stretch, and their sum
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < BPL; ++i) {
            c[i] = hist[(w * 64 + lane) * BPL + i];
            sum += c[i];
        }
        return sum;
    };
    uint32_t c[BPL];
    const uint32_t mine = lane_bins(wave, c);
    const uint32_t total = wave_suffix_sum(mine);           // lane 0: all keys in this wavefront's bins
    if (lane == 0) hist[4096 + wave] = total;          // (the words behind the bins: plain stores, nothing assumed about them)
    __syncthreads();
    // which wavefront's stretch holds the K-th largest key (stretches of higher wavefronts hold larger keys)
    const uint32_t tot = lane < WAVES ? hist[4096 + lane] : 0u;
    const uint32_t tsuf = row_suffix_sum(tot);              // (WAVES <= 16: one row)
    const unsigned long long hit_w = __ballot(lane < WAVES && tsuf >= K && tsuf - tot < K);      // exactly one: THREADS >= K keys in all
    const int W = __ffsll((long long)hit_w) - 1;
    const uint32_t above_w = (uint32_t)__builtin_amdgcn_readlane((int)(tsuf - tot), W);
    const uint32_t its = lane_bins(W, c);
    const uint32_t suf = wave_suffix_sum(its) + above_w;    // keys in the bins of this lane and above
    const unsigned long long hit_l = __ballot(suf >= K && suf - its < K);
    const int L = __ffsll((long long)hit_l) - 1;
    uint32_t above = suf - its;
    int b = BPL - 1;
    for (; b > 0; --b) {
        if (above + c[b] >= K) break;
        above += c[b];
    }
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((W * 64 + lane) * BPL + b, L);
    __syncthreads();     // the caller re-uses the bins (candidate list) and sc
    return 0x8000u | (d << 3);
}

// Which cells take part in a top-K.
//   TOPK_ZEROS_COMPETE : the K largest cells of the whole array are taken, zeros included, and zeros are then dropped
//                        (Compute_Similarity_Cython.pyx:523-555, Triangular_Matrix.get_scipy_csr :1384-1404)
//   TOPK_NONZERO       : only non-zero cells compete (similarityMatrixTopK, Base/Recommender_utils.py:100-104)
//   TOPK_FINITE        : every cell except -inf competes and nothing is dropped (BaseRecommender.recommend :182-205,
//                        where -inf marks excluded items)
enum { TOPK_ZEROS_COMPETE = 0, TOPK_NONZERO = 1, TOPK_FINITE = 2 };

// Top-K of the n floats in LDS array `acc` (K = topK <= MAX_TOPK).
//   npos / nneg     : TOPK_ZEROS_COMPETE / TOPK_NONZERO: number of strictly positive / negative cells;
//                     TOPK_FINITE: npos = number of cells > -inf, nneg = 0          (block-uniform, counted by the caller)
//   *ncand          : shared counter, must be 0 on entry
// Output: out_idx / out_val [topK] (out_val may be null), value-descending, ties broken towards the lower index,
// (-1, 0) padded.
//   idx_offset      : added to every emitted index (tiled callers: base of the tile)
//   idx_map         : if non-null the emitted index is idx_map[local index] (merge of per-tile candidates)
//   zero_count      : TOPK_ZEROS_COMPETE only: number of competing zeros; < 0 = n - npos - nneg
//   key_lo / key_hi : float_key range of the cells that can reach the top K (see block_select); for
//                     TOPK_ZEROS_COMPETE it is the range of the POSITIVE cells and is used when npos >= topK.
// Method: radix select with early exit to a superset of <= cap candidates, which are ranked by counting (every
// candidate counts the candidates above it: keys carry the index, so ranks are a permutation) -- or by a bitonic sort
// when there are more than 1024 of them -- and written straight to their output slot.
template <int THREADS>
__device__ void block_topk_emit(const float *acc, int n, int topK, uint32_t npos, uint32_t nneg, int mode,
                                uint32_t *aux, SelectScratch &sc, uint32_t *ncand_shared, int *out_idx, float *out_val,
                                int idx_offset = 0, const int *idx_map = nullptr, long long zero_count = -1,
                                uint32_t key_lo = 0u, uint32_t key_hi = 0xFFFFFFFFu) {
    const int tid = threadIdx.x, lane = tid & 63;
    MI355REC_LDS const float *const acc_l = lds_of(acc);
    MI355REC_LDS uint32_t *const aux_l = lds_of(aux), *const ncand_l = lds_of(ncand_shared);
    MI355REC_LDS SelectScratch &sc_l = *lds_of(&sc);
    const bool zeros_compete = mode == TOPK_ZEROS_COMPETE;
    const uint32_t nzero = zeros_compete ? (zero_count >= 0 ? (uint32_t)zero_count : (uint32_t)n - npos - nneg) : 0u;
    uint32_t K = (uint32_t)topK;
    if (!zeros_compete) K = min(K, npos + nneg);
    if (zeros_compete && npos < K) {   // zeros / negatives take part: full key range
        key_lo = 0u;
        key_hi = 0xFFFFFFFFu;
    }
    const uint32_t cap = (uint32_t)min(AUX_WORDS / 2, max(256, 2 * topK));
    uint32_t T = ZERO_KEY, need_eq = 0, eq_total = 0;
    if (tid == 0) sc_l.out_count = 0;
    // all positives fit and no negative can displace a zero: nothing to select
    bool take_all = zeros_compete ? (npos <= K && (nneg == 0 || npos + nzero >= K)) : (npos + nneg <= K);
    if (!zeros_compete && take_all) T = 0u;                      // every candidate key is > 0
    auto candidate = [&](float v) { return mode == TOPK_FINITE ? v > -INFINITY : v != 0.f; };
    auto value_key = [&](int, float v, uint32_t &key) {
        key = float_key(v);
        return candidate(v);
    };
    bool superset = false;
    if (!take_all && zeros_compete && npos >= K && npos <= cap) {
        superset = true;                                         // few enough positives: rank them all
        T = max(key_lo, ZERO_KEY + 1u);                          // positives only
    } else if (!take_all) {
        superset = block_select<THREADS>(value_key, acc_l, n, K, ZERO_KEY, nzero, aux_l, sc_l, key_lo, key_hi, cap, T, need_eq, eq_total);
        if (!superset && zeros_compete && T == ZERO_KEY) need_eq = 0;          // zeros are never emitted
    }
    uint32_t T2 = 0;  // tie-break on the index when more cells equal T than fit: lowest index wins
    const bool partial_ties = !superset && need_eq > 0 && need_eq < eq_total;
    if (partial_ties) {
        uint32_t dummy_need, dummy_tot;
        auto index_key = [&](int j, float v, uint32_t &key) {
            key = ~(uint32_t)j;
            return candidate(v) && float_key(v) == T;
        };
        block_select<THREADS>(index_key, acc_l, n, need_eq, 0u, 0u, aux_l, sc_l, 0u, 0xFFFFFFFFu, 0u, T2, dummy_need, dummy_tot);
    }
    __syncthreads();
    // ---- candidates -> aux as (key << 32 | ~index); slots are handed out per wavefront (one LDS atomic per wave) ----
    MI355REC_LDS uint64_t *cand = (MI355REC_LDS uint64_t *)aux_l;
    constexpr int CAND_MAX = AUX_WORDS / 2;
    for (int w0 = 0; w0 < (n + 3) / 4; w0 += THREADS) {
        const int w = w0 + tid;
        float vv[4] = {0.f, 0.f, 0.f, 0.f};
        if (4 * w < n) {
            const lds_f4 q = ((MI355REC_LDS const lds_f4 *)acc_l)[w];
            vv[0] = q.x; vv[1] = q.y; vv[2] = q.z; vv[3] = q.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * w + e;
            const float v = vv[e];
            const uint32_t key = float_key(v);
            const bool take = j < n && candidate(v) &&
                              (superset ? key >= T : (key > T || (need_eq > 0 && key == T && (!partial_ties || ~(uint32_t)j >= T2))));
            const unsigned long long m = __ballot(take);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if (lane == leader) base = lds_add(ncand_l, (uint32_t)__popcll(m));
                base = __shfl(base, leader);
                if (take) {
                    const uint32_t slot_c = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (slot_c < (uint32_t)CAND_MAX) cand[slot_c] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)j);
                }
            }
        }
    }
    __syncthreads();
    const int ncand = min((int)*ncand_l, CAND_MAX);
    block_rank_emit_lds<THREADS>(cand, ncand, topK, K, zeros_compete ? nzero : 0u, sc_l, out_idx, out_val, idx_offset, idx_map);
}

}  // namespace mi355rec
