// sim.hip -- Compute_Similarity on MI355X (gfx950).
//
// Replaces Base/Similarity/Cython/Compute_Similarity_Cython.pyx (reference): __init__ :72-213 (pre-processing,
// column norms, CSR+CSC views), computeItemSimilarities :325-406 (the hot loop), compute_similarity :411-607
// (normalisation :473-504, per-column top-K :523-562).
//
// Also serves Compute_Similarity_Euclidean.py (euclidean_cell below), the boolean-transpose products of P3alpha / RP3beta
// (unit_column_side) and the Gram step of EASE_R (dense output).
//
// Design: one persistent workgroup per CU pulls work items (columns, or parts of heavy
// columns) from a cost-ordered queue; the per-column accumulator `this_item_weights` lives in LDS (uint32 counts for
// all-ones data, exact int32 sums for quantised ratings, int64 fixed-point or float64 sums otherwise), the co-occurrence products are accumulated with LDS atomics from a padded
// uint16 profile stream, normalised in place and reduced to the top-K by an in-LDS radix select with early exit
// (bank-replicated histograms) and a counting rank of the survivors.  The URM is read through L2 / Infinity Cache;
// nothing but the K results per column (and the partial accumulators of split columns) is written to HBM.
#include "common.h"
#include "topk.cuh"

#include <rocprim/rocprim.hpp>

#include <chrono>

#include <algorithm>
#include <memory>
#include <type_traits>
#include <numeric>

namespace mi355rec {
namespace {

// chunks a lane group keeps in flight in the counts instance (round 5: 6 and 8 measured at ML-20M shape: 4.19 / 4.27 ms against 3.99-4.15:
// the stream is not what the accumulation waits for)
#ifndef SIM_DEPTH_UNIT
#define SIM_DEPTH_UNIT 4
#endif
constexpr int MAX_TILE = 32256;      // uint32 count cells of the LDS accumulator: 4 B * (32256 + 4) + 32 KiB selection scratch + statics <= 160 KiB
constexpr int MAX_TILE_F64 = 16128;  // float64 cells (real-valued data): 8 B * (16128 + 4) + 32 KiB
constexpr int NORM_PAD = 1024 + 4;     // zeros behind the norm arrays: the threshold-first selection reads whole rounds of 1024 cells
constexpr int F64_CELLS_PER_THREAD = 16;   // >= MAX_TILE_F64 / 1024 (and the 512-thread launches have <= 5116 cells)

struct SimParams {
    int n_rows, n_cols, n_cols_pad;    // n_cols_pad: neighbour cells of the LDS accumulator (tile width, multiple of 4)
    int acc_cells;                     // n_cols_pad + 4 spare cells that absorb the padding entries of the profiles
    int acc_words;                     // 32-bit words of the accumulator: acc_cells (uint32 counts) or 2 * acc_cells (float64)
    int topK;
    int kind, normalize, unit_col;
    int avg_row, euclid_mode;          // MI355REC_SIM_EUCLIDEAN
    float shrink, tversky_alpha, tversky_beta;
    const int *csr_ptr;
    // Profile stream: every (row, accumulator tile) segment of the CSR matrix, padded to a multiple of 8 entries so
    // that a 16-byte chunk is either entirely inside a segment or entirely outside (no per-entry bounds checks in the
    // hot loop).  Ids are uint16 relative to the tile base; padding entries carry the id of a spare cell and value 0.
    const int *seg_ptr;                // [n_rows * n_tiles + 1], multiples of 8
    const unsigned short *seg_idx16;
    const float *seg_val;
    const short *seg_val16;            // ACC_INT32: the stored values times 2^s as 16-bit integers (same entry order as seg_idx16); seg_val is absent then
    int tile_w, n_tiles;               // accumulator tile width and count (1 when n_cols fits the LDS)
    int *cand_idx;                     // n_tiles > 1: per-workgroup scratch [n_tiles * topK] of per-tile candidates
    float *cand_val;
    const int *csc_ptr, *csc_idx;
    const float *csc_val;
    // Walk lists (built by the constructor, build_walk): per column, what the accumulation walks -- one entry per SLICE of a user's
    // profile segment (at most WALK_SLICE chunks of 8 entries), longest slices first.  All-ones data: walk4 = the slice's number, whose
    // {first entry, end entry} in the profile stream are walk_tab[number] (a 4-byte entry: this list IS the column view of all-ones data,
    // sorted once); valued data: walk16 = {first, end, bits of the column-side value times the row weight, 0}.
    // With accumulator tiles (n_tiles > 1) an entry is a whole user: the row (walk4 / walk16.x), whose per-tile segment bounds come
    // from seg_ptr.
    const int *walk4;
    const uint2 *walk_tab;
    const uint4 *walk16;
    const float *row_w;
    const float *norm, *norm_alpha, *norm_1ma;
    const int4 *items;  // work items of this call, most expensive first: {column, part, n_parts, first part slot}
    const int2 *item_range;   // per work item: the column's [begin, end) in the CSC arrays (saves a dependent round trip per column)
    int n_items, start_col;
    const int *out_slot;    // interleaved parts: output row of every column of the call (NULL: column - start_col)
    float int_scale, int_inv;   // ACC_INT32: 4^s (both factors of a product carry 2^s) and its inverse
    float int_half;             // ACC_INT32: 2^s
    double fixed_scale;     // real-valued data: > 0 = the accumulator holds int64 fixed-point sums, products scaled by this power of two
    double fixed_inv;       //                   (1 / fixed_scale); 0 = float64 sums
    uint32_t *part_buf;     // [part slots][n_cols_pad] partial accumulators of split columns
    unsigned *part_count;   // arrival counters, indexed by the first part slot of a split column
    unsigned long long *phase_ticks;   // diagnostics (MI355REC_SIM_PHASES=1): 100 MHz ticks per phase, summed over workgroups
    int fast_topk;          // 1: threshold-first selection (fast_column_topk) where it applies; 0 (MI355REC_SIM_FAST_TOPK=0): always the full normalise + radix select
    // packed-counts launch + the 32-bit launch behind it (run_columns_lds): the second launch's work list is its own items followed by
    // the columns the packed kernel hands over; *retry_count = its length (read once at kernel start when n_items_dev is set)
    int *retry_count;
    int4 *retry_items;
    int2 *retry_ranges;
    const int *n_items_dev;
    unsigned long long *fast_stats;    // [0] columns finished by the fast path, [1] their candidates, [2] columns that fell back
    unsigned *queue;
    int *out_idx;
    float *out_val;
    float *out_dense;  // [n_local][n_cols] when topK == 0
};

// Denominators of compute_similarity (.pyx:473-504); the +1e-6 is the reference's.  norm_c / norm_j are the column
// norms of the two items (asymmetric cosine: norm^(2 alpha) of c and norm^(2 (1 - alpha)) of j).
__device__ __forceinline__ float normalise(const SimParams &p, float v, float norm_c, float norm_j) {
    if (p.normalize) return v / (norm_c * norm_j + p.shrink + 1e-6f);
    if (p.kind == MI355REC_SIM_JACCARD) return v / (norm_c + norm_j - v + p.shrink + 1e-6f);
    if (p.kind == MI355REC_SIM_DICE) return v / (norm_c + norm_j + p.shrink + 1e-6f);
    if (p.kind == MI355REC_SIM_TVERSKY)
        return v / (v + (norm_c - v) * p.tversky_alpha + (norm_j - v) * p.tversky_beta + p.shrink + 1e-6f);
    if (p.shrink != 0.f) return v / p.shrink;
    return v;
}

// Compute_Similarity_Euclidean.compute_similarity (Euclidean.py:167-203), one cell: the reference works in float32
// NumPy arithmetic (the dtype of the URM), one operation per statement -- restated with explicitly rounded float32
// operations so that no multiply-add is contracted.  sq_* = sum of squares of the column, rt_* = its square root.
// Deviation: a squared distance that rounds below zero is clamped to 0 (the reference takes sqrt of it and emits nan).
// row_weights (:62-72): the dot product is the weighted one (the accumulation multiplies every user's contribution by its weight,
// = dataMatrix_weighted.T.dot(item_data), :153), and the distance VECTOR over the columns is multiplied element by element by the
// weights of the ROWS (:174-175) -- defined for square inputs only, where column j meets row j's weight `w_j`.
__device__ __forceinline__ float euclidean_cell(const SimParams &p, float dot, float sq_c, float sq_j, float rt_c, float rt_j,
                                                float w_j = 1.f, bool weighted = false) {
    float d2 = __fsub_rn(__fadd_rn(sq_j, sq_c), __fmul_rn(2.f, dot));          // (a-b)^2 = a^2 + b^2 - 2ab   (:167-172)
    if (weighted) d2 = __fmul_rn(d2, w_j);                                     // :174-175
    if (p.normalize) d2 = __fdiv_rn(d2, __fmul_rn(rt_c, rt_j));                // :178-179
    if (p.avg_row) d2 = __fdiv_rn(d2, (float)p.n_rows);                        // :181-182
    const float d = __fsqrt_rn(fmaxf(d2, 0.f));                                // :184
    float f = d;                                                               // "lin" :189-190
    if (p.euclid_mode == MI355REC_EUCLID_EXP) f = expf(d);                     // :186-187
    else if (p.euclid_mode == MI355REC_EUCLID_LOG) f = logf(__fadd_rn(d, 1.f));  // :192-193
    return __fdiv_rn(1.f, __fadd_rn(__fadd_rn(f, p.shrink), 1e-9f));
}

// The denominator of `normalise` in ONE form for every mode, d = a * norm_j + (c * v + b), evaluated with two fused operations: used only
// to ORDER cells (the threshold-first selection); every value that is emitted comes from `normalise` itself.
//   normalize (cosine, asymmetric ...)   a = norm_c   b = shrink + 1e-6             c = 0
//   jaccard                              a = 1        b = norm_c + shrink + 1e-6    c = -1
//   dice                                 a = 1        b = norm_c + shrink + 1e-6    c = 0
//   tversky                              a = beta     b = alpha norm_c + shrink + 1e-6   c = 1 - alpha - beta
//   shrink only                          a = 0        b = shrink                    c = 0
//   none                                 a = 0        b = 1                         c = 0      (v * rcp(1) = v)
struct DenomForm {
    float a, b, c;
};
__device__ __forceinline__ DenomForm denominator_form(const SimParams &p, float norm_c) {
    const float s6 = p.shrink + 1e-6f;
    if (p.normalize) return {norm_c, s6, 0.f};
    if (p.kind == MI355REC_SIM_JACCARD) return {1.f, norm_c + s6, -1.f};
    if (p.kind == MI355REC_SIM_DICE) return {1.f, norm_c + s6, 0.f};
    if (p.kind == MI355REC_SIM_TVERSKY) return {p.tversky_beta, __builtin_fmaf(p.tversky_alpha, norm_c, s6), 1.f - p.tversky_alpha - p.tversky_beta};
    if (p.shrink != 0.f) return {0.f, p.shrink, 0.f};
    return {0.f, 1.f, 0.f};
}
__device__ __forceinline__ float approx_denominator(const DenomForm &f, float v, float norm_j) {
    return __builtin_fmaf(f.a, norm_j, __builtin_fmaf(f.c, v, f.b));
}

// float64 -> int64 by the "magic number" addition: for |x| < 2^51, bits(x + 1.5 * 2^52) - bits(1.5 * 2^52) = round-to-nearest-even(x)
constexpr double FIXED_MAGIC = 6755399441055744.0;
constexpr long long FIXED_MAGIC_BITS = 0x4338000000000000ll;

// THREADS: workgroup size; G: lanes that cooperate on one user profile (sub-wave group);
// MODE: what the accumulator cells hold.
//   ACC_COUNTS  all stored values are 1.0 (implicit / set-based data): uint32 co-occurrence counts, the value arrays are never read;
//   ACC_INT32   every stored value is a small multiple of a power of two (star ratings, half stars): the products, scaled by that
//               power of two squared, are small integers and their sums are EXACT in an int32 cell -- ds_add_u32 at the speed of the
//               counts, and one accumulator tile where 8-byte cells need two (26 744 columns at ML-20M shape);
//   ACC_WIDE    any other real-valued data (or row weights): int64 fixed-point or float64 sums in 8-byte cells.
enum { ACC_COUNTS = 0, ACC_INT32 = 1, ACC_WIDE = 2 };
struct alignas(16) SimShared {
    int4 item;
    int2 range;
    int col, last;
    uint32_t npos, nneg, ncand, kmin, kmax;
    SelectScratch sc;
};
// LDS byte address of the cell whose id is half `HI` of the packed id pair `w` (the accumulator starts at LDS address 0)
template <int HI, int SHIFT>
__device__ __forceinline__ unsigned lds_cell_address(unsigned w) {
    unsigned a;
    if (HI) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a) : "v"((unsigned)SHIFT), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a) : "v"((unsigned)SHIFT), "v"(w));
    return a;
}
typedef __attribute__((address_space(3))) unsigned lds_u32_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
typedef __attribute__((address_space(3))) double lds_f64_t;
__device__ __forceinline__ void lds_add_u32(unsigned byte_address, unsigned v) {
    __hip_atomic_fetch_add((lds_u32_t *)(size_t)byte_address, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add_u64(unsigned byte_address, unsigned long long v) {
    __hip_atomic_fetch_add((lds_u64_t *)(size_t)byte_address, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add_f64(unsigned byte_address, double v) {
    __hip_atomic_fetch_add((lds_f64_t *)(size_t)byte_address, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int THREADS, int G, int MODE>
__global__ __launch_bounds__(THREADS, 4) void sim_column_kernel(const SimParams p) {
    constexpr bool UNIT = MODE == ACC_COUNTS;        // no values
    constexpr bool CELL32 = MODE != ACC_WIDE;        // 4-byte integer cells
    // LDS: [accumulator | selection scratch | the workgroup's few shared scalars].  The kernel has NO static LDS, so the accumulator
    // starts at LDS address 0 and a cell's address is its id times the cell size -- one SDWA shift per entry instead of extract + shift
    // + base (lds_cell_address; checked once below).
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *acc = smem;
    uint32_t *aux = reinterpret_cast<uint32_t *>(smem + p.acc_words);
    SimShared &shared = *reinterpret_cast<SimShared *>(aux + AUX_WORDS);
    SelectScratch &sc = shared.sc;
    int &s_col = shared.col, &s_last = shared.last;
    int4 &s_item = shared.item;
    int2 &s_range = shared.range;
    uint32_t &s_npos = shared.npos, &s_nneg = shared.nneg, &s_ncand = shared.ncand, &s_kmin = shared.kmin, &s_kmax = shared.kmax;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) float *)smem != 0u) __builtin_trap();

    const int tid = threadIdx.x, lane = tid & 63;
    const int gl = tid % G;

    unsigned long long t_prev = p.phase_ticks ? wall_clock64() : 0ull;
    // (diagnostics: [8] earliest start, [9] latest end, [10] sum of the workgroups' own spans, [11] longest single work item, [12] its column)
    const unsigned long long t_start = t_prev;
    if (p.phase_ticks && tid == 0) atomicMin(&p.phase_ticks[8], t_start);
    auto mark = [&](int phase) {
        if (p.phase_ticks && tid == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&p.phase_ticks[phase], now - t_prev);
            t_prev = now;
        }
    };
    // The next work item is pulled while the current one is still in its normalisation / top-K phases: thread 0 issues the queue
    // atomic after the accumulation (A), requests the item's descriptor one phase later (B) and files both in LDS after the
    // top-K (C), where the loop head finds them -- without it every column starts with three dependent round trips (queue ->
    // descriptor -> CSC bounds: 2-3 us of ~20).  None of this state is live during the accumulation (the register peak).
    // (behind a packed-counts launch the list has grown by the columns that kernel handed over: its length is read from the device)
    const int n_items = p.n_items_dev ? *p.n_items_dev : p.n_items;
    int nx_slot = -1;                        // thread 0 only
    auto pull_now = [&]() {                  // thread 0, synchronous: the first item, and after a split column's part that does not finish the column
        const int sl = nx_slot >= 0 ? nx_slot : (int)atomicAdd(p.queue, 1u);
        s_col = sl;
        if (sl < n_items) {
            s_item = p.items[sl];
            s_range = p.item_range[sl];
        }
        nx_slot = -1;
    };
    if (tid == 0) pull_now();
    // (Measured and rejected, round 6: requesting the next column's first walk entries while the current column's survivors are ranked.
    // Every __syncthreads waits for ALL outstanding loads of the wavefront, so the requests were simply waited for at the next barrier
    // of the selection -- its phase grew by what the loop head saved, 3.85 ms against 3.81.)
    for (;;) {
        __syncthreads();
        const int slot = s_col;
        if (slot >= n_items) break;
        const int4 item = s_item;
        const unsigned long long t_item = p.phase_ticks ? wall_clock64() : 0ull;
        auto item_done = [&]() {
            if (p.phase_ticks && tid == 0) {
                const unsigned long long span = wall_clock64() - t_item;
                if (span > atomicMax(&p.phase_ticks[11], span)) p.phase_ticks[12] = (unsigned long long)item.x;
            }
        };
        const int c = item.x;
        const int cbeg = s_range.x, cend = s_range.y;    // the column's walk list
        int4 nx_item = make_int4(0, 0, 0, 0);
        int2 nx_range = make_int2(0, 0);
        auto request_next = [&]() {          // (B) thread 0
            if (nx_slot < n_items) {
                nx_item = p.items[nx_slot];
                nx_range = p.item_range[nx_slot];
            }
        };
        auto file_next = [&]() {             // (C) thread 0
            s_col = nx_slot;
            s_item = nx_item;
            s_range = nx_range;
            nx_slot = -1;
        };
        // (a heavy column split over several workgroups, item.z > 1: this one is part item.y, whose wavefronts take their stripes of
        // the walk list like the wavefronts of any other part -- see the dealing below)
        const size_t out_base = (size_t)(p.out_slot ? p.out_slot[c] : c - p.start_col) * p.topK;
        int *wg_cand_idx = p.cand_idx + (size_t)blockIdx.x * p.n_tiles * p.topK;
        float *wg_cand_val = p.cand_val + (size_t)blockIdx.x * p.n_tiles * p.topK;
        long long total_nonzero = 0;

        // Columns wider than the LDS accumulator are processed in tiles of tile_w neighbour ids: every CSR entry
        // belongs to exactly one (row, tile) segment of the profile stream (ids are stored tile-relative), so the
        // tiles together read each profile once.  n_tiles == 1 is the common case.
        for (int tile = 0; tile < p.n_tiles; ++tile) {
        const int tile_base = tile * p.tile_w;
        const int n_tile = min(p.tile_w, p.n_cols - tile_base);
        if (tid == 0) {
            s_npos = 0;
            s_nneg = 0;
            s_ncand = 0;
            s_kmin = 0xFFFFFFFFu;
            s_kmax = 0u;
        }

        // The column's walk list (slices of user profiles, longest first) is dealt to the wavefronts in stripes: units of GPW consecutive
        // entries -- one per lane group -- go to the NV = WAVES x parts "virtual wavefronts" of the column in serpentine order (every
        // other stripe reversed), so every wavefront of every part sees the same mix of lengths; entry q of virtual wavefront vw is
        //     cbeg + ((q / GPW) * NV + pos) * GPW + q % GPW,     pos = vw or NV - 1 - vw by the parity of the stripe q / GPW.
        // Consecutive entries of the sorted list have (nearly) the same number of chunks: the GPW groups of a wavefront finish their
        // entries of a round together and the wavefronts of a column finish together -- with the CSC's row order and whole profiles
        // one heavy user kept its lane group busy while the others idled (42 of 64 lanes per ds_add at ML-20M shape, scripts/analysis/
        // sim_lane_census.py; 59 with this dealing).
        constexpr int WAVES = THREADS / 64, GPW = 64 / G;
        const int wave = tid >> 6, sub = lane / G;
        const int NV = WAVES * item.z, vw = item.y * WAVES + wave;
        auto entry_in = [&](int first, int nv, int v, int q) {       // position in a column's walk list of the q-th entry of its virtual wavefront v of nv
            const int stripe = q / GPW, pos = (stripe & 1) ? nv - 1 - v : v;
            return first + (stripe * nv + pos) * GPW + (q % GPW);
        };
        auto entry_of = [&](int q) { return entry_in(cbeg, NV, vw, q); };
        // Walk entries (and, with accumulator tiles, the CSR bounds behind them) are the only dependent loads of the stream.  They
        // run two rounds (of 64 entries per wavefront) ahead: entries of round r+2 and bounds of round r+1 are requested while round r
        // streams, and the first round's entries are requested before the accumulator is cleared.
        auto load_entry = [&](int at, int end, int &ex, int &ey, float &cv) {
            if (at < end) {
                if (UNIT) {
                    ex = p.walk4[at];
                    ey = 0;
                    cv = 1.f;
                } else {
                    const uint4 e = p.walk16[at];
                    ex = (int)e.x;
                    ey = (int)e.y;
                    cv = __uint_as_float(e.z);
                }
            } else {
                ex = 0;
                ey = -1;                               // (marks a lane without an entry)
            }
        };
        auto load_user = [&](int q, int &ex, int &ey, float &cv) { load_entry(entry_of(q), cend, ex, ey, cv); };
        auto load_bounds = [&](int ex, int ey, float cv, int &rs, int &re, float &r) {
            r = cv;
            if (p.n_tiles == 1) {
                if (UNIT && ey >= 0) {                 // the slice's bounds: one more (L2-resident) look-up, a round ahead like the tiles' bounds
                    const uint2 e = p.walk_tab[ex];
                    ex = (int)e.x;
                    ey = (int)e.y;
                }
                rs = ex;
                re = ey;
            } else if (ey >= 0) {                      // accumulator tiles: .x is the row
                const int *sp = p.seg_ptr + ((size_t)ex * p.n_tiles + tile);
                rs = sp[0];
                re = sp[1];
            } else {
                rs = 0;
                re = -1;
            }
        };
        int x_first = 0, y_first = -1, x_next = 0, y_next = -1, t_rs = 0, t_re = -1;
        float cv_first = 1.f, cv_next = 1.f, t_r = 0.f;
        load_user(lane, x_first, y_first, cv_first);
        load_user(64 + lane, x_next, y_next, cv_next);

        // ---- clear this_item_weights (.pyx:365-370) ----
        {
            float4 *a4 = reinterpret_cast<float4 *>(acc);
            for (int w = tid; w < p.acc_words / 4; w += THREADS) a4[w] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        load_bounds(x_first, y_first, cv_first, t_rs, t_re, t_r);
        __syncthreads();
        mark(0);

        // ---- computeItemSimilarities (.pyx:376-406): users of column c, then every item of each user ----
        // A wavefront takes 64 of its users per round: lane l puts user l's CSR bounds and weight into a
        // wavefront-private table in the selection scratch.  Its GPW lane groups then walk the table round-robin,
        // streaming each profile segment in 16-byte chunks (8 uint16 column ids per lane).  Segments are padded to whole
        // chunks, so a lane's chunk is valid or not as a whole: the accumulation is 8 x (extract id, ds_add) with
        // nothing else -- on the 16-lane SIMDs of CDNA every wave64 VALU instruction costs 4 issue cycles, and the
        // per-entry bounds checks of an unpadded layout made this loop VALU-issue-bound (3.7 of 5.8 ms at ML-20M shape).
        // The stream is also latency-bound (one workgroup per CU = 16 wavefronts, each load ~1 us away), so every group
        // runs a fetch cursor DEPTH chunks ahead of its consume cursor: DEPTH loads per lane in flight, issued
        // unconditionally (finished groups re-read a hot line) so that the wait counters are static and the consume
        // side only ever waits for the oldest chunk.
        // UNIT data accumulates integer counts with ds_add_u32; real-valued data accumulates float64 products -- like the
        // reference, whose accumulator is a double array.  (Measured on gfx950, random cells, per CU and ns: ds_add_u32 21.6
        // lane-adds, ds_add_u64 13.2, ds_add_f64 7.2, ds_add_f32 0.8 -- the float32 LDS atomic is 27x slower than the integer
        // one and 9x slower than the float64 one.)  Because the 64-bit INTEGER atomic is 1.8x faster than the float64 one, the
        // products are accumulated as int64 fixed point whenever the host found a power-of-two scale that keeps every sum
        // inside 62 bits and every product's rounding below 1e-7 of the smallest normalised result (p.fixed_scale > 0):
        // x * scale is rounded to an integer by adding 1.5 * 2^52 in float64 (one fma) and subtracting that constant's bits;
        // integer sums are exact and independent of the order of the adds.
        // (exact int32 sums: ids and values are one 16-byte chunk each per lane -- three of them in flight)
        constexpr int DEPTH = UNIT ? SIM_DEPTH_UNIT : (MODE == ACC_INT32 ? 3 : 2);
        unsigned *acc_u = reinterpret_cast<unsigned *>(acc);
        double *acc_d = reinterpret_cast<double *>(acc);
        const bool fixed_point = MODE == ACC_WIDE && p.fixed_scale > 0.0;
        // (the id stream through a buffer descriptor: a 32-bit byte offset per load instead of 64-bit address arithmetic; the stream holds
        // fewer than 2^31 entries of 2 bytes -- checked by the constructor -- so the descriptor's 32-bit size covers it)
        const __amdgpu_buffer_rsrc_t idx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.seg_idx16), 0, (int)0xFFFFFFF0u, 0x00020000);
        const float4 *val4 = reinterpret_cast<const float4 *>(p.seg_val);
        const uint4 *val8 = reinterpret_cast<const uint4 *>(p.seg_val16);
        int4 *tab = reinterpret_cast<int4 *>(aux) + wave * 64;       // [64] x {rs, re, weight, -}: one 16-byte read per entry
        // All-ones data, one tile (the headline instance): the same walk with the bookkeeping pared down.  The table holds {first, end}
        // pairs (128 per wavefront, the upper half stays {0, 0}: a group that has run out of entries reads an empty slice and stays where
        // it is); per step a group checks whether its slice is used up, READS ITS NEXT TABLE ENTRY UNCONDITIONALLY (no branch, no nested
        // loop, no count of pending chunks) and applies it after the eight atomics of the oldest chunk, then fetches: 36 instructions
        // per step where the general loop below has 61.  Measured in round 6: the phase takes the same time either way (524 against
        // 518 workgroup-ms at ML-20M shape) -- like prefetch depth, the stream's origin and the atomics' count, the instruction count
        // is not what it waits for; kept because sim_packed_kernel shares the loop and it is the
        // simpler code.
        const bool lean = UNIT && p.n_tiles == 1;
        int2 *tab2 = reinterpret_cast<int2 *>(aux) + wave * 128;
        if (lean) tab2[64 + lane] = make_int2(0, 0);
        for (int base = 0; entry_of(base) < cend; base += 64) {
            const bool have = t_re >= 0;                 // (a prefix of the lanes: entry_of grows with q)
            const int n_here = __popcll(__ballot(have));
            if (lean) tab2[lane] = have ? make_int2(t_rs, t_re) : make_int2(0, 0);
            else if (have) tab[lane] = make_int4(t_rs, t_re, __float_as_int(t_r), 0);
            load_bounds(x_next, y_next, cv_next, t_rs, t_re, t_r);                     // for the next round
            load_user(base + 128 + lane, x_next, y_next, cv_next);                    // for the round after
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lean) {
                const int g8 = 8 * gl;
                int m = sub;
                int f_t, f_re;
                {
                    const int2 e = tab2[m];
                    f_t = e.x;
                    f_re = e.y;
                }
                uint4 ids[DEPTH];
                bool ok[DEPTH];
                auto fetch = [&](int d) {
                    const int at = f_t + g8;
                    ok[d] = at < f_re;
                    ids[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(idx_rsrc, at * 2, 0, 0));
                    f_t += 8 * G;
                };
                auto step = [&](int d) {
                    const bool done = f_t >= f_re;           // (group-uniform: the slice is used up)
                    m = min(m + (done ? GPW : 0), 127);
                    const int2 e = tab2[m];                  // requested before the atomics below, needed after them
                    if (ok[d]) {
                        const unsigned ww[4] = {ids[d].x, ids[d].y, ids[d].z, ids[d].w};
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            lds_add_u32((q & 1) ? lds_cell_address<1, 2>(ww[q >> 1]) : lds_cell_address<0, 2>(ww[q >> 1]), 1u);
                    }
                    f_t = done ? e.x : f_t;
                    f_re = done ? e.y : f_re;
                    fetch(d);
                };
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if (d) {
                        const bool done = f_t >= f_re;
                        m = min(m + (done ? GPW : 0), 127);
                        const int2 e = tab2[m];
                        f_t = done ? e.x : f_t;
                        f_re = done ? e.y : f_re;
                    }
                    fetch(d);
                }
                for (;;) {
                    bool any = false;
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        any |= ok[d];
                        step(d);
                    }
                    if (__ballot(any) == 0ull) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // the table is rewritten by the next round
                __builtin_amdgcn_wave_barrier();
                continue;
            }

            // fetch cursor of this lane group
            int m = sub - GPW, f_t = 0, f_re = 0;
            float f_r = 0.f;
            bool f_have = true;
            auto next_user = [&]() {        // moves the fetch cursor to the group's next non-empty segment
                do {
                    m += GPW;
                    f_have = m < n_here;
                    if (f_have) {
                        const int4 e = tab[m];
                        f_t = e.x;
                        f_re = e.y;
                        f_r = __int_as_float(e.z);
                    }
                } while (f_have && f_t >= f_re);      // empty segments exist only with accumulator tiles
            };
            next_user();
            uint4 ids[DEPTH];
            float4 vlo[DEPTH], vhi[DEPTH];
            int c_t[DEPTH], c_re[DEPTH];      // chunk position of the lane; end of the group's segment (0: no chunk)
            float c_r[DEPTH];
            int pending = 0;
            auto fetch = [&](int d) {
                const int at = f_have ? f_t + 8 * gl : 8 * gl;       // finished groups: a valid, cache-hot address
                c_t[d] = at;
                c_re[d] = f_have ? f_re : 0;
                c_r[d] = f_r;
                ids[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(idx_rsrc, at * 2, 0, 0));
                if (MODE == ACC_INT32) {
                    vlo[d] = __builtin_bit_cast(float4, val8[at >> 3]);       // eight int16 values
                } else if (!UNIT) {
                    vlo[d] = val4[at >> 2];
                    vhi[d] = val4[(at >> 2) + 1];
                }
                if (f_have) {
                    ++pending;
                    f_t += 8 * G;
                    if (f_t >= f_re) next_user();
                }
            };
            // (An interleaved lane <-> entry mapping -- neighbouring lanes on neighbouring profile entries, hoping for
            // neighbouring LDS banks -- was measured 11 % slower than 8 consecutive entries per lane.)
            auto consume = [&](int d) {
                if (c_re[d] > 0) --pending;
                if (c_t[d] < c_re[d]) {
                    const unsigned ww[4] = {ids[d].x, ids[d].y, ids[d].z, ids[d].w};
                    const float vv[8] = {vlo[d].x, vlo[d].y, vlo[d].z, vlo[d].w, vhi[d].x, vhi[d].y, vhi[d].z, vhi[d].w};
                    const double rd = (double)c_r[d];
                    if (MODE == ACC_INT32) {
                        // (column value * 2^s) * (row value * 2^s): two integers of at most 12 bits -- the row side comes from the stream
                        // as int16 (a third of the bytes of ids + float32 values: the stream, not the atomics, was what made this
                        // instance twice as slow as the counts), one 24-bit multiply, an integer add
                        const int ri = __float2int_rn(c_r[d] * p.int_half);
                        const uint4 vq = __builtin_bit_cast(uint4, vlo[d]);
                        const unsigned vw[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned at = (e & 1) ? lds_cell_address<1, 2>(ww[e >> 1]) : lds_cell_address<0, 2>(ww[e >> 1]);
                            const int v = (e & 1) ? (int)vw[e >> 1] >> 16 : (int)(short)(vw[e >> 1] & 0xFFFFu);
                            lds_add_u32(at, (unsigned)__mul24(ri, v));
                        }
                    } else if (MODE == ACC_WIDE && fixed_point) {
                        const double rs = rd * p.fixed_scale;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned at = (e & 1) ? lds_cell_address<1, 3>(ww[e >> 1]) : lds_cell_address<0, 3>(ww[e >> 1]);
                            const double q = __builtin_fma(rs, (double)vv[e], FIXED_MAGIC);
                            lds_add_u64(at, (unsigned long long)(__double_as_longlong(q) - FIXED_MAGIC_BITS));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (UNIT) lds_add_u32((e & 1) ? lds_cell_address<1, 2>(ww[e >> 1]) : lds_cell_address<0, 2>(ww[e >> 1]), 1u);
                            else lds_add_f64((e & 1) ? lds_cell_address<1, 3>(ww[e >> 1]) : lds_cell_address<0, 3>(ww[e >> 1]), rd * (double)vv[e]);
                        }
                    }
                }
            };
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) fetch(d);
            while (pending > 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    consume(d);
                    fetch(d);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // the table is rewritten by the next round
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        mark(1);
        if (tid == 0 && tile == 0) nx_slot = (int)atomicAdd(p.queue, 1u);       // next work item: requested now, looked at later
        if (UNIT && item.w < 0) {
            // A column with 65 536 users or more behind a packed-counts launch: its parts (each fewer users than that) were accumulated
            // there, two 16-bit counts per word, and published; this item -- {column, 0, 1, -(1 + first slot)} with the EMPTY walk list
            // [parts, parts) -- adds them up into 32-bit cells and selects.
            const int n_words = p.n_cols_pad / 2;
            const uint32_t *src = p.part_buf + (size_t)(-(item.w + 1)) * p.n_cols_pad;
            for (int w = tid; w < n_words; w += THREADS) {
                unsigned lo = 0u, hi = 0u;
                for (int q = 0; q < cbeg; ++q) {
                    const unsigned v = src[(size_t)q * p.n_cols_pad + w];
                    lo += v & 0xFFFFu;
                    hi += v >> 16;
                }
                acc_u[2 * w] = lo;
                acc_u[2 * w + 1] = hi;
            }
            __syncthreads();
            mark(2);
        }
        if (item.z > 1) {
            const int pub_words = CELL32 ? p.n_cols_pad : 2 * p.n_cols_pad;
            // Split column: publish this part's accumulator; the workgroup that arrives last adds the parts up (in
            // part order, so the float result does not depend on arrival order) and carries on with the column.
            // Nobody waits for anybody.
            {
                uint4 *dst = reinterpret_cast<uint4 *>(p.part_buf + (size_t)(item.w + item.y) * pub_words);   // spare cells are not published
                const uint4 *src = reinterpret_cast<const uint4 *>(acc);
                for (int w = tid; w < pub_words / 4; w += THREADS) dst[w] = src[w];
            }
            // Every wavefront waits until its own stores have reached the L2; after the barrier ONE thread makes them
            // visible device-wide (agent-scope release: L2 write-back) and counts the arrival; the last arriver
            // acquires (invalidates this CU's L1 / stale L2 lines) on behalf of the whole workgroup.  A fence per
            // thread costs ~50 us per part on gfx950 (16 wavefronts x write-back + invalidate).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                s_last = atomicAdd(&p.part_count[item.w], 1u) == (unsigned)(item.z - 1);
                if (s_last) __threadfence();
            }
            __syncthreads();
            if (!s_last) {
                mark(2);
                if (tid == 0) pull_now();
                continue;
            }
            const uint4 *src = reinterpret_cast<const uint4 *>(p.part_buf + (size_t)item.w * pub_words);
            const size_t stride4 = (size_t)pub_words / 4;
            for (int w = tid; w < pub_words / 4; w += THREADS) {
                uint4 a = src[w];
                for (int q = 1; q < item.z; ++q) {
                    const uint4 b = src[q * stride4 + w];
                    if (CELL32) {
                        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                    } else if (p.fixed_scale > 0.0) {   // two int64 cells
                        const unsigned long long a0 = (((unsigned long long)a.y << 32) | a.x) + (((unsigned long long)b.y << 32) | b.x);
                        const unsigned long long a1 = (((unsigned long long)a.w << 32) | a.z) + (((unsigned long long)b.w << 32) | b.z);
                        a = make_uint4((unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32));
                    } else {   // two float64 cells
                        const double a0 = __hiloint2double((int)a.y, (int)a.x) + __hiloint2double((int)b.y, (int)b.x);
                        const double a1 = __hiloint2double((int)a.w, (int)a.z) + __hiloint2double((int)b.w, (int)b.z);
                        a = make_uint4((unsigned)__double2loint(a0), (unsigned)__double2hiint(a0), (unsigned)__double2loint(a1),
                                       (unsigned)__double2hiint(a1));
                    }
                }
                reinterpret_cast<uint4 *>(acc)[w] = a;
            }
            __syncthreads();
            mark(2);
        }
        // the diagonal was accumulated like any other cell: clear it (the reference never adds to it, .pyx:392)
        if (tid == 0 && c >= tile_base && c < tile_base + n_tile) {
            if (CELL32) acc[c - tile_base] = 0.f;
            else acc_d[c - tile_base] = 0.0;
        }
        if (CELL32 && p.fast_topk) {
            // the histogram of block_kth_largest_prefix16 (256 words; the wavefront tables are dead) and, with the spare cells (they
            // absorbed the padding entries), the zeros the last round of the selection's scans reads behind the tile
            for (int w = tid; w < 1024; w += THREADS) aux[w] = 0u;
            if (tid < 4) acc[p.n_cols_pad + tid] = 0.f;
        }
        __syncthreads();

        // ---- threshold-first top-K (4-byte cells, one tile, topK > 0, a positive denominator) ----
        // The full path below divides every cell (IEEE division: ~10 VALU operations), counts signs and key ranges, and then
        // scans the 26 744 cells of an ML-20M column two or three more times for the radix select: 17.8 of the ~19 us a column costs
        // besides its accumulation.  Only the K winners need their exact value.  So: (A) every thread takes the maximum of
        // v * rcp(denominator) over its own cells (approximate: a few ulp) -- the K-th largest of these THREADS maxima is a lower
        // bound T0 on the K-th largest cell of the column, and a tight one (the winners of a column are spread over the threads);
        // (B) one more scan compares v with Tf * denominator, Tf = T0 (1 - 2^-19): no division, and the margin (32 ulp) covers the
        // rounding of both approximations (<= 4 ulp each), so every cell whose EXACT value reaches the exact K-th largest value
        // passes -- ties included; (C) the survivors (~1.05 K) are divided exactly (`normalise`, the same instructions as below),
        // ranked by (value, lowest index first) and the first K emitted.  The result is identical to the full path's, bit for bit
        // (tests/test_sim_gpu.py::test_fast_topk_equals_full_selection).  Fewer than K positive thread maxima (sparse columns) or
        // more survivors than the candidate buffer holds (4 096: masses of equal values): the full path runs, the accumulator is untouched.
        if (CELL32 && p.fast_topk && !(item.z == 1 && item.w == 1)) {          // (.w == 1: a light column, see the schedule)
            const bool asym = p.normalize && p.kind == MI355REC_SIM_ASYMMETRIC;
            const float norm_c = asym ? p.norm_alpha[c] : p.norm[c];
            const float4 *nj4 = reinterpret_cast<const float4 *>(asym ? p.norm_1ma : p.norm);
            const uint32_t K = (uint32_t)p.topK;
            // Thread t owns the cells t, t + THREADS, t + 2 THREADS, ...: neighbouring ids -- whose values are often neighbours too
            // (ids ordered by popularity or by age) -- sit in different threads, so a run of large cells is a run of large thread maxima.
            // (With four adjacent cells per thread the bound was loose: 262 survivors per column for K = 100.)  Cells and norms are
            // fetched in rounds of THREADS: ds_read_b32 at one address register + a constant offset; the norms with buffer loads (one
            // offset register, the round in the scalar offset, zeros beyond the array) -- nothing per cell is kept between the two
            // scans: 32 norms per thread do not fit next to the kernel's state in the 128 registers of a 1024-thread workgroup (they
            // went to scratch and came back one dependent reload per cell).  The round that straddles the end of the tile reads the
            // spare cells and the first words of the selection scratch: all zero (cleared above; the histogram is zero again when
            // block_kth_largest_prefix16 returns), and a zero cell neither raises a maximum nor passes the bar.
            constexpr int CPT = (MAX_TILE + 1023) / 1024;      // rounds (512-thread tiles are narrower than half of MAX_TILE)
            constexpr int CAND_MAX = AUX_WORDS / 2;            // 8-byte entries: (norm, id) of a survivor, then its (value key, ~id)
            constexpr int BATCH = 16, HALF = 8;
            const float *nj = reinterpret_cast<const float *>(nj4);
            const __amdgpu_buffer_rsrc_t nj_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(nj), 0, n_tile * 4, 0x00020000);
            int tid_o = tid;                                   // (opaque: or the 32 addresses are computed before the persistent loop and parked in scratch)
            asm volatile("" : "+v"(tid_o));
            const int n_rounds = (p.n_cols_pad + THREADS - 1) / THREADS;
            const DenomForm form = denominator_form(p, norm_c);
            auto cell_value = [&](unsigned q) { return UNIT ? (float)q : (float)(int)q * p.int_inv; };
            // one batch of rounds: the norms of BATCH cells are requested together, the cells are read from LDS while they are on their
            // way, then `use(round, value, norm)`.  Rounds behind the tile (a batch is not cut short) and the lanes of the last round
            // that lie behind it read the first spare cell: zero.
            const unsigned cell_at = (unsigned)tid_o * 4u, cell_end = (unsigned)p.n_cols_pad * 4u;      // byte offsets into the accumulator
            auto request_norms = [&](int b, float (&dst)[BATCH]) {
#pragma unroll
                for (int i = 0; i < BATCH; ++i)
                    dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nj_rsrc, tid_o * 4, (b + i) * THREADS * 4, 0));
            };
            // (sixteen norms per request and thread, nothing requested ahead: a scan of an ML-20M column is two L2 round trips instead of
            // the four that eight double-buffered norms made it -- working on eight cells never hid the next request's latency; the same
            // 24 registers: sixteen norms + eight cells)
            auto scan_cells = [&](auto &&use) {
#pragma unroll
                for (int bi = 0; bi < CPT / BATCH; ++bi) {
                    const int b = bi * BATCH;
                    if (b >= n_rounds) break;                              // (block-uniform)
                    float nrm[BATCH];
                    request_norms(b, nrm);
#pragma unroll
                    for (int hf = 0; hf < BATCH / HALF; ++hf) {
                        unsigned cnt[HALF];
#pragma unroll
                        for (int i = 0; i < HALF; ++i)
                            cnt[i] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(acc) +
                                                                         min(cell_at + (unsigned)(b + hf * HALF + i) * (THREADS * 4u), cell_end));
#pragma unroll
                        for (int i = 0; i < HALF; ++i) use(b + hf * HALF + i, cell_value(cnt[i]), nrm[hf * HALF + i]);
                    }
                }
            };
            // (A) thread maxima of the approximate values
            float m = 0.f;
            scan_cells([&](int, float v, float norm_j) { m = fmaxf(m, v * __builtin_amdgcn_rcpf(approx_denominator(form, v, norm_j))); });
            mark(5);
            if (p.phase_ticks) {          // (diagnostics only: the wait for the slowest wavefront of the scan, apart from the selection)
                __syncthreads();
                mark(7);
            }
            // (block_kth_largest_bin12 -- one 12-bit pass, three barriers, 4 060 cycles against 5 960 in scripts/micro/kth_select.hip --
            // was measured here: the phase went from 91 to 63 workgroup-ms, its 12 % more survivors cost 4 of them back, and the
            // un-instrumented kernel was 0.08-0.11 ms SLOWER in both sessions: the two-pass 16-bit prefix stays)
            const uint32_t p16 = block_kth_largest_prefix16<THREADS>(float_key(m), K, aux, sc);
            mark(3);
            bool done = p16 > (ZERO_KEY >> 16);                    // else: fewer than K threads hold a positive cell
            if (done) {
                if (tid == 0 && tile == 0) request_next();
                const float Tf = key_float(p16 << 16) * 0.99999809265136719f;        // 1 - 2^-19
                // (B) cells that can reach the top K -> list of (neighbour norm, cell id)
                uint64_t *cand = reinterpret_cast<uint64_t *>(aux);
                if (tid == 0) sc.out_count = 0;
                scan_cells([&](int round, float v, float norm_j) {
                    // (v > 0 is tested on its own: the rounds of a batch that lie behind the tile read zero CELLS by construction, but their
                    // NORMS rest on the buffer range check covering the scalar offset -- a zero cell must not pass on a stale norm)
                    if (v > 0.f && v >= Tf * approx_denominator(form, v, norm_j)) {
                        const uint32_t at = atomicAdd(&s_ncand, 1u);
                        if (at < (uint32_t)CAND_MAX) cand[at] = ((uint64_t)__float_as_uint(norm_j) << 32) | (uint32_t)(tid_o + round * THREADS);
                    }
                });
                __syncthreads();
                mark(6);
                const uint32_t n_cand = s_ncand;
                if (n_cand > (uint32_t)CAND_MAX || n_cand < K) {         // (n_cand < K cannot happen: at least K cells passed (A)'s bar)
                    __syncthreads();
                    if (tid == 0) s_ncand = 0;
                    if (p.fast_stats && tid == 0) atomicAdd(&p.fast_stats[2], 1ull);
                    if (p.fast_stats && tid == 0 && n_cand > (uint32_t)CAND_MAX) atomicAdd(&p.fast_stats[3], 1ull);
                    done = false;
                } else {
                    if (p.fast_stats && tid == 0) {
                        atomicAdd(&p.fast_stats[0], 1ull);
                        atomicAdd(&p.fast_stats[1], (unsigned long long)n_cand);
                    }
                    // (C) the survivors' exact values (one survivor per thread, in place), rank, emit
                    for (uint32_t t = tid; t < n_cand; t += THREADS) {
                        const uint64_t e = cand[t];
                        const uint32_t j = (uint32_t)e;
                        const float x = normalise(p, cell_value(acc_u[j]), norm_c, __uint_as_float((uint32_t)(e >> 32)));
                        cand[t] = ((uint64_t)float_key(x) << 32) | (uint32_t)(~j);
                    }
                    __syncthreads();
                    block_rank_emit<THREADS>(cand, (int)n_cand, p.topK, K, 0u, sc, p.out_idx + out_base, p.out_val + out_base);
                }
            }
            if (done) {
                if (tid == 0 && tile == 0) file_next();
                __syncthreads();
                mark(4);
                continue;
            }
        }

        // ---- normalisation (.pyx:473-504), in place; count signs for the selection ----
        uint32_t npos = 0, nneg = 0, kmin = 0xFFFFFFFFu, kmax = 0u;   // key range of the positive cells
        {
            const bool asym = p.normalize && p.kind == MI355REC_SIM_ASYMMETRIC;
            const bool euclid = p.kind == MI355REC_SIM_EUCLIDEAN;      // every cell but the diagonal gets a value
            const float norm_c = asym ? p.norm_alpha[c] : p.norm[c];
            const float sq_c = euclid ? p.norm_alpha[c] : 0.f;         // euclidean: norm_alpha holds the sums of squares
            const float *nj = (asym ? p.norm_1ma : p.norm) + tile_base;
            const float *sqj = p.norm_alpha + tile_base;
            auto account = [&](float v) {
                npos += v > 0.f;
                nneg += v < 0.f;
                if (v > 0.f) {
                    const uint32_t key = float_key(v);
                    kmin = min(kmin, key);
                    kmax = max(kmax, key);
                }
            };
            if (CELL32) {
                // counts, or exact integer sums of products scaled by int_scale (a power of four)
                auto cell_value = [&](unsigned q) { return UNIT ? (float)q : (float)(int)q * p.int_inv; };
                const float4 *nj4 = reinterpret_cast<const float4 *>(nj);
                float4 *a4 = reinterpret_cast<float4 *>(acc);
                const int n_quads = p.n_cols_pad / 4;
                // four cells per thread and step (the norm arrays are padded to a multiple of 4; cells beyond n_tile are 0)
                if (euclid) {
                    for (int w = tid; w < n_quads; w += THREADS) {
                        const uint4 qu = reinterpret_cast<const uint4 *>(acc)[w];
                        const float4 n4 = nj4[w];
                        const float4 s4 = reinterpret_cast<const float4 *>(sqj)[w];
                        float vv[4] = {cell_value(qu.x), cell_value(qu.y), cell_value(qu.z), cell_value(qu.w)};
                        const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
                        const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = 4 * w + e;
                            vv[e] = (j < n_tile && tile_base + j != c) ? euclidean_cell(p, vv[e], sq_c, ss[e], norm_c, nn[e]) : 0.f;
                            account(vv[e]);
                        }
                        a4[w] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                } else {
                    // The neighbours' norms come from L2: loaded inside the loop, behind the test for an all-zero quad, every
                    // step paid a full round trip (6.5 of them per column at ML-20M shape = the whole phase); all of a thread's
                    // quads are requested up front instead (8 steps cover MAX_TILE / 4 / 1024; 512-thread tiles are narrower).
                    constexpr int NPF = 8;
                    float4 npf[NPF];
                    const int n_quads_valid = (n_tile + 3) >> 2;          // (the last tile is narrower than the accumulator: the norm arrays end with it)
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int w = tid + i * THREADS;
                        npf[i] = nj4[w < n_quads_valid ? w : 0];
                    }
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int w = tid + i * THREADS;
                        if (w < n_quads) {
                            const uint4 qu = reinterpret_cast<const uint4 *>(acc)[w];
                            if ((qu.x | qu.y | qu.z | qu.w) != 0u) {
                                float vv[4] = {cell_value(qu.x), cell_value(qu.y), cell_value(qu.z), cell_value(qu.w)};
                                const float nn[4] = {npf[i].x, npf[i].y, npf[i].z, npf[i].w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (vv[e] != 0.f) {
                                        vv[e] = normalise(p, vv[e], norm_c, nn[e]);
                                        account(vv[e]);
                                    }
                                }
                                a4[w] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                            }
                        }
                    }
                }
            } else {
                // float64 sums -> normalised float32 values in the first half of the same LDS bytes.  In two batches of cells
                // (half the registers of one batch of 16: the 1024-thread instance sits at its 128-register cap): batch h reads
                // cells [8h T, 8(h+1) T) -- bytes [64h T, 64(h+1) T) -- into registers, barrier, writes float32 to bytes
                // [32h T, 32(h+1) T): batch 0 overwrites only cells it has read itself, batch 1 only cells batch 0 has read.
                // The norms are loaded unconditionally (not behind `v != 0`), so that the 8 loads of a batch are in flight together.
                constexpr int HALF = F64_CELLS_PER_THREAD / 2;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float reg[HALF], njv[HALF], sqv[HALF];
#pragma unroll
                    for (int k = 0; k < HALF; ++k) {
                        const int j = tid + (half * HALF + k) * THREADS;
                        njv[k] = nj[j < n_tile ? j : 0];
                        sqv[k] = euclid ? sqj[j < n_tile ? j : 0] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < HALF; ++k) {
                        const int j = tid + (half * HALF + k) * THREADS;
                        float v = 0.f;
                        if (j < n_tile) {
                            v = p.fixed_scale > 0.0 ? (float)((double)(long long)reinterpret_cast<const unsigned long long *>(acc)[j] * p.fixed_inv)
                                                    : (float)acc_d[j];
                            if (euclid) {
                                if (tile_base + j != c) {
                                    const bool weighted = p.row_w != nullptr;       // (weights always take this accumulator)
                                    v = euclidean_cell(p, v, sq_c, sqv[k], norm_c, njv[k], weighted ? p.row_w[tile_base + j] : 1.f, weighted);
                                } else {
                                    v = 0.f;
                                }
                                account(v);
                            } else if (v != 0.f) {
                                v = normalise(p, v, norm_c, njv[k]);
                                account(v);
                            }
                        }
                        reg[k] = v;
                    }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < HALF; ++k) {
                        const int j = tid + (half * HALF + k) * THREADS;
                        if (j < p.n_cols_pad) acc[j] = reg[k];
                    }
                }
            }
        }
        if (p.topK == 0) {  // dense output (.pyx:507-510)
            if (tid == 0 && tile == 0) {
                request_next();
                file_next();
            }
            __syncthreads();
            float *dst = p.out_dense + (size_t)(p.out_slot ? p.out_slot[c] : c - p.start_col) * p.n_cols + tile_base;
            for (int j = tid; j < n_tile; j += THREADS) dst[j] = acc[j];
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            npos += __shfl_down(npos, off);
            nneg += __shfl_down(nneg, off);
            kmin = min(kmin, (uint32_t)__shfl_down(kmin, off));
            kmax = max(kmax, (uint32_t)__shfl_down(kmax, off));
        }
        if (lane == 0) {
            if (npos) {
                atomicAdd(&s_npos, npos);
                atomicMin(&s_kmin, kmin);
                atomicMax(&s_kmax, kmax);
            }
            if (nneg) atomicAdd(&s_nneg, nneg);
        }
        __syncthreads();
        npos = s_npos;
        nneg = s_nneg;
        total_nonzero += npos + nneg;
        mark(3);
        if (tid == 0 && tile == 0) request_next();                               // its descriptor arrives during the top-K
        if (p.n_tiles == 1) {
            // ---- top-K: the K largest cells of the FULL column (zeros compete, then are dropped), value-descending,
            //      emitted like the COO triples of .pyx:550-562 with -1 padding ----
            block_topk_emit<THREADS>(acc, n_tile, p.topK, npos, nneg, TOPK_ZEROS_COMPETE, aux, sc, &s_ncand,
                                     p.out_idx + out_base, p.out_val + out_base, 0, nullptr, -1, s_kmin, s_kmax);
        } else {
            // the tile's K best non-zero cells go to the workgroup's scratch; zeros are accounted for in the merge
            block_topk_emit<THREADS>(acc, n_tile, p.topK, npos, nneg, TOPK_NONZERO, aux, sc, &s_ncand,
                                     wg_cand_idx + tile * p.topK, wg_cand_val + tile * p.topK, tile_base);
        }
        if (tid == 0 && tile == 0) file_next();
        __syncthreads();
        mark(4);
        }  // tiles

        if (p.n_tiles > 1 && p.topK > 0) {
            // ---- merge of the per-tile candidates: the K largest of the whole column, zeros competing ----
            const int n_m = p.n_tiles * p.topK;
            if (tid == 0) { s_npos = 0; s_nneg = 0; s_ncand = 0; }
            __threadfence_block();
            __syncthreads();
            uint32_t npos = 0, nneg = 0;
            for (int j = tid; j < n_m; j += THREADS) {
                const float v = wg_cand_idx[j] >= 0 ? wg_cand_val[j] : 0.f;
                acc[j] = v;
                npos += v > 0.f;
                nneg += v < 0.f;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                npos += __shfl_down(npos, off);
                nneg += __shfl_down(nneg, off);
            }
            if (lane == 0) {
                if (npos) atomicAdd(&s_npos, npos);
                if (nneg) atomicAdd(&s_nneg, nneg);
            }
            __syncthreads();
            block_topk_emit<THREADS>(acc, n_m, p.topK, s_npos, s_nneg, TOPK_ZEROS_COMPETE, aux, sc, &s_ncand,
                                     p.out_idx + out_base, p.out_val + out_base, 0, wg_cand_idx,
                                     (long long)p.n_cols - total_nonzero);
            __syncthreads();
        }
        item_done();
    }
    if (p.phase_ticks && tid == 0) {
        const unsigned long long t_end = wall_clock64();
        atomicMax(&p.phase_ticks[9], t_end);
        atomicAdd(&p.phase_ticks[10], t_end - t_start);
    }
}

// ---------------------------------------- packed counts: two workgroups per CU ----------------------------------------
// The column kernel above keeps one workgroup per CU: a 32-bit cell per neighbour takes most of the LDS at ML-20M / Netflix shape, and a
// column's phases run one after the other -- accumulation (LDS atomics, the stream), then four latency-bound selection phases during
// which the atomic unit idles; measured in round 6, neither phase comes near a hardware limit of its own (profiles/r6_sim_phases.txt).
// Two co-resident workgroups interleave them.  They fit because, for all-ones data, a cell (c, j) never exceeds the number of users of
// column c: a column with fewer than 65 536 users needs 16 bits per cell.  This kernel packs two neighbours per LDS word -- neighbour
// j lives in half j & 1 of word j >> 1 and is incremented by 1 or 65 536 with the same 32-bit atomic; a half cannot carry into the other
// -- so an ML-20M column takes 53 KiB, two 512-thread workgroups share a CU, and the same id stream, walk lists and threshold-first
// selection serve (thread maxima over the words' two cells each, exact values of the survivors, rank, emit: bit-identical output).
// NOT handled here, by the host's choice of work items: columns with 65 536 users or more, columns that the schedule would split,
// light columns -- the 32-bit kernel runs them in a second launch, together with the columns whose threshold-first selection does
// not go through (fewer than K positive thread maxima, more survivors than the buffer holds): this kernel appends those to the
// second launch's work list (p.retry_count / p.retry_items), the accumulator is simply abandoned.
// (2.0e6 until pieces of a multi-GPU part were measured on their own: the 512 most expensive columns of an 8-way part of the ML-20M
// shape, 1.07 M pair-adds per column, took 0.51 ms packed against 0.23 ms on the 32-bit kernel -- a few long columns and nothing to
// interleave them with --, columns [0, 4096) of the whole shape, 1.05 M, 1.51 against 1.45 ms; at 0.87 M and below packed wins)
constexpr double PACKED_MAX_PAIRS_PER_COLUMN = 1.0e6;
constexpr int PACKED_PART_ENTRIES = 49152;      // walk entries (>= users) of one part of a column with 65 536 users or more: its counts stay below 2^16
constexpr int PACKED_AUX_WORDS = 4096;          // 16 KiB: the wavefront tables (8 x 1 KiB), then histogram / candidates (2 048 x 8 B)
template <int THREADS, int G>
__global__ __launch_bounds__(THREADS, 4) void sim_packed_kernel(const SimParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned *accw = reinterpret_cast<unsigned *>(smem);                       // [acc_words] two 16-bit counts per word (+ spare words)
    uint32_t *aux = reinterpret_cast<uint32_t *>(smem) + p.acc_words;
    SimShared &shared = *reinterpret_cast<SimShared *>(aux + PACKED_AUX_WORDS);
    SelectScratch &sc = shared.sc;
    int &s_col = shared.col;
    int4 &s_item = shared.item;
    int2 &s_range = shared.range;
    uint32_t &s_ncand = shared.ncand;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) float *)smem != 0u) __builtin_trap();
    const int tid = threadIdx.x, lane = tid & 63;
    const int gl = tid % G;
    constexpr int WAVES = THREADS / 64, GPW = 64 / G, DEPTH = SIM_DEPTH_UNIT;
    const int wave = tid >> 6, sub = lane / G;
    const int n_words = p.n_cols_pad / 2;                                       // words that hold neighbours (n_cols_pad is a multiple of 4)

    unsigned long long t_prev = p.phase_ticks ? wall_clock64() : 0ull;
    const unsigned long long t_start = t_prev;
    if (p.phase_ticks && tid == 0) atomicMin(&p.phase_ticks[8], t_start);
    auto mark = [&](int phase) {
        if (p.phase_ticks && tid == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&p.phase_ticks[phase], now - t_prev);
            t_prev = now;
        }
    };
    int nx_slot = -1;                        // thread 0 only (the next work item is pulled early, see sim_column_kernel)
    if (tid == 0) {
        const int sl = (int)atomicAdd(p.queue, 1u);
        s_col = sl;
        if (sl < p.n_items) {
            s_item = p.items[sl];
            s_range = p.item_range[sl];
        }
    }
    const __amdgpu_buffer_rsrc_t idx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.seg_idx16), 0, (int)0xFFFFFFF0u, 0x00020000);
    for (;;) {
        __syncthreads();
        const int slot = s_col;
        if (slot >= p.n_items) break;
        const int4 item = s_item;
        const unsigned long long t_item = p.phase_ticks ? wall_clock64() : 0ull;
        const int c = item.x;
        const int cbeg = s_range.x, cend = s_range.y;
        int4 nx_item = make_int4(0, 0, 0, 0);
        int2 nx_range = make_int2(0, 0);
        const size_t out_base = (size_t)(p.out_slot ? p.out_slot[c] : c - p.start_col) * p.topK;

        // the wavefront's stripes of the column's walk list (serpentine over the WAVES x parts virtual wavefronts, as in sim_column_kernel)
        const int n_parts = item.z & 0xFFFF;
        const bool parts_only = (item.z >> 16) != 0;        // a column of 65 536 users or more: the 32-bit launch adds its parts up
        const int NV = WAVES * n_parts, vw = item.y * WAVES + wave;
        auto entry_of = [&](int q) {
            const int stripe = q / GPW, pos = (stripe & 1) ? NV - 1 - vw : vw;
            return cbeg + (stripe * NV + pos) * GPW + (q % GPW);
        };
        auto load_user = [&](int q, int &ex) {
            const int at = entry_of(q);
            ex = at < cend ? p.walk4[at] : -1;
        };
        auto load_bounds = [&](int ex, int &rs, int &re) {
            if (ex >= 0) {
                const uint2 e = p.walk_tab[ex];
                rs = (int)e.x;
                re = (int)e.y;
            } else {
                rs = 0;
                re = -1;
            }
        };
        int x_first, x_next, t_rs, t_re;
        load_user(lane, x_first);
        load_user(64 + lane, x_next);
        {
            uint4 *a4 = reinterpret_cast<uint4 *>(accw);
            for (int w = tid; w < p.acc_words / 4; w += THREADS) a4[w] = make_uint4(0u, 0u, 0u, 0u);
        }
        load_bounds(x_first, t_rs, t_re);
        __syncthreads();
        mark(0);

        // ---- accumulation: the lean walk of sim_column_kernel, the increment chosen by the id's lowest bit ----
        int2 *tab2 = reinterpret_cast<int2 *>(aux) + wave * 128;
        tab2[64 + lane] = make_int2(0, 0);
        for (int base = 0; entry_of(base) < cend; base += 64) {
            tab2[lane] = t_re >= 0 ? make_int2(t_rs, t_re) : make_int2(0, 0);
            load_bounds(x_next, t_rs, t_re);
            load_user(base + 128 + lane, x_next);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int g8 = 8 * gl;
            int m = sub, f_t, f_re;
            {
                const int2 e = tab2[m];
                f_t = e.x;
                f_re = e.y;
            }
            uint4 ids[DEPTH];
            bool ok[DEPTH];
            auto fetch = [&](int d) {
                const int at = f_t + g8;
                ok[d] = at < f_re;
                ids[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(idx_rsrc, at * 2, 0, 0));
                f_t += 8 * G;
            };
            auto add_pair = [&](unsigned w) {                 // the two ids of one stream word
                const unsigned a0 = lds_cell_address<0, 1>(w) & 0xFFFFFFFCu, a1 = lds_cell_address<1, 1>(w) & 0xFFFFFFFCu;
                lds_add_u32(a0, (w & 1u) ? 0x10000u : 1u);
                lds_add_u32(a1, (w & 0x10000u) ? 0x10000u : 1u);
            };
            auto step = [&](int d) {
                const bool done = f_t >= f_re;
                m = min(m + (done ? GPW : 0), 127);
                const int2 e = tab2[m];
                if (ok[d]) {
                    add_pair(ids[d].x);
                    add_pair(ids[d].y);
                    add_pair(ids[d].z);
                    add_pair(ids[d].w);
                }
                f_t = done ? e.x : f_t;
                f_re = done ? e.y : f_re;
                fetch(d);
            };
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (d) {
                    const bool done = f_t >= f_re;
                    m = min(m + (done ? GPW : 0), 127);
                    const int2 e = tab2[m];
                    f_t = done ? e.x : f_t;
                    f_re = done ? e.y : f_re;
                }
                fetch(d);
            }
            for (;;) {
                bool any = false;
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    any |= ok[d];
                    step(d);
                }
                if (__ballot(any) == 0ull) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        mark(1);
        if (tid == 0) nx_slot = (int)atomicAdd(p.queue, 1u);
        if (n_parts > 1 || parts_only) {
            // Split column (see sim_column_kernel): publish the part's words; the workgroup that arrives last adds the parts up -- the sums
            // stay below 65 536 per half, the column has fewer users than that -- and carries on with the column.  (parts_only: nobody
            // here adds anything up.)
            const int pub_words = n_words;
            {
                uint4 *dst = reinterpret_cast<uint4 *>(p.part_buf + (size_t)(item.w + item.y) * p.n_cols_pad);
                const uint4 *src = reinterpret_cast<const uint4 *>(accw);
                for (int w = tid; w < pub_words / 4; w += THREADS) dst[w] = src[w];
                if (tid < (pub_words & 3)) p.part_buf[(size_t)(item.w + item.y) * p.n_cols_pad + (pub_words & ~3) + tid] = accw[(pub_words & ~3) + tid];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                shared.last = parts_only ? 0 : (atomicAdd(&p.part_count[item.w], 1u) == (unsigned)(n_parts - 1));
                if (shared.last) __threadfence();
            }
            __syncthreads();
            if (!shared.last) {
                mark(2);
                if (tid == 0) {          // (synchronous: nothing of this item is left to hide the requests behind)
                    s_col = nx_slot;
                    if (nx_slot < p.n_items) {
                        s_item = p.items[nx_slot];
                        s_range = p.item_range[nx_slot];
                    }
                    nx_slot = -1;
                }
                continue;
            }
            const uint32_t *src = p.part_buf + (size_t)item.w * p.n_cols_pad;
            for (int w = tid; w < pub_words; w += THREADS) {
                uint32_t a = src[w];
                for (int q = 1; q < n_parts; ++q) a += src[(size_t)q * p.n_cols_pad + w];
                accw[w] = a;
            }
            __syncthreads();
            mark(2);
        }
        // the diagonal was accumulated like any other cell; the spare words absorbed the padding entries; the wavefront tables are dead:
        // zero the histogram of block_kth_largest_prefix16 and what the scans read behind the last word
        if (tid == 0) accw[c >> 1] &= (c & 1) ? 0x0000FFFFu : 0xFFFF0000u;
        for (int w = tid; w < 1024; w += THREADS) aux[w] = 0u;
        if (tid < p.acc_words - n_words) accw[n_words + tid] = 0u;
        if (tid == 0) s_ncand = 0;
        __syncthreads();

        // ---- threshold-first top-K over the words (see sim_column_kernel for the argument): thread t owns words t, t + THREADS, ... ----
        bool done = false;
        {
            const bool asym = p.normalize && p.kind == MI355REC_SIM_ASYMMETRIC;
            const float norm_c = asym ? p.norm_alpha[c] : p.norm[c];
            const float *nj = asym ? p.norm_1ma : p.norm;
            const uint32_t K = (uint32_t)p.topK;
            constexpr int CPT = (MAX_TILE / 2 + THREADS - 1) / THREADS;     // rounds of THREADS words
            constexpr int CAND_MAX = PACKED_AUX_WORDS / 2;
            constexpr int BATCH = 8;                                        // words: sixteen cells and norms
            const __amdgpu_buffer_rsrc_t nj_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(nj), 0, p.n_cols_pad * 4, 0x00020000);
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            const int n_rounds = (n_words + THREADS - 1) / THREADS;
            const DenomForm form = denominator_form(p, norm_c);
            const unsigned word_at = (unsigned)tid_o * 4u, word_end = (unsigned)n_words * 4u;      // byte offsets (behind the last word: a zeroed spare word)
            auto scan_words = [&](auto &&use) {
#pragma unroll
                for (int bi = 0; bi < CPT / BATCH; ++bi) {
                    const int b = bi * BATCH;
                    if (b >= n_rounds) break;
                    float2 nrm[BATCH];
#pragma unroll
                    for (int i = 0; i < BATCH; ++i)
                        nrm[i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(nj_rsrc, tid_o * 8, (b + i) * THREADS * 8, 0));
                    unsigned wd[BATCH];
#pragma unroll
                    for (int i = 0; i < BATCH; ++i)
                        wd[i] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(accw) + min(word_at + (unsigned)(b + i) * (THREADS * 4u), word_end));
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        use(b + i, 0, (float)(wd[i] & 0xFFFFu), nrm[i].x);
                        use(b + i, 1, (float)(wd[i] >> 16), nrm[i].y);
                    }
                }
            };
            // (two maxima per thread -- over its words' low and high cells: the bound on the K-th largest cell comes from 2 x THREADS keys, as
            // tight as the 1024-thread kernel's: 102 survivors per ML-20M column instead of 164 with one maximum over both)
            float mx0 = 0.f, mx1 = 0.f;
            scan_words([&](int, int half, float v, float norm_j) {
                const float a = v * __builtin_amdgcn_rcpf(approx_denominator(form, v, norm_j));
                if (half) mx1 = fmaxf(mx1, a);
                else mx0 = fmaxf(mx0, a);
            });
            mark(5);
            const uint32_t p16 = block_kth_largest_prefix16<THREADS, 2>(float_key(mx0), K, aux, sc, float_key(mx1));
            mark(3);
            done = p16 > (ZERO_KEY >> 16);
            if (done) {
                if (tid == 0 && nx_slot < p.n_items) {
                    nx_item = p.items[nx_slot];
                    nx_range = p.item_range[nx_slot];
                }
                const float Tf = key_float(p16 << 16) * 0.99999809265136719f;
                uint64_t *cand = reinterpret_cast<uint64_t *>(aux);
                if (tid == 0) sc.out_count = 0;
                scan_words([&](int round, int half, float v, float norm_j) {
                    if (v > 0.f && v >= Tf * approx_denominator(form, v, norm_j)) {
                        const uint32_t at = atomicAdd(&s_ncand, 1u);
                        if (at < (uint32_t)CAND_MAX) cand[at] = ((uint64_t)__float_as_uint(norm_j) << 32) | (uint32_t)(2 * (tid_o + round * THREADS) + half);
                    }
                });
                __syncthreads();
                mark(6);
                const uint32_t n_cand = s_ncand;
                if (n_cand > (uint32_t)CAND_MAX || n_cand < K) {
                    done = false;
                } else {
                    if (p.fast_stats && tid == 0) {
                        atomicAdd(&p.fast_stats[0], 1ull);
                        atomicAdd(&p.fast_stats[1], (unsigned long long)n_cand);
                    }
                    for (uint32_t t = tid; t < n_cand; t += THREADS) {
                        const uint64_t e = cand[t];
                        const uint32_t j = (uint32_t)e;
                        const unsigned wv = accw[j >> 1];
                        const float x = normalise(p, (float)((j & 1u) ? wv >> 16 : wv & 0xFFFFu), norm_c, __uint_as_float((uint32_t)(e >> 32)));
                        cand[t] = ((uint64_t)float_key(x) << 32) | (uint32_t)(~j);
                    }
                    __syncthreads();
                    block_rank_emit<THREADS>(cand, (int)n_cand, p.topK, K, 0u, sc, p.out_idx + out_base, p.out_val + out_base);
                }
            }
        }
        if (!done) {        // the 32-bit kernel's launch takes the column over (whole, whatever path failed here)
            if (tid == 0) {
                if (nx_slot < p.n_items && nx_item.z == 0) {          // (the descriptor of the next item was not asked for yet)
                    nx_item = p.items[nx_slot];
                    nx_range = p.item_range[nx_slot];
                }
                const int k = atomicAdd(p.retry_count, 1);
                p.retry_items[k] = make_int4(c, 0, 1, 0);
                p.retry_ranges[k] = make_int2(cbeg, cend);
            }
        }
        if (tid == 0) {
            s_col = nx_slot;
            s_item = nx_item;
            s_range = nx_range;
            nx_slot = -1;
            if (p.phase_ticks) {
                const unsigned long long span = wall_clock64() - t_item;
                if (span > atomicMax(&p.phase_ticks[11], span)) p.phase_ticks[12] = (unsigned long long)c;
            }
        }
        __syncthreads();
        mark(4);
    }
    if (p.phase_ticks && tid == 0) {
        const unsigned long long t_end = wall_clock64();
        atomicMax(&p.phase_ticks[9], t_end);
        atomicAdd(&p.phase_ticks[10], t_end - t_start);
    }
}

// ---------------------------------------- set-up kernels -------------------------------------------

// One pass over the stored values: out[0] bit s SET when some value times 2^s (s = 0..3) is not an integer, out[1] the bits of
// max |value|, out[2] non-zero when some value is not exactly 1.
__global__ void value_scan_kernel(const float *x, size_t n, unsigned *out) {
    unsigned bad = 0, top = 0, not_unit = 0;
    auto look = [&](float v) {
        top = max(top, __float_as_uint(fabsf(v)));
        not_unit |= v != 1.0f;
#pragma unroll
        for (int sh = 0; sh <= 3; ++sh) {
            const float t = v * (float)(1 << sh);
            if (!(t == rintf(t))) bad |= 1u << sh;       // (NaN / inf never qualify)
        }
    };
    // (16 bytes per load: with one float per thread and step the scan of 80 MB took 0.2 ms -- a tenth of the HBM rate)
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        look(v.x); look(v.y); look(v.z); look(v.w);
    }
    for (size_t i = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) look(x[i]);
    // one atomic per wavefront and word (a million threads on one address each cost 0.15 ms)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad |= (unsigned)__shfl_xor((int)bad, off);
        top = max(top, (unsigned)__shfl_xor((int)top, off));
        not_unit |= (unsigned)__shfl_xor((int)not_unit, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicOr(&out[0], bad);
        if (top) atomicMax(&out[1], top);
        if (not_unit) atomicOr(&out[2], 1u);
    }
}

__global__ void fill_kernel(float *x, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = v;
}

// NumPy's float32 pairwise summation (numpy/_core/src/umath/loops_utils.h, FLOAT_pairwise_sum): < 8 elements
// sequentially, <= 128 with eight running partial sums, above that split in halves (rounded to a multiple of 8).
__device__ float numpy_pairwise_sum(const float *a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
            r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
            r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
        }
        float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise_sum(a, n2) + numpy_pairwise_sum(a + n2, n - n2);
}

// Mean of the stored cells of every segment (CSR row or CSC column).  Mean-centred data is a difference of nearly
// equal numbers, so the float32 rounding of the SUM is visible in the result: the reference's
// `dataMatrix.sum(axis=...)` on a float32 matrix is np.add.reduceat(data, indptr), i.e. first element + NumPy's
// pairwise sum of the rest, in float32 -- reproduced bit for bit (checked against SciPy on the CPU).
__global__ void segment_mean_kernel(const int *ptr, const float *val, int n_segments, float *mean) {
    const int sgm = blockIdx.x * blockDim.x + threadIdx.x;
    if (sgm >= n_segments) return;
    const int s = ptr[sgm], e = ptr[sgm + 1];
    float sum = 0.f;
    if (e > s) sum = e - s > 1 ? val[s] + numpy_pairwise_sum(val + s + 1, e - s - 1) : val[s];
    mean[sgm] = e > s ? (float)((double)sum / (double)(e - s)) : 0.f;
}

// applyAdjustedCosine (.pyx:275-310): subtract from every stored cell the mean of its row.
__global__ void row_center_kernel(const int *ptr, float *val, int n_rows, const float *mean) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    const float m = mean[wave];
    for (int q = ptr[wave] + lane; q < ptr[wave + 1]; q += 64) val[q] -= m;
}

// Padded length (multiple of 8 entries) of every (row, accumulator tile) segment; slot n_seg gets 0 so that the
// exclusive scan over n_seg + 1 slots ends with the total.
__global__ void seg_len_kernel(const int *csr_ptr, const int *row_tile_ptr, int n_rows, int n_tiles, int *len_pad) {
    const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long n_seg = (long long)n_rows * n_tiles;
    if (k > n_seg) return;
    int len = 0;
    if (k < n_seg) {
        const int u = (int)(k / n_tiles), t = (int)(k % n_tiles);
        if (n_tiles == 1) len = csr_ptr[u + 1] - csr_ptr[u];
        else len = row_tile_ptr[(size_t)u * (n_tiles + 1) + t + 1] - row_tile_ptr[(size_t)u * (n_tiles + 1) + t];
    }
    len_pad[k] = (len + 7) & ~7;
}

// The profile stream of the column kernel (one wavefront per segment): ids relative to the tile base as uint16,
// values as they are after pre-processing; padding entries point at the 4 spare accumulator cells and carry 0.
__global__ void seg_fill_kernel(const int *csr_ptr, const int *row_tile_ptr, const int *csr_idx, const float *csr_val,
                                const int *seg_ptr, int n_rows, int n_tiles, int tile_w, unsigned short *seg_idx16,
                                float *seg_val, int group_lanes, short *seg_val16, float int_half) {
    const long long k = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (k >= (long long)n_rows * n_tiles) return;
    const int u = (int)(k / n_tiles), t = (int)(k % n_tiles);
    int a, b;
    if (n_tiles == 1) {
        a = csr_ptr[u];
        b = csr_ptr[u + 1];
    } else {
        a = row_tile_ptr[(size_t)u * (n_tiles + 1) + t];
        b = row_tile_ptr[(size_t)u * (n_tiles + 1) + t + 1];
    }
    const int dst = seg_ptr[k], padded = seg_ptr[k + 1] - dst, len = b - a;
    // Lane-interleaved order inside every FULL block of 8 G entries (G = lanes per profile in the column kernel, 0 = off): the
    // kernel's lane g loads the 16-byte chunk g of a block and its e-th ds_add takes the chunk's entry e, so with the entries
    // stored in row order one instruction carries entries g * 8 + e -- ids at stride 8 of a sorted profile, and where a long
    // profile is dense (the popular items of a heavy user: ids nearly consecutive) that is 4 distinct LDS banks for 32 lanes.
    // Stored as chunk g = entries {g, g + G, g + 2 G, ...}, one instruction carries G CONSECUTIVE entries of the profile: consecutive
    // ids, distinct banks.  The kernel does not care in which order a segment's entries arrive; the tail of a segment (less than a
    // block) stays in row order, so the chunk-granular end-of-segment test still holds.  Measured at ML-20M shape: accumulation
    // 659 -> 640 workgroup-ms, kernel 4.02 -> 3.98 ms (the atomic unit itself, not the bank pattern, is what bounds the scatter).
    // One lane per 16-byte chunk of the stream (8 entries): the row-order entries first, first + step, ... are read (neighbouring
    // lanes on neighbouring entries inside a block), packed and stored as ONE aligned 16-byte word per lane -- 2-byte stores
    // at a stride of 16 bytes took 0.18 ms for the 40 MB of ids.  The counts kernel never reads values: none are written for it.
    const int block = 8 * group_lanes, n_blocked = block > 0 ? (padded / block) * block : 0;
    const int n_chunks = padded >> 3, tile_base = t * tile_w;
    uint4 *idx_out = reinterpret_cast<uint4 *>(seg_idx16 + dst);
    for (int c = lane; c < n_chunks; c += 64) {
        int first = c * 8, step = 1;
        if (first < n_blocked) {
            const int blk = c / group_lanes;
            first = blk * block + (c - blk * group_lanes);
            step = group_lanes;
        }
        unsigned id[8];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int q = first + e * step;
            const bool real = q < len;
            id[e] = (unsigned)(real ? csr_idx[a + q] - tile_base : tile_w + (q & 3)) & 0xFFFFu;
            v[e] = (real && (seg_val || seg_val16)) ? csr_val[a + q] : 0.f;
        }
        idx_out[c] = make_uint4(id[0] | (id[1] << 16), id[2] | (id[3] << 16), id[4] | (id[5] << 16), id[6] | (id[7] << 16));
        if (seg_val16) {
            unsigned h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (unsigned)__float2int_rn(v[e] * int_half) & 0xFFFFu;     // exact: the value grid was checked
            reinterpret_cast<uint4 *>(seg_val16 + dst)[c] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        } else if (seg_val) {
            float4 *val_out = reinterpret_cast<float4 *>(seg_val + dst);
            val_out[2 * c] = make_float4(v[0], v[1], v[2], v[3]);
            val_out[2 * c + 1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// row_tile_ptr[u][t] = first position of CSR row u whose column id is >= t * tile_w (rows have sorted ids).
__global__ void row_tile_ptr_kernel(const int *ptr, const int *idx, int n_rows, int tile_w, int n_tiles, int *out) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)n_rows * (n_tiles + 1)) return;
    const int u = (int)(e / (n_tiles + 1)), t = (int)(e % (n_tiles + 1));
    int lo = ptr[u], hi = ptr[u + 1];
    const long long bound = (long long)t * tile_w;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (idx[mid] < bound) lo = mid + 1; else hi = mid;
    }
    out[e] = lo;
}

// CSC column pointers from the column keys sorted by the radix sort: csc_ptr[c] = first position whose key is >= c
// (one thread per column, binary search; a histogram with global atomics took 2.6 ms at ML-20M shape -- the head
// columns serialise on their counters).
__global__ void csc_ptr_kernel(const int *sorted_cols, size_t nnz, int n_cols, int *csc_ptr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cols) return;
    size_t lo = 0, hi = nnz;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (sorted_cols[mid] < c) lo = mid + 1; else hi = mid;
    }
    csc_ptr[c] = (int)lo;
}

// Row id of every stored cell (one wave per row): the payload of the CSR -> CSC sort for all-ones data.
__global__ void expand_rows_kernel(const int *ptr, int n_rows, int *row_of) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    for (int q = ptr[wave] + lane; q < ptr[wave + 1]; q += 64) row_of[q] = wave;
}

// Payload of the CSR -> CSC sort for valued data: (row id, value bits) of every stored cell in one 8-byte word, so that the sort
// itself carries the cells into column order (a sorted permutation + a gather of rows and values through it spent 0.7 ms on its
// 2 x 20 M random 4-byte reads at the ML-20M shape).
__global__ void expand_cells_kernel(const int *ptr, const float *val, int n_rows, unsigned long long *cell) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    for (int q = ptr[wave] + lane; q < ptr[wave + 1]; q += 64)
        cell[q] = (unsigned long long)(unsigned)wave | ((unsigned long long)__float_as_uint(val[q]) << 32);
}

// CSC view from the sorted cells: users inside a column stay in ascending order (the sort is stable).
__global__ void split_cells_kernel(const unsigned long long *cell, size_t nnz, int *csc_idx, float *csc_val) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long c = cell[i];
        csc_idx[i] = (int)(unsigned)c;
        csc_val[i] = __uint_as_float((unsigned)(c >> 32));
    }
}

// cost = sum of the profile lengths of a column's rows, for ALL columns with the cells dealt evenly: a wavefront per 4 096 cells of
// the column-ordered arrays (one wavefront per column spent 1.2 ms on the longest column of the ML-20M shape alone).  `col_of` is the
// sorted key array of the CSR -> CSC sort.  Runs of one column are summed in registers, one atomic per run and wavefront; the few
// cells of a stretch of 64 that spans several columns add themselves.
// COUNTED (the walk list of all-ones data as the column view: `csc_idx` = slice numbers, `csr_ptr` = scan of the rows' lengths filed
// under their FIRST slice): bits 40.. of the sum count the entries with a non-zero length = the column's users.
constexpr int COST_CHUNK = 4096;
constexpr int COST_COUNT_SHIFT = 40;
template <bool COUNTED>
__global__ __launch_bounds__(256) void column_cost_kernel(const int *col_of, const int *csc_idx, const int *csr_ptr, size_t nnz,
                                                          unsigned long long *cost) {
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t first = wave * COST_CHUNK, last = first + COST_CHUNK < nnz ? first + COST_CHUNK : nnz;
    if (first >= nnz) return;
    int cur = -1;
    unsigned long long part = 0;
    auto flush = [&]() {
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0 && cur >= 0 && part) atomicAdd(&cost[cur], part);
        part = 0;
    };
    for (size_t q0 = first; q0 < last; q0 += 64) {
        const size_t q = q0 + lane;
        const bool live = q < last;
        const int col = live ? col_of[q] : -1;
        const int u = live ? csc_idx[q] : 0;
        unsigned long long len = live ? (unsigned long long)(csr_ptr[u + 1] - csr_ptr[u]) : 0ull;
        if (COUNTED && len) len |= 1ull << COST_COUNT_SHIFT;
        const int col0 = __builtin_amdgcn_readfirstlane(col);
        if (__all(!live || col == col0)) {
            if (col0 != cur) {
                flush();
                cur = col0;
            }
            part += len;
        } else {
            flush();
            cur = -1;
            if (live) atomicAdd(&cost[col], len);
        }
    }
    flush();
}

// applyPearsonCorrelation (.pyx:234-271): subtract the column mean from every stored cell, both views.
__global__ void col_center_kernel(const int *col_of, float *val, size_t nnz, const float *mean) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x)
        val[i] -= mean[col_of[i]];
}
__global__ void col_center_csc_kernel(const int *csc_ptr, float *csc_val, int n_cols, const float *mean) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_cols) return;
    const float m = mean[wave];
    for (int q = csc_ptr[wave] + lane; q < csc_ptr[wave + 1]; q += 64) csc_val[q] -= m;
}

// numpy_pairwise_sum over the SQUARES of a[0 .. n) (each square rounded to float32 first, like dataMatrix.power(2))
__device__ float numpy_pairwise_sum_sq(const float *a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = __fadd_rn(r, __fmul_rn(a[i], a[i]));
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = __fmul_rn(a[j], a[j]);
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], __fmul_rn(a[i + j], a[i + j]));
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, __fmul_rn(a[i], a[i]));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(numpy_pairwise_sum_sq(a, n2), numpy_pairwise_sum_sq(a + n2, n - n2));
}

// `dataMatrix.power(2).sum(axis=0)` as the reference gets it (.pyx:169): float32 squares, float32 sums, in SciPy's order for the
// format at hand -- order 0 (CSR: ones @ X): a column's squares one after the other in row order (the CSC view built here keeps
// each column's cells in row order); order 1 (CSC: np.add.reduceat): first square + NumPy's pairwise sum of the rest.  One thread
// per column: the additions of a column are a dependent chain by definition (0.3 ms for the longest column at ML-20M shape).
__global__ void column_sumsq_f32_kernel(const int *csc_ptr, const float *csc_val, int n_cols, int order, double *sumsq) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const int s = csc_ptr[c], e = csc_ptr[c + 1];
    float sum = 0.f;
    if (order == 0) {
        for (int q = s; q < e; ++q) sum = __fadd_rn(sum, __fmul_rn(csc_val[q], csc_val[q]));
    } else if (e > s) {
        sum = __fmul_rn(csc_val[s], csc_val[s]);
        if (e - s > 1) sum = __fadd_rn(sum, numpy_pairwise_sum_sq(csc_val + s + 1, e - s - 1));
    }
    sumsq[c] = (double)sum;
}
// Order 0 for long columns, one WAVEFRONT per column: the chain of float32 additions cannot be split, but its operands can be
// fetched 64 at a time (coalesced, the next chunk in flight) and handed from lane to lane with v_readlane -- 8 cycles per cell
// instead of one exposed global load each (13 ms for the 100 000-cell columns of the ML-20M shape with one thread per column).
// Lanes past the end contribute +0.0f, which leaves a non-negative float32 sum unchanged.
__global__ __launch_bounds__(256) void column_sumsq_f32_rowwise_kernel(const int *csc_ptr, const float *csc_val, int n_cols, double *sumsq) {
    const int lane = threadIdx.x & 63;
    const int c = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (c >= n_cols) return;
    const int s = csc_ptr[c], e = csc_ptr[c + 1];
    float sum = 0.f;
    float v = s + lane < e ? csc_val[s + lane] : 0.f;
    for (int q = s; q < e; q += 64) {
        const float sq = __fmul_rn(v, v);
        v = q + 64 + lane < e ? csc_val[q + 64 + lane] : 0.f;            // next chunk
#pragma unroll
        for (int l = 0; l < 64; ++l)
            sum = __fadd_rn(sum, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), l)));
    }
    if (lane == 0) sumsq[c] = (double)sum;
}

// all-ones data: the sum of a column's squares is its number of stored cells, exactly, in any order (float32 holds every count
// below 2^24; beyond that the serial float32 chain decides)
__global__ void column_count_sumsq_kernel(const int *csc_ptr, int n_cols, double *sumsq) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cols) sumsq[c] = (double)(csc_ptr[c + 1] - csc_ptr[c]);
}
__global__ void iota_kernel(int *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// sumOfSquared -> norms (.pyx:169-177)
__global__ void norms_kernel(const double *sumsq, int n_cols, int set_based, int asymmetric, int euclidean, float alpha,
                             float *norm, float *norm_alpha, float *norm_1ma) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    double s = sumsq[c];
    if (euclidean) {   // float32 like the reference: item_distance_initial and its square root (Euclidean.py:112-113)
        norm_alpha[c] = (float)s;
        norm[c] = __fsqrt_rn((float)s);
        return;
    }
    if (!set_based) s = sqrt(s);
    norm[c] = (float)s;
    if (asymmetric) {
        norm_1ma[c] = (float)pow(s, 2.0 * (1.0 - (double)alpha));
        norm_alpha[c] = (float)pow(s, 2.0 * (double)alpha);
    }
}

// ---- walk lists: what the column kernel's accumulation walks (see SimParams::walk8) ------------------------------------------------
// A user's profile segment is cut into slices of at most WALK_SLICE chunks (of 8 entries); the slices of ALL rows are ordered by
// descending length once (a few hundred thousand of them), the (column, slice) pairs are generated in that order from the CSR rows
// and a STABLE sort by column leaves every column's slices longest first.
constexpr int WALK_SLICE = 128;

// slices per row (tiled accumulators: the row itself is the entry -- its segments differ per tile)
__global__ void walk_row_slices_kernel(const int *csr_ptr, const int *seg_ptr, int n_rows, int tiled, int *n_slices) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_rows) return;
    const int len = csr_ptr[u + 1] - csr_ptr[u];
    int n = 0;
    if (len > 0) n = tiled ? 1 : ((seg_ptr[u + 1] - seg_ptr[u]) / 8 + WALK_SLICE - 1) / WALK_SLICE;
    n_slices[u] = n;
}

// one record per slice: key = its chunks (tiled: the row's chunks over all tiles, clamped), value = row | slice << 32
__global__ void walk_slice_records_kernel(const int *csr_ptr, const int *seg_ptr, const int *slice_off, int n_rows, int tiled, int n_tiles,
                                          unsigned *key, unsigned long long *rec) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_rows) return;
    const int at = slice_off[u], n = slice_off[u + 1] - at;
    if (n == 0) return;
    if (tiled) {
        const int chunks = (seg_ptr[(size_t)(u + 1) * n_tiles] - seg_ptr[(size_t)u * n_tiles]) / 8;
        key[at] = (unsigned)min(chunks / 4, 255);
        rec[at] = (unsigned long long)(unsigned)u;
        return;
    }
    const int chunks = (seg_ptr[u + 1] - seg_ptr[u]) / 8;
    for (int j = 0; j < n; ++j) {
        key[at + j] = (unsigned)min(WALK_SLICE, chunks - j * WALK_SLICE);
        rec[at + j] = (unsigned long long)(unsigned)u | ((unsigned long long)j << 32);
    }
}

// per record (sorted order): the cells it emits (= the row's length), the row's length filed under its first slice only (what a
// column's cost and user count are summed from), and the slice's bounds in the profile stream
__global__ void walk_record_lengths_kernel(const unsigned long long *rec, const int *csr_ptr, const int *seg_ptr, int n_rec, int tiled,
                                           int *len, int *first_len, uint2 *tab) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rec) {
        const unsigned long long rc = rec[r];
        const int u = (int)(unsigned)rc, j = (int)(rc >> 32);
        const int n = csr_ptr[u + 1] - csr_ptr[u];
        len[r] = n;
        first_len[r] = j == 0 ? n : 0;
        if (!tiled) {
            const int s0 = seg_ptr[u], s1 = seg_ptr[u + 1];
            tab[r] = make_uint2((unsigned)(s0 + j * WALK_SLICE * 8), (unsigned)min(s1, s0 + (j + 1) * WALK_SLICE * 8));
        }
    }
    if (r == n_rec) {
        len[r] = 0;
        first_len[r] = 0;
    }
}

// the packed sums of column_cost_kernel<true>: cost, users (as the column's sum of squares and as an int)
__global__ void walk_unpack_cost_kernel(const unsigned long long *packed, int n_cols, long long *cost, double *sumsq, int *count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const unsigned long long v = packed[c];
    cost[c] = (long long)(v & ((1ull << COST_COUNT_SHIFT) - 1ull));
    sumsq[c] = (double)(v >> COST_COUNT_SHIFT);
    count[c] = (int)(v >> COST_COUNT_SHIFT);
}

// One workgroup per slice record (in sorted order): its (column, entry) pairs, one per stored cell of the row.  WIDE: 16-byte entries
// with the column-side value of the cell (times the row's weight), else 4-byte entries: the record's number (tiled: the row).
template <bool WIDE>
__global__ __launch_bounds__(256) void walk_generate_kernel(const unsigned long long *rec, const int *out_off, const int *csr_ptr, const int *csr_idx,
                                                            const float *csr_val, const int *seg_ptr, const float *row_w, int unit_col, int tiled,
                                                            int *key, void *entries) {
    const int r = blockIdx.x;
    const unsigned long long rc = rec[r];
    const int u = (int)(unsigned)rc, j = (int)(rc >> 32);
    const int a = csr_ptr[u], len = csr_ptr[u + 1] - a, at = out_off[r];
    unsigned ex = (unsigned)u, ey = 0u;
    if (!tiled && WIDE) {
        const int s0 = seg_ptr[u], s1 = seg_ptr[u + 1];
        ex = (unsigned)(s0 + j * WALK_SLICE * 8);
        ey = (unsigned)min(s1, s0 + (j + 1) * WALK_SLICE * 8);
    }
    const float w = row_w ? row_w[u] : 1.f;
    for (int i = threadIdx.x; i < len; i += 256) {
        key[at + i] = csr_idx[a + i];
        if (WIDE) {
            float cv = unit_col ? 1.f : csr_val[a + i];
            if (row_w) cv *= w;
            reinterpret_cast<uint4 *>(entries)[at + i] = make_uint4(ex, ey, __float_as_uint(cv), 0u);
        } else {
            reinterpret_cast<int *>(entries)[at + i] = tiled ? u : r;
        }
    }
}

// ---- BM25 / TF-IDF re-weighting of the stored values (Base/IR_feature_weighting.py:13-75) ---------------------------------
// Per row and per column of the CSR: the sum of the stored values and their number; one wavefront per row, the column side
// through atomics (20 M cells at ML-20M shape: a few hundred microseconds, once per build).  float64 throughout -- the
// reference mixes float32 (row sums, length norm) and float64 (idf); the float32 results agree to a few 1e-7 relative.
__global__ __launch_bounds__(256) void weighting_stats_kernel(const int *csr_ptr, const int *csr_idx, const float *csr_val, int n_rows,
                                                              double *row_sum, double *col_sum, int *col_cnt, double *total) {
    const int lane = threadIdx.x & 63;
    const int row = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (row >= n_rows) return;
    double sum = 0.0;
    for (int q = csr_ptr[row] + lane; q < csr_ptr[row + 1]; q += 64) {
        const double v = (double)csr_val[q];
        sum += v;
        atomicAdd(&col_sum[csr_idx[q]], v);
        atomicAdd(&col_cnt[csr_idx[q]], 1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) {
        row_sum[row] = sum;
        if (sum != 0.0) atomicAdd(&total[row & 63], sum);
    }
}

// okapi_BM_25 (:35-49): idf = log(N / (1 + cells of the term)), length_norm = (1 - B) + B * document_sum / mean document sum,
// value * (K1 + 1) / (K1 * length_norm + value) * idf, a zero denominator replaced by 1e-9;  TF_IDF (:69-73): sqrt(value) * idf.
__global__ __launch_bounds__(256) void weighting_apply_kernel(const int *csr_ptr, const int *csr_idx, float *csr_val, int n_rows, int n_cols,
                                                              const double *row_sum, const double *col_sum, const int *col_cnt,
                                                              const double *total, int mode, int documents_are_rows, double k1, double b) {
    const int lane = threadIdx.x & 63;
    const int row = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (row >= n_rows) return;
    double all = 0.0;
    for (int w = 0; w < 64; ++w) all += total[w];
    const double n_docs = documents_are_rows ? (double)n_rows : (double)n_cols;
    const double mean_len = all / n_docs;
    const int begin = csr_ptr[row], end = csr_ptr[row + 1];
    for (int q = begin + lane; q < end; q += 64) {
        const int col = csr_idx[q];
        const double v = (double)csr_val[q];
        const double cells_of_term = documents_are_rows ? (double)col_cnt[col] : (double)(end - begin);
        const double idf = log(n_docs / (1.0 + cells_of_term));
        double out;
        if (mode == MI355REC_WEIGHT_BM25) {
            const double doc_sum = documents_are_rows ? row_sum[row] : col_sum[col];
            double den = k1 * ((1.0 - b) + b * doc_sum / mean_len) + v;
            if (den == 0.0) den += 1e-9;
            out = v * (k1 + 1.0) / den * idf;
        } else {
            out = sqrt(v) * idf;
        }
        csr_val[q] = (float)out;
    }
}

// largest |value| (bit pattern of a non-negative float orders like the unsigned integer)
__global__ void absmax_kernel(const float *val, size_t nnz, unsigned *out) {
    unsigned m = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(fabsf(val[i])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// CSR assembly of the result (.pyx:603-605: row = neighbour, column = source item) -- sort keys: the neighbour id of
// every slab entry, padding entries (-1) mapped past the last row so that they sort to the end.
__global__ void csr_keys_kernel(const int *slab_idx, size_t n, int n_cols, int *key, int *pos) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int r = slab_idx[e];
        key[e] = r >= 0 ? r : n_cols;
        pos[e] = (int)e;
    }
}

// After the stable sort by neighbour id: entry t of the CSR arrays comes from slab position pos[t]; its column is the
// slab row (source item).  Positions ascend inside a row, hence so do the columns: indices come out sorted.
__global__ void csr_gather_kernel(const int *pos, const float *slab_val, size_t n, int topK, int start_col, int *indices,
                                  float *data) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int e = pos[t];
        indices[t] = start_col + e / topK;
        data[t] = slab_val[e];
    }
}

// [n_local][n_cols] -> [n_cols][n_local] (32x32 tiles through LDS)
__global__ void transpose_kernel(const float *in, float *out, int rows, int cols) {
    __shared__ float tile[32][33];
    int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 32 + threadIdx.y;
    for (int k = 0; k < 32; k += 8)
        if (x < cols && y + k < rows) tile[threadIdx.y + k][threadIdx.x] = in[(size_t)(y + k) * cols + x];
    __syncthreads();
    x = blockIdx.y * 32 + threadIdx.x;
    y = blockIdx.x * 32 + threadIdx.y;
    for (int k = 0; k < 32; k += 8)
        if (x < rows && y + k < cols) out[(size_t)(y + k) * rows + x] = tile[threadIdx.x][threadIdx.y + k];
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

struct mi355rec_sim {
    mi355rec_sim_config cfg{};
    int n_rows = 0, n_cols = 0;
    size_t nnz = 0;
    bool unit_values = false;
    hipStream_t stream = nullptr;
    StreamTimer timer;       // start/stop events carried by the column-kernel dispatch itself
    StreamTimer call_timer;  // events around the whole call (H2D of the schedule, kernel, D2H of the result)
    DeviceBuffer<int> csr_ptr, csr_idx, csc_ptr, csc_idx;
    DeviceBuffer<int4> items;
    DeviceBuffer<uint32_t> part_buf;
    DeviceBuffer<unsigned> part_count;
    DeviceBuffer<unsigned long long> phase_ticks;
    DeviceBuffer<unsigned long long> selection_counts;   // [0] columns finished by the threshold-first selection, [1] their candidates, [2] fall-backs after its scan
    DeviceBuffer<int> csr_key, csr_key_sorted, csr_pos, csr_pos_sorted, csr_indptr, csr_indices;   // mi355rec_sim_compute_csr
    DeviceBuffer<float> csr_data;
    DeviceBuffer<char> csr_sort_tmp;
    std::vector<int4> items_host;   // host staging for the current call
    std::vector<int2> ranges_host;
    DeviceBuffer<int2> item_range;
    int n_split_columns = 0, n_part_items = 0;
    DeviceBuffer<unsigned short> seg_idx16;
    DeviceBuffer<int> seg_ptr;
    DeviceBuffer<float> seg_val;
    DeviceBuffer<short> seg_val16;
    DeviceBuffer<int> row_tile_ptr, cand_idx;
    DeviceBuffer<float> cand_val;
    int tile_w = 0, n_tiles = 1;
    DeviceBuffer<float> csr_val, csc_val, row_w, norm, norm_alpha, norm_1ma;
    DeviceBuffer<int> out_slot;         // interleaved parts: output row per column
    DeviceBuffer<float> weighted_val;   // feature_weighting: the re-weighted values as handed back to the recommender
    DeviceBuffer<unsigned> queue;
    DeviceBuffer<int> out_idx;
    DeviceBuffer<float> out_val;
    std::vector<long long> cost;   // host copy
    std::vector<int> csc_ptr_host;
    DeviceBuffer<int> walk4;            // walk lists of the column kernel (all-ones data), see SimParams::walk4
    DeviceBuffer<uint2> walk_tab;       //   ... and the bounds of the slices they name
    DeviceBuffer<uint4> walk16;         // walk lists with the column-side weight (valued data)
    std::vector<int> walk_ptr_host;     // [n_cols + 1] a column's entries in the walk arrays
    std::vector<int> cost_order;   // all columns, most expensive first
    int group_lanes = 64;
    double fixed_scale = 0.0;      // real-valued data: power-of-two scale of the int64 fixed-point accumulator (0: float64 sums)
    bool wide_topk = false;
    double wide_kernel_ms = -1.0, wide_call_ms = 0.0;   // >= 0 after a build with topK > MAX_TOPK (several launches: the event pair of the last one is not the build)
    int int_shift = -1;            // >= 0: every stored value times 2^int_shift is a small integer -> exact int32 sums (ACC_INT32)
    int acc_mode() const { return unit_values && !row_w.ptr ? ACC_COUNTS : (int_shift >= 0 ? ACC_INT32 : ACC_WIDE); }
    mi355rec_stats stats{};
    // last call
    int last_start = -1, last_end = -1;

    ~mi355rec_sim() {   // also runs when mi355rec_sim_create fails half-way: nothing leaks
        if (stream) (void)hipStreamSynchronize(stream);
        timer.destroy();
        call_timer.destroy();
        ReleaseScope::forget(stream);
        if (stream) pooled_stream_return(stream);
    }
};

namespace {

template <int THREADS, int G>
void launch_sim(mi355rec_sim *h, const SimParams &p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    auto go = [&](auto k) {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipExtLaunchKernelGGL(k, dim3(grid), dim3(THREADS), (unsigned)lds, h->stream, e0, e1, 0, p);
    };
    switch (h->acc_mode()) {
        case ACC_COUNTS: go(sim_column_kernel<THREADS, G, ACC_COUNTS>); break;
        case ACC_INT32: go(sim_column_kernel<THREADS, G, ACC_INT32>); break;
        default: go(sim_column_kernel<THREADS, G, ACC_WIDE>); break;
    }
    MI_HIP(hipGetLastError());
}

template <int THREADS>
void launch_sim_g(mi355rec_sim *h, const SimParams &p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (h->group_lanes) {
        case 4: launch_sim<THREADS, 4>(h, p, grid, lds, e0, e1); break;
        case 8: launch_sim<THREADS, 8>(h, p, grid, lds, e0, e1); break;
        case 16: launch_sim<THREADS, 16>(h, p, grid, lds, e0, e1); break;
        case 32: launch_sim<THREADS, 32>(h, p, grid, lds, e0, e1); break;
        default: launch_sim<THREADS, 64>(h, p, grid, lds, e0, e1); break;
    }
}

// the packed-counts kernel (all-ones data, one tile, 512 threads, two workgroups per CU)
void launch_packed(mi355rec_sim *h, const SimParams &p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    auto go = [&](auto k) {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipExtLaunchKernelGGL(k, dim3(grid), dim3(512), (unsigned)lds, h->stream, e0, e1, 0, p);
    };
    switch (h->group_lanes) {
        case 4: go(sim_packed_kernel<512, 4>); break;
        case 16: go(sim_packed_kernel<512, 16>); break;
        default: go(sim_packed_kernel<512, 8>); break;
    }
    MI_HIP(hipGetLastError());
}

void clamp_range(const mi355rec_sim *h, int32_t &s, int32_t &e) {
    // same rule as .pyx:447-451: out-of-range bounds fall back to the full range
    int32_t s_in = s, e_in = e;
    s = 0;
    e = h->n_cols;
    if (s_in > 0 && s_in < h->n_cols) s = s_in;
    if (e_in > s && e_in < h->n_cols) e = e_in;
}

// Interleaved parts (multi-GPU): the columns in cost order are dealt to the parts in serpentine order -- position p of the
// cost order belongs to group p / n_parts and, inside the group, to part p % n_parts (even groups) or its mirror image (odd
// groups) -- so every part receives the same NUMBER of columns (+-1) and the same COST (the heavy head of the order is
// spread over all parts).  A part's output rows are its groups, in order.
inline int part_of_position(long long pos, int n_parts) {
    const long long group = pos / n_parts;
    const int within = (int)(pos % n_parts);
    return (group & 1) ? n_parts - 1 - within : within;
}

// Runs the column kernel for [start,end) -- or, with n_parts > 0, for part `start` of `n_parts` interleaved parts -- leaving
// results in d_idx/d_val (or d_dense when topK == 0).
void run_columns_lds(mi355rec_sim *h, int32_t start, int32_t end, int *d_idx, float *d_val, float *d_dense, int n_parts, int slot_first = 0,
                     int slot_count = 0x7fffffff);

// valid after the stream has been synchronised past the last build
void read_timers(mi355rec_sim *h) {
    if (h->wide_kernel_ms >= 0.0) {
        h->stats.kernel_ms = h->wide_kernel_ms;
        h->stats.call_ms = h->wide_call_ms;
    } else {
        h->stats.kernel_ms = h->timer.elapsed_ms();
        h->stats.call_ms = h->call_timer.elapsed_ms();
    }
}

// topK beyond the in-LDS selection (MAX_TOPK = 4096 candidates): the reference only clamps topK to n_cols (.pyx:146).  The columns
// are built DENSE into HBM (the kernel's topK == 0 path), every column is sorted by descending value with one segmented radix sort
// (rocPRIM; stable: equal values keep ascending neighbour ids, the in-LDS path's tie rule) and the K largest cells of the full
// column -- zeros compete, then are dropped (.pyx:523-555) -- are emitted.  Blocks of columns bound the scratch memory.
__global__ __launch_bounds__(256) void wide_topk_emit_kernel(const float *sorted_val, const int *sorted_id, int n_cols, int topK, int *out_idx,
                                                             float *out_val) {
    __shared__ int s_npos, s_nnonneg;
    const float *val = sorted_val + (size_t)blockIdx.x * n_cols;
    const int *id = sorted_id + (size_t)blockIdx.x * n_cols;
    if (threadIdx.x < 2) {           // first position whose value is <= 0 (thread 0) / < 0 (thread 1): descending order
        int lo = 0, hi = n_cols;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const bool before = threadIdx.x == 0 ? val[mid] > 0.f : val[mid] >= 0.f;
            if (before) lo = mid + 1; else hi = mid;
        }
        if (threadIdx.x == 0) s_npos = lo; else s_nnonneg = lo;
    }
    __syncthreads();
    const int npos = s_npos, nzero = s_nnonneg - s_npos;
    const int take_pos = min(topK, npos), take_neg = max(0, min(topK - npos - nzero, n_cols - npos - nzero));
    int *oi = out_idx + (size_t)blockIdx.x * topK;
    float *ov = out_val + (size_t)blockIdx.x * topK;
    for (int r = threadIdx.x; r < topK; r += 256) {
        int src = -1;
        if (r < take_pos) src = r;
        else if (r < take_pos + take_neg) src = npos + nzero + (r - take_pos);
        oi[r] = src >= 0 ? id[src] : -1;
        ov[r] = src >= 0 ? val[src] : 0.f;
    }
}

__global__ void wide_iota_kernel(int *ids, unsigned *offsets, int n_rows, int n_cols) {
    const size_t n = (size_t)n_rows * n_cols;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) ids[e] = (int)(e % n_cols);
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r <= (size_t)n_rows; r += (size_t)gridDim.x * blockDim.x)
        offsets[r] = (unsigned)(r * n_cols);
}

// n_parts > 0: rows [part_first, part_first + part_count) of interleaved part `start` only (a sharded build computes its part in pieces)
void run_columns_wide_topk(mi355rec_sim *h, int32_t start, int32_t end, int *d_idx, float *d_val, int n_parts, int part_first = 0,
                           int part_count = 0x7fffffff) {
    const int topK = h->cfg.topK, n_cols = h->n_cols;
    // rows of the output, in output order: a contiguous range, or one interleaved part (whose rows the kernel places by out_slot)
    int n_local = end - start;
    if (n_parts > 0) {
        n_local = 0;
        for (long long pos = 0; pos < n_cols; ++pos) n_local += part_of_position(pos, n_parts) == start;
        n_local = std::max(0, std::min(part_count, n_local - part_first));
    }
    if (n_local == 0) return;
    // 4 GiB per float buffer (MI355REC_SIM_WIDE_CELLS: a smaller bound, for tests of the block walk)
    const size_t cells_cap = getenv("MI355REC_SIM_WIDE_CELLS") ? (size_t)std::max(1ll, atoll(getenv("MI355REC_SIM_WIDE_CELLS"))) : (size_t)1 << 30;
    int block = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_local, cells_cap / (size_t)n_cols));
    DeviceBuffer<float> dense, sorted_val;
    DeviceBuffer<int> ids, sorted_id;
    DeviceBuffer<unsigned> offsets;
    DeviceBuffer<unsigned char> tmp;
    dense.alloc((size_t)block * n_cols); sorted_val.alloc((size_t)block * n_cols);
    ids.alloc((size_t)block * n_cols); sorted_id.alloc((size_t)block * n_cols);
    offsets.alloc((size_t)block + 1);
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(wide_iota_kernel, dim3(4096), dim3(256), 0, s, ids.ptr, offsets.ptr, block, n_cols);
    size_t tmp_bytes = 0;
    MI_HIP(rocprim::segmented_radix_sort_pairs_desc(nullptr, tmp_bytes, dense.ptr, sorted_val.ptr, ids.ptr, sorted_id.ptr,
                                                    (unsigned)((size_t)block * n_cols), (unsigned)block, offsets.ptr, offsets.ptr + 1, 0, 32, s));
    tmp.alloc(tmp_bytes + 256);
    const mi355rec_sim_config saved = h->cfg;
    double kernel_ms = 0, units = 0, bytes = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    MI_HIP(hipEventCreate(&t0));
    MI_HIP(hipEventCreate(&t1));
    MI_HIP(hipEventRecord(t0, s));
    try {
        for (int done = 0; done < n_local; done += block) {
            const int here = std::min(block, n_local - done);
            h->cfg.topK = 0;
            if (n_parts > 0) run_columns_lds(h, start, 0, nullptr, nullptr, dense.ptr, n_parts, part_first + done, here);
            else run_columns_lds(h, start + done, start + done + here, nullptr, nullptr, dense.ptr, 0);
            h->cfg = saved;
            MI_HIP(hipStreamSynchronize(s));
            kernel_ms += h->timer.elapsed_ms();
            units += h->stats.n_units;
            bytes += h->stats.algorithmic_bytes;
            size_t bytes_now = tmp_bytes;
            MI_HIP(rocprim::segmented_radix_sort_pairs_desc(tmp.ptr, bytes_now, dense.ptr, sorted_val.ptr, ids.ptr, sorted_id.ptr,
                                                            (unsigned)((size_t)here * n_cols), (unsigned)here, offsets.ptr, offsets.ptr + 1, 0, 32, s));
            hipLaunchKernelGGL(wide_topk_emit_kernel, dim3(here), dim3(256), 0, s, sorted_val.ptr, sorted_id.ptr, n_cols, topK,
                               d_idx + (size_t)done * topK, d_val + (size_t)done * topK);
            MI_HIP(hipGetLastError());
        }
    } catch (...) {
        h->cfg = saved;
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        throw;
    }
    // the handle's timers are read by the callers after this returns: make them cover the whole wide build
    MI_HIP(hipEventRecord(t1, s));
    MI_HIP(hipStreamSynchronize(s));
    float whole_ms = 0.f;
    MI_HIP(hipEventElapsedTime(&whole_ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    h->wide_kernel_ms = kernel_ms;
    h->wide_call_ms = whole_ms;
    h->stats.n_units = (int64_t)units;
    h->stats.algorithmic_bytes = bytes + 8.0 * (double)n_local * topK;
}

void run_columns(mi355rec_sim *h, int32_t start, int32_t end, int *d_idx, float *d_val, float *d_dense, int n_parts = 0) {
    h->wide_kernel_ms = -1.0;
    if (!d_dense && h->wide_topk && h->cfg.topK > 0) run_columns_wide_topk(h, start, end, d_idx, d_val, n_parts);
    else run_columns_lds(h, start, end, d_idx, d_val, d_dense, n_parts);
}

// n_parts > 0: interleaved part `start` of n_parts; of its columns (in output order) only slots [slot_first, slot_first + slot_count)
// are built, into output rows 0 .. slot_count - 1 (the wide top-K path walks a part in blocks).
void run_columns_lds(mi355rec_sim *h, int32_t start, int32_t end, int *d_idx, float *d_val, float *d_dense, int n_parts, int slot_first,
                     int slot_count) {
    const int part = start;
    std::vector<int> slot_host;
    int n_local = end - start;
    if (n_parts > 0) {
        slot_host.assign((size_t)h->n_cols, -1);
        n_local = 0;
        int seen = 0;
        for (long long pos = 0; pos < h->n_cols; ++pos)
            if (part_of_position(pos, n_parts) == part) {
                if (seen >= slot_first && n_local < slot_count) slot_host[h->cost_order[pos]] = n_local++;
                ++seen;
            }
        start = 0;
        end = h->n_cols;
    }
    auto in_call = [&](int c) { return n_parts > 0 ? slot_host[c] >= 0 : (c >= start && c < end); };
    const bool unit_kernel = h->acc_mode() != ACC_WIDE;          // 4-byte cells
    const int acc_words = (h->tile_w + 4) * (unit_kernel ? 1 : 2);
    const size_t lds = (size_t)acc_words * 4 + (size_t)AUX_WORDS * 4 + sizeof(SimShared);
    const int cus = multiprocessor_count();
    int threads = 1024, max_grid = cus;   // one 16-wave workgroup per CU when the accumulator owns the LDS
    if (lds <= 72 * 1024 && !getenv("MI355REC_SIM_ONE_WG_PER_CU")) {       // (the variable: measurements of one 16-wave workgroup against several 8-wave ones)
        const int per_cu = std::max(1, std::min(4, (int)((160 * 1024) / (lds + 1024))));
        threads = 512;
        max_grid = cus * per_cu;
    }

    // ---- schedule: work items, most expensive first (LPT).  A column whose cost exceeds 1/2 of a workgroup's fair
    //      share is split into parts (contiguous runs of its users) that different workgroups accumulate; otherwise
    //      the head items bound the build as soon as the range is spread over many CUs (at ML-20M shape the top
    //      column is 0.49 of a CU's share on one GPU, 3.9 on eight).  Not combined with accumulator tiling.
    long long cost_sum = 0;
    double nnz_range = 0;
    for (int c = start; c < end; ++c)
        if (in_call(c)) {
            cost_sum += h->cost[c];
            nnz_range += (double)(h->csc_ptr_host[c + 1] - h->csc_ptr_host[c]);
        }
    int min_part_users = 4 * threads;
    if (getenv("MI355REC_SIM_MIN_PART_USERS")) min_part_users = std::max(64, atoi(getenv("MI355REC_SIM_MIN_PART_USERS")));
    // threshold-first selection (fast_column_topk in the kernel): positive denominators only (the set-based modes and tversky's
    // alpha / beta inside the range the approximation's error bound was derived for), K well below the number of thread maxima
    auto fast_topk_for = [&](int n_threads) {
        const char *sw = getenv("MI355REC_SIM_FAST_TOPK");
        const bool tversky_ok = h->cfg.similarity != MI355REC_SIM_TVERSKY ||
                                (h->cfg.tversky_alpha >= 0.f && h->cfg.tversky_alpha <= 4.f && h->cfg.tversky_beta >= 0.f && h->cfg.tversky_beta <= 4.f);
        return !(sw && atoi(sw) == 0) && unit_kernel && h->n_tiles == 1 && h->cfg.topK > 0 && 4 * h->cfg.topK <= n_threads &&
               h->cfg.similarity != MI355REC_SIM_EUCLIDEAN && h->cfg.shrink >= 0 && tversky_ok;
    };
    // The packed-counts kernel (sim_packed_kernel: two 512-thread workgroups per CU) takes the columns it can: all-ones data, one
    // tile, the threshold-first selection applicable, fewer than 65 536 users, not light, cheap enough not to be split over its grid.
    // Everything else -- and whatever that kernel hands over -- goes to the 32-bit kernel's launch behind it.
    const int packed_words = ((h->tile_w / 2 + 2) + 3) & ~3;
    const size_t lds_packed = (size_t)packed_words * 4 + (size_t)PACKED_AUX_WORDS * 4 + sizeof(SimShared);
    // ... and only where a column's fixed phases weigh something next to its accumulation: below PACKED_MAX_PAIRS_PER_COLUMN
    // pair-adds per column of the call (ML-20M shape: 0.29 M, kernel 3.80 -> 3.03-3.10 ms; 138 493 x 9 000 with the same stored
    // values: 0.87 M, 2.44 -> 2.09-2.17 ms; Netflix shape: 3.0 M, accumulation 92 % of the kernel, 16.5 -> 17.0 ms: not packed; the
    // head of an 8-way part, 1.07 M: 0.51 against 0.23 ms: not packed).
    // MI355REC_SIM_PACKED=1 / 0 forces it on (where it applies) / off.
    const char *packed_env = getenv("MI355REC_SIM_PACKED");
    const bool packed_pays = packed_env ? atoi(packed_env) != 0 : (double)cost_sum < PACKED_MAX_PAIRS_PER_COLUMN * (double)std::max(1, n_local);
    const bool packed = h->acc_mode() == ACC_COUNTS && h->n_tiles == 1 && !d_dense && threads == 1024 && fast_topk_for(512) &&
                        2 * (lds_packed + 1024) <= 160 * 1024 && (h->group_lanes == 4 || h->group_lanes == 8 || h->group_lanes == 16) &&
                        packed_pays && !getenv("MI355REC_SIM_NO_PACKED");
    const int packed_grid = 2 * cus;
    std::vector<int4> packed_items, merge_items;       // merge_items: 32-bit launch, columns whose packed parts it adds up
    std::vector<int> merge_parts;
    std::vector<char> is_packed;
    long long legacy_cost = 0, packed_cost = 0;
    int part_slots = 0, n_split = 0;
    const bool heavy_parts = !(getenv("MI355REC_SIM_PACKED_HEAVY") && atoi(getenv("MI355REC_SIM_PACKED_HEAVY")) == 0);
    if (packed) {
        is_packed.assign((size_t)h->n_cols, 0);
        for (int c : h->cost_order) {
            if (!in_call(c)) continue;
            // (columns of 65 536 users or more: accumulated there in parts of fewer users each, added up by the 32-bit launch; more
            // than 64 such parts: left to the 32-bit kernel)
            const int n_c = h->walk_ptr_host[c + 1] - h->walk_ptr_host[c];
            const bool many = h->csc_ptr_host[c + 1] - h->csc_ptr_host[c] >= 65536;
            if ((many ? heavy_parts && n_c <= 64 * PACKED_PART_ENTRIES : true) && h->cost[c] >= 16ll * std::max(1, h->cfg.topK)) {
                is_packed[(size_t)c] = 1;
                packed_cost += h->cost[c];
            } else {
                legacy_cost += h->cost[c];
            }
        }
        // HEAVY columns gain nothing from the packed launch -- what it offers is a second workgroup's accumulation beside a column's
        // selection phases, and a heavy column is nearly all accumulation, on workgroups of 8 wavefronts instead of 16 (the 8 heaviest
        // columns of an 8-way part of the ML-20M shape: 0.37 ms packed against 0.125 ms; 504 columns of 0.93 M pair-adds: 0.245 against
        // 0.204 ms; 715 of 0.26 M: 0.102 against 0.115 ms -- packed wins).  Heavy = more than a quarter of a packed workgroup's fair share
        // of the call, and at least 0.5 M pair-adds.  Where such columns are a large share of the call (a part of an 8-way build: 60 %
        // of its pair-adds; the whole shape: 14 %, where a second launch of that size only adds a tail -- measured 3.05 against 3.00 ms)
        // they go to the 32-bit launch behind this one: slowest part of 8 0.56 -> 0.45-0.47 ms, identical output.
        // MI355REC_SIM_PACKED_DEMOTE=0 / 1 forces it off / on.
        {
            const long long heavy = std::max<long long>(500000, packed_cost / ((long long)packed_grid * 4));
            long long heavy_cost = 0;
            for (int c : h->cost_order)
                if (in_call(c) && is_packed[(size_t)c] && h->cost[c] > heavy) heavy_cost += h->cost[c];
            const char *dm = getenv("MI355REC_SIM_PACKED_DEMOTE");
            const bool demote = dm ? atoi(dm) != 0 : (double)heavy_cost >= 0.4 * (double)packed_cost;
            if (demote)
                for (int c : h->cost_order) {
                    if (!in_call(c) || !is_packed[(size_t)c] || h->cost[c] <= heavy) continue;
                    is_packed[(size_t)c] = 0;
                    packed_cost -= h->cost[c];
                    legacy_cost += h->cost[c];
                }
        }
        // its heavy columns are split like the 32-bit kernel's: a part is at most 1/4 of a workgroup's fair share (its workgroups have
        // 8 wavefronts: an unsplit column of 1/2 share kept one of them busy for a third of the launch)
        const long long plimit = std::max<long long>(1, packed_cost / ((long long)packed_grid * 4));
        std::vector<std::pair<long long, int>> pkeyed;
        for (int c : h->cost_order) {
            if (!in_call(c) || !is_packed[(size_t)c]) continue;
            const int n_c = h->walk_ptr_host[c + 1] - h->walk_ptr_host[c];
            const bool many_users = h->csc_ptr_host[c + 1] - h->csc_ptr_host[c] >= 65536;
            long long parts = 1;
            if (h->cost[c] > plimit) parts = std::max<long long>(1, std::min<long long>({(h->cost[c] + plimit - 1) / plimit, (long long)n_c / (4 * 512), 64ll}));
            if (many_users) parts = std::max<long long>(parts, (n_c + PACKED_PART_ENTRIES - 1) / PACKED_PART_ENTRIES);
            for (int q = 0; q < (int)parts; ++q) {
                pkeyed.emplace_back(h->cost[c] / parts, (int)packed_items.size());
                packed_items.push_back(make_int4(c, q, (int)parts | (many_users ? 1 << 16 : 0), parts > 1 || many_users ? part_slots : 0));
            }
            if (many_users) merge_items.push_back(make_int4(c, 0, 1, -(1 + part_slots)));
            if (many_users) merge_parts.push_back((int)parts);
            if (parts > 1 || many_users) {
                part_slots += (int)parts;
                ++n_split;
            }
        }
        if (n_split) {
            std::stable_sort(pkeyed.begin(), pkeyed.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            std::vector<int4> sorted(pkeyed.size());
            for (size_t i = 0; i < pkeyed.size(); ++i) sorted[i] = packed_items[pkeyed[i].second];
            packed_items.swap(sorted);
        }
    }
    const int n_packed = (int)packed_items.size();
    const int n_split_packed = n_split;
    // (the 32-bit launch behind a packed one splits ITS columns -- the heaviest of the call -- over its whole grid)
    const long long limit = std::max<long long>(1, (n_packed ? legacy_cost : cost_sum) / ((long long)max_grid * 2));
    h->items_host.clear();
    h->items_host.reserve((size_t)n_local + 8 * (size_t)max_grid);
    std::vector<std::pair<long long, int>> keyed;   // (item cost, index into items_host)
    keyed.reserve(h->items_host.capacity());
    for (int c : h->cost_order) {
        if (!in_call(c)) continue;
        if (n_packed && is_packed[(size_t)c]) continue;
        const int n_c = h->walk_ptr_host[c + 1] - h->walk_ptr_host[c];      // entries of the column's walk list
        long long parts = 1;
        if (h->n_tiles == 1 && h->cost[c] > limit)
            parts = std::max<long long>(1, std::min<long long>({(h->cost[c] + limit - 1) / limit, (long long)n_c / min_part_users, 64ll}));
        if (parts > 1) {
            for (int q = 0; q < (int)parts; ++q) {
                keyed.emplace_back(h->cost[c] / parts, (int)h->items_host.size());
                h->items_host.push_back(make_int4(c, q, (int)parts, part_slots));
            }
            part_slots += (int)parts;
            ++n_split;
        } else {
            keyed.emplace_back(h->cost[c], (int)h->items_host.size());
            // .w of an unsplit column: 1 = LIGHT -- fewer than 16 K pair-adds cannot leave K positive thread maxima behind (real
            // catalogues: half of ML-20M's items have fewer than 20 ratings), so the threshold-first selection would scan the
            // accumulator twice only to hand the column to the full path, which is quick on such columns anyway (all-zero quads
            // are skipped, nothing to select among fewer than K positives)
            h->items_host.push_back(make_int4(c, 0, 1, h->cost[c] < 16ll * std::max(1, h->cfg.topK) ? 1 : 0));
        }
    }
    if (n_split > n_split_packed) {
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
        std::vector<int4> sorted(keyed.size());
        for (size_t i = 0; i < keyed.size(); ++i) sorted[i] = h->items_host[keyed[i].second];
        h->items_host.swap(sorted);
    }
    h->items_host.insert(h->items_host.end(), merge_items.begin(), merge_items.end());        // (cheap: nothing to accumulate)
    // device layout of the work lists: [the packed kernel's items | the 32-bit kernel's items | room for every packed item handed over]
    const int n_legacy = (int)h->items_host.size();
    h->items_host.insert(h->items_host.begin(), packed_items.begin(), packed_items.end());
    const int n_items = (int)h->items_host.size();
    h->n_split_columns = n_split;
    h->n_part_items = part_slots;
    if (h->items.count < (size_t)n_items + (size_t)n_packed) {
        h->items.alloc((size_t)n_items + (size_t)n_packed + 1024);
        h->item_range.alloc((size_t)n_items + (size_t)n_packed + 1024);
    }
    h->ranges_host.resize((size_t)n_items);
    for (int i = 0; i < n_items; ++i) {
        const int c = h->items_host[i].x;
        h->ranges_host[i] = make_int2(h->walk_ptr_host[c], h->walk_ptr_host[c + 1]);
        if (i >= n_packed && h->items_host[i].w < 0) {            // a column added up from packed parts: the empty list [parts, parts)
            const int parts = merge_parts[(size_t)(i - (n_items - (int)merge_items.size()))];
            h->ranges_host[i] = make_int2(parts, parts);
        }
    }
    MI_HIP(hipMemcpyAsync(h->items.ptr, h->items_host.data(), sizeof(int4) * n_items, hipMemcpyHostToDevice, h->stream));
    MI_HIP(hipMemcpyAsync(h->item_range.ptr, h->ranges_host.data(), sizeof(int2) * n_items, hipMemcpyHostToDevice, h->stream));
    MI_HIP(hipMemsetAsync(h->queue.ptr, 0, 4 * sizeof(unsigned), h->stream));           // [0] the 32-bit launch's queue, [1] the packed launch's, [2] items of the 32-bit launch
    if (n_packed) MI_HIP(hipMemcpyAsync(h->queue.ptr + 2, &n_legacy, sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (part_slots) {
        const size_t pub_words = (size_t)h->tile_w * (unit_kernel ? 1 : 2);
        if (h->part_buf.count < (size_t)part_slots * pub_words) h->part_buf.alloc((size_t)part_slots * pub_words);
        if (h->part_count.count < (size_t)part_slots) h->part_count.alloc((size_t)part_slots);
        MI_HIP(hipMemsetAsync(h->part_count.ptr, 0, sizeof(unsigned) * part_slots, h->stream));
    }

    SimParams p{};
    p.n_rows = h->n_rows;
    p.n_cols = h->n_cols;
    p.n_cols_pad = h->tile_w;          // neighbour cells of the LDS accumulator
    p.acc_cells = h->tile_w + 4;
    p.acc_words = acc_words;
    p.tile_w = h->tile_w;
    p.n_tiles = h->n_tiles;
    p.topK = h->cfg.topK;
    p.kind = h->cfg.similarity;
    p.normalize = h->cfg.normalize;
    p.unit_col = h->cfg.unit_column_side;
    p.avg_row = h->cfg.normalize_avg_row;
    p.euclid_mode = h->cfg.euclidean_mode;
    p.shrink = (float)h->cfg.shrink;
    p.tversky_alpha = h->cfg.tversky_alpha;
    p.tversky_beta = h->cfg.tversky_beta;
    p.csr_ptr = h->csr_ptr.ptr;
    p.seg_ptr = h->seg_ptr.ptr;
    p.seg_idx16 = h->seg_idx16.ptr;
    p.seg_val = h->seg_val.ptr;
    p.seg_val16 = h->seg_val16.ptr;
    p.csc_ptr = h->csc_ptr.ptr;
    p.csc_idx = h->csc_idx.ptr;
    p.csc_val = h->csc_val.ptr;
    p.walk4 = h->walk4.ptr;
    p.walk_tab = h->walk_tab.ptr;
    p.walk16 = h->walk16.ptr;
    p.row_w = h->row_w.ptr;
    p.norm = h->norm.ptr;
    p.norm_alpha = h->norm_alpha.ptr;
    p.norm_1ma = h->norm_1ma.ptr;
    p.items = h->items.ptr + n_packed;
    p.item_range = h->item_range.ptr + n_packed;
    p.n_items = n_legacy;
    p.part_buf = h->part_buf.ptr;
    p.part_count = h->part_count.ptr;
    if (getenv("MI355REC_SIM_PHASES")) {
        if (!h->phase_ticks.ptr) h->phase_ticks.alloc(16);
        MI_HIP(hipMemsetAsync(h->phase_ticks.ptr, 0, 16 * sizeof(unsigned long long), h->stream));
        MI_HIP(hipMemsetAsync(h->phase_ticks.ptr + 8, 0xFF, sizeof(unsigned long long), h->stream));      // [8]: a minimum
        p.phase_ticks = h->phase_ticks.ptr;
    }
    if (!h->selection_counts.ptr) h->selection_counts.alloc(4);
    MI_HIP(hipMemsetAsync(h->selection_counts.ptr, 0, 4 * sizeof(unsigned long long), h->stream));
    p.fast_stats = h->selection_counts.ptr;
    p.fast_topk = fast_topk_for(threads);
    p.fixed_scale = unit_kernel ? 0.0 : h->fixed_scale;
    p.int_scale = h->int_shift >= 0 ? (float)(1 << (2 * h->int_shift)) : 1.f;
    p.int_half = h->int_shift >= 0 ? (float)(1 << h->int_shift) : 1.f;
    p.int_inv = 1.f / p.int_scale;
    p.fixed_inv = p.fixed_scale > 0.0 ? 1.0 / p.fixed_scale : 0.0;
    p.start_col = start;
    p.out_slot = nullptr;
    if (n_parts > 0) {
        if (h->out_slot.count < (size_t)h->n_cols) h->out_slot.alloc((size_t)h->n_cols);
        MI_HIP(hipMemcpyAsync(h->out_slot.ptr, slot_host.data(), sizeof(int) * (size_t)h->n_cols, hipMemcpyHostToDevice, h->stream));
        MI_HIP(hipStreamSynchronize(h->stream));       // (slot_host is a local)
        p.out_slot = h->out_slot.ptr;
    }
    p.queue = h->queue.ptr;
    p.out_idx = d_idx;
    p.out_val = d_val;
    p.out_dense = d_dense;

    const int grid = n_packed ? max_grid : std::min(n_legacy, max_grid);      // (behind a packed launch the list may grow)
    if (h->n_tiles > 1 && p.topK > 0) {
        const size_t need = (size_t)grid * h->n_tiles * p.topK;
        if (h->cand_idx.count < need) {
            h->cand_idx.alloc(need);
            h->cand_val.alloc(need);
        }
    }
    p.cand_idx = h->cand_idx.ptr;
    p.cand_val = h->cand_val.ptr;
    h->call_timer.start(h->stream);
    if (n_packed) {
        SimParams q = p;
        q.items = h->items.ptr;
        q.item_range = h->item_range.ptr;
        q.n_items = n_packed;
        q.acc_words = packed_words;
        q.queue = h->queue.ptr + 1;
        q.retry_count = reinterpret_cast<int *>(h->queue.ptr + 2);
        q.retry_items = h->items.ptr + n_packed;
        q.retry_ranges = h->item_range.ptr + n_packed;
        launch_packed(h, q, std::min(n_packed, packed_grid), lds_packed, h->timer.t0, nullptr);
        p.n_items_dev = reinterpret_cast<const int *>(h->queue.ptr + 2);
        launch_sim_g<1024>(h, p, grid, lds, nullptr, h->timer.t1);
    } else if (threads == 1024) {
        launch_sim_g<1024>(h, p, grid, lds, h->timer.t0, h->timer.t1);
    } else {
        launch_sim_g<512>(h, p, grid, lds, h->timer.t0, h->timer.t1);
    }
    h->call_timer.stop(h->stream);

    h->stats.n_launches = n_packed ? 2 : 1;
    h->stats.n_timed = 1;
    h->stats.n_units = n_local;
    // ALGORITHMIC bytes, SURVEY.md section 8(d): per column c, its CSC column (8 B x n_c) + the CSR row of each of its users
    // (8 B x L_u) + topK x 8 B of output, i.e. 8 * (nnz_range + cost_range) + 8 * n_local * topK.  (The kernel's own
    // layout moves less -- uint16 ids, no values for all-ones data -- see DESIGN.md section 4.)
    h->stats.algorithmic_bytes = 8.0 * (nnz_range + (double)cost_sum) + 8.0 * (double)n_local * (double)h->cfg.topK;
    h->stats.algorithmic_flops = 0;
    h->last_start = start;
    h->last_end = end;
}

}  // namespace

// `resident`: the three CSR arrays are device memory (mi355rec_sim_create_resident) -- copied at HBM speed instead of over PCIe
static int sim_create_from(mi355rec_sim_t *out, const mi355rec_sim_config *cfg, int32_t n_rows, int32_t n_cols,
                           const int32_t *csr_indptr, const int32_t *csr_indices, const float *csr_data,
                           const float *row_weights, bool resident) {
    return guarded([&] {
        MI_REQUIRE(out && cfg && csr_indptr && csr_indices && csr_data, "NULL argument");
        MI_REQUIRE(n_rows > 0 && n_cols > 0, "empty matrix (%d x %d)", n_rows, n_cols);
        MI_REQUIRE(cfg->similarity >= MI355REC_SIM_COSINE && cfg->similarity <= MI355REC_SIM_EUCLIDEAN,
                   "Cosine_Similarity: value for parameter 'mode' not recognized (%d)", cfg->similarity);
        const bool euclid = cfg->similarity == MI355REC_SIM_EUCLIDEAN;
        if (euclid) {
            MI_REQUIRE(cfg->euclidean_mode >= MI355REC_EUCLID_LIN && cfg->euclidean_mode <= MI355REC_EUCLID_EXP,
                       "Compute_Similarity_Euclidean: value for parameter 'mode' not recognized (%d)", cfg->euclidean_mode);
            // the reference multiplies the distances to the n_cols columns by the n_rows weights (Euclidean.py:174-175): NumPy refuses
            // that for any other shape ("operands could not be broadcast together")
            if (row_weights)
                MI_REQUIRE(n_rows == n_cols, "Compute_Similarity_Euclidean: row_weights need a square dataMatrix (the reference multiplies the "
                           "%d column distances by the %d row weights: operands could not be broadcast together)", n_cols, n_rows);
        }
        MI_REQUIRE(cfg->topK >= 0, "topK must be >= 0");
        const auto t_enter = std::chrono::steady_clock::now();
        ensure_device();
        std::unique_ptr<mi355rec_sim> h(new mi355rec_sim());
        h->cfg = *cfg;
        h->cfg.topK = std::min(cfg->topK, n_cols);  // .pyx:146
        // (topK > MAX_TOPK, the in-LDS selection's candidate buffer: dense columns + segmented sort, run_columns_wide_topk)
        const bool set_based = cfg->similarity == MI355REC_SIM_JACCARD || cfg->similarity == MI355REC_SIM_DICE ||
                               cfg->similarity == MI355REC_SIM_TVERSKY;
        if (set_based) h->cfg.normalize = 0;  // .pyx:124-135
        h->n_rows = n_rows;
        h->n_cols = n_cols;
        int32_t nnz_in = 0;
        if (resident) MI_HIP(hipMemcpy(&nnz_in, csr_indptr + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost));
        else nnz_in = csr_indptr[n_rows];
        h->nnz = (size_t)nnz_in;
        MI_REQUIRE(nnz_in > 0, "matrix has no stored values");
        h->stream = pooled_stream();
        ReleaseScope scope(h->stream);          // the constructor's temporaries wait for this stream, not for the device
        h->timer.init_pooled();
        h->call_timer.init_pooled();
        hipStream_t s = h->stream;
        const size_t nnz = h->nnz;
        // MI355REC_SIM_CREATE_PHASES=1: wall clock of the constructor's phases on stderr (each one drained before the next starts)
        const bool phases = getenv("MI355REC_SIM_CREATE_PHASES") != nullptr;
        auto t_phase = t_enter;
        auto phase = [&](const char *what) {
            if (!phases) return;
            (void)hipStreamSynchronize(s);
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[sim create] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
            t_phase = now;
        };
        // (the handle keeps its own copy either way: the values are re-weighted / centred in place and the arrays are padded)
        const hipMemcpyKind in_kind = resident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        phase("device, stream, events, nnz");
        h->csr_ptr.alloc((size_t)n_rows + 1);
        MI_HIP(hipMemcpyAsync(h->csr_ptr.ptr, csr_indptr, ((size_t)n_rows + 1) * sizeof(int), in_kind, s));
        // padding: the column kernel reads the profiles in aligned 16-byte chunks, a whole lane group at a time
        h->csr_idx.alloc(nnz + 520);                 // (only the padding needs the zeros)
        h->csr_val.alloc(nnz + 520);
        MI_HIP(hipMemsetAsync(h->csr_idx.ptr + nnz, 0, 520 * sizeof(int), s));
        MI_HIP(hipMemsetAsync(h->csr_val.ptr + nnz, 0, 520 * sizeof(float), s));
        MI_HIP(hipMemcpyAsync(h->csr_idx.ptr, csr_indices, nnz * sizeof(int), in_kind, s));
        MI_HIP(hipMemcpyAsync(h->csr_val.ptr, csr_data, nnz * sizeof(float), in_kind, s));
        if (row_weights) h->row_w.upload(row_weights, n_rows, s);
        phase(resident ? "allocate + copy of the resident URM" : "allocate + upload (PCIe)");
        const int eb = 256, eg = std::min<size_t>((nnz + eb - 1) / eb, 4096);
        // One pass over the values as they came in: the reference's dispatcher asserts that they are finite (Compute_Similarity.py:34-36,
        // np.isfinite over the whole array: 9 of the 18 ms of an ItemKNN fit at ML-20M shape when the front-end did it on the host) -- the
        // largest |value|'s bits say so here -- and, where nothing re-writes the values before the build, the same pass answers "all
        // ones?" and "which 2^s grid?" below.
        unsigned info0[3] = {0xFu, 0u, 1u};
        {
            DeviceBuffer<unsigned> scan;
            scan.alloc_zero(3, s);
            hipLaunchKernelGGL(value_scan_kernel, dim3(eg), dim3(eb), 0, s, h->csr_val.ptr, nnz, scan.ptr);
            MI_HIP(hipGetLastError());
            scan.download(info0, 3, s);
            MI_HIP(hipStreamSynchronize(s));
            if (info0[1] >= 0x7F800000u) fail(MI355REC_E_INVALID, "Compute_Similarity: Data matrix contains non finite values");
        }

        // optional pre-pass: BM25 / TF-IDF on the stored values (what the KNN recommenders do to the matrix before the build)
        if (cfg->feature_weighting != MI355REC_WEIGHT_NONE) {
            MI_REQUIRE(cfg->feature_weighting == MI355REC_WEIGHT_BM25 || cfg->feature_weighting == MI355REC_WEIGHT_TFIDF,
                       "Value for 'feature_weighting' not recognized (%d)", cfg->feature_weighting);
            if (cfg->feature_weighting == MI355REC_WEIGHT_BM25) {
                MI_REQUIRE(cfg->bm25_b > 0.f && cfg->bm25_b < 1.f, "okapi_BM_25: B must be in (0,1)");
                MI_REQUIRE(cfg->bm25_k1 > 0.f, "okapi_BM_25: K1 must be > 0");
            }
            DeviceBuffer<double> row_sum, col_sum, total;
            DeviceBuffer<int> col_cnt;
            row_sum.alloc((size_t)n_rows);
            col_sum.alloc_zero((size_t)n_cols, s);
            col_cnt.alloc_zero((size_t)n_cols, s);
            total.alloc_zero(64, s);
            const int wg = div_up((int64_t)n_rows * 64, 256);
            hipLaunchKernelGGL(weighting_stats_kernel, dim3(wg), dim3(256), 0, s, h->csr_ptr.ptr, h->csr_idx.ptr, h->csr_val.ptr, n_rows,
                               row_sum.ptr, col_sum.ptr, col_cnt.ptr, total.ptr);
            hipLaunchKernelGGL(weighting_apply_kernel, dim3(wg), dim3(256), 0, s, h->csr_ptr.ptr, h->csr_idx.ptr, h->csr_val.ptr, n_rows, n_cols,
                               row_sum.ptr, col_sum.ptr, col_cnt.ptr, total.ptr, cfg->feature_weighting, cfg->weighting_documents,
                               (double)cfg->bm25_k1, (double)cfg->bm25_b);
            MI_HIP(hipGetLastError());
            h->weighted_val.alloc(nnz);
            MI_HIP(hipMemcpyAsync(h->weighted_val.ptr, h->csr_val.ptr, nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
            MI_HIP(hipStreamSynchronize(s));            // (the statistics buffers go out of scope here)
        }
        // pre-processing of the stored values (.pyx:158-164)
        if (set_based) hipLaunchKernelGGL(fill_kernel, dim3(eg), dim3(eb), 0, s, h->csr_val.ptr, nnz, 1.0f);
        // All-ones data (implicit URMs, every set-based similarity) takes the integer-count kernel, which never reads
        // the value arrays; mean-centred data never qualifies.
        h->unit_values = set_based;
        // Quantised values (star ratings, half stars, counts): if every stored value times 2^s (s <= 3) is an integer of at most
        // 2048 and n_rows products of that size cannot overflow an int32, the column sums are exact integers (ACC_INT32).  Not for
        // mean-centred data (adjusted / pearson centre the values later) nor with row weights.  One pass answers both questions.
        if (!set_based && cfg->similarity != MI355REC_SIM_ADJUSTED && cfg->similarity != MI355REC_SIM_PEARSON) {
            // [0] bit s set: some value times 2^s is not an integer; [1] bits of max |value|; [2] not all ones
            unsigned info[3] = {info0[0], info0[1], info0[2]};
            if (cfg->feature_weighting != MI355REC_WEIGHT_NONE) {          // (the weighting has re-written the values: look again)
                DeviceBuffer<unsigned> scan;
                scan.alloc_zero(3, s);
                hipLaunchKernelGGL(value_scan_kernel, dim3(eg), dim3(eb), 0, s, h->csr_val.ptr, nnz, scan.ptr);
                MI_HIP(hipGetLastError());
                info[0] = 0xFu; info[1] = 0u; info[2] = 1u;
                scan.download(info, 3, s);
                MI_HIP(hipStreamSynchronize(s));
            }
            h->unit_values = (info[2] == 0);
            if (!h->unit_values && !row_weights && !getenv("MI355REC_SIM_F64_SUMS") && !getenv("MI355REC_SIM_NO_INT32")) {
                float vmax_f;
                memcpy(&vmax_f, &info[1], sizeof(float));
                for (int sh = 0; sh <= 3; ++sh) {
                    const double m = (double)vmax_f * (double)(1 << sh);
                    if (!((info[0] >> sh) & 1u) && m <= 2048.0 && (double)n_rows * m * m < 2147483648.0) {
                        h->int_shift = sh;
                        break;
                    }
                }
            }
        }
        phase("value checks (weighting, units, grid)");
        // accumulator tiling: the LDS holds MAX_TILE 4-byte cells (counts, exact integer sums) or MAX_TILE_F64 8-byte cells (other
        // real-valued data, row weights) next to the 32 KiB selection scratch
        const int max_tile = h->acc_mode() != ACC_WIDE ? MAX_TILE : MAX_TILE_F64;
        h->tile_w = n_cols <= max_tile ? ((n_cols + 3) & ~3) : max_tile;
        h->n_tiles = (n_cols + h->tile_w - 1) / h->tile_w;
        // topK beyond the in-LDS selection, or per-tile candidates (n_tiles x topK) beyond the merge buffer: dense columns + segmented sort
        h->wide_topk = h->cfg.topK > MAX_TOPK || (long long)h->n_tiles * h->cfg.topK > h->tile_w;
        DeviceBuffer<float> row_mean;
        if (cfg->similarity == MI355REC_SIM_ADJUSTED) {
            row_mean.alloc((size_t)n_rows);
            hipLaunchKernelGGL(segment_mean_kernel, dim3(div_up(n_rows, 256)), dim3(256), 0, s, h->csr_ptr.ptr, h->csr_val.ptr,
                               n_rows, row_mean.ptr);
            hipLaunchKernelGGL(row_center_kernel, dim3(div_up((int64_t)n_rows * 64, 256)), dim3(256), 0, s, h->csr_ptr.ptr,
                               h->csr_val.ptr, n_rows, row_mean.ptr);
        }

        const bool stream_order = !(getenv("MI355REC_SIM_STREAM_ORDER") && atoi(getenv("MI355REC_SIM_STREAM_ORDER")) == 0);
        const long long n_seg = (long long)n_rows * h->n_tiles;
        // the profile stream: (row, tile) segments padded to whole 16-byte chunks, from the pre-processed values.  Its offsets
        // (build_seg_ptr) depend on the row lengths alone; its contents (fill_stream) on the pre-processed values and on group_lanes
        auto build_seg_ptr = [&]() {
            DeviceBuffer<int> len_pad;
            DeviceBuffer<char> scan_tmp;
            len_pad.alloc((size_t)n_seg + 1);
            h->seg_ptr.alloc((size_t)n_seg + 1);
            hipLaunchKernelGGL(seg_len_kernel, dim3(div_up(n_seg + 1, 256)), dim3(256), 0, s, h->csr_ptr.ptr, h->row_tile_ptr.ptr,
                               n_rows, h->n_tiles, len_pad.ptr);
            size_t scan_bytes = 0;
            MI_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, len_pad.ptr, h->seg_ptr.ptr, 0, (size_t)(n_seg + 1), rocprim::plus<int>(), s));
            scan_tmp.alloc(scan_bytes);
            MI_HIP(rocprim::exclusive_scan(scan_tmp.ptr, scan_bytes, len_pad.ptr, h->seg_ptr.ptr, 0, (size_t)(n_seg + 1), rocprim::plus<int>(), s));
        };
        auto fill_stream = [&]() {
            const size_t seg_cap = nnz + 7 * (size_t)n_seg + 520;     // every segment grows by at most 7 entries
            MI_REQUIRE(seg_cap < (size_t)INT32_MAX, "matrix too large for 32-bit segment offsets");
            h->seg_idx16.alloc_zero(seg_cap, s);
            const bool int16_values = h->acc_mode() == ACC_INT32;      // (ids + int16 values: 4 B per entry instead of 6)
            if (int16_values) h->seg_val16.alloc_zero(seg_cap, s);
            else if (h->acc_mode() != ACC_COUNTS) h->seg_val.alloc_zero(seg_cap, s);     // (the counts kernel reads ids only)
            hipLaunchKernelGGL(seg_fill_kernel, dim3(div_up(n_seg * 64, 256)), dim3(256), 0, s, h->csr_ptr.ptr, h->row_tile_ptr.ptr,
                               h->csr_idx.ptr, h->csr_val.ptr, h->seg_ptr.ptr, n_rows, h->n_tiles, h->tile_w, h->seg_idx16.ptr,
                               h->seg_val.ptr, stream_order ? h->group_lanes : 0, h->seg_val16.ptr, (float)(1 << std::max(0, h->int_shift)));
            MI_HIP(hipGetLastError());
            MI_HIP(hipStreamSynchronize(s));      // the temporaries above go out of scope
        };
        // the walk lists (what the column kernel's accumulation walks instead of the CSC arrays): slices of the profile segments,
        // every column's longest first.  All-ones data: 4-byte entries (slice numbers) -- the sorted list is the column view itself,
        // `walk_keys` (its sorted column keys) and `walk_len_ptr` (scan of the rows' lengths by first slice) are what the columns'
        // costs and user counts are then summed from, and no CSC is ever built (walk_only).
        DeviceBuffer<int> walk_keys, walk_len_ptr;
        int n_walk_entries = 0;
        auto build_walk = [&]() {
            const int tiled = h->n_tiles > 1;
            const bool wide = h->acc_mode() != ACC_COUNTS;
            DeviceBuffer<int> n_slices, slice_off, rec_len, first_len, out_off, key_in, key_sorted, walk_ptr;
            DeviceBuffer<unsigned> rec_key, rec_key_sorted;
            DeviceBuffer<unsigned long long> rec, rec_sorted;
            DeviceBuffer<char> tmp;
            n_slices.alloc((size_t)n_rows + 1);
            slice_off.alloc((size_t)n_rows + 1);
            MI_HIP(hipMemsetAsync(n_slices.ptr + n_rows, 0, sizeof(int), s));
            hipLaunchKernelGGL(walk_row_slices_kernel, dim3(div_up(n_rows, 256)), dim3(256), 0, s, h->csr_ptr.ptr, h->seg_ptr.ptr, n_rows, tiled, n_slices.ptr);
            size_t bytes = 0;
            MI_HIP(rocprim::exclusive_scan(nullptr, bytes, n_slices.ptr, slice_off.ptr, 0, (size_t)n_rows + 1, rocprim::plus<int>(), s));
            tmp.alloc(bytes + 16);
            MI_HIP(rocprim::exclusive_scan(tmp.ptr, bytes, n_slices.ptr, slice_off.ptr, 0, (size_t)n_rows + 1, rocprim::plus<int>(), s));
            int n_rec = 0;
            MI_HIP(hipMemcpyAsync(&n_rec, slice_off.ptr + n_rows, sizeof(int), hipMemcpyDeviceToHost, s));
            MI_HIP(hipStreamSynchronize(s));
            MI_REQUIRE(n_rec > 0, "matrix has no stored values");
            rec_key.alloc((size_t)n_rec); rec_key_sorted.alloc((size_t)n_rec);
            rec.alloc((size_t)n_rec); rec_sorted.alloc((size_t)n_rec);
            hipLaunchKernelGGL(walk_slice_records_kernel, dim3(div_up(n_rows, 256)), dim3(256), 0, s, h->csr_ptr.ptr, h->seg_ptr.ptr, slice_off.ptr, n_rows,
                               tiled, h->n_tiles, rec_key.ptr, rec.ptr);
            bytes = 0;
            MI_HIP(rocprim::radix_sort_pairs_desc(nullptr, bytes, rec_key.ptr, rec_key_sorted.ptr, rec.ptr, rec_sorted.ptr, (size_t)n_rec, 0, 8, s));
            DeviceBuffer<char> tmp2;
            tmp2.alloc(bytes + 16);
            MI_HIP(rocprim::radix_sort_pairs_desc(tmp2.ptr, bytes, rec_key.ptr, rec_key_sorted.ptr, rec.ptr, rec_sorted.ptr, (size_t)n_rec, 0, 8, s));
            rec_len.alloc((size_t)n_rec + 1);
            first_len.alloc((size_t)n_rec + 1);
            out_off.alloc((size_t)n_rec + 1);
            if (!wide && !tiled) h->walk_tab.alloc((size_t)n_rec);
            hipLaunchKernelGGL(walk_record_lengths_kernel, dim3(div_up(n_rec + 1, 256)), dim3(256), 0, s, rec_sorted.ptr, h->csr_ptr.ptr, h->seg_ptr.ptr,
                               n_rec, (int)(wide || tiled), rec_len.ptr, first_len.ptr, h->walk_tab.ptr);
            bytes = 0;
            MI_HIP(rocprim::exclusive_scan(nullptr, bytes, rec_len.ptr, out_off.ptr, 0, (size_t)n_rec + 1, rocprim::plus<int>(), s));
            DeviceBuffer<char> tmp3;
            tmp3.alloc(bytes + 16);
            MI_HIP(rocprim::exclusive_scan(tmp3.ptr, bytes, rec_len.ptr, out_off.ptr, 0, (size_t)n_rec + 1, rocprim::plus<int>(), s));
            if (!wide && !tiled) {         // (tiled: the entries are rows, whose lengths the CSR pointers give)
                walk_len_ptr.alloc((size_t)n_rec + 1);
                bytes = tmp3.count;
                MI_HIP(rocprim::exclusive_scan(tmp3.ptr, bytes, first_len.ptr, walk_len_ptr.ptr, 0, (size_t)n_rec + 1, rocprim::plus<int>(), s));
            }
            int n_walk = 0;
            MI_HIP(hipMemcpyAsync(&n_walk, out_off.ptr + n_rec, sizeof(int), hipMemcpyDeviceToHost, s));
            MI_HIP(hipStreamSynchronize(s));
            MI_REQUIRE(n_walk > 0 && (size_t)n_walk >= nnz, "walk list: %d entries for %zu stored values", n_walk, nnz);
            key_in.alloc((size_t)n_walk); key_sorted.alloc((size_t)n_walk);
            walk_ptr.alloc((size_t)n_cols + 1);
            int key_bits = 1;
            while ((1ll << key_bits) < (long long)n_cols) ++key_bits;
            DeviceBuffer<char> tmp4;
            if (wide) {
                DeviceBuffer<uint4> gen;
                gen.alloc((size_t)n_walk);
                h->walk16.alloc((size_t)n_walk);
                hipLaunchKernelGGL(walk_generate_kernel<true>, dim3(n_rec), dim3(256), 0, s, rec_sorted.ptr, out_off.ptr, h->csr_ptr.ptr, h->csr_idx.ptr,
                                   h->csr_val.ptr, h->seg_ptr.ptr, h->row_w.ptr, (int)cfg->unit_column_side, tiled, key_in.ptr, (void *)gen.ptr);
                bytes = 0;
                MI_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key_in.ptr, key_sorted.ptr, gen.ptr, h->walk16.ptr, (size_t)n_walk, 0, key_bits, s));
                tmp4.alloc(bytes + 16);
                MI_HIP(rocprim::radix_sort_pairs(tmp4.ptr, bytes, key_in.ptr, key_sorted.ptr, gen.ptr, h->walk16.ptr, (size_t)n_walk, 0, key_bits, s));
                MI_HIP(hipStreamSynchronize(s));       // (gen goes out of scope)
            } else {
                DeviceBuffer<int> gen;
                gen.alloc((size_t)n_walk);
                h->walk4.alloc((size_t)n_walk);
                hipLaunchKernelGGL(walk_generate_kernel<false>, dim3(n_rec), dim3(256), 0, s, rec_sorted.ptr, out_off.ptr, h->csr_ptr.ptr, h->csr_idx.ptr,
                                   h->csr_val.ptr, h->seg_ptr.ptr, h->row_w.ptr, (int)cfg->unit_column_side, tiled, key_in.ptr, (void *)gen.ptr);
                bytes = 0;
                MI_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key_in.ptr, key_sorted.ptr, gen.ptr, h->walk4.ptr, (size_t)n_walk, 0, key_bits, s));
                tmp4.alloc(bytes + 16);
                MI_HIP(rocprim::radix_sort_pairs(tmp4.ptr, bytes, key_in.ptr, key_sorted.ptr, gen.ptr, h->walk4.ptr, (size_t)n_walk, 0, key_bits, s));
                MI_HIP(hipStreamSynchronize(s));
            }
            hipLaunchKernelGGL(csc_ptr_kernel, dim3(div_up(n_cols + 1, 256)), dim3(256), 0, s, key_sorted.ptr, (size_t)n_walk, n_cols, walk_ptr.ptr);
            MI_HIP(hipGetLastError());
            h->walk_ptr_host.resize((size_t)n_cols + 1);
            walk_ptr.download(h->walk_ptr_host.data(), (size_t)n_cols + 1, s);
            MI_HIP(hipStreamSynchronize(s));
            n_walk_entries = n_walk;
            walk_keys.swap(key_sorted);
        };
        // gather the pre-processed values into the column order
        DeviceBuffer<float> mean;
        DeviceBuffer<double> sumsq;
        DeviceBuffer<long long> cost;
        // all-ones data: the walk list is the column view (one sort of 4-byte entries instead of the CSC's sort + the list's)
        const bool walk_only = h->acc_mode() == ACC_COUNTS && n_rows < (1 << 23) && !getenv("MI355REC_SIM_WALK_WITH_CSC");
        if (!walk_only) {
            h->csc_idx.alloc(nnz);
            h->csc_val.alloc(nnz);
        }
        if (h->n_tiles > 1) {
            h->row_tile_ptr.alloc((size_t)n_rows * (h->n_tiles + 1));
            hipLaunchKernelGGL(row_tile_ptr_kernel, dim3(div_up((int64_t)n_rows * (h->n_tiles + 1), 256)), dim3(256), 0, s,
                               h->csr_ptr.ptr, h->csr_idx.ptr, n_rows, h->tile_w, h->n_tiles, h->row_tile_ptr.ptr);
        }
        // CSR -> CSC (.pyx:203-207), on the device: stable radix sort of the pre-processed cells by column.  All-ones data sorts the
        // row ids alone (the values of the column view are a fill); valued data sorts (row id, value) words and splits them.
        // (Measured and rejected: the sort on a second stream behind the upload of the values -- the upload of pageable memory and
        // the sort's kernels got into each other's way, 5.96 ms for the two against 2.92 + 1.16 ms one after the other.)
        DeviceBuffer<int> key_out, row_of;
        DeviceBuffer<unsigned long long> cell_in, cell_out;
        DeviceBuffer<char> sort_tmp;
        if (walk_only) {
            build_seg_ptr();
            build_walk();
        } else {
            h->csc_ptr.alloc((size_t)n_cols + 1);
            key_out.alloc(nnz);
            int key_bits = 1;
            while ((1ll << key_bits) < (long long)n_cols) ++key_bits;
            size_t tmp_bytes = 0;
            const int rg = div_up((int64_t)n_rows * 64, 256);
            if (h->unit_values) {
                row_of.alloc(nnz);
                hipLaunchKernelGGL(expand_rows_kernel, dim3(rg), dim3(256), 0, s, h->csr_ptr.ptr, n_rows, row_of.ptr);
                MI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->csr_idx.ptr, key_out.ptr, row_of.ptr, h->csc_idx.ptr,
                                                          (int)nnz, 0, key_bits, s));
                sort_tmp.alloc(tmp_bytes);
                MI_HIP(rocprim::radix_sort_pairs(sort_tmp.ptr, tmp_bytes, h->csr_idx.ptr, key_out.ptr, row_of.ptr, h->csc_idx.ptr,
                                                          (int)nnz, 0, key_bits, s));
                hipLaunchKernelGGL(fill_kernel, dim3(eg), dim3(eb), 0, s, h->csc_val.ptr, nnz, 1.0f);
            } else {
                cell_in.alloc(nnz);
                cell_out.alloc(nnz);
                hipLaunchKernelGGL(expand_cells_kernel, dim3(rg), dim3(256), 0, s, h->csr_ptr.ptr, h->csr_val.ptr, n_rows, cell_in.ptr);
                MI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->csr_idx.ptr, key_out.ptr, cell_in.ptr, cell_out.ptr,
                                                          (int)nnz, 0, key_bits, s));
                sort_tmp.alloc(tmp_bytes);
                MI_HIP(rocprim::radix_sort_pairs(sort_tmp.ptr, tmp_bytes, h->csr_idx.ptr, key_out.ptr, cell_in.ptr, cell_out.ptr,
                                                          (int)nnz, 0, key_bits, s));
                hipLaunchKernelGGL(split_cells_kernel, dim3(eg), dim3(eb), 0, s, cell_out.ptr, nnz, h->csc_idx.ptr, h->csc_val.ptr);
            }
            hipLaunchKernelGGL(csc_ptr_kernel, dim3(div_up(n_cols + 1, 256)), dim3(256), 0, s, key_out.ptr, nnz, n_cols, h->csc_ptr.ptr);
            MI_HIP(hipGetLastError());
        }

        phase(walk_only ? "walk lists = the column view (slices, radix sort of the entries)" : "CSR -> CSC (allocations, radix sort of the cells)");
        const int cg = div_up((int64_t)n_cols * 64, 256);
        if (cfg->similarity == MI355REC_SIM_PEARSON) {
            mean.alloc((size_t)n_cols);
            hipLaunchKernelGGL(segment_mean_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, s, h->csc_ptr.ptr, h->csc_val.ptr, n_cols,
                               mean.ptr);
            hipLaunchKernelGGL(col_center_kernel, dim3(eg), dim3(eb), 0, s, h->csr_idx.ptr, h->csr_val.ptr, nnz, mean.ptr);
            hipLaunchKernelGGL(col_center_csc_kernel, dim3(cg), dim3(256), 0, s, h->csc_ptr.ptr, h->csc_val.ptr, n_cols, mean.ptr);
        }
        sumsq.alloc((size_t)n_cols);
        cost.alloc((size_t)n_cols);
        MI_HIP(hipMemsetAsync(cost.ptr, 0, sizeof(long long) * (size_t)n_cols, s));
        DeviceBuffer<int> user_count;
        DeviceBuffer<unsigned long long> packed;
        if (walk_only) {
            // cost and users of every column from the sorted entries: the rows' lengths filed under their first slices (tiled: rows)
            const size_t n_walk = (size_t)n_walk_entries;
            packed.alloc_zero((size_t)n_cols, s);
            user_count.alloc((size_t)n_cols);
            hipLaunchKernelGGL(column_cost_kernel<true>, dim3(div_up((int64_t)div_up((int64_t)n_walk, COST_CHUNK) * 64, 256)), dim3(256), 0, s, walk_keys.ptr,
                               h->walk4.ptr, h->n_tiles > 1 ? h->csr_ptr.ptr : walk_len_ptr.ptr, n_walk, packed.ptr);
            hipLaunchKernelGGL(walk_unpack_cost_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, s, packed.ptr, n_cols, cost.ptr, sumsq.ptr, user_count.ptr);
        } else
        hipLaunchKernelGGL(column_cost_kernel<false>, dim3(div_up((int64_t)div_up((int64_t)nnz, COST_CHUNK) * 64, 256)), dim3(256), 0, s, key_out.ptr,
                           h->csc_idx.ptr, h->csr_ptr.ptr, nnz, reinterpret_cast<unsigned long long *>(cost.ptr));
        MI_REQUIRE(cfg->norm_sum_order == 0 || cfg->norm_sum_order == 1, "norm_sum_order must be 0 (CSR order) or 1 (CSC order)");
        if (walk_only) {
            // (sumsq = the users counted above)
        } else if (h->unit_values && n_rows < (1 << 24))
            hipLaunchKernelGGL(column_count_sumsq_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, s, h->csc_ptr.ptr, n_cols, sumsq.ptr);
        else if (cfg->norm_sum_order == 0)
            hipLaunchKernelGGL(column_sumsq_f32_rowwise_kernel, dim3(div_up((int64_t)n_cols * 64, 256)), dim3(256), 0, s, h->csc_ptr.ptr,
                               h->csc_val.ptr, n_cols, sumsq.ptr);
        else
            hipLaunchKernelGGL(column_sumsq_f32_kernel, dim3(div_up(n_cols, 64)), dim3(64), 0, s, h->csc_ptr.ptr, h->csc_val.ptr, n_cols,
                               cfg->norm_sum_order, sumsq.ptr);
        h->norm.alloc_zero((size_t)n_cols + NORM_PAD, s);
        const bool asym = cfg->similarity == MI355REC_SIM_ASYMMETRIC;
        if (euclid) h->norm_alpha.alloc_zero((size_t)n_cols + NORM_PAD, s);     // sums of squares
        if (asym) {
            h->norm_alpha.alloc_zero((size_t)n_cols + NORM_PAD, s);
            h->norm_1ma.alloc_zero((size_t)n_cols + NORM_PAD, s);
        }
        hipLaunchKernelGGL(norms_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, s, sumsq.ptr, n_cols, (int)set_based,
                           (int)asym, (int)euclid, cfg->asymmetric_alpha, h->norm.ptr, h->norm_alpha.ptr, h->norm_1ma.ptr);
        MI_HIP(hipGetLastError());
        // columns by descending cost (stable: ties keep ascending column ids), sorted where the costs are
        DeviceBuffer<long long> cost_sorted;
        DeviceBuffer<int> col_ids, col_order;
        DeviceBuffer<char> order_tmp;
        cost_sorted.alloc((size_t)n_cols); col_ids.alloc((size_t)n_cols); col_order.alloc((size_t)n_cols);
        hipLaunchKernelGGL(iota_kernel, dim3(div_up(n_cols, 256)), dim3(256), 0, s, col_ids.ptr, n_cols);
        size_t order_bytes = 0;
        int cost_bits = 1;                          // a column's cost counts every stored cell at most once: cost <= nnz
        while ((1ull << cost_bits) <= (unsigned long long)nnz) ++cost_bits;
        MI_HIP(rocprim::radix_sort_pairs_desc(nullptr, order_bytes, cost.ptr, cost_sorted.ptr, col_ids.ptr, col_order.ptr, (size_t)n_cols, 0, cost_bits, s));
        order_tmp.alloc(order_bytes + 16);
        MI_HIP(rocprim::radix_sort_pairs_desc(order_tmp.ptr, order_bytes, cost.ptr, cost_sorted.ptr, col_ids.ptr, col_order.ptr, (size_t)n_cols, 0, cost_bits, s));
        h->cost_order.resize(n_cols);
        col_order.download(h->cost_order.data(), (size_t)n_cols, s);
        h->cost.resize(n_cols);
        cost.download(h->cost.data(), n_cols, s);
        h->csc_ptr_host.resize((size_t)n_cols + 1);
        if (walk_only) {
            user_count.download(h->csc_ptr_host.data() + 1, (size_t)n_cols, s);
            MI_HIP(hipStreamSynchronize(s));
            h->csc_ptr_host[0] = 0;
            for (int c = 0; c < n_cols; ++c) h->csc_ptr_host[c + 1] += h->csc_ptr_host[c];
            MI_REQUIRE((size_t)h->csc_ptr_host[n_cols] == nnz, "walk lists: %d users counted for %zu stored values", h->csc_ptr_host[n_cols], nnz);
        } else {
            h->csc_ptr.download(h->csc_ptr_host.data(), (size_t)n_cols + 1, s);
            MI_HIP(hipStreamSynchronize(s));
        }
        phase("column costs + norms + downloads");

        // lanes per user profile: sized to the profile length seen from an item (cost-weighted mean)
        {
            long long total_cost = 0;
            for (long long c : h->cost) total_cost += c;
            const double weighted_len = (double)total_cost / (double)nnz;
            // each lane covers 8 profile entries per load: G lanes span 8*G entries
            // (with the walk lists a lane group never sees more than WALK_SLICE chunks at once and the groups of a round get slices of
            // the same length: 8 lanes per slice are fastest at every shape measured -- ML-20M shape 3.76 ms against 3.77 / 3.91 / 4.48 with
            // 16 / 32 / 64, Netflix shape 15.7 against 16.2 / 17.8 with 16 / 32, star ratings 5.16 against 5.31 / 5.88 / 7.39)
            (void)weighted_len;
            h->group_lanes = 8;
            // (where the packed-counts kernel will run -- see run_columns_lds -- sixteen: 3.03 against 3.10 ms at ML-20M shape)
            if (h->acc_mode() == ACC_COUNTS && h->n_tiles == 1 && (double)total_cost < PACKED_MAX_PAIRS_PER_COLUMN * (double)n_cols) h->group_lanes = 16;
            // the float64 kernel has half the loads in flight per lane (DEPTH 2)
            if (h->acc_mode() == ACC_WIDE) h->group_lanes = 16;
            if (getenv("MI355REC_SIM_G")) h->group_lanes = atoi(getenv("MI355REC_SIM_G"));
            MI_REQUIRE(h->group_lanes == 4 || h->group_lanes == 8 || h->group_lanes == 16 || h->group_lanes == 32 || h->group_lanes == 64,
                       "MI355REC_SIM_G must be 4, 8, 16, 32 or 64");
        }
        if (!walk_only) build_seg_ptr();
        fill_stream();
        phase("profile stream");
        if (!walk_only) {
            build_walk();
            phase("walk lists");
        }

        // Real-valued data: can the column sums be kept as int64 fixed point (ds_add_u64 is 1.8x faster than ds_add_f64)?
        // Every product is at most P = max weight * max |column-side value| * max |value|; a cell sums at most N = longest
        // column of them.  Scale 2^S with P * 2^S <= 2^50 (the float64 rounding trick needs |x| < 2^51) and N * P * 2^S <= 2^62.
        // A cell is then off by at most N / 2 units of 2^-S; the smallest denominator it can meet is the smallest non-zero
        // column norm squared (normalised similarities) -- accept when that WORST-CASE error stays below 1e-6 (a tenth of the
        // parity bar; rounding errors of random sign add up to ~sqrt(N), not N), otherwise keep float64.
        if (h->acc_mode() == ACC_WIDE && !getenv("MI355REC_SIM_F64_SUMS")) {
            DeviceBuffer<unsigned> d_vmax;
            d_vmax.alloc_zero(1, s);
            hipLaunchKernelGGL(absmax_kernel, dim3(eg), dim3(eb), 0, s, h->csc_val.ptr, nnz, d_vmax.ptr);
            MI_HIP(hipGetLastError());
            unsigned vbits = 0;
            d_vmax.download(&vbits, 1, s);
            std::vector<double> sq((size_t)n_cols);
            sumsq.download(sq.data(), (size_t)n_cols, s);
            MI_HIP(hipStreamSynchronize(s));
            float vmax_f;
            memcpy(&vmax_f, &vbits, sizeof(float));
            const double vmax = (double)vmax_f;
            double wmax = 1.0;
            if (row_weights)
                for (int r = 0; r < n_rows; ++r) wmax = std::max(wmax, (double)std::fabs(row_weights[r]));
            double longest = 1.0, min_sq = 0.0;
            for (int c = 0; c < n_cols; ++c) {
                longest = std::max(longest, (double)(h->csc_ptr_host[c + 1] - h->csc_ptr_host[c]));
                if (sq[c] > 0.0 && (min_sq == 0.0 || sq[c] < min_sq)) min_sq = sq[c];
            }
            const double prod = wmax * (cfg->unit_column_side ? 1.0 : vmax) * vmax;
            if (prod > 0.0 && std::isfinite(prod)) {
                const int S = (int)std::floor(std::min(50.0 - std::log2(prod), 62.0 - std::log2(prod * longest)));
                const double unit = std::ldexp(1.0, -S);
                // denominators: norm_c * norm_j >= min_sq (normalised); otherwise the results are the sums themselves, whose
                // scale is at least the smallest non-zero product -- bounded below by min_sq as well only for single-cell
                // columns, so the same bar is applied (conservative for everything else)
                const double worst = 0.5 * longest * unit / std::max(min_sq, 1e-300);
                if (S > -1000 && S < 1000 && worst <= 1e-6) h->fixed_scale = std::ldexp(1.0, S);
            }
        }

        // the column view has served (norms, costs, value range): the accumulation walks the lists above
        h->csc_idx.release();
        h->csc_val.release();
        h->queue.alloc(4);
        phase("fixed-point check + cost order (host)");
        *out = h.release();
    });
}

extern "C" int mi355rec_sim_create(mi355rec_sim_t *out, const mi355rec_sim_config *cfg, int32_t n_rows, int32_t n_cols,
                                   const int32_t *csr_indptr, const int32_t *csr_indices, const float *csr_data,
                                   const float *row_weights) {
    return sim_create_from(out, cfg, n_rows, n_cols, csr_indptr, csr_indices, csr_data, row_weights, false);
}

extern "C" int mi355rec_sim_create_resident(mi355rec_sim_t *out, const mi355rec_sim_config *cfg, int32_t n_rows, int32_t n_cols,
                                            const int32_t *d_csr_indptr, const int32_t *d_csr_indices, const float *d_csr_data,
                                            const float *row_weights) {
    return sim_create_from(out, cfg, n_rows, n_cols, d_csr_indptr, d_csr_indices, d_csr_data, row_weights, true);
}

extern "C" int mi355rec_sim_compute_device(mi355rec_sim_t h, int32_t start_col, int32_t end_col, int32_t *d_nbr_idx,
                                           float *d_nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && d_nbr_idx && d_nbr_val, "NULL argument");
        MI_REQUIRE(h->cfg.topK > 0, "topK == 0: use mi355rec_sim_compute_dense");
        ensure_device();
        clamp_range(h, start_col, end_col);
        ReleaseScope scope(h->stream);
        run_columns(h, start_col, end_col, d_nbr_idx, d_nbr_val, nullptr);
    });
}

extern "C" int mi355rec_sim_compute_part_device(mi355rec_sim_t h, int32_t part, int32_t n_parts, int32_t *d_nbr_idx, float *d_nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && d_nbr_idx && d_nbr_val, "NULL argument");
        MI_REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts, "part %d of %d", part, n_parts);
        if (h->cfg.topK == 0) fail(MI355REC_E_INVALID, "topK == 0: use mi355rec_sim_compute_dense");
        ensure_device();
        run_columns(h, part, 0, d_nbr_idx, d_nbr_val, nullptr, n_parts);
    });
}

extern "C" int mi355rec_sim_compute_part_chunk_device(mi355rec_sim_t h, int32_t part, int32_t n_parts, int32_t slot_first, int32_t slot_count,
                                                      int32_t *d_nbr_idx, float *d_nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && d_nbr_idx && d_nbr_val, "NULL argument");
        MI_REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts, "part %d of %d", part, n_parts);
        MI_REQUIRE(slot_first >= 0 && slot_count >= 0, "rows %d + %d of a part", slot_first, slot_count);
        if (h->cfg.topK == 0) fail(MI355REC_E_INVALID, "topK == 0: use mi355rec_sim_compute_dense");
        ensure_device();
        if (slot_count == 0) return;
        h->wide_kernel_ms = -1.0;
        // (topK beyond the in-LDS selection, or more per-tile candidates than the merge buffer holds: dense columns + segmented sort,
        // walking the same rows of the part)
        if (h->wide_topk) run_columns_wide_topk(h, part, 0, d_nbr_idx, d_nbr_val, n_parts, slot_first, slot_count);
        else run_columns_lds(h, part, 0, d_nbr_idx, d_nbr_val, nullptr, n_parts, slot_first, slot_count);
    });
}

// The 6-byte cells of the sharded build's exchange (n_cols <= 65 535): n_cells float32 values, then n_cells 16-bit neighbour ids
// (0xFFFF = the empty slot's -1), the whole padded to 4-byte words.  Two cells per thread: one packed id word per store.
__global__ void sim_pack_slab_kernel(const int *idx, const float *val, long long n_cells, float *out_val, unsigned *out_ids) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;          // pair number
    const long long q = 2 * p;
    if (q >= n_cells) return;
    const bool two = q + 1 < n_cells;
    const unsigned lo = (unsigned)idx[q] & 0xFFFFu, hi = two ? ((unsigned)idx[q + 1] & 0xFFFFu) : 0xFFFFu;
    out_val[q] = val[q];
    if (two) out_val[q + 1] = val[q + 1];
    out_ids[p] = lo | (hi << 16);
}
__global__ void sim_unpack_slab_kernel(const float *in_val, const unsigned *in_ids, long long n_cells, int *idx, float *val) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long q = 2 * p;
    if (q >= n_cells) return;
    const unsigned w = in_ids[p];
    const unsigned lo = w & 0xFFFFu, hi = w >> 16;
    idx[q] = lo == 0xFFFFu ? -1 : (int)lo;
    val[q] = in_val[q];
    if (q + 1 < n_cells) {
        idx[q + 1] = hi == 0xFFFFu ? -1 : (int)hi;
        val[q + 1] = in_val[q + 1];
    }
}

extern "C" int mi355rec_sim_pack_slab_device(mi355rec_sim_t h, const int32_t *d_nbr_idx, const float *d_nbr_val, int64_t n_cells, void *d_packed) {
    return guarded([&] {
        MI_REQUIRE(h && d_nbr_idx && d_nbr_val && d_packed, "NULL argument");
        MI_REQUIRE(n_cells >= 0, "n_cells = %lld", (long long)n_cells);
        MI_REQUIRE(h->n_cols <= 65535, "n_cols = %d: neighbour ids do not fit 16 bits", h->n_cols);
        ensure_device();
        if (n_cells == 0) return;
        const long long pairs = (n_cells + 1) / 2;
        hipLaunchKernelGGL(sim_pack_slab_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, d_nbr_idx, d_nbr_val, (long long)n_cells,
                           (float *)d_packed, (unsigned *)d_packed + n_cells);
        MI_HIP(hipGetLastError());
    });
}

extern "C" int mi355rec_sim_unpack_slab_device(mi355rec_sim_t h, const void *d_packed, int64_t n_cells, int32_t *d_nbr_idx, float *d_nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && d_nbr_idx && d_nbr_val && d_packed, "NULL argument");
        MI_REQUIRE(n_cells >= 0, "n_cells = %lld", (long long)n_cells);
        ensure_device();
        if (n_cells == 0) return;
        const long long pairs = (n_cells + 1) / 2;
        hipLaunchKernelGGL(sim_unpack_slab_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, (const float *)d_packed,
                           (const unsigned *)d_packed + n_cells, (long long)n_cells, d_nbr_idx, d_nbr_val);
        MI_HIP(hipGetLastError());
    });
}

extern "C" int mi355rec_sim_part_columns(mi355rec_sim_t h, int32_t part, int32_t n_parts, int32_t *columns, int32_t *n_columns) {
    return guarded([&] {
        MI_REQUIRE(h && n_columns, "NULL argument");
        MI_REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts, "part %d of %d", part, n_parts);
        int n = 0;
        for (long long pos = 0; pos < h->n_cols; ++pos)
            if (part_of_position(pos, n_parts) == part) {
                if (columns) columns[n] = h->cost_order[pos];
                ++n;
            }
        *n_columns = n;
    });
}

extern "C" int mi355rec_sim_get_weighted_values(mi355rec_sim_t h, float *csr_data) {
    return guarded([&] {
        MI_REQUIRE(h && csr_data, "NULL argument");
        MI_REQUIRE(h->weighted_val.ptr, "the handle was created without feature_weighting");
        ensure_device();
        h->weighted_val.download(csr_data, h->nnz, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_sim_compute(mi355rec_sim_t h, int32_t start_col, int32_t end_col, int32_t *nbr_idx, float *nbr_val) {
    return guarded([&] {
        MI_REQUIRE(h && nbr_idx && nbr_val, "NULL argument");
        MI_REQUIRE(h->cfg.topK > 0, "topK == 0: use mi355rec_sim_compute_dense");
        ensure_device();
        clamp_range(h, start_col, end_col);
        ReleaseScope scope(h->stream);
        const size_t n = (size_t)(end_col - start_col) * h->cfg.topK;
        if (h->out_idx.count < n) {
            h->out_idx.alloc(n);
            h->out_val.alloc(n);
        }
        run_columns(h, start_col, end_col, h->out_idx.ptr, h->out_val.ptr, nullptr);
        h->out_idx.download(nbr_idx, n, h->stream);
        h->out_val.download(nbr_val, n, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
        read_timers(h);
        if (h->phase_ticks.ptr && getenv("MI355REC_SIM_PHASES")) {
            unsigned long long t[12], w[16];
            h->phase_ticks.download(w, 16, h->stream);
            h->selection_counts.download(t + 8, 4, h->stream);
            MI_HIP(hipStreamSynchronize(h->stream));
            memcpy(t, w, 8 * sizeof(unsigned long long));
            fprintf(stderr, "[mi355rec sim spans] first start to last end %.3f ms; workgroups' own spans %.2f workgroup-ms; longest work item %.3f ms (column %llu)\n",
                    (double)(w[9] - w[8]) * 1e-5, (double)w[10] * 1e-5, (double)w[11] * 1e-5, w[12]);
            fprintf(stderr, "[mi355rec sim phases, workgroup-ms] fetch+clear %.2f  accumulate %.2f  split-merge %.2f  normalise %.2f  topk %.2f  (kernel %.3f ms)"
                            "  threshold-first: maxima scan %.2f  (K-th maximum under `normalise`)  survivor scan %.2f  (exact values + rank + emit under `topk`)"
                            "  columns %llu (candidates %.1f per column), full-selection fall-backs after the scan %llu (%llu: buffer full); wait for the scan's slowest wavefront %.2f (instrumented runs only)\n",
                    t[0] * 1e-5, t[1] * 1e-5, t[2] * 1e-5, t[3] * 1e-5, t[4] * 1e-5, h->stats.kernel_ms, t[5] * 1e-5, t[6] * 1e-5, t[8],
                    t[8] ? (double)t[9] / (double)t[8] : 0.0, t[10], t[11], t[7] * 1e-5);
        }
    });
}

extern "C" int mi355rec_sim_compute_csr(mi355rec_sim_t h, int32_t start_col, int32_t end_col, int32_t *indptr, int32_t *indices,
                                        float *data, int64_t *nnz_out) {
    return guarded([&] {
        MI_REQUIRE(h && indptr && indices && data && nnz_out, "NULL argument");
        MI_REQUIRE(h->cfg.topK > 0, "topK == 0: use mi355rec_sim_compute_dense");
        ensure_device();
        clamp_range(h, start_col, end_col);
        ReleaseScope scope(h->stream);
        const size_t n = (size_t)(end_col - start_col) * h->cfg.topK;
        MI_REQUIRE(n < (size_t)INT32_MAX, "result too large for 32-bit CSR offsets");
        if (h->out_idx.count < n) {
            h->out_idx.alloc(n);
            h->out_val.alloc(n);
        }
        if (h->csr_key.count < n) {
            h->csr_key.alloc(n); h->csr_key_sorted.alloc(n); h->csr_pos.alloc(n); h->csr_pos_sorted.alloc(n);
            h->csr_indices.alloc(n); h->csr_data.alloc(n);
        }
        if (!h->csr_indptr.ptr) h->csr_indptr.alloc((size_t)h->n_cols + 1);
        hipStream_t s = h->stream;
        run_columns(h, start_col, end_col, h->out_idx.ptr, h->out_val.ptr, nullptr);
        const int eb = 256, eg = (int)std::min<size_t>((n + eb - 1) / eb, 4096);
        hipLaunchKernelGGL(csr_keys_kernel, dim3(eg), dim3(eb), 0, s, h->out_idx.ptr, n, h->n_cols, h->csr_key.ptr, h->csr_pos.ptr);
        int key_bits = 1;
        while ((1ll << key_bits) < (long long)h->n_cols + 1) ++key_bits;
        size_t tmp_bytes = 0;
        MI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->csr_key.ptr, h->csr_key_sorted.ptr, h->csr_pos.ptr,
                                                  h->csr_pos_sorted.ptr, (int)n, 0, key_bits, s));
        if (h->csr_sort_tmp.count < tmp_bytes) h->csr_sort_tmp.alloc(tmp_bytes);
        MI_HIP(rocprim::radix_sort_pairs(h->csr_sort_tmp.ptr, tmp_bytes, h->csr_key.ptr, h->csr_key_sorted.ptr,
                                                  h->csr_pos.ptr, h->csr_pos_sorted.ptr, (int)n, 0, key_bits, s));
        // indptr[r] = first sorted position whose key is >= r; indptr[n_cols] = number of real entries (padding sorts last)
        hipLaunchKernelGGL(csc_ptr_kernel, dim3(div_up(h->n_cols + 1, 256)), dim3(256), 0, s, h->csr_key_sorted.ptr, n, h->n_cols,
                           h->csr_indptr.ptr);
        hipLaunchKernelGGL(csr_gather_kernel, dim3(eg), dim3(eb), 0, s, h->csr_pos_sorted.ptr, h->out_val.ptr, n, h->cfg.topK,
                           start_col, h->csr_indices.ptr, h->csr_data.ptr);
        MI_HIP(hipGetLastError());
        h->csr_indptr.download(indptr, (size_t)h->n_cols + 1, s);
        MI_HIP(hipStreamSynchronize(s));
        const size_t nnz = (size_t)indptr[h->n_cols];
        *nnz_out = (int64_t)nnz;
        h->csr_indices.download(indices, nnz, s);
        h->csr_data.download(data, nnz, s);
        MI_HIP(hipStreamSynchronize(s));
        read_timers(h);
    });
}

extern "C" int mi355rec_sim_compute_dense(mi355rec_sim_t h, int32_t start_col, int32_t end_col, float *W, int64_t ld) {
    return guarded([&] {
        MI_REQUIRE(h && W, "NULL argument");
        ensure_device();
        clamp_range(h, start_col, end_col);
        ReleaseScope scope(h->stream);
        const int n_local = end_col - start_col;
        MI_REQUIRE(ld >= n_local, "ld (%lld) < number of columns (%d)", (long long)ld, n_local);
        mi355rec_sim_config saved = h->cfg;
        h->cfg.topK = 0;
        DeviceBuffer<float> slab, slab_t;
        slab.alloc((size_t)n_local * h->n_cols);
        slab_t.alloc((size_t)n_local * h->n_cols);
        try {
            run_columns(h, start_col, end_col, nullptr, nullptr, slab.ptr);
        } catch (...) {
            h->cfg = saved;
            throw;
        }
        h->cfg = saved;
        hipLaunchKernelGGL(transpose_kernel, dim3(div_up(h->n_cols, 32), div_up(n_local, 32)), dim3(32, 8), 0, h->stream,
                           slab.ptr, slab_t.ptr, n_local, h->n_cols);
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpy2DAsync(W, (size_t)ld * sizeof(float), slab_t.ptr, (size_t)n_local * sizeof(float),
                                (size_t)n_local * sizeof(float), (size_t)h->n_cols, hipMemcpyDeviceToHost, h->stream));
        MI_HIP(hipStreamSynchronize(h->stream));
        read_timers(h);
    });
}

extern "C" int mi355rec_sim_column_costs(mi355rec_sim_t h, int64_t *cost) {
    return guarded([&] {
        MI_REQUIRE(h && cost, "NULL argument");
        for (int c = 0; c < h->n_cols; ++c) cost[c] = h->cost[c];
    });
}

extern "C" int mi355rec_sim_schedule_info(mi355rec_sim_t h, int32_t *n_items, int32_t *n_split_columns, int32_t *n_parts) {
    return guarded([&] {
        MI_REQUIRE(h && n_items && n_split_columns && n_parts, "NULL argument");
        *n_items = (int32_t)h->items_host.size();
        *n_split_columns = h->n_split_columns;
        *n_parts = h->n_part_items;
    });
}

extern "C" int mi355rec_sim_selection_info(mi355rec_sim_t h, int64_t *threshold_first_columns, int64_t *candidates, int64_t *fallbacks) {
    return guarded([&] {
        MI_REQUIRE(h && threshold_first_columns && candidates && fallbacks, "NULL argument");
        unsigned long long t[4] = {0, 0, 0, 0};
        if (h->selection_counts.ptr) {
            ensure_device();
            h->selection_counts.download(t, 4, h->stream);
            MI_HIP(hipStreamSynchronize(h->stream));
        }
        *threshold_first_columns = (int64_t)t[0];
        *candidates = (int64_t)t[1];
        *fallbacks = (int64_t)t[2];
    });
}

extern "C" int mi355rec_sim_accumulator_info(mi355rec_sim_t h, int32_t *kind, double *fixed_scale) {
    return guarded([&] {
        MI_REQUIRE(h && kind && fixed_scale, "NULL argument");
        const int mode = h->acc_mode();
        *kind = mode == ACC_COUNTS ? 0 : (mode == ACC_INT32 ? 3 : (h->fixed_scale > 0.0 ? 1 : 2));
        *fixed_scale = mode == ACC_COUNTS ? 0.0 : (mode == ACC_INT32 ? (double)(1 << (2 * h->int_shift)) : h->fixed_scale);
    });
}

// ---- the bound the column kernel is priced against, measured on the device at hand -----------------------------------------
// ds_add_u32 lane-adds per second of the whole device: one 1024-thread workgroup per CU adding to pseudo-random cells of a 128 KiB LDS
// array (uniformly random addresses: the bank conflicts of a random scatter are part of the figure; scripts/micro/lds_atomics.hip is
// the same loop stand-alone, 21.6 lane-adds per ns and CU in round 1).
namespace {
__global__ __launch_bounds__(1024) void lds_atomic_rate_kernel(int iters, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned rate_cells[];
    constexpr unsigned CELLS = 128 * 1024 / sizeof(unsigned);
    for (unsigned i = threadIdx.x; i < CELLS; i += 1024) rate_cells[i] = 0u;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            atomicAdd(&rate_cells[(s >> 8) % CELLS], s & 7u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = rate_cells[blockIdx.x % CELLS];
}
}  // namespace

extern "C" int mi355rec_lds_atomic_rate(double *lane_adds_per_second) {
    return guarded([&] {
        MI_REQUIRE(lane_adds_per_second, "NULL argument");
        ensure_device();
        const int cus = multiprocessor_count(), iters = 2048;
        DeviceBuffer<unsigned> sink;
        sink.alloc((size_t)cus);
        auto kern = lds_atomic_rate_kernel;
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipEvent_t a = pooled_event(), b = pooled_event();
        hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), 128 * 1024, 0, 64, sink.ptr);             // warm-up
        double best = 0.0;
        for (int rep = 0; rep < 3; ++rep) {
            MI_HIP(hipEventRecord(a, 0));
            hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), 128 * 1024, 0, iters, sink.ptr);
            MI_HIP(hipEventRecord(b, 0));
            MI_HIP(hipEventSynchronize(b));
            float ms = 0.f;
            MI_HIP(hipEventElapsedTime(&ms, a, b));
            if (ms > 0.f) best = std::max(best, (double)cus * 1024.0 * iters * 8.0 / (ms * 1e-3));
        }
        MI_HIP(hipGetLastError());
        pooled_event_return(a);
        pooled_event_return(b);
        *lane_adds_per_second = best;
    });
}

extern "C" int mi355rec_sim_sync(mi355rec_sim_t h) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_HIP(hipStreamSynchronize(h->stream));
        if (h->last_start >= 0) read_timers(h);
    });
}

extern "C" int mi355rec_sim_get_stats(mi355rec_sim_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_sim_destroy(mi355rec_sim_t h) {
    if (!h) return;
    ReleaseScope scope(h->stream);
    delete h;
}
