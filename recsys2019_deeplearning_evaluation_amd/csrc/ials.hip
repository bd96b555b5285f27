// ials.hip -- IALS solve step on MI355X (gfx950).
//
// Replaces MatrixFactorization/IALSRecommender.py (reference, pure NumPy): _run_epoch :137-166 (user pass against
// V and V^T V, then item pass against the UPDATED U and U^T U) and _update_row :170-201
//      x = inv(YtY + Y_I^T (C_I - 1) Y_I + reg I) . (Y_I^T c_I).
//
// Design.  This is the one dense contraction on the hot path; it is done in float64 like the
// reference (the systems are ill-conditioned for small `reg`, fp32 cannot meet the 1e-5 parity bar), on the FP64 MATRIX pipe
// (v_mfma_f64_16x16x4_f64; on CDNA4 its peak equals the FP64 vector peak, 78.6 TFLOP/s, but one instruction carries 512
// multiply-adds and needs 2 LDS operand reads, against 14 reads per 28 multiply-adds on the vector pipe).
//   gram_kernel      G = Y^T Y, register tiles + double atomics (tiny: n * k^2).
//   ials_row_kernel  one 512-thread workgroup per row, rows pulled longest-profile-first from a queue; see the comment above
//                    the kernel: augmented system in 16 x 16 register tiles, Gramian and trailing updates as MFMAs, blocked
//                    Cholesky with an in-register diagonal-tile factorisation, back substitution through the tile inverses.
#include "common.h"
#include "ials_diag.cuh"

#include <algorithm>
#include <memory>
#include <numeric>

namespace mi355rec {
namespace {

constexpr int CHUNK = 16;   // profile rows staged per LDS round
#ifndef MI355REC_IALS_GRAM_CHUNK
#define MI355REC_IALS_GRAM_CHUNK 32
#endif
constexpr int GRAM_CHUNK = MI355REC_IALS_GRAM_CHUNK;   // ... by the Gramian stage of a two-stage epoch

struct IalsParams {
    int k;
    double reg;
    const int *ptr, *idx;      // sparse rows being solved (users: C as CSR; items: C as CSC)
    const float *conf;
    const double *Y;           // fixed side factors (n_other x k)
    const double *G;           // Y^T Y (k x k)
    double *X;                 // factors being solved (n x k)
    const int4 *items;         // work items of this call, most expensive first: {row, part, n_parts, first part slot}
    int n_local;               // number of work items
    double *part_buf;          // [part slots][SLOTS * 4 * ROW_THREADS] partial augmented Gramians of split rows (accumulator layout)
    unsigned *part_count;      // arrival counters, indexed by the first part slot of a split row
    unsigned *queue;
    double *systems;           // two-stage epochs: [rows of the batch][SLOTS * 4 * ROW_THREADS] augmented systems (accumulator layout)
    const int *sys_slot;       // two-stage epochs: slab of every work item's row in `systems`
    int same_panel_wave;       // diagnostics (MI355REC_IALS_SAME_PANEL_WAVE=1): every workgroup's panel wavefront is wavefront 0
    unsigned long long *phases;   // optional (MI355REC_IALS_PHASES=1): shader-clock totals of {base, Gramian, Cholesky, back substitution}, rows
};

__device__ __forceinline__ unsigned long long ials_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// G += Y[rows]^T Y[rows]; 256 threads as a 16 x 16 grid, thread owns G[ty + 16 (a0 + a)][tx + 16 (b0 + b)], a, b < KT16: the
// whole matrix in one launch (a0 = b0 = 0, KT16 = tiles per side; up to 14 x 14 cells = 392 registers per thread), or -- 15 and 16
// tiles per side, k > 224 -- one launch per 8 x 8 quadrant.
template <int KT16>
__global__ __launch_bounds__(256) void gram_kernel(const double *Y, int n, int k, double *G, int a0, int b0) {
    __shared__ double ys[8][256];
    const int tid = threadIdx.x, ty = tid >> 4 | a0 << 4, tx = (tid & 15) | b0 << 4;
    double acc[KT16][KT16];
#pragma unroll
    for (int a = 0; a < KT16; ++a)
#pragma unroll
        for (int b = 0; b < KT16; ++b) acc[a][b] = 0.0;
    const int rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
    for (int base = r0; base < r1; base += 8) {
        const int nr = min(8, r1 - base);
        __syncthreads();
        for (int e = tid; e < 8 * 256; e += 256) {
            const int r = e >> 8, f = e & 255;
            ys[r][f] = (r < nr && f < k) ? Y[(size_t)(base + r) * k + f] : 0.0;
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            double ya[KT16], yb[KT16];
#pragma unroll
            for (int a = 0; a < KT16; ++a) {
                ya[a] = ys[r][(ty + 16 * a) & 255];        // (tiles past the 16th do not exist: masked at the end)
                yb[a] = ys[r][(tx + 16 * a) & 255];
            }
#pragma unroll
            for (int a = 0; a < KT16; ++a)
#pragma unroll
                for (int b = 0; b < KT16; ++b) acc[a][b] += ya[a] * yb[b];
        }
    }
#pragma unroll
    for (int a = 0; a < KT16; ++a)
#pragma unroll
        for (int b = 0; b < KT16; ++b) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < k && c < k && r < 256 && c < 256) atomicAdd(&G[(size_t)r * k + c], acc[a][b]);
        }
}

// ---- the row solver ---------------------------------------------------------------------------------------------------------
// One 512-thread workgroup (8 wavefronts) per row.  The augmented system
//        [ B    rhs ]      B = YtY + Y_I^T (C_I - 1) Y_I + reg I,  rhs = Y_I^T c_I
//        [ rhs^T  * ]
// lives in REGISTERS as 16 x 16 tiles of its lower triangle in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l,
// element i holds row 4 i + l / 16, column l % 16; checked on the device, scripts/micro/mfma_f64.hip); tile t belongs to
// wavefront t % 8 (up to 15 tiles = 120 VGPRs per wavefront at k = 224).  Row k of the augmented matrix carries rhs, so its Cholesky row IS the forward substitution.
//   (1) tiles = G + reg I;
//   (2) Gramian: profile rows staged through LDS 16 at a time, every tile takes one MFMA per 4 profile rows
//       (A operand (c - 1) y, B operand y; the rhs row takes c instead of (c - 1) y) -- the FP64 matrix pipe, 2 LDS reads per
//       512 multiply-adds instead of 14 per 28;
//   (3) blocked right-looking Cholesky, 16-column panels: panel tiles -> LDS; the diagonal tile is factored by 16 lanes of one
//       wavefront (a row each, rows broadcast with v_readlane: no barriers inside); triangular solve of the panel, one lane per
//       row; trailing update tile -= X_I X_J^T on the matrix pipe.  4 barriers per panel (52 at k = 200) instead of one per
//       column twice over (400);
//   (4) back substitution by tile rows: x_I from the diagonal tile (16 lanes), then every tile (I, J < I) subtracts its
//       L^T x_I from segment J.
// Round 1 kept 32 x 32 register tiles on the FP64 vector pipe with one barrier per column: VALU-issue bound in the
// factorisation (2240 cycles per column: 28 multiply-adds among ~180 instructions on 16 wavefronts) and staging-latency bound
// in the Gramian (1440 cycles per profile row); measured 832 k cycles per user row at ML-20M shape, k = 200.
constexpr int MAX_KT = 16, MAX_NT = MAX_KT * (MAX_KT + 1) / 2;
constexpr int ROW_THREADS = 512, ROW_WAVES = ROW_THREADS / 64;   // 2 wavefronts per SIMD: 256 VGPRs for the tiles of a wavefront
constexpr double AUG_DIAG = 1e200;       // diagonal of the rhs row: keeps the last pivot positive, never used

__device__ __forceinline__ double swap_sum16(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[0] << 32) | (unsigned)lo[0]) +
           __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[1] << 32) | (unsigned)lo[1]);
}
__device__ __forceinline__ double swap_sum32(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[0] << 32) | (unsigned)lo[0]) +
           __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[1] << 32) | (unsigned)lo[1]);
}

// doubles of dynamic LDS the kernel needs for k factors
static inline size_t ials_lds_doubles(int k, int stage = 0) {
    const int KT = (k + 1 + 15) / 16, KP = KT * 16;
    const int first = stage == 2 ? KT * 16 * TP : (stage == 1 ? 2 * GRAM_CHUNK * KP : std::max(2 * CHUNK * KP, KT * 16 * TP));
    return (size_t)first + (size_t)KT * 16 * TP + (size_t)KT * 16 + 3 * (size_t)KP;   // staging | panel, Linv, z / x / acc
}

// The tile coordinates of a wavefront never change, so the compiler hoists every per-tile address computation (4 global
// addresses + 8 LDS addresses per tile) out of the row loop and then spills them: ~150 live address registers at 13 tiles.
// Passing the lane coordinates through an empty asm at the start of each phase keeps those computations where they are used.
__device__ __forceinline__ int not_invariant(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// STAGE 0: the whole row in one workgroup (Gramian, Cholesky, back substitution).
// Two-stage epochs (the default where it pays, half_step): STAGE 1 builds the augmented systems of a batch of rows and leaves them in
// HBM in the layout the MFMA accumulators have (512 consecutive doubles per store, 196 KB per row at k = 200); STAGE 2 loads them and
// solves.  The factorisation is a chain of 13 panels whose diagonal tile is factored and inverted by ONE wavefront (58 % of the
// Cholesky's cycles, measured) while the matrix pipe idles; the fused kernel holds the tiles AND the Gramian's staging in 256
// registers, one workgroup per CU, so nothing else can run meanwhile.  The solve alone fits 128 registers: two workgroups per CU, and
// one row's panel chain hides behind the other's trailing updates -- what the wide register file of a CU is for.
// THREADS: 512 for the one-kernel epoch (256 registers per lane: tiles + staging of a whole row); the Gramian stage of a two-stage epoch
// runs 1024 threads -- sixteen wavefronts with half the tiles each (6 slots at k = 200), four per SIMD instead of two: the row's
// profile is staged once, and while two wavefronts of a SIMD wait at the chunk's barriers or for their operands the other two feed the
// matrix pipe (with two per SIMD the Gramian reached 47 % of it).  (Measured and not kept: two chunks staged in LDS, chunk n + 1 written
// while chunk n is multiplied, one barrier per chunk -- 161 k cycles per user row against 99 k.)
template <int SLOTS, int STAGE, int THREADS>
__global__ __launch_bounds__(THREADS) void ials_row_kernel(const IalsParams p) {
    static_assert(STAGE == 0 || STAGE == 1, "the solve stage is ials_solve_kernel");
    static_assert(THREADS == 512 || THREADS == 1024, "CHUNK profile rows are staged by THREADS / 64 wavefronts");
    // profile rows staged per LDS round: 16; the Gramian stage (which has the LDS to itself) takes GRAM_CHUNK -- half as many barriers per row
    constexpr int CHUNK = STAGE == 1 ? GRAM_CHUNK : mi355rec::CHUNK;
    constexpr int WAVES = THREADS / 64, RPW = CHUNK / WAVES;      // profile rows a wavefront stages per chunk
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ short s_tI[MAX_NT], s_tJ[MAX_NT];
    __shared__ int s_row;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = p.k, KT = (k + 1 + 15) / 16, KP = KT * 16, NT = KT * (KT + 1) / 2;
    double *const ya = lds;                                         // [CHUNK][KP]   A side: (c - 1) y, c in column k
    double *const yb = ya + CHUNK * KP;                             // [CHUNK][KP]   B side: y
    double *const P = lds;                                          // [KT][16][TP]  panel tiles (aliases the staging area)
    double *const Linv = lds + (STAGE == 1 ? 2 * GRAM_CHUNK * KP : max(2 * CHUNK * KP, KT * 16 * TP));   // [KT][16][TP]  inverses of the factored diagonal tiles
    double *const zv = Linv + KT * 16 * TP + KT * 16;               // [KP] forward-substituted rhs
    double *const xv = zv + KP;                                     // [KP] solution
    double *const acc = xv + KP;                                    // [KP] sum of L[I][J]^T x_I over the tile rows already solved

    for (int t = tid; t < NT; t += THREADS) {                   // tile t = I (I + 1) / 2 + J of the lower triangle
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        s_tI[t] = (short)I;
        s_tJ[t] = (short)(t - I * (I + 1) / 2);
    }
    __syncthreads();
    int tIs[SLOTS], tJs[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + WAVES * s;
        tIs[s] = t < NT ? __builtin_amdgcn_readfirstlane((int)s_tI[t]) : -1;
        tJs[s] = t < NT ? __builtin_amdgcn_readfirstlane((int)s_tJ[t]) : -1;
    }
    const int Ik = k >> 4, rk = k & 15;                             // tile row and local row of the rhs row
    // staging roles: wavefront w fetches profile rows RPW w .. RPW w + RPW - 1 of a chunk, lanes across the factors (4 x 64 >= 224 + 1)
    constexpr int SPL = 4;

    for (;;) {
        __syncthreads();
        if (tid == 0) s_row = (int)atomicAdd(p.queue, 1u);
        __syncthreads();
        const int slot = s_row;
        if (slot >= p.n_local) break;
        const int4 item = p.items[slot];
        const int row = item.x;
        int beg = p.ptr[row], end = p.ptr[row + 1];
        // A row whose Gramian alone would keep one workgroup busy for a large share of the call -- the most popular item at ML-20M
        // shape: 10^5 profile rows x k^2 on ONE CU = 26 ms, every partition of the item half-step waits for it -- is split: part q of
        // n accumulates the profile rows [beg, end) below into its own tiles (part 0 starts from YtY + reg I, the others from zero),
        // publishes them, and the part that arrives last adds the parts up IN PART ORDER (the result does not depend on who is
        // last) and solves.  Nobody waits for anybody.
        if (item.z > 1) {
            const int per = (((end - beg + item.z - 1) / item.z) + CHUNK - 1) / CHUNK * CHUNK;
            beg = min(end, beg + item.y * per);
            end = min(end, beg + per);
        }
        const bool first_part = item.y == 0;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (p.phases && tid == 0) t0 = ials_stamp();

        // (2a) the first chunk of profile rows is requested before anything else; item ids and confidences run one chunk
        // further ahead than the factor rows they address (two dependent global round trips otherwise)
        constexpr size_t SYS_DOUBLES = (size_t)SLOTS * 4 * THREADS;
        d4 C[SLOTS];
        double pre[RPW][SPL];
        double pre_c[RPW], next_c[RPW];
        int next_item[RPW];
        auto fetch_ids = [&](int base) {
#pragma unroll
            for (int h = 0; h < RPW; ++h) {
                const int r = base + RPW * wave + h;
                next_item[h] = r < end ? p.idx[r] : -1;
                next_c[h] = r < end ? (double)p.conf[r] : 0.0;
            }
        };
        auto fetch_rows = [&]() {                                    // rows of the ids fetched last
#pragma unroll
            for (int h = 0; h < RPW; ++h) {
                const bool ok = next_item[h] >= 0;
                pre_c[h] = next_c[h];
                const double *src = p.Y + (size_t)(ok ? next_item[h] : 0) * k;
#pragma unroll
                for (int q = 0; q < SPL; ++q) {
                    const int f = lane + 64 * q;
                    pre[h][q] = ok && f < k ? src[f] : 0.0;
                }
            }
        };
        fetch_ids(beg);
        fetch_rows();
        fetch_ids(beg + CHUNK);

        // (1) B = YtY + reg I                                      (IALSRecommender.py:199)
        // Every lane's 4 x SLOTS cells of G are requested first, unconditionally (clamped addresses), and patched afterwards: with the
        // load inside the `r < k && c < k` test every cell waited for its own round trip to the L2 (37 k cycles per row, a quarter
        // of the Gramian stage).
        {
        const int g = not_invariant(lane >> 4), cl = not_invariant(lane & 15);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int tI = max(tIs[s], 0), tJ = max(tJs[s], 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) C[s][i] = p.G[(size_t)min(16 * tI + 4 * i + g, k - 1) * k + min(16 * tJ + cl, k - 1)];
            // (one kernel: four tiles' loads in flight at a time -- its registers are full; the Gramian stage of a two-stage epoch
            // requests all of them at once)
            if (STAGE == 0 && (s & 3) == 3) asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * tIs[s] + 4 * i + g, c = 16 * tJs[s] + cl;
                double v = C[s][i];
                if (r == c) v += p.reg;
                if (r >= k || c >= k) v = r == c ? (r == k ? AUG_DIAG : 1.0) : 0.0;   // identity padding keeps the factorisation well defined
                if (tIs[s] < 0 || !first_part) v = 0.0;
                C[s][i] = v;
            }
        }
        }
        if (p.phases && tid == 0) { asm volatile("" ::"v"(C[0][0])); t1 = ials_stamp(); }

        // (2) A = Y_I^T ((c - 1) o Y_I), rhs = Y_I^T c             (:197, :201): the matrix pipe works on chunk n while the
        // loads of chunk n + 1 are in flight
        for (int base = beg; base < end; base += CHUNK) {
            const int nr = min(CHUNK, end - base);
            __syncthreads();                                         // the previous chunk has been consumed
#pragma unroll
            for (int h = 0; h < RPW; ++h) {
                const int r = RPW * wave + h;
                const double c = pre_c[h];
#pragma unroll
                for (int q = 0; q < SPL; ++q) {
                    const int f = lane + 64 * q;
                    if (f < KP) {
                        const double y = pre[h][q];
                        ya[r * KP + f] = f == k ? c : y * (c - 1.0);     // (y is 0 beyond k and for rows past the profile)
                        yb[r * KP + f] = y;
                    }
                }
            }
            if (base + CHUNK < end) {
                fetch_rows();
                fetch_ids(base + 2 * CHUNK);
            }
            __syncthreads();
            const int groups = (nr + 3) >> 2;
            const int g = not_invariant(lane >> 4), cl = not_invariant(lane & 15);
            // all operand reads of a group of 4 profile rows first, then the MFMAs (no branch between them: a wavefront with
            // fewer tiles than slots runs its spare slot on tile (0, 0) and never looks at the result) -- otherwise every MFMA
            // waits for its own two LDS reads (117 instead of 32 cycles per MFMA, measured)
            for (int q = 0; q < groups; ++q) {
                const double *ar = ya + (4 * q + g) * KP + cl, *br = yb + (4 * q + g) * KP + cl;
                double av[SLOTS], bv[SLOTS];
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    av[s] = ar[16 * max(tIs[s], 0)];
                    bv[s] = br[16 * max(tJs[s], 0)];
                }
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) C[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], C[s], 0, 0, 0);
            }
        }
        if (item.z > 1) {
            constexpr size_t PART_DOUBLES = (size_t)SLOTS * 4 * THREADS;
            {
                double *dst = p.part_buf + (size_t)(item.w + item.y) * PART_DOUBLES;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[(size_t)(s * 4 + i) * THREADS + tid] = C[s][i];
            }
            // every wavefront waits for its own stores; ONE thread then makes them visible device-wide (agent-scope release) and
            // counts the arrival; the last arriver acquires on behalf of the workgroup (same protocol as the similarity kernel's
            // split columns)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                const bool last = atomicAdd(&p.part_count[item.w], 1u) == (unsigned)(item.z - 1);
                if (last) __threadfence();
                s_row = last ? 1 : 0;
            }
            __syncthreads();
            const bool last = s_row != 0;
            if (!last) continue;
            const double *src = p.part_buf + (size_t)item.w * PART_DOUBLES;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double v = 0.0;
                    for (int q = 0; q < item.z; ++q) v += src[(size_t)q * PART_DOUBLES + (size_t)(s * 4 + i) * THREADS + tid];
                    C[s][i] = v;
                }
        }
        if (p.phases && tid == 0) { asm volatile("" ::"v"(C[0][0])); t2 = ials_stamp(); }
        if (STAGE == 1) {                                            // hand the system to the solve stage
            double *dst = p.systems + (size_t)p.sys_slot[slot] * SYS_DOUBLES;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[(size_t)(s * 4 + i) * THREADS + tid] = C[s][i];
            if (p.phases && tid == 0) {
                atomicAdd(&p.phases[0], t1 - t0);
                atomicAdd(&p.phases[1], t2 - t1);
                atomicAdd(&p.phases[4], 1ull);
            }
            continue;
        }

        // (3) blocked Cholesky of the augmented matrix, panel J = columns 16 J .. 16 J + 15
        for (int J = 0; J < KT; ++J) {
            const int g = not_invariant(lane >> 4), cl = not_invariant(lane & 15);
            __syncthreads();                                         // (the staging area / the previous panel is no longer read)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (tJs[s] == J) {
                    double *tile = P + (tIs[s] - J) * 16 * TP;
#pragma unroll
                    for (int i = 0; i < 4; ++i) tile[(4 * i + g) * TP + cl] = C[s][i];
                }
            __syncthreads();
            unsigned long long c0 = 0, c1 = 0, c2 = 0;
            if (p.phases && tid == 0) c0 = ials_stamp();
            if (wave == 0) factor_and_invert_diagonal_tile(P, Linv + J * 16 * TP, lane);
            if (p.phases && tid == 0) { asm volatile("" ::: "memory"); c1 = ials_stamp(); }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                if (tJs[s] == J && tIs[s] > J) {                     // X = T L_JJ^-T on the matrix pipe: the owner's registers ARE the result
                    double *tile = P + (tIs[s] - J) * 16 * TP;
                    const double *li = Linv + J * 16 * TP;
                    d4 X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) X = __builtin_amdgcn_mfma_f64_16x16x4f64(tile[cl * TP + 4 * q + g], li[cl * TP + 4 * q + g], X, 0, 0, 0);
                    C[s] = X;
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // (the tile is rewritten below: its operand reads must have issued)
#pragma unroll
                    for (int i = 0; i < 4; ++i) tile[(4 * i + g) * TP + cl] = X[i];
                } else if (tJs[s] == J) {                            // the diagonal tile: L_JJ
#pragma unroll
                    for (int i = 0; i < 4; ++i) C[s][i] = P[(4 * i + g) * TP + cl];
                }
            }
            __syncthreads();
            if (p.phases && tid == 0) c2 = ials_stamp();
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (tJs[s] > J) {                                    // trailing update: tile -= X_I X_J'^T
                    const double *xa = P + ((tIs[s] - J) * 16 + cl) * TP, *xb = P + ((tJs[s] - J) * 16 + cl) * TP;
                    double av[4], bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { av[q] = -xa[4 * q + g]; bv[q] = xb[4 * q + g]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) C[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], C[s], 0, 0, 0);
                }
            if (p.phases && tid == 0) {
                asm volatile("" ::"v"(C[0][0]));
                const unsigned long long c3 = ials_stamp();
                atomicAdd(&p.phases[5], c1 - c0);
                atomicAdd(&p.phases[6], c2 - c1);
                atomicAdd(&p.phases[7], c3 - c2);
            }
        }
        if (p.phases && tid == 0) { asm volatile("" ::"v"(C[0][0])); t3 = ials_stamp(); }

        // (4) back substitution L^T x = z; z is row k of L (the forward substitution came with the factorisation)
        __syncthreads();
        for (int f = tid; f < KP; f += THREADS) { xv[f] = 0.0; acc[f] = 0.0; zv[f] = 0.0; }
        __syncthreads();
        const int g = not_invariant(lane >> 4), cl = not_invariant(lane & 15);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (tIs[s] == Ik && g == (rk & 3)) zv[16 * tJs[s] + cl] = C[s][rk >> 2];
        __syncthreads();
        for (int I = KT - 1; I >= 0; --I) {
            if (wave == 0) {                                         // x_I = L_II^-T (z_I - acc_I): lane c < 16 takes entry c
                const int c = lane & 15;
                double t = zv[16 * I + c] - acc[16 * I + c];
                if (16 * I + c >= k) t = 0.0;                        // the rhs row and the padding are not unknowns
                double x = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m) x += Linv[(I * 16 + m) * TP + c] * lane_bcast(t, m);   // (L^-T)[c][m] = Linv[m][c]
                if (16 * I + c >= k) x = 0.0;
                if (lane < 16) xv[16 * I + c] = x;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (tIs[s] == I && tJs[s] < I) {                     // segment J loses L[I][J]^T x_I
                    double part = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) part += C[s][i] * xv[16 * I + 4 * i + g];
                    part = swap_sum32(swap_sum16(part));
                    if (g == 0) atomicAdd(&acc[16 * tJs[s] + cl], part);
                }
            __syncthreads();
        }
        if (tid < k) p.X[(size_t)row * k + tid] = xv[tid];
        if (p.phases && tid == 0) {
            const unsigned long long t4 = ials_stamp();
            atomicAdd(&p.phases[0], t1 - t0);
            atomicAdd(&p.phases[1], t2 - t1);
            atomicAdd(&p.phases[4], 1ull);
            atomicAdd(&p.phases[2], t3 - t2);
            atomicAdd(&p.phases[3], t4 - t3);
        }
    }
}

// CSR -> CSC of the confidence matrix on the device (row ids inside a column end up unordered: irrelevant here,
// the Gramian is a sum).
// ---- the solve stage of a two-stage epoch ----------------------------------------------------------------------------------
// One 512-thread workgroup per row, TWO workgroups per CU (128 registers per lane): wavefront 0 is the PANEL wavefront -- it factors
// and inverts the diagonal tile of every panel and solves the diagonal blocks of the back substitution -- and holds no tile;
// wavefronts 1..7 hold the lower-triangle tiles (tile t on wavefront 1 + t % 7, slot t / 7: 13 slots = 104 registers at k = 200) and
// do the matrix-pipe work: panel solves and trailing updates.  The two roles run different code between the same barriers, so the
// register allocation is the larger of the two, not their sum (tiles 104 + operands; diagonal tile 64 + the inverse) -- in the
// one-kernel epoch every wavefront carries both.  While one workgroup's panel wavefront works through its 16 pivots, the other
// workgroup of the CU has the matrix pipe.  Algorithm and order of operations: those of ials_row_kernel steps (3), (4); the factors
// of a two-stage epoch agree with the one-kernel epoch's to 1e-12 (two compilations of the same expressions; the back substitution's
// LDS atomics add in arrival order in both).
// Measured and not kept (round 5): the diagonal tile with DPP row broadcasts instead of v_readlane (factor alone 103 k cycles per row
// against ~80 k; factor + inverse 559 k against 158 k), and no inverse at all -- panel tiles solved by substitution on the tile
// wavefronts (DPP), back substitution by substitution on the panel wavefront: the Cholesky took 274 k cycles per row against 267 k, the
// back substitution 60 k against 35 k; the columns of L through LDS (broadcast reads) instead of SGPRs: diagonal tiles 317 k cycles per
// row against 158 k (the panel wavefront's share of the 128 registers is small: its rows went to scratch).
constexpr int TILE_WAVES = ROW_WAVES - 1;
template <int SLOTS>
__global__ __launch_bounds__(ROW_THREADS, 4) void ials_solve_kernel(const IalsParams p, int gram_slots, int gram_threads) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ short s_tI[MAX_NT], s_tJ[MAX_NT];
    __shared__ int s_row;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = p.k, KT = (k + 1 + 15) / 16, KP = KT * 16, NT = KT * (KT + 1) / 2;
    double *const P = lds;                                          // [KT][16][TP]  panel tiles
    double *const Ld = lds + KT * 16 * TP;                          // [KT][16][TP]  inverses of the factored diagonal tiles
    double *const zv = Ld + KT * 16 * TP + KT * 16;                 // [KP] forward-substituted rhs
    double *const xv = zv + KP;                                     // [KP] solution
    double *const acc = xv + KP;                                    // [KP] sum of L[I][J]^T x_I over the tile rows already solved
    for (int t = tid; t < NT; t += ROW_THREADS) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        s_tI[t] = (short)I;
        s_tJ[t] = (short)(t - I * (I + 1) / 2);
    }
    __syncthreads();
    const int Ik = k >> 4, rk = k & 15;                             // tile row and local row of the rhs row
    const size_t sys_doubles = (size_t)gram_slots * 4 * gram_threads;      // (the Gramian stage's workgroup wrote the slab)
    const int gram_waves = gram_threads / 64;
    // The two workgroups of a CU should not have their panel wavefronts on the same SIMD (both chains would share one issue port:
    // measured 120 k -> 158 k cycles of diagonal-tile work per row).  Wavefront w of a workgroup sits on SIMD w % 4 and the second
    // half of the grid is what doubles the CUs up (both observed, neither promised: a wrong guess costs speed only) -- the second
    // half takes wavefront 2 as its panel wavefront.  Tile wavefronts are numbered 0..6 in the order of the remaining indices.
    const int panel_wave = 2 * (int)(blockIdx.x >= gridDim.x / 2 && gridDim.x > 1 && !p.same_panel_wave);
    const bool is_panel = wave == panel_wave;
    const int tile_wave = wave - (wave > panel_wave);               // 0 .. 6 for the others

    for (;;) {
        __syncthreads();
        if (tid == 0) s_row = (int)atomicAdd(p.queue, 1u);
        __syncthreads();
        const int slot = s_row;
        if (slot >= p.n_local) break;
        const int row = p.items[slot].x;
        unsigned long long t2 = 0, t3 = 0;
        const bool clock = p.phases && is_panel && lane == 0;
        if (clock) t2 = ials_stamp();
        if (is_panel) {
            // ---- the panel wavefront ----
            for (int J = 0; J < KT; ++J) {
                __syncthreads();
                __syncthreads();                                     // the panel's tiles are in LDS
                unsigned long long c0 = 0;
                if (clock) c0 = ials_stamp();
                // (s_setprio 3 around it -- the chain everybody waits for first at the issue port -- changes nothing: measured)
                factor_and_invert_diagonal_tile(P, Ld + J * 16 * TP, lane);
                if (clock) { asm volatile("" ::: "memory"); atomicAdd(&p.phases[5], ials_stamp() - c0); }
                __syncthreads();
                __syncthreads();
            }
            if (clock) t3 = ials_stamp();
            __syncthreads();
            __syncthreads();
            __syncthreads();
            for (int I = KT - 1; I >= 0; --I) {                      // x_I = L_II^-T (z_I - acc_I) by substitution: lane c < 16 takes entry c
                const int c = lane & 15;
                double t = zv[16 * I + c] - acc[16 * I + c];
                if (16 * I + c >= k) t = 0.0;                        // the rhs row and the padding are not unknowns
                double x = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m) x += Ld[(I * 16 + m) * TP + c] * lane_bcast(t, m);   // (L^-T)[c][m] = Linv[m][c]
                if (16 * I + c >= k) x = 0.0;
                if (lane < 16) xv[16 * I + c] = x;
                __syncthreads();
                __syncthreads();
            }
        } else {
            // ---- a tile wavefront ----
            int tIs[SLOTS], tJs[SLOTS];
            d4 C[SLOTS];
            {
                const double *src = p.systems + (size_t)p.sys_slot[slot] * sys_doubles;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int t = tile_wave + TILE_WAVES * s;
                    tIs[s] = t < NT ? __builtin_amdgcn_readfirstlane((int)s_tI[t]) : -1;
                    tJs[s] = t < NT ? __builtin_amdgcn_readfirstlane((int)s_tJ[t]) : -1;
                    // stage 1 left tile t in slot t / W of its wavefront t % W (W wavefronts)
                    const double *ts = src + (size_t)(t / gram_waves) * 4 * gram_threads + (t % gram_waves) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) C[s][i] = t < NT ? ts[(size_t)i * gram_threads] : 0.0;
                }
            }
            for (int J = 0; J < KT; ++J) {
                const int g = not_invariant(lane >> 4), cl = not_invariant(lane & 15);
                __syncthreads();                                     // (the previous panel is no longer read)
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (tJs[s] == J) {
                        double *tile = P + (tIs[s] - J) * 16 * TP;
#pragma unroll
                        for (int i = 0; i < 4; ++i) tile[(4 * i + g) * TP + cl] = C[s][i];
                    }
                __syncthreads();
                __syncthreads();                                     // the panel wavefront has factored the diagonal tile
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    if (tJs[s] == J && tIs[s] > J) {                 // X = T L_JJ^-T on the matrix pipe: the owner's registers ARE the result
                        double *tile = P + (tIs[s] - J) * 16 * TP;
                        const double *li = Ld + J * 16 * TP;
                        d4 X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < 4; ++q) X = __builtin_amdgcn_mfma_f64_16x16x4f64(tile[cl * TP + 4 * q + g], li[cl * TP + 4 * q + g], X, 0, 0, 0);
                        C[s] = X;
                        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // (the tile is rewritten below: its operand reads must have issued)
#pragma unroll
                        for (int i = 0; i < 4; ++i) tile[(4 * i + g) * TP + cl] = X[i];
                    } else if (tJs[s] == J) {                        // the diagonal tile: L_JJ
#pragma unroll
                        for (int i = 0; i < 4; ++i) C[s][i] = P[(4 * i + g) * TP + cl];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (tJs[s] > J) {                                // trailing update: tile -= X_I X_J'^T
                        const double *xa = P + ((tIs[s] - J) * 16 + cl) * TP, *xb = P + ((tJs[s] - J) * 16 + cl) * TP;
                        double av[4], bv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { av[q] = -xa[4 * q + g]; bv[q] = xb[4 * q + g]; }
#pragma unroll
                        for (int q = 0; q < 4; ++q) C[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], C[s], 0, 0, 0);
                    }
            }
            // (4) back substitution L^T x = z; z is row k of L (the forward substitution came with the factorisation)
            __syncthreads();
            for (int f = tile_wave * 64 + lane; f < KP; f += ROW_THREADS - 64) { xv[f] = 0.0; acc[f] = 0.0; zv[f] = 0.0; }
            __syncthreads();
            const int g = not_invariant(lane >> 4), cl = not_invariant(lane & 15);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (tIs[s] == Ik && g == (rk & 3)) zv[16 * tJs[s] + cl] = C[s][rk >> 2];
            __syncthreads();
            for (int I = KT - 1; I >= 0; --I) {
                __syncthreads();                                     // the panel wavefront has x_I
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (tIs[s] == I && tJs[s] < I) {                 // segment J loses L[I][J]^T x_I
                        double part = 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) part += C[s][i] * xv[16 * I + 4 * i + g];
                        part = swap_sum32(swap_sum16(part));
                        if (g == 0) atomicAdd(&acc[16 * tJs[s] + cl], part);
                    }
                __syncthreads();
            }
        }
        if (tid < k) p.X[(size_t)row * k + tid] = xv[tid];
        if (clock) {
            const unsigned long long t4 = ials_stamp();
            atomicAdd(&p.phases[2], t3 - t2);
            atomicAdd(&p.phases[3], t4 - t3);
        }
    }
}

__global__ void ials_count_kernel(const int *idx, size_t nnz, int *cnt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nnz; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[idx[i]], 1);
}
__global__ __launch_bounds__(1024) void ials_scan_kernel(const int *cnt, int *out, int *cursor, int n) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int b = tid * chunk, e = min(n, b + chunk);
    long long s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    long long run = tid ? part[tid - 1] : 0;
    for (int i = b; i < e; ++i) {
        out[i] = (int)run;
        cursor[i] = (int)run;
        run += cnt[i];
    }
    if (tid == 1023) out[n] = (int)part[1023];
}
__global__ void ials_scatter_kernel(const int *ptr, const int *idx, const float *val, int n_rows, int *cursor, int *t_idx,
                                    float *t_val) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    for (int q = ptr[wave] + lane; q < ptr[wave + 1]; q += 64) {
        const int pos = atomicAdd(&cursor[idx[q]], 1);
        t_idx[pos] = wave;
        t_val[pos] = val[q];
    }
}

}  // namespace
}  // namespace mi355rec

using namespace mi355rec;

struct mi355rec_ials {
    int n_users = 0, n_items = 0, k = 0;
    double reg = 0;
    size_t nnz = 0;
    hipStream_t stream = nullptr;
    StreamTimer call_timer;
    DispatchTimers dispatch_timers;
    DeviceBuffer<int> u_ptr, u_idx, i_ptr, i_idx;
    DeviceBuffer<int4> items;
    DeviceBuffer<double> part_buf;
    DeviceBuffer<unsigned> part_count;
    DeviceBuffer<double> systems;          // two-stage epochs: the augmented systems of one batch of rows
    DeviceBuffer<int> sys_slot;
    std::vector<int> sys_slot_host;
    int n_batches = 0;
    double system_gib = 0.0;             // size of the two-stage epochs' systems buffer (0: not decided yet, see rows_per_batch)
    std::vector<int4> items_host;
    int n_split_rows = 0, n_part_items = 0;
    DeviceBuffer<float> u_conf, i_conf;
    DeviceBuffer<double> U, V, G;
    DeviceBuffer<unsigned> queue;
    DeviceBuffer<unsigned long long> phases;
    std::vector<int> user_order, item_order;     // longest profile first
    std::vector<int> u_ptr_host, i_ptr_host;
    std::vector<int> staging;
    mi355rec_stats stats{};
    double flops_acc = 0, bytes_acc = 0;
    long long rows_acc = 0, launches_acc = 0;

    ~mi355rec_ials() {
        if (stream) (void)hipStreamSynchronize(stream);
        call_timer.destroy();
        dispatch_timers.destroy();
        ReleaseScope::forget(stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

template <int KT16>
void launch_gram_t(mi355rec_ials *h, const double *Y, int n, int grid) {
    hipLaunchKernelGGL(gram_kernel<KT16>, dim3(grid), dim3(256), 0, h->stream, Y, n, h->k, h->G.ptr, 0, 0);
}

void launch_gram(mi355rec_ials *h, const double *Y, int n) {
    MI_HIP(hipMemsetAsync(h->G.ptr, 0, sizeof(double) * h->k * h->k, h->stream));
    const int grid = std::max(1, std::min(2 * multiprocessor_count(), (n + 63) / 64));
    switch ((h->k + 15) / 16) {
        case 1: launch_gram_t<1>(h, Y, n, grid); break;
        case 2: launch_gram_t<2>(h, Y, n, grid); break;
        case 3: launch_gram_t<3>(h, Y, n, grid); break;
        case 4: launch_gram_t<4>(h, Y, n, grid); break;
        case 5: launch_gram_t<5>(h, Y, n, grid); break;
        case 6: launch_gram_t<6>(h, Y, n, grid); break;
        case 7: launch_gram_t<7>(h, Y, n, grid); break;
        case 8: launch_gram_t<8>(h, Y, n, grid); break;
        case 9: launch_gram_t<9>(h, Y, n, grid); break;
        case 10: launch_gram_t<10>(h, Y, n, grid); break;
        case 11: launch_gram_t<11>(h, Y, n, grid); break;
        case 12: launch_gram_t<12>(h, Y, n, grid); break;
        case 13: launch_gram_t<13>(h, Y, n, grid); break;
        case 14: launch_gram_t<14>(h, Y, n, grid); break;
        default:        // 15 or 16 tiles per side: four quadrants of 8 x 8 tiles
            for (int a0 = 0; a0 < 16; a0 += 8)
                for (int b0 = 0; b0 < 16; b0 += 8)
                    hipLaunchKernelGGL(gram_kernel<8>, dim3(grid), dim3(256), 0, h->stream, Y, n, h->k, h->G.ptr, a0, b0);
            break;
    }
}

template <int SLOTS, int STAGE, int THREADS>
void launch_rows_ts(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = sizeof(double) * ials_lds_doubles(h->k, STAGE);
    auto kern = ials_row_kernel<SLOTS, STAGE, THREADS>;
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), (unsigned)lds, h->stream, e0, e1, 0, p);
}

template <int SLOTS>
void launch_rows_t(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1, int) {
    launch_rows_ts<SLOTS, 0, ROW_THREADS>(h, p, grid, e0, e1);
}

// The Gramian stage of a two-stage epoch: 1024 threads, tiles over sixteen wavefronts
constexpr int GRAM_THREADS = 1024, GRAM_WAVES = GRAM_THREADS / 64;
int gram_slots_of(int NT) { return (NT + GRAM_WAVES - 1) / GRAM_WAVES; }
void launch_gram_stage(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1) {
    const int KT = (h->k + 1 + 15) / 16, NT = KT * (KT + 1) / 2;
    switch (gram_slots_of(NT)) {
        case 1: launch_rows_ts<1, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        case 2: launch_rows_ts<2, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        case 3: launch_rows_ts<3, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        case 4: launch_rows_ts<4, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        case 5: launch_rows_ts<5, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        case 6: launch_rows_ts<6, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        case 7: launch_rows_ts<7, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
        default: launch_rows_ts<8, 1, GRAM_THREADS>(h, p, grid, e0, e1); break;
    }
}

template <int SLOTS>
void launch_solve_t(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1, int gram_slots) {
    const size_t lds = sizeof(double) * ials_lds_doubles(h->k, 2);
    auto kern = ials_solve_kernel<SLOTS>;
    MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(ROW_THREADS), (unsigned)lds, h->stream, e0, e1, 0, p, gram_slots, (int)GRAM_THREADS);
}

// the solve stage holds the tiles on 7 wavefronts: up to 13 slots (k <= 207) fit two workgroups per CU
constexpr int MAX_SOLVE_SLOTS = 13;
int solve_slots(int NT) { return (NT + TILE_WAVES - 1) / TILE_WAVES; }

void launch_solve(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1, int gram_slots) {
    const int KT = (h->k + 1 + 15) / 16, NT = KT * (KT + 1) / 2;
    switch (solve_slots(NT)) {
        case 1: launch_solve_t<1>(h, p, grid, e0, e1, gram_slots); break;
        case 2: launch_solve_t<2>(h, p, grid, e0, e1, gram_slots); break;
        case 3: case 4: launch_solve_t<4>(h, p, grid, e0, e1, gram_slots); break;
        case 5: case 6: launch_solve_t<6>(h, p, grid, e0, e1, gram_slots); break;
        case 7: case 8: launch_solve_t<8>(h, p, grid, e0, e1, gram_slots); break;
        case 9: case 10: launch_solve_t<10>(h, p, grid, e0, e1, gram_slots); break;
        case 11: launch_solve_t<11>(h, p, grid, e0, e1, gram_slots); break;
        case 12: launch_solve_t<12>(h, p, grid, e0, e1, gram_slots); break;
        default: launch_solve_t<13>(h, p, grid, e0, e1, gram_slots); break;
    }
}

// SLOTS of the ials_row_kernel instance launch_rows picks for NT lower-triangle tiles (must mirror its switch)
int row_slots(int NT) {
    const int need = (NT + ROW_WAVES - 1) / ROW_WAVES;
    if (need <= 2) return need;
    if (need <= 12) return (need + 1) / 2 * 2;
    return need == 13 ? 13 : (need <= 15 ? 15 : 17);
}

void launch_rows(mi355rec_ials *h, const IalsParams &p, int grid, hipEvent_t e0, hipEvent_t e1, int stage = 0) {
    const int KT = (h->k + 1 + 15) / 16, NT = KT * (KT + 1) / 2;
    switch ((NT + ROW_WAVES - 1) / ROW_WAVES) {          // lower-triangle tiles per wavefront
        case 1: launch_rows_t<1>(h, p, grid, e0, e1, stage); break;
        case 2: launch_rows_t<2>(h, p, grid, e0, e1, stage); break;
        case 3: case 4: launch_rows_t<4>(h, p, grid, e0, e1, stage); break;
        case 5: case 6: launch_rows_t<6>(h, p, grid, e0, e1, stage); break;
        case 7: case 8: launch_rows_t<8>(h, p, grid, e0, e1, stage); break;
        case 9: case 10: launch_rows_t<10>(h, p, grid, e0, e1, stage); break;
        case 11: case 12: launch_rows_t<12>(h, p, grid, e0, e1, stage); break;
        case 13: launch_rows_t<13>(h, p, grid, e0, e1, stage); break;
        case 14: case 15: launch_rows_t<15>(h, p, grid, e0, e1, stage); break;
        default: launch_rows_t<17>(h, p, grid, e0, e1, stage); break;      // 16 x 16 tiles: k <= 255
    }
}

// doubles of one row's augmented system (or of one part of a split row) as the kernel that builds it lays it out
size_t slab_doubles(const mi355rec_ials *h, bool two_stage) {
    const int KT = (h->k + 1 + 15) / 16, NT = KT * (KT + 1) / 2;
    return two_stage ? (size_t)gram_slots_of(NT) * 4 * GRAM_THREADS : (size_t)row_slots(NT) * 4 * ROW_THREADS;
}

// Two-stage epochs: on unless MI355REC_IALS_TWO_STAGE=0 or the solve stage's 13 tile slots do not cover k (k > 207).
bool two_stage_epochs(const mi355rec_ials *h) {
    const char *ts = getenv("MI355REC_IALS_TWO_STAGE");
    const int KT = (h->k + 1 + 15) / 16;
    return (ts ? atoi(ts) != 0 : true) && solve_slots(KT * (KT + 1) / 2) <= MAX_SOLVE_SLOTS;
}

// Rows whose systems the buffer holds at a time: MI355REC_IALS_SYSTEM_GIB worth of slabs (default: 8, but never more than a quarter
// of the memory the device has free when the handle first asks -- several handles on one device, or a device shared with PyTorch /
// RCCL, each take a share instead of 8 GiB), at most the longer side.  A buffer that cannot be had is halved (ensure_systems).
int rows_per_batch(mi355rec_ials *h) {
    const size_t sys_bytes = slab_doubles(h, true) * sizeof(double);
    if (h->system_gib <= 0.0) {
        double gib = 8.0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) gib = std::min(gib, (double)free_b / 4.0 / 1073741824.0);
        else (void)hipGetLastError();
        if (getenv("MI355REC_IALS_SYSTEM_GIB")) gib = atof(getenv("MI355REC_IALS_SYSTEM_GIB"));
        h->system_gib = std::max(0.001, gib);
    }
    const size_t longest = (size_t)std::max(h->n_users, h->n_items);
    return (int)std::max<size_t>(1, std::min<size_t>(longest, (size_t)(h->system_gib * 1073741824.0) / sys_bytes));
}

// The systems buffer for half-steps of up to `rows` rows: min(rows, rows_per_batch) slabs.  Blocks above 1 GiB bypass the block cache,
// so this is a real hipMalloc -- done once per handle (and again only if a larger half-step arrives).  If the device cannot give it,
// the batch is halved (more, smaller batches: same results) down to 64 slabs; below that the caller falls back to the one-kernel epoch.
bool ensure_systems(mi355rec_ials *h, int rows) {
    const size_t slab = slab_doubles(h, true);
    for (;;) {
        const int per_batch = std::min(std::max(1, rows), rows_per_batch(h));
        if (h->systems.count >= (size_t)per_batch * slab) return true;
        try {
            h->systems.alloc((size_t)per_batch * slab);
            return true;
        } catch (const Error &) {
            (void)hipGetLastError();
            if (per_batch <= 64) return false;
            h->system_gib = std::max(0.001, 0.5 * (double)per_batch * (double)slab * sizeof(double) / 1073741824.0);
        }
    }
}

// Solve the rows [r0, r1) of one side.
void half_step(mi355rec_ials *h, bool users, int r0, int r1) {
    const int n_side = users ? h->n_users : h->n_items;
    MI_REQUIRE(r0 >= 0 && r1 <= n_side && r0 <= r1, "row range [%d,%d) outside [0,%d)", r0, r1, n_side);
    const std::vector<int> &full = users ? h->user_order : h->item_order;
    const std::vector<int> &ptr = users ? h->u_ptr_host : h->i_ptr_host;
    h->staging.clear();
    double nnz_rows = 0;
    for (int r : full)
        if (r >= r0 && r < r1 && ptr[r + 1] > ptr[r]) {       // warm rows only (IALSRecommender.py:78-82)
            h->staging.push_back(r);
            nnz_rows += ptr[r + 1] - ptr[r];
        }
    const int n_local = (int)h->staging.size();
    launch_gram(h, users ? h->V.ptr : h->U.ptr, users ? h->n_items : h->n_users);
    if (n_local == 0) return;
    // Work items, most expensive first.  cost(row) = L k^2 (Gramian) + k^3 / 3 (solve), in units of k^2: L + k / 3.  A row with more
    // than 2 x PART_ROWS profile entries is split into parts of PART_ROWS (at most 64 parts); the part that arrives last merges and
    // solves (ials_row_kernel).  The split depends on the row alone -- not on what else the call holds -- so a row-sharded epoch adds
    // every row up in the same order as the single-GPU epoch.  (MI355REC_IALS_PART_ROWS / MI355REC_IALS_NO_SPLIT: tests.)
    const int grid_cap = multiprocessor_count();
    const double solve_cost = h->k / 3.0;
    int part_rows = 4096;
    if (getenv("MI355REC_IALS_PART_ROWS")) part_rows = std::max(CHUNK, atoi(getenv("MI355REC_IALS_PART_ROWS")) / CHUNK * CHUNK);
    const bool may_split = !getenv("MI355REC_IALS_NO_SPLIT");
    h->items_host.clear();
    std::vector<std::pair<double, int>> keyed;
    int part_slots = 0, n_split = 0;
    for (int r : h->staging) {
        const int L = ptr[r + 1] - ptr[r];
        int parts = 1;
        if (may_split && L > 2 * part_rows) parts = std::min(64, (L + part_rows - 1) / part_rows);
        if (parts > 1) {
            for (int q = 0; q < parts; ++q) {
                keyed.emplace_back((double)L / parts + (q == 0 ? solve_cost : 0.0), (int)h->items_host.size());
                h->items_host.push_back(make_int4(r, q, parts, part_slots));
            }
            part_slots += parts;
            ++n_split;
        } else {
            keyed.emplace_back(L + solve_cost, (int)h->items_host.size());
            h->items_host.push_back(make_int4(r, 0, 1, 0));
        }
    }
    if (n_split) {
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
        std::vector<int4> sorted(keyed.size());
        for (size_t i = 0; i < keyed.size(); ++i) sorted[i] = h->items_host[keyed[i].second];
        h->items_host.swap(sorted);
    }
    const int n_work = (int)h->items_host.size();
    h->n_split_rows = n_split;
    h->n_part_items = part_slots;
    if (h->items.count < (size_t)n_work) h->items.alloc((size_t)n_work + 1024);
    MI_HIP(hipMemcpyAsync(h->items.ptr, h->items_host.data(), sizeof(int4) * n_work, hipMemcpyHostToDevice, h->stream));
    MI_HIP(hipMemsetAsync(h->queue.ptr, 0, sizeof(unsigned), h->stream));
    if (part_slots) {
        const size_t part_doubles = slab_doubles(h, two_stage_epochs(h));
        if (h->part_buf.count < (size_t)part_slots * part_doubles) h->part_buf.alloc((size_t)part_slots * part_doubles);
        if (h->part_count.count < (size_t)part_slots) h->part_count.alloc((size_t)part_slots);
        MI_HIP(hipMemsetAsync(h->part_count.ptr, 0, sizeof(unsigned) * part_slots, h->stream));
    }
    IalsParams p{};
    p.k = h->k;
    p.reg = h->reg;
    p.ptr = users ? h->u_ptr.ptr : h->i_ptr.ptr;
    p.idx = users ? h->u_idx.ptr : h->i_idx.ptr;
    p.conf = users ? h->u_conf.ptr : h->i_conf.ptr;
    p.Y = users ? h->V.ptr : h->U.ptr;
    p.G = h->G.ptr;
    p.X = users ? h->U.ptr : h->V.ptr;
    p.items = h->items.ptr;
    p.n_local = n_work;
    p.part_buf = h->part_buf.ptr;
    p.part_count = h->part_count.ptr;
    p.queue = h->queue.ptr;
    p.phases = h->phases.ptr;
    // Two-stage epochs (ials_row_kernel, STAGE 1 / 2): the rows of the half-step, in cost order, in batches whose systems fit the
    // buffer (MI355REC_IALS_SYSTEM_GIB, default 8: 43 000 rows at k = 200); per batch one launch builds the systems (work items =
    // the batch's rows and parts, still most expensive first) and one solves them, two workgroups per CU.  Same arithmetic as the
    // one-kernel epoch (tests/test_ials_gpu.py::test_two_stage_epochs_equal_one_kernel_epochs: 1e-12).
    const bool two_stage = two_stage_epochs(h) && ensure_systems(h, n_local);      // (no memory for 64 systems: the one-kernel epoch)
    h->n_batches = 0;
    if (!two_stage) {
        const int grid = std::min(n_work, grid_cap);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        h->dispatch_timers.reserve(h->dispatch_timers.used + 1);
        h->dispatch_timers.next(e0, e1, 1 << 30);
        launch_rows(h, p, grid, e0, e1);
        MI_HIP(hipGetLastError());
    } else {
        const int KT_ = (h->k + 1 + 15) / 16, NT_ = KT_ * (KT_ + 1) / 2;
        const size_t sys_doubles = slab_doubles(h, true);
        const int per_batch = (int)std::min<size_t>((size_t)std::min(std::max(1, n_local), rows_per_batch(h)), h->systems.count / sys_doubles);
        const int n_batches = (n_local + per_batch - 1) / per_batch;
        h->n_batches = n_batches;
        // position of every row in the cost order -> (batch, slab)
        std::vector<int> pos_of_row((size_t)n_side, -1);
        for (int i = 0; i < n_local; ++i) pos_of_row[h->staging[i]] = i;
        // stage-1 items per batch (the sorted list filtered), then one stage-2 item per row
        std::vector<std::vector<int4>> gram_items((size_t)n_batches);
        for (const int4 &it : h->items_host) gram_items[(size_t)(pos_of_row[it.x] / per_batch)].push_back(it);
        std::vector<int4> all;
        h->sys_slot_host.clear();
        std::vector<int> gram_off((size_t)n_batches), gram_n((size_t)n_batches), solve_off((size_t)n_batches), solve_n((size_t)n_batches);
        for (int b = 0; b < n_batches; ++b) {
            gram_off[b] = (int)all.size();
            gram_n[b] = (int)gram_items[b].size();
            for (const int4 &it : gram_items[b]) {
                all.push_back(it);
                h->sys_slot_host.push_back(pos_of_row[it.x] % per_batch);
            }
            solve_off[b] = (int)all.size();
            const int r_end = std::min(n_local, (b + 1) * per_batch);
            solve_n[b] = r_end - b * per_batch;
            for (int i = b * per_batch; i < r_end; ++i) {
                all.push_back(make_int4(h->staging[i], 0, 1, 0));
                h->sys_slot_host.push_back(i % per_batch);
            }
        }
        if (h->items.count < all.size()) h->items.alloc(all.size() + 1024);
        if (h->sys_slot.count < all.size()) h->sys_slot.alloc(all.size() + 1024);
        MI_HIP(hipMemcpyAsync(h->items.ptr, all.data(), sizeof(int4) * all.size(), hipMemcpyHostToDevice, h->stream));
        MI_HIP(hipMemcpyAsync(h->sys_slot.ptr, h->sys_slot_host.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice, h->stream));
        MI_HIP(hipStreamSynchronize(h->stream));         // (`all` is a local)
        h->dispatch_timers.reserve(h->dispatch_timers.used + 2 * n_batches);
        p.systems = h->systems.ptr;
        p.same_panel_wave = getenv("MI355REC_IALS_SAME_PANEL_WAVE") != nullptr;
        for (int b = 0; b < n_batches; ++b) {
            for (int stage = 1; stage <= 2; ++stage) {
                p.items = h->items.ptr + (stage == 1 ? gram_off[b] : solve_off[b]);
                p.sys_slot = h->sys_slot.ptr + (stage == 1 ? gram_off[b] : solve_off[b]);
                p.n_local = stage == 1 ? gram_n[b] : solve_n[b];
                MI_HIP(hipMemsetAsync(h->queue.ptr, 0, sizeof(unsigned), h->stream));
                hipEvent_t e0 = nullptr, e1 = nullptr;
                h->dispatch_timers.next(e0, e1, 1 << 30);
                if (stage == 1) launch_gram_stage(h, p, std::min(p.n_local, grid_cap), e0, e1);
                else launch_solve(h, p, std::min(p.n_local, 2 * grid_cap), e0, e1, gram_slots_of(NT_));
                MI_HIP(hipGetLastError());
            }
        }
    }
    if (h->phases.ptr) {
        unsigned long long ph[8];
        MI_HIP(hipMemcpyAsync(ph, h->phases.ptr, sizeof(ph), hipMemcpyDeviceToHost, h->stream));
        MI_HIP(hipStreamSynchronize(h->stream));
        fprintf(stderr, "[ials phases] %s half: %llu rows; shader cycles per row: base %.0f, Gramian %.0f, Cholesky %.0f (diagonal tiles %.0f, panel solves %.0f, "
                        "trailing updates of wavefront 0 %.0f), back-substitution %.0f\n",
                users ? "user" : "item", ph[4], (double)ph[0] / ph[4], (double)ph[1] / ph[4], (double)ph[2] / ph[4], (double)ph[5] / ph[4], (double)ph[6] / ph[4],
                (double)ph[7] / ph[4], (double)ph[3] / ph[4]);
        MI_HIP(hipMemsetAsync(h->phases.ptr, 0, sizeof(ph), h->stream));
    }
    // ALGORITHMIC work, SURVEY.md section 8(d): Gramian 2 * nnz * k^2 flop per pass (+ the k x k base Gramian 2 n k^2),
    // solve k^3/3 + 2 k^2 per row; bytes: gathered factor rows + confidences + the solved rows.
    const double k = h->k;
    h->flops_acc += 2.0 * nnz_rows * k * k + (double)n_local * (k * k * k / 3.0 + 2.0 * k * k) +
                    2.0 * (double)(users ? h->n_items : h->n_users) * k * k;
    h->bytes_acc += nnz_rows * (8.0 * k + 8.0) + (double)n_local * 8.0 * k;
    h->rows_acc += n_local;
    h->launches_acc += 1;
}

// `rows`: the longest half-step the call will run
void begin_call(mi355rec_ials *h, int rows) {
    // (the buffer of the two-stage epochs is allocated here, before the call's clock starts: 8 GiB of hipMalloc took 0.4 s of the
    // first epoch's "call_ms" when half_step did it)
    if (two_stage_epochs(h)) (void)ensure_systems(h, rows);
    h->dispatch_timers.reset();
    h->flops_acc = h->bytes_acc = 0;
    h->rows_acc = h->launches_acc = 0;
    h->call_timer.start(h->stream);
}

void end_call(mi355rec_ials *h, bool sync) {
    h->call_timer.stop(h->stream);
    if (!sync) return;
    MI_HIP(hipStreamSynchronize(h->stream));
    h->stats.call_ms = h->call_timer.elapsed_ms();
    h->stats.kernel_ms = h->dispatch_timers.total_ms();
    h->stats.n_timed = h->dispatch_timers.used;
    h->stats.n_launches = h->launches_acc;
    h->stats.n_units = h->rows_acc;
    h->stats.algorithmic_bytes = h->bytes_acc;
    h->stats.algorithmic_flops = h->flops_acc;
    h->stats.loss = 0;
}

}  // namespace

extern "C" int mi355rec_ials_create(mi355rec_ials_t *out, int32_t n_users, int32_t n_items, int32_t n_factors, double reg,
                                    const int32_t *indptr, const int32_t *indices, const float *confidence, const double *U0,
                                    const double *V0) {
    return guarded([&] {
        MI_REQUIRE(out && indptr && indices && confidence && V0, "NULL argument");
        MI_REQUIRE(n_users > 0 && n_items > 0, "empty URM");
        MI_REQUIRE(n_factors >= 1, "num_factors must be >= 1");
        if (n_factors > 255)   // 16 x 16 tiles of 16 (the rhs row included): 17 tiles per wavefront; the search space stops at 200
            fail(MI355REC_E_UNSUPPORTED, "num_factors = %d: the register-resident solver covers num_factors <= 255", n_factors);
        ensure_device();
        std::unique_ptr<mi355rec_ials> h(new mi355rec_ials());
        h->n_users = n_users;
        h->n_items = n_items;
        h->k = n_factors;
        h->reg = reg;
        h->nnz = (size_t)indptr[n_users];
        MI_REQUIRE(h->nnz > 0, "URM has no interactions");
        MI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->call_timer.init();
        h->dispatch_timers.reserve(8);
        hipStream_t s = h->stream;
        const size_t nnz = h->nnz;
        h->u_ptr.upload(indptr, (size_t)n_users + 1, s);
        h->u_idx.upload(indices, nnz, s);
        h->u_conf.upload(confidence, nnz, s);
        // item-major view (the reference keeps C_csc, IALSRecommender.py:106), built on the device
        DeviceBuffer<int> cnt, cursor;
        cnt.alloc_zero((size_t)n_items, s);
        cursor.alloc((size_t)n_items);
        h->i_ptr.alloc((size_t)n_items + 1);
        h->i_idx.alloc(nnz);
        h->i_conf.alloc(nnz);
        const int eb = 256, eg = (int)std::min<size_t>((nnz + eb - 1) / eb, 4096);
        hipLaunchKernelGGL(ials_count_kernel, dim3(eg), dim3(eb), 0, s, h->u_idx.ptr, nnz, cnt.ptr);
        hipLaunchKernelGGL(ials_scan_kernel, dim3(1), dim3(1024), 0, s, cnt.ptr, h->i_ptr.ptr, cursor.ptr, n_items);
        hipLaunchKernelGGL(ials_scatter_kernel, dim3(div_up((int64_t)n_users * 64, 256)), dim3(256), 0, s, h->u_ptr.ptr,
                           h->u_idx.ptr, h->u_conf.ptr, n_users, cursor.ptr, h->i_idx.ptr, h->i_conf.ptr);
        MI_HIP(hipGetLastError());
        const size_t nu = (size_t)n_users * h->k, ni = (size_t)n_items * h->k;
        if (U0) h->U.upload(U0, nu, s); else h->U.alloc_zero(nu, s);
        h->V.upload(V0, ni, s);
        h->G.alloc((size_t)h->k * h->k);
        h->queue.alloc(1);
        if (getenv("MI355REC_IALS_PHASES")) h->phases.alloc_zero(8, s);
        h->u_ptr_host.assign(indptr, indptr + n_users + 1);
        h->i_ptr_host.resize((size_t)n_items + 1);
        h->i_ptr.download(h->i_ptr_host.data(), (size_t)n_items + 1, s);
        MI_HIP(hipStreamSynchronize(s));
        auto by_length = [](const std::vector<int> &ptr, int n) {
            std::vector<int> o(n);
            std::iota(o.begin(), o.end(), 0);
            std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return ptr[a + 1] - ptr[a] > ptr[b + 1] - ptr[b]; });
            return o;
        };
        h->user_order = by_length(h->u_ptr_host, n_users);
        h->item_order = by_length(h->i_ptr_host, n_items);
        *out = h.release();
    });
}

extern "C" int mi355rec_ials_run_epochs(mi355rec_ials_t h, int32_t n_epochs) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        MI_REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        ensure_device();
        ReleaseScope scope(h->stream);
        h->dispatch_timers.reserve(2 * std::max(1, n_epochs));
        begin_call(h, std::max(h->n_users, h->n_items));
        for (int e = 0; e < n_epochs; ++e) {
            half_step(h, true, 0, h->n_users);      // fit user factors against V     (IALSRecommender.py:141-152)
            half_step(h, false, 0, h->n_items);     // then item factors against the UPDATED U (:156-166)
        }
        end_call(h, true);
    });
}

extern "C" int mi355rec_ials_user_half(mi355rec_ials_t h, int32_t u0, int32_t u1) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        ReleaseScope scope(h->stream);
        h->dispatch_timers.reserve(2);
        begin_call(h, std::max(0, u1 - u0));
        half_step(h, true, u0, u1);
        end_call(h, false);
    });
}

extern "C" int mi355rec_ials_item_half(mi355rec_ials_t h, int32_t i0, int32_t i1) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        ReleaseScope scope(h->stream);
        h->dispatch_timers.reserve(2);
        begin_call(h, std::max(0, i1 - i0));
        half_step(h, false, i0, i1);
        end_call(h, false);
    });
}

extern "C" int mi355rec_ials_device_factors(mi355rec_ials_t h, double **d_U, double **d_V) {
    return guarded([&] {
        MI_REQUIRE(h && d_U && d_V, "NULL argument");
        *d_U = h->U.ptr;
        *d_V = h->V.ptr;
    });
}

extern "C" int mi355rec_ials_sync(mi355rec_ials_t h) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        MI_HIP(hipStreamSynchronize(h->stream));
        h->stats.call_ms = h->call_timer.elapsed_ms();
        h->stats.kernel_ms = h->dispatch_timers.total_ms();
        h->stats.n_timed = h->dispatch_timers.used;
        h->stats.n_launches = h->launches_acc;
        h->stats.n_units = h->rows_acc;
        h->stats.algorithmic_bytes = h->bytes_acc;
        h->stats.algorithmic_flops = h->flops_acc;
    });
}

extern "C" int mi355rec_ials_get_factors(mi355rec_ials_t h, double *U, double *V) {
    return guarded([&] {
        MI_REQUIRE(h, "NULL handle");
        ensure_device();
        if (U) h->U.download(U, (size_t)h->n_users * h->k, h->stream);
        if (V) h->V.download(V, (size_t)h->n_items * h->k, h->stream);
        MI_HIP(hipStreamSynchronize(h->stream));
    });
}

extern "C" int mi355rec_ials_schedule_info(mi355rec_ials_t h, int32_t *n_split_rows, int32_t *n_parts) {
    return guarded([&] {
        MI_REQUIRE(h && n_split_rows && n_parts, "NULL argument");
        *n_split_rows = h->n_split_rows;
        *n_parts = h->n_part_items;
    });
}

extern "C" int mi355rec_ials_get_stats(mi355rec_ials_t h, mi355rec_stats *stats) {
    return guarded([&] {
        MI_REQUIRE(h && stats, "NULL argument");
        *stats = h->stats;
    });
}

extern "C" void mi355rec_ials_destroy(mi355rec_ials_t h) {
    if (!h) return;
    ReleaseScope scope(h->stream);
    delete h;
}
