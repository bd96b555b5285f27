// ials_diag.cuh -- the 16 x 16 diagonal tile of a Cholesky panel: factor and inverse by ONE wavefront (gfx950).
// Shared by csrc/ials.hip (the row solver of IALSRecommender._update_row, IALSRecommender.py:170-201) and the stand-alone check
// scripts/micro/diag_tile.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace mi355rec {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int TP = 17;                   // padded row length of a 16 x 16 tile staged in LDS (conflict-free operand reads)

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, src_lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// 1 / sqrt(d) in full double precision from v_rsq_f64 and two Newton steps (the IEEE sqrt + divide sequences are ~50
// instructions on the critical path of every pivot)
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    y = y * (1.5 - 0.5 * d * y * y);
    return y;
}

// The 16 x 16 diagonal tile of a panel (row-major in LDS, TP doubles per row) -> its Cholesky factor in place (upper part
// cleared) and the inverse of that factor in `inv_tile`.  One wavefront, no barriers inside.
__device__ __forceinline__ void factor_and_invert_diagonal_tile(double *P, double *inv_tile, int lane) {
    const int r = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = P[r * TP + c];
    double my_inv = 0.0;
    // right-looking, all in registers: lane r holds row r; the pivot and the scaled column travel between lanes as v_readlane
    // broadcasts (no LDS round trip on the dependent chain: 128 cycles per column when they did)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = lane_bcast(a[j], j);
        const double inv = fast_rsqrt(d);
        const double lj = r > j ? a[j] * inv : (r == j ? d * inv : 0.0);     // L[r][j]
        a[j] = lj;
        if (r == j) my_inv = inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] -= lj * lane_bcast(lj, c);      // only cells with c <= r are ever used
    }
    // the inverse, lane c computes column c: x starts as e_c; once x_m is final every later entry loses L[rr][m] x_m
    double x[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) x[rr] = r == rr ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        x[m] *= lane_bcast(my_inv, m);
        // (The broadcasts of column m of L depend on nothing the loop computes, and left to itself the compiler reads all 120 of
        // them into SGPRs up front -- twice as many as there are, so they are parked in VGPR lanes: 200 v_writelane + as many
        // v_readlane per tile on the one chain every other wavefront of the workgroup waits for.  Tying the column to x[m] keeps
        // each broadcast next to its use.)
        double am = a[m];
        asm volatile("" : "+v"(am) : "v"(x[m]));
#pragma unroll
        for (int rr = m + 1; rr < 16; ++rr) x[rr] -= lane_bcast(am, rr) * x[m];
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            P[r * TP + c] = c <= r ? a[c] : 0.0;                               // L_JJ, upper part cleared
            inv_tile[c * TP + r] = x[c];                                       // inverse[c][column r]
        }
    }
}


// The same factor, the inverse on the MATRIX pipe (the solve stage's panel wavefront: the pipe is idle while it works, and the
// scalar inversion above is half of the chain every other wavefront of the workgroup waits for).  With B = the four 4 x 4 diagonal
// blocks of L and N = the rest, L = B (I - W), W = -B^-1 N strictly block-lower, so W^4 = 0 and
//     L^-1 = (I + W)(I + W^2) B^-1.
// B^-1 (four 4 x 4 triangles: three dependent steps each, sixteen lanes) is scalar work; then four 16 x 16 x 16 products = sixteen
// v_mfma_f64_16x16x4, their operands re-laid-out through `scratch` (3 tiles of 16 x TP doubles).  MEASURED (scripts/micro/diag_tile.hip, one
// wavefront alone on a CU): correct to 3e-16 (|M L - I|), but 8 002 cycles per tile against 6 148 for the scalar version -- four dependent
// products, each behind an LDS round trip, cost what the 120 broadcast + multiply-add pairs of the scalar inversion cost.  NOT USED by
// ials.hip; kept with its check as the record of the attempt.
// Operand layouts of v_mfma_f64_16x16x4 (scripts/micro/mfma_f64.hip): lane (g = lane / 16, c = lane % 16) supplies A[c][4 q + g] and
// B[4 q + g][c] of chunk q and holds D[4 i + g][c], i = 0..3.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ d4 tile_product(const double *X, const double *Y, d4 acc, int g, int c, double sign = 1.0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * X[c * TP + 4 * q + g], Y[(4 * q + g) * TP + c], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void store_tile(double *Z, const d4 &v, int g, int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) Z[(4 * i + g) * TP + c] = v[i];
}
__device__ __forceinline__ void factor_and_invert_diagonal_tile_mfma(double *P, double *inv_tile, double *scratch, int lane) {
    const int r = lane & 15, g = lane >> 4;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = P[r * TP + c];
    double my_inv = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = lane_bcast(a[j], j);
        const double inv = fast_rsqrt(d);
        const double lj = r > j ? a[j] * inv : (r == j ? d * inv : 0.0);     // L[r][j]
        a[j] = lj;
        if (r == j) my_inv = inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] -= lj * lane_bcast(lj, c);      // only cells with c <= r are ever used
    }
    double *const Dm = scratch, *const Wm = scratch + 16 * TP, *const Vm = scratch + 32 * TP;    // B^-1 | W, later (I + W)(I + W^2) | W^2
    // L_JJ (upper part cleared), N = L without its diagonal blocks (into Vm for now), B^-1 zeroed, 1 / diagonal in the spare column
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double l = c <= r ? a[c] : 0.0;
            P[r * TP + c] = l;
            Vm[r * TP + c] = (c >> 2) < (r >> 2) ? l : 0.0;
            Dm[r * TP + c] = 0.0;
        }
        Dm[r * TP + 16] = my_inv;
    }
    wave_sync_lds();
    // B^-1: lane t < 16 takes column t % 4 of block t / 4
    if (lane < 16) {
        const int b4 = 4 * (lane >> 2), cc = lane & 3;
        double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            if (rr == cc) x[rr] = Dm[(b4 + rr) * TP + 16];
            else if (rr > cc) {
                double sum = 0.0;
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    if (kk >= cc && kk < rr) sum += P[(b4 + rr) * TP + b4 + kk] * x[kk];
                x[rr] = -sum * Dm[(b4 + rr) * TP + 16];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            if (rr >= cc) Dm[(b4 + rr) * TP + b4 + cc] = x[rr];
    }
    wave_sync_lds();
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const d4 W = tile_product(Dm, Vm, zero, g, r, -1.0);              // W = -B^-1 N
    store_tile(Wm, W, g, r);
    wave_sync_lds();
    const d4 W2 = tile_product(Wm, Wm, zero, g, r);
    store_tile(Vm, W2, g, r);                                            // (N is not needed any more)
    wave_sync_lds();
    d4 S = W + W2;
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] += 4 * i + g == r ? 1.0 : 0.0;
    S = tile_product(Wm, Vm, S, g, r);                                   // I + W + W^2 + W W^2
    wave_sync_lds();                                                     // (every lane has read W before it is overwritten)
    store_tile(Wm, S, g, r);
    wave_sync_lds();
    const d4 M = tile_product(Wm, Dm, zero, g, r);                       // L^-1 = S B^-1
    store_tile(inv_tile, M, g, r);                                       // row-major, like the scalar version's
}

}  // namespace mi355rec
