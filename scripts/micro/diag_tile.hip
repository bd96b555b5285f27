// diag_tile.hip -- the 16 x 16 diagonal tile of the IALS solve stage, stand-alone: the scalar factor + inverse against the version that
// inverts on the matrix pipe (csrc/ials_diag.cuh).  Checks L L^T = A, M L = I for both, their agreement, and times them (shader clock,
// one wavefront, 256 repetitions).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I recsys2019_deeplearning_evaluation_amd/csrc scripts/micro/diag_tile.hip -o /tmp/diag_tile && /tmp/diag_tile
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "ials_diag.cuh"

using namespace mi355rec;

__global__ void diag_kernel(const double *A, double *L_out, double *M_out, unsigned long long *cycles, int variant, int reps) {
    __shared__ double P[16 * TP], inv[16 * TP], scratch[48 * TP];
    const int lane = threadIdx.x;
    unsigned long long t0 = 0, total = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int q = lane; q < 256; q += 64) P[(q >> 4) * TP + (q & 15)] = A[q];
        __syncthreads();
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if (variant == 0) factor_and_invert_diagonal_tile(P, inv, lane);
        else factor_and_invert_diagonal_tile_mfma(P, inv, scratch, lane);
        __syncthreads();
        unsigned long long t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        total += t1 - t0;
    }
    for (int q = lane; q < 256; q += 64) {
        L_out[q] = P[(q >> 4) * TP + (q & 15)];
        M_out[q] = inv[(q >> 4) * TP + (q & 15)];
    }
    if (lane == 0) *cycles = total / reps;
}

int main() {
    std::mt19937_64 rng(5);
    std::normal_distribution<double> nd;
    std::vector<double> Y(64 * 16), A(256);
    for (auto &v : Y) v = nd(rng);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = i == j ? 1e-3 : 0.0;
            for (int k = 0; k < 64; ++k) s += Y[k * 16 + i] * Y[k * 16 + j];
            A[i * 16 + j] = s;
        }
    double *dA, *dL, *dM;
    unsigned long long *dC;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dL, 256 * 8); hipMalloc(&dM, 256 * 8); hipMalloc(&dC, 8);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    std::vector<double> L[2], M[2];
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(diag_kernel, dim3(1), dim3(64), 0, 0, dA, dL, dM, dC, variant, 256);
        hipDeviceSynchronize();
        L[variant].resize(256); M[variant].resize(256);
        unsigned long long cyc = 0;
        hipMemcpy(L[variant].data(), dL, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(M[variant].data(), dM, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost);
        double e_fac = 0, e_inv = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double llt = 0, ml = 0;
                for (int k = 0; k < 16; ++k) {
                    llt += L[variant][i * 16 + k] * L[variant][j * 16 + k];
                    ml += M[variant][i * 16 + k] * L[variant][k * 16 + j];
                }
                e_fac = std::fmax(e_fac, std::fabs(llt - A[i * 16 + j]) / std::fabs(A[i * 16 + i]));
                e_inv = std::fmax(e_inv, std::fabs(ml - (i == j ? 1.0 : 0.0)));
            }
        printf("%-28s %6llu cycles per tile   |L L^T - A| / diag %.2e   |M L - I| %.2e\n", variant ? "inverse on the matrix pipe" : "scalar (v_readlane)", cyc, e_fac, e_inv);
    }
    double dl = 0, dm = 0, ml = 0;
    for (int q = 0; q < 256; ++q) {
        dl = std::fmax(dl, std::fabs(L[0][q] - L[1][q]));
        dm = std::fmax(dm, std::fabs(M[0][q] - M[1][q]));
        ml = std::fmax(ml, std::fabs(M[0][q]));
    }
    printf("between the two: max |dL| %.2e, max |dM| %.2e (max |M| %.2e)\n", dl, dm, ml);
    return 0;
}
