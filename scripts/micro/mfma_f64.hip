// layout check of v_mfma_f64_16x16x4_f64 on gfx950: D = A(16x4) * B(4x16) + C
//   A: lane l holds A[m = l % 16][k = l / 16];  B: lane l holds B[k = l / 16][n = l % 16];
//   D: lane l, element i holds D[m = 4 * i + l / 16][n = l % 16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * i + l / 16) * 16 + l % 16] = c[i];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + i * 0.37; hB[i] = 2 - i * 0.11; }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[m * 4 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("mfma_f64_16x16x4 layout check: max err %g (%s)\n", err, err < 1e-9 ? "layout as stated" : "LAYOUT MISMATCH");
    return 0;
}
