// HBM rate for the access pattern of the mini-batch kernels (diagnostic, not part of the library): rows of 512 B (k = 128 float32)
// at random positions of tables far larger than the memory-side cache, three rows read and three rows written (to a second
// buffer) per "sample", nothing else -- the ceiling a BPR mini-batch kernel can reach on this pattern.
//   lanes per row 32 (one 16-byte chunk per lane, as mf_batch_kernel<.., 4, 32, 1>), 2 samples per wavefront and round
//   sweep: rounds in flight per wavefront (1: load, wait, store, like the kernel; 2 / 4: software-pipelined), wavefronts per CU
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/row_gather_scatter.hip -o /tmp/rgs && /tmp/rgs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>

constexpr int K = 128;

template <int DEPTH>
__global__ __launch_bounds__(256) void rows_kernel(const float4 *src, float4 *dst, const int *ids, int n_samples) {
    const int lane = threadIdx.x & 63, g = lane >> 5, li = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int s0 = wave * 2 * DEPTH; s0 < n_samples; s0 += n_waves * 2 * DEPTH) {
        float4 v[DEPTH][3];
        int r[DEPTH][3];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = min(s0 + 2 * d + g, n_samples - 1);
#pragma unroll
            for (int e = 0; e < 3; ++e) r[d][e] = ids[3 * s + e];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int e = 0; e < 3; ++e) v[d][e] = src[(size_t)r[d][e] * (K / 4) + li];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                float4 o = v[d][e];
                o.x += 1.f;
                dst[(size_t)r[d][e] * (K / 4) + li] = o;
            }
    }
}

int main() {
    const size_t n_rows = 6u << 20;                      // 6 M rows x 512 B = 3 GB per buffer
    const int n_samples = 1 << 20;
    float4 *src, *dst;
    int *ids;
    hipMalloc(&src, n_rows * K * 4);
    hipMalloc(&dst, n_rows * K * 4);
    hipMalloc(&ids, sizeof(int) * 3 * n_samples);
    hipMemset(src, 0, n_rows * K * 4);
    hipMemset(dst, 0, n_rows * K * 4);
    std::mt19937 rng(3);
    std::vector<int> h(3 * (size_t)n_samples);
    for (auto &x : h) x = (int)(rng() % n_rows);
    hipMemcpy(ids, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)n_samples * 3 * 2 * K * 4;
    auto run = [&](auto kernel, const char *name, int wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        kernel<<<grid, 256>>>(src, dst, ids, n_samples);
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) kernel<<<grid, 256>>>(src, dst, ids, n_samples);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %2d workgroups of 256 per CU: %7.3f ms per pass, %6.2f TB/s (read + write)\n", name, wgs_per_cu, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    };
    for (int w : {2, 4, 8}) {
        run(rows_kernel<1>, "1 round in flight", w);
        run(rows_kernel<2>, "2 rounds in flight", w);
        run(rows_kernel<4>, "4 rounds in flight", w);
    }
    return 0;
}
