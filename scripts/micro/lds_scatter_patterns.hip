// ds_add_u32 scatter rate on gfx950 by address pattern (diagnostic, not part of the library).  What the similarity kernel's
// accumulation could gain from re-ordering its id stream, and what hot cells cost:
//   uniform     every lane a uniformly random cell of a 26 744-cell accumulator (the microbenchmark behind the "ds_add_u32 rate")
//   own-bank    lane l only touches cells with cell % 64 == l % 64: no two lanes of an instruction share a bank (upper bound of
//               any bank-aware ordering)
//   zipf        cells drawn from a Zipf(1) popularity over the accumulator (the items of a recommender profile), random lanes:
//               bank conflicts AND same-address collisions inside an instruction
//   zipf-head   the same without the 1024 most popular cells (what is left if the head is counted some other way)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_scatter_patterns.hip -o /tmp/lsp && /tmp/lsp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

constexpr int CELLS = 26744, THREADS = 1024, PER_THREAD = 8;

// ids[iter][e][thread]: one coalesced load per instruction slot
__global__ __launch_bounds__(THREADS) void scatter(const unsigned short *ids, int iters, unsigned *sink) {
    __shared__ unsigned acc[CELLS];
    for (int i = threadIdx.x; i < CELLS; i += THREADS) acc[i] = 0;
    __syncthreads();
    const unsigned short *p = ids + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        unsigned short id[PER_THREAD];
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) id[e] = p[((size_t)(it & 63) * PER_THREAD + e) * THREADS];
#pragma unroll
        for (int e = 0; e < PER_THREAD; ++e) atomicAdd(&acc[id[e]], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc[blockIdx.x % CELLS];
}

int main() {
    std::mt19937 rng(7);
    std::vector<double> cdf(CELLS);
    double z = 0;
    for (int r = 0; r < CELLS; ++r) { z += 1.0 / (r + 1); cdf[r] = z; }
    auto zipf = [&](int skip) {
        std::uniform_real_distribution<double> U(skip ? cdf[skip - 1] : 0.0, z);
        const double x = U(rng);
        return (int)(std::lower_bound(cdf.begin(), cdf.end(), x) - cdf.begin());
    };
    std::vector<int> perm(CELLS);          // popularity rank -> cell id (random placement of the popular items over the banks)
    for (int i = 0; i < CELLS; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    const size_t n = (size_t)64 * PER_THREAD * THREADS;
    const char *names[4] = {"uniform", "own-bank", "zipf", "zipf-head"};
    unsigned *sink;
    hipMalloc(&sink, 256 * 4);
    unsigned short *d;
    hipMalloc(&d, n * 2);
    for (int pat = 0; pat < 4; ++pat) {
        std::vector<unsigned short> ids(n);
        for (size_t q = 0; q < n; ++q) {
            const int thread = (int)(q % THREADS);
            int c;
            if (pat == 0) c = (int)(rng() % CELLS);
            else if (pat == 1) c = (int)((rng() % (CELLS / 64)) * 64 + thread % 64);
            else c = perm[zipf(pat == 3 ? 1024 : 0)];
            ids[q] = (unsigned short)c;
        }
        hipMemcpy(d, ids.data(), n * 2, hipMemcpyHostToDevice);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        const int iters = 4096;
        hipLaunchKernelGGL(scatter, dim3(256), dim3(THREADS), 0, 0, d, 64, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(scatter, dim3(256), dim3(THREADS), 0, 0, d, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double adds = 256.0 * THREADS * PER_THREAD * iters;
        printf("%-10s %8.3f ms  %7.2f G lane-adds/s  = %.2f per CU per ns\n", names[pat], ms, adds / ms * 1e-6, adds / ms * 1e-6 / 256);
    }
    return 0;
}
