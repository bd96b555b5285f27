// LDS atomic throughput on gfx950: u32 / u64 / f32 / f64 adds to pseudo-random cells (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int MODE>
__global__ __launch_bounds__(1024) void k(int iters, unsigned long long *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    T *acc = reinterpret_cast<T *>(raw);
    const int cells = 128 * 1024 / sizeof(T);
    for (int i = threadIdx.x; i < cells; i += 1024) acc[i] = T(0);
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            const unsigned j = (s >> 8) % (unsigned)cells;
            if (MODE == 0) atomicAdd(&acc[j], T(1));
            else atomicAdd(&acc[j], T(s & 7));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = (unsigned long long)acc[blockIdx.x % cells];
}
template <typename T>
void run(const char *name) {
    unsigned long long *sink;
    hipMalloc(&sink, 256 * 8);
    auto kern = k<T, 1>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 128 * 1024, 0, 64, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 128 * 1024, 0, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double n = 256.0 * 1024 * iters * 8;
    printf("%-8s %8.3f ms  %.2f G lane-atomics/s  = %.2f per CU per ns\n", name, ms, n / ms * 1e-6, n / ms * 1e-6 / 256);
    hipFree(sink);
}
int main() {
    run<unsigned>("u32");
    run<unsigned long long>("u64");
    run<float>("f32");
    run<double>("f64");
    return 0;
}
