// Latency of the two ways to order dependent phases on gfx950 (diagnostic, not part of the library):
//   A. a chain of dependent (empty) kernel nodes replayed from a hipGraph          -> us per launch boundary
//   B. a grid-wide barrier inside one persistent kernel (one workgroup per CU, atomic arrive + spin, with the
//      release / acquire fences data exchange needs)                               -> us per barrier
// The BPR-MF epoch is a chain of 2 x 139 dependent phases per epoch; this decides whether a persistent kernel can
// beat graph replay.  Spins are bounded, so a mis-launch cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

template <bool FENCES>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned *count, int iters, int *timed_out, float *data) {
    const unsigned n = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        if (data) data[(blockIdx.x * 256 + threadIdx.x + it) & 65535] += 1.f;   // something to publish
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCES) __threadfence();
            atomicAdd(count, 1u);
            const unsigned target = (unsigned)(it + 1) * n;
            long spins = 0;
            while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 20000000) { *timed_out = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (FENCES) __threadfence();
        }
        __syncthreads();
        if (*timed_out) return;
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.gcnArchName, cus);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;

    // ---- A: graph of N dependent empty kernels
    for (int threads : {64, 1024}) {
        for (int grid : {1, 256, 1000}) {
            const int N = 2000;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, s, (int *)nullptr);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            printf("A graph chain   grid %4d x %4d threads: %.3f us per dependent launch\n", grid, threads, ms * 1e3 / N);
            // plain stream launches for comparison
            CK(hipEventRecord(a, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, s, (int *)nullptr);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            printf("A stream chain  grid %4d x %4d threads: %.3f us per launch\n", grid, threads, ms * 1e3 / N);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }

    // ---- B: grid barrier, one workgroup per CU (co-resident by construction)
    unsigned *count; int *flag; float *data;
    CK(hipMalloc(&count, 4)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&data, 65536 * 4));
    CK(hipMemset(data, 0, 65536 * 4));
    for (int grid : {64, cus}) {
        for (int fences = 0; fences < 2; ++fences) {
            const int iters = 2000;
            CK(hipMemsetAsync(count, 0, 4, s)); CK(hipMemsetAsync(flag, 0, 4, s));
            CK(hipEventRecord(a, s));
            if (fences) hipLaunchKernelGGL(barrier_kernel<true>, dim3(grid), dim3(256), 0, s, count, iters, flag, data);
            else hipLaunchKernelGGL(barrier_kernel<false>, dim3(grid), dim3(256), 0, s, count, iters, flag, (float *)nullptr);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            int to = 0;
            CK(hipMemcpy(&to, flag, 4, hipMemcpyDeviceToHost));
            printf("B grid barrier  %3d workgroups, fences %d: %.3f us per barrier%s\n", grid, fences, ms * 1e3 / iters, to ? "  (TIMED OUT)" : "");
        }
    }
    return 0;
}
