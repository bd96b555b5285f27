// lds_atomic_lanes.hip -- does a wave64 ds_add_u32 cost per INSTRUCTION or per live LANE?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_atomic_lanes.hip -o /tmp/lds_atomic_lanes && /tmp/lds_atomic_lanes
// One 1024-thread workgroup per CU (the column kernel's shape), every wavefront issues the same number of ds_add_u32 instructions
// to uniformly random cells of a 128 KiB array with only the first `live` lanes of every wavefront enabled (live = 8 .. 64), and
// in a second series with the live lanes spread over the wavefront (every (64 / live)-th lane).  Reported: instructions per ns and
// CU, lane-adds per ns and CU.  If the unit is bound per instruction the first figure is flat and the second grows with `live`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool SPREAD>
__global__ __launch_bounds__(1024) void kern(int iters, int live, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned cells[];
    constexpr unsigned CELLS = 128 * 1024 / sizeof(unsigned);
    for (unsigned i = threadIdx.x; i < CELLS; i += 1024) cells[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = SPREAD ? (lane % (64 / live) == 0) : (lane < live);
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            if (on) atomicAdd(&cells[(s >> 8) % CELLS], s & 7u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = cells[blockIdx.x % CELLS];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 4096;
    unsigned *sink;
    CK(hipMalloc(&sink, cus * sizeof(unsigned)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("%s, %d CUs; 16 wavefronts per CU, %d x 8 ds_add_u32 each, uniformly random cells\n", prop.name, cus, iters);
    for (int spread = 0; spread < 2; ++spread)
        for (int live : {8, 16, 24, 32, 40, 48, 56, 64}) {
            if (spread && 64 % live) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a, 0));
                if (spread) hipLaunchKernelGGL(kern<true>, dim3(cus), dim3(1024), 128 * 1024, 0, iters, live, sink);
                else hipLaunchKernelGGL(kern<false>, dim3(cus), dim3(1024), 128 * 1024, 0, iters, live, sink);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            const double instr = 16.0 * iters * 8.0;                 // per CU
            printf("%-10s live %2d: %7.3f ms  %.3f instructions per ns and CU  %6.2f lane-adds per ns and CU\n", spread ? "spread" : "contiguous", live, best,
                   instr / (best * 1e6), instr * live / (best * 1e6));
        }
    return 0;
}
