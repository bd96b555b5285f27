// kth_select.hip -- what the "K-th largest of the 1024 thread maxima" step of the threshold-first selection costs, stand-alone:
// block_kth_largest_prefix16 (two 8-bit passes, ballot-aggregated atomics) against block_kth_largest_bin12 (one 12-bit pass) and against
// pieces of them (barriers only; the histogram atomics on conflict-free addresses), for keys clustered like one column's maxima and for
// uniformly random keys.  One 1024-thread workgroup per CU, 140 KB of LDS each (the similarity kernel's occupancy).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I recsys2019_deeplearning_evaluation_amd/csrc scripts/micro/kth_select.hip -o /tmp/kth_select && /tmp/kth_select
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"

#include <cstdio>
#include <vector>

#include "topk.cuh"

using namespace mi355rec;

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int VARIANT>
__global__ __launch_bounds__(1024) void kth_kernel(unsigned long long *ticks, unsigned *result, int reps, int clustered) {
    extern __shared__ uint32_t lds[];
    uint32_t *aux = lds;                      // 8192 words
    __shared__ SelectScratch sc;
    const int tid = threadIdx.x;
    for (int w = tid; w < AUX_WORDS; w += 1024) aux[w] = 0u;
    __syncthreads();
    unsigned long long t0, t1;
    unsigned acc = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int rep = 0; rep < reps; ++rep) {
        const unsigned h = hash32((unsigned)tid * 2654435761u + (unsigned)rep * 40503u + blockIdx.x * 977u);
        const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
        float v = clustered ? 0.05f + 0.25f * u * u * u : u;
        if (clustered && (h & 15u) == 0u) v = 0.f;          // threads without a positive cell
        const uint32_t key = float_key(v);
        uint32_t p16 = 0;
        if (VARIANT == 0) {
            p16 = block_kth_largest_prefix16<1024>(key, 100u, aux, sc);
        } else if (VARIANT == 1) {
            p16 = block_kth_largest_bin12<1024>(key, 100u, aux, sc);
            if (rep < 64 && blockIdx.x < 4) {          // check: at least K keys at or above the bin's edge, fewer than K above the bin
                __shared__ unsigned n_ge, n_gt;
                if (tid == 0) { n_ge = 0; n_gt = 0; }
                __syncthreads();
                const unsigned d = (p16 & 0x7FFFu) >> 3, mine = (key >> 19) & 0xFFFu;
                if (mine >= d) atomicAdd(&n_ge, 1u);
                if (mine > d) atomicAdd(&n_gt, 1u);
                __syncthreads();
                if (tid == 0 && !(n_ge >= 100u && n_gt < 100u)) atomicAdd(&result[2], 1u);
                if (tid == 0) atomicAdd(&result[3], 1u);
                __syncthreads();
            }
            reinterpret_cast<uint4 *>(aux)[tid] = make_uint4(0u, 0u, 0u, 0u);          // (the caller's share: the bins are left dirty)
            __syncthreads();
        } else if (VARIANT == 2) {          // barriers only
            __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
            p16 = key >> 16;
        } else if (VARIANT == 3) {          // one atomic per thread on its own word + the four barriers
            atomicAdd(&aux[tid], 1u);
            __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
            p16 = aux[(tid * 7) & 1023];
        } else if (VARIANT == 4) {          // one atomic per thread where the 12-bit bins put it + the four barriers
            atomicAdd(&aux[(key >> 19) & 0xFFFu], 1u);
            __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
            p16 = aux[(tid * 7) & 1023];
        }
        acc += p16;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc == 0xFFFFFFFFu) result[0] = acc;
    if (tid == 0 && blockIdx.x == 0) result[1] = acc;
}

template <int VARIANT>
static void run(const char *what, int clustered) {
    const int reps = 2000, blocks = 256;
    unsigned long long *d_ticks;
    unsigned *d_res;
    hipMalloc(&d_ticks, blocks * 8);
    hipMalloc(&d_res, 16);
    hipMemset(d_res, 0, 16);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kth_kernel<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int warm = 0; warm < 2; ++warm) {
        hipLaunchKernelGGL(kth_kernel<VARIANT>, dim3(blocks), dim3(1024), 140 * 1024, 0, d_ticks, d_res, reps, clustered);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> t(blocks);
    hipMemcpy(t.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : t) sum += (double)x;
    unsigned res[4];
    hipMemcpy(res, d_res, 16, hipMemcpyDeviceToHost);
    printf("%-78s %7.0f shader cycles per call (s_memtime, mean over %d workgroups)", what, sum / blocks / reps, blocks);
    if (res[3]) printf("   checked %u calls: %u wrong", res[3], res[2]);
    printf("\n");
    hipFree(d_ticks);
    hipFree(d_res);
}

int main() {
    run<2>("four barriers only", 1);
    run<3>("four barriers + one LDS atomic per thread, own word", 1);
    run<4>("four barriers + one LDS atomic per thread, 12-bit bin of a clustered key", 1);
    run<0>("block_kth_largest_prefix16, clustered keys (a column's maxima)", 1);
    run<0>("block_kth_largest_prefix16, uniformly random keys", 0);
    run<1>("block_kth_largest_bin12 + zeroing its bins, clustered keys", 1);
    run<1>("block_kth_largest_bin12 + zeroing its bins, uniformly random keys", 0);
    return 0;
}
