// wave.cuh -- wavefront reductions without LDS traffic (gfx950), shared by the SGD kernels.
#pragma once

#include <hip/hip_runtime.h>

namespace mi355rec {

// ---- wavefront reductions without LDS traffic ----------------------------------------------------------------------
// row_ror:n rotates inside each row of 16 lanes; v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows / halves.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float swap16_sum(float v) {   // lane l gets v[l] + v[l ^ 16] (same operand order in both rows)
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float swap32_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ double swap16_sum(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double a = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[0] << 32) | (unsigned)lo[0]);
    const double c = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[1] << 32) | (unsigned)lo[1]);
    return a + c;
}
__device__ __forceinline__ double swap32_sum(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double a = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[0] << 32) | (unsigned)lo[0]);
    const double c = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi[1] << 32) | (unsigned)lo[1]);
    return a + c;
}
// sum over aligned groups of LPR lanes, every lane of the group gets the (bitwise identical) result
template <int LPR, class T> __device__ __forceinline__ T group_sum(T v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1
    if (LPR >= 32) v = swap16_sum(v);
    if (LPR >= 64) v = swap32_sum(v);
    return v;
}
// sum ACROSS the 64 / LPR groups (lane l of every group gets the total of the lanes l of all groups)
template <int LPR, class T> __device__ __forceinline__ T cross_group_sum(T v) {
    if (LPR <= 16) v = swap16_sum(v);
    if (LPR <= 32) v = swap32_sum(v);
    return v;
}
template <class T> __device__ __forceinline__ T wave_sum(T v) { return group_sum<64>(v); }

}  // namespace mi355rec
