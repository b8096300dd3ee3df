// Device-side helpers shared by the single-workgroup kernels: DPP / readlane reductions
// (no LDS traffic, no ds_bpermute latency) and Newton-refined f64 reciprocal / rsqrt.
// gfx950 (wave64, GFX9 DPP controls) only.
#pragma once

#include <hip/hip_runtime.h>

namespace rc {

// ---- DPP moves -----------------------------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ inline float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ inline int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ inline double dpp_mov(double v) {
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)rhi << 32) | (unsigned int)rlo);
}

// sum over an aligned group of W lanes (W = 4, 8 or 16); every lane of the group gets the
// total, bitwise identical across the group (the butterfly is symmetric)
template <int W, typename T>
__device__ inline T group_sum_dpp(T v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    if (W >= 8) v += dpp_mov<0x141>(v);
    if (W >= 16) v += dpp_mov<0x140>(v);
    return v;
}
template <int W, typename T>
__device__ inline T group_max_dpp(T v) {
    v = max(v, dpp_mov<0xB1>(v));
    v = max(v, dpp_mov<0x4E>(v));
    if (W >= 8) v = max(v, dpp_mov<0x141>(v));
    if (W >= 16) v = max(v, dpp_mov<0x140>(v));
    return v;
}
template <int W>
__device__ inline int group_min_dpp(int v) {
    v = min(v, dpp_mov<0xB1>(v));
    v = min(v, dpp_mov<0x4E>(v));
    if (W >= 8) v = min(v, dpp_mov<0x141>(v));
    if (W >= 16) v = min(v, dpp_mov<0x140>(v));
    return v;
}

// ---- readlane (wave-uniform result) ------------------------------------------------
__device__ inline int read_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ inline float read_lane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ inline double read_lane(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// full-wave (64 lanes) reductions: DPP inside the four rows, then four readlanes
template <typename T>
__device__ inline T wave_sum_dpp(T v) {
    v = group_sum_dpp<16>(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
template <typename T>
__device__ inline T wave_max_dpp(T v) {
    v = group_max_dpp<16>(v);
    return max(max(read_lane(v, 0), read_lane(v, 16)), max(read_lane(v, 32), read_lane(v, 48)));
}
__device__ inline int wave_min_dpp(int v) {
    v = group_min_dpp<16>(v);
    return min(min(read_lane(v, 0), read_lane(v, 16)), min(read_lane(v, 32), read_lane(v, 48)));
}

// ---- fast, fully accurate reciprocal / reciprocal square root --------------------------
// v_rcp_f64 / v_rsq_f64 deliver ~2^-23 relative accuracy; two Newton steps reach the f64
// rounding level (error <= ~2 ulp) at a fraction of the IEEE division / sqrt expansions.
__device__ inline double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ inline float fast_rcp(float x) { return 1.0f / x; }
__device__ inline double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    // r <- r * (1.5 - 0.5 x r^2)
    r = r * fma(-0.5 * x, r * r, 1.5);
    r = r * fma(-0.5 * x, r * r, 1.5);
    return r;
}
__device__ inline float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }

}  // namespace rc
