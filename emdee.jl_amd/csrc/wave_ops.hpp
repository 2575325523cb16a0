// wave_ops.hpp -- wave64 cross-lane primitives for gfx950 (CDNA4).
// The reference exchanges per-lane values with 32-wide CUDA.shfl_sync (src/vec3.jl:9-13,
// src/lennard_jones.jl:20-23, src/nonbonded.jl:70-84).  Here a wavefront is 64 lanes and
// the exchange is done with DPP row operations (no LDS traffic) plus row broadcasts.
#pragma once

#include <hip/hip_runtime.h>

namespace emdee {

constexpr int WAVE = 64;

// DPP control words (GFX9 encoding)
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// v_mov_b32 with a DPP modifier; lanes whose source is out of range or masked off read 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true));
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wavefront; the total is valid in lane 63 only.
// 4 in-row shifted adds (row = 16 lanes), then row_bcast:15 into rows 1,3 and
// row_bcast:31 into rows 2,3.
template <typename T>
__device__ __forceinline__ T wave_sum_to_lane63(T v) {
    v += dpp_mov<DPP_ROW_SHR1, 0xf, 0xf>(v);
    v += dpp_mov<DPP_ROW_SHR2, 0xf, 0xf>(v);
    v += dpp_mov<DPP_ROW_SHR4, 0xf, 0xf>(v);
    v += dpp_mov<DPP_ROW_SHR8, 0xf, 0xf>(v);
    v += dpp_mov<DPP_ROW_BCAST15, 0xa, 0xf>(v);
    v += dpp_mov<DPP_ROW_BCAST31, 0xc, 0xf>(v);
    return v;
}

// Sum over groups of G consecutive lanes (G = 4, 8, 16, 32, 64); the group total is valid in
// the LAST lane of each group.
template <int G, typename T>
__device__ __forceinline__ T group_sum_to_last(T v) {
    v += dpp_mov<DPP_ROW_SHR1, 0xf, 0xf>(v);
    if (G >= 4) v += dpp_mov<DPP_ROW_SHR2, 0xf, 0xf>(v);
    if (G >= 8) v += dpp_mov<DPP_ROW_SHR4, 0xf, 0xf>(v);
    if (G >= 16) v += dpp_mov<DPP_ROW_SHR8, 0xf, 0xf>(v);
    if (G >= 32) v += dpp_mov<DPP_ROW_BCAST15, 0xa, 0xf>(v);
    if (G >= 64) v += dpp_mov<DPP_ROW_BCAST31, 0xc, 0xf>(v);
    return v;
}

__device__ __forceinline__ double readlane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Number of set bits of a 64-bit ballot below this lane (compaction offset).
__device__ __forceinline__ int prefix_popc(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

}  // namespace emdee
