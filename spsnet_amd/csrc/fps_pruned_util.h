// fps_pruned_util.h -- device helpers shared by the spatially pruned FPS kernels (fps_pruned.hip: a scene's points in
// the registers of one CU; fps_pruned_big.hip: larger scenes, points in L2).
#pragma once
#include "sps_common.h"
#include "spatial_grid.h"

#include <math.h>

namespace sps {

constexpr int PF_WAVES = 8;
constexpr int PF_THREADS = PF_WAVES * 64;

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_or(int v, int identity) {
    // lanes without a DPP source read `identity`
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_max_i32_id(int v) {
    constexpr int ID = (int)0x80000000;
    v = imax(v, dpp_or<DPP_ROW_SHR1>(v, ID));
    v = imax(v, dpp_or<DPP_ROW_SHR2>(v, ID));
    v = imax(v, dpp_or<DPP_ROW_SHR4>(v, ID));
    v = imax(v, dpp_or<DPP_ROW_SHR8>(v, ID));
    v = imax(v, dpp_or<DPP_ROW_BCAST15, 0xA>(v, ID));
    v = imax(v, dpp_or<DPP_ROW_BCAST31, 0xC>(v, ID));
    return __builtin_amdgcn_readlane(v, 63);
}
// v_min_f32 without the canonicalising v_max hipcc puts in front of fminf(): IEEE mode already returns the
// non-NaN operand, which is all the reference's min() needs
__device__ __forceinline__ float fmin_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int wave_min_i32_id(int v) { return ~wave_max_i32_id(~v); }

// wave-wide float min / max through DPP (rows, then row broadcasts; result read from lane 63): ~10 cycles per step
// where the ds_bpermute behind __shfl_xor costs ~60 -- the bucket boxes need 6 x 32 of these reductions per wave
template <bool MAX>
__device__ __forceinline__ float wave_all_f32(float v) {
    const int id = __float_as_int(MAX ? -INFINITY : INFINITY);
    auto step = [&](float o) { v = MAX ? fmaxf(v, o) : fminf(v, o); };
    step(__int_as_float(dpp_or<DPP_ROW_SHR1>(__float_as_int(v), id)));
    step(__int_as_float(dpp_or<DPP_ROW_SHR2>(__float_as_int(v), id)));
    step(__int_as_float(dpp_or<DPP_ROW_SHR4>(__float_as_int(v), id)));
    step(__int_as_float(dpp_or<DPP_ROW_SHR8>(__float_as_int(v), id)));
    step(__int_as_float(dpp_or<DPP_ROW_BCAST15, 0xA>(__float_as_int(v), id)));
    step(__int_as_float(dpp_or<DPP_ROW_BCAST31, 0xC>(__float_as_int(v), id)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_allmin_f32(float v) { return wave_all_f32<false>(v); }
__device__ __forceinline__ float wave_allmax_f32(float v) { return wave_all_f32<true>(v); }

// v[LANE] = value (wave-uniform), LANE a compile-time constant
template <int LANE>
__device__ __forceinline__ void put_lane(int &v, int value) {
    asm volatile("s_nop 0\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(value)), "n"(LANE));
}

// tie-break rank of point k under the reference's block size bs = 2^l2: bit-reversed (k mod bs), then k / bs
__device__ __forceinline__ unsigned pf_rank(unsigned k, int bs, int l2, int rb) {
    const unsigned lowrev = (l2 == 0) ? 0u : (__brev(k & (unsigned)(bs - 1)) >> (32 - l2));
    return (lowrev << rb) | (k >> l2);
}
__device__ __forceinline__ unsigned pf_unrank(unsigned rank, int l2, int rb) {
    const unsigned hi = rank >> rb, lo = rank & ((1u << rb) - 1u);
    const unsigned low = (l2 == 0) ? 0u : (__brev(hi) >> (32 - l2));
    return (lo << l2) | low;
}

}  // namespace sps
