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
// max over the 64 lanes, wave-uniform.  One v_max_i32_dpp per step: a lane without a DPP source (row_shr past the row
// start, rows outside row_mask) is write-disabled and keeps its value, which is what max(v, identity) would give.  hipcc
// turns update_dpp(identity, v) + max into v_mov + v_mov_dpp + v_max -- three dependent VALU slots per step, and these
// reductions ARE the critical path of an FPS round.  s_nop 1 = the two wait states between a VALU write and a DPP read.
__device__ __forceinline__ int wave_max_i32_id(int v) {
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
// three independent reductions interleaved: each chain's next step is three instructions away, so no wait states
__device__ __forceinline__ void wave_max_i32_id3(int &a, int &b, int &c) {
#define SPS_STEP3(CTRL)                                  \
    "v_max_i32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t" \
    "v_max_i32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t" \
    "v_max_i32_dpp %2, %2, %2 " CTRL " bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" SPS_STEP3("row_shr:1 row_mask:0xf") SPS_STEP3("row_shr:2 row_mask:0xf")
                 SPS_STEP3("row_shr:4 row_mask:0xf") SPS_STEP3("row_shr:8 row_mask:0xf")
                 SPS_STEP3("row_bcast:15 row_mask:0xa") SPS_STEP3("row_bcast:31 row_mask:0xc")
                 : "+v"(a), "+v"(b), "+v"(c));
#undef SPS_STEP3
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63);
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
// a bucket's box: min and max of x, y, z over the wave, six chains interleaved (no wait states), results wave-uniform.
// v_min/v_max_f32 return the non-NaN operand, like fminf/fmaxf: NaN coordinates do not poison a box.
__device__ __forceinline__ void wave_box6(float &lx, float &ly, float &lz, float &hx, float &hy, float &hz) {
#define SPS_STEP6(CTRL)                                  \
    "v_min_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %2, %2, %2 " CTRL " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %3, %3, %3 " CTRL " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %4, %4, %4 " CTRL " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %5, %5, %5 " CTRL " bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" SPS_STEP6("row_shr:1 row_mask:0xf") SPS_STEP6("row_shr:2 row_mask:0xf")
                 SPS_STEP6("row_shr:4 row_mask:0xf") SPS_STEP6("row_shr:8 row_mask:0xf")
                 SPS_STEP6("row_bcast:15 row_mask:0xa") SPS_STEP6("row_bcast:31 row_mask:0xc")
                 : "+v"(lx), "+v"(ly), "+v"(lz), "+v"(hx), "+v"(hy), "+v"(hz));
#undef SPS_STEP6
    lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lx), 63));
    ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly), 63));
    lz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lz), 63));
    hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hx), 63));
    hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hy), 63));
    hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hz), 63));
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
