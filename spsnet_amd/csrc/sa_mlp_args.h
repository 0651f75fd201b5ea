// sa_mlp_args.h -- kernel argument block shared by the fp32 and split-fp16 grouped-MLP kernels.
#pragma once
#include "sps_common.h"

namespace sps {

struct SaMlpArgs {
    int n, m, c_feat, units;       // points/scene, centroids/scene, feature channels, total units
    int j0, ups;                   // centroid range start within every scene; units per scene
    int ks1;                       // layer-1 k-steps = ceil((3 + c_feat) / 4)
    int c3;                        // padded last-layer width (multiple of 16)
    int c3_real;                   // channels actually written
    int out_c_total, out_c_off;    // out is (B, out_c_total, M); this scale writes [off, off + c3_real)
    const float *xyz, *new_xyz, *feat;
    const int *idx;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;
    int feat_pm;                   // 1: `feat` is point-major (B, N, c_feat) and layer 1's channel order is [features, xyz]
                                   // (mode 3, pure fp16: `feat` points at halves)
    int *overflow;                 // split-fp16 kernel: set to 1 if an operand exceeded the exactly splittable range
};


// max over the 16 lanes of each DPP row (= the 16 columns of an MFMA tile) for the four rows a lane holds, valid in
// every lane: xor butterfly quad_perm [1,0,3,2] / [2,3,0,1] / row_half_mirror / row_mirror.  Written as v_max_f32_dpp:
// hipcc emits v_mov_b32_dpp + v_max_f32 per step from update_dpp + fmaxf (twice the VALU slots, and on this chip VALU
// work is not hidden behind other waves' MFMAs); the four independent chains cover each other's DPP wait states.
#ifdef __HIPCC__
typedef float sps_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sps_f32x4 row_allmax4(sps_f32x4 v) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
#define SPS_DPP_MAX4(CTRL)                                                    \
    "v_max_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n" SPS_DPP_MAX4("quad_perm:[1,0,3,2]") SPS_DPP_MAX4("quad_perm:[2,3,0,1]")
                 SPS_DPP_MAX4("row_half_mirror") SPS_DPP_MAX4("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef SPS_DPP_MAX4
    return (sps_f32x4){a, b, c, d};
}
#endif

extern int g_mlp_f16;
// sa_mlp_f16.hip: split-fp16 variant; same argument block (units = scenes, ups = centroids per scene on entry)
int launch_sa_mlp_f16(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st, bool pure = false);
// sa_mlp_f16_lds.hip: the same arithmetic with the weight stream shared through LDS (a.w1 = concatenated stream)
int launch_sa_mlp_f16_lds(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st);

}  // namespace sps
