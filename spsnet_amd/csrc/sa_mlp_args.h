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
    int *overflow;                 // split-fp16 kernel: set to 1 if an operand exceeded the exactly splittable range
};


extern int g_mlp_f16;
// sa_mlp_f16.hip: split-fp16 variant; same argument block (units = scenes, ups = centroids per scene on entry)
int launch_sa_mlp_f16(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st);
// sa_mlp_f16_lds.hip: the same arithmetic with the weight stream shared through LDS (a.w1 = concatenated stream)
int launch_sa_mlp_f16_lds(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st);

}  // namespace sps
