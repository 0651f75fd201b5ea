// score_topk.hip -- fused "score + top-k" sampler of the SA module dispatcher.
//
// Replaces the chain of small torch kernels in the reference's
// PointnetSAModuleMSG_WithSampling.forward (pcdet/ops/pointnet2/pointnet2_batch/
// pointnet2_modules.py:287-303):
//     cls_features.max(-1) -> sigmoid -> [ * (1 - sigmoid(stds/8 - 3)) ] -> topk -> .int()
// by ONE launch, one workgroup per scene.
//
// torch leaves two things unspecified that an index-exact contract needs pinned:
//   * the last ulp of sigmoid (device libm) -> sps_sigmoid below uses only correctly rounded
//     fp32 operations (+,-,*,fma,/,rint), so host and device agree bit for bit;
//   * the order of equal scores in topk -> (score descending, index ascending).
// Scores are in [0,1], so their fp32 bit patterns order like unsigned integers; the sort key is
// (score_bits << 32) | ~index and the top-k is a descending bitonic sort of those keys in LDS.
#include "sps_common.h"

namespace sps {

__device__ __forceinline__ float sps_exp(float x) {
    x = x > 88.0f ? 88.0f : x;
    x = x < -87.0f ? -87.0f : x;
    const float nf = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(-nf, 0.693145751953125f, x);
    r = __builtin_fmaf(-nf, 1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const float scale = __int_as_float(((int)nf + 127) << 23);
    return p * scale;
}

__device__ __forceinline__ float sps_sigmoid(float x) { return 1.0f / (1.0f + sps_exp(-x)); }

constexpr int TOPK_MAX_N = 16384;

__global__ __launch_bounds__(1024) void score_topk_kernel(
    int n, int c, int k, int np2, const float *__restrict__ cls, const float *__restrict__ stds,
    int *__restrict__ idx, float *__restrict__ score_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int scene = blockIdx.x;
    cls += (size_t)scene * n * c;
    for (int e = threadIdx.x; e < np2; e += blockDim.x) {
        unsigned long long key = 0ull;  // padding sorts below every real entry
        if (e < n) {
            float mx = cls[(size_t)e * c];
            for (int ch = 1; ch < c; ++ch) mx = fmaxf(mx, cls[(size_t)e * c + ch]);
            float s = sps_sigmoid(mx);
            if (stds) {
                const float sta = 1.0f - sps_sigmoid(stds[(size_t)scene * n + e] / 8.0f - 3.0f);
                s = s * sta;
            }
            if (score_out) score_out[(size_t)scene * n + e] = s;
            key = ((unsigned long long)(unsigned)__float_as_int(s) << 32) | (unsigned)(~e);
        }
        keys[e] = key;
    }
    __syncthreads();
    // descending bitonic sort of np2 keys
    const int half = np2 >> 1;
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < half; t += blockDim.x) {
                const int i = 2 * stride * (t / stride) + (t % stride);
                const int l = i + stride;
                const bool desc = (i & size) == 0;
                const unsigned long long a = keys[i], bkey = keys[l];
                if ((a < bkey) == desc) { keys[i] = bkey; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < k; r += blockDim.x)
        idx[(size_t)scene * k + r] = (int)(~(unsigned)(keys[r] & 0xFFFFFFFFull));
}

}  // namespace sps

extern "C" int sps_score_topk(int b, int n, int c, int k, const float *cls, const float *stds, int *idx,
                              float *score_out, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || c <= 0 || k < 0 || k > n)
        return fail(SPS_ERR_INVALID, "score_topk: bad shape b=%d n=%d c=%d k=%d", b, n, c, k);
    if (n > TOPK_MAX_N) return fail(SPS_ERR_INVALID, "score_topk: n=%d exceeds %d", n, TOPK_MAX_N);
    if (b == 0) return SPS_OK;
    if (!cls || (!idx && k > 0)) return fail(SPS_ERR_INVALID, "score_topk: null pointer");
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    int threads = np2 / 2;
    threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : threads);
    const size_t lds = (size_t)np2 * sizeof(unsigned long long);
    if (lds > 64 * 1024) {
        static LdsLimitOnce raised;
        const int rc = raise_lds_limit((const void *)score_topk_kernel, 160 * 1024, raised, "score_topk");
        if (rc != SPS_OK) return rc;
    }
    hipLaunchKernelGGL(score_topk_kernel, dim3(b), dim3(threads), lds, as_stream(stream), n, c, k, np2, cls, stds,
                       idx, score_out);
    return check_launch("score_topk_kernel");
}
