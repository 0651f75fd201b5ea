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

// torch.max keeps a NaN (pointnet2_modules.py:288, 296) and torch.topk ranks NaN above every number: a point whose class
// scores or stability hold a NaN is sampled first.  fmaxf would drop the NaN and sample by the remaining classes.
__device__ __forceinline__ float nan_max(float m, float v) { return (v > m || v != v) ? v : m; }
// scores are products of sigmoids, >= +0: their bit patterns order like the numbers; every NaN becomes the one quiet NaN
// pattern, which lies above all of them (ties among NaNs: lower index first, like any other tie)
__device__ __forceinline__ unsigned score_bits(float s) { return s != s ? 0x7fc00000u : (unsigned)__float_as_int(s); }

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
            for (int ch = 1; ch < c; ++ch) mx = nan_max(mx, cls[(size_t)e * c + ch]);
            float s = sps_sigmoid(mx);
            if (stds) {
                const float sta = 1.0f - sps_sigmoid(stds[(size_t)scene * n + e] / 8.0f - 3.0f);
                s = s * sta;
            }
            if (score_out) score_out[(size_t)scene * n + e] = s;
            key = ((unsigned long long)score_bits(s) << 32) | (unsigned)(~e);
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

// The same result by RANKING instead of sorting, for scenes of up to 2048 points (IA-SSD layers 2-3: 1024 / 512): a
// workgroup of 4 waves owns 64 points; every wave counts, over a quarter of the scene, the keys greater than its lane's key
// (the rank in the descending order -- keys are distinct); the four partial ranks meet in LDS and the points ranked below k
// write idx[rank] -- and, fused, the centroid row new_xyz[rank] = xyz[point] (the gather_operation that follows the sampler,
// pointnet2_modules.py:423-424).  The bitonic sort above is 55 barrier-separated passes of one workgroup per scene (18 us
// at 1024 points); this is one pass over 16 x more workgroups.
constexpr int RANK_MAX_N = 4096;   // (16 KiB of keys)
__global__ __launch_bounds__(256) void score_topk_rank_kernel(
    int n, int c, int k, const float *__restrict__ cls, const float *__restrict__ stds, const float *__restrict__ xyz,
    int *__restrict__ idx, float *__restrict__ new_xyz, float *__restrict__ score_out) {
    __shared__ __attribute__((aligned(16))) unsigned sbits[RANK_MAX_N + 4];
    __shared__ int partial[4][64];
    const int scene = blockIdx.y;
    cls += (size_t)scene * n * c;
    const int npad = (n + 3) & ~3;
    for (int e = threadIdx.x; e < npad; e += blockDim.x) {
        unsigned bits = 0u;   // padding: score +0 with index >= n never outranks a real entry of equal score (index rule)
        if (e < n) {
            float mx = cls[(size_t)e * c];
            for (int ch = 1; ch < c; ++ch) mx = nan_max(mx, cls[(size_t)e * c + ch]);
            float s = sps_sigmoid(mx);
            if (stds) {
                const float sta = 1.0f - sps_sigmoid(stds[(size_t)scene * n + e] / 8.0f - 3.0f);
                s = s * sta;
            }
            if (score_out && blockIdx.x == 0) score_out[(size_t)scene * n + e] = s;
            bits = score_bits(s);
        }
        sbits[e] = bits;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                  // my point
    const unsigned si = i < n ? sbits[i] : 0u;
    const int quarter = ((npad / 4) + 3) & ~3;
    const int jbeg = seg * quarter, jend = min(n, jbeg + quarter);
    int rank = 0;
    for (int j = jbeg; j < jend; j += 4) {                  // LDS broadcast reads: every lane compares with the same four keys
        const uint4 v = *reinterpret_cast<const uint4 *>(&sbits[j]);
        // key_j > key_i  <=>  s_j > s_i, or s_j == s_i and j < i   (key = score bits : ~index)
        rank += (int)((v.x > si) | ((v.x == si) & (j + 0 < i))) & (int)(j + 0 < jend);
        rank += (int)((v.y > si) | ((v.y == si) & (j + 1 < i))) & (int)(j + 1 < jend);
        rank += (int)((v.z > si) | ((v.z == si) & (j + 2 < i))) & (int)(j + 2 < jend);
        rank += (int)((v.w > si) | ((v.w == si) & (j + 3 < i))) & (int)(j + 3 < jend);
    }
    partial[seg][lane] = rank;
    __syncthreads();
    if (seg == 0 && i < n) {
        rank = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
        if (rank < k) {
            idx[(size_t)scene * k + rank] = i;
            if (new_xyz) {
                const float *p = xyz + ((size_t)scene * n + i) * 3;
                float *o = new_xyz + ((size_t)scene * k + rank) * 3;
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            }
        }
    }
}

}  // namespace sps

// sps_score_topk with the centroid gather fused: xyz (b, n, 3) -> new_xyz (b, k, 3) = xyz[idx] (both may be NULL).
extern "C" int sps_score_topk_gather(int b, int n, int c, int k, const float *cls, const float *stds, const float *xyz,
                                     int *idx, float *new_xyz, float *score_out, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || c <= 0 || k < 0 || k > n)
        return fail(SPS_ERR_INVALID, "score_topk: bad shape b=%d n=%d c=%d k=%d", b, n, c, k);
    if (b == 0) return SPS_OK;
    if (!cls || (!idx && k > 0) || (new_xyz && !xyz)) return fail(SPS_ERR_INVALID, "score_topk: null pointer");
    if (n > RANK_MAX_N || b > 65535) {   // large scenes: the sorting kernel, then the plain gather
        const int rc = sps_score_topk(b, n, c, k, cls, stds, idx, score_out, stream);
        if (rc != SPS_OK || !new_xyz) return rc;
        return sps_gather_xyz(b, n, k, xyz, idx, new_xyz, stream);
    }
    hipLaunchKernelGGL(score_topk_rank_kernel, dim3(divup(n, 64), b), dim3(256), 0, as_stream(stream), n, c, k, cls, stds, xyz,
                       idx, new_xyz, score_out);
    return check_launch("score_topk_rank_kernel");
}

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
