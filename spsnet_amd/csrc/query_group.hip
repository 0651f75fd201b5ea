// query_group.hip -- fused QueryAndGroup: ball query + grouping of xyz (centred) and features.
//
// Replaces the per-scale sequence in the reference's QueryAndGroup.forward
// (pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py:299-322):
//     idx = ball_query(...)                    (zero-filled buffer + kernel)
//     xyz.transpose(1,2).contiguous()           (copy)
//     grouped_xyz = grouping_operation(...)     (kernel)
//     grouped_xyz -= new_xyz.transpose(...)     (kernel)
//     grouped_features = grouping_operation()   (kernel)
//     torch.cat([grouped_xyz, grouped_features])(copy of the whole grouped tensor)
// by the segmented ball-query kernel of ball_query.hip in its "write every row" mode (zeros for
// empty balls) and ONE grouping launch that writes the concatenated (B, 3+C, M, ns) tensor directly, reading xyz in its
// native (B,N,3) layout.
#include "sps_common.h"

namespace sps {

constexpr int QG_THREADS = 256;
constexpr int QG_CCHUNK = 16;

// channel chunk z = 0 additionally emits the 3 centred xyz channels when use_xyz
__global__ __launch_bounds__(QG_THREADS) void group_concat_kernel(
    int n, int m, int c, int nsample, int use_xyz, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ features,
    const int *__restrict__ idx, float *__restrict__ out) {
    const int scene = blockIdx.y;
    const int cols = m * nsample;
    const int e = blockIdx.x * QG_THREADS + threadIdx.x;
    if (e >= cols) return;
    const int src = idx[(size_t)scene * cols + e];
    const int cout = c + (use_xyz ? 3 : 0);
    float *o = out + (size_t)scene * cout * cols + e;
    if (use_xyz) {
        if (blockIdx.z == 0) {
            const int j = e / nsample;
            const float *p = xyz + ((size_t)scene * n + src) * 3;
            const float *q = new_xyz + ((size_t)scene * m + j) * 3;
            o[0] = p[0] - q[0];
            o[(size_t)cols] = p[1] - q[1];
            o[(size_t)2 * cols] = p[2] - q[2];
        }
        o += (size_t)3 * cols;
    }
    const int c0 = blockIdx.z * QG_CCHUNK;
    const int c1 = (c0 + QG_CCHUNK < c) ? c0 + QG_CCHUNK : c;
    const float *f = features + ((size_t)scene * c + c0) * n + src;
    o += (size_t)c0 * cols;
    for (int ch = c0; ch < c1; ++ch, f += n, o += cols) *o = *f;
}

}  // namespace sps

extern "C" int sps_ball_query_full(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                   const float *xyz, int *idx, sps_stream_t stream) {
    return sps::launch_ball_query(false, /*fill_empty=*/true, b, n, m, radius, 0.f, nsample, new_xyz, xyz, idx,
                                  sps::as_stream(stream));
}

extern "C" int sps_query_and_group(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                                   const float *xyz, const float *new_xyz, const float *features, int *idx,
                                   float *out, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || c < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "query_and_group: bad shape b=%d n=%d m=%d c=%d nsample=%d", b, n, m, c, nsample);
    if (!use_xyz && c == 0) return fail(SPS_ERR_INVALID, "query_and_group: no features and use_xyz == 0");
    if ((long long)m * nsample > 0x7FFFFFFFLL) return fail(SPS_ERR_INVALID, "query_and_group: m*nsample overflows int");
    if (b == 0 || m == 0 || nsample == 0) return SPS_OK;
    if (!xyz || !new_xyz || !idx || !out || (c > 0 && !features))
        return fail(SPS_ERR_INVALID, "query_and_group: null pointer");
    hipStream_t st = as_stream(stream);
    int rc = launch_ball_query(false, /*fill_empty=*/true, b, n, m, radius, 0.f, nsample, new_xyz, xyz, idx, st);
    if (rc != SPS_OK) return rc;
    return sps_group_concat(b, n, m, c, nsample, use_xyz, xyz, new_xyz, features, idx, out, stream);
}

// The grouping half of sps_query_and_group on neighbour indices the caller already has (a dual-radius scan, or the
// streamed queries of a training pass): out (B, 3+C, M, nsample) = [xyz[idx] - new_xyz ; features[idx]].
extern "C" int sps_group_concat(int b, int n, int m, int c, int nsample, int use_xyz, const float *xyz,
                                const float *new_xyz, const float *features, const int *idx, float *out,
                                sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || c < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "group_concat: bad shape b=%d n=%d m=%d c=%d nsample=%d", b, n, m, c, nsample);
    if (!use_xyz && c == 0) return fail(SPS_ERR_INVALID, "group_concat: no features and use_xyz == 0");
    if ((long long)m * nsample > 0x7FFFFFFFLL) return fail(SPS_ERR_INVALID, "group_concat: m*nsample overflows int");
    if (b == 0 || m == 0 || nsample == 0) return SPS_OK;
    if (!xyz || !new_xyz || !idx || !out || (c > 0 && !features))
        return fail(SPS_ERR_INVALID, "group_concat: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "group_concat: batch too large for the launch grid");
    hipStream_t st = as_stream(stream);
    const int cols = m * nsample;
    const int zchunks = c > 0 ? divup(c, QG_CCHUNK) : 1;
    hipLaunchKernelGGL(group_concat_kernel, dim3(divup(cols, QG_THREADS), b, zchunks), dim3(QG_THREADS), 0, st, n,
                       m, c, nsample, use_xyz, xyz, new_xyz, features, idx, out);
    return check_launch("group_concat_kernel");
}
