// ball_query.hip -- radius neighbour search with the reference's ordered-fill semantics.
//
// Replaces ball_query_kernel_fast / ball_query_dilated_kernel_fast (reference
// pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:9-45, 70-117):
//   row j of idx = the first `nsample` points, in ascending point index, with
//   d2 < radius^2 (strict, fp32 product); on the first hit the whole row is filled with
//   that index; rows of empty balls are NOT written (the caller zero-fills,
//   pointnet2_utils.py:246).  Dilated: a point is appended if d2 == 0 and, independently,
//   if min_r^2 <= d2 < max_r^2 (so a coincident point counts twice when min_r == 0).
//
// Mapping: one lane per centroid (64 centroids per workgroup so that even the 512-centroid
// layers spread over the chip); the scanned point is wave-uniform, so its coordinates
// arrive through the scalar cache (s_load) and feed the VALU as SGPR operands -- no LDS,
// no per-lane global reads (the reference re-reads 12 B/point/centroid through L2).
#include "sps_common.h"

namespace sps {

constexpr int BQ_THREADS = 64;
constexpr int BQ_CHUNK = 8;  // points between two wave-uniform "everyone full?" checks

template <bool DILATED>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    int n, int m, float r2max, float r2min, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx) {
    const int scene = blockIdx.y;
    const int j = blockIdx.x * BQ_THREADS + threadIdx.x;
    const bool active = j < m;
    xyz += (size_t)scene * n * 3;
    const float *ctr = new_xyz + ((size_t)scene * m + (active ? j : 0)) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
    int *row = idx + ((size_t)scene * m + (active ? j : 0)) * nsample;

    int cnt = active ? 0 : nsample;  // inactive lanes count as full
    int first = 0;
    for (int k0 = 0; k0 < n; k0 += BQ_CHUNK) {
        if (__all(cnt >= nsample)) break;
        const int kend = (k0 + BQ_CHUNK < n) ? k0 + BQ_CHUNK : n;
        for (int k = k0; k < kend; ++k) {
            const float d2 = sqdist(cx, cy, cz, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
            if (DILATED) {
                if (d2 == 0.f && cnt < nsample) {
                    if (cnt == 0) first = k;
                    row[cnt++] = k;
                }
                if (d2 >= r2min && d2 < r2max && cnt < nsample) {
                    if (cnt == 0) first = k;
                    row[cnt++] = k;
                }
            } else {
                if (d2 < r2max && cnt < nsample) {
                    if (cnt == 0) first = k;
                    row[cnt++] = k;
                }
            }
        }
    }
    if (active && cnt > 0)
        for (int l = cnt; l < nsample; ++l) row[l] = first;
}

static int launch_ball_query(bool dilated, int b, int n, int m, float rmax, float rmin, int nsample,
                             const float *new_xyz, const float *xyz, int *idx, hipStream_t st) {
    if (b < 0 || n < 0 || m < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "ball_query: bad shape b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (b == 0 || m == 0 || nsample == 0 || n == 0) return SPS_OK;
    if (!new_xyz || !xyz || !idx) return fail(SPS_ERR_INVALID, "ball_query: null pointer");
    dim3 grid(divup(m, BQ_THREADS), b), block(BQ_THREADS);
    // radius*radius in fp32, as the kernel computes it (ball_query_gpu.cu:23, 84-85)
    const float r2max = rmax * rmax, r2min = rmin * rmin;
    if (dilated)
        hipLaunchKernelGGL(ball_query_kernel<true>, grid, block, 0, st, n, m, r2max, r2min, nsample, new_xyz, xyz, idx);
    else
        hipLaunchKernelGGL(ball_query_kernel<false>, grid, block, 0, st, n, m, r2max, r2min, nsample, new_xyz, xyz, idx);
    return check_launch("ball_query_kernel");
}

}  // namespace sps

extern "C" int sps_ball_query_kernel_launcher_fast(int b, int n, int m, float radius, int nsample,
                                                   const float *new_xyz, const float *xyz, int *idx,
                                                   sps_stream_t stream) {
    return sps::launch_ball_query(false, b, n, m, radius, 0.f, nsample, new_xyz, xyz, idx, sps::as_stream(stream));
}

extern "C" int sps_ball_query_dilated_kernel_launcher_fast(int b, int n, int m, float max_radius,
                                                           float min_radius, int nsample,
                                                           const float *new_xyz, const float *xyz, int *idx,
                                                           sps_stream_t stream) {
    return sps::launch_ball_query(true, b, n, m, max_radius, min_radius, nsample, new_xyz, xyz, idx,
                                  sps::as_stream(stream));
}
