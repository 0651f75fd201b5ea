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
// Design for gfx950:
//   * one LANE per centroid, 64 centroids per workgroup; the scanned point is wave-uniform, so
//     its coordinates arrive through the scalar cache (s_load_dwordx8, 8 points per batch) and feed
//     the VALU as SGPR operands -- the reference re-reads 12 B/point/centroid per thread via L2.
//   * the point range is cut into S segments scanned by S waves of the same workgroup, so that
//     even a 512-centroid layer puts several waves on every SIMD; each wave records its first
//     `nsample` hits per centroid in LDS ([slot][lane]: conflict-free), and the ordered semantics
//     are restored by concatenating the segments in order (prefix sum of the per-segment counts).
//   * the finished 64 x nsample block of idx is contiguous in HBM and is written coalesced from a
//     padded LDS image instead of one 4-byte store per hit per lane.
#include "sps_common.h"

namespace sps {

constexpr int BQ_LANES = 64;
constexpr int BQ_MAX_SEG = 8;
constexpr int BQ_BATCH = 8;  // points per scalar-load batch

__device__ __forceinline__ void bq_append(int *hits, int nsample, int lane, int k, int &cnt) {
    if (cnt < nsample) {
        hits[cnt * BQ_LANES + lane] = k;
        ++cnt;
    }
}

template <bool DILATED>
__device__ __forceinline__ void bq_test(float d2, float r2max, float r2min, int *hits, int nsample,
                                        int lane, int k, int &cnt) {
    if (DILATED) {
        if (d2 == 0.f) bq_append(hits, nsample, lane, k, cnt);
        if (d2 >= r2min && d2 < r2max) bq_append(hits, nsample, lane, k, cnt);
    } else {
        if (d2 < r2max) bq_append(hits, nsample, lane, k, cnt);
    }
}

// LDS layout (ints): hits[S][nsample][64] | cnt[S][64] | final[nsample][65]
template <bool DILATED>
__global__ __launch_bounds__(BQ_LANES *BQ_MAX_SEG) void ball_query_seg_kernel(
    int n, int m, int seg_len, float r2max, float r2min, int nsample, int fill_empty,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) int bq_lds[];
    const int S = blockDim.x / BQ_LANES;
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *hits = bq_lds + seg * nsample * BQ_LANES;
    int *cnts = bq_lds + S * nsample * BQ_LANES;
    int *final_img = cnts + S * BQ_LANES;

    const int scene = blockIdx.y;
    const int j0 = blockIdx.x * BQ_LANES;
    const int j = j0 + lane;
    const bool active = j < m;
    xyz += (size_t)scene * n * 3;
    const float *ctr = new_xyz + ((size_t)scene * m + (active ? j : 0)) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];

    int cnt = active ? 0 : nsample;  // inactive lanes never record
    const int kbeg = seg * seg_len;
    const int kend = (kbeg + seg_len < n) ? kbeg + seg_len : n;
    int k0 = kbeg;
    for (; k0 + BQ_BATCH <= kend; k0 += BQ_BATCH) {
        if (__all(cnt >= nsample)) break;
        float p[BQ_BATCH * 3];
        const float *src = xyz + (size_t)k0 * 3;
#pragma unroll
        for (int u = 0; u < BQ_BATCH * 3; ++u) p[u] = src[u];  // wave-uniform -> scalar loads
#pragma unroll
        for (int u = 0; u < BQ_BATCH; ++u) {
            const float d2 = sqdist(cx, cy, cz, p[u * 3], p[u * 3 + 1], p[u * 3 + 2]);
            bq_test<DILATED>(d2, r2max, r2min, hits, nsample, lane, k0 + u, cnt);
        }
    }
    if (!__all(cnt >= nsample)) {
        for (; k0 < kend; ++k0) {
            const float d2 = sqdist(cx, cy, cz, xyz[(size_t)k0 * 3], xyz[(size_t)k0 * 3 + 1], xyz[(size_t)k0 * 3 + 2]);
            bq_test<DILATED>(d2, r2max, r2min, hits, nsample, lane, k0, cnt);
        }
    }
    cnts[seg * BQ_LANES + lane] = active ? cnt : 0;
    __syncthreads();

    // ordered concatenation of the segments: this wave's hits start at the sum of the earlier counts
    int before = 0, total = 0;
    for (int s = 0; s < S; ++s) {
        const int c = cnts[s * BQ_LANES + lane];
        before += (s < seg) ? c : 0;
        total += c;
    }
    const int mine = active ? cnt : 0;
    for (int i = 0; i < mine && before + i < nsample; ++i)
        final_img[(before + i) * (BQ_LANES + 1) + lane] = hits[i * BQ_LANES + lane];
    __syncthreads();
    // rows with fewer than nsample hits are padded with their first hit (zeros when empty)
    if (seg == 0) {
        const int kept = total < nsample ? total : nsample;
        const int pad = kept > 0 ? final_img[lane] : 0;
        for (int i = kept; i < nsample; ++i) final_img[i * (BQ_LANES + 1) + lane] = pad;
        cnts[lane] = total;
    }
    __syncthreads();
    // coalesced write-out of the 64 x nsample block (rows of consecutive centroids are adjacent)
    const int rows = (m - j0 < BQ_LANES) ? m - j0 : BQ_LANES;
    int *dst = idx + ((size_t)scene * m + j0) * nsample;
    for (int e = threadIdx.x; e < rows * nsample; e += blockDim.x) {
        const int c = e / nsample, i = e - c * nsample;
        if (fill_empty || cnts[c] > 0) dst[e] = final_img[i * (BQ_LANES + 1) + c];
    }
}

int launch_ball_query(bool dilated, bool fill_empty, int b, int n, int m, float rmax, float rmin, int nsample,
                      const float *new_xyz, const float *xyz, int *idx, hipStream_t st) {
    if (b < 0 || n < 0 || m < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "ball_query: bad shape b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (b == 0 || m == 0 || nsample == 0) return SPS_OK;
    if (n == 0 && !fill_empty) return SPS_OK;
    if (!new_xyz || (!xyz && n > 0) || !idx) return fail(SPS_ERR_INVALID, "ball_query: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "ball_query: batch %d exceeds the grid limit", b);
    const int groups = divup(m, BQ_LANES);
    // enough segments to put ~4 waves on every SIMD of the chip, within the LDS budget (64 KiB)
    int S = divup(4096, b * groups);
    S = S < 1 ? 1 : (S > BQ_MAX_SEG ? BQ_MAX_SEG : S);
    while (S > 1 && n / S < 256) --S;
    auto lds_bytes = [&](int s) { return (size_t)4 * ((size_t)s * nsample * BQ_LANES + s * BQ_LANES + (size_t)nsample * (BQ_LANES + 1)); };
    while (S > 1 && lds_bytes(S) > 64 * 1024) --S;
    if (lds_bytes(S) > 64 * 1024) return fail(SPS_ERR_INVALID, "ball_query: nsample=%d needs more LDS than a workgroup has", nsample);
    int seg_len = divup(n > 0 ? n : 1, S);
    seg_len = divup(seg_len, BQ_BATCH) * BQ_BATCH;
    dim3 grid(groups, b), block(BQ_LANES * S);
    // radius*radius in fp32, as the reference kernel computes it (ball_query_gpu.cu:23, 84-85)
    const float r2max = rmax * rmax, r2min = rmin * rmin;
    if (dilated)
        hipLaunchKernelGGL(ball_query_seg_kernel<true>, grid, block, lds_bytes(S), st, n, m, seg_len, r2max, r2min,
                           nsample, fill_empty ? 1 : 0, new_xyz, xyz, idx);
    else
        hipLaunchKernelGGL(ball_query_seg_kernel<false>, grid, block, lds_bytes(S), st, n, m, seg_len, r2max, r2min,
                           nsample, fill_empty ? 1 : 0, new_xyz, xyz, idx);
    return check_launch("ball_query_seg_kernel");
}

}  // namespace sps

extern "C" int sps_ball_query_kernel_launcher_fast(int b, int n, int m, float radius, int nsample,
                                                   const float *new_xyz, const float *xyz, int *idx,
                                                   sps_stream_t stream) {
    return sps::launch_ball_query(false, false, b, n, m, radius, 0.f, nsample, new_xyz, xyz, idx,
                                  sps::as_stream(stream));
}

extern "C" int sps_ball_query_dilated_kernel_launcher_fast(int b, int n, int m, float max_radius,
                                                           float min_radius, int nsample,
                                                           const float *new_xyz, const float *xyz, int *idx,
                                                           sps_stream_t stream) {
    return sps::launch_ball_query(true, false, b, n, m, max_radius, min_radius, nsample, new_xyz, xyz, idx,
                                  sps::as_stream(stream));
}
