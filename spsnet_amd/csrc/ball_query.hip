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

#include <stdlib.h>
#include "spatial_grid.h"

namespace sps {

constexpr int BQ_LANES = 64;
constexpr int BQ_MAX_SEG = 8;        // segments (waves) per workgroup, general launches
constexpr int BQ_MAX_SEG_DUAL = 16;  // dual kernel: small launches (streamed chunks) use up to 16, LDS permitting
constexpr int BQ_BATCH = 8;  // points per scalar-load batch

template <typename HitT>
__device__ __forceinline__ void bq_append(HitT *hits, int nsample, int lane, int k, int &cnt) {
    if (cnt < nsample) {
        hits[cnt * BQ_LANES + lane] = (HitT)k;
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
    // Per batch of 8 points: 8 branch-free distance evaluations, ONE wave-uniform test "can any lane still
    // take any of them?" (min of the 8 distances against the lane's remaining threshold), and only then the
    // ordered per-point append path.  Hits are rare (a point lies in few of the 64 balls), and a lone wave pays
    // ~40 cycles per taken branch, so testing per point cost 3x more than the arithmetic.
    // Scalar loads return out of order (hipcc can only wait lgkmcnt(0)), so the next batch is requested right
    // AFTER the wait for the current one and BEFORE its arithmetic, which then hides the scalar-cache latency.
    int k0 = kbeg;
    float p[BQ_BATCH * 3], pn[BQ_BATCH * 3];
    if (k0 + BQ_BATCH <= kend) {
#pragma unroll
        for (int u = 0; u < BQ_BATCH * 3; ++u) p[u] = xyz[(size_t)k0 * 3 + u];
    }
    float thr = (cnt < nsample) ? r2max : -1.f;  // full / inactive lanes can no longer be hit
    for (; k0 + BQ_BATCH <= kend; k0 += BQ_BATCH) {
        if (__all(thr < 0.f)) break;
        asm volatile("" ::"s"(p[0]), "s"(p[BQ_BATCH * 3 - 1]));  // the current batch has landed
        __builtin_amdgcn_sched_barrier(0);
        const bool more = k0 + 2 * BQ_BATCH <= kend;
        const float *nsrc = xyz + (size_t)(more ? k0 + BQ_BATCH : k0) * 3;
#pragma unroll
        for (int u = 0; u < BQ_BATCH * 3; ++u) pn[u] = nsrc[u];
        __builtin_amdgcn_sched_barrier(0);
        float d2[BQ_BATCH];
#pragma unroll
        for (int u = 0; u < BQ_BATCH; ++u) d2[u] = sqdist(cx, cy, cz, p[u * 3], p[u * 3 + 1], p[u * 3 + 2]);
        float dmin = d2[0];
#pragma unroll
        for (int u = 1; u < BQ_BATCH; ++u) dmin = fminf(dmin, d2[u]);
        const bool maybe = DILATED ? (dmin < thr || (dmin == 0.f && thr >= 0.f)) : (dmin < thr);
        if (__any(maybe)) {
#pragma unroll
            for (int u = 0; u < BQ_BATCH; ++u) bq_test<DILATED>(d2[u], r2max, r2min, hits, nsample, lane, k0 + u, cnt);
            thr = (cnt < nsample) ? r2max : -1.f;
        }
#pragma unroll
        for (int u = 0; u < BQ_BATCH * 3; ++u) p[u] = pn[u];
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

// Two radii in ONE scan (the two grouping scales of an SA layer share centroids and points): the distance
// is evaluated once per pair and tested against both radii; each radius keeps its own ordered hit list.
// LDS: cnt[2][S][64] | final[max(nsA,nsB)][65] (ints) | hitsA[S][nsA][64] | hitsB[S][nsB][64] (HitT).  The hit lists
// hold point indices RELATIVE to the segment start, as 16-bit values whenever a segment is shorter than 65 536
// points: half the LDS, so twice the segments (waves) per workgroup fit -- the kernel is latency-bound per wave.
// Rows are always fully written (zeros for empty balls), like sps_ball_query_full.
template <typename HitT>
__global__ __launch_bounds__(BQ_LANES *BQ_MAX_SEG_DUAL) void ball_query_dual_kernel(
    int n, int m, int seg_len, float r2a, float r2b, int nsa, int nsb, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx_a, int *__restrict__ idx_b, const int *__restrict__ perm,
    int jbeg, int jend, const int *__restrict__ run_if) {
    if (run_if && *run_if == 0) return;   // predicated launch (sps_ball_query_full2_range): workgroup-uniform
    extern __shared__ __attribute__((aligned(16))) int bq_lds[];
    const int S = blockDim.x / BQ_LANES;
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *cnts = bq_lds;  // [2][S][64]
    int *final_img = cnts + 2 * S * BQ_LANES;
    HitT *hits_base = (HitT *)(final_img + (nsa > nsb ? nsa : nsb) * (BQ_LANES + 1) + 1);  // +1: keeps 8-byte alignment irrelevant, HitT <= 4 B
    HitT *hits_a = hits_base + seg * nsa * BQ_LANES;
    HitT *hits_b = hits_base + S * nsa * BQ_LANES + seg * nsb * BQ_LANES;

    const int scene = blockIdx.y;
    const int j0 = jbeg + blockIdx.x * BQ_LANES;  // centroid range [jbeg, jend) of every scene
    const bool active = j0 + lane < jend;
    // With `perm` (centroid_order_kernel) the 64 lanes take 64 spatially neighbouring centroids instead of 64
    // consecutive ones: far fewer 8-point batches contain a hit for ANY lane, and neighbours fill up together.
    const int j = active ? (perm ? perm[(size_t)scene * m + j0 + lane] : j0 + lane) : 0;
    xyz += (size_t)scene * n * 3;
    const float *ctr = new_xyz + ((size_t)scene * m + j) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];

    int ca = active ? 0 : nsa, cb = active ? 0 : nsb;
    const int kbeg = seg * seg_len;
    const int kend = (kbeg + seg_len < n) ? kbeg + seg_len : n;
    int k0 = kbeg;
    // largest squared radius this lane can still use (-1 when both lists are full / the lane is inactive)
    auto lane_thr = [&]() { return fmaxf(ca < nsa ? r2a : -1.f, cb < nsb ? r2b : -1.f); };
    float thr = lane_thr();
    // one batch of BQ_BATCH points (wave-uniform coordinates q[]) starting at index kb
    auto batch = [&](int kb, const float *q) {
        float d2[BQ_BATCH];
#pragma unroll
        for (int u = 0; u < BQ_BATCH; ++u) d2[u] = sqdist(cx, cy, cz, q[u * 3], q[u * 3 + 1], q[u * 3 + 2]);
        float dmin = d2[0];
#pragma unroll
        for (int u = 1; u < BQ_BATCH; ++u) dmin = fminf(dmin, d2[u]);
        if (__any(dmin < thr)) {
            // branch-free per-lane hit bitmasks of the 8 points, then each lane walks only its own set bits
            // (ascending point index, so the ordered-fill rule holds); most lanes have none, the rest one or two
            unsigned bits_a = 0u, bits_b = 0u;
#pragma unroll
            for (int u = 0; u < BQ_BATCH; ++u) {
                bits_a |= (d2[u] < r2a ? 1u : 0u) << u;
                bits_b |= (d2[u] < r2b ? 1u : 0u) << u;
            }
            while (bits_a != 0u && ca < nsa) {
                const int u = __builtin_ctz(bits_a);
                bits_a &= bits_a - 1u;
                hits_a[ca * BQ_LANES + lane] = (HitT)(kb - kbeg + u);
                ++ca;
            }
            while (bits_b != 0u && cb < nsb) {
                const int u = __builtin_ctz(bits_b);
                bits_b &= bits_b - 1u;
                hits_b[cb * BQ_LANES + lane] = (HitT)(kb - kbeg + u);
                ++cb;
            }
            thr = lane_thr();
        }
    };
    {
        float p[BQ_BATCH * 3], pn[BQ_BATCH * 3];  // current / next scalar-load batch (see ball_query_seg_kernel)
        if (k0 + BQ_BATCH <= kend) {
#pragma unroll
            for (int u = 0; u < BQ_BATCH * 3; ++u) p[u] = xyz[(size_t)k0 * 3 + u];
        }
        for (; k0 + BQ_BATCH <= kend; k0 += BQ_BATCH) {
            if (__all(thr < 0.f)) break;
            asm volatile("" ::"s"(p[0]), "s"(p[BQ_BATCH * 3 - 1]));
            __builtin_amdgcn_sched_barrier(0);
            const bool more = k0 + 2 * BQ_BATCH <= kend;
            const float *nsrc = xyz + (size_t)(more ? k0 + BQ_BATCH : k0) * 3;
#pragma unroll
            for (int u = 0; u < BQ_BATCH * 3; ++u) pn[u] = nsrc[u];
            __builtin_amdgcn_sched_barrier(0);
            batch(k0, p);
#pragma unroll
            for (int u = 0; u < BQ_BATCH * 3; ++u) p[u] = pn[u];
        }
    }
    if (!__all(ca >= nsa && cb >= nsb)) {
        for (; k0 < kend; ++k0) {
            const float d2 = sqdist(cx, cy, cz, xyz[(size_t)k0 * 3], xyz[(size_t)k0 * 3 + 1], xyz[(size_t)k0 * 3 + 2]);
            if (d2 < r2a) bq_append(hits_a, nsa, lane, k0 - kbeg, ca);
            if (d2 < r2b) bq_append(hits_b, nsb, lane, k0 - kbeg, cb);
        }
    }
    cnts[seg * BQ_LANES + lane] = active ? ca : 0;
    cnts[(S + seg) * BQ_LANES + lane] = active ? cb : 0;
    __syncthreads();
    const int rows = (jend - j0 < BQ_LANES) ? jend - j0 : BQ_LANES;
    // merge + write-out, one radius after the other through the shared `final` image
    for (int which = 0; which < 2; ++which) {
        const int ns = which ? nsb : nsa;
        const HitT *hits = which ? hits_b : hits_a;
        const int *cn = cnts + which * S * BQ_LANES;
        int before = 0, total = 0;
        for (int s = 0; s < S; ++s) {
            const int c = cn[s * BQ_LANES + lane];
            before += (s < seg) ? c : 0;
            total += c;
        }
        const int mine = cn[seg * BQ_LANES + lane];
        for (int i = 0; i < mine && before + i < ns; ++i)
            final_img[(before + i) * (BQ_LANES + 1) + lane] = kbeg + (int)hits[i * BQ_LANES + lane];
        __syncthreads();
        if (seg == 0) {
            const int kept = total < ns ? total : ns;
            const int pad = kept > 0 ? final_img[lane] : 0;
            for (int i = kept; i < ns; ++i) final_img[i * (BQ_LANES + 1) + lane] = pad;
        }
        __syncthreads();
        int *out = (which ? idx_b : idx_a) + (size_t)scene * m * ns;
        for (int e = threadIdx.x; e < rows * ns; e += blockDim.x) {
            const int c = e / ns, i = e - c * ns;
            const int row = perm ? perm[(size_t)scene * m + j0 + c] : j0 + c;  // rows stay ns*4-byte contiguous
            out[(size_t)row * ns + i] = final_img[i * (BQ_LANES + 1) + c];
        }
        __syncthreads();
    }
}

// ---- wave-per-centroid variant ---------------------------------------------------------------------------
// One WAVE per centroid, one LANE per scanned point (64 points per step, coalesced 12-byte reads served by
// L1/L2): hits are ordered with ballot + prefix popcount, and the scan of a centroid stops as soon as ITS lists
// are full.  In lidar clouds most balls fill after a fraction of the cloud, which the lane-per-centroid kernel
// cannot exploit (it stops only when all 64 centroids of a wave are full) and whose per-hit divergent appends
// cost more than its distance arithmetic (tools/bq_time.py: 353 us vs 150 us without hits at the layer-0 shape).
constexpr int BQW_WAVES = 4;   // centroids per workgroup
constexpr long long BQ_WAVE_MAX_CENTROIDS = 8192;
constexpr long long BQ_SEG_MAX_CENTROIDS = 2048;   // ... and up to which four waves share a centroid  // b * centroids up to which sps_ball_query_full2_range prefers this kernel
constexpr int BQW_UNROLL = 4;  // 64-point steps in flight per loop trip

__global__ __launch_bounds__(64 * BQW_WAVES) void ball_query_wave_dual_kernel(
    int n, int m, float r2a, float r2b, int nsa, int nsb, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx_a, int *__restrict__ idx_b, int jbeg, int jend,
    const int *__restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    const int scene = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = jbeg + blockIdx.x * BQW_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (j >= jend) return;
    xyz += (size_t)scene * n * 3;
    const float *ctr = new_xyz + ((size_t)scene * m + j) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
    int *row_a = idx_a + ((size_t)scene * m + j) * nsa;
    int *row_b = idx_b + ((size_t)scene * m + j) * nsb;
    const unsigned long long below = (1ull << lane) - 1ull;  // lanes (= points) before mine in this step

    int ca = 0, cb = 0, first_a = 0, first_b = 0;  // wave-uniform
    for (int base = 0; base < n; base += 64 * BQW_UNROLL) {
        float px[BQW_UNROLL], py[BQW_UNROLL], pz[BQW_UNROLL];
#pragma unroll
        for (int u = 0; u < BQW_UNROLL; ++u) {
            const int k = base + u * 64 + lane;
            const int kk = k < n ? k : n - 1;
            px[u] = xyz[kk * 3 + 0]; py[u] = xyz[kk * 3 + 1]; pz[u] = xyz[kk * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < BQW_UNROLL; ++u) {
            const int k = base + u * 64 + lane;
            const float d2 = sqdist(cx, cy, cz, px[u], py[u], pz[u]);
            const bool in = k < n;
            const bool ha = in && d2 < r2a, hb = in && d2 < r2b;
            const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
            if (ma != 0ull && ca < nsa) {
                if (ca == 0) first_a = base + u * 64 + __builtin_ctzll(ma);
                const int pos = ca + __builtin_popcountll(ma & below);
                if (ha && pos < nsa) row_a[pos] = k;
                ca += __builtin_popcountll(ma);
            }
            if (mb != 0ull && cb < nsb) {
                if (cb == 0) first_b = base + u * 64 + __builtin_ctzll(mb);
                const int pos = cb + __builtin_popcountll(mb & below);
                if (hb && pos < nsb) row_b[pos] = k;
                cb += __builtin_popcountll(mb);
            }
        }
        if (ca >= nsa && cb >= nsb) break;
    }
    // pad with the first hit (zeros for an empty ball: first_* is still 0)
    const int ka = ca < nsa ? ca : nsa, kb = cb < nsb ? cb : nsb;
    for (int l = ka + lane; l < nsa; l += 64) row_a[l] = first_a;
    for (int l = kb + lane; l < nsb; l += 64) row_b[l] = first_b;
}

// The same with the four waves of a workgroup sharing ONE centroid (each scans a quarter of the cloud, ordered hit
// lists meet in LDS): for launches of a few hundred centroids, where the scan of a whole cloud by one wave (~38 us at
// 16 384 points) is the launch time.
constexpr int BQS_SEG = 4;
constexpr int BQS_MAX_NS = 64;
__global__ __launch_bounds__(64 * BQS_SEG) void ball_query_wave_seg_kernel(
    int n, int m, float r2a, float r2b, int nsa, int nsb, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx_a, int *__restrict__ idx_b, int jbeg,
    const int *__restrict__ run_if) {
    __shared__ int hits[2][BQS_SEG][BQS_MAX_NS];
    __shared__ int cnt[2][BQS_SEG];
    if (run_if && *run_if == 0) return;
    const int scene = blockIdx.y, j = jbeg + blockIdx.x;
    const int lane = threadIdx.x & 63, seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    xyz += (size_t)scene * n * 3;
    const float *ctr = new_xyz + ((size_t)scene * m + j) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
    const unsigned long long below = (1ull << lane) - 1ull;
    const int seg_len = ((n + BQS_SEG - 1) / BQS_SEG + 63) & ~63;
    const int kbeg = seg * seg_len, kend = (kbeg + seg_len < n) ? kbeg + seg_len : n;
    int ca = 0, cb = 0;
    for (int base = kbeg; base < kend; base += 64 * BQW_UNROLL) {
        float px[BQW_UNROLL], py[BQW_UNROLL], pz[BQW_UNROLL];
#pragma unroll
        for (int u = 0; u < BQW_UNROLL; ++u) {
            const int k = base + u * 64 + lane;
            const int kk = k < n ? k : n - 1;
            px[u] = xyz[kk * 3 + 0]; py[u] = xyz[kk * 3 + 1]; pz[u] = xyz[kk * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < BQW_UNROLL; ++u) {
            const int k = base + u * 64 + lane;
            const float d2 = sqdist(cx, cy, cz, px[u], py[u], pz[u]);
            const bool in = k < kend;
            const bool ha = in && d2 < r2a, hb = in && d2 < r2b;
            const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
            if (ma != 0ull && ca < nsa) {
                const int pos = ca + __builtin_popcountll(ma & below);
                if (ha && pos < nsa) hits[0][seg][pos] = k;
                ca += __builtin_popcountll(ma);
            }
            if (mb != 0ull && cb < nsb) {
                const int pos = cb + __builtin_popcountll(mb & below);
                if (hb && pos < nsb) hits[1][seg][pos] = k;
                cb += __builtin_popcountll(mb);
            }
        }
        if (ca >= nsa && cb >= nsb) break;
    }
    if (lane == 0) { cnt[0][seg] = ca < nsa ? ca : nsa; cnt[1][seg] = cb < nsb ? cb : nsb; }
    __syncthreads();
    if (seg < 2) {  // wave 0 merges radius a, wave 1 radius b: segment order = index order
        const int ns = seg ? nsb : nsa;
        int *row = (seg ? idx_b : idx_a) + ((size_t)scene * m + j) * ns;
        int c[BQS_SEG], total = 0;
#pragma unroll
        for (int s2 = 0; s2 < BQS_SEG; ++s2) { c[s2] = cnt[seg][s2]; total += c[s2]; }
        int first = 0;  // empty ball: zeros
#pragma unroll
        for (int s2 = BQS_SEG - 1; s2 >= 0; --s2) first = c[s2] > 0 ? hits[seg][s2][0] : first;
        for (int p = lane; p < ns; p += 64) {
            int v = first, q = p;
#pragma unroll
            for (int s2 = 0; s2 < BQS_SEG; ++s2) {
                if (q >= 0 && q < c[s2]) v = hits[seg][s2][q];
                q = (q >= 0 && q < c[s2]) ? -1 : q - c[s2];
            }
            row[p] = p < total ? v : first;
        }
    }
}

// Several centroids per wave.  The per-centroid kernels above re-read the cloud once per centroid: 2048 centroids x 16 384
// points (the last chunk of a streamed layer) or 8192 x 4096 (IA-SSD layer 1) pull ~400 MB through L2 per launch -- at 27-30 us
// that IS the L2 bandwidth.  Here a wave takes C consecutive centroids of a scene (centres in SGPRs) and tests every 64-point
// step it loads against all of them; SEG waves of a workgroup split the point range, their ordered hit lists meet in LDS and
// are concatenated in segment order (= index order), as in ball_query_wave_seg_kernel.  Same rows, C times less traffic.
template <int C, int SEG, int UNR>
__global__ __launch_bounds__(64 * SEG) void ball_query_wave_multi_kernel(
    int n, int m, float r2a, float r2b, int nsa, int nsb, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx_a, int *__restrict__ idx_b, int jbeg,
    const int *__restrict__ run_if, const int *__restrict__ alt, const int *__restrict__ gather_idx,
    int kfirst = 0, int kcount = -1, const int *__restrict__ all_points_if = nullptr,
    const int *__restrict__ all_points_if_any = nullptr, int any_count = 0) {
    // [kfirst, kfirst + kcount): only these points of every scene are scanned (kcount < 0: all n) -- the rows then hold the
    // first nsample hits AMONG THEM in index order, and a row without a hit is filled with -1 (a later range may still hit;
    // zeros would read as "point 0"); *all_points_if != 0 widens the scan to the whole scene again (rows as always)
    // (sps_ball_query_full2_points: a layer whose cloud arrives piecewise queries the early part first and the rest later).
    // gather_idx != NULL: the centroids are xyz[gather_idx[scene][j]] (the sampler's picks, clamped into the cloud) and this
    // launch ALSO writes them to new_xyz -- the gather_operation between sampler and query (pointnet2_modules.py:423-424)
    // fused in, one launch less on the critical chain of a layer
    __shared__ int hits[2][C][SEG][BQS_MAX_NS];
    __shared__ int cnt[2][C][SEG];
    if (run_if && *run_if == 0) return;
    // *alt != 0: all m centroids of every scene instead of [jbeg, jbeg + C gridDim.x) (self-repairing range launch)
    const bool whole = alt && *alt != 0;
    const int jfirst = whole ? 0 : jbeg, groups = whole ? m / C : (int)gridDim.x;
    const int scene = blockIdx.y;
    const int lane = threadIdx.x & 63, seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    xyz += (size_t)scene * n * 3;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int j0 = jfirst + grp * C;
    __syncthreads();   // the previous group's hit lists are dead
    float cx[C], cy[C], cz[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float *ctr = new_xyz + ((size_t)scene * m + j0 + c) * 3;
        if (gather_idx) {
            int src = __builtin_amdgcn_readfirstlane(gather_idx[(size_t)scene * m + j0 + c]);
            src = src < 0 ? 0 : (src >= n ? n - 1 : src);
            ctr = xyz + (size_t)src * 3;
        }
        cx[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ctr[0])));
        cy[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ctr[1])));
        cz[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ctr[2])));
        if (gather_idx && seg == 0 && lane == 0) {
            float *dst = const_cast<float *>(new_xyz) + ((size_t)scene * m + j0 + c) * 3;
            dst[0] = cx[c]; dst[1] = cy[c]; dst[2] = cz[c];
        }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool ranged = kcount >= 0 && !flag_or_any(all_points_if, all_points_if_any, any_count);
    const int k_lo = ranged ? kfirst : 0, k_hi = ranged ? kfirst + kcount : n;
    const int seg_len = ((k_hi - k_lo + SEG - 1) / SEG + 63) & ~63;
    const int kbeg = k_lo + seg * seg_len, kend = (kbeg + seg_len < k_hi) ? kbeg + seg_len : k_hi;
    int ca[C], cb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ca[c] = cb[c] = 0;
    // UNR 64-point steps are in flight per trip: the scan is a chain of L2 round trips, so their number sets the launch time
    for (int base = kbeg; base < kend; base += 64 * UNR) {
        float px[UNR], py[UNR], pz[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = base + u * 64 + lane;
            const int kk = k < n ? k : n - 1;
            px[u] = xyz[kk * 3 + 0]; py[u] = xyz[kk * 3 + 1]; pz[u] = xyz[kk * 3 + 2];
        }
        bool all_full = true;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = base + u * 64 + lane;
            const bool in = k < kend;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d2 = sqdist(cx[c], cy[c], cz[c], px[u], py[u], pz[u]);
                const bool ha = in && d2 < r2a, hb = in && d2 < r2b;
                const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
                if (ma != 0ull && ca[c] < nsa) {
                    const int pos = ca[c] + __builtin_popcountll(ma & below);
                    if (ha && pos < nsa) hits[0][c][seg][pos] = k;
                    ca[c] += __builtin_popcountll(ma);
                }
                if (mb != 0ull && cb[c] < nsb) {
                    const int pos = cb[c] + __builtin_popcountll(mb & below);
                    if (hb && pos < nsb) hits[1][c][seg][pos] = k;
                    cb[c] += __builtin_popcountll(mb);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) all_full = all_full && ca[c] >= nsa && cb[c] >= nsb;
        if (all_full) break;
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) { cnt[0][c][seg] = ca[c] < nsa ? ca[c] : nsa; cnt[1][c][seg] = cb[c] < nsb ? cb[c] : nsb; }
    }
    __syncthreads();
    for (int pair = seg; pair < 2 * C; pair += SEG) {   // (radius, centroid) rows, concatenated in segment = index order
        const int which = pair & 1, c = pair >> 1;
        const int ns = which ? nsb : nsa;
        int *row = (which ? idx_b : idx_a) + ((size_t)scene * m + j0 + c) * ns;
        int cs[SEG], total = 0;
#pragma unroll
        for (int s2 = 0; s2 < SEG; ++s2) { cs[s2] = cnt[which][c][s2]; total += cs[s2]; }
        int first = ranged ? -1 : 0;  // empty ball: zeros like the reference -- over a point RANGE: -1s ("no hit here", not "point 0")
#pragma unroll
        for (int s2 = SEG - 1; s2 >= 0; --s2) first = cs[s2] > 0 ? hits[which][c][s2][0] : first;
        for (int p = lane; p < ns; p += 64) {
            int v = first, q = p;
#pragma unroll
            for (int s2 = 0; s2 < SEG; ++s2) {
                if (q >= 0 && q < cs[s2]) v = hits[which][c][s2][q];
                q = (q >= 0 && q < cs[s2]) ? -1 : q - cs[s2];
            }
            row[p] = p < total ? v : first;
        }
    }
    }   // group loop
}

// perm[b, :] = the scene's centroids sorted by 12-bit cell key (counting sort in LDS, one workgroup per scene;
// order inside a cell is arbitrary -- it only affects speed, never results).
constexpr int CO_THREADS = 512;
__global__ __launch_bounds__(CO_THREADS) void centroid_order_kernel(int m, const float *__restrict__ new_xyz,
                                                                    int *__restrict__ perm) {
    __shared__ int hist[PF_BINS];
    __shared__ float red[6][CO_THREADS / 64];
    __shared__ int wsum[CO_THREADS / 64];
    const int scene = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *c = new_xyz + (size_t)scene * m * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = tid; k < m; k += CO_THREADS)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], c[k * 3 + a]); hi[a] = fmaxf(hi[a], c[k * 3 + a]); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    for (int i = tid; i < PF_BINS; i += CO_THREADS) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = red[a][0]; hi[a] = red[3 + a][0];
        for (int w = 1; w < CO_THREADS / 64; ++w) { lo[a] = fminf(lo[a], red[a][w]); hi[a] = fmaxf(hi[a], red[3 + a][w]); }
    }
    const PfGrid grid = pf_make_grid(lo, hi);
    for (int k = tid; k < m; k += CO_THREADS) atomicAdd(&hist[pf_cell_key(grid, c[k * 3], c[k * 3 + 1], c[k * 3 + 2])], 1);
    __syncthreads();
    {
        constexpr int PER = PF_BINS / CO_THREADS;
        int loc[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { loc[i] = hist[tid * PER + i]; sum += loc[i]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int run = base + incl - sum;
#pragma unroll
        for (int i = 0; i < PER; ++i) { hist[tid * PER + i] = run; run += loc[i]; }
    }
    __syncthreads();
    for (int k = tid; k < m; k += CO_THREADS) {
        const int pos = atomicAdd(&hist[pf_cell_key(grid, c[k * 3], c[k * 3 + 1], c[k * 3 + 2])], 1);
        perm[(size_t)scene * m + pos] = k;
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

extern "C" int sps_ball_query_full2(int b, int n, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                    const float *new_xyz, const float *xyz, int *idx_a, int *idx_b, int *perm_work,
                                    sps_stream_t stream) {
    return sps_ball_query_full2_range(b, n, m, 0, m, radius_a, nsample_a, radius_b, nsample_b, new_xyz, xyz, idx_a, idx_b,
                                      perm_work, nullptr, nullptr, nullptr, stream);
}

extern "C" int sps_ball_query_full2_range(int b, int n, int m, int j0, int jcount, float radius_a, int nsample_a,
                                          float radius_b, int nsample_b, const float *new_xyz, const float *xyz,
                                          int *idx_a, int *idx_b, int *perm_work, const int *run_if, const int *full_range_if,
                                          const int *gather_idx, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0 || m < 0 || nsample_a <= 0 || nsample_b <= 0 || j0 < 0 || jcount < 0 || j0 + jcount > m)
        return fail(SPS_ERR_INVALID, "ball_query_full2: bad shape b=%d n=%d m=%d ns=(%d,%d) range [%d,+%d)", b, n, m,
                    nsample_a, nsample_b, j0, jcount);
    if (perm_work && !(j0 == 0 && jcount == m)) return fail(SPS_ERR_INVALID, "ball_query_full2: perm_work needs the full range");
    if (b == 0 || jcount == 0) return SPS_OK;
    if (!new_xyz || (!xyz && n > 0) || !idx_a || !idx_b) return fail(SPS_ERR_INVALID, "ball_query_full2: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "ball_query_full2: batch %d exceeds the grid limit", b);
    if (n == 0) return fail(SPS_ERR_INVALID, "ball_query_full2: n == 0");
    // Few centroids: the lane-per-centroid kernel below would put a 64-centroid group's whole scan on ONE CU (its
    // segments are waves of one workgroup); one wave per centroid spreads the same work over the chip.
    bool per_wave = !perm_work && (long long)b * jcount <= BQ_WAVE_MAX_CENTROIDS;
    if (const char *force = getenv("SPS_BQ_WAVE")) per_wave = force[0] == '1';  // diagnostic override (tools/bq_time.py)
    static const bool multi_ok = !(getenv("SPS_BQ_MULTI") && getenv("SPS_BQ_MULTI")[0] == '0');
    if (full_range_if && !(jcount % 4 == 0 && m % 4 == 0 && (long long)b * jcount <= BQ_WAVE_MAX_CENTROIDS && !perm_work))
        return fail(SPS_ERR_INVALID, "ball_query_full2: full_range_if needs a per-wave launch with jcount and m multiples of 4");
    if (gather_idx && !(jcount % 4 == 0 && (long long)b * jcount <= BQ_WAVE_MAX_CENTROIDS && !perm_work && n >= 256 &&
                        nsample_a <= BQS_MAX_NS && nsample_b <= BQS_MAX_NS))
        return fail(SPS_ERR_INVALID, "ball_query_full2: gather_idx needs a per-wave launch (range of at most %d centroids, a multiple of 4)",
                    BQ_WAVE_MAX_CENTROIDS);
    if (per_wave && (multi_ok || full_range_if || gather_idx) && jcount % 4 == 0 && nsample_a <= BQS_MAX_NS && nsample_b <= BQS_MAX_NS && n >= 256) {
        // four centroids per wave share every point load; the point range is split so that the launch still has waves
        const float ra2 = radius_a * radius_a, rb2 = radius_b * radius_b;
        const dim3 grid(jcount / 4, b);
        if (n >= 8192)
            hipLaunchKernelGGL((ball_query_wave_multi_kernel<4, 8, 8>), grid, dim3(64 * 8), 0, as_stream(stream), n, m, ra2, rb2,
                               nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, j0, run_if, full_range_if, gather_idx);
        else if (n >= 2048)
            hipLaunchKernelGGL((ball_query_wave_multi_kernel<4, 4, 8>), grid, dim3(64 * 4), 0, as_stream(stream), n, m, ra2, rb2,
                               nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, j0, run_if, full_range_if, gather_idx);
        else
            hipLaunchKernelGGL((ball_query_wave_multi_kernel<4, 4, 4>), grid, dim3(64 * 4), 0, as_stream(stream), n, m, ra2, rb2,
                               nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, j0, run_if, full_range_if, gather_idx);
        return check_launch("ball_query_wave_multi_kernel");
    }
    if (per_wave && (long long)b * jcount <= BQ_SEG_MAX_CENTROIDS && nsample_a <= BQS_MAX_NS && nsample_b <= BQS_MAX_NS &&
        n >= 4096) {
        hipLaunchKernelGGL(ball_query_wave_seg_kernel, dim3(jcount, b), dim3(64 * BQS_SEG), 0, as_stream(stream), n, m,
                           radius_a * radius_a, radius_b * radius_b, nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, j0, run_if);
        return check_launch("ball_query_wave_seg_kernel");
    }
    if (per_wave) {
        hipLaunchKernelGGL(ball_query_wave_dual_kernel, dim3(divup(jcount, BQW_WAVES), b), dim3(64 * BQW_WAVES), 0,
                           as_stream(stream), n, m, radius_a * radius_a, radius_b * radius_b, nsample_a, nsample_b, new_xyz,
                           xyz, idx_a, idx_b, j0, j0 + jcount, run_if);
        return check_launch("ball_query_wave_dual_kernel");
    }
    const int groups = divup(jcount, BQ_LANES);
    int S = divup(4096, b * groups);
    S = S < 1 ? 1 : (S > BQ_MAX_SEG_DUAL ? BQ_MAX_SEG_DUAL : S);
    while (S > 1 && n / S < 256) --S;
    const int nsmax = nsample_a > nsample_b ? nsample_a : nsample_b;
    auto seg_of = [&](int s) { return divup(divup(n, s), BQ_BATCH) * BQ_BATCH; };
    auto lds_bytes = [&](int s) {
        const size_t hit = seg_of(s) <= 65536 ? 2 : 4;
        return hit * (size_t)s * (nsample_a + nsample_b) * BQ_LANES +
               (size_t)4 * (2 * s * BQ_LANES + (size_t)nsmax * (BQ_LANES + 1) + 1);
    };
    // 64 KiB keeps two workgroups on a CU; a launch with fewer workgroups than CUs may take up to 150 KiB each
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    const size_t lds_cap = (b * groups <= num_cu) ? 150 * 1024 : 64 * 1024;
    while (S > 1 && lds_bytes(S) > lds_cap) --S;
    if (lds_bytes(S) > lds_cap) return fail(SPS_ERR_INVALID, "ball_query_full2: nsample too large for LDS");
    const int seg_len = seg_of(S);
    const int *perm = nullptr;
    if (perm_work && m >= 4 * BQ_LANES) {  // worth a sort only when a scene has several groups of centroids
        hipLaunchKernelGGL(centroid_order_kernel, dim3(b), dim3(CO_THREADS), 0, as_stream(stream), m, new_xyz, perm_work);
        const int rc = check_launch("centroid_order_kernel");
        if (rc != SPS_OK) return rc;
        perm = perm_work;
    }
    typedef void (*dual_fn)(int, int, int, float, float, int, int, const float *, const float *, int *, int *, const int *, int, int,
                            const int *);
    const dual_fn fn = seg_len <= 65536 ? ball_query_dual_kernel<unsigned short> : ball_query_dual_kernel<int>;
    if (lds_bytes(S) > 64 * 1024) {
        static LdsLimitOnce raised[2];
        const int rc = raise_lds_limit((const void *)fn, 150 * 1024, raised[seg_len <= 65536 ? 0 : 1], "ball_query_full2");
        if (rc != SPS_OK) return rc;
    }
    hipLaunchKernelGGL(fn, dim3(groups, b), dim3(BQ_LANES * S), lds_bytes(S), as_stream(stream), n, m, seg_len,
                       radius_a * radius_a, radius_b * radius_b, nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, perm, j0,
                       j0 + jcount, run_if);
    return check_launch("ball_query_dual_kernel");
}

// Both radii for ALL m centroids of every scene, but only over the points [k0, k0 + kcount) of every scene: row = the first
// nsample hits among THOSE points in index order, padded with the first of them; -1 in every slot when there is none.  When
// *all_points_if != 0 or any of all_points_if_any[0 .. any_count) != 0 (device ints, may be NULL) the launch scans the whole
// scene instead -- the self-repair of a caller that queried a partly written cloud (spsnet_amd/sa_stack.py: the next layer's
// queries start while this layer's FPS still runs).  b * m <= 8192 centroids, m a multiple of 4, n >= 256, nsample <= 64.
extern "C" int sps_ball_query_full2_points(int b, int n, int m, int k0, int kcount, float radius_a, int nsample_a, float radius_b,
                                           int nsample_b, const float *new_xyz, const float *xyz, int *idx_a, int *idx_b,
                                           const int *all_points_if, const int *all_points_if_any, int any_count,
                                           sps_stream_t stream) {
    return sps_ball_query_full2_points_gather(b, n, m, k0, kcount, radius_a, nsample_a, radius_b, nsample_b, new_xyz, xyz, idx_a,
                                              idx_b, all_points_if, all_points_if_any, any_count, nullptr, stream);
}

// gather_idx (device i32 (b, m), may be NULL): the centroids are xyz[gather_idx[scene][j]] and the launch also WRITES them to
// new_xyz (as in sps_ball_query_full2_range) -- the last stage of a layer whose centroids were guessed takes the sampler's
// verified picks here instead of a gather launch in front of it.
extern "C" int sps_ball_query_full2_points_gather(int b, int n, int m, int k0, int kcount, float radius_a, int nsample_a,
                                                  float radius_b, int nsample_b, const float *new_xyz, const float *xyz, int *idx_a,
                                                  int *idx_b, const int *all_points_if, const int *all_points_if_any, int any_count,
                                                  const int *gather_idx, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 256 || m < 0 || (m % 4) || nsample_a <= 0 || nsample_b <= 0 || nsample_a > BQS_MAX_NS || nsample_b > BQS_MAX_NS ||
        k0 < 0 || kcount < 0 || k0 + kcount > n || (long long)b * m > BQ_WAVE_MAX_CENTROIDS || b > 65535)
        return fail(SPS_ERR_INVALID, "ball_query_full2_points: bad shape b=%d n=%d m=%d ns=(%d,%d) points [%d,+%d)", b, n, m,
                    nsample_a, nsample_b, k0, kcount);
    if (b == 0 || m == 0) return SPS_OK;
    if (!new_xyz || !xyz || !idx_a || !idx_b) return fail(SPS_ERR_INVALID, "ball_query_full2_points: null pointer");
    const float ra2 = radius_a * radius_a, rb2 = radius_b * radius_b;
    const dim3 grid(m / 4, b);
    // (the segment count follows the range that is normally scanned; the widened scan of a repair is correct with any)
    if (kcount >= 2048)
        hipLaunchKernelGGL((ball_query_wave_multi_kernel<4, 4, 8>), grid, dim3(64 * 4), 0, as_stream(stream), n, m, ra2, rb2,
                           nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, 0, (const int *)nullptr, (const int *)nullptr,
                           gather_idx, k0, kcount, all_points_if, all_points_if_any, any_count);
    else
        hipLaunchKernelGGL((ball_query_wave_multi_kernel<4, 4, 4>), grid, dim3(64 * 4), 0, as_stream(stream), n, m, ra2, rb2,
                           nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, 0, (const int *)nullptr, (const int *)nullptr,
                           gather_idx, k0, kcount, all_points_if, all_points_if_any, any_count);
    return check_launch("ball_query_wave_multi_kernel<points>");
}

// wave-per-centroid variant of sps_ball_query_full2 (same result; faster when balls fill early, i.e. large radii)
extern "C" int sps_ball_query_full2_wave(int b, int n, int m, float radius_a, int nsample_a, float radius_b,
                                         int nsample_b, const float *new_xyz, const float *xyz, int *idx_a, int *idx_b,
                                         sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || nsample_a <= 0 || nsample_b <= 0)
        return fail(SPS_ERR_INVALID, "ball_query_full2_wave: bad shape b=%d n=%d m=%d ns=(%d,%d)", b, n, m, nsample_a, nsample_b);
    if (b == 0 || m == 0) return SPS_OK;
    if (!new_xyz || !xyz || !idx_a || !idx_b) return fail(SPS_ERR_INVALID, "ball_query_full2_wave: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "ball_query_full2_wave: batch %d exceeds the grid limit", b);
    hipLaunchKernelGGL(ball_query_wave_dual_kernel, dim3(divup(m, BQW_WAVES), b), dim3(64 * BQW_WAVES), 0, as_stream(stream),
                       n, m, radius_a * radius_a, radius_b * radius_b, nsample_a, nsample_b, new_xyz, xyz, idx_a, idx_b, 0, m,
                       (const int *)nullptr);
    return check_launch("ball_query_wave_dual_kernel");
}

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
