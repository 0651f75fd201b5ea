// fps_verify.hip -- D-FPS of an FPS-ordered cloud: verify the identity prefix instead of recomputing it.
//
// In IA-SSD / SPSNet layer k+1 runs D-FPS on the centroids layer k's D-FPS just produced, in the order they
// were produced (IASSD_backbone.py:128-134 -> pointnet2_modules.py:307-310).  Farthest point sampling has a
// prefix property: every sample it picks maximises the running min-distance over the WHOLE cloud, hence also
// over the subset of already-picked-plus-later-picked points, so re-running FPS on that ordered subset picks
// positions 0, 1, 2, ... again -- unless exact fp32 distance ties meet the (block-size dependent) tie rule,
// which differs between the two runs.  So instead of m-1 serial argmax rounds (0.84 ms for 8 x 4096 -> 1024)
// this file CHECKS the guess "idx = 0..m-1" in two embarrassingly parallel passes (tens of microseconds):
//   pass 1: T[j] = running distance of point j when it is picked = min(temp[j], min_{i<j} d(j, i))
//   pass 2: every point k replays its own running distance t_k over the centres 0..m-2 and, at every step j,
//           checks (t_k, rank k) against (T[j], rank j) under the reference's total order; any point that
//           would have beaten j flags its scene.
// Scenes whose guess is confirmed get idx = 0..m-1 and the exact final `temp`; flagged scenes are recomputed
// by the ordinary kernels (fps.hip / fps_pruned.hip skip the confirmed ones), so the result is always
// bit-identical to running the reference kernel -- the shortcut only changes the schedule.
#include "sps_common.h"

namespace sps {

constexpr int FV_THREADS = 256;
constexpr int FV_BATCH = 8;  // centres per scalar-load batch

__device__ __forceinline__ unsigned fv_rank(unsigned k, int bs, int l2, int rb) {
    const unsigned lowrev = (l2 == 0) ? 0u : (__brev(k & (unsigned)(bs - 1)) >> (32 - l2));
    return (lowrev << rb) | (k >> l2);
}

// pass 1: thread j computes T[j] over the centres i < j (j < m)
__global__ __launch_bounds__(FV_THREADS) void fps_prefix_dist_kernel(int n, int m, const float *__restrict__ xyz,
                                                                     const float *__restrict__ temp, float *__restrict__ T) {
    const int scene = blockIdx.y;
    const int j = blockIdx.x * FV_THREADS + threadIdx.x;
    xyz += (size_t)scene * n * 3;
    const bool live = j < m;
    const int jj = live ? j : 0;
    const float px = xyz[jj * 3], py = xyz[jj * 3 + 1], pz = xyz[jj * 3 + 2];
    float t = temp[(size_t)scene * n + jj];
    // all lanes of the wave walk the centres up to the wave's largest j (wave-uniform trip count -> scalar loads);
    // a lane stops taking the min once i reaches its own j
    const int jmax = __builtin_amdgcn_readfirstlane(blockIdx.x * FV_THREADS + (threadIdx.x | 63));
    const int stop = jmax < m ? jmax : m - 1;
    int i = 0;
    for (; i + FV_BATCH <= stop; i += FV_BATCH) {  // wave-uniform -> one scalar-load batch of 8 centres
        float c[FV_BATCH * 3];
#pragma unroll
        for (int u = 0; u < FV_BATCH * 3; ++u) c[u] = xyz[i * 3 + u];
#pragma unroll
        for (int u = 0; u < FV_BATCH; ++u) {
            const float d = sqdist(px, py, pz, c[u * 3], c[u * 3 + 1], c[u * 3 + 2]);
            t = (i + u < j) ? fminf(d, t) : t;
        }
    }
    for (; i < stop; ++i) {
        const float d = sqdist(px, py, pz, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
        t = (i < j) ? fminf(d, t) : t;
    }
    if (live) T[(size_t)scene * m + j] = t;
}

// pass 2: thread k replays its running distance and checks every step of the guess
__global__ __launch_bounds__(FV_THREADS) void fps_prefix_check_kernel(
    int n, int m, int bs, int l2, int rb, const float *__restrict__ xyz, const float *__restrict__ temp,
    const float *__restrict__ T, float *__restrict__ temp_done, int *__restrict__ idx, int *__restrict__ bad) {
    const int scene = blockIdx.y;
    const int k = blockIdx.x * FV_THREADS + threadIdx.x;
    xyz += (size_t)scene * n * 3;
    T += (size_t)scene * m;
    const bool live = k < n;
    const int kk = live ? k : 0;
    const float px = xyz[kk * 3], py = xyz[kk * 3 + 1], pz = xyz[kk * 3 + 2];
    float t = temp[(size_t)scene * n + kk];
    const unsigned myrank = fv_rank((unsigned)kk, bs, l2, rb);
    bool violated = false;
    // step j (1 <= j < m): centre j-1 has been applied, the guess says point j is picked now
    int j = 1;
    for (; j + FV_BATCH <= m; j += FV_BATCH) {  // centres j-1 .. j+6, one scalar-load batch
        float c[FV_BATCH * 3], tj[FV_BATCH];
#pragma unroll
        for (int u = 0; u < FV_BATCH * 3; ++u) c[u] = xyz[(j - 1) * 3 + u];
#pragma unroll
        for (int u = 0; u < FV_BATCH; ++u) tj[u] = T[j + u];
#pragma unroll
        for (int u = 0; u < FV_BATCH; ++u) {
            const float d = sqdist(px, py, pz, c[u * 3], c[u * 3 + 1], c[u * 3 + 2]);
            t = fminf(d, t);
            const unsigned rj = fv_rank((unsigned)(j + u), bs, l2, rb);
            violated |= (k != j + u) && (t > tj[u] || (t == tj[u] && myrank < rj));
        }
    }
    for (; j < m; ++j) {
        const int c = j - 1;
        const float d = sqdist(px, py, pz, xyz[c * 3], xyz[c * 3 + 1], xyz[c * 3 + 2]);
        t = fminf(d, t);
        const float tj = T[j];
        const unsigned rj = fv_rank((unsigned)j, bs, l2, rb);
        violated |= (k != j) && (t > tj || (t == tj && myrank < rj));
    }
    if (live) {
        temp_done[(size_t)scene * n + k] = t;  // the reference's final `temp` if the guess holds
        if (k < m) idx[(size_t)scene * m + k] = k;
        if (violated) bad[scene] = 1;
    }
}

}  // namespace sps

// Workspace (device, caller-allocated): work_T (B*m f32), work_temp (B*n f32), flags (B i32).
// `temp` is the usual caller-filled running-distance buffer (1e10); on return it holds the final values.
extern "C" int sps_fps_ordered_prefix(int b, int n, int m, const float *xyz, float *temp, int *idxs, float *work_T,
                                      float *work_temp, int *flags, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || m > n) return fail(SPS_ERR_INVALID, "fps_ordered_prefix: bad shape b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return SPS_OK;
    if (!xyz || !temp || !idxs || !work_T || !work_temp || !flags) return fail(SPS_ERR_INVALID, "fps_ordered_prefix: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "fps_ordered_prefix: batch too large");
    hipStream_t st = as_stream(stream);
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    if (hipMemsetAsync(flags, 0, sizeof(int) * (size_t)b, st) != hipSuccess) return fail(SPS_ERR_LAUNCH, "fps_ordered_prefix: memset failed");
    hipLaunchKernelGGL(fps_prefix_dist_kernel, dim3(divup(m, FV_THREADS), b), dim3(FV_THREADS), 0, st, n, m, xyz, temp, work_T);
    int rc = check_launch("fps_prefix_dist_kernel");
    if (rc != SPS_OK) return rc;
    hipLaunchKernelGGL(fps_prefix_check_kernel, dim3(divup(n, FV_THREADS), b), dim3(FV_THREADS), 0, st, n, m, bs, l2, rb, xyz,
                       temp, work_T, work_temp, idxs, flags);
    rc = check_launch("fps_prefix_check_kernel");
    if (rc != SPS_OK) return rc;
    // confirmed scenes: copy work_temp -> temp and stop; flagged scenes: the ordinary FPS kernel recomputes them
    return launch_fps_resolve(b, n, m, xyz, temp, idxs, flags, work_temp, st);
}
