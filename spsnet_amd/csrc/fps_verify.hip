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

constexpr int FV_SEG = 8;                    // waves per workgroup = segments of the centre range
constexpr int FV_THREADS = 64 * FV_SEG;
constexpr int FV_FINE_MAX_POINTS = 4096;     // launches of at most this many points (all scenes) take the 16-point check kernel
constexpr int FV_MAX_M = 7168;               // centres staged in LDS: 20 B each (+ 2 KiB) of the 160 KiB, dynamic

__device__ __forceinline__ unsigned fv_rank(unsigned k, int bs, int l2, int rb) {
    const unsigned lowrev = (l2 == 0) ? 0u : (__brev(k & (unsigned)(bs - 1)) >> (32 - l2));
    return (lowrev << rb) | (k >> l2);
}

// Both passes use the same shape: a workgroup owns 64 points (one per lane) and its FV_SEG waves split the centre
// range into equal segments.  The centres are staged once in LDS as {x, y, z, T} and read back with wave-uniform
// (broadcast) ds_read_b128 -- the first version walked them with one scalar-load batch in flight per wave and was
// bound by that load's latency (136 us for 8 x 4096 points x 1024 centres with 4 waves per CU).
// min is order-independent, so splitting the running minimum over segments changes no value:
//   phase 1: wave s reduces its own segment to segmin[s][lane];
//   phase 2 (check only): wave s starts from min(temp, segmin[0..s-1]) and replays its segment step by step.

// pass 1: T[j] = min(temp[j], min_{i<j} d(j, i)) for j < m
__global__ __launch_bounds__(FV_THREADS) void fps_prefix_dist_kernel(int n, int m, const float *__restrict__ xyz,
                                                                     const float *__restrict__ temp, float *__restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) char fv_smem[];
    float4 *ctr = reinterpret_cast<float4 *>(fv_smem);                                   // [m]
    float (*segmin)[64] = reinterpret_cast<float (*)[64]>(fv_smem + (size_t)m * 16);     // [FV_SEG][64]
    const int scene = blockIdx.y;
    const int lane = threadIdx.x & 63, seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = blockIdx.x * 64, j = j0 + lane;
    xyz += (size_t)scene * n * 3;
    const int need = (j0 + 63 < m ? j0 + 63 : m - 1);  // centres 0 .. need-1 matter to this workgroup
    for (int i = threadIdx.x; i < need; i += FV_THREADS) ctr[i] = make_float4(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], 0.f);
    __syncthreads();
    const bool live = j < m;
    const int jj = live ? j : 0;
    const float px = xyz[jj * 3], py = xyz[jj * 3 + 1], pz = xyz[jj * 3 + 2];
    const int len = (need + FV_SEG - 1) / FV_SEG;
    const int ibeg = seg * len, iend = (ibeg + len < need) ? ibeg + len : need;
    float t = INFINITY;
    for (int i = ibeg; i < iend; ++i) {
        const float4 c = ctr[i];
        const float d = sqdist(px, py, pz, c.x, c.y, c.z);
        t = (i < j) ? fminf(d, t) : t;
    }
    segmin[seg][lane] = t;
    __syncthreads();
    if (seg == 0 && live) {
        float r = temp[(size_t)scene * n + j];
#pragma unroll
        for (int s = 0; s < FV_SEG; ++s) r = fminf(segmin[s][lane], r);
        T[(size_t)scene * m + j] = r;
    }
}

// pass 2: point k replays its running distance and checks every step of the guess.
// PTS points per workgroup (64 or 16): with 16, a wave holds four segments of the centre range side by side (lane = 16 sub +
// point), 32 segments in all -- a quarter of the serial chain per lane.  The chain (two passes over a segment's centres, ~14 us
// of a 21 us launch at 1024 centres whatever the number of workgroups) is all a small launch costs, and the streamed layer's
// last piece -- 256 centroids per scene, on the critical path behind the producer's last pick -- is such a launch.
template <int PTS>
__global__ __launch_bounds__(FV_THREADS) void fps_prefix_check_kernel(
    int n, int m, int bs, int l2, int rb, const float *__restrict__ xyz, const float *__restrict__ temp,
    const float *__restrict__ T, float *__restrict__ temp_done, int *__restrict__ idx, int *__restrict__ bad,
    const int *__restrict__ force_bad, int k0) {
    constexpr int SUB = 64 / PTS;            // segments side by side in a wave
    constexpr int NSEG = FV_SEG * SUB;
    extern __shared__ __attribute__((aligned(16))) char fv_smem[];
    float4 *step = reinterpret_cast<float4 *>(fv_smem);                                   // step j (1 <= j < m): {centre j-1, T[j]} at step[j-1]
    unsigned *srank = reinterpret_cast<unsigned *>(fv_smem + (size_t)m * 16);            // tie-break rank of point j, at srank[j-1]
    float (*segmin)[PTS] = reinterpret_cast<float (*)[PTS]>(fv_smem + (size_t)m * 20);   // [NSEG][PTS]
    const int scene = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & (PTS - 1), seg = wave * SUB + lane / PTS;
    const int k = k0 + blockIdx.x * PTS + pt;   // (points are independent: a launch may cover any range of them)
    xyz += (size_t)scene * n * 3;
    T += (size_t)scene * m;
    const int steps = m - 1;
    for (int i = threadIdx.x; i < steps; i += FV_THREADS) {
        step[i] = make_float4(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], T[i + 1]);
        srank[i] = fv_rank((unsigned)(i + 1), bs, l2, rb);
    }
    __syncthreads();
    const bool live = k < n;
    const int kk = live ? k : 0;
    const float px = xyz[kk * 3], py = xyz[kk * 3 + 1], pz = xyz[kk * 3 + 2];
    const int len = (steps + NSEG - 1) / NSEG;
    const int ibeg = seg * len < steps ? seg * len : steps, iend = (ibeg + len < steps) ? ibeg + len : steps;
    float local = INFINITY;
    for (int i = ibeg; i < iend; ++i) {
        const float4 c = step[i];
        local = fminf(sqdist(px, py, pz, c.x, c.y, c.z), local);
    }
    segmin[seg][pt] = local;
    __syncthreads();
    float t = temp[(size_t)scene * n + kk];
    for (int s = 0; s < seg; ++s) t = fminf(segmin[s][pt], t);
    const unsigned myrank = fv_rank((unsigned)kk, bs, l2, rb);
    bool violated = false;
    for (int i = ibeg; i < iend; ++i) {  // step j = i + 1: centre i has been applied, the guess says point j is picked now
        const float4 c = step[i];
        t = fminf(sqdist(px, py, pz, c.x, c.y, c.z), t);
        violated |= (k != i + 1) & ((t > c.w) | ((t == c.w) & (myrank < srank[i])));
    }
    if (live) {
        if (seg == NSEG - 1) {
            float r = fminf(local, t);  // == t when the last segment is not empty
            temp_done[(size_t)scene * n + k] = r;  // the reference's final `temp` if the guess holds
            if (k < m) idx[(size_t)scene * m + k] = k;
        }
        if (violated) bad[scene] = 1;
    }
    // the caller knows the inputs of pass 1 were not ready (a streamed layer's bounded wait gave up): every scene is
    // recomputed by the ordinary kernel, whatever this check concluded from them
    if (force_bad && threadIdx.x == 0 && *force_bad != 0) bad[scene] = 1;
}

}  // namespace sps

// Workspace (device, caller-allocated): work_T (B*m f32), work_temp (B*n f32), flags (B i32).
// `temp` is the usual caller-filled running-distance buffer (1e10); on return it holds the final values.
//
// The two passes are also exported separately: pass 1 reads only the first m points, so a caller that receives the
// cloud piecewise (sa_stack's streamed first layer) can run it as soon as those exist and pass 2 when the rest is in.
static int fv_check_args(const char *who, int b, int n, int m, const void *xyz, const void *temp, const void *work_T,
                         const void *flags) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || m > n) return fail(SPS_ERR_INVALID, "%s: bad shape b=%d n=%d m=%d", who, b, n, m);
    if (!xyz || !temp || !work_T || !flags) return fail(SPS_ERR_INVALID, "%s: null pointer", who);
    if (b > 65535) return fail(SPS_ERR_INVALID, "%s: batch too large", who);
    if (m > FV_MAX_M) return fail(SPS_ERR_INVALID, "%s: m=%d exceeds the %d centres the kernels stage in LDS", who, m, FV_MAX_M);
    return SPS_OK;
}

extern "C" int sps_fps_ordered_prefix_begin(int b, int n, int m, const float *xyz, const float *temp, float *work_T,
                                            int *flags, sps_stream_t stream) {
    using namespace sps;
    int rc = fv_check_args("fps_ordered_prefix_begin", b, n, m, xyz, temp, work_T, flags);
    if (rc != SPS_OK || b == 0 || m == 0) return rc;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(flags, 0, sizeof(int) * (size_t)b, st) != hipSuccess) return fail(SPS_ERR_LAUNCH, "fps_ordered_prefix: memset failed");
    const size_t lds = (size_t)m * 16 + sizeof(float) * FV_SEG * 64;
    static LdsLimitOnce raised;
    if (lds > 64 * 1024) {
        rc = raise_lds_limit((const void *)fps_prefix_dist_kernel, 160 * 1024, raised, "fps_prefix_dist_kernel");
        if (rc != SPS_OK) return rc;
    }
    hipLaunchKernelGGL(fps_prefix_dist_kernel, dim3(divup(m, 64), b), dim3(FV_THREADS), lds, st, n, m, xyz, temp, work_T);
    return check_launch("fps_prefix_dist_kernel");
}

// pass 2 over the points [k0, k0 + kcount) of every scene (k0 a multiple of 64; the range is clipped to the cloud)
static int fv_launch_check(int b, int n, int m, int k0, int kcount, const float *xyz, const float *temp, int *idxs,
                           const float *work_T, float *work_temp, int *flags, const int *force_redo, hipStream_t st,
                           bool may_be_fine) {
    using namespace sps;
    if (k0 < 0 || kcount < 0 || (k0 & 63)) return fail(SPS_ERR_INVALID, "fps_ordered_prefix: bad point range [%d,+%d)", k0, kcount);
    const int kend = (long long)k0 + kcount < n ? k0 + kcount : n;
    if (kend <= k0) return SPS_OK;
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const size_t lds = (size_t)m * 20 + sizeof(float) * FV_SEG * 64;   // (segmin: NSEG * PTS = 512 floats either way)
    // the SMALL launch somebody waits for (the streamed layer's last piece): the 16-point form -- more, shorter workgroups; a
    // whole cloud is throughput, not latency, and the 16-point form's extra staging makes it 5-8 % slower there (measured:
    // 8 x 4096 -> 1024 50.5 vs 53.4 us per call, 1 x 16 384 -> 4096 129 vs 137 us).  The early pieces run BESIDE the producer and nobody waits for them: four times the
    // workgroups (each staging all m centres) only take cycles from its compute units (config 5: the clustered FPS
    // 7.88 -> 8.36 ms with 16-point pieces beside it), so they keep the 64-point form.
    const bool fine = may_be_fine && (long long)b * (kend - k0) <= FV_FINE_MAX_POINTS;
    const void *fn = fine ? (const void *)fps_prefix_check_kernel<16> : (const void *)fps_prefix_check_kernel<64>;
    static LdsLimitOnce raised64, raised16;
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit(fn, 160 * 1024, fine ? raised16 : raised64, "fps_prefix_check_kernel");
        if (rc != SPS_OK) return rc;
    }
    if (fine)
        hipLaunchKernelGGL(fps_prefix_check_kernel<16>, dim3(divup(kend - k0, 16), b), dim3(FV_THREADS), lds, st, n, m, bs, l2, rb,
                           xyz, temp, work_T, work_temp, idxs, flags, force_redo, k0);
    else
        hipLaunchKernelGGL(fps_prefix_check_kernel<64>, dim3(divup(kend - k0, 64), b), dim3(FV_THREADS), lds, st, n, m, bs, l2, rb,
                           xyz, temp, work_T, work_temp, idxs, flags, force_redo, k0);
    return check_launch("fps_prefix_check_kernel");
}

extern "C" int sps_fps_ordered_prefix_check_range(int b, int n, int m, int k0, int kcount, const float *xyz, const float *temp,
                                                  int *idxs, const float *work_T, float *work_temp, int *flags,
                                                  sps_stream_t stream) {
    using namespace sps;
    int rc = fv_check_args("fps_ordered_prefix_check_range", b, n, m, xyz, temp, work_T, flags);
    if (rc != SPS_OK || b == 0 || m == 0) return rc;
    if (!idxs || !work_temp) return fail(SPS_ERR_INVALID, "fps_ordered_prefix_check_range: null pointer");
    return fv_launch_check(b, n, m, k0, kcount, xyz, temp, idxs, work_T, work_temp, flags, nullptr, as_stream(stream), false);
}

extern "C" int sps_fps_ordered_prefix_finish_from(int b, int n, int m, int k_from, const float *xyz, float *temp, int *idxs,
                                                  const float *work_T, float *work_temp, int *flags, const int *force_redo,
                                                  sps_stream_t stream) {
    using namespace sps;
    int rc = fv_check_args("fps_ordered_prefix_finish", b, n, m, xyz, temp, work_T, flags);
    if (rc != SPS_OK || b == 0 || m == 0) return rc;
    if (!idxs || !work_temp) return fail(SPS_ERR_INVALID, "fps_ordered_prefix_finish: null pointer");
    if (k_from < 0 || k_from >= n || (k_from & 63))   // (at least one block is left: it is the one that reads force_redo)
        return fail(SPS_ERR_INVALID, "fps_ordered_prefix_finish: k_from=%d must be a multiple of 64 below n=%d", k_from, n);
    hipStream_t st = as_stream(stream);
    rc = fv_launch_check(b, n, m, k_from, n - k_from, xyz, temp, idxs, work_T, work_temp, flags, force_redo, st, true);
    if (rc != SPS_OK) return rc;
    // confirmed scenes: copy work_temp -> temp and stop; flagged scenes: the ordinary FPS kernel recomputes them
    return launch_fps_resolve(b, n, m, xyz, temp, idxs, flags, work_temp, st);
}

extern "C" int sps_fps_ordered_prefix_finish(int b, int n, int m, const float *xyz, float *temp, int *idxs,
                                             const float *work_T, float *work_temp, int *flags, const int *force_redo,
                                             sps_stream_t stream) {
    return sps_fps_ordered_prefix_finish_from(b, n, m, 0, xyz, temp, idxs, work_T, work_temp, flags, force_redo, stream);
}

extern "C" int sps_fps_ordered_prefix(int b, int n, int m, const float *xyz, float *temp, int *idxs, float *work_T,
                                      float *work_temp, int *flags, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || m > n) return fail(SPS_ERR_INVALID, "fps_ordered_prefix: bad shape b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return SPS_OK;
    if (!xyz || !temp || !idxs || !work_T || !work_temp || !flags) return fail(SPS_ERR_INVALID, "fps_ordered_prefix: null pointer");
    // more centres than the verification kernels stage in LDS: no shortcut, the ordinary kernel computes the result
    if (m > FV_MAX_M) return sps_farthest_point_sampling_kernel_launcher(b, n, m, xyz, temp, idxs, stream);
    const int rc = sps_fps_ordered_prefix_begin(b, n, m, xyz, temp, work_T, flags, stream);
    if (rc != SPS_OK) return rc;
    return sps_fps_ordered_prefix_finish(b, n, m, xyz, temp, idxs, work_T, work_temp, flags, nullptr, stream);
}
