// fps.hip -- farthest point sampling for gfx950 (MI355X).
//
// Replaces farthest_point_sampling_kernel<BS> + launcher (reference
// pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:93-253) and the with-dist variant
// (:256-416).  Results are bit-identical to the reference algorithm, INCLUDING its tie rule,
// which is an artefact of its block size BS = opt_n_threads(n) and shared-memory tree:
//
//   winner = the point with the largest running min-distance; among equal distances the one
//            whose reference thread (k mod BS) is smallest in BIT-REVERSED order, then the
//            smallest k inside that thread.
//
// (Slot t of the tree absorbs slot t+stride only when strictly larger, for strides
// BS/2..1, so slot 0 prefers even threads over odd ones, then multiples of 4, ... :86-91,148-203.)
//
// Design (one workgroup per scene, like the reference, but nothing per-iteration touches HBM):
//   * thread T of the workgroup plays reference thread bitrev(T); with that relabelling the
//     tie rule becomes "lowest lane, then lowest wave", which is what ballot + ctz give for free.
//   * every thread keeps its P = ceil(n/BS) points AND their running min-distances in VGPRs for
//     the whole kernel (16 384 points -> 16 x 4 registers per lane); the reference re-reads
//     20 B/point/iteration through L2.
//   * per iteration: P fused distance/min/argmax updates, a DPP wave max, one LDS record per
//     wave {max, k, x, y, z}, ONE s_barrier (records are double-buffered), then every wave
//     reduces the <=16 records itself.  The reference needs 11 __syncthreads per iteration.
#include "sps_common.h"

#include <math.h>

namespace sps {

constexpr int FPS_MAX_THREADS = 1024;  // cuda_utils.h:6 TOTAL_THREADS
constexpr int FPS_MAX_WAVES = FPS_MAX_THREADS / 64;

struct FpsShared {
    float4 rec[2][FPS_MAX_WAVES];  // {x, y, z, k-as-bits} of each wave's candidate
    int best[2][FPS_MAX_WAVES];    // candidate distance (fp32 bits; non-negative, or -1.0f when idle)
};

// Reduce the per-wave candidates: every wave calls this after the barrier and gets the same
// answer.  Ties go to the lowest wave (= lowest relabelled thread).
__device__ __forceinline__ float4 fps_pick(const FpsShared &sh, int buf, int nwaves, int lane) {
    const int mine = (lane < nwaves) ? sh.best[buf][lane] : (int)0x80000000;
    int v = row_scan_max_i32(mine);
    const int gmax = __builtin_amdgcn_readlane(v, 15);
    const unsigned long long eq = __ballot(mine == gmax) & 0xFFFFull;
    const int ww = __builtin_ctzll(eq);
    return sh.rec[buf][ww];
}

// Register-resident FPS.  blockDim.x = max(64, BS) where BS = opt_n_threads(n) (a power of two),
// P = ceil(n / BS) slots per thread; slot s of reference thread r holds point s*BS + r.
// DIST = true is the with-dist variant: d comes from row `old` of an n x n matrix.
template <int P, bool DIST>
__global__ __launch_bounds__(FPS_MAX_THREADS) void fps_reg_kernel(
    int n, int m, int bs, int log2bs, const float *__restrict__ dataset, float *__restrict__ temp,
    int *__restrict__ idxs, const int *__restrict__ redo = nullptr, const float *__restrict__ temp_done = nullptr) {
    if (m <= 0) return;  // sampling_gpu.cu:101
    __shared__ FpsShared sh;

    const int scene = blockIdx.x;
    if (fps_already_done(redo, temp_done, temp, scene, n)) return;
    const float *xyz = dataset + (size_t)scene * n * (DIST ? (size_t)n : 3);
    temp += (size_t)scene * n;
    idxs += (size_t)scene * m;

    const int T = threadIdx.x;
    const int lane = T & 63, wave = T >> 6;
    const int nwaves = blockDim.x >> 6;
    // reference thread this lane plays (bit reversal over log2(BS) bits); lanes >= BS idle
    const bool live = T < bs;
    const int tref = (log2bs == 0) ? 0 : (int)(__brev((unsigned)T) >> (32 - log2bs));

    float x[P], y[P], z[P], t[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const int k = s * bs + tref;
        const bool ok = live && k < n;
        if (!DIST) {
            x[s] = ok ? xyz[k * 3 + 0] : 0.f;
            y[s] = ok ? xyz[k * 3 + 1] : 0.f;
            z[s] = ok ? xyz[k * 3 + 2] : 0.f;
        }
        t[s] = ok ? temp[k] : -1.f;  // -1 never beats the per-thread start value
    }

    int old = 0;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (!DIST) { cx = xyz[0]; cy = xyz[1]; cz = xyz[2]; }
    if (T == 0) idxs[0] = 0;

    for (int j = 1; j < m; ++j) {
        float best = -1.f;
        int bslot = 0;
        const float *drow = DIST ? xyz + (size_t)old * n : nullptr;
#pragma unroll
        for (int s = 0; s < P; ++s) {
            float d;
            if (DIST) {
                const int k = s * bs + tref;
                d = (live && k < n) ? drow[k] : 0.f;
            } else {
                d = sqdist(x[s], y[s], z[s], cx, cy, cz);
            }
            const float d2 = fminf(d, t[s]);
            t[s] = d2;
            const bool gt = d2 > best;  // strict: lowest slot (= lowest k) wins inside a thread
            bslot = gt ? s : bslot;
            best = gt ? d2 : best;
        }
        // distances are >= +0 (or the -1.0f idle marker): fp32 order == signed-int order
        const int bi = __float_as_int(best);
        const int wmax = wave_max_i32(bi);
        const unsigned long long eq = __ballot(bi == wmax);
        const int wl = __builtin_ctzll(eq);  // lowest lane = bit-reversed-smallest reference thread
        const int slot = __builtin_amdgcn_readlane(bslot, wl);
        const int k_ref = __builtin_amdgcn_readlane(tref, wl);
        float px = 0.f, py = 0.f, pz = 0.f;
        if (!DIST) {
            // fetch the candidate's coordinates out of the winner lane's registers
#pragma unroll
            for (int s = 0; s < P; ++s) {
                if (slot == s) {
                    px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[s]), wl));
                    py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[s]), wl));
                    pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[s]), wl));
                }
            }
        }
        const int buf = j & 1;
        if (lane == 0) {
            sh.rec[buf][wave] = make_float4(px, py, pz, __int_as_float(slot * bs + k_ref));
            sh.best[buf][wave] = wmax;
        }
        __syncthreads();
        const float4 w = fps_pick(sh, buf, nwaves, lane);
        cx = w.x; cy = w.y; cz = w.z;
        old = __float_as_int(w.w);
        if (T == 0) idxs[j] = old;
    }

    // the reference leaves the final running min-distances in `temp`
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const int k = s * bs + tref;
        if (live && k < n) temp[k] = t[s];
    }
}

// Any-n fallback: points and running distances stay in global memory (L2), 1024 threads in the
// reference's own thread<->point mapping, explicit (distance, bit-reversed thread) keys.
template <bool DIST>
__global__ __launch_bounds__(FPS_MAX_THREADS) void fps_stream_kernel(
    int n, int m, int bs, int log2bs, const float *__restrict__ dataset, float *__restrict__ temp,
    int *__restrict__ idxs, const int *__restrict__ redo = nullptr, const float *__restrict__ temp_done = nullptr) {
    if (m <= 0) return;
    __shared__ unsigned long long key[2][FPS_MAX_WAVES];
    __shared__ int cand[2][FPS_MAX_WAVES];

    const int scene = blockIdx.x;
    if (fps_already_done(redo, temp_done, temp, scene, n)) return;
    const float *xyz = dataset + (size_t)scene * n * (DIST ? (size_t)n : 3);
    temp += (size_t)scene * n;
    idxs += (size_t)scene * m;
    const int T = threadIdx.x, lane = T & 63, wave = T >> 6, nwaves = blockDim.x >> 6;
    const bool live = T < bs;
    const unsigned rank = (log2bs == 0) ? 0u : (__brev((unsigned)T) >> (32 - log2bs));

    int old = 0;
    if (T == 0) idxs[0] = 0;
    for (int j = 1; j < m; ++j) {
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (!DIST) { cx = xyz[old * 3]; cy = xyz[old * 3 + 1]; cz = xyz[old * 3 + 2]; }
        float best = -1.f;
        int besti = 0;
        if (live) {
            for (int k = T; k < n; k += bs) {
                const float d = DIST ? xyz[(size_t)old * n + k]
                                     : sqdist(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2], cx, cy, cz);
                const float d2 = fminf(d, temp[k]);
                temp[k] = d2;
                if (d2 > best) { best = d2; besti = k; }
            }
        }
        // key: distance bits (sign flipped so -1.0f sorts below +0), then inverted bit-reversed tid
        const unsigned hi = (unsigned)__float_as_int(best) ^ 0x80000000u;
        unsigned long long kkey = ((unsigned long long)hi << 32) | (0xFFFFFFFFu - rank);
        int kidx = besti;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long ok = __shfl_xor(kkey, off);
            const int oi = __shfl_xor(kidx, off);
            if (ok > kkey) { kkey = ok; kidx = oi; }
        }
        const int buf = j & 1;
        if (lane == 0) { key[buf][wave] = kkey; cand[buf][wave] = kidx; }
        __syncthreads();
        unsigned long long bk = key[buf][0];
        int bidx = cand[buf][0];
        for (int w = 1; w < nwaves; ++w) {
            const unsigned long long ok = key[buf][w];
            if (ok > bk) { bk = ok; bidx = cand[buf][w]; }
        }
        old = bidx;
        if (T == 0) idxs[j] = old;
        // temp[] written above is only re-read by the same thread: no extra barrier needed
    }
}

static int g_fps_mode = 0;  // 0 = auto (pruned where it applies), 1 = brute-force kernels only
int fps_mode() { return g_fps_mode; }

static int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }

template <bool DIST>
static int launch_fps(int b, int n, int m, const float *dataset, float *temp, int *idxs, hipStream_t st,
                      const int *redo = nullptr, const float *temp_done = nullptr) {
    if (b < 0 || n <= 0 || m < 0) return fail(SPS_ERR_INVALID, "fps: bad shape b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return SPS_OK;
    if (!dataset || !temp || !idxs) return fail(SPS_ERR_INVALID, "fps: null pointer");
    if (!DIST && g_fps_mode == 0) {  // spatially pruned variant (fps_pruned.hip) where it applies
        const int rc = launch_fps_pruned(b, n, m, dataset, temp, idxs, st, redo, temp_done);
        if (rc >= 0) return rc;
    }
    const int bs = sps_opt_n_threads(n);
    const int l2 = ilog2(bs);
    const int threads = bs < 64 ? 64 : bs;
    const int P = divup(n, bs);
    dim3 grid(b), block(threads);
#define SPS_FPS_CASE(PP)                                                                              \
    if (P <= PP) {                                                                                    \
        hipLaunchKernelGGL((fps_reg_kernel<PP, DIST>), grid, block, 0, st, n, m, bs, l2, dataset, temp, idxs, redo, temp_done); \
        return check_launch("fps_reg_kernel");                                                       \
    }
    SPS_FPS_CASE(1)
    SPS_FPS_CASE(2)
    SPS_FPS_CASE(4)
    SPS_FPS_CASE(8)
    SPS_FPS_CASE(12)
    SPS_FPS_CASE(16)
    SPS_FPS_CASE(20)
    SPS_FPS_CASE(24)
#undef SPS_FPS_CASE
    hipLaunchKernelGGL((fps_stream_kernel<DIST>), grid, block, 0, st, n, m, bs, l2, dataset, temp, idxs, redo, temp_done);
    return check_launch("fps_stream_kernel");
}

// fps_verify.hip: finish a batch whose guess was checked -- confirmed scenes only copy their final temp,
// flagged scenes (redo[scene] != 0) run the ordinary kernel
int launch_fps_resolve(int b, int n, int m, const float *dataset, float *temp, int *idxs, const int *redo,
                       const float *temp_done, hipStream_t st) {
    return launch_fps<false>(b, n, m, dataset, temp, idxs, st, redo, temp_done);
}

}  // namespace sps

extern "C" int sps_set_fps_mode(int mode) {
    const int old = sps::g_fps_mode;
    sps::g_fps_mode = mode;
    return old;
}

// Diagnostic only (tools/fps_profile.py): runs the s_memtime-instrumented build of the pruned kernel and
// fills dbg[b][8 waves][8] = {test, update, wave-reduce, publish, barrier, pick, touched, tie-path} sums.
extern "C" int sps_debug_fps_profile(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                     unsigned long long *dbg, sps_stream_t stream) {
    return sps::launch_fps_pruned_profile(b, n, m, dataset, temp, idxs, dbg, sps::as_stream(stream));
}

// One workgroup, one wave: spin (bounded, with s_sleep) until every scene has published `need` samples.
static unsigned g_wait_spins = 1u << 22;   // seconds: the producer is gone

// DIAGNOSTIC: the spin bound of sps_wait_progress (0 restores the default).  A tiny bound makes every wait give up at once,
// which is how the tests drive sa_stack's redo path; 0xFFFFFFFF makes every wait give up WITHOUT looking at the counter (a
// producer that is already done would otherwise let a one-spin wait through, and a test that asserts on the flag would
// depend on the host's pace).  Returns the previous bound.
extern "C" unsigned sps_debug_set_wait_spins(unsigned spins) {
    const unsigned old = g_wait_spins;
    g_wait_spins = spins ? spins : (1u << 22);
    return old;
}

__global__ __launch_bounds__(64) void wait_progress_kernel(const int *progress, int b, int need, int *timed_out, unsigned bound) {
    const int lane = threadIdx.x;
    if (bound == 0xFFFFFFFFu) {   // forced by sps_debug_set_wait_spins
        for (int s2 = lane; s2 < b; s2 += 64) timed_out[s2] = 1;
        return;
    }
    for (int base = 0; base < b; base += 64) {
        const int sc = base + lane;
        bool done = sc >= b;
        for (unsigned spins = 0;; ++spins) {
            if (!done) done = __hip_atomic_load(&progress[sc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
            if (__all(done)) break;
            if (spins >= bound) {
                for (int s2 = lane; s2 < b; s2 += 64) timed_out[s2] = 1;   // one flag per scene (all set: the batch is redone)
                return;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
}

// FPS whose output can be consumed while it runs (pruned kernel sizes only: 6144 <= n <= 16384).
// progress (B i32, device) counts the samples each scene has published.  The CALLER zeroes it -- before it lets
// any consumer stream go (a memset enqueued here would race with a wait already running on another stream).
extern "C" int sps_fps_publish(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress,
                               sps_stream_t stream) {
    using namespace sps;
    if (b <= 0 || n <= 0 || m <= 0 || !dataset || !idxs || !progress)   // temp may be NULL: start from 1e10, no write-back
        return fail(SPS_ERR_INVALID, "fps_publish: bad arguments");
    hipStream_t st = as_stream(stream);
    const int rc = launch_fps_pruned_publish(b, n, m, dataset, temp, idxs, progress, st);
    if (rc < 0) return fail(SPS_ERR_INVALID, "fps_publish: no publishing kernel for n=%d", n);
    return rc;
}

// The same for scenes of 16 385 .. 262 144 points: `work` = b * sps_fps_workspace_floats(n) floats (the clustered
// large-scene kernel publishes; batches it does not take -- more than 64 workgroups -- return SPS_ERR_INVALID and the caller
// runs the layer unstreamed).  Smaller scenes: sps_fps_publish, `work` unused.
extern "C" int sps_fps_publish_ws(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress,
                                  float *work, sps_stream_t stream) {
    using namespace sps;
    if (b <= 0 || n <= 0 || m <= 0 || !dataset || !idxs || !progress) return fail(SPS_ERR_INVALID, "fps_publish_ws: bad arguments");
    if (n <= 32 * 512) {   // register-resident kernel; with a workspace its scenes are sorted by a pre-pass (fps_presort.hip)
        if (!work || fps_mode() != 0) return sps_fps_publish(b, n, m, dataset, temp, idxs, progress, stream);
        const int rc = launch_fps_pruned_publish(b, n, m, dataset, temp, idxs, progress, as_stream(stream), work,
                                                 sps_fps_workspace_floats(n));
        if (rc < 0) return fail(SPS_ERR_INVALID, "fps_publish_ws: no publishing kernel for n=%d", n);
        return rc;
    }
    if (!work) return fail(SPS_ERR_INVALID, "fps_publish_ws: bad arguments");
    if (fps_mode() != 0) return fail(SPS_ERR_INVALID, "fps_publish_ws: brute-force mode has no publishing kernel");
    const int rc = launch_fps_big_publish(b, n, m, dataset, temp, idxs, progress, work, as_stream(stream));
    if (rc < 0) return fail(SPS_ERR_INVALID, "fps_publish_ws: no publishing kernel for b=%d n=%d", b, n);
    return rc;
}

// Block `stream` until every scene's progress counter reaches `need` (a tiny spinning kernel, bounded).
// timed_out (device i32, caller-zeroed) is set to 1 if the bound was hit.
extern "C" int sps_wait_progress(const int *progress, int b, int need, int *timed_out, sps_stream_t stream) {
    return sps_wait_progress_ex(progress, b, need, timed_out, 0, stream);
}

// patient != 0: the wait for the producer's LAST sample.  It spins ~64 times longer (minutes) and ignores the diagnostic bound:
// everything behind it -- the predicated repair of earlier waits included -- relies on the samples being complete, and a
// producer that has not finished after minutes means a hung device whatever this wait does.
extern "C" int sps_wait_progress_ex(const int *progress, int b, int need, int *timed_out, int patient, sps_stream_t stream) {
    using namespace sps;
    if (b <= 0 || !progress || !timed_out) return fail(SPS_ERR_INVALID, "wait_progress: bad arguments");
    hipLaunchKernelGGL(wait_progress_kernel, dim3(1), dim3(64), 0, as_stream(stream), progress, b, need, timed_out,
                       patient ? (1u << 28) : g_wait_spins);
    return check_launch("wait_progress_kernel");
}

// The ordinary FPS for the scenes with redo[scene] != 0 only; the others keep idxs and temp as they are.  With redo = the
// per-scene timed_out flags of sps_wait_progress this is the predicated repair of a D-FPS result that was derived from
// samples a timed-out wait let through (sa_stack._streamed_first_layer): normally one launch that does nothing.
extern "C" int sps_fps_redo_where(int b, int n, int m, const float *dataset, float *temp, int *idxs, const int *redo,
                                  sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0) return fail(SPS_ERR_INVALID, "fps_redo_where: bad shape b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return SPS_OK;
    if (!dataset || !temp || !idxs || !redo) return fail(SPS_ERR_INVALID, "fps_redo_where: null pointer");
    return launch_fps_resolve(b, n, m, dataset, temp, idxs, redo, temp, as_stream(stream));   // temp_done = temp: kept as is
}

extern "C" int sps_opt_n_threads(int work_size) {
    // cuda_utils.h:10-14 (double log, truncation, clamp to [1, 1024])
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > sps::FPS_MAX_THREADS) t = sps::FPS_MAX_THREADS;
    if (t < 1) t = 1;
    return t;
}

extern "C" int sps_farthest_point_sampling_kernel_launcher(int b, int n, int m, const float *dataset,
                                                           float *temp, int *idxs, sps_stream_t stream) {
    return sps::launch_fps<false>(b, n, m, dataset, temp, idxs, sps::as_stream(stream));
}

extern "C" int sps_furthest_point_sampling_with_dist_kernel_launcher(int b, int n, int m,
                                                                     const float *dataset, float *temp,
                                                                     int *idxs, sps_stream_t stream) {
    return sps::launch_fps<true>(b, n, m, dataset, temp, idxs, sps::as_stream(stream));
}
