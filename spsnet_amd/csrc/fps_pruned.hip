// fps_pruned.hip -- exact farthest point sampling with spatial pruning (gfx950).
//
// Same contract and bit-identical results as fps.hip / the reference kernel
// (pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:93-208), including the
// block-size-dependent tie rule (winner = largest running min-distance; ties -> smallest
// bit-reversed reference thread (k mod BS), then smallest k), but it does NOT touch every point in
// every iteration.
//
// Observation: once j samples exist, a new centre only lowers the running distance of the few points
// that are closer to it than to every earlier sample (about N/j of them).  The kernel therefore
//   1. sorts the scene's points by a 12-bit Morton-like cell key (in-kernel LDS counting sort), so
//      that every run of 64 consecutive sorted points -- a BUCKET -- is spatially compact;
//   2. keeps each bucket's points and running distances in the VGPRs of one wave (bucket g lives in
//      slot g / W of wave g % W: neighbouring buckets sit in different waves) together with the bucket's
//      bounding box, its current maximum distance and that maximum's tie-break rank and coordinates;
//   3. per iteration tests every bucket against the new centre with ONE lane per bucket:
//           lb = sqdist(clamp(centre, box), centre)     (same fp32 operation sequence as the real distance)
//      Rounding is monotone, so lb <= the computed distance of every point in the box; if lb >= the
//      bucket's maximum running distance, min(d, temp) == temp for all its points and the bucket is
//      skipped WITHOUT changing any result.  On KITTI-shaped clouds ~6 of 256 buckets survive the test;
//   4. re-evaluates only the surviving buckets (64 points, one per lane), refreshes their cached
//      maximum, and reduces the per-bucket maxima: DPP inside the wave, one 64-bit LDS atomic max
//      {distance | inverted rank | wave} across waves, ONE s_barrier per iteration.
// The arithmetic on every point that IS evaluated is exactly the reference's, and skipped points are
// provably unchanged, so indices and the final `temp` array are bit-identical to the brute-force sweep.
#include "fps_pruned_util.h"
#include "fps_sort_split.h"

#include <math.h>

#include <type_traits>

namespace sps {

struct PfShared {
    int hist[PF_BINS];
    // per round parity: soa[field][record], fields {dist, keylo, x, y, z, bound}, records 2w and 2w+1 from wave w
    __attribute__((aligned(16))) int soa[2][6][2 * PF_WAVES];
    float red[6][PF_WAVES];
    int wsum[PF_WAVES];
};

// P = bucket slots per wave: the wave holds P buckets x 64 points; N <= P * PF_THREADS.
// PROF = diagnostic build: per-wave s_memtime sums of the loop segments go to `dbg` (never shipped on the
// product path; the timed kernel is the PROF = false instantiation).
// RESOLVE = this launch follows a checked guess (fps_verify.hip): scenes whose guess was confirmed only install
// their final running distances.  A separate instantiation so that the plain kernel is untouched by it.
// PUBLISH = consumers on other CUs read idxs while this kernel is still running (sa_stack's chunked layer 0):
// the samples are stored write-through (sc1) by ONE lane, and after every 64th that lane drains its stores
// (s_waitcnt vmcnt(0)) and stores progress[scene] = number of samples written, also sc1 -- the R1 hand-off of
// cdna_hip_programming.md Guideline 16 (one lane signals for all its own stores; consumers poll relaxed).
// PRESORT = the scene arrives sorted: `presorted` holds, per scene, `pstride` floats of which 5 arrays of npad = 64 ceil(n / 64)
// elements {x, y, z, running distance, rank} in bucket order, padding included (fps_presort.hip) -- the set-up is 5 P
// coalesced loads per lane instead of the in-kernel sort.
template <int P, bool PROF = false, bool RESOLVE = false, bool PUBLISH = false, bool PRESORT = false>
__global__ __launch_bounds__(PF_THREADS) void fps_pruned_kernel(
    int n, int m, int bs, int l2, int rb, const float *__restrict__ dataset, float *__restrict__ temp,
    int *__restrict__ idxs, unsigned long long *__restrict__ dbg = nullptr, const int *__restrict__ redo = nullptr,
    const float *__restrict__ temp_done = nullptr, int *__restrict__ progress = nullptr,
    const float *__restrict__ presorted = nullptr, long long pstride = 0,
    const unsigned long long *__restrict__ gate = nullptr, int gate_stride = 0, unsigned gate_tag = 0) {
    if (m <= 0) return;
    __shared__ PfShared sh;
    __shared__ unsigned short sorted[PRESORT ? 64 : P * PF_THREADS];

    const int scene = blockIdx.x;
    // gate = the sorting pre-pass's per-scene give-up words (fps_presort.hip): the PRESORT launch skips a scene whose
    // pre-pass gave up (its workspace is incomplete), the launcher's follow-up launch of the self-sorting kernel takes
    // exactly those scenes -- normally none, a launch that ends here
    if (gate) {
        const bool raised = (unsigned)(__hip_atomic_load(gate + (size_t)scene * gate_stride, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT) >> 32) == gate_tag;
        if (raised == PRESORT) return;
    }
    if constexpr (RESOLVE) {
        if (fps_already_done(redo, temp_done, temp, scene, n)) return;
    }
    const float *xyz = dataset + (size_t)scene * n * 3;
    // PUBLISH launches may pass temp == nullptr: "all running distances start at 1e10 (what the reference's caller fills in,
    // pointnet2_utils.py:26) and nobody wants them back" -- the streamed layer then needs neither the fill launch in front of
    // the producer nor the write-back behind it
    const bool has_temp = !PUBLISH || temp != nullptr;
    if (has_temp) temp += (size_t)scene * n;
    idxs += (size_t)scene * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if constexpr (!PRESORT) {
        // ------------------------------------------------------------------ spatial sort (once)
        float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
        // eight points per trip: their 24 loads are in flight together (a plain strided loop waits ~1 us per iteration)
        for (int k0 = tid; k0 < n; k0 += 8 * PF_THREADS) {
            float v[8][3];
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * PF_THREADS;
                const int kk = k < n ? k : k0;
    #pragma unroll
                for (int a = 0; a < 3; ++a) v[u][a] = xyz[kk * 3 + a];
            }
    #pragma unroll
            for (int u = 0; u < 8; ++u)
    #pragma unroll
                for (int a = 0; a < 3; ++a) {
                    lo3[a] = fminf(lo3[a], v[u][a]);
                    hi3[a] = fmaxf(hi3[a], v[u][a]);
                }
        }
    #pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo3[a] = wave_allmin_f32(lo3[a]);
            hi3[a] = wave_allmax_f32(hi3[a]);
            if (lane == 0) { sh.red[a][wave] = lo3[a]; sh.red[3 + a][wave] = hi3[a]; }
        }
        for (int i = tid; i < PF_BINS; i += PF_THREADS) sh.hist[i] = 0;
        __syncthreads();
        float glo[3], ghi[3];
    #pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = sh.red[a][0], h = sh.red[3 + a][0];
    #pragma unroll
            for (int w = 1; w < PF_WAVES; ++w) { l = fminf(l, sh.red[a][w]); h = fmaxf(h, sh.red[3 + a][w]); }
            glo[a] = l; ghi[a] = h;
        }
        const PfGrid grid = pf_make_grid(glo, ghi);
        // cell keys of this thread's points stay in registers between the histogram and the scatter pass
        int ckey[P];
        constexpr int KB = P % 8 == 0 ? 8 : 4;  // points per batch: loads first (unconditional, clamped), then keys + atomics
    #pragma unroll
        for (int i0 = 0; i0 < P; i0 += KB) {
            float v[KB][3];
    #pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int k = tid + (i0 + u) * PF_THREADS;
                const int kk = k < n ? k : 0;
    #pragma unroll
                for (int a = 0; a < 3; ++a) v[u][a] = xyz[kk * 3 + a];
            }
    #pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int k = tid + (i0 + u) * PF_THREADS;
                ckey[i0 + u] = 0;
                if (k < n) {
                    ckey[i0 + u] = pf_cell_key(grid, v[u][0], v[u][1], v[u][2]);
                    atomicAdd(&sh.hist[ckey[i0 + u]], 1);
                }
            }
        }
        __syncthreads();
        {   // exclusive prefix sum of the histogram: 8 bins per thread, wave scan, cross-wave offsets
            constexpr int PER = PF_BINS / PF_THREADS;
            int loc[PER], sum = 0;
    #pragma unroll
            for (int i = 0; i < PER; ++i) { loc[i] = sh.hist[tid * PER + i]; sum += loc[i]; }
            int incl = sum;
    #pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            if (lane == 63) sh.wsum[wave] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += sh.wsum[w];
            int run = base + incl - sum;
    #pragma unroll
            for (int i = 0; i < PER; ++i) { sh.hist[tid * PER + i] = run; run += loc[i]; }
        }
        __syncthreads();
    #pragma unroll
        for (int i = 0; i < P; ++i) {
            const int k = tid + i * PF_THREADS;
            if (k < n) sorted[atomicAdd(&sh.hist[ckey[i]], 1)] = (unsigned short)k;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ load the wave's buckets
    // bucket g = s * PF_WAVES + wave lives in slot s; lane l holds sorted position g*64 + l
    // Register-resident buckets as VECTORS: a wave-uniform runtime slot index then lowers to VGPR-index mode
    // (s_set_gpr_idx_on + v_mov) instead of a branch tree over P cases -- taken branches are what a lone wave
    // pays most for (instruction-buffer refill), and there is one dispatch per touched bucket.
    typedef float vfP __attribute__((ext_vector_type(P)));
    typedef int viP __attribute__((ext_vector_type(P)));
    vfP x, y, z, t;
    viP rk;  // tie-break rank of the point (0x0FFFFFFF for padding lanes)
    if constexpr (PRESORT) {
        const int npad = ((n + 63) >> 6) << 6;
        const float *px = presorted + (size_t)scene * pstride, *py = px + npad, *pz = py + npad, *pt = pz + npad;
        const int *prk = reinterpret_cast<const int *>(pt + npad);
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int pos = (s * PF_WAVES + wave) * 64 + lane;
            const bool ok = pos < npad;      // (entries n .. npad-1 are the pre-pass's padding: NaN, -1, worst rank)
            const int q = ok ? pos : 0;
            x[s] = ok ? px[q] : NAN;
            y[s] = ok ? py[q] : NAN;
            z[s] = ok ? pz[q] : NAN;
            t[s] = ok ? pt[q] : -1.f;
            rk[s] = ok ? prk[q] : 0x0FFFFFFF;
        }
    } else {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int pos = (s * PF_WAVES + wave) * 64 + lane;
            const bool ok = pos < n;
            const int k = ok ? (int)sorted[pos] : 0;
            rk[s] = ok ? (int)pf_rank((unsigned)k, bs, l2, rb) : 0x0FFFFFFF;
            x[s] = ok ? xyz[k * 3 + 0] : NAN;  // NaN coordinates: never inside a box, distance stays -1
            y[s] = ok ? xyz[k * 3 + 1] : NAN;
            z[s] = ok ? xyz[k * 3 + 2] : NAN;
            t[s] = ok ? (has_temp ? temp[k] : 1e10f) : -1.f;
        }
    }

    // per-bucket metadata, bucket s of this wave in lane s
    float blo_x = INFINITY, blo_y = INFINITY, blo_z = INFINITY, bhi_x = -INFINITY, bhi_y = -INFINITY, bhi_z = -INFINITY;
    int bmax = __float_as_int(-1.f);          // bits of the bucket's largest running distance
    unsigned bkeylo = 0;                       // (0x0FFFFFFF - rank of that point) << 4
    float bpx = 0.f, bpy = 0.f, bpz = 0.f;    // its coordinates

    // recompute bucket S's cached maximum (after its distances changed)
    unsigned long long nslow = 0;  // (PROF) refreshes that needed the tie-break path
    int bhold = 0;                              // lane holding the bucket's maximum
    // scalar results of the last refresh; committed to lane `slot` of the metadata registers by commit()
    int r_vmax = 0, r_keylo = 0, r_px = 0, r_py = 0, r_pz = 0, r_wl = 0;
    // recompute a bucket's maximum from its 64 running distances `tv` (ranks / coordinates of its points in
    // rv, xv, yv, zv) -> r_*
    auto refresh = [&](float tv, int rv, float xv, float yv, float zv) {
        const int tb = __float_as_int(tv);
        const int vmax = wave_max_i32_id(tb);
        unsigned long long eq = __ballot(tb == vmax);
        int wl = __builtin_ctzll(eq);
        if (__builtin_popcountll(eq) > 1) {  // equal distances: the reference's tie rule decides
            if constexpr (PROF) nslow += 1;
            const int inv = (tb == vmax) ? (0x0FFFFFFF - rv) : -1;
            const int best = wave_max_i32_id(inv);
            wl = __builtin_ctzll(__ballot(inv == best));
        }
        const int rank = __builtin_amdgcn_readlane(rv, wl);
        r_px = __builtin_amdgcn_readlane(__float_as_int(xv), wl);
        r_py = __builtin_amdgcn_readlane(__float_as_int(yv), wl);
        r_pz = __builtin_amdgcn_readlane(__float_as_int(zv), wl);
        r_vmax = vmax;
        r_keylo = (int)((0x0FFFFFFFu - (unsigned)rank) << 4);
        r_wl = wl;
    };
    // bucket metadata lives in lane `slot`: six v_writelane with an SGPR lane select (no builtin exists).  The
    // SGPR sources were produced by v_readlane / SALU; s_nop 3 covers the VALU-writes-SGPR wait states hipcc
    // cannot see inside asm.
    auto commit = [&](int slot) {
        int m0 = bmax, m1 = (int)bkeylo, m2 = __float_as_int(bpx), m3 = __float_as_int(bpy), m4 = __float_as_int(bpz), m5 = bhold;
        // two different SGPRs (data + lane select) exceed gfx9's one-scalar-operand limit, so the lane select goes
        // through M0 (saved and restored: M0 belongs to the compiler)
        unsigned keep;
        asm volatile("s_mov_b32 %6, m0\n\t"
                     "s_mov_b32 m0, %13\n\t"
                     "s_nop 3\n\t"
                     "v_writelane_b32 %0, %7, m0\n\t"
                     "v_writelane_b32 %1, %8, m0\n\t"
                     "v_writelane_b32 %2, %9, m0\n\t"
                     "v_writelane_b32 %3, %10, m0\n\t"
                     "v_writelane_b32 %4, %11, m0\n\t"
                     "v_writelane_b32 %5, %12, m0\n\t"
                     "s_mov_b32 m0, %6"
                     : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "=&s"(keep)
                     : "s"(r_vmax), "s"(r_keylo), "s"(r_px), "s"(r_py), "s"(r_pz), "s"(r_wl), "s"(slot));
        bmax = m0; bkeylo = (unsigned)m1; bpx = __int_as_float(m2); bpy = __int_as_float(m3); bpz = __int_as_float(m4); bhold = m5;
    };
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const float lx = wave_allmin_f32(x[s]), ly = wave_allmin_f32(y[s]), lz = wave_allmin_f32(z[s]);
        const float hx = wave_allmax_f32(x[s]), hy = wave_allmax_f32(y[s]), hz = wave_allmax_f32(z[s]);
        if (lane == s) { blo_x = lx; blo_y = ly; blo_z = lz; bhi_x = hx; bhi_y = hy; bhi_z = hz; }
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
        refresh(t[s], rk[s], x[s], y[s], z[s]);
        commit(s);
    }

    if (tid == 0) {
        if constexpr (PUBLISH) __hip_atomic_store(&idxs[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else idxs[0] = 0;
    }
    __syncthreads();

    // ------------------------------------------------------------------ sampling loop, several picks per round
    // Every round each wave publishes its best point e_w (largest key = {running distance, inverted rank}) plus a
    // BOUND_w >= the running distance of every other point it holds.  With the records sorted by key,
    // e_(1) > e_(2) > ..., sequential FPS is known to pick e_(1), ..., e_(L) in that order as long as for every k <= L
    //   (a) no earlier e_(i) lowers e_(k)'s running distance:  !(sqdist(e_(k), e_(i)) < t_(k)),     and
    //   (b) t_(k) > BOUND of every earlier e_(i)'s wave (which includes e_(i)'s own distance after it was picked)
    // because running distances only ever decrease: after the earlier picks e_(k) is unchanged while every other
    // point is still <= its old value, which was below e_(k)'s key -- (b) covers the points that were not
    // published.  The round accepts the longest such prefix (L >= 1: e_(1) is the plain FPS pick), then applies the L
    // updates (min is order-independent).  Rejecting is always safe, so every comparison that involves a NaN
    // rejects.  One barrier and one LDS exchange per ROUND instead of per sample.
    constexpr int IMIN = (int)0x80000000;
    int crec = 0;  // the wave's two records as lanes 0..11 publish them: lane 2f+r = field f {dist, keylo, x, y, z, bound} of record r
    bool cand_stale = true;
    int cand_slot = -1, cand_slot2 = -1;  // bucket slots the two records came from: only their refresh changes the records
    // centres accepted by the previous round and still to be applied: record r in lane r of (ax, ay, az)
    float ax = xyz[0], ay = xyz[1], az = xyz[2];
    unsigned long long pend = m > 1 ? 1ull : 0ull;
    int j = 1;  // picks made so far
    int round = 0;

    unsigned long long tseg[6] = {0, 0, 0, 0, 0, 0}, ntouch = 0, why[4] = {0, 0, 0, 0};
    nslow = 0;
    auto stamp = [&]() -> unsigned long long {
        if constexpr (PROF) {
            unsigned long long tt;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            return tt;
        } else {
            return 0ull;
        }
    };
    // Two waves share a SIMD and the arbiter serves the older one (waves 0..PF_WAVES/2-1) first: measured, the younger
    // took 3.2 k cycles for the apply phase and 1.66 k for the accept phase against 2.5 k and 1.23 k, and the older then
    // idled ~1 k cycles at the barrier.  Alternating the user priority by phase splits the penalty between the two.
    const bool younger = wave >= PF_WAVES / 2;
    for (;;) {
        if (younger) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        const unsigned long long s0 = stamp();
        while (pend) {
            const int r = __builtin_ctzll(pend);
            pend &= pend - 1;
            const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), r));
            const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), r));
            const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), r));
            // 1. one lane per bucket: can the new centre lower any distance in the box?
            const float qx = __builtin_amdgcn_fmed3f(cx, blo_x, bhi_x);
            const float qy = __builtin_amdgcn_fmed3f(cy, blo_y, bhi_y);
            const float qz = __builtin_amdgcn_fmed3f(cz, blo_z, bhi_z);
            const float lb = sqdist(qx, qy, qz, cx, cy, cz);
            const bool skip = lb >= __int_as_float(bmax);  // NaN -> not skipped
            unsigned long long todo = __ballot(!skip && lane < P);
            if constexpr (PROF) ntouch += __builtin_popcountll(todo);
            // 2. re-evaluate the surviving buckets (exactly the reference arithmetic, one point per lane)
            while (todo) {
                const int s = __builtin_ctzll(todo);
                todo &= todo - 1;
                // the cached maximum only changes if the point holding it moved closer to a sample
                const int hl = __builtin_amdgcn_readlane(bhold, s);
                const int oldmax = __builtin_amdgcn_readlane(bmax, s);
                const float xs = x[s], ys = y[s], zs = z[s];  // wave-uniform dynamic index -> VGPR-index mode
                const float d = sqdist(xs, ys, zs, cx, cy, cz);
                const float tn = fmin_raw(d, t[s]);
                t[s] = tn;
                const bool changed = __builtin_amdgcn_readlane(__float_as_int(tn), hl) != oldmax;
                if (changed) refresh(tn, rk[s], xs, ys, zs);
                if (changed) {
                    commit(s);
                    if (s == cand_slot || s == cand_slot2) cand_stale = true;
                }
            }
        }
        if (j >= m) break;
        const unsigned long long s1 = stamp();
        // 3. the wave's two records: the maxima of its two best buckets (largest distance, then largest inverted rank)
        //    and, for each, the bound that takes over once it has been picked
        if (cand_stale) {
            const bool mine = lane < P;
            // best bucket among the lanes in `in`: its distance bits and lane
            auto best_bucket = [&](bool in, int &vmax) -> int {
                vmax = wave_max_i32_id(in ? bmax : IMIN);
                const unsigned long long eq = __ballot(in && bmax == vmax);
                int wl = __builtin_ctzll(eq);
                if (__builtin_popcountll(eq) > 1) {
                    const int kl = (in && bmax == vmax) ? (int)(bkeylo >> 4) : -1;
                    const int kbest = wave_max_i32_id(kl);
                    wl = __builtin_ctzll(__ballot(kl == kbest));
                }
                return wl;
            };
            int v1, v2;
            const int wl1 = best_bucket(mine, v1);
            const int wl2 = best_bucket(mine && lane != wl1, v2);
            // the best of the other buckets, and the runner-up inside each of the two buckets: one interleaved reduction
            const int hl1 = __builtin_amdgcn_readlane(bhold, wl1), hl2 = __builtin_amdgcn_readlane(bhold, wl2);
            const float tw1 = t[wl1], tw2 = t[wl2];
            int v3 = (mine && lane != wl1 && lane != wl2) ? bmax : IMIN;
            int t21 = lane != hl1 ? __float_as_int(tw1) : IMIN;
            int t22 = lane != hl2 ? __float_as_int(tw2) : IMIN;
            wave_max_i32_id3(v3, t21, t22);
            auto record = [&](int wl, int vmax, int others, auto r) {
                constexpr int R = decltype(r)::value;
                const int klo = __builtin_amdgcn_readlane((int)bkeylo, wl);
                const int px = __builtin_amdgcn_readlane(__float_as_int(bpx), wl);
                const int py = __builtin_amdgcn_readlane(__float_as_int(bpy), wl);
                const int pz = __builtin_amdgcn_readlane(__float_as_int(bpz), wl);
                // the point's own running distance once it has been picked (0 unless its coordinates are Inf/NaN:
                // then the update leaves it where it is and the reference picks it again)
                const float fx = __int_as_float(px), fy = __int_as_float(py), fz = __int_as_float(pz);
                const float own = fmin_raw(sqdist(fx, fy, fz, fx, fy, fz), __int_as_float(vmax));
                const int bound = imax(others, __builtin_amdgcn_readfirstlane(__float_as_int(own)));
                put_lane<0 + R>(crec, vmax);
                put_lane<2 + R>(crec, klo);
                put_lane<4 + R>(crec, px);
                put_lane<6 + R>(crec, py);
                put_lane<8 + R>(crec, pz);
                put_lane<10 + R>(crec, bound);
            };
            // once record 1 is gone: the rest of its bucket (everything else is still below record 2);
            // once record 2 is gone too: the rest of its bucket and all other buckets
            record(wl1, v1, t21, std::integral_constant<int, 0>{});
            record(wl2, v2, imax(t22, v3), std::integral_constant<int, 1>{});
            cand_slot = wl1;
            cand_slot2 = wl2;
            cand_stale = false;
        }
        const unsigned long long s2 = stamp();
        // 4. exchange: lanes 0..11 store the two records field by field (soa[field][record]), ONE barrier, then every
        //    wave evaluates all 16x16 ordered pairs, four per lane: lane 4j+b holds record j against records 4b..4b+3
        const int buf = round & 1;
        if (lane < 12) sh.soa[buf][lane >> 1][2 * wave + (lane & 1)] = crec;
        const unsigned long long s3 = stamp();
        __syncthreads();
        const unsigned long long s4 = stamp();
        if (younger) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
        const int rj = lane >> 2, rb4 = lane & 3;
        const int4 id = *(const int4 *)&sh.soa[buf][0][rb4 * 4], ik = *(const int4 *)&sh.soa[buf][1][rb4 * 4];
        const int4 ixv = *(const int4 *)&sh.soa[buf][2][rb4 * 4], iyv = *(const int4 *)&sh.soa[buf][3][rb4 * 4];
        const int4 izv = *(const int4 *)&sh.soa[buf][4][rb4 * 4], ibv = *(const int4 *)&sh.soa[buf][5][rb4 * 4];
        const int jd = sh.soa[buf][0][rj], jk = sh.soa[buf][1][rj];
        const float jx = __int_as_float(sh.soa[buf][2][rj]), jy = __int_as_float(sh.soa[buf][3][rj]);
        const float jz = __int_as_float(sh.soa[buf][4][rj]);
        const float jt = __int_as_float(jd);
        int nbef = 0, nbad = 0, nlow = 0;  // nlow (PROF): earlier records that lower this one's distance
        auto pair = [&](int idist, int iklo, int ixb, int iyb, int izb, int ibound) {
            const bool before = (idist > jd) | ((idist == jd) & ((unsigned)iklo > (unsigned)jk));  // no short circuit: no branches
            // as the update would compute it: point j, centre i
            const float dij = sqdist(jx, jy, jz, __int_as_float(ixb), __int_as_float(iyb), __int_as_float(izb));
            const bool lowered = !(dij >= jt);
            const bool hidden = !(jt > __int_as_float(ibound));
            nbef += before ? 1 : 0;
            nbad += (before & (lowered | hidden)) ? 1 : 0;
            if constexpr (PROF) nlow += (before & lowered) ? 1 : 0;
        };
        pair(id.x, ik.x, ixv.x, iyv.x, izv.x, ibv.x);
        pair(id.y, ik.y, ixv.y, iyv.y, izv.y, ibv.y);
        pair(id.z, ik.z, ixv.z, iyv.z, izv.z, ibv.z);
        pair(id.w, ik.w, ixv.w, iyv.w, izv.w, ibv.w);
        // sum over the four lanes of record j: its position in the order, and whether it may follow
        int cnt = nbef | (nbad << 8);
        asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                     : "+v"(cnt));
        const int pos = cnt & 0xFF;
        const int firstbad = -wave_max_i32_id((cnt >> 8) ? -pos : -2 * PF_WAVES);
        int L = firstbad < m - j ? firstbad : m - j;
        if constexpr (PROF) {
            // why did the accepted prefix end?  the first rejected record was lowered by an earlier one (a), only hidden
            // behind an unpublished point (b), or nothing was rejected (all 2 * PF_WAVES records taken / end of the run)
            const unsigned long long stop = __ballot(rb4 == 0 && pos == firstbad && ((cnt >> 8) & 0xFF));
            const unsigned long long low = __ballot(nlow > 0) & 0x1111111111111111ull;
            unsigned long long lowrec = low | ((__ballot(nlow > 0) >> 1) & 0x1111111111111111ull) |
                                        ((__ballot(nlow > 0) >> 2) & 0x1111111111111111ull) | ((__ballot(nlow > 0) >> 3) & 0x1111111111111111ull);
            if (!stop) why[2] += 1;
            else if (stop & lowrec) why[0] += 1;
            else why[1] += 1;
            why[3] += (unsigned long long)L;
        }
        const bool taken = rb4 == 0 && pos < L;
        if (tid < 64 && taken) {
            const unsigned rank = 0x0FFFFFFFu - ((unsigned)jk >> 4);
            const int picked = (int)pf_unrank(rank, l2, rb);
            if constexpr (PUBLISH) __hip_atomic_store(&idxs[j + pos], picked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else idxs[j + pos] = picked;
        }
        if constexpr (PUBLISH) {
            if (tid == 0 && (((j + L) >> 6) != (j >> 6) || j + L == m)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left
                __hip_atomic_store(&progress[blockIdx.x], j + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the reference never applies its last pick to `temp`
        pend = __ballot(taken && (j + pos) != m - 1);  // bit 4r: record r
        ax = jx; ay = jy; az = jz;
        j += L;
        round += 1;
        if constexpr (PROF) {
            const unsigned long long s5 = stamp();
            tseg[0] += s1 - s0; tseg[1] += s2 - s1; tseg[2] += s3 - s2; tseg[3] += s4 - s3; tseg[4] += s5 - s4;
            tseg[5] += 1;
        }
    }
    if constexpr (PROF) {
        if (lane == 0 && dbg) {
            unsigned long long *o = dbg + ((size_t)scene * PF_WAVES + wave) * 12;
            for (int i = 0; i < 6; ++i) o[i] = tseg[i];
            o[6] = ntouch; o[7] = nslow;
            for (int i = 0; i < 4; ++i) o[8 + i] = why[i];
        }
    }

    // the reference leaves the final running min-distances in `temp`
    if (has_temp) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int pos = (s * PF_WAVES + wave) * 64 + lane;
            if (pos < n) temp[PRESORT ? (int)pf_unrank((unsigned)rk[s], l2, rb) : (int)sorted[pos]] = t[s];
        }
    }
}

// diagnostic: P = 32 profile build on one configuration (tools/fps_profile.py)
int launch_fps_pruned_profile(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                              unsigned long long *dbg, hipStream_t st) {
    if (divup(n, PF_THREADS) > 32 || n < 2048) return fail(SPS_ERR_INVALID, "fps profile build: 2048 <= n <= 16384");
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    hipLaunchKernelGGL((fps_pruned_kernel<32, true>), dim3(b), dim3(PF_THREADS), 0, st, n, m, bs, l2, rb, dataset, temp,
                       idxs, dbg);
    return check_launch("fps_pruned_kernel<profile>");
}

// publishing variant for chunked consumers; -1 if the pruned kernel does not apply to this size
int launch_fps_presort(int b, int n, const float *dataset, const float *temp, float *work, long long stride, hipStream_t st,
                       PresortGate *gate);

int launch_fps_pruned_publish(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress,
                              hipStream_t st, float *work, long long stride) {
    if (n < 6144 || n > 32 * PF_THREADS || m < 2) return -1;
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int P = divup(n, PF_THREADS);
    dim3 grid(b), block(PF_THREADS);
    PresortGate gate{};
    if (work && launch_fps_presort(b, n, dataset, temp, work, stride, st, &gate) == SPS_OK) {   // sorted by a pre-pass of K workgroups per scene
        // (behind it: the scenes the pre-pass gave up on -- a bounded poll that ran out, normally none -- through the kernel
        //  that sorts for itself; it publishes the same way)
#define SPS_PFS_CASE(PP)                                                                                         \
        if (P <= PP) {                                                                                           \
            hipLaunchKernelGGL((fps_pruned_kernel<PP, false, false, true, true>), grid, block, 0, st, n, m, bs, l2, rb, dataset, \
                               temp, idxs, (unsigned long long *)nullptr, (const int *)nullptr, (const float *)nullptr, \
                               progress, (const float *)work, stride, gate.word, gate.stride, gate.tag);         \
            if (gate.word)                                                                                       \
            hipLaunchKernelGGL((fps_pruned_kernel<PP, false, false, true>), grid, block, 0, st, n, m, bs, l2, rb, dataset, \
                               temp, idxs, (unsigned long long *)nullptr, (const int *)nullptr, (const float *)nullptr, \
                               progress, (const float *)nullptr, 0ll, gate.word, gate.stride, gate.tag);         \
            return check_launch("fps_pruned_kernel<publish, presorted>");                                       \
        }
        SPS_PFS_CASE(16)
        SPS_PFS_CASE(32)
#undef SPS_PFS_CASE
    }
#define SPS_PFP_CASE(PP)                                                                                         \
    if (P <= PP) {                                                                                               \
        hipLaunchKernelGGL((fps_pruned_kernel<PP, false, false, true>), grid, block, 0, st, n, m, bs, l2, rb, dataset, \
                           temp, idxs, (unsigned long long *)nullptr, (const int *)nullptr, (const float *)nullptr, \
                           progress);                                                                            \
        return check_launch("fps_pruned_kernel<publish>");                                                      \
    }
    SPS_PFP_CASE(16)
    SPS_PFP_CASE(32)
#undef SPS_PFP_CASE
    return -1;
}

// returns SPS_OK after launching, or -1 if this variant does not apply (caller falls back to fps.hip)
int launch_fps_pruned(int b, int n, int m, const float *dataset, float *temp, int *idxs, hipStream_t st,
                      const int *redo, const float *temp_done, float *work, long long stride) {
    // measured cross-over (tools/fps_time.py): below ~6k points the brute-force register kernel's iteration
    // (N/1024 points per lane) is shorter than the pruned kernel's fixed test/reduce chain
    if (n < 6144 || n > 32 * PF_THREADS) return -1;  // 32 bucket slots per wave = 16 384 points
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int P = divup(n, PF_THREADS);
    dim3 grid(b), block(PF_THREADS);
    PresortGate gate{};
    if (work && !redo && P > 8 && launch_fps_presort(b, n, dataset, temp, work, stride, st, &gate) == SPS_OK) {
#define SPS_PFS_CASE(PP)                                                                                       \
        if (P <= PP) {                                                                                         \
            hipLaunchKernelGGL((fps_pruned_kernel<PP, false, false, false, true>), grid, block, 0, st, n, m, bs, l2, rb, dataset, \
                               temp, idxs, (unsigned long long *)nullptr, (const int *)nullptr, (const float *)nullptr, \
                               (int *)nullptr, (const float *)work, stride, gate.word, gate.stride, gate.tag); \
            if (gate.word)                                                                                     \
            hipLaunchKernelGGL((fps_pruned_kernel<PP, false, false>), grid, block, 0, st, n, m, bs, l2, rb, dataset,  \
                               temp, idxs, (unsigned long long *)nullptr, (const int *)nullptr, (const float *)nullptr, \
                               (int *)nullptr, (const float *)nullptr, 0ll, gate.word, gate.stride, gate.tag); \
            return check_launch("fps_pruned_kernel<presorted>");                                              \
        }
        SPS_PFS_CASE(16)
        SPS_PFS_CASE(32)
#undef SPS_PFS_CASE
    }
#define SPS_PF_CASE(PP)                                                                                        \
    if (P <= PP) {                                                                                             \
        if (redo)                                                                                              \
            hipLaunchKernelGGL((fps_pruned_kernel<PP, false, true>), grid, block, 0, st, n, m, bs, l2, rb, dataset,   \
                               temp, idxs, (unsigned long long *)nullptr, redo, temp_done);                    \
        else                                                                                                   \
            hipLaunchKernelGGL((fps_pruned_kernel<PP, false, false>), grid, block, 0, st, n, m, bs, l2, rb, dataset,  \
                               temp, idxs, (unsigned long long *)nullptr, (const int *)nullptr,                \
                               (const float *)nullptr);                                                        \
        return check_launch("fps_pruned_kernel");                                                             \
    }
    SPS_PF_CASE(4)
    SPS_PF_CASE(8)
    SPS_PF_CASE(16)
    SPS_PF_CASE(32)
#undef SPS_PF_CASE
    return -1;
}

}  // namespace sps
