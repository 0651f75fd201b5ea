// fps_pruned4.hip -- the register-resident exact FPS of fps_pruned.hip on FOUR waves (one per SIMD) instead of eight.
//
// Why.  fps_pruned_kernel runs two waves per SIMD and its phases are issue-bound: of the ~820 wave-instructions a pick costs
// the compute unit, ~240 are the accept phase, which all eight waves evaluate redundantly, ~135 the bucket test, which uses
// half the lanes of every wave (32 buckets per wave), and the candidate reductions cost the same for 32 or 64 buckets
// (DESIGN.md 4.1).  Four waves with 64 bucket slots each halve all three.  What made the variant impossible before is the
// register file: 64 slots x {x, y, z, running distance, rank} = 320 registers per lane.  Measured on gfx950
// (tools/agpr_idx_probe.hip): VGPR-index mode (s_set_gpr_idx_on) DOES apply to the ACC operand of v_accvgpr_read_b32 /
// v_accvgpr_write_b32, source and destination -- so x, y, z and the running distances live in the 256 AGPRs of a wave that
// owns a SIMD (a[0..63] = x, a[64..127] = y, a[128..191] = z, a[192..255] = t), selected by a wave-uniform slot index exactly
// like the architectural registers of the eight-wave kernel, and the ranks (14 bits for <= 16 384 points) pack two per VGPR.
// A lone wave has nobody to fill its issue bubbles, so the round is restructured for instruction-level parallelism inside
// the wave: accepted centres are applied two at a time (their bucket tests and their touched buckets are independent chains),
// and the cached maxima of the buckets that changed are refreshed once per round, two interleaved reductions at a time.
//
// Same contract, same arithmetic, bit-identical picks and running distances (sampling_gpu.cu:93-208; the tie rule, the
// exact pruning test and the multi-pick acceptance proof are those of fps_pruned.hip).  Every wave publishes FOUR records
// (the maxima of its four best buckets), so a round still ranks sixteen.  Input: the scene as sorted by the pre-pass
// (fps_presort.hip); 8192 < n <= 16 384.
#include "fps_pruned_util.h"
#include "fps_sort_split.h"

#include <math.h>

#include <type_traits>

namespace sps {
int launch_fps_presort(int b, int n, const float *dataset, const float *temp, float *work, long long stride, hipStream_t st,
                       PresortGate *gate);

namespace {

constexpr int P4_WAVES = 4, P4_THREADS = 256, P4_SLOTS = 64, P4_T = 4, P4_REC = P4_WAVES * P4_T;   // 16 records per round

struct P4Shared {
    // per round parity: soa[field][record], fields {dist, keylo, x, y, z, bound}, records 4 w .. 4 w + 3 from wave w
    __attribute__((aligned(16))) int soa[2][6][P4_REC];
};

// ---- the AGPR-resident arrays (see the header): every access is one idx-mode bracket.  (s_set_gpr_idx_on writes M0, which
// is a reserved register the compiler never keeps a value in: its own idx-mode sequences overwrite it the same way.)
#define P4_AGPR_CLOBBERS "a0", "a63", "a64", "a127", "a128", "a191", "a192", "a255"
__device__ __forceinline__ void p4_read4(int s, float &x, float &y, float &z, float &t) {
    asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\t"
                 "v_accvgpr_read_b32 %0, a0\n\t"
                 "v_accvgpr_read_b32 %1, a64\n\t"
                 "v_accvgpr_read_b32 %2, a128\n\t"
                 "v_accvgpr_read_b32 %3, a192\n\t"
                 "s_set_gpr_idx_off"
                 : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(t)
                 : "s"(s)
                 : "memory");
}
__device__ __forceinline__ float p4_read_t(int s) {
    float t;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "v_accvgpr_read_b32 %0, a192\n\t"
                 "s_set_gpr_idx_off"
                 : "=&v"(t)
                 : "s"(s)
                 : "memory");
    return t;
}
__device__ __forceinline__ void p4_write_t(int s, float t) {
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(DST)\n\t"
                 "v_accvgpr_write_b32 a192, %0\n\t"
                 "s_set_gpr_idx_off"
                 :
                 : "v"(t), "s"(s)
                 : P4_AGPR_CLOBBERS, "memory");
}
__device__ __forceinline__ void p4_write4(int s, float x, float y, float z, float t) {
    asm volatile("s_set_gpr_idx_on %4, gpr_idx(DST)\n\t"
                 "v_accvgpr_write_b32 a0, %0\n\t"
                 "v_accvgpr_write_b32 a64, %1\n\t"
                 "v_accvgpr_write_b32 a128, %2\n\t"
                 "v_accvgpr_write_b32 a192, %3\n\t"
                 "s_set_gpr_idx_off"
                 :
                 : "v"(x), "v"(y), "v"(z), "v"(t), "s"(s)
                 : P4_AGPR_CLOBBERS, "memory");
}

// two independent wave-wide maxima interleaved (each chain's next step is two instructions away)
__device__ __forceinline__ void wave_max_i32_id2(int &a, int &b) {
#define SPS_STEP2(CTRL)                                  \
    "v_max_i32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t" \
    "v_max_i32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t" \
    "s_nop 0\n\t"
    asm volatile("s_nop 1\n\t" SPS_STEP2("row_shr:1 row_mask:0xf") SPS_STEP2("row_shr:2 row_mask:0xf")
                 SPS_STEP2("row_shr:4 row_mask:0xf") SPS_STEP2("row_shr:8 row_mask:0xf")
                 SPS_STEP2("row_bcast:15 row_mask:0xa") SPS_STEP2("row_bcast:31 row_mask:0xc")
                 : "+v"(a), "+v"(b));
#undef SPS_STEP2
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
}

}  // namespace

// PROF = diagnostic build (tools/fps4_profile.py): per-wave s_memtime sums of the round's segments go to `dbg`
template <bool PUBLISH, bool PROF = false>
__global__ __launch_bounds__(P4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void fps_pruned4_kernel(int n, int m, int l2, int rb, const float *__restrict__ dataset, float *__restrict__ temp,
                        int *__restrict__ idxs, int *__restrict__ progress, const float *__restrict__ presorted, long long pstride,
                        const unsigned long long *__restrict__ gate, int gate_stride, unsigned gate_tag,
                        unsigned long long *__restrict__ dbg = nullptr) {
    if (m <= 0) return;
    __shared__ P4Shared sh;
    const int scene = blockIdx.x;
    if (gate) {   // a scene the sorting pre-pass gave up on belongs to the launcher's follow-up launch (fps_pruned.hip)
        const bool raised = (unsigned)(__hip_atomic_load(gate + (size_t)scene * gate_stride, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT) >> 32) == gate_tag;
        if (raised) return;
    }
    const float *xyz = dataset + (size_t)scene * n * 3;
    const bool has_temp = !PUBLISH || temp != nullptr;
    if (has_temp) temp += (size_t)scene * n;
    idxs += (size_t)scene * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ------------------------------------------------------------------ load the wave's buckets
    // bucket g = s * P4_WAVES + wave lives in slot s; lane l holds sorted position g * 64 + l
    const int npad = ((n + 63) >> 6) << 6;
    const float *px = presorted + (size_t)scene * pstride, *py = px + npad, *pz = py + npad, *pt = pz + npad;
    const int *prk = reinterpret_cast<const int *>(pt + npad);
    typedef int vi32 __attribute__((ext_vector_type(32)));
    vi32 rk2;   // ranks, two per register: slot s in half (s & 1) of element s >> 1 (0xFFFF: padding)
    float blo_x = INFINITY, blo_y = INFINITY, blo_z = INFINITY, bhi_x = -INFINITY, bhi_y = -INFINITY, bhi_z = -INFINITY;
    int bmax = __float_as_int(-1.f);          // lane s: bits of slot s's largest running distance
    unsigned bkeylo = 0;                       // (0x0FFFFFFF - rank of that point) << 4
    float bpx = 0.f, bpy = 0.f, bpz = 0.f;    // its coordinates
    int bhold = 0;                             // lane holding it
#pragma unroll 1
    for (int s0 = 0; s0 < P4_SLOTS; s0 += 8) {
        float vx[8], vy[8], vz[8], vt[8];
        int vr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pos = ((s0 + u) * P4_WAVES + wave) * 64 + lane;
            const bool ok = pos < npad;      // (entries n .. npad-1 are the pre-pass's padding: NaN, -1, worst rank)
            const int q = ok ? pos : 0;
            vx[u] = px[q]; vy[u] = py[q]; vz[u] = pz[q]; vt[u] = pt[q]; vr[u] = prk[q];
            if (!ok) { vx[u] = NAN; vy[u] = NAN; vz[u] = NAN; vt[u] = -1.f; vr[u] = 0x0FFFFFFF; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u;
            p4_write4(s, vx[u], vy[u], vz[u], vt[u]);
            const int r16 = vr[u] > 0xFFFF ? 0xFFFF : vr[u];
            const int prev = rk2[s >> 1];
            rk2[s >> 1] = (s & 1) ? (prev | (r16 << 16)) : r16;
            float lx = vx[u], ly = vy[u], lz = vz[u], hx = vx[u], hy = vy[u], hz = vz[u];
            wave_box6(lx, ly, lz, hx, hy, hz);
            if (lane == s) { blo_x = lx; blo_y = ly; blo_z = lz; bhi_x = hx; bhi_y = hy; bhi_z = hz; }
        }
    }
    // rank of slot s's point in this lane
    auto rank_of = [&](int s) -> int {
        const int w = rk2[s >> 1];
        const int r = (s & 1) ? (int)((unsigned)w >> 16) : (w & 0xFFFF);
        return r == 0xFFFF ? 0x0FFFFFFF : r;
    };
    // scalar results of a refresh; committed to lane `slot` of the metadata registers by commit()
    int r_vmax = 0, r_keylo = 0, r_px = 0, r_py = 0, r_pz = 0, r_wl = 0;
    // finish a refresh whose maximum `vmax` over the bucket's running distances `tv` is known
    auto refresh_tail = [&](int s, float tv, int vmax, float xv, float yv, float zv) {
        const int tb = __float_as_int(tv);
        unsigned long long eq = __ballot(tb == vmax);
        int wl = __builtin_ctzll(eq);
        const int rv = rank_of(s);
        if (__builtin_popcountll(eq) > 1) {  // equal distances: the reference's tie rule decides
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
    auto commit = [&](int slot) {
        int m0 = bmax, m1 = (int)bkeylo, m2 = __float_as_int(bpx), m3 = __float_as_int(bpy), m4 = __float_as_int(bpz), m5 = bhold;
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
    auto refresh_one = [&](int s) {
        float xs, ys, zs, ts;
        p4_read4(s, xs, ys, zs, ts);
        const int vmax = wave_max_i32_id(__float_as_int(ts));
        refresh_tail(s, ts, vmax, xs, ys, zs);
        commit(s);
    };
    auto refresh_two = [&](int s0, int s1) {
        float x0, y0, z0, t0, x1, y1, z1, t1;
        p4_read4(s0, x0, y0, z0, t0);
        p4_read4(s1, x1, y1, z1, t1);
        int v0 = __float_as_int(t0), v1 = __float_as_int(t1);
        wave_max_i32_id2(v0, v1);
        refresh_tail(s0, t0, v0, x0, y0, z0);
        commit(s0);
        refresh_tail(s1, t1, v1, x1, y1, z1);
        commit(s1);
    };
#pragma unroll 1
    for (int s = 0; s < P4_SLOTS; s += 2) refresh_two(s, s + 1);

    if (tid == 0) {
        if constexpr (PUBLISH) __hip_atomic_store(&idxs[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else idxs[0] = 0;
    }
    __syncthreads();
    // centres accepted by the previous round and still to be applied: record r in lane 4 r of (ax, ay, az), bit 4 r of pend
    float ax = xyz[0], ay = xyz[1], az = xyz[2];
    unsigned long long pend = m > 1 ? 1ull : 0ull;

    // ------------------------------------------------------------------ sampling loop, several picks per round (fps_pruned.hip)
    constexpr int IMIN = (int)0x80000000;
    int crec = 0;  // the wave's records as lanes 0..23 publish them: lane 4 f + r = field f {dist, keylo, x, y, z, bound} of record r
    bool cand_stale = true;
    unsigned long long cand_slots = 0;  // bucket slots the records came from: only their refresh changes the records
    int j = 1;  // picks made so far
    int round = 0;
    unsigned long long tseg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ntouch = 0, nrefresh = 0;
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
    const unsigned long long t_setup = stamp();

    for (;;) {
        const unsigned long long s0 = stamp();
        // ---- 1. apply the accepted centres, two at a time; the cached maxima of changed buckets are refreshed afterwards
        unsigned long long dirty = 0;
        // one (centre, slot): re-evaluate the slot's 64 points (exactly the reference arithmetic, one point per lane)
        auto touch_one = [&](float cx, float cy, float cz, int s) {
            float xs, ys, zs, ts;
            p4_read4(s, xs, ys, zs, ts);
            const float tn = fmin_raw(sqdist(xs, ys, zs, cx, cy, cz), ts);
            p4_write_t(s, tn);
            if (!((dirty >> s) & 1ull)) {   // the cached maximum only changes if the point holding it moved closer to a sample
                const int hl = __builtin_amdgcn_readlane(bhold, s);
                const int oldmax = __builtin_amdgcn_readlane(bmax, s);
                if (__builtin_amdgcn_readlane(__float_as_int(tn), hl) != oldmax) dirty |= 1ull << s;
            }
        };
        auto touch_two = [&](float cx0, float cy0, float cz0, int s0, float cx1, float cy1, float cz1, int s1) {   // s0 != s1
            float x0, y0, z0, t0, x1, y1, z1, t1;
            p4_read4(s0, x0, y0, z0, t0);
            p4_read4(s1, x1, y1, z1, t1);
            const float n0 = fmin_raw(sqdist(x0, y0, z0, cx0, cy0, cz0), t0);
            const float n1 = fmin_raw(sqdist(x1, y1, z1, cx1, cy1, cz1), t1);
            p4_write_t(s0, n0);
            p4_write_t(s1, n1);
            const int hl0 = __builtin_amdgcn_readlane(bhold, s0), hl1 = __builtin_amdgcn_readlane(bhold, s1);
            const int om0 = __builtin_amdgcn_readlane(bmax, s0), om1 = __builtin_amdgcn_readlane(bmax, s1);
            const bool ch0 = __builtin_amdgcn_readlane(__float_as_int(n0), hl0) != om0;
            const bool ch1 = __builtin_amdgcn_readlane(__float_as_int(n1), hl1) != om1;
            dirty |= (ch0 ? 1ull << s0 : 0ull) | (ch1 ? 1ull << s1 : 0ull);
        };
        // one lane per bucket: can the centre lower any distance in the box?  (lb computed like the real distance: monotone)
        auto test = [&](float cx, float cy, float cz) -> unsigned long long {
            const float qx = __builtin_amdgcn_fmed3f(cx, blo_x, bhi_x);
            const float qy = __builtin_amdgcn_fmed3f(cy, blo_y, bhi_y);
            const float qz = __builtin_amdgcn_fmed3f(cz, blo_z, bhi_z);
            const float lb = sqdist(qx, qy, qz, cx, cy, cz);
            return __ballot(!(lb >= __int_as_float(bmax)));   // NaN -> not skipped
        };
        while (pend) {
            const int r = __builtin_ctzll(pend);
            pend &= pend - 1;
            const bool two = pend != 0;
            const int r1 = two ? __builtin_ctzll(pend) : r;
            pend &= pend - 1;               // (0 & -1 = 0 when there was no second one)
            const float cx0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), r));
            const float cy0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), r));
            const float cz0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), r));
            const float cx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), r1));
            const float cy1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), r1));
            const float cz1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), r1));
            unsigned long long todo0 = test(cx0, cy0, cz0);
            unsigned long long todo1 = two ? test(cx1, cy1, cz1) : 0ull;
            if constexpr (PROF) ntouch += __builtin_popcountll(todo0) + __builtin_popcountll(todo1);
            while (todo0 | todo1) {
                if (todo0 && todo1) {
                    const int s0 = __builtin_ctzll(todo0);
                    int s1 = __builtin_ctzll(todo1);
                    todo0 &= todo0 - 1;
                    if (s1 == s0) {   // the same bucket for both centres: one after the other
                        todo1 &= todo1 - 1;
                        touch_one(cx0, cy0, cz0, s0);
                        touch_one(cx1, cy1, cz1, s1);
                    } else {
                        todo1 &= todo1 - 1;
                        touch_two(cx0, cy0, cz0, s0, cx1, cy1, cz1, s1);
                    }
                } else if (todo0) {
                    const int s0 = __builtin_ctzll(todo0);
                    todo0 &= todo0 - 1;
                    if (todo0) {
                        const int s1 = __builtin_ctzll(todo0);
                        todo0 &= todo0 - 1;
                        touch_two(cx0, cy0, cz0, s0, cx0, cy0, cz0, s1);
                    } else {
                        touch_one(cx0, cy0, cz0, s0);
                    }
                } else {
                    const int s0 = __builtin_ctzll(todo1);
                    todo1 &= todo1 - 1;
                    if (todo1) {
                        const int s1 = __builtin_ctzll(todo1);
                        todo1 &= todo1 - 1;
                        touch_two(cx1, cy1, cz1, s0, cx1, cy1, cz1, s1);
                    } else {
                        touch_one(cx1, cy1, cz1, s0);
                    }
                }
            }
        }
        if (dirty & cand_slots) cand_stale = true;
        const unsigned long long s1 = stamp();
        if constexpr (PROF) nrefresh += __builtin_popcountll(dirty);
        while (dirty) {
            const int s0 = __builtin_ctzll(dirty);
            dirty &= dirty - 1;
            if (dirty) {
                const int s1 = __builtin_ctzll(dirty);
                dirty &= dirty - 1;
                refresh_two(s0, s1);
            } else {
                refresh_one(s0);
            }
        }
        if (j >= m) break;
        const unsigned long long s2 = stamp();

        // ---- 2. the wave's four records: the maxima of its four best buckets (largest distance, then largest inverted rank),
        //         each with the bound that takes over once it has been picked (the rest of its bucket; for the last one also
        //         every other bucket)
        if (cand_stale) {
            // best bucket among the lanes in `in` (distance bits vmax known): its lane
            auto pick_lane = [&](bool in, int vmax) -> int {
                const unsigned long long eq = __ballot(in && bmax == vmax);
                int wl = __builtin_ctzll(eq);
                if (__builtin_popcountll(eq) > 1) {
                    const int kl = (in && bmax == vmax) ? (int)(bkeylo >> 4) : -1;
                    const int kbest = wave_max_i32_id(kl);
                    wl = __builtin_ctzll(__ballot(kl == kbest));
                }
                return wl;
            };
            int wl[P4_T], vm[P4_T], ru[P4_T];   // lanes (= slots) of the four buckets, their maxima, the runner-ups inside them
            unsigned long long taken = 0;
            int vnext = 0;
            {
                int v = bmax;
                vm[0] = wave_max_i32_id(v);
                wl[0] = pick_lane(true, vm[0]);
                taken = 1ull << wl[0];
            }
#pragma unroll
            for (int k = 1; k <= P4_T; ++k) {
                // the next bucket's maximum together with the runner-up inside the previous bucket: two interleaved chains
                const bool in = !((taken >> lane) & 1ull);
                int a = in ? bmax : IMIN;
                const int hl = __builtin_amdgcn_readlane(bhold, wl[k - 1]);
                const float tw = p4_read_t(wl[k - 1]);
                int b = lane != hl ? __float_as_int(tw) : IMIN;
                wave_max_i32_id2(a, b);
                ru[k - 1] = b;
                if (k < P4_T) {
                    vm[k] = a;
                    wl[k] = pick_lane(in, a);
                    taken |= 1ull << wl[k];
                } else {
                    vnext = a;
                }
            }
            cand_slots = taken;
            auto record = [&](int k, int others, auto rr) {
                constexpr int R = decltype(rr)::value;
                const int klo = __builtin_amdgcn_readlane((int)bkeylo, wl[k]);
                const int qx = __builtin_amdgcn_readlane(__float_as_int(bpx), wl[k]);
                const int qy = __builtin_amdgcn_readlane(__float_as_int(bpy), wl[k]);
                const int qz = __builtin_amdgcn_readlane(__float_as_int(bpz), wl[k]);
                // the point's own running distance once it has been picked (0 unless its coordinates are Inf/NaN: then the
                // update leaves it where it is and the reference picks it again)
                const float fx = __int_as_float(qx), fy = __int_as_float(qy), fz = __int_as_float(qz);
                const float own = fmin_raw(sqdist(fx, fy, fz, fx, fy, fz), __int_as_float(vm[k]));
                const int bound = imax(others, __builtin_amdgcn_readfirstlane(__float_as_int(own)));
                put_lane<0 + R>(crec, vm[k]);
                put_lane<4 + R>(crec, klo);
                put_lane<8 + R>(crec, qx);
                put_lane<12 + R>(crec, qy);
                put_lane<16 + R>(crec, qz);
                put_lane<20 + R>(crec, bound);
            };
            record(0, ru[0], std::integral_constant<int, 0>{});
            record(1, ru[1], std::integral_constant<int, 1>{});
            record(2, ru[2], std::integral_constant<int, 2>{});
            record(3, imax(ru[3], vnext), std::integral_constant<int, 3>{});
            cand_stale = false;
        }
        // ---- 3. exchange: lanes 0..23 store the four records field by field, ONE barrier, then every wave evaluates all
        //         16 x 16 ordered pairs, four per lane: lane 4 j + b holds record j against records 4 b .. 4 b + 3
        const unsigned long long s3 = stamp();
        const int buf = round & 1;
        if (lane < 24) sh.soa[buf][lane >> 2][4 * wave + (lane & 3)] = crec;
        __syncthreads();
        const unsigned long long s4 = stamp();
        const int rj = lane >> 2, rb4 = lane & 3;
        const int4 id = *(const int4 *)&sh.soa[buf][0][rb4 * 4], ik = *(const int4 *)&sh.soa[buf][1][rb4 * 4];
        const int4 ixv = *(const int4 *)&sh.soa[buf][2][rb4 * 4], iyv = *(const int4 *)&sh.soa[buf][3][rb4 * 4];
        const int4 izv = *(const int4 *)&sh.soa[buf][4][rb4 * 4], ibv = *(const int4 *)&sh.soa[buf][5][rb4 * 4];
        const int jd = sh.soa[buf][0][rj], jk = sh.soa[buf][1][rj];
        const float jx = __int_as_float(sh.soa[buf][2][rj]), jy = __int_as_float(sh.soa[buf][3][rj]);
        const float jz = __int_as_float(sh.soa[buf][4][rj]);
        const float jt = __int_as_float(jd);
        int nbef = 0, nbad = 0;
        auto pair = [&](int idist, int iklo, int ixb, int iyb, int izb, int ibound) {
            const bool before = (idist > jd) | ((idist == jd) & ((unsigned)iklo > (unsigned)jk));  // no short circuit: no branches
            const float dij = sqdist(jx, jy, jz, __int_as_float(ixb), __int_as_float(iyb), __int_as_float(izb));   // point j, centre i
            const bool lowered = !(dij >= jt);
            const bool hidden = !(jt > __int_as_float(ibound));
            nbef += before ? 1 : 0;
            nbad += (before & (lowered | hidden)) ? 1 : 0;
        };
        pair(id.x, ik.x, ixv.x, iyv.x, izv.x, ibv.x);
        pair(id.y, ik.y, ixv.y, iyv.y, izv.y, ibv.y);
        pair(id.z, ik.z, ixv.z, iyv.z, izv.z, ibv.z);
        pair(id.w, ik.w, ixv.w, iyv.w, izv.w, ibv.w);
        int cnt = nbef | (nbad << 8);
        asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                     : "+v"(cnt));
        const int pos = cnt & 0xFF;
        const int firstbad = -wave_max_i32_id((cnt >> 8) ? -pos : -P4_REC);
        const int L = firstbad < m - j ? firstbad : m - j;
        const bool taken_rec = rb4 == 0 && pos < L;
        if (tid < 64 && taken_rec) {
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
        pend = __ballot(taken_rec && (j + pos) != m - 1);  // bit 4 r: record r
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
            unsigned long long *o = dbg + ((size_t)scene * P4_WAVES + wave) * 12;
            for (int i = 0; i < 6; ++i) o[i] = tseg[i];
            o[6] = ntouch; o[7] = nrefresh; o[8] = t_setup; o[9] = stamp();
        }
    }

    // the reference leaves the final running min-distances in `temp`
    if (has_temp) {
#pragma unroll 1
        for (int s = 0; s < P4_SLOTS; ++s) {
            const int pos = (s * P4_WAVES + wave) * 64 + lane;
            const float ts = p4_read_t(s);
            const int rk = rank_of(s);
            if (pos < n) temp[(int)pf_unrank((unsigned)rk, l2, rb)] = ts;
        }
    }
}

// the four-wave kernel behind the sorting pre-pass; -1 if it does not apply (the caller then takes fps_pruned_kernel)
int launch_fps_pruned4(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress, const float *work,
                       long long stride, const PresortGate &gate, hipStream_t st) {
    if (n <= 8192 || n > P4_SLOTS * P4_THREADS || m < 2 || !work) return -1;
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    if (progress)
        hipLaunchKernelGGL((fps_pruned4_kernel<true>), dim3(b), dim3(P4_THREADS), 0, st, n, m, l2, rb, dataset, temp, idxs, progress, work, stride,
                           gate.word, gate.stride, gate.tag);
    else
        hipLaunchKernelGGL((fps_pruned4_kernel<false>), dim3(b), dim3(P4_THREADS), 0, st, n, m, l2, rb, dataset, temp, idxs, progress, work, stride,
                           gate.word, gate.stride, gate.tag);
    return check_launch("fps_pruned4_kernel");
}

// DIAGNOSTIC (tools/fps4_profile.py): the s_memtime-instrumented build behind an un-gated pre-pass
int launch_fps_pruned4_profile(int b, int n, int m, const float *dataset, float *temp, int *idxs, const float *work, long long stride,
                               unsigned long long *dbg, hipStream_t st) {
    if (n <= 8192 || n > P4_SLOTS * P4_THREADS || m < 2 || !work) return fail(SPS_ERR_INVALID, "fps4 profile: 8192 < n <= 16384");
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    hipLaunchKernelGGL((fps_pruned4_kernel<false, true>), dim3(b), dim3(P4_THREADS), 0, st, n, m, l2, rb, dataset, temp, idxs,
                       (int *)nullptr, work, stride, (const unsigned long long *)nullptr, 0, 0u, dbg);
    return check_launch("fps_pruned4_kernel<profile>");
}

}  // namespace sps

// DIAGNOSTIC ONLY: the pre-pass + the instrumented four-wave kernel; dbg (B, 4 waves, 12) u64: cycle sums of {apply, refresh,
// candidates, publish + barrier, accept}, rounds, touched (centre, bucket) pairs, refreshed buckets, first / last stamp
extern "C" int sps_debug_fps4_profile(int b, int n, int m, const float *dataset, float *temp, int *idxs, float *work,
                                      unsigned long long *dbg, sps_stream_t stream) {
    using namespace sps;
    PresortGate gate{};
    const long long stride = sps_fps_workspace_floats(n);
    if (stride <= 0 || launch_fps_presort(b, n, dataset, temp, work, stride, as_stream(stream), &gate) != SPS_OK)
        return fail(SPS_ERR_INVALID, "fps4 profile: the sorting pre-pass declined (sps_init?)");
    return launch_fps_pruned4_profile(b, n, m, dataset, temp, idxs, work, stride, dbg, as_stream(stream));
}
