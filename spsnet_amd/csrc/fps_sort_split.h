// fps_sort_split.h -- the spatial counting sort of the pruned FPS kernels, split over the K workgroups of a scene
// (fps_pruned_cluster.hip: the K workgroups that go on to sample the scene; fps_presort.hip: a pre-pass in front of the
// register-resident kernel).
//
// Workgroup c owns the points [n c / K, n (c+1) / K): (A) every workgroup takes the bounding box of the same strided sample
// of the cloud; (B) its cell histogram -> ghist[c][.] in the workspace (write-through stores + flag); everybody adds the K
// histograms up (exclusive scan over the cells + the counts of the workgroups before it = its own first slot in every cell);
// (C) it scatters its points {x, y, z, running distance, rank} with LDS atomics on those slots; optionally release + flag
// again so that every workgroup may read the whole sorted scene.  The order inside a cell is as arbitrary as with one
// workgroup's atomics and as irrelevant (ties are decided by rank).  The granules must be zeroed before the launch;
// all K workgroups of a scene should be resident at once (they spin on each other).  The spins are BOUNDED and a poll that
// gives up never traps: it raises the scene's give-up word, every workgroup of the scene leaves (pc_sort_split returns
// false to all its threads), and the launcher's predicated follow-up launch redoes the scene with the one-workgroup
// kernel -- correct or redo, the pattern of sa_stack's bounded progress waits.
#pragma once
#include "fps_pruned_util.h"

namespace sps {

constexpr int PC_MAXK = 16;       // workgroups per scene at most (layout of the flags / histograms of the exchange area; round 5: 16)
constexpr int PS_MAXK = 8;        // ... of the sorting pre-pass in front of the register-resident kernel (fps_presort.hip)
constexpr int PC_MAXT = 8;        // records a workgroup publishes per round (fps_pruned_cluster.hip)
constexpr int PC_MAXR = 64;       // records per round (K T) at most: one lane per record in the acceptance (round 5: 64, was 32)
// the exchange area in 8-byte granules: [8 ..): the rounds' records [parity][record][field]; the sort's two rounds of flags
// [2][K]; behind the granules the K cell histograms of the sort (ints).  The launchers zero the granules per launch.
constexpr int PC_FLAG_AT = 8 + 2 * PC_MAXR * 6;
constexpr int PC_GRANULES = PC_FLAG_AT + 2 * PC_MAXK + 8;
constexpr unsigned PC_SPIN_LIMIT = 1u << 24;          // polls before a stuck exchange gives up (~ seconds)
constexpr unsigned PC_SPIN_FORCED = 0xFFFFFFFFu;      // diagnostic bound: every poll gives up without looking (tests)
// the bound in force (fps_presort.hip; sps_debug_set_exchange_spins)
unsigned pc_spin_limit();
// where a launch behind the sorting pre-pass reads the pre-pass's per-scene give-up words (fps_presort.hip)
struct PresortGate {
    const unsigned long long *word;   // scene 0's word (NULL: no gate)
    int stride;                       // granules between the words of consecutive scenes
    unsigned tag;                     // a word is raised when its tag half equals this
};

__device__ __forceinline__ void granule_store(unsigned long long *p, int value, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same exchange for workgroups that share an XCD (round 5).  An agent-scope access (sc1) is coherent across the chip's eight L2
// caches, i.e. it is served from the memory side: ~1 us per hand-off (MI355X_MICROARCH.md).  Workgroups on ONE XCD share ONE L2, and
// that L2 executes every atomic read-modify-write itself: a swap without scope bits publishes a granule at the L2, an `or 0` WITH
// return reads it back from there (written as instructions: the compiler folds an idempotent read-modify-write into a plain load,
// which a CU may serve from its own L1 for ever).  Only legal between workgroups that have checked that they sit on the same XCD
// (fps_pruned_cluster.hip compares the XCC_ID registers the K workgroups publish with their sort flags).
__device__ __forceinline__ void granule_store_xcd(unsigned long long *p, int value, unsigned tag) {
    const unsigned long long v = ((unsigned long long)tag << 32) | (unsigned)value;
    asm volatile("global_atomic_swap_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long granule_load_xcd(const unsigned long long *p) {
    unsigned long long r;
    const unsigned long long zero = 0ull;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(zero) : "memory");
    return r;
}

// LDS the sort needs (a member of the caller's shared struct)
struct PcSortShared {
    int hist[PF_BINS];
    float red[6][PF_WAVES];
    int wsum[PF_WAVES];
    int giveup;          // a bounded poll of this workgroup gave up (or saw the scene's give-up word): everybody leaves
};

// A scene's give-up word: an 8-byte granule whose tag says "this launch"; raised by the first poll that gives up, read by
// the siblings' polls (they leave early instead of running into their own bounds) and by the launcher's follow-up launch.
struct PcGiveUp {
    unsigned long long *word;
    unsigned tag;
    unsigned limit;      // spin bound of every poll (PC_SPIN_FORCED: give up at once)
    __device__ __forceinline__ bool raised() const { return (unsigned)(granule_load(word) >> 32) == tag; }
    __device__ __forceinline__ void raise() const { granule_store(word, 1, tag); }
};
// spin until `ready()`; false when the bound was hit or the scene has been given up by a sibling
template <class Ready>
__device__ __forceinline__ bool pc_bounded_poll(const PcGiveUp &gu, int sleep, Ready ready) {
    if (gu.limit == PC_SPIN_FORCED) return false;
    unsigned spins = 0;
    while (!ready()) {
        if (sleep == 1) __builtin_amdgcn_s_sleep(1);
        else if (sleep == 2) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(4);
        if (++spins > gu.limit) return false;
        if ((spins & 1023u) == 0u && gu.raised()) return false;
    }
    return true;
}

// xg: the scene's exchange area (the cell histograms live behind its granules); sx .. srk: the scene's sorted arrays (npad entries each); hand_over: every workgroup
// waits until ALL of them have scattered (needed when they go on to read each other's points in the same launch).
// -> false (to every thread of the workgroup alike) when a poll gave up: the scene's give-up word is raised, nothing this
// workgroup wrote may be used, the caller returns.
// KMAX: compile-time bound of K (the histogram exchange keeps KMAX x 4 eight-byte loads in flight per thread)
template <int KMAX>
__device__ __forceinline__ bool pc_sort_split(PcSortShared &sh, int cu, int K, int n, int npad, int bs, int l2, int rb,
                                              const float *__restrict__ xyz, const float *__restrict__ temp,
                                              unsigned long long *xg, float *sx, float *sy, float *sz, float *st, int *srk,
                                              bool hand_over, const PcGiveUp &gu, unsigned long long *xflag = nullptr,
                                              unsigned tag0 = 2u, int flag_value = 1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) sh.giveup = 0;      // (several barriers lie between this store and the first poll)
    {
        // the two rounds of flags [2][PC_MAXK]: in the (zeroed) exchange area with tags 2 / 3, or wherever the caller keeps
        // them with a tag of its own (fps_presort.hip: a library-owned pool and a launch epoch -- nothing to zero per launch)
        if (!xflag) xflag = xg + PC_FLAG_AT;
        int *ghist = reinterpret_cast<int *>(xg + PC_GRANULES);
        const int s_beg = (int)((long long)n * cu / K), s_end = (int)((long long)n * (cu + 1) / K);
        auto flag_and_wait = [&](int slot, unsigned tg) {   // my stores -> visible; then wait for everybody's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                granule_store(xflag + slot * PC_MAXK + cu, 1, tg);
            }
            if (wave == 0) {
                bool ok = true;
                if (lane < K)
                    ok = pc_bounded_poll(gu, 4, [&] { return (unsigned)(granule_load(xflag + slot * PC_MAXK + lane) >> 32) == tg; });
                if (__ballot(!ok) != 0ull && lane == 0) { sh.giveup = 1; gu.raise(); }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();   // (this CU's vector L1 is invalidated: plain loads of the others' stores from here on)
        };
        // (A) bounding box of every K-th point, computed by every workgroup for itself (no exchange): the box only shapes the
        // sort cells -- keys are clamped into it, and no result depends on how good the sort is -- and a strided sample sees the
        // whole index range whatever order the cloud comes in
        float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
        const int nsample = (n + K - 1) / K;
        for (int k0 = tid; k0 < nsample; k0 += 8 * PF_THREADS) {
            float v[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * PF_THREADS;
                const size_t kk = (size_t)(k < nsample ? k : k0) * K;
#pragma unroll
                for (int a = 0; a < 3; ++a) v[u][a] = xyz[kk * 3 + a];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo3[a] = fminf(lo3[a], v[u][a]); hi3[a] = fmaxf(hi3[a], v[u][a]); }
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
        // (B) my histogram, then everybody's
        for (int k0 = s_beg + tid; k0 < s_end; k0 += 8 * PF_THREADS) {
            float v[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * PF_THREADS;
                const int kk = k < s_end ? k : k0;
#pragma unroll
                for (int a = 0; a < 3; ++a) v[u][a] = xyz[(size_t)kk * 3 + a];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u * PF_THREADS < s_end) atomicAdd(&sh.hist[pf_cell_key(grid, v[u][0], v[u][1], v[u][2])], 1);
        }
        __syncthreads();
        constexpr int PER = PF_BINS / PF_THREADS;
        // write-through (sc1) stores, drained by every wave, the workgroup's barrier, then the flag; the readers poll the K
        // flags and read with sc1 loads -- no L2 write-back / L1 invalidate on either side (MI355X_MICROARCH.md, hand-offs
        // measured with sc1 loads in place of the acquire)
        // (a thread's PER = 8 cells as four 8-byte stores, and below as 4 K 8-byte loads all in flight together: a dword
        //  at a time the exchange was a chain of K round trips, ~16 us of a 31 us pre-pass)
        static_assert(PER % 2 == 0, "cells are exchanged in pairs");
        unsigned long long *gh64 = reinterpret_cast<unsigned long long *>(ghist);
#pragma unroll
        for (int i = 0; i < PER / 2; ++i) {
            const unsigned lo = (unsigned)sh.hist[tid * PER + 2 * i], hi = (unsigned)sh.hist[tid * PER + 2 * i + 1];
            __hip_atomic_store(&gh64[((size_t)cu * PF_BINS + tid * PER) / 2 + i], ((unsigned long long)hi << 32) | lo,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) granule_store(xflag + cu, flag_value, tag0);   // (flag_value: whatever the caller wants its siblings to read)
        if (wave == 0) {
            bool ok = true;
            if (lane < K) ok = pc_bounded_poll(gu, 2, [&] { return (unsigned)(granule_load(xflag + lane) >> 32) == tag0; });
            if (__ballot(!ok) != 0ull && lane == 0) { sh.giveup = 1; gu.raise(); }
        }
        __syncthreads();
        if (sh.giveup) return false;   // (the histograms of a sibling that never arrived would be garbage offsets)
        {   // exclusive prefix sum over the cells of the summed histograms, plus what the workgroups before me put in each cell
            int loc[PER], before[PER], sum = 0;
            static_assert(KMAX <= PC_MAXK, "the exchange area holds PC_MAXK histograms");
            unsigned long long h64[KMAX][PER / 2];
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
#pragma unroll
                for (int i = 0; i < PER / 2; ++i)
                    h64[c][i] = c < K ? __hip_atomic_load(&gh64[((size_t)c * PF_BINS + tid * PER) / 2 + i], __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT)
                                      : 0ull;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int tot = 0, bef = 0;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    const int h = (int)(unsigned)(h64[c][i / 2] >> (32 * (i & 1)));
                    tot += h;
                    bef += c < cu ? h : 0;
                }
                loc[i] = tot; before[i] = bef; sum += tot;
            }
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
            for (int i = 0; i < PER; ++i) { sh.hist[tid * PER + i] = run + before[i]; run += loc[i]; }
        }
        __syncthreads();
        // (C) scatter my points
        for (int k0 = s_beg + tid; k0 < s_end; k0 += 8 * PF_THREADS) {
            float v[8][3], tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * PF_THREADS;
                const int kk = k < s_end ? k : k0;
#pragma unroll
                for (int a = 0; a < 3; ++a) v[u][a] = xyz[(size_t)kk * 3 + a];
                tv[u] = temp ? temp[kk] : 1e10f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * PF_THREADS;
                if (k < s_end) {
                    const int pos = atomicAdd(&sh.hist[pf_cell_key(grid, v[u][0], v[u][1], v[u][2])], 1);
                    sx[pos] = v[u][0]; sy[pos] = v[u][1]; sz[pos] = v[u][2]; st[pos] = tv[u];
                    srk[pos] = (int)pf_rank((unsigned)k, bs, l2, rb);
                }
            }
        }
        if (cu == 0)
            for (int p = n + tid; p < npad; p += PF_THREADS) {  // padding: never inside a box, distance stays -1, worst rank
                sx[p] = NAN; sy[p] = NAN; sz[p] = NAN; st[p] = -1.f; srk[p] = 0x0FFFFFFF;
            }
        if (hand_over) {
            flag_and_wait(1, tag0 + 1u);
            if (sh.giveup) return false;
        }
    }
    return true;
}

}  // namespace sps

