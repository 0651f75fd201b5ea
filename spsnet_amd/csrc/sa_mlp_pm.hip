// sa_mlp_pm.hip -- the exact-fp32 "group -> shared MLP -> max-pool" kernel of sa_mlp.hip for layers whose input features
// come with a point-major twin (B, N, C) (every SA layer behind the first: the aggregation kernel writes one, pw_mlp.hip).
//
// Why a second fp32 kernel.  sa_mlp.hip gathers layer 1's input one channel per lane and k-step from the reference's
// channel-major (B, C, N) tensor: 64 scattered dwords per instruction, each k-step's gather requested one step (16-32 MFMAs
// = 0.2 us) ahead of its use while a gather takes ~1 us to come back -- with one or two waves per SIMD nobody hides that, and
// layer 1 (15-25 % of the arithmetic) cost as much as layers 2 and 3 together: the 131-128-256-256 scale ran the fp32 matrix
// pipe at 60 % (profiles/round2/pmc_mfma.json), the 67-64-64-128 scale at a third.  Here
//   * a lane's four channels 16 t + 4 q .. + 3 of a grouped point are ONE 16-byte load from the point-major twin -- exactly
//     the B-operand layout the chained layers use (k-step (t, r) <-> channel 16 t + 4 q + r), so layer 1 over the feature
//     channels is just another chained layer whose "activations" are the loaded registers, plus one k-step for the three
//     centred coordinates;
//   * ALL of a unit's inputs (C / 16 float4 + 1 coordinate per lane and column tile) are requested while the PREVIOUS unit's
//     layer 3 runs (their indices while its layer 2 runs): one exposed gather latency per unit instead of one per k-step;
//   * layer 1's weights stream in the same double-buffered dwordx4 chunks as the other layers'.
// Same arithmetic as sa_mlp.hip -- v_mfma_f32_16x16x4_f32, bias as the C operand, ReLU / max-pool as integer maxima -- but
// layer 1 accumulates its k-steps in a different ORDER (coordinates first, then the features in the twin's order), so its
// results differ from sa_mlp.hip's in the last bits, like any two fp32 GEMM schedules; both are within 1e-4 of torch.
#include "sps_common.h"
#include "sa_mlp_args.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 wload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return (f32x4){__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
}

// One chained layer: out tile mt = bias + sum over the CIN input channels.  Fragments packed [mt][CIN/16][lane][4] (one
// dwordx4 per lane = k-steps (t, 0..3)) -- i.e. LINEAR in the chunk index c = mt * NCH + ch -- and streamed through a ring of
// RING register buffers, RING - 1 chunks ahead of the multiplies: one chunk ahead (1024 cycles of MFMAs at KCH = 16) is less
// than an L2 round trip under load, and with one wave per SIMD nobody else covers the difference.
// pre(mt, acc): called once per output tile before the chain (adds the coordinate k-step in layer 1); sink(mt, acc): the
// finished accumulators.  Fully unrolled: `mt` is a compile-time constant for pre / sink (they index register arrays).
constexpr int RING = 4;

template <int CIN>
struct ChainShape {
    static constexpr int KS = CIN / 4;
    static constexpr int KCH = (KS % 16 == 0) ? 16 : ((KS % 8 == 0) ? 8 : 4);
    static constexpr int NCH = KS / KCH;
    static constexpr int Q4 = KCH / 4;
};

// BRIDGED use (all layers of a kernel with Q4 == 4 and a chunk count that is a multiple of RING): the ring lives in the caller
// and runs on across layer and unit boundaries -- PRIMED: this layer's first RING - 1 chunks were requested by the layer
// before it (or the prime in front of the unit loop); NEXT: while this layer multiplies its last chunks it requests the first
// RING - 1 chunks of the layer behind it (`wnext`: that layer's fragments; the last layer hands on to layer 1 of the next
// unit).  Without it every layer began with an exposed L2 round trip, three per unit.  The next tile's bias is read from LDS
// while the current tile multiplies (its lgkmcnt wait otherwise sat in front of every tile's first MFMA).
typedef f32x4 WeightRing[RING][4];

template <int CIN, int MT, int NT, bool PRIMED, bool NEXT, class Pre, class Sink>
__device__ __forceinline__ void chain_layer(const float *w, const float *bias, int lane, int q, const f32x4 (&hin)[CIN / 16][NT],
                                            WeightRing &wb, const float *wnext, int next_chunks, Pre pre, Sink sink) {
    using S = ChainShape<CIN>;
    constexpr int KCH = S::KCH, NCH = S::NCH, Q4 = S::Q4, G = MT * NCH;
    static_assert(!(PRIMED || NEXT) || (Q4 == 4 && G % RING == 0), "a bridged layer keeps the ring's phase");
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc(w, (unsigned)(MT * S::KS * 64 * 4));
    const __amdgpu_buffer_rsrc_t rsn = weight_rsrc(wnext, (unsigned)(next_chunks * 4 * 1024));
    if constexpr (!PRIMED) {
#pragma unroll
        for (int p = 0; p < RING - 1; ++p)
            if (p < G) {
#pragma unroll
                for (int u = 0; u < Q4; ++u) wb[p][u] = wload4(rs, lane * 16, (p * Q4 + u) * 1024);
            }
    }
    // A tile's epilogue (ReLU, hand-over) is issued BEHIND the first chunk of the next tile's MFMAs: a wave issues in order, so
    // an epilogue placed right behind its own tile waits for the matrix pipe to drain (its VALU ops read the accumulators)
    // and the pipe then idles until the next tile's first MFMA -- 40 such bubbles per unit at the widest scale.  Two
    // accumulator sets alternate by tile parity.
    f32x4 acc[2][NT];
    f32x4 bnext = *reinterpret_cast<const f32x4 *>(bias + 4 * q);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int mt = g / NCH, ch = g % NCH, par = mt & 1;
        if (g + RING - 1 < G) {
#pragma unroll
            for (int u = 0; u < Q4; ++u) wb[(g + RING - 1) % RING][u] = wload4(rs, lane * 16, ((g + RING - 1) * Q4 + u) * 1024);
        } else if constexpr (NEXT) {
#pragma unroll
            for (int u = 0; u < 4; ++u) wb[(g + RING - 1) % RING][u] = wload4(rsn, lane * 16, ((g + RING - 1 - G) * 4 + u) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[par][nt] = bnext;
            pre(mt, acc[par]);
            if (mt + 1 < MT) bnext = *reinterpret_cast<const f32x4 *>(bias + 16 * (mt + 1) + 4 * q);
        }
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk) {
            const int ks = ch * KCH + kk;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[par][nt] = mfma16(wb[g % RING][kk / 4][kk % 4], hin[ks / 4][nt][ks % 4], acc[par][nt]);
        }
        if (ch == 0 && mt > 0) sink(mt - 1, acc[par ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    sink(MT - 1, acc[(MT - 1) & 1]);
}

// The same with a RUNTIME loop over groups of U output tiles (the last layer: its sink needs no compile-time tile index, and
// 64 unrolled tile x chunk blocks exceed what the unroller accepts -- it then unrolls partially and the INPUT tiles, indexed by
// a no longer constant k-step, land in scratch).  U tiles = U * NCH chunks = a whole number of turns of the ring, so the
// buffer a chunk lands in is a compile-time constant inside the unrolled group; the requests run RING - 1 chunks ahead across
// tile and group boundaries (the chunk index is linear in memory).
template <int CIN, int MT, int NT, bool PRIMED, bool NEXT, class Sink>
__device__ __forceinline__ void chain_layer_rt(const float *w, const float *bias, int lane, int q, const f32x4 (&hin)[CIN / 16][NT],
                                               WeightRing &wb, const float *wnext, int next_chunks, Sink sink) {
    using S = ChainShape<CIN>;
    constexpr int KCH = S::KCH, NCH = S::NCH, Q4 = S::Q4, G = MT * NCH;
    constexpr int U = (NCH % RING == 0) ? 1 : ((2 * NCH) % RING == 0 ? 2 : RING);   // tiles per unrolled group
    static_assert((U * NCH) % RING == 0 && MT % U == 0, "a group of tiles is a whole number of ring turns");
    static_assert(!(PRIMED || NEXT) || Q4 == 4, "a bridged layer keeps the ring's phase");
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc(w, (unsigned)(MT * S::KS * 64 * 4));
    const __amdgpu_buffer_rsrc_t rsn = weight_rsrc(wnext, (unsigned)(next_chunks * 4 * 1024));
    if constexpr (!PRIMED) {
#pragma unroll
        for (int p = 0; p < RING - 1; ++p)
            if (p < G) {
#pragma unroll
                for (int u = 0; u < Q4; ++u) wb[p][u] = wload4(rs, lane * 16, (p * Q4 + u) * 1024);
            }
    }
    // (the epilogue of tile mt - 1 -- pool, stores -- is issued behind the first chunk of tile mt's MFMAs, see chain_layer)
    f32x4 accp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accp[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bnext = *reinterpret_cast<const f32x4 *>(bias + 4 * q);
    for (int mt0 = 0; mt0 < MT; mt0 += U) {
        const int c0 = mt0 * NCH;                     // first chunk of the group (wave-uniform)
#pragma unroll
        for (int tu = 0; tu < U; ++tu) {
            const int mt = mt0 + tu;
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = bnext;
            {
                const int mtn = mt + 1 < MT ? mt + 1 : mt;
                bnext = *reinterpret_cast<const f32x4 *>(bias + 16 * mtn + 4 * q);
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int cl = tu * NCH + ch;           // chunk within the group: compile-time
                int cn = c0 + cl + RING - 1;            // the chunk to request now
                if (NEXT && cn >= G) {                  // (wave-uniform; the last RING - 1 chunks of the layer only)
#pragma unroll
                    for (int u = 0; u < 4; ++u) wb[(cl + RING - 1) % RING][u] = wload4(rsn, lane * 16, ((cn - G) * 4 + u) * 1024);
                } else {
                    cn = cn < G ? cn : G - 1;           // (unbridged: the last ones re-read the last chunk)
#pragma unroll
                    for (int u = 0; u < Q4; ++u) wb[(cl + RING - 1) % RING][u] = wload4(rs, lane * 16, (cn * Q4 + u) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < KCH; ++kk) {
                    const int ks = ch * KCH + kk;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wb[cl % RING][kk / 4][kk % 4], hin[ks / 4][nt][ks % 4], acc[nt]);
                }
                if (ch == 0 && mt > 0) sink(mt - 1, accp);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) accp[nt] = acc[nt];
        }
    }
    sink(MT - 1, accp);
}

}  // namespace

// CF: feature channels of the point-major twin (a multiple of 16); C1, C2, C3: padded widths; NT: 16-column tiles per unit.
// Weights: a.w1 = [C1/16 tiles][lane] coordinate fragments (k-slot q < 3 = centred x, y, z; slot 3 = zero), then layer 1 over
// the features as [tile][CF/16][lane][4]; a.w2, a.w3 as in sa_mlp.hip (fused._pack_next).
// PACKED: the columns come from sps_pack_columns (only the distinct neighbours of every ball, in power-of-two slots): a unit is
// COLS consecutive entries of a.cols / a.meta, the pool is segmented per slot (pool_write_packed) -- same pooled values bit for bit.
// HOIST1: a.feat is (B, N, C1) = b1 + W1f . features[point] (point_layer1_pm_kernel below): per grouped point layer 1 is then
// ONE k-step (the centred coordinates) on top of the gathered row instead of CF / 4 + 1 -- a grouped MLP multiplies every
// point's features by the same W1f once per ball the point falls into (nsample M / N times: 16 at IA-SSD layer 2).  The sums
// are formed in a different order than without it (features first, per point; then the coordinates), like any two fp32 GEMM
// schedules; both are within 1e-4 of torch.
template <int CF, int C1, int C2, int C3, int NT, int NS, bool PACKED, bool HOIST1 = false>
__global__ __launch_bounds__(256) void sa_group_mlp_pm_kernel(SaMlpArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    static_assert(!HOIST1 || CF == C1, "the hoisted form gathers C1 channels per point into the registers of CF");
    constexpr int T0 = CF / 16, T1 = C1 / 16, T2 = C2 / 16, MT3 = C3 / 16;
    constexpr int COLS = 16 * NT;
    constexpr int CPP = COLS >= NS ? COLS / NS : 1;   // whole centroids per unit ...
    constexpr bool PART = COLS < NS;                  // ... or a unit is a slice of one centroid's samples (nsample 64)
    static_assert((COLS % NS == 0 || NS % COLS == 0) && (NS % 16) == 0 && CF % 16 == 0, "units and centroids must nest");

    // Biases (and layer 1's coordinate fragments) live in LDS for the whole launch: read from global memory where a tile
    // begins, the load drew an `s_waitcnt vmcnt(0)` that also waited for the weight chunk requested just before it -- a full
    // memory round trip per output tile, 40 per unit at the widest scale, which is what held sa_mlp.hip at 60 % of the pipe.
    // LDS reads count on lgkmcnt and leave the weight stream's vmcnt alone.
    __shared__ __attribute__((aligned(16))) float sbias[C1 + C2 + C3];
    __shared__ float swx[T1 * 64];
    for (int i = threadIdx.x; i < C1; i += blockDim.x) sbias[i] = a.b1[i];
    for (int i = threadIdx.x; i < C2; i += blockDim.x) sbias[C1 + i] = a.b2[i];
    for (int i = threadIdx.x; i < C3; i += blockDim.x) sbias[C1 + C2 + i] = a.b3[i];
    for (int i = threadIdx.x; i < T1 * 64; i += blockDim.x) swx[i] = a.w1[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const MlpRange rg = mlp_range(a);
    const int nunits = PACKED ? (*a.ntiles) / NT : rg.units;   // packed: as many units as pack_columns produced tiles for
    const bool merge = PACKED && a.merge_max && !flag_or_any(a.merge_unless, a.merge_unless_any, a.merge_unless_count);
    const float *w1f = a.w1 + (size_t)T1 * 64;   // layer 1 over the features, behind the coordinate fragments

    // a unit's inputs: requested one unit ahead
    f32x4 xin[T0][NT];
    float xq[NT];
    int src[NT];
    PackedUnit<NT> pun, pu;      // meta words of the unit being requested / being computed (PACKED)
    auto col0_of = [&](int unit, int &ub) -> long long {
        ub = unit / rg.ups;
        return ((long long)ub * a.m + rg.j0) * NS + (long long)(unit - ub * rg.ups) * COLS;
    };
    auto load_idx = [&](int unit) {
        if constexpr (PACKED) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const size_t e = (size_t)unit * COLS + nt * 16 + c;
                src[nt] = a.cols[e];
                pun.w[nt] = a.meta[e];
            }
        } else {
            int ub;
            const long long col0 = col0_of(unit, ub);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) src[nt] = a.idx[col0 + nt * 16 + c];
        }
    };
    auto load_inputs = [&](int unit) {   // (uses src[] / pun of the same unit)
        int ub = 0;
        long long col0 = 0;
        if constexpr (!PACKED) col0 = col0_of(unit, ub);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            long long bj;
            if constexpr (PACKED) {   // a tile never straddles scenes: the scene stays wave-uniform (scalar address arithmetic)
                ub = __builtin_amdgcn_readfirstlane((int)((pun.w[nt] >> 20) & 0xFFu));
                bj = (long long)ub * a.m + (int)(pun.w[nt] & 0xFFFFFu);
            } else {
                bj = (col0 + nt * 16 + c) / NS;
            }
            const size_t pt = (size_t)ub * a.n + src[nt];
            const float *row = a.feat + pt * CF + 4 * q;
#pragma unroll
            for (int t = 0; t < T0; ++t) xin[t][nt] = *reinterpret_cast<const f32x4 *>(row + 16 * t);
            const int ax = q < 3 ? q : 0;
            const float d = a.xyz[pt * 3 + ax] - a.new_xyz[(size_t)bj * 3 + ax];
            xq[nt] = q < 3 ? d : 0.f;
        }
    };
    // The widest scales (IA-SSD layer 5: 256 feature channels, up to 512 hidden) hold a unit's activations in 384 of the 512
    // registers a wave owns, so the next unit's inputs cannot travel beside them; a unit is 8-22 k MFMAs there (120-340 us),
    // the one exposed gather at its head is noise.
    constexpr bool PREFETCH = CF <= 128 && C2 <= 256;
    if (PREFETCH && wave < nunits) {
        load_idx(wave);
        load_inputs(wave);
    }
    // one weight ring for the whole kernel when every layer keeps its phase (all IA-SSD widths but the 96-wide one)
    using S0 = ChainShape<CF>; using S1 = ChainShape<C1>; using S2 = ChainShape<C2>;
    constexpr bool BRIDGE = (HOIST1 || (S0::Q4 == 4 && (T1 * S0::NCH) % RING == 0)) && S1::Q4 == 4 && S2::Q4 == 4 &&
                            (T2 * S1::NCH) % RING == 0 && (MT3 * S2::NCH) % RING == 0;
    // the layer that follows layer 3 in the ring: layer 1's feature chain, or (HOIST1: there is none) layer 2
    const float *wfirst = HOIST1 ? a.w2 : w1f;
    constexpr int FIRST_CHUNKS = HOIST1 ? T2 * S1::NCH : T1 * S0::NCH;
    WeightRing wb;
    if constexpr (BRIDGE) {
        const __amdgpu_buffer_rsrc_t rs1 = weight_rsrc(wfirst, (unsigned)(FIRST_CHUNKS * 4 * 1024));
#pragma unroll
        for (int p = 0; p < RING - 1; ++p)
#pragma unroll
            for (int u = 0; u < 4; ++u) wb[p][u] = wload4(rs1, lane * 16, (p * 4 + u) * 1024);
    }
    for (int unit = wave; unit < nunits; unit += nwaves) {
        int ub = 0;
        long long col0 = 0;
        if constexpr (!PACKED) col0 = col0_of(unit, ub);
        if constexpr (!PREFETCH) {
            load_idx(unit);
            load_inputs(unit);
        }
        if constexpr (PACKED) pu = pun;
        const int nxt = unit + nwaves;
        const bool more = PREFETCH && nxt < nunits;
        // ---------------- layer 1: the coordinate k-step, then the chain over the twin's channels ----------------
        f32x4 h1[T1][NT];
        {
            float wx[T1];
#pragma unroll
            for (int t = 0; t < T1; ++t) wx[t] = swx[t * 64 + lane];
            if constexpr (HOIST1) {
#pragma unroll
                for (int t = 0; t < T1; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x4 acc = mfma16(wx[t], xq[nt], xin[t][nt]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) h1[t][nt][r] = relu_keep_nan(acc[r]);
                    }
            } else {
            chain_layer<CF, T1, NT, BRIDGE, BRIDGE>(w1f, sbias, lane, q, xin, wb, a.w2, T2 * S1::NCH,
                [&](int mt, f32x4 (&acc)[NT]) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wx[mt], xq[nt], acc[nt]);
                },
                [&](int mt, f32x4 (&acc)[NT]) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h1[mt][nt][r] = relu_keep_nan(acc[nt][r]);
                });
            }
        }
        // the next unit's neighbour indices travel while layer 2 runs ...
        if (more) load_idx(nxt);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 h2[T2][NT];
        // (at most 16 output tiles per fully unrolled call: beyond ~64 tile x chunk blocks the unroller gives up and the
        // activations, then indexed by a run-time tile, would live in scratch)
        constexpr int H2 = T2 > 16 ? T2 / 2 : T2;
        static_assert(T2 == H2 || (T2 == 2 * H2 && (H2 * S1::NCH) % RING == 0), "layer 2 in two halves keeps the ring's phase");
#pragma unroll
        for (int half = 0; half < T2 / H2; ++half) {
            const bool last = half + 1 == T2 / H2;
            const float *wh = a.w2 + (size_t)half * H2 * S1::KS * 64;
            chain_layer<C1, H2, NT, BRIDGE, BRIDGE>(wh, sbias + C1 + 16 * H2 * half, lane, q, h1, wb,
                last ? a.w3 : wh + (size_t)H2 * S1::KS * 64, last ? MT3 * S2::NCH : H2 * S1::NCH, [](int, f32x4 (&)[NT]) {},
                [&](int mt, f32x4 (&acc)[NT]) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h2[half * H2 + mt][nt][r] = relu_keep_nan(acc[nt][r]);
                });
        }
        // ... and its inputs while layer 3 does (loads return in order: the first weight wait behind this point also waits
        // for these gathers -- once per unit)
        if (more) load_inputs(nxt);
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- layer 3 + max-pool over the unit's columns ----------------
        const long long bj0 = col0 / NS;
        chain_layer_rt<C2, MT3, NT, BRIDGE, BRIDGE>(a.w3, sbias + C1 + C2, lane, q, h2, wb, wfirst, FIRST_CHUNKS,
            [&](int mt, f32x4 (&acc)[NT]) {
                if constexpr (PACKED) {
                    pool_write_packed<NT>(a, acc, pu, mt, q, c, false, merge);
                    return;
                }
                f32x4 best[CPP];
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) best[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cc = PART ? 0 : (nt * 16) / NS;
#pragma unroll
                    for (int r = 0; r < 4; ++r) best[cc][r] = imaxf(best[cc][r], acc[nt][r]);   // ReLU + pool: one max from +0
                }
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) {
                    const f32x4 pooled4 = row_allmax4i(best[cc]);
                    if (c == 0) {
                        const long long cen = bj0 + cc;
                        const int j = (int)(cen - (long long)ub * a.m);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * mt + 4 * q + r;
                            if (row < a.c3_real) {
                                float *dst = a.out_pm ? a.out + ((size_t)ub * a.m + j) * a.out_c_total + a.out_c_off + row
                                                      : a.out + ((size_t)ub * a.out_c_total + a.out_c_off + row) * a.m + j;
                                if constexpr (PART) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(pooled4[r]));
                                else *dst = pooled4[r];
                            }
                        }
                    }
                }
            });
    }
}

// b1 + W1f . features[point] for every point of the (B, N, CF) twin -> out (B N, C1) point-major: what HOIST1 gathers.  The same
// chained layer on the same packed fragments, over 16 consecutive points per wave instead of 16 grouped ones.
template <int CF, int C1>
__global__ __launch_bounds__(256) void point_layer1_pm_kernel(int npts, const float *__restrict__ feat, const float *__restrict__ w1,
                                                              const float *__restrict__ b1, float *__restrict__ out) {
    constexpr int T0 = CF / 16, T1 = C1 / 16;
    __shared__ __attribute__((aligned(16))) float sbias[C1];
    for (int i = threadIdx.x; i < C1; i += blockDim.x) sbias[i] = b1[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const float *w1f = w1 + (size_t)T1 * 64;   // (behind the coordinate fragments)
    WeightRing wb;
    for (int tile = wave; tile * 16 < npts; tile += nwaves) {
        const int pt = tile * 16 + c;
        const int ptc = pt < npts ? pt : npts - 1;
        f32x4 xin[T0][1];
#pragma unroll
        for (int t = 0; t < T0; ++t) xin[t][0] = *reinterpret_cast<const f32x4 *>(feat + (size_t)ptc * CF + 16 * t + 4 * q);
        chain_layer<CF, T1, 1, false, false>(w1f, sbias, lane, q, xin, wb, w1f, 0, [](int, f32x4 (&)[1]) {},
            [&](int mt, f32x4 (&acc)[1]) {
                if (pt < npts) *reinterpret_cast<f32x4 *>(out + (size_t)pt * C1 + 16 * mt + 4 * q) = acc[0];
            });
    }
}

int launch_point_layer1_pm(int npts, int c_feat, int c1, const float *feat, const float *w1, const float *b1, float *out, hipStream_t st) {
    const int tiles = divup(npts, 16);
    int blocks = divup(tiles, 4);
    if (blocks > 1024) blocks = 1024;
#define SPS_PL1_CASE(CF, C1) \
    if (c_feat == CF && c1 == C1) { \
        hipLaunchKernelGGL((point_layer1_pm_kernel<CF, C1>), dim3(blocks), dim3(256), 0, st, npts, feat, w1, b1, out); \
        return check_launch("point_layer1_pm_kernel"); \
    }
    SPS_PL1_CASE(64, 64)
    SPS_PL1_CASE(128, 128)
    SPS_PL1_CASE(256, 256)
#undef SPS_PL1_CASE
    return fail(SPS_ERR_INVALID, "sa_layer1_per_point: no kernel for %d feature channels -> %d", c_feat, c1);
}

template <int CF, int C1, int C2, int C3, int NT, int NS>
static int launch_pm_variant(const SaMlpArgs &a, hipStream_t st) {
    constexpr int UNIT = 16 * NT;
    SaMlpArgs k = a;
    if (a.cols) {                      // packed columns: `units` = tile capacity on entry; the kernel reads the real count
        k.ups = 1;
        k.units = a.units / NT;
    } else {
        const long long cols_scene = (long long)a.ups * NS;   // caller passes centroids per scene in `ups`, scenes in `units`
        if (cols_scene % UNIT != 0)
            return fail(SPS_ERR_INVALID, "sa_group_mlp(pm): centroids*nsample per scene (%lld) not a multiple of %d", cols_scene, UNIT);
        k.ups = (int)(cols_scene / UNIT);
        k.units = a.units * k.ups;
        k.alt_j0 = 0;
        k.alt_ups = (int)((long long)a.m * NS / UNIT);
        k.alt_units = a.units * k.alt_ups;
        if (a.alt && ((long long)a.m * NS) % UNIT != 0)
            return fail(SPS_ERR_INVALID, "sa_group_mlp(pm): centroids*nsample per scene not a multiple of %d", UNIT);
    }
    const int waves_per_block = 4;
    // as many workgroups as the chip holds at this kernel's occupancy, dealt evenly: every wave walks the same number of
    // units (+-1), each of which prefetches the next one's inputs
    static std::atomic<int> occ_cached{0};     // (per instantiation; the same on every device of the node; racing callers
    int occ = occ_cached.load(std::memory_order_relaxed);   //  compute and store the same value)
    if (occ == 0) {
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, sa_group_mlp_pm_kernel<CF, C1, C2, C3, NT, NS, false>, 64 * waves_per_block, 0)
                != hipSuccess || o < 1)
            o = 1;
        occ = o > 4 ? 4 : o;
        occ_cached.store(occ, std::memory_order_relaxed);
    }
    int blocks = divup(k.units, waves_per_block);
    int max_blocks = device_cu_count() * occ;
    if (const char *e = getenv("SPS_MLP_PM_BLOCKS")) {   // DIAGNOSTIC (tools/mlp_time.py): how the launch scales with the CUs it uses
        const int v = atoi(e);
        if (v > 0) max_blocks = v;
    }
    if (blocks > max_blocks) blocks = max_blocks;
    if constexpr (CF == C1 && NS <= 32) {
        if (a.hoist1) {
            if (a.cols)
                hipLaunchKernelGGL((sa_group_mlp_pm_kernel<CF, C1, C2, C3, NT, NS, true, true>), dim3(blocks), dim3(64 * waves_per_block), 0, st, k);
            else
                hipLaunchKernelGGL((sa_group_mlp_pm_kernel<CF, C1, C2, C3, NT, NS, false, true>), dim3(blocks), dim3(64 * waves_per_block), 0, st, k);
            return check_launch("sa_group_mlp_pm_kernel<hoisted layer 1>");
        }
    }
    if (a.hoist1) return fail(SPS_ERR_INVALID, "sa_group_mlp(pm): no per-point layer-1 form for these widths / nsample %d", NS);
    if (a.cols)
        hipLaunchKernelGGL((sa_group_mlp_pm_kernel<CF, C1, C2, C3, NT, NS, true>), dim3(blocks), dim3(64 * waves_per_block), 0, st, k);
    else
        hipLaunchKernelGGL((sa_group_mlp_pm_kernel<CF, C1, C2, C3, NT, NS, false>), dim3(blocks), dim3(64 * waves_per_block), 0, st, k);
    return check_launch("sa_group_mlp_pm_kernel");
}

// arith 0 + point-major features (mode 4 of sps_sa_group_mlp_packed)
int launch_sa_mlp_pm(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st) {
#define SPS_MLPPM_CASE(CF, C1, C2, C3, NT, NS) \
    if (a.c_feat == CF && c1 == C1 && c2 == C2 && a.c3 == C3 && nsample == NS) return launch_pm_variant<CF, C1, C2, C3, NT, NS>(a, st);
    SPS_MLPPM_CASE(64, 64, 64, 128, 2, 16)      // IA-SSD L1 [67,64,64,128]
    SPS_MLPPM_CASE(64, 64, 96, 128, 2, 32)      // L1 [67,64,96,128]
    SPS_MLPPM_CASE(128, 128, 128, 256, 2, 16)   // L2 [131,128,128,256]
    SPS_MLPPM_CASE(128, 128, 256, 256, 2, 32)   // L2 [131,128,256,256]
    SPS_MLPPM_CASE(256, 256, 256, 512, 2, 16)   // IA-SSD L5 (vote centres) [259,256,256,512]
    SPS_MLPPM_CASE(256, 256, 512, 1024, 2, 32)  // L5 [259,256,512,1024]
    SPS_MLPPM_CASE(64, 64, 64, 128, 2, 64)      // nsample 64: a centroid spans two units, atomic max onto zeros
    SPS_MLPPM_CASE(64, 64, 96, 128, 2, 64)
    SPS_MLPPM_CASE(128, 128, 128, 256, 2, 64)
    SPS_MLPPM_CASE(128, 128, 256, 256, 2, 64)
#undef SPS_MLPPM_CASE
    return fail(SPS_ERR_INVALID, "sa_group_mlp(pm): no kernel for %d feature channels, widths (%d, %d, %d), nsample %d", a.c_feat, c1,
                c2, a.c3, nsample);
}

}  // namespace sps

// 1 if the point-major fp32 kernel (mode 4 of sps_sa_group_mlp_packed) serves these feature channels / padded widths
extern "C" int sps_sa_group_mlp_pm_supported(int c_feat, int c1, int c2, int c3, int nsample) {
    static const int tab[][5] = {{64, 64, 64, 128, 16}, {64, 64, 96, 128, 32}, {128, 128, 128, 256, 16}, {128, 128, 256, 256, 32},
                                 {256, 256, 256, 512, 16}, {256, 256, 512, 1024, 32},
                                 {64, 64, 64, 128, 64}, {64, 64, 96, 128, 64}, {128, 128, 128, 256, 64}, {128, 128, 256, 256, 64}};
    for (auto &t : tab)
        if (t[0] == c_feat && t[1] == c1 && t[2] == c2 && t[3] == c3 && t[4] == nsample) return 1;
    return 0;
}

// b1 + W1f . features[point] of the exact-fp32 point-major kernel's layer 1, once per point: features_pm (npts, c_feat) ->
// out (npts, c1); w1 / b1 as packed for sps_sa_group_mlp mode 4 (fused._pack_first_pm).  The grouped launch then takes `out` as
// its feature tensor with mode bit 32.
extern "C" int sps_sa_layer1_per_point(int npts, int c_feat, int c1, const float *features_pm, const float *w1, const float *b1,
                                       float *out, sps_stream_t stream) {
    using namespace sps;
    if (npts < 0 || !features_pm || !w1 || !b1 || !out) return fail(SPS_ERR_INVALID, "sa_layer1_per_point: bad arguments");
    if (npts == 0) return SPS_OK;
    return launch_point_layer1_pm(npts, c_feat, c1, features_pm, w1, b1, out, as_stream(stream));
}

// 1 if mode bit 32 (layer 1's feature product per point) is served for these widths
extern "C" int sps_sa_layer1_per_point_supported(int c_feat, int c1, int nsample) {
    return ((c_feat == 64 && c1 == 64) || (c_feat == 128 && c1 == 128) || (c_feat == 256 && c1 == 256)) && nsample <= 32;
}
