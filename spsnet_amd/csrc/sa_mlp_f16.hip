// sa_mlp_f16.hip -- the fused group -> shared MLP -> max-pool kernel of sa_mlp.hip on the fp16 matrix cores,
// with every fp32 operand carried as a SPLIT pair of halves (x = hi + lo, hi = fp16(x), lo = fp16(x - hi)) and
// three MFMAs per product block (hi*hi + hi*lo + lo*hi, fp32 accumulate).  The pair holds ~22 significant bits,
// the dropped lo*lo term is 2^-22 relative, so features stay within ~1e-6 of the fp32 path -- far inside the
// 1e-4 bar of BASELINE.json -- while v_mfma_f32_16x16x16_f16 runs at 16x the rate of the fp32 MFMA (which on
// gfx950 is no faster than the VALU).  |x| is clamped to the fp16 range before the split, so an out-of-range
// activation saturates (and is reported) instead of producing inf/NaN.
//
// Layout: v_mfma_f32_16x16x32_f16 wants 8 consecutive k per lane.  The D tiles of layer l (lane (q,c): rows 4q..4q+3 of
// column c) of TWO consecutive 16-row tiles form the B operand of one k32-step of layer l+1 if k-slot (q, j) of step s
// stands for channel 32 s + 16 (j / 4) + 4 q + j % 4 -- activations chain register to register exactly as in the fp32
// kernel, the host packs the weights in that order: [tile][k32][hi | lo][lane][8 halves], 2 KiB per fragment
// (fused._pack_f16).  (The K = 16 instruction this file first used issues at a quarter of the rate, profiles/round1/microbench_*.txt.)
#include "sps_common.h"
#include "sa_mlp_args.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// K = 32: the K = 16 form issues at 32 cycles on gfx950, this one at 16 for twice the work (profiles/round2/microbench_mfma_gfx950.txt)
__device__ __forceinline__ f32x4 mfma32h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Split 4 fp32 values into halves OFF..OFF+3 of the hi / lo operand vectors.  The value is clamped to the fp16 range
// first (RELU: to [0, 65504], which is the layer's ReLU as well), so hi is finite and lo = fp16(c - hi) is tiny: no
// inf/NaN can arise.  |x| <= 65504 splits exactly to 22 bits; `mx` tracks the largest magnitude seen so that the
// launch can report operands beyond that (fused.check_overflow) -- never silent.
template <int OFF, bool RELU, bool PURE = false>
__device__ __forceinline__ void split4(const f32x4 v, h8 &hi, h8 &lo, float &mx) {
    mx = __builtin_amdgcn_fmed3f(mx, INFINITY, fmaxf(fabsf(v[0]), fabsf(v[1])));  // max3(mx, |v0|, |v1|) for mx >= 0
    mx = __builtin_amdgcn_fmed3f(mx, INFINITY, fmaxf(fabsf(v[2]), fabsf(v[3])));
    if constexpr (!RELU) {
        // layer-1 inputs come from outside: v_max drops a NaN, so NaN / Inf are caught by 0 * v (NaN for both), which sends
        // mx to +Inf.  Activations (RELU) can only turn NaN after an Inf that mx has seen.
        const float z = __builtin_fmaf(v[0], 0.f, __builtin_fmaf(v[1], 0.f, __builtin_fmaf(v[2], 0.f, v[3] * 0.f)));
        mx = (z == 0.f) ? mx : INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float c = __builtin_amdgcn_fmed3f(v[r], RELU ? 0.f : -65504.f, 65504.f);
        const _Float16 h = (_Float16)c;
        hi[OFF + r] = h;
        if constexpr (!PURE) lo[OFF + r] = (_Float16)(c - (float)h);  // PURE (fp16 features): operands ARE halves, no lo part
    }
}

// a fragment = (16-row tile, 32-channel step): split mode 2 KiB = [lane][hi x8] then [lane][lo x8]; pure-fp16 mode 1 KiB =
// [lane][8 halves] (the weights themselves are rounded to fp16 by the host, fused._pack_h16)
struct WFrag { h8 hi, lo; };
template <bool PURE> constexpr int frag_bytes() { return PURE ? 1024 : 2048; }
template <bool PURE>
__device__ __forceinline__ WFrag wload_h(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    union { i32x4 i; h8 h; } a, b;
    a.i = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    if constexpr (PURE) return WFrag{a.h, a.h};
    b.i = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + 1024, 0);
    return WFrag{a.h, b.h};
}
// Weight source of a layer: a buffer resource (fragments streamed from L2 by every wave) or, when all three layers fit
// (WLDS: the 64-wide scales, <= 96 KiB), a copy in LDS made once per workgroup -- the L1 scales moved 805 MB of
// weights per launch through L2 (8.4 TB/s: that bandwidth, not the matrix cores, bounded them).
template <bool WLDS, bool PURE>
struct WSrc {
    __amdgpu_buffer_rsrc_t rs;
    unsigned lds;  // byte address of the layer's first fragment in LDS
    __device__ __forceinline__ WFrag load(int lane16, int soff) const {
        if constexpr (WLDS) {
            typedef const __attribute__((address_space(3))) i32x4 lds_v;
            lds_v *p = (lds_v *)(size_t)(lds + (unsigned)soff + (unsigned)lane16);
            union { i32x4 i; h8 h; } a, b;
            a.i = p[0];
            if constexpr (PURE) return WFrag{a.h, a.h};
            b.i = p[64];
            return WFrag{a.h, b.h};
        } else {
            return wload_h<PURE>(rs, lane16, soff);
        }
    }
};

template <int NT, bool PURE = false>
__device__ __forceinline__ void mac3(const WFrag &w, const h8 (&xh)[NT], const h8 (&xl)[NT], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32h(w.hi, xh[nt], acc[nt]);
    if constexpr (!PURE) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32h(w.hi, xl[nt], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32h(w.lo, xh[nt], acc[nt]);
    }
}


// a.ks1 = layer-1 k-steps of 32 grouped channels = ceil((3 + c_feat) / 32)
// PURE: the feature tensor holds halves (fp16 features in HBM, BASELINE configs[4]) and every operand is ONE half -- one MFMA
// per product block instead of three; coordinates are still differenced in fp32 and rounded once.
template <int C1, int C2, int NT, int NS, bool WLDS, bool PURE, bool PACKED>
__global__ __launch_bounds__(WLDS ? 512 : 256) void sa_group_mlp_f16_kernel(SaMlpArgs a) {
    if (a.run_if && *a.run_if == 0) return;   // workgroup-uniform, before any barrier
    constexpr int FRAG = frag_bytes<PURE>();
    const _Float16 *feat_h = reinterpret_cast<const _Float16 *>(a.feat);
    constexpr int T1 = C1 / 16, T2 = C2 / 16;
    constexpr int S1 = T1 / 2, S2 = T2 / 2;  // k32-steps over the previous layer's channels
    constexpr int UNIT = 16 * NT;
    constexpr int CPP = UNIT >= NS ? UNIT / NS : 1;  // whole centroids per unit ...
    constexpr bool PART = UNIT < NS;                  // ... or a unit is a slice of one centroid's samples (nsample 64)
    static_assert((UNIT % NS == 0 || NS % UNIT == 0) && (NS % 16) == 0, "units and centroids must nest");
    static_assert(T1 % 2 == 0 && T2 % 2 == 0, "two 16-row tiles chain into one K = 32 operand");
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);

    WSrc<WLDS, PURE> ws1, ws2, ws3;
    {
        const int MT3 = a.c3 / 16;
        const unsigned n1 = (unsigned)(T1 * a.ks1 * FRAG), n2 = (unsigned)(T2 * S1 * FRAG), n3 = (unsigned)(MT3 * S2 * FRAG);
        if constexpr (WLDS) {
            extern __shared__ __attribute__((aligned(16))) char wlds[];
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)wlds;
            ws1.lds = base; ws2.lds = base + n1; ws3.lds = base + n1 + n2;
            i32x4 *dst = reinterpret_cast<i32x4 *>(wlds);
            const i32x4 *s1 = reinterpret_cast<const i32x4 *>(a.w1), *s2 = reinterpret_cast<const i32x4 *>(a.w2),
                        *s3 = reinterpret_cast<const i32x4 *>(a.w3);
            for (unsigned i = threadIdx.x; i < n1 / 16; i += blockDim.x) dst[i] = s1[i];
            for (unsigned i = threadIdx.x; i < n2 / 16; i += blockDim.x) dst[n1 / 16 + i] = s2[i];
            for (unsigned i = threadIdx.x; i < n3 / 16; i += blockDim.x) dst[(n1 + n2) / 16 + i] = s3[i];
            __syncthreads();
        } else {
            ws1.rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.w1, 0, n1, 0x00020000);
            ws2.rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, n2, 0x00020000);
            ws3.rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.w3, 0, n3, 0x00020000);
        }
    }
    // biases in LDS for the whole launch (see sa_mlp.hip: a global load where a tile begins draws a vmcnt(0) wait -- with
    // the weights in LDS that is the bias's own round trip, exposed once per output tile)
    __shared__ __attribute__((aligned(16))) float sbias[C1 + C2 + 256];
    for (int i = threadIdx.x; i < C1; i += blockDim.x) sbias[i] = a.b1[i];
    for (int i = threadIdx.x; i < C2; i += blockDim.x) sbias[C1 + i] = a.b2[i];
    for (int i = threadIdx.x; i < a.c3; i += blockDim.x) sbias[C1 + C2 + i] = a.b3[i];
    __syncthreads();
    const float *b1l = sbias, *b2l = sbias + C1, *b3l = sbias + C1 + C2;
    bool any_bad = false;   // some unit of this wave met an operand beyond the representable range (or NaN / Inf)
    constexpr bool packed = PACKED;   // a separate instantiation: the padded form keeps its register budget
    // (packed columns: pooled rows merged into `out` by atomic max -- sa_mlp_args.h, the staged grouping of sa_stack)
    const bool merge = PACKED && a.merge_max && !flag_or_any(a.merge_unless, a.merge_unless_any, a.merge_unless_count);
    const MlpRange rg = mlp_range(a);
    const int nunits = packed ? (*a.ntiles) / NT : rg.units;   // packed: as many units as pack_columns produced tiles for
    for (int unit = wave; unit < nunits; unit += nwaves) {
        const int ub = packed ? 0 : unit / rg.ups;
        const long long col0 = ((long long)ub * a.m + rg.j0) * NS + (long long)(unit - ub * rg.ups) * UNIT;
        h8 h2hi[S2][NT], h2lo[S2][NT];
        PackedUnit<NT> pu;
        float mx = 0.f;  // largest operand magnitude this lane has split in this unit
        {
            int src[NT];
            long long bj[NT];
            int bb[NT];
            if constexpr (packed) {
                load_packed_unit<NT>(a, unit, c, src, bj, bb, pu);
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const long long e = col0 + nt * 16 + c;
                    bj[nt] = e / NS;
                    bb[nt] = ub;  // a unit never straddles scenes
                    src[nt] = a.idx[e];
                }
            }
            // ---------------- layer 1: k-steps of 32 gathered channels ----------------
            f32x4 acc1[T1][NT];
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(b1l + 16 * t + 4 * q);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc1[t][nt] = bias;
            }
            auto gather4 = [&](int k16, int nt) -> f32x4 {
                f32x4 v;
                // a block of padding only (its weights are zero): no loads, no address arithmetic -- the narrow scales
                // (3 + 1 grouped channels) are VALU-bound and half of their gather was this
                if (16 * k16 >= 3 + a.c_feat) return (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.feat_pm) {
                    // point-major features (B, N, C), C % 4 == 0, grouped channel order [features, xyz, pad]: a lane's
                    // four channels are one 16-byte load, the four q-lanes of a column read 64 contiguous bytes
                    const int ch0 = 16 * k16 + 4 * q;
                    if (ch0 < a.c_feat) {
                        if constexpr (PURE) {
                            const h4 hv = *reinterpret_cast<const h4 *>(feat_h + ((size_t)bb[nt] * a.n + src[nt]) * a.c_feat + ch0);
                            v = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                        } else {
                            v = *reinterpret_cast<const f32x4 *>(a.feat + ((size_t)bb[nt] * a.n + src[nt]) * a.c_feat + ch0);
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int ax = ch0 + jj - a.c_feat;
                            v[jj] = ax < 3 ? a.xyz[((size_t)bb[nt] * a.n + src[nt]) * 3 + ax] - a.new_xyz[(size_t)bj[nt] * 3 + ax] : 0.f;
                        }
                    }
                    return v;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int ch = 16 * k16 + 4 * q + jj;  // grouped channel: 0..2 centred xyz, 3.. features
                    float x;
                    if (ch < 3) {
                        x = a.xyz[((size_t)bb[nt] * a.n + src[nt]) * 3 + ch] - a.new_xyz[(size_t)bj[nt] * 3 + ch];
                    } else if (a.c_feat == 0) {
                        x = 0.f;
                    } else {
                        int cf = ch - 3;
                        cf = cf < a.c_feat ? cf : a.c_feat - 1;  // padded channel: finite data times a zero weight
                        const size_t at = ((size_t)bb[nt] * a.c_feat + cf) * a.n + src[nt];
                        x = PURE ? (float)feat_h[at] : a.feat[at];
                    }
                    v[jj] = x;
                }
                return v;
            };
            f32x4 xcur[2][NT], xnext[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { xcur[0][nt] = gather4(0, nt); xcur[1][nt] = gather4(1, nt); }
            for (int ks = 0; ks < a.ks1; ++ks) {
                const bool more = ks + 1 < a.ks1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    xnext[0][nt] = more ? gather4(2 * ks + 2, nt) : xcur[0][nt];
                    xnext[1][nt] = more ? gather4(2 * ks + 3, nt) : xcur[1][nt];
                }
                h8 xhi[NT], xlo[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    split4<0, false, PURE>(xcur[0][nt], xhi[nt], xlo[nt], mx);
                    if (16 * (2 * ks + 1) >= 3 + a.c_feat) {  // padding block (zeros): nothing to split
#pragma unroll
                        for (int r = 4; r < 8; ++r) { xhi[nt][r] = (_Float16)0.f; xlo[nt][r] = (_Float16)0.f; }
                    } else {
                        split4<4, false, PURE>(xcur[1][nt], xhi[nt], xlo[nt], mx);
                    }
                }
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    const WFrag w = ws1.load(lane * 16, (t * a.ks1 + ks) * FRAG);
                    mac3<NT, PURE>(w, xhi, xlo, acc1[t]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { xcur[0][nt] = xnext[0][nt]; xcur[1][nt] = xnext[1][nt]; }
            }
            h8 h1hi[S1][NT], h1lo[S1][NT];
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (t % 2 == 0) split4<0, true, PURE>(acc1[t][nt], h1hi[t / 2][nt], h1lo[t / 2][nt], mx);
                    else split4<4, true, PURE>(acc1[t][nt], h1hi[t / 2][nt], h1lo[t / 2][nt], mx);
                }

            // ---------------- layer 2 ----------------
            {
                constexpr int KCH = (S1 % 4 == 0) ? 4 : ((S1 % 2 == 0) ? 2 : 1);  // k32-steps per prefetched chunk
                constexpr int NCH = S1 / KCH;
                constexpr int G = T2 * NCH;
                WFrag w[2][KCH];
#pragma unroll
                for (int u = 0; u < KCH; ++u) w[0][u] = ws2.load(lane * 16, u * FRAG);
                f32x4 acc[NT];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int mt = g / NCH, ch = g % NCH;
                    if (g + 1 < G) {
#pragma unroll
                        for (int u = 0; u < KCH; ++u) w[(g + 1) & 1][u] = ws2.load(lane * 16, ((g + 1) * KCH + u) * FRAG);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ch == 0) {
                        const f32x4 bias = *reinterpret_cast<const f32x4 *>(b2l + 16 * mt + 4 * q);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
                    }
#pragma unroll
                    for (int u = 0; u < KCH; ++u) mac3<NT, PURE>(w[g & 1][u], h1hi[ch * KCH + u], h1lo[ch * KCH + u], acc);
                    if (ch == NCH - 1) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (mt % 2 == 0) split4<0, true, PURE>(acc[nt], h2hi[mt / 2][nt], h2lo[mt / 2][nt], mx);
                            else split4<4, true, PURE>(acc[nt], h2hi[mt / 2][nt], h2lo[mt / 2][nt], mx);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // Every operand of this unit has been split by now (layer 3's outputs go to the pool as fp32).  One beyond the
        // exactly representable range -- or NaN / Inf in the gathered inputs -- POISONS the unit: its pooled rows are written
        // as NaN instead of a silently clamped value, and the launch reports it (fused.check_overflow).
        const bool poison = __builtin_amdgcn_ballot_w64(mx > 65504.f) != 0ull;
        any_bad |= poison;
        const float nan_or = __int_as_float(0x7fc00000);
        // ---------------- layer 3 (runtime width) + max-pool ----------------
        const long long bj0 = col0 / NS;
        {
            constexpr int KCH = (S2 % 4 == 0) ? 4 : ((S2 % 2 == 0) ? 2 : 1);
            constexpr int NCH = S2 / KCH;
            const int MT3 = a.c3 / 16;
            WFrag wfirst[KCH];
#pragma unroll
            for (int u = 0; u < KCH; ++u) wfirst[u] = ws3.load(lane * 16, u * FRAG);
            for (int mt = 0; mt < MT3; ++mt) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(b3l + 16 * mt + 4 * q);
                const int tile_off = mt * S2 * FRAG;
                const int next_off = ((mt + 1 < MT3) ? mt + 1 : mt) * S2 * FRAG;
                WFrag w[2][KCH];
#pragma unroll
                for (int u = 0; u < KCH; ++u) w[0][u] = wfirst[u];
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (ch + 1 < NCH) {
#pragma unroll
                        for (int u = 0; u < KCH; ++u) w[(ch + 1) & 1][u] = ws3.load(lane * 16, tile_off + ((ch + 1) * KCH + u) * FRAG);
                    } else {
#pragma unroll
                        for (int u = 0; u < KCH; ++u) wfirst[u] = ws3.load(lane * 16, next_off + u * FRAG);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < KCH; ++u) mac3<NT, PURE>(w[ch & 1][u], h2hi[ch * KCH + u], h2lo[ch * KCH + u], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (packed) {
                    pool_write_packed<NT>(a, acc, pu, mt, q, c, poison, merge);
                    continue;
                }
                f32x4 best[CPP];
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) best[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};   // ReLU + pool = one integer max from +0
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cc = PART ? 0 : (nt * 16) / NS;
#pragma unroll
                    for (int r = 0; r < 4; ++r) best[cc][r] = imaxf(best[cc][r], acc[nt][r]);
                }
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) {
                    const f32x4 pooled4 = row_allmax4i(best[cc]);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = poison ? nan_or : pooled4[r];
                    if (c == 0) {
                        const long long cen = bj0 + cc;
                        const int b = ub, j = (int)(cen - (long long)ub * a.m);  // no 64-bit division: the unit lies inside scene ub
                        // PART = a slice of the centroid's samples: combine with the other slices (values >= 0 after the
                        // ReLU order like ints; the caller zero-fills `out` for nsample 64)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * mt + 4 * q + r;
                            if (row < a.c3_real) {
                                float *dst = a.out + ((size_t)b * a.out_c_total + a.out_c_off + row) * a.m + j;
                                if constexpr (PART) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(v[r]));
                                else *dst = v[r];
                            }
                        }
                    }
                }
            }
        }
    }
    if (any_bad && a.overflow) *a.overflow = 1;
}

template <int C1, int C2, int NT, int NS, bool PURE>
static int launch_f16_variant(const SaMlpArgs &a, hipStream_t st) {
    constexpr int UNIT = 16 * NT;
    constexpr int FRAG = frag_bytes<PURE>();
    SaMlpArgs k = a;
    if (a.cols) {                      // packed columns: `units` = tile capacity on entry; the kernel reads the real count
        k.ups = 1;
        k.units = a.units / NT;
    } else {
        const long long cols_scene = (long long)a.ups * NS;
        if (cols_scene % UNIT != 0)
            return fail(SPS_ERR_INVALID, "sa_group_mlp(f16): centroids*nsample per scene (%lld) not a multiple of %d", cols_scene, UNIT);
        k.ups = (int)(cols_scene / UNIT);
        k.units = a.units * k.ups;
        k.alt_j0 = 0;                                     // the whole layer: (scenes) x (all m centroids)
        k.alt_ups = (int)((long long)a.m * NS / UNIT);
        k.alt_units = a.units * k.alt_ups;
        if (a.alt && ((long long)a.m * NS) % UNIT != 0)
            return fail(SPS_ERR_INVALID, "sa_group_mlp: centroids*nsample per scene not a multiple of %d", UNIT);
    }
    k.ks1 = (3 + a.c_feat + 31) / 32;
    // all three layers' fragments in LDS when they fit beside nothing else (one workgroup of 8 waves per CU) and the
    // launch is big enough to amortise the copy
    const size_t wbytes = (size_t)FRAG * ((size_t)(C1 / 16) * k.ks1 + (size_t)(C2 / 16) * (C1 / 32) + (size_t)(a.c3 / 16) * (C2 / 32));
    if (C1 >= 64 && wbytes <= 128 * 1024 && k.units >= 2048) {  // (the 32-wide scale is faster streaming: measured)
        int blocks = divup(k.units, 8);
        if (blocks > 256) blocks = 256;
        if (a.cols) {
            static LdsLimitOnce raised;  // one per instantiation
            const int rc = raise_lds_limit((const void *)sa_group_mlp_f16_kernel<C1, C2, NT, NS, true, PURE, true>, 128 * 1024,
                                           raised, "sa_group_mlp(f16)");
            if (rc != SPS_OK) return rc;
            hipLaunchKernelGGL((sa_group_mlp_f16_kernel<C1, C2, NT, NS, true, PURE, true>), dim3(blocks), dim3(512), wbytes, st, k);
        } else {
            static LdsLimitOnce raised;
            const int rc = raise_lds_limit((const void *)sa_group_mlp_f16_kernel<C1, C2, NT, NS, true, PURE, false>, 128 * 1024,
                                           raised, "sa_group_mlp(f16)");
            if (rc != SPS_OK) return rc;
            hipLaunchKernelGGL((sa_group_mlp_f16_kernel<C1, C2, NT, NS, true, PURE, false>), dim3(blocks), dim3(512), wbytes, st, k);
        }
        return check_launch("sa_group_mlp_f16_kernel<lds weights>");
    }
    int blocks = divup(k.units, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (a.cols) hipLaunchKernelGGL((sa_group_mlp_f16_kernel<C1, C2, NT, NS, false, PURE, true>), dim3(blocks), dim3(256), 0, st, k);
    else hipLaunchKernelGGL((sa_group_mlp_f16_kernel<C1, C2, NT, NS, false, PURE, false>), dim3(blocks), dim3(256), 0, st, k);
    return check_launch("sa_group_mlp_f16_kernel");
}

// pure = the feature tensor and every MFMA operand are fp16 (mode 3 of sps_sa_group_mlp_ex); otherwise split-fp16 (mode 1)
int launch_sa_mlp_f16(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st, bool pure) {
#define SPS_MLPH_CASE(C1, C2, NT, NS)                                                                \
    if (c1 == C1 && c2 == C2 && nsample == NS)                                                       \
        return pure ? launch_f16_variant<C1, C2, NT, NS, true>(a, st) : launch_f16_variant<C1, C2, NT, NS, false>(a, st);
    SPS_MLPH_CASE(32, 32, 2, 32)
    SPS_MLPH_CASE(64, 64, 2, 16)
    SPS_MLPH_CASE(64, 96, 2, 32)
    SPS_MLPH_CASE(128, 128, 2, 16)
    SPS_MLPH_CASE(128, 256, 2, 32)
    SPS_MLPH_CASE(32, 32, 2, 16)
    SPS_MLPH_CASE(128, 64, 2, 16)
    SPS_MLPH_CASE(128, 96, 2, 32)
    SPS_MLPH_CASE(32, 32, 2, 64)
    SPS_MLPH_CASE(64, 64, 2, 64)
    SPS_MLPH_CASE(64, 96, 2, 64)
    SPS_MLPH_CASE(128, 128, 2, 64)
    SPS_MLPH_CASE(128, 256, 2, 64)
#undef SPS_MLPH_CASE
    return fail(SPS_ERR_INVALID, "sa_group_mlp(f16): no kernel for widths (%d, %d) nsample %d", c1, c2, nsample);
}

}  // namespace sps
