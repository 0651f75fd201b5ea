// sa_mlp.hip -- fused "group -> shared MLP -> max-pool" of one SA scale on the gfx950 matrix cores.
//
// Replaces, for inference (BatchNorm folded into the 1x1 convolutions), the per-scale chain of the
// reference's PointnetSAModuleMSG_WithSampling.forward
// (pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:429-447, built at :199-211):
//     grouping_operation x2 + subtract + cat          (pointnet2_utils.py:312-320)
//     [Conv2d 1x1 -> BatchNorm2d -> ReLU] x 3         (MIOpen conv + BN + clamp kernels)
//     F.max_pool2d(kernel=[1, nsample])
// -- ~14 launches and 8 HBM round trips of the (B, C, M, nsample) activation tensor per scale --
// by ONE kernel whose activations never leave registers.
//
// Matrix-core mapping (v_mfma_f32_16x16x4_f32: exact fp32, D = A(16x4) * B(4x16) + C):
//   A = weights  (16 output channels x 4 input channels), one f32 per lane
//   B = activations (4 input channels x 16 grouped points = "columns"), one f32 per lane
//   D = 16 output channels x 16 columns: lane (q = lane>>4, c = lane&15) holds rows 4q..4q+3 of column c.
// The D layout of layer l is ALREADY a valid B operand of layer l+1 if the k-steps of layer l+1 walk the
// input channels in the order (tile t, register r) -> channels {16t + 4q + r : q = 0..3}; the weight
// fragments are pre-permuted on the host accordingly (pack_* in spsnet_amd/fused.py; layer 1 as
// [tile][k-step][lane], layers 2-3 as [tile][k-step/4][lane][4] so that one dwordx4 feeds 4 MFMAs), so the activations
// chain register to register with no LDS and no cross-lane traffic.  Bias enters as the C input of the
// first MFMA, ReLU is one v_max per register, and the max-pool over the nsample columns of a centroid is a
// 4-step DPP row reduction on the last layer's accumulators (relu(max x) == max relu(x)).
//
// Work split: one wave owns one UNIT = one centroid x nsample columns (two centroids when nsample = 16 and
// NT = 2) at a time and walks units grid-stride; waves never synchronise.  The grouped input (3 centred xyz
// channels + C feature channels of the nsample ball-query neighbours) is gathered straight from the
// (B,N,3) / (B,C,N) tensors, one prefetched 4-channel k-step ahead of the MFMAs.
#include "sps_common.h"
#include "sa_mlp_args.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SPS_MLP_MAX_C3 = 256;   // widest last layer of the per-wave kernels (its biases are staged in LDS)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Weight fragments are fetched with buffer loads: SGPR descriptor + SGPR chunk offset + lane offset, so the
// unrolled layers need no per-load 64-bit address registers (global_load immediates only reach 4 KiB and
// hipcc otherwise materialises -- and spills -- one pointer per fragment).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 wload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return (f32x4){__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
}


// C1, C2: padded widths of layers 1 and 2 (multiples of 16).  NT: 16-column tiles per MFMA pass.
// NS: nsample (16, 32 or 64), with 16*NT >= NS: a unit is 16*NT columns = 16*NT/NS whole centroids.
template <int C1, int C2, int NT, int NS, bool PACKED>
__global__ __launch_bounds__(256) void sa_group_mlp_kernel(SaMlpArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    constexpr int T1 = C1 / 16, T2 = C2 / 16;
    constexpr int COLS = 16 * NT;
    constexpr int UNIT = COLS;
    constexpr int CPP = COLS >= NS ? COLS / NS : 1;  // whole centroids per unit (1 or 2) ...
    constexpr bool PART = COLS < NS;                  // ... or a unit is a slice of one centroid's samples (nsample 64)
    static_assert((COLS % NS == 0 || NS % COLS == 0) && (NS % 16) == 0, "units and centroids must nest");

    // The biases live in LDS for the whole launch.  Loaded from global memory where an output tile begins, each drew an
    // `s_waitcnt vmcnt(0)` -- which also waits for the weight chunk requested just before it: one full memory round trip per
    // output tile (40 per unit at the widest scale) with nothing else to hide it at one or two waves per SIMD.  An LDS read
    // counts on lgkmcnt and leaves the weight stream's counted vmcnt waits alone.
    __shared__ __attribute__((aligned(16))) float sbias[C1 + C2 + SPS_MLP_MAX_C3];
    for (int i = threadIdx.x; i < C1; i += blockDim.x) sbias[i] = a.b1[i];
    for (int i = threadIdx.x; i < C2; i += blockDim.x) sbias[C1 + i] = a.b2[i];
    for (int i = threadIdx.x; i < a.c3; i += blockDim.x) sbias[C1 + C2 + i] = a.b3[i];
    __syncthreads();
    const float *b1l = sbias, *b2l = sbias + C1, *b3l = sbias + C1 + C2;

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, c = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);

    constexpr bool packed = PACKED;   // a separate instantiation: the padded form keeps its register budget
    const bool merge = PACKED && a.merge_max && !flag_or_any(a.merge_unless, a.merge_unless_any, a.merge_unless_count);
    const MlpRange rg = mlp_range(a);
    const int nunits = packed ? (*a.ntiles) / NT : rg.units;   // packed: as many units as pack_columns produced tiles for
    for (int unit = wave; unit < nunits; unit += nwaves) {
        // first flattened (b, j, s) column of the unit: scene = unit / ups, centroids from j0 on
        const int ub = packed ? 0 : unit / rg.ups;
        const long long col0 = ((long long)ub * a.m + rg.j0) * NS + (long long)(unit - ub * rg.ups) * UNIT;
        f32x4 h2[T2][NT];
        PackedUnit<NT> pu;
        {
            // column owned by this lane in tile nt
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
            // ---------------- layer 1: k-steps over the gathered channels (runtime count) ----------------
            f32x4 h1[T1][NT];
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(b1l + 16 * t + 4 * q);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) h1[t][nt] = bias;
            }
            auto gather = [&](int ks, int nt) -> float {
                const int ch = 4 * ks + q;  // grouped channel: 0..2 = centred xyz, 3.. = features
                if (ks == 0 && q < 3) {
                    const float p = a.xyz[((size_t)bb[nt] * a.n + src[nt]) * 3 + q];
                    return p - a.new_xyz[(size_t)bj[nt] * 3 + q];
                }
                int cf = ch - 3;
                cf = cf < a.c_feat ? cf : a.c_feat - 1;  // padded k: finite data times a zero weight
                if (a.c_feat == 0) return 0.f;
                return a.feat[((size_t)bb[nt] * a.c_feat + cf) * a.n + src[nt]];
            };
            const __amdgpu_buffer_rsrc_t rs1 = weight_rsrc(a.w1, (unsigned)(T1 * a.ks1 * 64 * 4));
            float xcur[NT], xnext[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xcur[nt] = gather(0, nt);
            for (int ks = 0; ks < a.ks1; ++ks) {
                const bool more = ks + 1 < a.ks1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) xnext[nt] = more ? gather(ks + 1, nt) : 0.f;
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    const float wf = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, lane * 4, (t * a.ks1 + ks) * 256, 0));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) h1[t][nt] = mfma16(wf, xcur[nt], h1[t][nt]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) xcur[nt] = xnext[nt];
            }
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[t][nt][r] = relu_keep_nan(h1[t][nt][r]);

            // ---------------- layer 2: activations chain register-to-register ----------------
            // Weight fragments stream from L2 in chunks of KCH k-steps, double-buffered in registers: the
            // chunk after the one being multiplied is already in flight (the sched_barriers keep hipcc from
            // hoisting every load of the unrolled layer to the top and spilling).
            {
                // fragments packed [mt][KS/4][lane][4]: one dwordx4 per lane = 4 consecutive k-steps
                constexpr int KS = C1 / 4;
                constexpr int KCH = (KS % 16 == 0) ? 16 : ((KS % 8 == 0) ? 8 : 4);
                constexpr int NCH = KS / KCH;
                constexpr int G = T2 * NCH;
                constexpr int Q4 = KCH / 4;  // dwordx4 loads per chunk
                const __amdgpu_buffer_rsrc_t rs = weight_rsrc(a.w2, (unsigned)(T2 * KS * 64 * 4));
                f32x4 w[2][Q4];
#pragma unroll
                for (int u = 0; u < Q4; ++u) w[0][u] = wload4(rs, lane * 16, u * 1024);
                f32x4 acc[NT];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int mt = g / NCH, ch = g % NCH;
                    if (g + 1 < G) {
#pragma unroll
                        for (int u = 0; u < Q4; ++u) w[(g + 1) & 1][u] = wload4(rs, lane * 16, ((g + 1) * Q4 + u) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ch == 0) {
                        const f32x4 bias = *reinterpret_cast<const f32x4 *>(b2l + 16 * mt + 4 * q);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
                    }
#pragma unroll
                    for (int kk = 0; kk < KCH; ++kk) {
                        const int ks = ch * KCH + kk;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt] = mfma16(w[g & 1][kk / 4][kk % 4], h1[ks / 4][nt][ks % 4], acc[nt]);
                    }
                    if (ch == NCH - 1) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) h2[mt][nt][r] = relu_keep_nan(acc[nt][r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // ---------------- layer 3 (runtime width) + max-pool over the unit's columns ----------------
        const long long bj0 = col0 / NS;  // first centroid of the unit
        {
            constexpr int KS = C2 / 4;
            constexpr int KCH = (KS % 16 == 0) ? 16 : ((KS % 8 == 0) ? 8 : 4);
            constexpr int NCH = KS / KCH;
            constexpr int Q4 = KCH / 4;
            const int MT3 = a.c3 / 16;
            const __amdgpu_buffer_rsrc_t rs = weight_rsrc(a.w3, (unsigned)(MT3 * KS * 64 * 4));
            f32x4 wfirst[Q4];  // first chunk of the NEXT output tile, prefetched during the current one
#pragma unroll
            for (int u = 0; u < Q4; ++u) wfirst[u] = wload4(rs, lane * 16, u * 1024);
            for (int mt = 0; mt < MT3; ++mt) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(b3l + 16 * mt + 4 * q);
                const int tile_off = mt * (KS / 4) * 1024;                                  // bytes, wave-uniform
                const int next_off = ((mt + 1 < MT3) ? mt + 1 : mt) * (KS / 4) * 1024;
                f32x4 w[2][Q4];
#pragma unroll
                for (int u = 0; u < Q4; ++u) w[0][u] = wfirst[u];
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    if (ch + 1 < NCH) {
#pragma unroll
                        for (int u = 0; u < Q4; ++u) w[(ch + 1) & 1][u] = wload4(rs, lane * 16, tile_off + ((ch + 1) * Q4 + u) * 1024);
                    } else {
#pragma unroll
                        for (int u = 0; u < Q4; ++u) wfirst[u] = wload4(rs, lane * 16, next_off + u * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < KCH; ++kk) {
                        const int ks = ch * KCH + kk;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt] = mfma16(w[ch & 1][kk / 4][kk % 4], h2[ks / 4][nt][ks % 4], acc[nt]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (packed) {
                    pool_write_packed<NT>(a, acc, pu, mt, q, c, false, merge);
                    continue;
                }
                f32x4 best[CPP];
#pragma unroll
                // ReLU and pooling as ONE integer max starting from +0 (relu(max x) == max(0, x...)): NaN-keeping, see
                // relu_keep_nan
                for (int cc = 0; cc < CPP; ++cc) best[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cc = PART ? 0 : (nt * 16) / NS;  // which centroid of the unit this tile belongs to
#pragma unroll
                    for (int r = 0; r < 4; ++r) best[cc][r] = imaxf(best[cc][r], acc[nt][r]);
                }
#pragma unroll
                for (int cc = 0; cc < CPP; ++cc) {
                    const f32x4 pooled4 = row_allmax4i(best[cc]);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = pooled4[r];
                    if (c == 0) {
                        const long long cen = bj0 + cc;
                        const int b = ub, j = (int)(cen - (long long)ub * a.m);  // no 64-bit division: the unit lies inside scene ub
                        // PART = a slice of the centroid's samples: combine with the other slices.  Pooled values are
                        // >= 0 after the ReLU, so their bit patterns order like ints and the caller's zero fill is the
                        // identity (sps_sa_group_mlp_ex requires a zeroed `out` for nsample 64)
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
}

template <int C1, int C2, int NT, int NS>
static int launch_variant(const SaMlpArgs &a, hipStream_t st) {
    constexpr int UNIT = 16 * NT;
    SaMlpArgs k = a;
    if (a.cols) {                      // packed columns: `units` = tile capacity on entry; the kernel reads the real count
        k.ups = 1;
        k.units = a.units / NT;
    } else {
        const long long cols_scene = (long long)a.ups * NS;  // caller passes centroids per scene in `ups`, scenes in `units`
        if (cols_scene % UNIT != 0)
            return fail(SPS_ERR_INVALID, "sa_group_mlp: centroids*nsample per scene (%lld) not a multiple of %d", cols_scene, UNIT);
        k.ups = (int)(cols_scene / UNIT);
        k.units = a.units * k.ups;
        k.alt_j0 = 0;                                     // the whole layer: (scenes) x (all m centroids)
        k.alt_ups = (int)((long long)a.m * NS / UNIT);
        k.alt_units = a.units * k.alt_ups;
        if (a.alt && ((long long)a.m * NS) % UNIT != 0)
            return fail(SPS_ERR_INVALID, "sa_group_mlp: centroids*nsample per scene not a multiple of %d", UNIT);
    }
    const int waves_per_block = 4;
    int blocks = divup(k.units, waves_per_block);
    const int max_blocks = 256 * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    if (a.cols) hipLaunchKernelGGL((sa_group_mlp_kernel<C1, C2, NT, NS, true>), dim3(blocks), dim3(64 * waves_per_block), 0, st, k);
    else hipLaunchKernelGGL((sa_group_mlp_kernel<C1, C2, NT, NS, false>), dim3(blocks), dim3(64 * waves_per_block), 0, st, k);
    return check_launch("sa_group_mlp_kernel");
}

}  // namespace sps

namespace sps { int g_mlp_f16 = 0; }

// 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32), 1 = split-fp16 (hi+lo) on v_mfma_f32_16x16x16_f16 (sa_mlp_f16.hip).
// The weight buffers handed to sps_sa_group_mlp must be packed for the mode in force.  Returns the old mode.
extern "C" int sps_set_mlp_precision(int mode) {
    const int old = sps::g_mlp_f16;
    sps::g_mlp_f16 = mode ? 1 : 0;
    return old;
}

// Widths are the PADDED widths (multiples of 16) the weight fragments were packed for; c3_real <= c3 is the
// number of output channels written.  Supported (c1, c2, nsample) combinations are the IA-SSD / SPSNet ones.
extern "C" int sps_sa_group_mlp(int b, int n, int m, int c_feat, int nsample, const float *xyz,
                                const float *new_xyz, const float *features, const int *idx, int c1, int c2,
                                int c3, int c3_real, const float *w1, const float *b1, const float *w2,
                                const float *b2, const float *w3, const float *b3, float *out, int out_c_total,
                                int out_c_off, sps_stream_t stream) {
    return sps_sa_group_mlp_range(b, n, m, 0, m, c_feat, nsample, xyz, new_xyz, features, idx, c1, c2, c3, c3_real, w1, b1,
                                  w2, b2, w3, b3, out, out_c_total, out_c_off, stream);
}

extern "C" int sps_sa_group_mlp_range(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                                      const float *new_xyz, const float *features, const int *idx, int c1, int c2,
                                      int c3, int c3_real, const float *w1, const float *b1, const float *w2,
                                      const float *b2, const float *w3, const float *b3, float *out, int out_c_total,
                                      int out_c_off, sps_stream_t stream) {
    return sps_sa_group_mlp_ex(b, n, m, j0, jcount, c_feat, nsample, xyz, new_xyz, features, idx, c1, c2, c3, c3_real, w1, b1,
                               w2, b2, w3, b3, out, out_c_total, out_c_off, sps::g_mlp_f16, nullptr, stream);
}

extern "C" int sps_sa_group_mlp_ex(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                                   const float *new_xyz, const float *features, const int *idx, int c1, int c2,
                                   int c3, int c3_real, const float *w1, const float *b1, const float *w2,
                                   const float *b2, const float *w3, const float *b3, float *out, int out_c_total,
                                   int out_c_off, int split_fp16, int *overflow_flag, sps_stream_t stream) {
    return sps_sa_group_mlp_packed(b, n, m, j0, jcount, c_feat, nsample, xyz, new_xyz, features, idx, nullptr, nullptr, nullptr,
                                   0, c1, c2, c3, c3_real, w1, b1, w2, b2, w3, b3, out, out_c_total, out_c_off, split_fp16,
                                   overflow_flag, nullptr, nullptr, stream);
}

// The general entry point.  Either idx (b, m, nsample) and the centroid range [j0, j0 + jcount), or -- cols != NULL -- the
// packed column stream of sps_pack_columns (cols / meta / *ntiles on the device, tile_cap = the capacity the pack call was
// given; idx, j0 and jcount are then ignored).  mode = the split_fp16 word of sps_sa_group_mlp_ex, + 8: `out` is point-major
// (b, m, out_c_total) instead of (b, out_c_total, m).  Packed columns are served by modes 0, 1 and 3.  run_if (device, may be
// NULL): the launch does nothing when *run_if == 0.  full_range_if (device, may be NULL; idx form only): when
// *full_range_if != 0 the launch covers all centroids [0, m) of every scene instead of [j0, j0 + jcount) -- the last chunk of
// a streamed layer repairs the chunks whose bounded progress wait gave up, at no cost when none did.
extern "C" int sps_sa_group_mlp_packed(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                                       const float *new_xyz, const float *features, const int *idx, const int *cols,
                                       const unsigned *meta, const int *ntiles, long long tile_cap, int c1, int c2, int c3,
                                       int c3_real, const float *w1, const float *b1, const float *w2, const float *b2,
                                       const float *w3, const float *b3, float *out, int out_c_total, int out_c_off,
                                       int split_fp16, int *overflow_flag, const int *run_if, const int *full_range_if,
                                       sps_stream_t stream) {
    return sps_sa_group_mlp_packed_merge(b, n, m, j0, jcount, c_feat, nsample, xyz, new_xyz, features, idx, cols, meta, ntiles,
                                         tile_cap, c1, c2, c3, c3_real, w1, b1, w2, b2, w3, b3, out, out_c_total, out_c_off,
                                         split_fp16, overflow_flag, run_if, full_range_if, nullptr, 0, stream);
}

// The same with the merge mode's second "plain stores after all" predicate (mode + 16, packed columns): the launch stores
// instead of merging while *full_range_if != 0 OR any of unless_any[0 .. unless_count) != 0 (device ints, may be NULL).
extern "C" int sps_sa_group_mlp_packed_merge(int b, int n, int m, int j0, int jcount, int c_feat, int nsample, const float *xyz,
                                             const float *new_xyz, const float *features, const int *idx, const int *cols,
                                             const unsigned *meta, const int *ntiles, long long tile_cap, int c1, int c2, int c3,
                                             int c3_real, const float *w1, const float *b1, const float *w2, const float *b2,
                                             const float *w3, const float *b3, float *out, int out_c_total, int out_c_off,
                                             int split_fp16, int *overflow_flag, const int *run_if, const int *full_range_if,
                                             const int *unless_any, int unless_count, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || c_feat < 0 || nsample <= 0 || c3 <= 0 || (c3 % 16) || c3_real > c3 ||
        out_c_off < 0 || out_c_off + c3_real > out_c_total || j0 < 0 || jcount < 0 || j0 + jcount > m)
        return fail(SPS_ERR_INVALID, "sa_group_mlp: bad shape");
    if (b == 0 || (jcount == 0 && !cols)) return SPS_OK;
    // (the channel-major / split-fp16 per-wave kernels stage at most 256 last-layer biases in LDS; the shared-stream kernel,
    //  mode 2, and the exact-fp32 point-major kernel, mode 4, serve the 512 / 1024-wide scales of IA-SSD layer 5)
    if ((split_fp16 & 3) != 2 && (split_fp16 & 7) != 4 && c3 > 256)
        return fail(SPS_ERR_INVALID, "sa_group_mlp: last layer wider than 256 (%d)", c3);
    if (!xyz || !new_xyz || (!idx && !cols) || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out || (c_feat > 0 && !features))
        return fail(SPS_ERR_INVALID, "sa_group_mlp: null pointer");
    if (cols && (!meta || !ntiles || tile_cap <= 0 || tile_cap > 0x7FFFFFF || (tile_cap & 3) || m >= (1 << 20) || b > 256))
        return fail(SPS_ERR_INVALID, "sa_group_mlp: packed columns need meta, ntiles, a tile capacity that is a multiple of 4, "
                                     "m < 2^20 and b <= 256");
    if ((split_fp16 & 8) && !cols && (split_fp16 & 3) != 2 && (split_fp16 & 7) != 4)
        return fail(SPS_ERR_INVALID, "sa_group_mlp: a point-major `out` is served by the packed-column, shared-stream and "
                                     "point-major fp32 kernels");
    if (cols && (split_fp16 & 3) == 2)
        return fail(SPS_ERR_INVALID, "sa_group_mlp: the shared-stream kernel (mode 2) does not take packed columns");
    const long long cols_total = (long long)b * m * nsample;
    if (cols_total > 0x7FFFFFFFLL) return fail(SPS_ERR_INVALID, "sa_group_mlp: too many grouped points");
    SaMlpArgs a;
    a.n = n; a.m = m; a.c_feat = c_feat;
    a.units = cols ? (int)tile_cap : b;   // scenes (launch_variant turns this into the total unit count) / tile capacity
    a.ups = jcount;    // centroids per scene in the range (launch_variant turns this into units per scene)
    a.j0 = j0;
    a.cols = cols; a.meta = meta; a.ntiles = ntiles;
    a.out_pm = (split_fp16 & 8) ? 1 : 0;
    a.run_if = run_if;
    a.alt = cols ? nullptr : full_range_if;
    // mode + 16 (packed columns): merge the pooled rows into `out` by atomic max;
    // full_range_if then means "plain stores after all" (this launch covers every column: a repair)
    a.merge_max = (cols && (split_fp16 & 16)) ? 1 : 0;
    a.merge_unless = a.merge_max ? full_range_if : nullptr;
    a.merge_unless_any = a.merge_max ? unless_any : nullptr;
    a.merge_unless_count = a.merge_max ? unless_count : 0;
    if ((split_fp16 & 16) && !(cols && (split_fp16 & 3) != 2))
        return fail(SPS_ERR_INVALID, "sa_group_mlp: merge mode (16) needs packed columns (every kernel but the shared-stream one)");
    a.alt_j0 = 0; a.alt_ups = 0; a.alt_units = 0;
    a.ks1 = (3 + c_feat + 3) / 4;
    a.c3 = c3; a.c3_real = c3_real; a.out_c_total = out_c_total; a.out_c_off = out_c_off;
    a.xyz = xyz; a.new_xyz = new_xyz; a.feat = features; a.idx = idx;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.out = out;
    a.overflow = overflow_flag;
    a.hoist1 = (split_fp16 & 32) ? 1 : 0;   // `features` = layer 1's feature product per point (sps_sa_layer1_per_point)
    if (a.hoist1 && !((split_fp16 & 7) == 4 && c_feat == c1))
        return fail(SPS_ERR_INVALID, "sa_group_mlp: mode bit 32 belongs to the exact-fp32 point-major kernel (mode 4) and takes a "
                                     "(B, N, c1) feature tensor");
    split_fp16 &= ~(16 | 32);   // (consumed above)
    const int arith = split_fp16 & 3;
    a.feat_pm = (split_fp16 & 4) ? 1 : 0;
    if (a.feat_pm && (c_feat < 4 || (c_feat % 4)))
        return fail(SPS_ERR_INVALID, "sa_group_mlp: point-major features need c_feat %% 4 == 0 (got %d)", c_feat);
    if (a.feat_pm && arith == 0 && !sps_sa_group_mlp_pm_supported(c_feat, c1, c2, c3, nsample))
        return fail(SPS_ERR_INVALID, "sa_group_mlp: no exact-fp32 kernel for point-major features with %d channels, widths "
                                     "(%d, %d, %d), nsample %d", c_feat, c1, c2, c3, nsample);
    hipStream_t st = as_stream(stream);
    if (a.feat_pm && arith == 0) return launch_sa_mlp_pm(a, c1, c2, nsample, st);   // sa_mlp_pm.hip
    if (arith == 2) return launch_sa_mlp_f16_lds(a, c1, c2, nsample, st);
    if (arith) return launch_sa_mlp_f16(a, c1, c2, nsample, st, arith == 3);   // 3: `features` holds halves (fp16 in HBM)
#define SPS_MLP_CASE(C1, C2, NT, NS) \
    if (c1 == C1 && c2 == C2 && nsample == NS) return launch_variant<C1, C2, NT, NS>(a, st);
    SPS_MLP_CASE(16, 16, 2, 16)    // IA-SSD L0 r=0.2 [4,16,16,32]
    SPS_MLP_CASE(32, 32, 2, 32)    // IA-SSD L0 r=0.8 [4,32,32,64]
    SPS_MLP_CASE(64, 64, 2, 16)    // L1 [67,64,64,128]
    SPS_MLP_CASE(64, 96, 2, 32)    // L1 [67,64,96,128]
    SPS_MLP_CASE(128, 128, 2, 16)  // L2 [131,128,128,256]
    SPS_MLP_CASE(128, 256, 2, 32)  // L2 [131,128,256,256]
    SPS_MLP_CASE(16, 16, 2, 32)
    SPS_MLP_CASE(32, 32, 2, 16)
    SPS_MLP_CASE(128, 64, 2, 16)   // SPSNet L1 [127,124->128,64,128]
    SPS_MLP_CASE(128, 96, 2, 32)   // SPSNet L1 [127,124->128,96,128]
    SPS_MLP_CASE(16, 16, 2, 64)    // nsample 64 (BASELINE config 5): a centroid spans two units, atomic max
    SPS_MLP_CASE(32, 32, 2, 64)
    SPS_MLP_CASE(64, 64, 2, 64)
    SPS_MLP_CASE(64, 96, 2, 64)
    SPS_MLP_CASE(128, 128, 2, 64)
    SPS_MLP_CASE(128, 256, 2, 64)
#undef SPS_MLP_CASE
    return fail(SPS_ERR_INVALID, "sa_group_mlp: no kernel for widths (%d, %d) nsample %d", c1, c2, nsample);
}

// 1 if sps_sa_group_mlp has a kernel for these padded widths / nsample
// widths that only the shared-stream split-fp16 kernel (mode 2) serves
extern "C" int sps_sa_group_mlp_supported_stream(int c1, int c2, int c3, int nsample) {
    static const int tab[][4] = {{256, 256, 512, 16}, {256, 512, 1024, 32}};
    for (auto &t : tab)
        if (t[0] == c1 && t[1] == c2 && t[2] == c3 && t[3] == nsample) return 1;
    return 0;
}

extern "C" int sps_sa_group_mlp_supported(int c1, int c2, int nsample) {
    static const int tab[][3] = {{16, 16, 16}, {32, 32, 32}, {64, 64, 16}, {64, 96, 32}, {128, 128, 16}, {128, 256, 32},
                                 {16, 16, 32}, {32, 32, 16}, {128, 64, 16}, {128, 96, 32},
                                 {16, 16, 64}, {32, 32, 64}, {64, 64, 64}, {64, 96, 64}, {128, 128, 64}, {128, 256, 64}};
    for (auto &t : tab)
        if (t[0] == c1 && t[1] == c2 && t[2] == nsample) return 1;
    return 0;
}
