// sa_mlp_args.h -- kernel argument block shared by the fp32 and split-fp16 grouped-MLP kernels.
#pragma once
#include "sps_common.h"

namespace sps {

struct SaMlpArgs {
    int n, m, c_feat, units;       // points/scene, centroids/scene, feature channels, total units
    int j0, ups;                   // centroid range start within every scene; units per scene
    int ks1;                       // layer-1 k-steps = ceil((3 + c_feat) / 4)
    int c3;                        // padded last-layer width (multiple of 16)
    int c3_real;                   // channels actually written
    int out_c_total, out_c_off;    // out is (B, out_c_total, M); this scale writes [off, off + c3_real)
    const float *xyz, *new_xyz, *feat;
    const int *idx;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;
    int feat_pm;                   // 1: `feat` is point-major (B, N, c_feat) and layer 1's channel order is [features, xyz]
                                   // (mode 3, pure fp16: `feat` points at halves)
    int *overflow;                 // split-fp16 kernel: set to 1 if an operand exceeded the exactly splittable range
    // Packed columns (pack_columns.hip): when `cols` is set, `idx` / j0 / ups are ignored, a unit is UNIT consecutive
    // entries of cols / meta and the number of units is *ntiles / (UNIT / 16), read on the device.
    const int *cols;
    const unsigned *meta;
    const int *ntiles;
    int out_pm;                    // 1: `out` is point-major (B, M, out_c_total): a centroid's pooled rows are contiguous
    const int *run_if;             // predicated launch: the kernel returns at once when *run_if == 0 (NULL: always runs)
    // Self-repairing range launch: when *alt != 0 the kernel covers ALL centroids of every scene (alt_j0 = 0, alt_ups,
    // alt_units) instead of its range -- the last chunk of a streamed layer redoes the chunks whose bounded wait gave up,
    // without any extra launch when none did.  The unit loops are grid-stride, so the chunk's grid serves either range.
    const int *alt;
    int alt_j0, alt_ups, alt_units;
    // Packed columns only: the pooled rows are MERGED into `out` with an atomic max (values >= 0 order like ints) instead
    // of stored -- `out` already holds the maxima over columns that went through an earlier launch (sa_stack: the next
    // layer's early columns).  While *merge_unless != 0 (a repair: this launch covers every column) plain stores instead.
    int merge_max;
    const int *merge_unless, *merge_unless_any;
    int merge_unless_count;
    // Exact-fp32 kernel on point-major features (sa_mlp_pm.hip) only: layer 1's product over the FEATURE channels was computed
    // once per point (sps_sa_layer1_per_point) -- `feat` then holds b1 + W1f . features[point] as (B, N, C1) and layer 1 in
    // the grouped kernel is the coordinate k-step on top of the gathered row.
    int hoist1;
};

// the (units, units per scene, first centroid) a launch works on: its range, or the whole layer when *alt is set
struct MlpRange { int units, ups, j0; };
#ifdef __HIPCC__
__device__ __forceinline__ MlpRange mlp_range(const SaMlpArgs &a) {
    MlpRange r = {a.units, a.ups, a.j0};
    if (a.alt && *a.alt != 0) { r.units = a.alt_units; r.ups = a.alt_ups; r.j0 = a.alt_j0; }
    return r;
}
#endif


// max over the 16 lanes of each DPP row (= the 16 columns of an MFMA tile) for the four rows a lane holds, valid in
// every lane: xor butterfly quad_perm [1,0,3,2] / [2,3,0,1] / row_half_mirror / row_mirror.  Written as v_max_f32_dpp:
// hipcc emits v_mov_b32_dpp + v_max_f32 per step from update_dpp + fmaxf (twice the VALU slots, and on this chip VALU
// work is not hidden behind other waves' MFMAs); the four independent chains cover each other's DPP wait states.
#ifdef __HIPCC__
typedef float sps_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sps_f32x4 row_allmax4(sps_f32x4 v) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
#define SPS_DPP_MAX4(CTRL)                                                    \
    "v_max_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n" SPS_DPP_MAX4("quad_perm:[1,0,3,2]") SPS_DPP_MAX4("quad_perm:[2,3,0,1]")
                 SPS_DPP_MAX4("row_half_mirror") SPS_DPP_MAX4("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef SPS_DPP_MAX4
    return (sps_f32x4){a, b, c, d};
}

// ReLU and max-pooling in the INTEGER domain.  Non-negative floats (and +Inf, +NaN) order like their bit patterns as signed
// ints, everything with the sign bit set is a negative int: max_i32(bits(x), 0) is ReLU, and it keeps +Inf / +NaN where
// v_max_f32 would drop a NaN -- torch's ReLU and max_pool2d propagate NaN (pointnet2_modules.py:432-440 runs them).
// (A NaN with the sign bit set becomes 0; the arithmetic here produces the default +NaN.)
__device__ __forceinline__ float relu_keep_nan(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ float imaxf(float a, float b) {   // both >= 0 (or +Inf / +NaN) as floats
    const int x = __float_as_int(a), y = __float_as_int(b);
    return __int_as_float(x > y ? x : y);
}
// row_allmax4 on non-negative values, NaN-keeping: v_max_i32_dpp
__device__ __forceinline__ sps_f32x4 row_allmax4i(sps_f32x4 v) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
#define SPS_DPP_MAX4(CTRL)                                                    \
    "v_max_i32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_i32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_i32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_i32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n" SPS_DPP_MAX4("quad_perm:[1,0,3,2]") SPS_DPP_MAX4("quad_perm:[2,3,0,1]")
                 SPS_DPP_MAX4("row_half_mirror") SPS_DPP_MAX4("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef SPS_DPP_MAX4
    return (sps_f32x4){a, b, c, d};
}

// ---- packed-column epilogue shared by the per-wave kernels (sa_mlp.hip, sa_mlp_f16.hip) ------------------------------------
// meta word of a column: centroid j within its scene [19:0], scene [27:20], log2(slot) [30:28], unused [31].  A unit keeps
// only these words (one VGPR per tile) across the three layers; everything else is decoded where it is used.
template <int NT>
struct PackedUnit {
    unsigned w[NT];
};

template <int NT>
__device__ __forceinline__ void load_packed_unit(const SaMlpArgs &a, int unit, int c, int (&src)[NT], long long (&bj)[NT],
                                                 int (&bb)[NT], PackedUnit<NT> &pu) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const size_t e = (size_t)unit * (16 * NT) + nt * 16 + c;
        src[nt] = a.cols[e];
        pu.w[nt] = a.meta[e];
        // a unit never straddles windows (pack_columns pads a window to whole groups of four tiles), hence never scenes:
        // the scene stays wave-uniform, and with it the gathers' address arithmetic stays on the scalar unit
        bb[nt] = __builtin_amdgcn_readfirstlane((int)((pu.w[nt] >> 20) & 0xFFu));
        bj[nt] = (long long)bb[nt] * a.m + (int)(pu.w[nt] & 0xFFFFFu);
    }
}

// row_allmax4i (values already >= 0: ReLU applied) restricted to each lane's slot of 2^lg columns: step s of the xor butterfly runs only in the lanes with
// lg > s (EXEC masked; the mask comes from one v_cmp per step); their partners lie in the same aligned slot, hence are
// enabled too.  Afterwards every lane of a slot holds the slot's maximum.
__device__ __forceinline__ sps_f32x4 slot_allmax4(sps_f32x4 v, unsigned lg) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
    unsigned long long saved, mask;
#define SPS_DPP_MAX4(CTRL)                                                    \
    "v_max_i32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_i32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_i32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n"           \
    "v_max_i32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n"
#define SPS_SLOT_STEP(S, CTRL)                                                \
    "v_cmp_lt_u32_e64 %5, " S ", %6\n"                                        \
    "s_and_saveexec_b64 %4, %5\n"                                             \
    "s_nop 4\n" SPS_DPP_MAX4(CTRL)                                            \
    "s_mov_b64 exec, %4\n"
    asm volatile(SPS_SLOT_STEP("0", "quad_perm:[1,0,3,2]") SPS_SLOT_STEP("1", "quad_perm:[2,3,0,1]")
                 SPS_SLOT_STEP("2", "row_half_mirror") SPS_SLOT_STEP("3", "row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(saved), "=&s"(mask)
                 : "v"(lg)
                 : "scc");
#undef SPS_SLOT_STEP
#undef SPS_DPP_MAX4
    return (sps_f32x4){a, b, c, d};
}

// rows 16 mt + 4 q .. + 3 of centroid (b, j): plain stores, or an atomic max of the (non-negative) values onto zeros
// PM: the instantiation honours a point-major `out` (the packed kernels; the padded per-wave kernels keep the reference
// layout only, and with it their register budget)
template <bool PM>
__device__ __forceinline__ void store_pooled_rows(const SaMlpArgs &a, int b, int j, int mt, int q, const float (&v)[4], bool atomic) {
    const int row0 = 16 * mt + 4 * q;
    if (PM && a.out_pm) {
        float *dst = a.out + ((size_t)b * a.m + j) * a.out_c_total + a.out_c_off + row0;
        if (!atomic && row0 + 3 < a.c3_real && ((a.out_c_total | a.out_c_off) & 3) == 0) {
            *reinterpret_cast<sps_f32x4 *>(dst) = (sps_f32x4){v[0], v[1], v[2], v[3]};
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < a.c3_real) {
                if (atomic) atomicMax(reinterpret_cast<int *>(dst + r), __float_as_int(v[r]));
                else dst[r] = v[r];
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (row0 + r < a.c3_real) {
            float *dst = a.out + ((size_t)b * a.out_c_total + a.out_c_off + row0 + r) * a.m + j;
            if (atomic) atomicMax(reinterpret_cast<int *>(dst), __float_as_int(v[r]));
            else *dst = v[r];
        }
}

// Pool one 16-row output tile of a PACKED unit over each centroid's slot (ReLU folded into the integer max), write.
// poison: the unit met an operand it could not represent (split-fp16 kernels): its rows are written as NaN, never clamped.
template <int NT>
__device__ __forceinline__ void pool_write_packed(const SaMlpArgs &a, const sps_f32x4 (&acc)[NT], const PackedUnit<NT> &pu,
                                                  int mt, int q, int c, bool poison = false, bool merge = false) {
    const int lg0 = __builtin_amdgcn_readfirstlane((int)((pu.w[0] >> 28) & 7u));
    const int scene = __builtin_amdgcn_readfirstlane((int)((pu.w[0] >> 20) & 0xFFu));
    const float nan = __int_as_float(0x7fc00000);
    if (lg0 >= 5) {   // the whole unit is (part of) ONE centroid's slot of 32 or 64 columns
        sps_f32x4 best = (sps_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) best[r] = imaxf(best[r], acc[nt][r]);
        const sps_f32x4 p = row_allmax4i(best);
        const float v[4] = {poison ? nan : p[0], poison ? nan : p[1], poison ? nan : p[2], poison ? nan : p[3]};
        // 64 columns = two units: they meet through an atomic max on the zero-filled `out` (values >= 0 order like ints)
        if (c == 0) store_pooled_rows<true>(a, scene, (int)(pu.w[0] & 0xFFFFFu), mt, q, v, lg0 >= 6 || merge);
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const unsigned w = pu.w[nt], lg = (w >> 28) & 7u;
        const sps_f32x4 relu = (sps_f32x4){relu_keep_nan(acc[nt][0]), relu_keep_nan(acc[nt][1]), relu_keep_nan(acc[nt][2]),
                                           relu_keep_nan(acc[nt][3])};
        const sps_f32x4 p = slot_allmax4(relu, lg);
        const float v[4] = {poison ? nan : p[0], poison ? nan : p[1], poison ? nan : p[2], poison ? nan : p[3]};
        const bool writer = (w >> 31) == 0u && (c & ((1 << lg) - 1)) == 0;   // first lane of a slot, never an unused lane
        if (writer) store_pooled_rows<true>(a, scene, (int)(w & 0xFFFFFu), mt, q, v, merge);
    }
}

#endif  // __HIPCC__

extern int g_mlp_f16;
// sa_mlp_f16.hip: split-fp16 variant; same argument block (units = scenes, ups = centroids per scene on entry)
int launch_sa_mlp_f16(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st, bool pure = false);
// sa_mlp_f16_lds.hip: the same arithmetic with the weight stream shared through LDS (a.w1 = concatenated stream)
int launch_sa_mlp_f16_lds(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st);
// exact fp32 with point-major features (sa_mlp_pm.hip)
int launch_sa_mlp_pm(const SaMlpArgs &a, int c1, int c2, int nsample, hipStream_t st);

}  // namespace sps
