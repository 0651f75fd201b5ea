// mlp_train.hip -- the grouped MLP of an SA layer in TRAINING mode, fused along its memory passes.
//
// pointnet2_modules.py:203-211 / :432-444 of the reference run, per scale, [Conv2d 1x1 -> BatchNorm2d (batch statistics) ->
// ReLU] x 3 and a max-pool on (B, C, M, nsample) activations.  BatchNorm needs the statistics of ALL grouped points before the
// next convolution can start, so the layers cannot be chained in registers as in inference; what can be removed are the
// elementwise passes between the convolutions.  Op by op (conv, statistics, normalise + ReLU, pool; backward: pool, BatchNorm
// reduce, BatchNorm apply, data gradient, weight gradient) every pre-activation tensor crosses HBM ~5 times forward and ~10
// times backward, and those passes already ran at 5-6 TB/s (bn_relu_train.hip), i.e. only fusion removes them.  Here the ONLY
// tensors that exist are the pre-BatchNorm convolution outputs Y_l (forward) and the gradients dA_l w.r.t. the post-ReLU
// activations (backward); everything else is recomputed where it is consumed:
//
//   forward   Y_l = W_l . T(Y_{l-1})        T = relu(fma(y, scale, shift)) applied while the operand is loaded; the epilogue
//                                           accumulates sum / sum of squares of Y_l per channel (tconv_kernel, TEPI_STATS)
//             pooled = max_s T(Y_3)         BatchNorm + ReLU + max-pool + arg-max in one pass (tpool_fwd_kernel)
//   backward  dY_l = scale (dZ - c1 - xhat c2), dZ = dA_l [z > 0]   recomputed from (dA_l, Y_l) in the operand load of BOTH
//             dA_{l-1} = W_l^T . dY_l       gradient kernels; for the last layer dA_3 is never materialised either: it is the
//             dW_l = dY_l . T(Y_{l-1})^T    pooled gradient routed by the arg-max (TIN_BNBWD_POOL).  The data-gradient
//                                           kernel's epilogue accumulates the two BatchNorm-backward sums of layer l-1.
//
// Arithmetic: split-fp16 on v_mfma_f32_16x16x32_f16 (every fp32 operand as hi + lo halves, three products, fp32 accumulate:
// ~22 significant bits, the same scheme as the inference kernels of sa_mlp_f16.hip) -- at 8 TB/s these GEMMs (64-256 channels
// on either side) need ~500 TFLOP/s of issued matrix work to stay memory-bound, three times what the fp32 MFMA pipe has.
// Every operand tensor is first multiplied by an exact power of two that brings its largest magnitude near 2^10 (see
// tpow2_scale: unscaled, the low half of a 1e-5 gradient is an fp16 denormal).  Operands beyond +-65504 after scaling, or
// NaN / Inf, are never clamped silently: the outputs that depend on them (or the whole weight gradient) are written as NaN
// and the overflow flag is raised.  Statistics are accumulated per lane in fp32 over a workgroup's columns, then in fp64
// in a fixed order; weight-gradient partials are summed in launch order (bit-reproducible from run to run).
// The kernels are HBM-streaming (3.5-4.6 TB/s at a million columns, tools/tconv_bench.py); DESIGN.md 4.6 / profiles/README.md have what
// bounds them at the IA-SSD shapes and what was measured on the way.
#include "sps_common.h"

namespace sps {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int TP = 8;   // floats per channel of a parameter block: mean, invstd, scale, shift, gamma, beta, c1, c2
enum { TIN_RAW = 0, TIN_BNRELU = 1, TIN_BNBWD = 2, TIN_BNBWD_POOL = 3 };
enum { TEPI_NONE = 0, TEPI_STATS = 1, TEPI_BWD = 2 };

__device__ __forceinline__ f32x4 tmfma(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// EXACT fp32 (the reference's arithmetic, fused.set_train_precision("fp32") / sps_set_train_precision(1)): the same kernels with
// their operands left as they are -- no scaling, no split, no range check -- on v_mfma_f32_16x16x4_f32.  A k-step of 32 input
// rows is then eight MFMAs of four rows (lane (q, c) holds rows 8 q + e: MFMA e takes row 8 q + e of every q as its k index q,
// on both operands alike -- a sum over k does not care in which order its terms meet) where the split form issues three of 32.
__device__ __forceinline__ f32x4 tmfma32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// relu that keeps a NaN (torch's clamp_min does); the pool compares its results with `>`, hence the float form here
__device__ __forceinline__ float trelu(float z) { return (z > 0.f || z != z) ? z : 0.f; }

// hi / lo halves of a value clamped into the fp16 range (the weights: scaled to 2^8 before, so the clamp never acts)
__device__ __forceinline__ void tsplit(float v, _Float16 &hi, _Float16 &lo) {
    const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    hi = (_Float16)c;
    lo = (_Float16)(c - (float)hi);
}
// The VALU, not the matrix pipe, is what these memory-bound kernels keep busy, so the per-element work is kept short:
//   * two values -> packed halves with ONE v_cvt_pkrtz_f16_f32 (round toward zero: hi + lo still carries 22 bits, the low half
//     takes what the high half dropped); round-toward-zero saturates at 65504 instead of producing Inf, so the range check
//     stays explicit: one v_max3_f32 (|a|, |b| as input modifiers) per pair.  A NaN passes through the conversion and the
//     matrix pipe and is caught on the accumulators.
//   * ReLU as an integer max (keeps +NaN like torch's clamp_min, one instruction).
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tsplit2(float a, float b, unsigned int &hi, unsigned int &lo, float &mx) {
    mx = fmaxf(fmaxf(mx, fabsf(a)), fabsf(b));
    const hp2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const hp2 l = __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]);
    hi = __builtin_bit_cast(unsigned int, h);
    lo = __builtin_bit_cast(unsigned int, l);
}
__device__ __forceinline__ float trelu_i(float z) {
    const int b = __float_as_int(z);
    return __int_as_float(b > 0 ? b : 0);
}
union TFrag { h8 v; unsigned int u[4]; };
// per-row constants of the BatchNorm backward: dY = ks dZ + ka y + kb  with  ks = scale sx, ka = -ks c2 invstd,
// kb = ks (c2 invstd mean - c1)   (= ks (dZ - c1 - xhat c2), xhat = (y - mean) invstd)
struct TBwdRow { float sc, sh, ks, ka, kb; };
__device__ __forceinline__ TBwdRow tbwd_row(const float *P, float sx) {
    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P), p1 = *reinterpret_cast<const f32x4 *>(P + 4);
    TBwdRow r;
    r.sc = p0[2]; r.sh = p0[3];
    r.ks = p0[2] * sx;
    const float t = r.ks * p1[3] * p0[1];
    r.ka = -t;
    r.kb = __builtin_fmaf(t, p0[0], -r.ks * p1[2]);
    return r;
}
__device__ __forceinline__ float tbwd_apply(const TBwdRow &r, float g, float y) {
    const float z = __builtin_fmaf(y, r.sc, r.sh);
    const float dz = z > 0.f ? g : 0.f;
    return __builtin_fmaf(r.ks, dz, __builtin_fmaf(r.ka, y, r.kb));
}
// Exact power-of-two scaling in front of the split: fp16 keeps 11 bits per half only for |v| >= 2^-14, and the low half of a
// value below ~0.06 is already a half denormal (absolute error 3e-8, not 2^-22 relative).  Gradients are routinely 1e-5 and
// smaller, so every operand tensor is multiplied by 2^k with the tensor's largest magnitude brought near 2^10 (weights: near
// 2^8), and the accumulators are multiplied by the inverse afterwards -- both exact.  The largest magnitude of a gradient
// tensor is tracked by the kernel that produces it (atomic max on the float's bits: order-independent, reproducible).
__device__ __forceinline__ float tpow2_scale(float target, float amax) {
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.f;
    int e = (int)floorf(log2f(target / amax));
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    return ldexpf(1.f, e);
}
// max over the workgroup (all threads get it); `scratch` = 8 floats of LDS
__device__ __forceinline__ float tblock_max(float v, float *scratch, int nwaves) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < nwaves; ++w) r = fmaxf(r, scratch[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ void tatomic_amax(float *dst, float v) {   // v >= 0 (or NaN, which then sticks)
    // (thousands of waves on one address: the value only grows, so whoever cannot raise it need not queue up for the atomic)
    const unsigned int bits = __float_as_uint(v);
    if (bits > __atomic_load_n(reinterpret_cast<unsigned int *>(dst), __ATOMIC_RELAXED))
        atomicMax(reinterpret_cast<unsigned int *>(dst), bits);
}

struct TConvArgs {
    int b, ci, co, S;            // scenes, input rows, output rows, k-steps of 32 input rows
    long long l;                 // columns per scene (multiple of 64)
    int trans;                   // A[o][i] = trans ? w[i * co + o] : w[o * ci + i]
    const float *w;
    const float *in, *in2;       // RAW / BNRELU: in = the operand; BNBWD: in = dA, in2 = Y; BNBWD_POOL: in2 = Y
    const float *gout;           // BNBWD_POOL: pooled gradient (b, ci, m) ...
    const unsigned char *arg;    // ... and the arg-max of the pool (b, ci, m)
    int ns, m;
    const float *pin;            // (ci, 8) parameter block of the input rows
    float *out;                  // (b, co, l)
    const float *epi_y;          // TEPI_BWD: Y of the OUTPUT rows (b, co, l)
    const float *pout;           // TEPI_BWD: (co, 8) parameter block of the output rows
    double *partial;             // [gridDim.x][co][2]
    const float *amax_in;        // BNBWD modes: largest |incoming gradient| (device scalar)
    float *amax_out;             // TEPI_BWD: atomic max of |out| (device scalar, zeroed by the caller)
    const float *wamax;          // largest |w| (device scalar)
    int *overflow;
    // K slabs (layers with more than 256 input rows, sps_tconv): this launch multiplies the input rows [k0, k0 + ci) of an
    // operand that has ci_total rows per scene -- in / in2 / gout / arg / pin already point at row k0, w at column k0 of
    // its rows (ld = w_ld floats; transposed: at row k0) -- and, behind the first slab, ADDS its product to what `out` holds
    // (accum; the statistics epilogue runs in the last slab only, on the complete sums).
    int ci_total, w_ld, accum;
};

// One 64-column block per wave at a time; the 16 RT output rows of blockIdx.z against all input rows.  Four waves per
// workgroup, ONE workgroup per CU at the wide shapes: what keeps HBM busy is not occupancy but the loads each wave has in
// flight -- the operand rows of a k-step (8 KiB per wave and tensor) are requested TWO k-steps ahead of their use, across
// block boundaries, from a three-deep ring of register buffers.
//   B operand (activations): lane (q, c) loads rows 32 s + 8 q + e, e = 0..7, as float4 = columns 4c..4c+3; element j of the
//   float4 feeds column tile j, so loads and stores stay 16 bytes per lane and 256 contiguous bytes per row.
//   A operand (weights): scaled and split once per workgroup into LDS as [tile][k-step][hi | lo][lane][8 halves]; F32: the
//   plain floats as [tile][k-step][rows e < 4 | e >= 4][lane][4 floats] -- the same 2 KiB per tile and k-step.
template <int RT, int IN, int EPI, bool F32>
__global__ __launch_bounds__(256) void tconv_kernel(TConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool GRAD = (IN == TIN_BNBWD || IN == TIN_BNBWD_POOL);
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wave = threadIdx.x >> 6;
    const int S = a.S, cip = 32 * S;
    const int rows_wg = 16 * RT;
    const int row0 = blockIdx.z * rows_wg;
    char *wl = smem;
    float *pin_l = reinterpret_cast<float *>(smem + (size_t)RT * S * 2048);
    float *pout_l = pin_l + (IN != TIN_RAW ? cip * TP : 0);
    double *red = reinterpret_cast<double *>(pout_l + (EPI == TEPI_BWD ? rows_wg * TP : 0));
    float *scratch = reinterpret_cast<float *>(red + (EPI != TEPI_NONE ? (size_t)4 * 16 * RT * 2 : 0));

    // this wave's blocks: id = (blockIdx.x + i gridDim.x) 4 + wave, i = 0 .. nblk - 1; its pieces: (block, k-step), in order
    const long long nb64 = a.l >> 6;
    const long long total = (long long)a.b * nb64;
    const long long first = (long long)blockIdx.x * 4 + wave, stride = (long long)gridDim.x * 4;
    const long long nblk = first < total ? (total - first + stride - 1) / stride : 0;
    const long long npieces = nblk * S;

    // (POOL mode: the pooled gradient and the raw arg-max byte; NO arithmetic on loaded values in here -- it would wait for
    //  the load, and with it for everything requested before it, where the request is issued)
    constexpr bool POOL = (IN == TIN_BNBWD_POOL);
    struct Piece { f32x4 xa[POOL ? 1 : 8], xb[GRAD ? 8 : 1]; float pg[POOL ? 8 : 1]; int pi[POOL ? 8 : 1]; };
    auto load = [&](long long piece, Piece &pc) {
        const long long i = piece / S;
        const int s = (int)(piece - i * S);
        const long long id = first + i * stride;
        const int scene = (int)(id / nb64);
        const long long col0 = (id - (long long)scene * nb64) * 64 + 4 * c;
        const float *src = a.in + (size_t)scene * a.ci_total * a.l + col0;
        const float *src2 = a.in2 + (size_t)scene * a.ci_total * a.l + col0;
        const long long cen = (IN == TIN_BNBWD_POOL) ? col0 / a.ns : 0;     // the four columns belong to ONE centroid
        const size_t pool_base = (size_t)scene * a.ci_total * a.m + cen;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 32 * s + 8 * q + e;
            if (!POOL) pc.xa[POOL ? 0 : e] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (POOL) { pc.pg[POOL ? e : 0] = 0.f; pc.pi[POOL ? e : 0] = 255; }
            if (GRAD) pc.xb[GRAD ? e : 0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (row < a.ci) {
                if (POOL) {
                    pc.pg[POOL ? e : 0] = a.gout[pool_base + (size_t)row * a.m];
                    pc.pi[POOL ? e : 0] = a.arg[pool_base + (size_t)row * a.m];
                } else {
                    pc.xa[POOL ? 0 : e] = *reinterpret_cast<const f32x4 *>(src + (size_t)row * a.l);
                }
                if (GRAD) pc.xb[GRAD ? e : 0] = *reinterpret_cast<const f32x4 *>(src2 + (size_t)row * a.l);
            }
        }
    };
    // (the gradient modes at 128 rows: one k-step ahead only -- two tensors per piece, and 192 accumulator / statistics registers)
    constexpr int DEPTH = (GRAD && RT == 8) ? 1 : 2;
    Piece b0, b1, b2;
    if (npieces > 0) load(0, b0);            // (in flight while the weights are staged)
    if (DEPTH == 2 && npieces > 1) load(1, b1);

    // Weights -> LDS: 16 independent loads per thread and trip (a plain loop waits for every load: 128 round trips to L2 per
    // workgroup at the widest shapes, longer than the workgroup's whole column range took); the transposed operand walks
    // the rows fastest so that consecutive lanes still read consecutive floats.
    const float sw = F32 ? 1.f : tpow2_scale(256.f, *a.wamax);
    const bool quads = (reinterpret_cast<uintptr_t>(a.w) & 15) == 0 && ((a.trans ? a.co : (a.ci | a.w_ld)) & 3) == 0;
    if (quads) {
        // 16-byte loads along the contiguous dimension of w (k for the plain operand, the output row for the transposed one)
        constexpr int WQ = 8;
        const int fast = a.trans ? rows_wg : cip;                  // extent of the contiguous dimension within this workgroup
        const int nq = (rows_wg * cip) >> 2;
        for (int base = 0; base < nq; base += 256 * WQ) {
            f32x4 wq[WQ];
#pragma unroll
            for (int u = 0; u < WQ; ++u) {
                const int idx = (base + u * 256 + (int)threadIdx.x) << 2;
                const int slow = idx / fast, f0 = idx - slow * fast;
                const int o = a.trans ? f0 : slow, k = a.trans ? slow : f0;
                const int row = row0 + o;
                wq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (idx < 4 * nq && row < a.co && k < a.ci)
                    wq[u] = *reinterpret_cast<const f32x4 *>(a.trans ? a.w + (size_t)k * a.co + row : a.w + (size_t)row * a.w_ld + k);
            }
#pragma unroll
            for (int u = 0; u < WQ; ++u) {
                const int idx = (base + u * 256 + (int)threadIdx.x) << 2;
                if (idx >= 4 * nq) continue;
                const int slow = idx / fast, f0 = idx - slow * fast;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = a.trans ? f0 + j : slow, k = a.trans ? slow : f0 + j;
                    const int t = o >> 4, i = o & 15, ss = k >> 5, qq = (k & 31) >> 3, e = k & 7;
                    if (F32) {
                        *reinterpret_cast<float *>(wl + ((size_t)(t * S + ss) * 2 + (e >> 2)) * 1024 + (qq * 16 + i) * 16 + (e & 3) * 4) = wq[u][j];
                        continue;
                    }
                    _Float16 hi, lo;
                    tsplit(wq[u][j] * sw, hi, lo);
                    char *dst = wl + ((size_t)(t * S + ss) * 2) * 1024 + (qq * 16 + i) * 16 + e * 2;
                    *reinterpret_cast<_Float16 *>(dst) = hi;
                    *reinterpret_cast<_Float16 *>(dst + 1024) = lo;
                }
            }
        }
    } else {
        constexpr int WB = 16;
        const int nel = rows_wg * cip;
        for (int base = 0; base < nel; base += 256 * WB) {
            float wv[WB];
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int idx = base + u * 256 + (int)threadIdx.x;
                const int o = a.trans ? idx % rows_wg : idx / cip, k = a.trans ? idx / rows_wg : idx % cip;
                const int row = row0 + o;
                wv[u] = 0.f;
                if (idx < nel && row < a.co && k < a.ci) wv[u] = a.trans ? a.w[(size_t)k * a.co + row] : a.w[(size_t)row * a.w_ld + k];
            }
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int idx = base + u * 256 + (int)threadIdx.x;
                if (idx >= nel) continue;
                const int o = a.trans ? idx % rows_wg : idx / cip, k = a.trans ? idx / rows_wg : idx % cip;
                const int t = o >> 4, i = o & 15, ss = k >> 5, qq = (k & 31) >> 3, e = k & 7;
                if (F32) {
                    *reinterpret_cast<float *>(wl + ((size_t)(t * S + ss) * 2 + (e >> 2)) * 1024 + (qq * 16 + i) * 16 + (e & 3) * 4) = wv[u];
                    continue;
                }
                _Float16 hi, lo;
                tsplit(wv[u] * sw, hi, lo);
                char *dst = wl + ((size_t)(t * S + ss) * 2) * 1024 + (qq * 16 + i) * 16 + e * 2;
                *reinterpret_cast<_Float16 *>(dst) = hi;
                *reinterpret_cast<_Float16 *>(dst + 1024) = lo;
            }
        }
    }
    float sx = 1.f;
    if (IN != TIN_RAW) {
        float smax = 0.f;
        for (int idx = threadIdx.x; idx < cip * TP; idx += 256) {
            const float v = (idx / TP < a.ci) ? a.pin[idx] : 0.f;
            pin_l[idx] = v;
            if ((idx & (TP - 1)) == 2) smax = fmaxf(smax, fabsf(v));
        }
        if (GRAD && !F32) sx = tpow2_scale(1024.f, tblock_max(smax, scratch, 4) * (*a.amax_in));
    }
    if (EPI == TEPI_BWD)
        for (int idx = threadIdx.x; idx < rows_wg * TP; idx += 256)
            pout_l[idx] = (row0 + idx / TP < a.co) ? a.pout[(size_t)row0 * TP + idx] : 0.f;
    __syncthreads();
    const float inv = 1.f / (sx * sw);                                  // powers of two: exact

    float st1[RT][4], st2[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[t][r] = 0.f; st2[t][r] = 0.f; }
    bool any_bad = false, nonfinite = false;
    float omax = 0.f;

    f32x4 acc[RT][4];
    float mx = 0.f;
    int sidx0 = 0;
    int s = 0;
    long long blk = 0;
    for (long long piece = 0; piece < npieces; ++piece) {
        if (DEPTH == 2) {
            if (piece + 2 < npieces) load(piece + 2, b2);
        } else if (piece + 1 < npieces) {
            load(piece + 1, b1);
        }
        if (s == 0) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mx = 0.f;
            if (POOL) {
                const long long id = first + blk * stride;
                const long long col0 = (id - (id / nb64) * nb64) * 64 + 4 * c;
                sidx0 = (int)(col0 % a.ns);
            }
        }
        TFrag bh[4], bl[4];
        f32x4 bv[F32 ? 8 : 1];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            f32x4 v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * e2 + h;
                const float *P = pin_l + (32 * s + 8 * q + e) * TP;
                if (IN == TIN_RAW) {
                    v[h] = b0.xa[POOL ? 0 : e];
                } else if (IN == TIN_BNRELU) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[h][j] = trelu_i(__builtin_fmaf(b0.xa[POOL ? 0 : e][j], p0[2], p0[3]));
                } else {
                    const TBwdRow rw = tbwd_row(P, sx);
                    f32x4 g = b0.xa[POOL ? 0 : e];
                    if (POOL) {
                        const int am = b0.pi[POOL ? e : 0] - sidx0;     // position of the arg-max relative to this lane's columns
                        const float gg = b0.pg[POOL ? e : 0];
                        g = (f32x4){am == 0 ? gg : 0.f, am == 1 ? gg : 0.f, am == 2 ? gg : 0.f, am == 3 ? gg : 0.f};
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[h][j] = tbwd_apply(rw, g[j], b0.xb[GRAD ? e : 0][j]);
                }
            }
            if (F32) {
                bv[F32 ? 2 * e2 : 0] = v[0];
                bv[F32 ? 2 * e2 + 1 : 0] = v[1];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) tsplit2(v[0][j], v[1][j], bh[j].u[e2], bl[j].u[e2], mx);
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const char *fr = wl + ((size_t)(t * S + s) * 2) * 1024 + lane * 16;
            if (F32) {
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(fr), a1 = *reinterpret_cast<const f32x4 *>(fr + 1024);
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = tmfma32(e < 4 ? a0[e & 3] : a1[e & 3], bv[F32 ? e : 0][j], acc[t][j]);
                continue;
            }
            const h8 ah = *reinterpret_cast<const h8 *>(fr), al = *reinterpret_cast<const h8 *>(fr + 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[t][j] = tmfma(ah, bh[j].v, acc[t][j]);
                acc[t][j] = tmfma(ah, bl[j].v, acc[t][j]);
                acc[t][j] = tmfma(al, bh[j].v, acc[t][j]);
            }
        }
        b0 = b1;
        if (DEPTH == 2) b1 = b2;
        if (++s < S) continue;
        // ---- the block is complete: rows 16 t + 4 q + r, columns col0 .. col0 + 3 ----
        s = 0;
        const long long id = first + blk * stride;
        ++blk;
        const int scene = (int)(id / nb64);
        const long long col0 = (id - (long long)scene * nb64) * 64 + 4 * c;
        const bool poison = !F32 && __builtin_amdgcn_ballot_w64(mx > 65504.f) != 0ull;   // an unrepresentable operand: NaN, never clamped
        any_bad |= poison;
        const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = 16 * t + 4 * q + r, row = row0 + orow;
                if (row >= a.co) continue;
                f32x4 o = (f32x4){acc[t][0][r] * inv, acc[t][1][r] * inv, acc[t][2][r] * inv, acc[t][3][r] * inv};
                const size_t at = ((size_t)scene * a.co + row) * a.l + col0;
                if (a.accum) {          // (a later K slab: on top of the earlier slabs' sums; a NaN row stays NaN)
                    const f32x4 prev = *reinterpret_cast<const f32x4 *>(a.out + at);
                    o = (f32x4){o[0] + prev[0], o[1] + prev[1], o[2] + prev[2], o[3] + prev[3]};
                }
                const bool finite = (fabsf(o[0]) + fabsf(o[1])) + (fabsf(o[2]) + fabsf(o[3])) < INFINITY;   // (false for a NaN too)
                // (exact fp32: an Inf / NaN is whatever the arithmetic made it, as in the reference -- nothing to flag or poison)
                nonfinite |= !F32 && !finite;
                if (!F32 && (poison || !finite)) o = (f32x4){nanv, nanv, nanv, nanv};
                *reinterpret_cast<f32x4 *>(a.out + at) = o;
                if (EPI == TEPI_STATS) {
                    st1[t][r] += (o[0] + o[1]) + (o[2] + o[3]);
                    st2[t][r] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                } else if (EPI == TEPI_BWD) {
                    const f32x4 yp = *reinterpret_cast<const f32x4 *>(a.epi_y + at);
                    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(pout_l + orow * TP);
                    omax = fmaxf(omax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float z = __builtin_fmaf(yp[j], p0[2], p0[3]);
                        const float dz = z > 0.f ? o[j] : 0.f;
                        st1[t][r] += dz;
                        st2[t][r] += dz * ((yp[j] - p0[0]) * p0[1]);
                    }
                }
            }
    }
    any_bad |= __builtin_amdgcn_ballot_w64(nonfinite) != 0ull;
    if (any_bad && a.overflow) *a.overflow = 1;
    if (EPI == TEPI_BWD && a.amax_out) {
        const float wgmax = tblock_max(any_bad ? INFINITY : omax, scratch, 4);
        if (threadIdx.x == 0) tatomic_amax(a.amax_out, wgmax);
    }
    if (EPI != TEPI_NONE) {
        // per-lane fp32 partials -> fp64, the 16 lanes of a row meet in a butterfly, the four waves in LDS (fixed order)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double d1 = st1[t][r], d2 = st2[t][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
                if (c == 0) {
                    double *o = red + ((size_t)wave * 16 * RT + 16 * t + 4 * q + r) * 2;
                    o[0] = d1; o[1] = d2;
                }
            }
        __syncthreads();
        for (int o = threadIdx.x; o < rows_wg; o += 256) {
            if (row0 + o >= a.co) continue;
            double d1 = 0.0, d2 = 0.0;
            for (int w = 0; w < 4; ++w) { d1 += red[((size_t)w * 16 * RT + o) * 2]; d2 += red[((size_t)w * 16 * RT + o) * 2 + 1]; }
            double *dst = a.partial + ((size_t)blockIdx.x * a.co + row0 + o) * 2;
            dst[0] = d1; dst[1] = d2;
        }
    }
}

// ---- weight gradient: dW (co x ci) = sum over all columns of dY (x) T(X) -------------------------------------------------
struct TWgradArgs {
    int b, co, ci;
    long long l;                 // columns per scene (multiple of 32 NK)
    int dmode, xmode;            // TIN_BNBWD / TIN_BNBWD_POOL for dY; TIN_RAW / TIN_BNRELU for the other operand
    const float *dA, *y;         // (b, co, l)
    const float *gout;
    const unsigned char *arg;
    int ns, m;
    const float *pd;             // (co, 8)
    const float *x;              // (b, ci, l)
    const float *px;             // (ci, 8)
    float *partial;              // [gridDim.x][cop][cip]  (cop, cip = co, ci rounded up to 16)
    const float *amax_in;        // largest |incoming gradient|
    int *overflow;
    // blocks of a wide layer (sps_twgrad: more than 256 rows on either side): the operands have co_total / ci_total rows per
    // scene, this launch takes co / ci of them starting where dA / y / gout / arg / pd and x / px point
    int co_total, ci_total;
};

// 8 waves; a stage = 32 NK columns of every row of both operands, transformed and split ONCE into LDS (rows 64 NK + 16 bytes
// apart: the 16 rows of a fragment start 16 bytes apart modulo 256, no bank conflicts), then each wave multiplies its output
// tiles (t = wt, wt + 4, ..; u = wu, wu + 2, ..).  The next stage's global loads are in flight during the MFMAs.  NK widens
// the stage for narrow layers (4 passes of 512 / (8 NK) rows cover 64 NK... rows): a stage then moves enough bytes to hide its
// two barriers.
// F32 (exact fp32, see tmfma32): ONE plane of floats per operand, rows 128 NK + 16 bytes apart (the same 16-bytes-modulo-256
// walk over the 16 rows of a fragment); a fragment read is the lane's eight columns 8 q .. 8 q + 7 of its row, MFMA e takes
// column 8 q + e of every q as its k index q.
template <int NK, bool DB, bool F32>
__global__ __launch_bounds__(512) void twgrad_kernel(TWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = F32 ? 128 * NK + 16 : 64 * NK + 16;            // LDS bytes per operand row (and plane)
    constexpr int CG = 8 * NK, RR = 512 / CG;   // float4 groups per row of a stage, rows per pass
    const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, c = lane & 15, wave = tid >> 6;
    const int wt = wave >> 1, wu = wave & 1;
    const int T = (a.co + 15) >> 4, U = (a.ci + 15) >> 4;
    const int cop = 16 * T, cip = 16 * U;
    // DB: two operand buffers, so that the waves that are done with a stage's products write the next stage while the others
    // still multiply -- ONE barrier per stage instead of two (whoever writes buffer p again has passed the barrier of the stage
    // in between, which everybody reaches only after its products on p)
    const size_t opbytes = (size_t)(cop + cip) * (F32 ? 1 : 2) * RS;
    char *ahi0 = smem, *alo0 = ahi0 + (F32 ? 0 : (size_t)cop * RS), *bhi0 = alo0 + (size_t)cop * RS, *blo0 = bhi0 + (F32 ? 0 : (size_t)cip * RS);
    float *pd_l = reinterpret_cast<float *>(smem + (DB ? 2 : 1) * opbytes);
    float *px_l = pd_l + cop * TP;
    float *scratch = px_l + cip * TP;           // 8 floats + the poison flag
    float smax = 0.f;
    for (int i = tid; i < cop * TP; i += 512) {
        const float v = (i / TP < a.co) ? a.pd[i] : 0.f;
        pd_l[i] = v;
        if ((i & (TP - 1)) == 2) smax = fmaxf(smax, fabsf(v));
    }
    for (int i = tid; i < cip * TP; i += 512) px_l[i] = (a.xmode != TIN_RAW && i / TP < a.ci) ? a.px[i] : 0.f;
    const float bmax = tblock_max(smax, scratch, 8);                                       // (ends with a barrier: LDS is ready)
    const float sx = F32 ? 1.f : tpow2_scale(1024.f, bmax * (*a.amax_in));

    const int cg = tid % CG, rr = tid / CG;                      // float4 group within the stage, row within a pass
    const long long per_scene = a.l / (32 * NK);
    const long long total = (long long)a.b * per_scene;
    const long long s0 = total * blockIdx.x / gridDim.x, s1 = total * (blockIdx.x + 1) / gridDim.x;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mx = 0.f;

    f32x4 pa[4], pb[4], pxv[4];
    float pg[4];
    int pi[4], psub = 0;     // POOL mode: pooled gradient, raw arg-max byte (no arithmetic on loaded values while prefetching)
    auto prefetch = [&](long long st) {
        const int scene = (int)(st / per_scene);
        const long long col = (st - (long long)scene * per_scene) * (32 * NK) + 4 * cg;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = rr + RR * p;
            pa[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            pb[p] = pa[p];
            pxv[p] = pa[p];
            pg[p] = 0.f;
            pi[p] = 255;
            if (row < a.co) {
                const size_t at = ((size_t)scene * a.co_total + row) * a.l + col;
                if (a.dmode == TIN_BNBWD_POOL) {
                    const long long cen = col / a.ns;
                    const size_t pat = ((size_t)scene * a.co_total + row) * a.m + cen;
                    pg[p] = a.gout[pat];
                    pi[p] = a.arg[pat];
                    psub = (int)(col - cen * a.ns);
                } else {
                    pa[p] = *reinterpret_cast<const f32x4 *>(a.dA + at);
                }
                pb[p] = *reinterpret_cast<const f32x4 *>(a.y + at);
            }
            if (row < a.ci) pxv[p] = *reinterpret_cast<const f32x4 *>(a.x + ((size_t)scene * a.ci_total + row) * a.l + col);
        }
    };
    auto put = [&](char *hi_base, char *lo_base, int row, const f32x4 v) {
        if (F32) {
            *reinterpret_cast<f32x4 *>(hi_base + (size_t)row * RS + cg * 16) = v;
            return;
        }
        uint2 hi, lo;
        tsplit2(v[0], v[1], hi.x, lo.x, mx);
        tsplit2(v[2], v[3], hi.y, lo.y, mx);
        *reinterpret_cast<uint2 *>(hi_base + (size_t)row * RS + cg * 8) = hi;
        *reinterpret_cast<uint2 *>(lo_base + (size_t)row * RS + cg * 8) = lo;
    };
    if (s0 < s1) prefetch(s0);
    for (long long st = s0; st < s1; ++st) {
        const size_t boff = (DB && ((st - s0) & 1)) ? opbytes : 0;
        char *ahi = ahi0 + boff, *alo = alo0 + boff, *bhi = bhi0 + boff, *blo = blo0 + boff;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = rr + RR * p;
            if (row < cop) {
                const TBwdRow rw = tbwd_row(pd_l + row * TP, sx);
                f32x4 g = pa[p], v;
                if (a.dmode == TIN_BNBWD_POOL) {
                    const int am = pi[p] - psub;
                    const float gg = pg[p];
                    g = (f32x4){am == 0 ? gg : 0.f, am == 1 ? gg : 0.f, am == 2 ? gg : 0.f, am == 3 ? gg : 0.f};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tbwd_apply(rw, g[j], pb[p][j]);
                put(ahi, alo, row, v);
            }
            if (row < cip) {
                f32x4 v = pxv[p];
                if (a.xmode == TIN_BNRELU) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(px_l + row * TP);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = trelu_i(__builtin_fmaf(v[j], p0[2], p0[3]));
                }
                put(bhi, blo, row, v);
            }
        }
        __syncthreads();
        if (st + 1 < s1) prefetch(st + 1);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = wt + 4 * i;
                if (t >= T) continue;
                if (F32) {
                    const char *ar = ahi + (size_t)(16 * t + c) * RS + kk * 128 + q * 32;
                    const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(ar), fa1 = *reinterpret_cast<const f32x4 *>(ar + 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int u = wu + 2 * j;
                        if (u >= U) continue;
                        const char *br = bhi + (size_t)(16 * u + c) * RS + kk * 128 + q * 32;
                        const f32x4 fb0 = *reinterpret_cast<const f32x4 *>(br), fb1 = *reinterpret_cast<const f32x4 *>(br + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = tmfma32(fa0[e], fb0[e], acc[i][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j] = tmfma32(fa1[e], fb1[e], acc[i][j]);
                    }
                    continue;
                }
                const h8 fah = *reinterpret_cast<const h8 *>(ahi + (size_t)(16 * t + c) * RS + kk * 64 + q * 16);
                const h8 fal = *reinterpret_cast<const h8 *>(alo + (size_t)(16 * t + c) * RS + kk * 64 + q * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int u = wu + 2 * j;
                    if (u >= U) continue;
                    const h8 fbh = *reinterpret_cast<const h8 *>(bhi + (size_t)(16 * u + c) * RS + kk * 64 + q * 16);
                    const h8 fbl = *reinterpret_cast<const h8 *>(blo + (size_t)(16 * u + c) * RS + kk * 64 + q * 16);
                    acc[i][j] = tmfma(fah, fbh, acc[i][j]);
                    acc[i][j] = tmfma(fah, fbl, acc[i][j]);
                    acc[i][j] = tmfma(fal, fbh, acc[i][j]);
                }
            }
        if (!DB) __syncthreads();
    }
    // an unrepresentable operand anywhere in this workgroup's columns: its whole partial is NaN
    float accbad = 0.f;     // a NaN operand went through the conversion and the matrix pipe: it shows on the accumulators
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!((fabsf(acc[i][j][0]) + fabsf(acc[i][j][1])) + (fabsf(acc[i][j][2]) + fabsf(acc[i][j][3])) < INFINITY)) accbad = INFINITY;
    const bool poison = !F32 && tblock_max(fmaxf(mx, accbad), scratch, 8) > 65504.f;
    if (poison && tid == 0 && a.overflow) *a.overflow = 1;
    const float nanv = __int_as_float(0x7fc00000), inv = 1.f / sx;
    float *dst = a.partial + (size_t)blockIdx.x * cop * cip;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wt + 4 * i;
        if (t >= T) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = wu + 2 * j;
            if (u >= U) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(16 * t + 4 * q + r) * cip + 16 * u + c] = poison ? nanv : acc[i][j][r] * inv;
        }
    }
}

// dw[o][i] = sum over the workgroup partials in a FIXED order: a workgroup owns 32 elements, its eight 32-thread groups
// take the partials k = g, g + 8, ... (eight loads in flight each), the eight group sums meet in LDS and are added in group
// order.  (One thread per element walking all partials was 32 dependent round trips: ~12 us per launch whatever the size,
// eighteen launches per training step.)
__global__ __launch_bounds__(256) void twgrad_reduce_kernel(int co, int ci, int cip, int cop, int nparts, const float *__restrict__ partial,
                                                            float *__restrict__ dw, int ld) {
    __shared__ float part[8][32];
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const bool live = e < co * ci;
    float s = 0.f;
    if (live) {
        const int o = e / ci, i = e - o * ci;
        const float *p = partial + (size_t)o * cip + i;
        const size_t stride = (size_t)cop * cip;
        int k = g;
        for (; k + 56 < nparts; k += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nparts; k += 8) s += p[(size_t)k * stride];
    }
    part[g][el] = s;
    __syncthreads();
    if (g == 0 && live) {
        float t = part[0][el];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += part[u][el];
        dw[(size_t)(e / ci) * ld + (e % ci)] = t;      // (ld = ci: the whole matrix; wider: a block of it)
    }
}

// ---- statistics -> parameter blocks -----------------------------------------------------------------------------------------
// one wave per channel: lane l adds parts l, l + 64, ..., the lane sums meet in a butterfly (a fixed order)
__device__ __forceinline__ void tsum_parts(const double *partial, int c, int ch, int nparts, double &s1, double &s2) {
    const int lane = threadIdx.x & 63;
    double d1 = 0.0, d2 = 0.0;
    for (int k = lane; k < nparts; k += 64) {
        d1 += partial[((size_t)k * c + ch) * 2];
        d2 += partial[((size_t)k * c + ch) * 2 + 1];
    }
    for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
    s1 = d1; s2 = d2;
}

// count_dev != NULL: the element count lives on the device (SyncBatchNorm: the all-reduced sum of the ranks' counts, which
// may differ from rank to rank -- torch's SyncBatchNorm gathers them too)
__global__ __launch_bounds__(256) void tbn_finalize_kernel(int c, int nparts, double count, const double *__restrict__ partial,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                           float momentum, float *__restrict__ running_mean,
                                                           float *__restrict__ running_var, float *__restrict__ P,
                                                           long long *__restrict__ num_batches_tracked,
                                                           const double *__restrict__ count_dev) {
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= c) return;
    if (count_dev) count = *count_dev;
    if (ch == 0 && (threadIdx.x & 63) == 0 && num_batches_tracked) *num_batches_tracked += 1;   // nn.BatchNorm2d.forward's counter
    double s1, s2;
    tsum_parts(partial, c, ch, nparts, s1, s2);
    if ((threadIdx.x & 63) != 0) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
    const float scale = g * invstd;
    float *p = P + (size_t)ch * TP;
    p[0] = mean; p[1] = invstd; p[2] = scale; p[3] = bt - mean * scale; p[4] = g; p[5] = bt; p[6] = 0.f; p[7] = 0.f;
    if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(256) void tbn_bwd_finalize_kernel(int c, int nparts, double inv_count, const double *__restrict__ partial,
                                                               float *__restrict__ P, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               const double *__restrict__ count_dev) {
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= c) return;
    if (count_dev) inv_count = 1.0 / *count_dev;
    double s1, s2;
    tsum_parts(partial, c, ch, nparts, s1, s2);
    if ((threadIdx.x & 63) != 0) return;
    P[(size_t)ch * TP + 6] = (float)(s1 * inv_count);
    P[(size_t)ch * TP + 7] = (float)(s2 * inv_count);
    if (dbeta) dbeta[ch] = (float)s1;
    if (dgamma) dgamma[ch] = (float)s2;
}

// ---- BatchNorm + ReLU + max over the samples (+ arg-max), one pass over Y_3 ------------------------------------------------
// the FIRST maximum wins; a NaN wins and of several NaNs the last keeps the index (torch's max_pool2d; group_gather.hip)
__device__ __forceinline__ bool tpool_takes(float b, int bi, float a, int ai) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return bn && (!an || bi > ai);
    return b > a || (b == a && bi < ai);
}
// G = nsample / 4 lanes share a row of nsample values (16 bytes each).  yarg = the pre-BatchNorm value at the arg-max: what
// the backward needs of Y for the BatchNorm sums of the last layer, so that it does not have to gather it again.
template <int G>
__global__ __launch_bounds__(256) void tpool_fwd_kernel(long long rows, int c, int m, const float *__restrict__ y,
                                                        const float *__restrict__ P, float *__restrict__ out,
                                                        unsigned char *__restrict__ arg, float *__restrict__ yarg) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = e / G;
    const int sub = (int)(e - r * G);
    const bool live = r < rows;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    float sc = 0.f, sh = 0.f;
    if (live) {
        v = *reinterpret_cast<const f32x4 *>(y + e * 4);
        const int ch = (int)((r / m) % c);
        sc = P[(size_t)ch * TP + 2];
        sh = P[(size_t)ch * TP + 3];
    }
    float best = trelu(__builtin_fmaf(v[0], sc, sh)), by = v[0];
    int bi = sub * 4;
#pragma unroll
    for (int u = 1; u < 4; ++u) {
        const float z = trelu(__builtin_fmaf(v[u], sc, sh));
        if (z > best || z != z) { best = z; bi = sub * 4 + u; by = v[u]; }
    }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        const float ob = __shfl_xor(best, off), oy = __shfl_xor(by, off);
        const int oi = __shfl_xor(bi, off);
        if (tpool_takes(ob, oi, best, bi)) { best = ob; bi = oi; by = oy; }
    }
    if (live && sub == 0) { out[r] = best; arg[r] = (unsigned char)bi; yarg[r] = by; }
}

// The two BatchNorm-backward sums of the LAST layer from the pooled gradient alone: dZ_3 is gout at the arg-max (where the
// pooled value is positive) and zero elsewhere.  One workgroup per (channel, scene); partial[(scene * c + ch) * 2].
__global__ __launch_bounds__(256) void tpool_bwd_stats_kernel(int c, int m, const float *__restrict__ yarg,
                                                              const float *__restrict__ gout, const float *__restrict__ P,
                                                              double *__restrict__ partial, float *__restrict__ amax_out) {
    __shared__ double sh1[4], sh2[4];
    const int ch = blockIdx.x, scene = blockIdx.y;
    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P + (size_t)ch * TP);
    const size_t base = ((size_t)scene * c + ch) * m;
    double d1 = 0.0, d2 = 0.0;
    float gm = 0.f;
    for (int j = threadIdx.x; j < m; j += 256) {
        const float g = gout[base + j];
        const float yy = yarg[base + j];
        const float z = __builtin_fmaf(yy, p0[2], p0[3]);
        const float dz = z > 0.f ? g : 0.f;
        d1 += dz;
        d2 += (double)dz * ((yy - p0[0]) * p0[1]);
        gm = (g != g) ? INFINITY : fmaxf(gm, fabsf(g));
    }
    __shared__ float shm[4];
    for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); gm = fmaxf(gm, __shfl_xor(gm, off)); }
    if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = d1; sh2[threadIdx.x >> 6] = d2; shm[threadIdx.x >> 6] = gm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = partial + ((size_t)scene * c + ch) * 2;
        o[0] = (sh1[0] + sh1[1]) + (sh1[2] + sh1[3]);
        o[1] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
        if (amax_out) tatomic_amax(amax_out, fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3])));   // one per workgroup
    }
}

// workgroups along the columns: as many as stay resident (one per CU at 128 rows per workgroup -- the weights then fill the
// LDS and a wave holds ~400 registers --, two otherwise), each walking its share of the 64-column blocks
// ---- the same stack WITHOUT a pool (the aggregation / confidence layers: Conv1d + BatchNorm1d + ReLU on (b, c, m)) ----------
// out = relu(fma(y, scale, shift)); one workgroup per (channel, scene) row
__global__ __launch_bounds__(256) void tbn_apply_relu_kernel(int c, long long l, const float *__restrict__ y, const float *__restrict__ P,
                                                             float *__restrict__ out) {
    const int ch = blockIdx.x, scene = blockIdx.y;
    const float sc = P[(size_t)ch * TP + 2], sh = P[(size_t)ch * TP + 3];
    const size_t base = ((size_t)scene * c + ch) * l;
    for (long long e = 4 * threadIdx.x; e < l; e += 4 * 256) {     // l % 4 == 0 (host)
        const f32x4 v = *reinterpret_cast<const f32x4 *>(y + base + e);
        *reinterpret_cast<f32x4 *>(out + base + e) = (f32x4){trelu_i(__builtin_fmaf(v[0], sc, sh)), trelu_i(__builtin_fmaf(v[1], sc, sh)),
                                                              trelu_i(__builtin_fmaf(v[2], sc, sh)), trelu_i(__builtin_fmaf(v[3], sc, sh))};
    }
}
// the BatchNorm-backward sums of the last layer from a DENSE incoming gradient; partial[(scene * c + ch) * 2], amax of |dA|
__global__ __launch_bounds__(256) void tbn_bwd_stats_kernel(int c, long long l, const float *__restrict__ y, const float *__restrict__ dA,
                                                            const float *__restrict__ P, double *__restrict__ partial,
                                                            float *__restrict__ amax_out) {
    __shared__ double sh1[4], sh2[4];
    __shared__ float shm[4];
    const int ch = blockIdx.x, scene = blockIdx.y;
    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P + (size_t)ch * TP);
    const size_t base = ((size_t)scene * c + ch) * l;
    double d1 = 0.0, d2 = 0.0;
    float gm = 0.f;
    for (long long e = 4 * threadIdx.x; e < l; e += 4 * 256) {
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + base + e), gv = *reinterpret_cast<const f32x4 *>(dA + base + e);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = __builtin_fmaf(yv[j], p0[2], p0[3]);
            const float dz = z > 0.f ? gv[j] : 0.f;
            s1 += dz;
            s2 += dz * ((yv[j] - p0[0]) * p0[1]);
            gm = (gv[j] != gv[j]) ? INFINITY : fmaxf(gm, fabsf(gv[j]));
        }
        d1 += s1;
        d2 += s2;
    }
    for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); gm = fmaxf(gm, __shfl_xor(gm, off)); }
    if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = d1; sh2[threadIdx.x >> 6] = d2; shm[threadIdx.x >> 6] = gm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = partial + ((size_t)scene * c + ch) * 2;
        o[0] = (sh1[0] + sh1[1]) + (sh1[2] + sh1[3]);
        o[1] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
        if (amax_out) tatomic_amax(amax_out, fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3])));
    }
}

int tconv_grid_x(int b, long long l, int co) {
    const long long groups = ((long long)b * (l >> 6) + 3) / 4;
    const long long cap = co > 64 ? 256 : 512;
    return (int)(groups < cap ? groups : cap);
}

template <int RT, int IN, bool F32>
int tconv_launch_epi(const TConvArgs &a, int epi, hipStream_t st) {
    const int cip = 32 * a.S, rows_wg = 16 * RT;
    const size_t lds = (size_t)RT * a.S * 2048 + (IN != TIN_RAW ? cip * TP * 4 : 0) + (epi == TEPI_BWD ? rows_wg * TP * 4 : 0) +
                       (epi != TEPI_NONE ? (size_t)4 * 16 * RT * 2 * 8 : 0) + 64;
    if (lds > 156 * 1024) return fail(SPS_ERR_INVALID, "tconv: %zu bytes of LDS (ci = %d)", lds, a.ci);
    const dim3 grid(tconv_grid_x(a.b, a.l, a.co), 1, divup(a.co, rows_wg));
#define SPS_TCONV_GO(EPI)                                                                                               \
    {                                                                                                                   \
        static LdsLimitOnce raised;                                                                                     \
        if (lds > 64 * 1024) {                                                                                          \
            const int rc = raise_lds_limit((const void *)tconv_kernel<RT, IN, EPI, F32>, 156 * 1024, raised, "tconv"); \
            if (rc != SPS_OK) return rc;                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL((tconv_kernel<RT, IN, EPI, F32>), grid, dim3(256), lds, st, a);                              \
    }
    if (epi == TEPI_NONE) SPS_TCONV_GO(TEPI_NONE)
    else if (epi == TEPI_STATS) SPS_TCONV_GO(TEPI_STATS)
    else SPS_TCONV_GO(TEPI_BWD)
#undef SPS_TCONV_GO
    return check_launch("tconv_kernel");
}

template <int RT, bool F32>
int tconv_launch_prec(const TConvArgs &a, int in_mode, int epi, hipStream_t st) {
    switch (in_mode) {
        case TIN_RAW: return tconv_launch_epi<RT, TIN_RAW, F32>(a, epi, st);
        case TIN_BNRELU: return tconv_launch_epi<RT, TIN_BNRELU, F32>(a, epi, st);
        case TIN_BNBWD: return tconv_launch_epi<RT, TIN_BNBWD, F32>(a, epi, st);
        default: return tconv_launch_epi<RT, TIN_BNBWD_POOL, F32>(a, epi, st);
    }
}

// 0 = split-fp16 (hi + lo halves, ~22-bit products), 1 = exact fp32: the arithmetic of every sps_tconv / sps_twgrad launch
std::atomic<int> g_train_f32{0};

template <int RT>
int tconv_launch_in(const TConvArgs &a, int in_mode, int epi, hipStream_t st) {
    return g_train_f32.load(std::memory_order_relaxed) ? tconv_launch_prec<RT, true>(a, in_mode, epi, st)
                                                       : tconv_launch_prec<RT, false>(a, in_mode, epi, st);
}

template <int NK, bool F32>
int twgrad_launch_prec(const TWgradArgs &a, int parts, hipStream_t st) {
    const int cop = 16 * divup(a.co, 16), cip = 16 * divup(a.ci, 16);
    const size_t op = (size_t)(cop + cip) * (F32 ? (128 * NK + 16) : 2 * (64 * NK + 16)), rest = (size_t)(cop + cip) * TP * 4 + 64;
    const bool db = 2 * op + rest <= 150 * 1024;
    const size_t lds = (db ? 2 : 1) * op + rest;
    static LdsLimitOnce raised[2];
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit(db ? (const void *)twgrad_kernel<NK, true, F32> : (const void *)twgrad_kernel<NK, false, F32>,
                                       150 * 1024, raised[db ? 1 : 0], "twgrad");
        if (rc != SPS_OK) return rc;
    }
    if (db) hipLaunchKernelGGL((twgrad_kernel<NK, true, F32>), dim3(parts), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((twgrad_kernel<NK, false, F32>), dim3(parts), dim3(512), lds, st, a);
    return SPS_OK;
}

template <int NK>
int twgrad_launch(const TWgradArgs &a, int parts, hipStream_t st) {
    return g_train_f32.load(std::memory_order_relaxed) ? twgrad_launch_prec<NK, true>(a, parts, st)
                                                       : twgrad_launch_prec<NK, false>(a, parts, st);
}

}  // namespace
}  // namespace sps

using namespace sps;

// Arithmetic of every sps_tconv / sps_twgrad launch (and of sps_mlp_train_forward / _backward, which compose them):
// 0 = split-fp16 (default), 1 = exact fp32 on v_mfma_f32_16x16x4_f32 -- the reference's arithmetic.  Returns the old mode.
extern "C" int sps_set_train_precision(int mode) {
    return g_train_f32.exchange(mode ? 1 : 0, std::memory_order_relaxed);
}

// Workgroups along the columns of sps_tconv = first dimension of its `partial` output ([parts][co][2] doubles).
extern "C" int sps_tconv_parts(int b, long long l, int co) {
    if (b <= 0 || l <= 0 || co <= 0) return 0;
    return tconv_grid_x(b, l, co);
}

extern "C" int sps_tconv(int b, int ci, int co, long long l, int in_mode, int epi_mode, int trans, const float *w, const float *in,
                         const float *in2, const float *gout, const unsigned char *arg, int nsample, int m, const float *pin,
                         float *out, const float *epi_y, const float *pout, double *partial, const float *amax_in,
                         float *amax_out, const float *wamax, int *overflow, sps_stream_t stream) {
    if (b < 0 || ci <= 0 || co <= 0 || l < 0 || ci > 4096 || co > 4096)
        return fail(SPS_ERR_INVALID, "tconv: bad shape b=%d ci=%d co=%d l=%lld", b, ci, co, l);
    if (b == 0 || l == 0) return SPS_OK;
    if (l % 64) return fail(SPS_ERR_INVALID, "tconv: l = %lld must be a multiple of 64", l);
    if (in_mode < TIN_RAW || in_mode > TIN_BNBWD_POOL || epi_mode < TEPI_NONE || epi_mode > TEPI_BWD)
        return fail(SPS_ERR_INVALID, "tconv: unknown mode");
    const bool f32 = g_train_f32.load(std::memory_order_relaxed) != 0;     // (exact fp32 scales nothing: wamax / amax_in unused)
    if (!w || (!wamax && !f32) || !out || (in_mode != TIN_BNBWD_POOL && !in) || (in_mode != TIN_RAW && !pin) || (in_mode >= TIN_BNBWD && !in2) ||
        (epi_mode != TEPI_NONE && !partial) || (epi_mode == TEPI_BWD && (!epi_y || !pout)) || (in_mode >= TIN_BNBWD && !amax_in && !f32))
        return fail(SPS_ERR_INVALID, "tconv: null pointer");
    if (in_mode == TIN_BNBWD_POOL && (!gout || !arg || nsample <= 0 || (nsample % 4) || m <= 0 || (long long)m * nsample != l))
        return fail(SPS_ERR_INVALID, "tconv: the pooled-gradient operand needs gout, arg, nsample %% 4 == 0 and m * nsample == l");
    hipStream_t st = as_stream(stream);
    // The weights of a workgroup's output rows live in LDS for the whole launch, which holds 288 input rows of them: wider
    // layers (IA-SSD layer 5: 512 and 1024 channels) run as K SLABS of 256 input rows, one launch each, every slab behind the
    // first adding to `out`; the statistics / BatchNorm-backward epilogue belongs to the last slab (the complete sums).
    const int slab = ci <= 288 ? ci : 256;
    for (int k0 = 0; k0 < ci; k0 += slab) {
        const int cs = ci - k0 < slab ? ci - k0 : slab;
        const bool last = k0 + cs >= ci;
        const size_t rowoff = (size_t)k0 * (size_t)l;
        TConvArgs a;
        a.b = b; a.ci = cs; a.co = co; a.S = (cs + 31) / 32; a.l = l; a.trans = trans ? 1 : 0;
        a.ci_total = ci; a.w_ld = ci; a.accum = k0 > 0 ? 1 : 0;
        a.w = trans ? w + (size_t)k0 * co : w + k0;
        const float *in_k = in ? in + rowoff : nullptr, *in2_k = in2 ? in2 + rowoff : nullptr;
        a.in = in_k ? in_k : in2_k; a.in2 = in2_k ? in2_k : in_k;
        a.gout = gout ? gout + (size_t)k0 * m : nullptr; a.arg = arg ? arg + (size_t)k0 * m : nullptr;
        a.ns = nsample > 0 ? nsample : 4; a.m = m;
        a.pin = pin ? pin + (size_t)k0 * TP : nullptr; a.out = out; a.epi_y = epi_y; a.pout = pout; a.partial = partial;
        a.amax_in = amax_in; a.amax_out = amax_out; a.wamax = wamax; a.overflow = overflow;
        const int epi = last ? epi_mode : TEPI_NONE;
        int rc;
        if (co <= 16) rc = tconv_launch_in<1>(a, in_mode, epi, st);
        else if (co <= 32) rc = tconv_launch_in<2>(a, in_mode, epi, st);
        else if (co <= 64 || a.S > 8) rc = tconv_launch_in<4>(a, in_mode, epi, st);   // (nine k-steps: 128 rows of weights would not fit the LDS)
        else rc = tconv_launch_in<8>(a, in_mode, epi, st);        // 128 rows per workgroup (blockIdx.z walks the rest)
        if (rc != SPS_OK) return rc;
    }
    return SPS_OK;
}

// out[k] = max |p_k[0 .. n_k)| for up to four arrays (the weight matrices of one grouped MLP): one launch, one workgroup each
namespace sps { namespace {
struct TAmaxArgs { const float *p[4]; long long n[4]; float *out; };
__global__ __launch_bounds__(256) void tamax_kernel(TAmaxArgs a) {
    __shared__ float scratch[8];
    const float *p = a.p[blockIdx.x];
    const long long n = a.n[blockIdx.x];
    float m = 0.f;
    auto take = [&](float v) { m = (v != v) ? INFINITY : fmaxf(m, fabsf(v)); };
    long long done = 0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {            // eight 16-byte loads in flight per thread
        const long long n4 = n >> 2;
        for (long long base = 0; base < n4; base += 256 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long i = base + u * 256 + threadIdx.x;
                v[u] = i < n4 ? *reinterpret_cast<const f32x4 *>(p + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { take(v[u][0]); take(v[u][1]); take(v[u][2]); take(v[u][3]); }
        }
        done = n4 << 2;
    }
    for (long long i = done + threadIdx.x; i < n; i += 256) take(p[i]);
    m = tblock_max(m, scratch, 4);
    if (threadIdx.x == 0) a.out[blockIdx.x] = m;
}
} }

extern "C" int sps_tamax4(int count, const float *p0, long long n0, const float *p1, long long n1, const float *p2, long long n2,
                          const float *p3, long long n3, float *out, sps_stream_t stream) {
    if (count < 1 || count > 4 || !out) return fail(SPS_ERR_INVALID, "tamax4: 1..4 arrays and an output");
    TAmaxArgs a;
    a.p[0] = p0; a.p[1] = p1; a.p[2] = p2; a.p[3] = p3; a.n[0] = n0; a.n[1] = n1; a.n[2] = n2; a.n[3] = n3; a.out = out;
    for (int k = 0; k < count; ++k)
        if (!a.p[k] || a.n[k] < 0) return fail(SPS_ERR_INVALID, "tamax4: array %d", k);
    hipLaunchKernelGGL(tamax_kernel, dim3(count), dim3(256), 0, as_stream(stream), a);
    return check_launch("tamax_kernel");
}

extern "C" int sps_tbn_finalize(int c, int nparts, double count, const double *partial, const float *gamma, const float *beta,
                                float eps, float momentum, float *running_mean, float *running_var, float *params,
                                long long *num_batches_tracked, sps_stream_t stream) {
    if (count <= 0.0) return fail(SPS_ERR_INVALID, "tbn_finalize: bad arguments");
    return sps_tbn_finalize_dc(c, nparts, count, nullptr, partial, gamma, beta, eps, momentum, running_mean, running_var, params,
                               num_batches_tracked, stream);
}

// count_dev (device double, may be NULL): when given it replaces `count` -- the element count of a SyncBatchNorm's GLOBAL
// batch, all-reduced together with the sums, so that ranks with different local batch sizes normalise alike
extern "C" int sps_tbn_finalize_dc(int c, int nparts, double count, const double *count_dev, const double *partial,
                                   const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                   float *running_var, float *params, long long *num_batches_tracked, sps_stream_t stream) {
    if (c <= 0 || nparts <= 0 || (!count_dev && count <= 0.0) || !partial || !params)
        return fail(SPS_ERR_INVALID, "tbn_finalize: bad arguments");
    hipLaunchKernelGGL(tbn_finalize_kernel, dim3(divup(c, 4)), dim3(256), 0, as_stream(stream), c, nparts, count, partial, gamma, beta,
                       eps, momentum, running_mean, running_var, params, num_batches_tracked, count_dev);
    return check_launch("tbn_finalize_kernel");
}

extern "C" int sps_tbn_bwd_finalize(int c, int nparts, double count, const double *partial, float *params, float *dgamma,
                                    float *dbeta, sps_stream_t stream) {
    if (count <= 0.0) return fail(SPS_ERR_INVALID, "tbn_bwd_finalize: bad arguments");
    return sps_tbn_bwd_finalize_dc(c, nparts, count, nullptr, partial, params, dgamma, dbeta, stream);
}

extern "C" int sps_tbn_bwd_finalize_dc(int c, int nparts, double count, const double *count_dev, const double *partial,
                                       float *params, float *dgamma, float *dbeta, sps_stream_t stream) {
    if (c <= 0 || nparts <= 0 || (!count_dev && count <= 0.0) || !partial || !params)
        return fail(SPS_ERR_INVALID, "tbn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(tbn_bwd_finalize_kernel, dim3(divup(c, 4)), dim3(256), 0, as_stream(stream), c, nparts,
                       count_dev ? 0.0 : 1.0 / count, partial, params, dgamma, dbeta, count_dev);
    return check_launch("tbn_bwd_finalize_kernel");
}

extern "C" int sps_tpool_fwd(int b, int c, int m, int nsample, const float *y, const float *params, float *out,
                             unsigned char *arg, float *yarg, sps_stream_t stream) {
    if (b < 0 || c <= 0 || m < 0 || nsample <= 0) return fail(SPS_ERR_INVALID, "tpool_fwd: bad shape");
    if (b == 0 || m == 0) return SPS_OK;
    if (nsample != 4 && nsample != 8 && nsample != 16 && nsample != 32 && nsample != 64)
        return fail(SPS_ERR_INVALID, "tpool_fwd: nsample %d not in {4, 8, 16, 32, 64}", nsample);
    if (!y || !params || !out || !arg || !yarg) return fail(SPS_ERR_INVALID, "tpool_fwd: null pointer");
    const long long rows = (long long)b * c * m;
    const int G = nsample / 4;
    const long long g = (rows * G + 255) / 256;
    if (g > 0x7fffffffLL) return fail(SPS_ERR_INVALID, "tpool_fwd: too many rows");
    hipStream_t st = as_stream(stream);
#define SPS_TPOOL(GG) hipLaunchKernelGGL(tpool_fwd_kernel<GG>, dim3((unsigned)g), dim3(256), 0, st, rows, c, m, y, params, out, arg, yarg)
    if (G == 1) SPS_TPOOL(1);
    else if (G == 2) SPS_TPOOL(2);
    else if (G == 4) SPS_TPOOL(4);
    else if (G == 8) SPS_TPOOL(8);
    else SPS_TPOOL(16);
#undef SPS_TPOOL
    return check_launch("tpool_fwd_kernel");
}

// partial: (b, c, 2) doubles = `b` parts for sps_tbn_bwd_finalize
extern "C" int sps_tpool_bwd_stats(int b, int c, int m, const float *yarg, const float *gout, const float *params, double *partial,
                                   float *amax_out, sps_stream_t stream) {
    if (b <= 0 || c <= 0 || m <= 0 || b > 65535) return fail(SPS_ERR_INVALID, "tpool_bwd_stats: bad shape");
    if (!yarg || !gout || !params || !partial) return fail(SPS_ERR_INVALID, "tpool_bwd_stats: null pointer");
    hipLaunchKernelGGL(tpool_bwd_stats_kernel, dim3(c, b), dim3(256), 0, as_stream(stream), c, m, yarg, gout, params, partial, amax_out);
    return check_launch("tpool_bwd_stats_kernel");
}

extern "C" int sps_tbn_apply_relu(int b, int c, long long l, const float *y, const float *params, float *out, sps_stream_t stream) {
    if (b < 0 || c <= 0 || l < 0 || (l % 4) || b > 65535) return fail(SPS_ERR_INVALID, "tbn_apply_relu: bad shape b=%d c=%d l=%lld", b, c, l);
    if (b == 0 || l == 0) return SPS_OK;
    if (!y || !params || !out) return fail(SPS_ERR_INVALID, "tbn_apply_relu: null pointer");
    hipLaunchKernelGGL(tbn_apply_relu_kernel, dim3(c, b), dim3(256), 0, as_stream(stream), c, l, y, params, out);
    return check_launch("tbn_apply_relu_kernel");
}

// partial: (b, c, 2) doubles = `b` parts for sps_tbn_bwd_finalize
extern "C" int sps_tbn_bwd_stats(int b, int c, long long l, const float *y, const float *dA, const float *params, double *partial,
                                 float *amax_out, sps_stream_t stream) {
    if (b <= 0 || c <= 0 || l <= 0 || (l % 4) || b > 65535) return fail(SPS_ERR_INVALID, "tbn_bwd_stats: bad shape b=%d c=%d l=%lld", b, c, l);
    if (!y || !dA || !params || !partial) return fail(SPS_ERR_INVALID, "tbn_bwd_stats: null pointer");
    hipLaunchKernelGGL(tbn_bwd_stats_kernel, dim3(c, b), dim3(256), 0, as_stream(stream), c, l, y, dA, params, partial, amax_out);
    return check_launch("tbn_bwd_stats_kernel");
}

static int twgrad_nk(int co, int ci, long long l) {
    const int rows = 16 * divup(co > ci ? co : ci, 16);
    if (rows <= 32 && l % 256 == 0) return 8;
    if (rows <= 64 && l % 128 == 0) return 4;
    if (rows <= 128 && l % 64 == 0) return 2;
    return 1;
}
static int twgrad_parts(int b, int co, int ci, long long l) {
    const long long stages = (long long)b * (l / (32 * twgrad_nk(co, ci, l)));
    return (int)(stages < 256 ? stages : 256);
}

extern "C" long long sps_twgrad_workspace_floats(int b, int co, int ci, long long l) {
    if (b <= 0 || co <= 0 || ci <= 0 || l <= 0) return 0;
    const int cb = co < 256 ? co : 256, ib = ci < 256 ? ci : 256;     // (wider layers run block by block through one workspace)
    return (long long)twgrad_parts(b, cb, ib, l) * (16 * divup(cb, 16)) * (16 * divup(ib, 16));
}

extern "C" int sps_twgrad(int b, int co, int ci, long long l, int dmode, int xmode, const float *dA, const float *y,
                          const float *gout, const unsigned char *arg, int nsample, int m, const float *pd, const float *x,
                          const float *px, const float *amax_in, float *dw, float *work, int *overflow, sps_stream_t stream) {
    if (b <= 0 || co <= 0 || ci <= 0 || l <= 0 || co > 4096 || ci > 4096)
        return fail(SPS_ERR_INVALID, "twgrad: bad shape b=%d co=%d ci=%d l=%lld", b, co, ci, l);
    if (l % 32) return fail(SPS_ERR_INVALID, "twgrad: l = %lld must be a multiple of 32", l);
    if ((dmode != TIN_BNBWD && dmode != TIN_BNBWD_POOL) || (xmode != TIN_RAW && xmode != TIN_BNRELU))
        return fail(SPS_ERR_INVALID, "twgrad: unknown mode");
    if (!y || !pd || !x || !dw || !work || (!amax_in && !g_train_f32.load(std::memory_order_relaxed)) || (dmode == TIN_BNBWD && !dA) ||
        (xmode == TIN_BNRELU && !px))
        return fail(SPS_ERR_INVALID, "twgrad: null pointer");
    if (dmode == TIN_BNBWD_POOL && (!gout || !arg || nsample <= 0 || (nsample % 4) || m <= 0 || (long long)m * nsample != l))
        return fail(SPS_ERR_INVALID, "twgrad: the pooled-gradient operand needs gout, arg, nsample %% 4 == 0 and m * nsample == l");
    hipStream_t st = as_stream(stream);
    // both operands of a stage live in LDS, which holds 256 rows of each: wider layers (IA-SSD layer 5) run as blocks of at
    // most 256 x 256 of dW, one launch pair each through the same workspace (the stream orders them), in a fixed order
    for (int r0 = 0; r0 < co; r0 += 256)
        for (int c0 = 0; c0 < ci; c0 += 256) {
            const int cb = co - r0 < 256 ? co - r0 : 256, ib = ci - c0 < 256 ? ci - c0 : 256;
            TWgradArgs a;
            a.b = b; a.co = cb; a.ci = ib; a.co_total = co; a.ci_total = ci; a.l = l; a.dmode = dmode; a.xmode = xmode;
            a.dA = dA ? dA + (size_t)r0 * l : nullptr; a.y = y + (size_t)r0 * l;
            a.gout = gout ? gout + (size_t)r0 * m : nullptr; a.arg = arg ? arg + (size_t)r0 * m : nullptr;
            a.ns = nsample > 0 ? nsample : 4; a.m = m; a.pd = pd + (size_t)r0 * TP; a.x = x + (size_t)c0 * l;
            a.px = px ? px + (size_t)c0 * TP : nullptr; a.partial = work; a.amax_in = amax_in; a.overflow = overflow;
            const int cop = 16 * divup(cb, 16), cip = 16 * divup(ib, 16), nk = twgrad_nk(cb, ib, l), parts = twgrad_parts(b, cb, ib, l);
            const int rc = nk == 8 ? twgrad_launch<8>(a, parts, st)
                                   : (nk == 4 ? twgrad_launch<4>(a, parts, st) : (nk == 2 ? twgrad_launch<2>(a, parts, st) : twgrad_launch<1>(a, parts, st)));
            if (rc != SPS_OK) return rc;
            hipLaunchKernelGGL(twgrad_reduce_kernel, dim3(divup(cb * ib, 32)), dim3(256), 0, st, cb, ib, cip, cop, parts, work,
                               dw + (size_t)r0 * ci + c0, ci);
        }
    return check_launch("twgrad_kernel");
}

// ---- one call per grouped MLP (host side of a training step) ------------------------------------------------------------
// The launches above, composed in C: a training step of IASSD_Backbone issues ~220 of them from Python (24-argument ctypes
// calls, a torch.empty per output), ~25 us of host time each, and is host-bound.  The caller allocates every buffer once
// (sizes: sps_mlp_train_partial_doubles / sps_twgrad_workspace_floats) and makes ONE call for the forward of a grouped MLP
// with its max-pool, one for its backward.  Same kernels, same order, same results as the launch-by-launch form
// (pointnet2_modules._GroupedMLPPoolTrain without it; SyncBatchNorm keeps that form: its all-reduces sit between the launches).
extern "C" long long sps_struct_size(int which) {
    switch (which) {
    case 0: return (long long)sizeof(sps_mlp_train_desc);
    default: return -1;
    }
}

extern "C" long long sps_mlp_train_partial_doubles(const sps_mlp_train_desc *d) {
    if (!d || d->n < 1 || d->n > 4) return 0;
    const long long l = (long long)d->m * (d->ns > 0 ? d->ns : 1);
    long long need = (long long)d->b * d->c[d->n] * 2;
    for (int k = 0; k <= d->n; ++k) {
        const long long v = (long long)sps_tconv_parts(d->b, l, d->c[k]) * d->c[k] * 2;
        need = v > need ? v : need;
    }
    return need;
}

static int mlp_train_check(const sps_mlp_train_desc *d, const char *what) {
    if (!d || d->n < 1 || d->n > 4 || d->b <= 0 || d->m <= 0 || d->ns < 0)
        return fail(SPS_ERR_INVALID, "%s: bad descriptor (n, b, m, ns)", what);
    for (int k = 0; k <= d->n; ++k)
        if (d->c[k] <= 0) return fail(SPS_ERR_INVALID, "%s: c[%d] = %d", what, k, d->c[k]);
    for (int k = 0; k < d->n; ++k)
        if (!d->w[k] || !d->y[k] || !d->params[k]) return fail(SPS_ERR_INVALID, "%s: null pointer at layer %d", what, k);
    if (!d->x || !d->partial || (!d->wamax && !g_train_f32.load(std::memory_order_relaxed)))
        return fail(SPS_ERR_INVALID, "%s: null pointer", what);
    return SPS_OK;
}

extern "C" int sps_mlp_train_forward(const sps_mlp_train_desc *d, sps_stream_t stream) {
    int rc = mlp_train_check(d, "mlp_train_forward");
    if (rc != SPS_OK) return rc;
    // nsample = 0: the same stack WITHOUT a pool on (b, c, m) tensors -- an aggregation / confidence / vote stack
    // ([Conv1d, BatchNorm1d, ReLU] x n, pointnet2_modules.py:213-245): out = relu(bn_n(...)) dense, no arg / yarg
    const bool pool = d->ns > 0;
    if (!d->out || (pool && (!d->arg || !d->yarg))) return fail(SPS_ERR_INVALID, "mlp_train_forward: null output");
    const int n = d->n;
    const long long l = (long long)d->m * (pool ? d->ns : 1);
    const double count = (double)d->b * (double)l;
    const bool f32 = g_train_f32.load(std::memory_order_relaxed) != 0;
    const float *wp[4] = {nullptr, nullptr, nullptr, nullptr};
    long long wn[4] = {0, 0, 0, 0};
    for (int k = 0; k < n; ++k) { wp[k] = d->w[k]; wn[k] = (long long)d->c[k + 1] * d->c[k]; }
    if (!f32) {       // (the split form scales the weights by a power of two derived from their largest magnitude)
        rc = sps_tamax4(n, wp[0], wn[0], wp[1], wn[1], wp[2], wn[2], wp[3], wn[3], d->wamax, stream);
        if (rc != SPS_OK) return rc;
    }
    const float *operand = d->x, *pin = nullptr;
    for (int k = 0; k < n; ++k) {
        rc = sps_tconv(d->b, d->c[k], d->c[k + 1], l, k ? TIN_BNRELU : TIN_RAW, TEPI_STATS, 0, d->w[k], operand, nullptr, nullptr,
                       nullptr, 0, 0, pin, d->y[k], nullptr, nullptr, d->partial, nullptr, nullptr, f32 ? nullptr : d->wamax + k, d->overflow,
                       stream);
        if (rc != SPS_OK) return rc;
        rc = sps_tbn_finalize_dc(d->c[k + 1], sps_tconv_parts(d->b, l, d->c[k + 1]), count, nullptr, d->partial, d->gamma[k], d->beta[k],
                                 d->eps[k], d->momentum[k], d->running_mean[k], d->running_var[k], d->params[k],
                                 d->num_batches_tracked[k], stream);
        if (rc != SPS_OK) return rc;
        operand = d->y[k];
        pin = d->params[k];
    }
    if (!pool) return sps_tbn_apply_relu(d->b, d->c[n], l, d->y[n - 1], d->params[n - 1], d->out, stream);
    return sps_tpool_fwd(d->b, d->c[n], d->m, d->ns, d->y[n - 1], d->params[n - 1], d->out, d->arg, d->yarg, stream);
}

extern "C" int sps_mlp_train_backward(const sps_mlp_train_desc *d, sps_stream_t stream) {
    int rc = mlp_train_check(d, "mlp_train_backward");
    if (rc != SPS_OK) return rc;
    const int n = d->n;
    const bool pool = d->ns > 0;
    const bool f32 = g_train_f32.load(std::memory_order_relaxed) != 0;
    if (!d->gout || (pool && (!d->arg || !d->yarg)) || !d->amax) return fail(SPS_ERR_INVALID, "mlp_train_backward: null pointer");
    for (int k = 0; k < n; ++k) {
        if (!d->dgamma[k] || !d->dbeta[k] || (k > 0 && !d->dA[k]) || (d->dw[k] && !d->work))
            return fail(SPS_ERR_INVALID, "mlp_train_backward: null pointer at layer %d", k);
    }
    const long long l = (long long)d->m * (pool ? d->ns : 1);
    const double count = (double)d->b * (double)l;
    if (hipMemsetAsync(d->amax, 0, (size_t)n * sizeof(float), as_stream(stream)) != hipSuccess)
        return fail(SPS_ERR_LAUNCH, "mlp_train_backward: hipMemsetAsync");
    // the last layer's BatchNorm-backward sums from the pooled gradient alone (its dA is never materialised) -- or, for a
    // stack without a pool, from the dense incoming gradient
    rc = pool ? sps_tpool_bwd_stats(d->b, d->c[n], d->m, d->yarg, d->gout, d->params[n - 1], d->partial, d->amax + (n - 1), stream)
              : sps_tbn_bwd_stats(d->b, d->c[n], l, d->y[n - 1], d->gout, d->params[n - 1], d->partial, d->amax + (n - 1), stream);
    if (rc != SPS_OK) return rc;
    rc = sps_tbn_bwd_finalize_dc(d->c[n], d->b, count, nullptr, d->partial, d->params[n - 1], d->dgamma[n - 1], d->dbeta[n - 1], stream);
    if (rc != SPS_OK) return rc;
    for (int k = n - 1; k >= 0; --k) {
        const bool routed = pool && k == n - 1;      // the incoming gradient is the pooled one, routed by the arg-max in the operand load
        const float *dA_in = routed ? nullptr : (k == n - 1 ? d->gout : d->dA[k + 1]);
        const float *g = routed ? d->gout : nullptr;
        const unsigned char *ar = routed ? d->arg : nullptr;
        const int ns = routed ? d->ns : 0, mm = routed ? d->m : 0;
        if (d->dw[k]) {
            rc = sps_twgrad(d->b, d->c[k + 1], d->c[k], l, routed ? TIN_BNBWD_POOL : TIN_BNBWD, k ? TIN_BNRELU : TIN_RAW, dA_in, d->y[k],
                            g, ar, ns, mm, d->params[k], k ? d->y[k - 1] : d->x, k ? d->params[k - 1] : nullptr, d->amax + k, d->dw[k],
                            d->work, d->overflow, stream);
            if (rc != SPS_OK) return rc;
        }
        const int mode = routed ? TIN_BNBWD_POOL : TIN_BNBWD;
        if (k > 0) {
            rc = sps_tconv(d->b, d->c[k + 1], d->c[k], l, mode, TEPI_BWD, 1, d->w[k], dA_in, d->y[k], g, ar, ns, mm, d->params[k], d->dA[k],
                           d->y[k - 1], d->params[k - 1], d->partial, d->amax + k, d->amax + (k - 1), f32 ? nullptr : d->wamax + k, d->overflow,
                           stream);
            if (rc != SPS_OK) return rc;
            rc = sps_tbn_bwd_finalize_dc(d->c[k], sps_tconv_parts(d->b, l, d->c[k]), count, nullptr, d->partial, d->params[k - 1],
                                         d->dgamma[k - 1], d->dbeta[k - 1], stream);
            if (rc != SPS_OK) return rc;
        } else if (d->dA[0]) {
            rc = sps_tconv(d->b, d->c[1], d->c[0], l, mode, TEPI_NONE, 1, d->w[0], dA_in, d->y[0], g, ar, ns, mm, d->params[0], d->dA[0],
                           nullptr, nullptr, nullptr, d->amax, nullptr, f32 ? nullptr : d->wamax, d->overflow, stream);
            if (rc != SPS_OK) return rc;
        }
    }
    return SPS_OK;
}
