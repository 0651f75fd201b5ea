// dense_edge_conv_bwd.hip -- gradient of DenseEdgeConv (surface_feature.py:45-116) for TRAINING: one kernel recomputes
// the convolution of a centre point and back-propagates through it, so that neither the (B, N, K, 3d) edge features nor the
// three concatenated activations of the op-by-op form ever exist.  FeatureExtraction's forward + backward took 52 ms at
// 8 x 16 384 points through torch ops (cat / Linear / max and their backward kernels); the stack of SA layers next to it in
// PAGNet_Backbone takes 12.6 ms.
//
// A wave owns one centre p; its K = 16 neighbours are the 16 columns of v_mfma_f32_16x16x4_f32 (as in the forward).
//   1. forward recompute (42 MFMAs): z1, z2, z3, the ReLU masks, the row maxima and the column that attains each.
//   2. data gradient: with dZ of a layer in the accumulator layout (lane (q, c): rows 4q..4q+3 of column c) it is directly
//      the B operand of W^T dZ, W^T fragments packed by the host: dY2 = W3a^T dZ3 + G2, dY1 = W3b^T dZ3 + W2a^T dZ2 + G1,
//      dX_i (centre) = sum_c [W3c^T dZ3 + W2c^T dZ2 + (W1a - W1c)^T dZ1] + dOut[36:60],  dX_j (neighbours) = (W1b + W1c)^T dZ1
//      (44 MFMAs; the difference-only layer: -W1^T / W1^T).  The neighbour part goes to a (B, 24, N*16) buffer and is
//      reduced by the LDS-row scatter kernel of group_gather.hip instead of 50 M global atomics.
//   3. weight gradient dW = sum over all columns of dZ (x) input: an MFMA whose reduction index is the COLUMN, i.e. both
//      operands transposed with respect to how the wave holds them; they are written to LDS ([row][17]) and read back in
//      the other orientation (52 MFMAs, 13 persistent accumulator tiles per wave).  A row of ones rides along as channel
//      24 of the 24-wide inputs, so the bias gradients fall out of the same products.  Per-workgroup partial sums are
//      written out and added by a second, tiny kernel in a fixed order (no atomics, reproducible).
// fp32 throughout.  Ties in the max over K go to the first column (any choice gives the same gradient up to summation
// order: tied columns of a padded ball are the same neighbour).
#include "sps_common.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int DB_D = 24, DB_K = 16, DB_G = 12, DB_OUT = 60, DB_P = 17, DB_WAVES = 4;
constexpr int DB_WT = 44;  // transposed-weight k-steps: t3a 4 | t3b 4 | t2a 4 | t3c 8 | t2c 8 | t1c 8 | t1n 8
constexpr int DB_LDS_WAVE = 5 * 16 * DB_P + 3 * 32 * DB_P;  // Z[3], Y[2] (16 rows), X[3] (32 rows)

struct DbArgs {
    int n;
    long long units;
    const float *x;    // (B, N, 24)
    const int *idx;    // (B, N, 16)
    const float *g;    // dOut (B, N, 60)
    const float *wf;   // forward fragments [w1 (18 | 6) | w2 (10) | w3 (14)] x 64 lanes
    const float *wt;   // transposed fragments, DB_WT x 64 lanes
    const float *b1, *b2, *b3;  // padded to 16
    float *dxc;        // (B, N, 24): gradient through the centre role (+ the pass-through channels)
    float *dxn;        // (B, 24, N * 16): gradient through the neighbour role, per (centre, column)
    float *partial;    // (workgroups, tiles * 256) weight-gradient partial sums
};

__device__ __forceinline__ f32x4 db_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float db_row_max(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ float db_row_sum(float v) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
// is this lane's column the FIRST of its 16-lane row whose value equals the row maximum?
__device__ __forceinline__ bool db_first_max(float v, float rowmax, int q, int c) {
    const unsigned long long eq = __ballot(v == rowmax);
    const unsigned row = (unsigned)(eq >> (16 * q)) & 0xFFFFu;
    return row != 0u && (__ffs(row) - 1) == c;
}

template <bool REL>
__global__ __launch_bounds__(64 * DB_WAVES, 2) void dense_edge_conv_bwd_kernel(DbArgs a) {
    constexpr int KS1 = REL ? 6 : 18;
    constexpr int TILES = REL ? 9 : 13;
    extern __shared__ float db_lds[];
    float *wts = db_lds;                                        // DB_WT x 64
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wave = threadIdx.x >> 6;
    float *Z = db_lds + DB_WT * 64 + wave * DB_LDS_WAVE;        // [3][16][P]
    float *Y = Z + 3 * 16 * DB_P;                               // [2][16][P]
    float *X = Y + 2 * 16 * DB_P;                               // [3][32][P]: xn, xd, xc (row 24 = ones)
    for (int e = threadIdx.x; e < DB_WT * 64; e += blockDim.x) wts[e] = a.wt[e];
    for (int e = lane; e < 3 * 32 * DB_P; e += 64) {
        const int row = (e / DB_P) % 32;
        X[e] = row == 24 ? 1.f : 0.f;
    }
    float w1r[KS1], w2r[10], w3r[14];
#pragma unroll
    for (int k = 0; k < KS1; ++k) w1r[k] = a.wf[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 10; ++k) w2r[k] = a.wf[(KS1 + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < 14; ++k) w3r[k] = a.wf[(KS1 + 10 + k) * 64 + lane];
    const f32x4 b1v = *reinterpret_cast<const f32x4 *>(a.b1 + 4 * q);
    const f32x4 b2v = *reinterpret_cast<const f32x4 *>(a.b2 + 4 * q);
    const f32x4 b3v = *reinterpret_cast<const f32x4 *>(a.b3 + 4 * q);
    __syncthreads();
    const float *t3a = wts, *t3b = wts + 4 * 64, *t2a = wts + 8 * 64, *t3c = wts + 12 * 64, *t2c = wts + 20 * 64,
                *t1c = wts + 28 * 64, *t1n = wts + 36 * 64;

    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long long nwaves = (long long)gridDim.x * DB_WAVES;
    for (long long p = (long long)blockIdx.x * DB_WAVES + wave; p < a.units; p += nwaves) {
        const long long scene = p / a.n;
        const int nb = a.idx[p * DB_K + c];
        const f32x2 *xcp = reinterpret_cast<const f32x2 *>(a.x + p * DB_D + 6 * q);
        const f32x2 *xnp = reinterpret_cast<const f32x2 *>(a.x + (scene * a.n + nb) * DB_D + 6 * q);
        float xc[6], xn[6], xd[6];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const f32x2 u = xcp[h], v = xnp[h];
            xc[2 * h] = u[0]; xc[2 * h + 1] = u[1];
            xn[2 * h] = v[0]; xn[2 * h + 1] = v[1];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) xd[j] = xn[j] - xc[j];
        // ---- 1. forward recompute ----
        f32x4 y1 = b1v;
        if (!REL) {
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = db_mfma(w1r[j], xc[j], y1);
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = db_mfma(w1r[6 + j], xn[j], y1);
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = db_mfma(w1r[12 + j], xd[j], y1);
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = db_mfma(w1r[j], xd[j], y1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) y1[r] = fmaxf(y1[r], 0.f);
        f32x4 y2 = b2v;
#pragma unroll
        for (int r = 0; r < 4; ++r) y2 = db_mfma(w2r[r], y1[r], y2);
#pragma unroll
        for (int j = 0; j < 6; ++j) y2 = db_mfma(w2r[4 + j], xc[j], y2);
#pragma unroll
        for (int r = 0; r < 4; ++r) y2[r] = fmaxf(y2[r], 0.f);
        f32x4 y3 = b3v;
#pragma unroll
        for (int r = 0; r < 4; ++r) y3 = db_mfma(w3r[r], y2[r], y3);
#pragma unroll
        for (int r = 0; r < 4; ++r) y3 = db_mfma(w3r[4 + r], y1[r], y3);
#pragma unroll
        for (int j = 0; j < 6; ++j) y3 = db_mfma(w3r[8 + j], xc[j], y3);
        // ---- gradient of the pooled outputs, routed to the first maximal column of each row ----
        const float *gp = a.g + p * DB_OUT;
        f32x4 dz3 = {0.f, 0.f, 0.f, 0.f}, dy2 = dz3, dy1 = dz3;
        if (q < 3) {
            const f32x4 g3 = *reinterpret_cast<const f32x4 *>(gp + 4 * q);
            const f32x4 g2 = *reinterpret_cast<const f32x4 *>(gp + DB_G + 4 * q);
            const f32x4 g1 = *reinterpret_cast<const f32x4 *>(gp + 2 * DB_G + 4 * q);
            dz3 = g3; dy2 = g2; dy1 = g1;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool s3 = db_first_max(y3[r], db_row_max(y3[r]), q, c);
            const bool s2 = db_first_max(y2[r], db_row_max(y2[r]), q, c);
            const bool s1 = db_first_max(y1[r], db_row_max(y1[r]), q, c);
            dz3[r] = s3 ? dz3[r] : 0.f;
            dy2[r] = s2 ? dy2[r] : 0.f;
            dy1[r] = s1 ? dy1[r] : 0.f;
        }
        // ---- 2. data gradient ----
#pragma unroll
        for (int r = 0; r < 4; ++r) dy2 = db_mfma(t3a[r * 64 + lane], dz3[r], dy2);
        f32x4 dz2;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz2[r] = y2[r] > 0.f ? dy2[r] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) dy1 = db_mfma(t3b[r * 64 + lane], dz3[r], dy1);
#pragma unroll
        for (int r = 0; r < 4; ++r) dy1 = db_mfma(t2a[r * 64 + lane], dz2[r], dy1);
        f32x4 dz1;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz1[r] = y1[r] > 0.f ? dy1[r] : 0.f;
        const long long centre = p - scene * a.n;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 dc = {0.f, 0.f, 0.f, 0.f}, dn = dc;
#pragma unroll
            for (int r = 0; r < 4; ++r) dc = db_mfma(t3c[(t * 4 + r) * 64 + lane], dz3[r], dc);
#pragma unroll
            for (int r = 0; r < 4; ++r) dc = db_mfma(t2c[(t * 4 + r) * 64 + lane], dz2[r], dc);
#pragma unroll
            for (int r = 0; r < 4; ++r) dc = db_mfma(t1c[(t * 4 + r) * 64 + lane], dz1[r], dc);
#pragma unroll
            for (int r = 0; r < 4; ++r) dn = db_mfma(t1n[(t * 4 + r) * 64 + lane], dz1[r], dn);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * t + 4 * q + r;
                const float s = db_row_sum(dc[r]);
                if (ch < DB_D) {
                    a.dxn[((scene * DB_D + ch) * a.n + centre) * DB_K + c] = dn[r];
                    if (c == 0) a.dxc[p * DB_D + ch] = s + gp[3 * DB_G + ch];
                }
            }
        }
        // ---- 3. weight gradient: stage both operands in LDS, read them back column-major ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * q + r;
            Z[(0 * 16 + row) * DB_P + c] = dz1[r];
            Z[(1 * 16 + row) * DB_P + c] = dz2[r];
            Z[(2 * 16 + row) * DB_P + c] = dz3[r];
            Y[(0 * 16 + row) * DB_P + c] = y1[r];
            Y[(1 * 16 + row) * DB_P + c] = y2[r];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int row = 6 * q + j;
            X[(0 * 32 + row) * DB_P + c] = xn[j];
            X[(1 * 32 + row) * DB_P + c] = xd[j];
            X[(2 * 32 + row) * DB_P + c] = xc[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rd = c * DB_P + q;  // row (lane & 15), column 4 ks + q
        float za[3][4];
#pragma unroll
        for (int z = 0; z < 3; ++z)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) za[z][ks] = Z[z * 16 * DB_P + rd + 4 * ks];
        auto tile = [&](const float *base, int z, f32x4 &dst) {
            float bb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bb[ks] = base[rd + 4 * ks];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dst = db_mfma(za[z][ks], bb[ks], dst);
        };
        const float *Y1 = Y, *Y2 = Y + 16 * DB_P;
        const float *Xn0 = X, *Xn1 = X + 16 * DB_P, *Xd0 = X + 32 * DB_P, *Xd1 = X + 48 * DB_P, *Xc0 = X + 64 * DB_P,
                    *Xc1 = X + 80 * DB_P;
        tile(Y2, 2, acc[0]); tile(Y1, 2, acc[1]); tile(Xc0, 2, acc[2]); tile(Xc1, 2, acc[3]);   // dW3
        tile(Y1, 1, acc[4]); tile(Xc0, 1, acc[5]); tile(Xc1, 1, acc[6]);                        // dW2
        if (!REL) {
            tile(Xc0, 0, acc[7]); tile(Xc1, 0, acc[8]); tile(Xn0, 0, acc[9]); tile(Xn1, 0, acc[10]);
            tile(Xd0, 0, acc[11]); tile(Xd1, 0, acc[12]);
        } else {
            tile(Xd0, 0, acc[7]); tile(Xd1, 0, acc[8]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- per-workgroup partial sums of the weight gradients (added across waves through LDS, then written) ----
    __syncthreads();
    float *red = db_lds + DB_WT * 64;  // the staging area is free now: TILES x 256 floats
    for (int e = threadIdx.x; e < TILES * 256; e += blockDim.x) red[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < DB_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[t * 256 + (4 * q + r) * 16 + c] += acc[t][r];
        }
        __syncthreads();
    }
    float *out = a.partial + (size_t)blockIdx.x * TILES * 256;
    for (int e = threadIdx.x; e < TILES * 256; e += blockDim.x) out[e] = red[e];
}

// sum of the workgroups' partial tiles, fixed order: one wave per element, lane l adds blocks l, l + 64, ... and the 64
// lane sums meet in a butterfly (the order does not depend on timing: reproducible)
__global__ __launch_bounds__(256) void dec_bwd_reduce_kernel(int blocks, int count, const float *__restrict__ partial, float *__restrict__ out) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= count) return;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64) s += partial[(size_t)b * count + e];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[e] = s;
}

// ---- gradient of the row-wise Linear (+ ReLU) of FCLayer (sps_linear_rows): dX = dY' W, dW = dY'^T X, db = sum dY' -------
// (torch hands the weight gradient, a 24 x cin product over 131 072 rows, to a GEMM kernel that takes 0.4 ms for it.)
// A workgroup stages 64 rows of x and dY' (= dY where the output was positive, if ReLU) in LDS; four threads per row form
// dX, every thread accumulates its share of the 24 x cin + 24 weight / bias gradient entries over the 64 rows; workgroup
// partials are summed by dec_bwd_reduce_kernel in a fixed order.
constexpr int LB_ROWS = 64, LB_COUT = 24, LB_MAX_CIN = 64, LB_SLOTS = (LB_COUT * LB_MAX_CIN + 255) / 256;
__global__ __launch_bounds__(256) void linear_rows24_bwd_kernel(long long rows, int cin, const float *__restrict__ x,
                                                                const float *__restrict__ y, const float *__restrict__ dy,
                                                                const float *__restrict__ w, int relu, float *__restrict__ dx,
                                                                float *__restrict__ partial) {
    __shared__ float ws[LB_COUT * LB_MAX_CIN];          // [o][i]
    __shared__ float xs[LB_ROWS * (LB_MAX_CIN + 1)];
    __shared__ float ds[LB_ROWS * (LB_COUT + 1)];
    const int px = cin + 1, pd = LB_COUT + 1;
    for (int e = threadIdx.x; e < LB_COUT * cin; e += blockDim.x) ws[e] = w[e];
    float acc[LB_SLOTS], accb = 0.f;
#pragma unroll
    for (int k = 0; k < LB_SLOTS; ++k) acc[k] = 0.f;
    const int entries = LB_COUT * cin;
    for (long long r0 = (long long)blockIdx.x * LB_ROWS; r0 < rows; r0 += (long long)gridDim.x * LB_ROWS) {
        const int here = (rows - r0 < LB_ROWS) ? (int)(rows - r0) : LB_ROWS;
        __syncthreads();
        for (int e = threadIdx.x; e < LB_ROWS * cin; e += blockDim.x) {
            const int r = e / cin, i = e - r * cin;
            xs[r * px + i] = r < here ? x[r0 * cin + e] : 0.f;
        }
        for (int e = threadIdx.x; e < LB_ROWS * LB_COUT; e += blockDim.x) {
            const int r = e / LB_COUT, o = e - r * LB_COUT;
            float g = 0.f;
            if (r < here) {
                g = dy[r0 * LB_COUT + e];
                if (relu && !(y[r0 * LB_COUT + e] > 0.f)) g = 0.f;
            }
            ds[r * pd + o] = g;
        }
        __syncthreads();
        {   // dX: four threads per row, outputs part, part + 4, ...
            const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
            if (r < here) {
                for (int i = part; i < cin; i += 4) {
                    float s = 0.f;
#pragma unroll
                    for (int o = 0; o < LB_COUT; ++o) s = __builtin_fmaf(ds[r * pd + o], ws[o * cin + i], s);
                    dx[(r0 + r) * cin + i] = s;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < LB_SLOTS; ++k) {
            const int e = threadIdx.x + 256 * k;
            if (e < entries) {
                const int o = e / cin, i = e - o * cin;
                float s = acc[k];
                for (int r = 0; r < LB_ROWS; ++r) s = __builtin_fmaf(ds[r * pd + o], xs[r * px + i], s);
                acc[k] = s;
            }
        }
        if (threadIdx.x < LB_COUT) {
            float s = accb;
            for (int r = 0; r < LB_ROWS; ++r) s += ds[r * pd + threadIdx.x];
            accb = s;
        }
    }
    float *out = partial + (size_t)blockIdx.x * (entries + LB_COUT);
#pragma unroll
    for (int k = 0; k < LB_SLOTS; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (e < entries) out[e] = acc[k];
    }
    if (threadIdx.x < LB_COUT) out[entries + threadIdx.x] = accb;
}

}  // namespace sps

extern "C" int sps_dense_edge_conv_bwd_blocks(void) { return 512; }

extern "C" int sps_dense_edge_conv_bwd(int b, int n, int d, int k, int growth, int relative_only, const float *x, const int *idx,
                                       const float *grad_out, const float *w_fwd, const float *w_transposed, const float *b1,
                                       const float *b2, const float *b3, float *dx_centre, float *dx_neighbour, float *partial,
                                       float *grad_tiles, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0) return fail(SPS_ERR_INVALID, "dense_edge_conv_bwd: bad shape b=%d n=%d", b, n);
    if (d != DB_D || k != DB_K || growth != DB_G)
        return fail(SPS_ERR_INVALID, "dense_edge_conv_bwd: built for d=%d, knn=%d, growth=%d (got %d, %d, %d)", DB_D, DB_K, DB_G, d, k, growth);
    if (!x || !idx || !grad_out || !w_fwd || !w_transposed || !b1 || !b2 || !b3 || !dx_centre || !dx_neighbour || !partial || !grad_tiles)
        return fail(SPS_ERR_INVALID, "dense_edge_conv_bwd: null pointer");
    const int tiles = relative_only ? 9 : 13;
    const int blocks = sps_dense_edge_conv_bwd_blocks();
    hipStream_t st = as_stream(stream);
    if (b == 0 || n == 0) {
        if (hipMemsetAsync(grad_tiles, 0, sizeof(float) * tiles * 256, st) != hipSuccess)
            return fail(SPS_ERR_LAUNCH, "dense_edge_conv_bwd: memset failed");
        return SPS_OK;
    }
    DbArgs a;
    a.n = n; a.units = (long long)b * n; a.x = x; a.idx = idx; a.g = grad_out; a.wf = w_fwd; a.wt = w_transposed;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.dxc = dx_centre; a.dxn = dx_neighbour; a.partial = partial;
    const size_t lds = sizeof(float) * ((size_t)DB_WT * 64 + (size_t)DB_WAVES * DB_LDS_WAVE);
    static LdsLimitOnce raised[2];
    const void *fn = relative_only ? (const void *)dense_edge_conv_bwd_kernel<true> : (const void *)dense_edge_conv_bwd_kernel<false>;
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit(fn, 100 * 1024, raised[relative_only ? 1 : 0], "dense_edge_conv_bwd");
        if (rc != SPS_OK) return rc;
    }
    if (relative_only) hipLaunchKernelGGL(dense_edge_conv_bwd_kernel<true>, dim3(blocks), dim3(64 * DB_WAVES), lds, st, a);
    else hipLaunchKernelGGL(dense_edge_conv_bwd_kernel<false>, dim3(blocks), dim3(64 * DB_WAVES), lds, st, a);
    hipLaunchKernelGGL(dec_bwd_reduce_kernel, dim3(divup(tiles * 256, 4)), dim3(256), 0, st, blocks, tiles * 256, partial, grad_tiles);
    return check_launch("dense_edge_conv_bwd_kernel");
}

extern "C" int sps_linear_rows_bwd_blocks(void) { return 512; }

extern "C" int sps_linear_rows_bwd(long long rows, int cin, int cout, const float *x, const float *y, const float *dy,
                                   const float *w, int relu, float *dx, float *partial, float *grad_w_b, sps_stream_t stream) {
    using namespace sps;
    if (rows < 0 || cin <= 0) return fail(SPS_ERR_INVALID, "linear_rows_bwd: bad shape rows=%lld cin=%d", rows, cin);
    if (cout != LB_COUT || cin > LB_MAX_CIN) return fail(SPS_ERR_INVALID, "linear_rows_bwd: built for cout=%d, cin<=%d (got %d, %d)", LB_COUT, LB_MAX_CIN, cout, cin);
    if (!x || !dy || !w || !dx || !partial || !grad_w_b || (relu && !y)) return fail(SPS_ERR_INVALID, "linear_rows_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int count = LB_COUT * cin + LB_COUT, blocks = sps_linear_rows_bwd_blocks();
    if (rows == 0) {
        if (hipMemsetAsync(grad_w_b, 0, sizeof(float) * count, st) != hipSuccess)
            return fail(SPS_ERR_LAUNCH, "linear_rows_bwd: memset failed");
        return SPS_OK;
    }
    hipLaunchKernelGGL(linear_rows24_bwd_kernel, dim3(blocks), dim3(256), 0, st, rows, cin, x, y, dy, w, relu, dx, partial);
    hipLaunchKernelGGL(dec_bwd_reduce_kernel, dim3(divup(count, 4)), dim3(256), 0, st, blocks, count, partial, grad_w_b);
    return check_launch("linear_rows24_bwd_kernel");
}
