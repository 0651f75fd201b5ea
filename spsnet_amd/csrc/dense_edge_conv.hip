// dense_edge_conv.hip -- the reference's DenseEdgeConv (surface_feature.py:45-116) as ONE kernel per convolution, plus
// the row-wise Linear ("transform", surface_feature.py:8-27,147-152) that precedes it.
//
// The reference materialises, per convolution, the grouped tensor (B, d, N, K), its (B, N, K, 3d) edge features and
// three concatenated activations (B, N, K, d + i*c) through ~15 cuBLAS/elementwise launches; at B x N = 8 x 16384,
// K = 16, d = 24 that is ~1.3 GB of traffic per convolution.  Here a wave owns a tile of 16 centre points and every
// activation stays in registers:
//   * the part of each layer that depends on the CENTRE only (W1[:, :d] x_i, W2[:, c:] x_i, W3[:, 2c:] x_i -- the edge
//     feature and both dense concatenations repeat x_i for all K neighbours) is computed once per centre, for the 16
//     centres of the tile at a time: they are the 16 columns of v_mfma_f32_16x16x4_f32, the three 12-row blocks (padded
//     to 16) are three output tiles;
//   * then, centre by centre, its K = 16 neighbours are the 16 columns: the centre terms enter as the accumulator (a row
//     broadcast of the centre's column), the neighbour-dependent parts follow as MFMAs; a D fragment of layer i
//     is directly the B operand of layer i+1 (lane (q, col) holds rows 4q..4q+3 of column col = k-slot q of k-steps
//     r = 0..3), and the max over K is a DPP row reduction.
// 24 MFMAs per centre (18 for the first, difference-only convolution) instead of the 42 (30) of a per-centre GEMM chain;
// the only HBM traffic is the 96-byte feature rows (L2-resident), the neighbour table and the 240-byte output row.
//
// fp32 throughout (MFMA fp32 = an fmaf chain per output); the difference x_j - x_i is rounded before it is multiplied,
// as in the reference.  An optional merged form of the first layer, (W1a - W1c) x_i + (W1b + W1c) x_j with the weight
// sums formed on the host (18 MFMAs, ~1e-6 relative off), exists for static graphs (fused.DEC_MERGED); it is OFF by
// default because in a DYNAMIC graph the features are the next convolution's query positions, and every last-bit
// change moves some point across a ball's boundary.  Channel -> k-slot mapping of the d = 24 input channels: lane q of k-step j carries channel 6q + j, so
// that a lane loads 24 contiguous bytes of a feature row.  The host packs the weights to match (fused.py:
// pack_dense_edge_conv): w1 = [centre | neighbour | difference] k-steps (the difference block alone when relative_only),
// w2 = [y1 (4) | centre (6)], w3 = [y2 (4) | y1 (4) | centre (6)].
//
// Output row (reference order, surface_feature.py:98-116): [max_k y3 (12) | max_k y2 (12) | max_k y1 (12) | x (24)].
#include "sps_common.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int DEC_D = 24, DEC_K = 16, DEC_G = 12, DEC_OUT = DEC_D + 3 * DEC_G, DEC_TILE = 16;

struct DecArgs {
    int n;
    long long units;  // B * N centres
    const float *x;   // (B, N, 24)
    const int *idx;   // (B, N, 16)
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;       // (B, N, 60)
};

__device__ __forceinline__ f32x4 dec_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// y[r] = max over the 16 lanes of its DPP row, in every lane, for the three activations at once: one v_max_f32_dpp per
// value and rotation (8, 4, 2, 1).  hipcc splits update_dpp + fmaxf into v_mov_dpp + v_max (and a v_mov for `old`), i.e.
// three VALU slots per step; written out, the twelve independent chains also cover each other's DPP read-after-write
// wait states (the leading s_nop covers the first).
#define DEC_ROR_STEP(ROT)                                                                                            \
    asm volatile("s_nop 1\n"                                                                                         \
                 "v_max_f32_dpp %0, %0, %0 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %1, %1, %1 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %2, %2, %2 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %3, %3, %3 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %4, %4, %4 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %5, %5, %5 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %6, %6, %6 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %7, %7, %7 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %8, %8, %8 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %9, %9, %9 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                             \
                 "v_max_f32_dpp %10, %10, %10 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                          \
                 "v_max_f32_dpp %11, %11, %11 row_ror:" #ROT " row_mask:0xf bank_mask:0xf\n"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c0), "+v"(c1),    \
                   "+v"(c2), "+v"(c3))
__device__ __forceinline__ void row_allmax3(f32x4 &y1, f32x4 &y2, f32x4 &y3) {
    float a0 = y1[0], a1 = y1[1], a2 = y1[2], a3 = y1[3], b0 = y2[0], b1 = y2[1], b2 = y2[2], b3 = y2[3], c0 = y3[0],
          c1 = y3[1], c2 = y3[2], c3 = y3[3];
    DEC_ROR_STEP(8);
    DEC_ROR_STEP(4);
    DEC_ROR_STEP(2);
    DEC_ROR_STEP(1);
    y1 = (f32x4){a0, a1, a2, a3};
    y2 = (f32x4){b0, b1, b2, b3};
    y3 = (f32x4){c0, c1, c2, c3};
}
#undef DEC_ROR_STEP
__device__ __forceinline__ void dec_load6(const float *row, float (&v)[6]) {
    const f32x2 *p = reinterpret_cast<const f32x2 *>(row);
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const f32x2 u = p[h];
        v[2 * h] = u[0];
        v[2 * h + 1] = u[1];
    }
}
// lane `src` (0..15, wave-uniform) of every 16-lane row, broadcast to the row
__device__ __forceinline__ float row_pick(float v, int addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }

// FORM 0: layer_first on [x_i | x_j | x_j - x_i] as the reference forms it (w1 = 18 k-steps); 1: on x_j - x_i only (6);
// 2: the algebraically merged (W1a - W1c) x_i + (W1b + W1c) x_j (12 k-steps, a third fewer MFMAs, ~1e-6 relative off)
template <int FORM>
__global__ __launch_bounds__(256, 4) void dense_edge_conv_kernel(DecArgs a) {
    constexpr bool REL = FORM == 1;
    constexpr int CEN = REL ? 0 : 6;      // k-steps of w1 that multiply the centre
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    // weights of the per-centre steps live in registers; those of the per-tile centre terms (and the biases) are
    // re-read from LDS once per tile, which keeps the kernel at four waves per SIMD
    __shared__ float cw[18 + 12][64];
    if (threadIdx.x < 64) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            cw[j][lane] = REL ? 0.f : a.w1[j * 64 + lane];
            cw[6 + j][lane] = a.w2[(4 + j) * 64 + lane];
            cw[12 + j][lane] = a.w3[(8 + j) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cw[18 + r][lane] = a.b1[4 * q + r];
            cw[22 + r][lane] = a.b2[4 * q + r];
            cw[26 + r][lane] = a.b3[4 * q + r];
        }
    }
    float w1x[6], w1d[FORM == 0 ? 6 : 1], w2y[4], w3y[8];  // w1x multiplies x_j (or x_j - x_i when REL), w1d the difference
#pragma unroll
    for (int j = 0; j < 6; ++j) w1x[j] = a.w1[(CEN + j) * 64 + lane];
    if (FORM == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) w1d[j] = a.w1[(12 + j) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) w2y[r] = a.w2[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < 8; ++r) w3y[r] = a.w3[r * 64 + lane];
    __syncthreads();

    const long long tiles = (a.units + DEC_TILE - 1) / DEC_TILE;
    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    const long long last = a.units - 1;
    for (long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < tiles; tile += nwaves) {
        const long long p0 = tile * DEC_TILE;
        const int count = (a.units - p0 < DEC_TILE) ? (int)(a.units - p0) : DEC_TILE;
        const long long pc = (p0 + c < last) ? p0 + c : last;  // column c's centre (clamped in a short tile)
        float xc[6];
        dec_load6(a.x + pc * DEC_D + 6 * q, xc);
        const int srow = (int)((pc / a.n) * a.n);  // first row of that centre's scene
        // ---- centre terms of the three layers for the 16 centres of the tile: column c = centre c ----
        f32x4 t0, t1, t2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t0[r] = cw[18 + r][lane];
            t1[r] = cw[22 + r][lane];
            t2[r] = cw[26 + r][lane];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (!REL) t0 = dec_mfma(cw[j][lane], xc[j], t0);
            t1 = dec_mfma(cw[6 + j][lane], xc[j], t1);
            t2 = dec_mfma(cw[12 + j][lane], xc[j], t2);
        }
        // the centre's own channels are the tail of its output row
        if (p0 + c < a.units) {
            f32x2 *o = reinterpret_cast<f32x2 *>(a.out + (p0 + c) * DEC_OUT + 3 * DEC_G + 6 * q);
#pragma unroll
            for (int h = 0; h < 3; ++h) o[h] = (f32x2){xc[2 * h], xc[2 * h + 1]};
        }
        // ---- centre by centre; the neighbour table runs two centres ahead, the neighbour rows one ----
        const int *ip = a.idx + p0 * DEC_K + c;
        int nb1 = ip[(count > 1 ? 1 : 0) * DEC_K];
        float xn[6];
        dec_load6(a.x + ((long long)__builtin_amdgcn_readlane(srow, 0) + ip[0]) * DEC_D + 6 * q, xn);
        for (int cc = 0; cc < count; ++cc) {
            const int c1 = (cc + 1 < count) ? cc + 1 : cc, c2 = (cc + 2 < count) ? cc + 2 : c1;
            const int nb2 = ip[c2 * DEC_K];
            float xnext[6];
            dec_load6(a.x + ((long long)__builtin_amdgcn_readlane(srow, c1) + nb1) * DEC_D + 6 * q, xnext);
            const int pick = ((lane & 48) | cc) << 2;
            float xd[6];
            if (FORM != 2) {  // the reference's rounded difference x_j - x_i
#pragma unroll
                for (int j = 0; j < 6; ++j) xd[j] = xn[j] - row_pick(xc[j], pick);
            }
            // ---- layer_first ----
            f32x4 y1;
#pragma unroll
            for (int r = 0; r < 4; ++r) y1[r] = row_pick(t0[r], pick);
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = dec_mfma(w1x[j], REL ? xd[j] : xn[j], y1);
            if (FORM == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) y1 = dec_mfma(w1d[j], xd[j], y1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) y1[r] = fmaxf(y1[r], 0.f);
            // ---- layers.0: [y1 | centre] ----
            f32x4 y2;
#pragma unroll
            for (int r = 0; r < 4; ++r) y2[r] = row_pick(t1[r], pick);
#pragma unroll
            for (int r = 0; r < 4; ++r) y2 = dec_mfma(w2y[r], y1[r], y2);
#pragma unroll
            for (int r = 0; r < 4; ++r) y2[r] = fmaxf(y2[r], 0.f);
            // ---- layer_last: [y2 | y1 | centre], no activation ----
            f32x4 y3;
#pragma unroll
            for (int r = 0; r < 4; ++r) y3[r] = row_pick(t2[r], pick);
#pragma unroll
            for (int r = 0; r < 4; ++r) y3 = dec_mfma(w3y[r], y2[r], y3);
#pragma unroll
            for (int r = 0; r < 4; ++r) y3 = dec_mfma(w3y[4 + r], y1[r], y3);
            // ---- max over the K columns; three lanes of each row q < 3 store rows 4q..4q+3 of one segment each ----
            row_allmax3(y1, y2, y3);
            float *o = a.out + (p0 + cc) * DEC_OUT + 4 * q;
            if (q < 3) {
                if (c == 0) *reinterpret_cast<f32x4 *>(o) = y3;
                if (c == 1) *reinterpret_cast<f32x4 *>(o + DEC_G) = y2;
                if (c == 2) *reinterpret_cast<f32x4 *>(o + 2 * DEC_G) = y1;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) xn[j] = xnext[j];
            nb1 = nb2;
        }
    }
}

// out[r][o] = act(b[o] + sum_i x[r][i] * w[o][i]) for COUT = 24 outputs; one thread per row, weights broadcast from LDS
constexpr int LIN_COUT = 24, LIN_MAX_CIN = 64;
__global__ __launch_bounds__(256) void linear_rows24_kernel(long long rows, int cin, const float *__restrict__ x,
                                                            const float *__restrict__ w, const float *__restrict__ bias,
                                                            int relu, float *__restrict__ out) {
    __shared__ float ws[LIN_MAX_CIN * LIN_COUT];  // [i][o]
    __shared__ float bs[LIN_COUT];
    for (int e = threadIdx.x; e < cin * LIN_COUT; e += blockDim.x) {
        const int i = e / LIN_COUT, o = e % LIN_COUT;
        ws[e] = w[o * cin + i];
    }
    if (threadIdx.x < LIN_COUT) bs[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float acc[LIN_COUT];
#pragma unroll
    for (int o = 0; o < LIN_COUT; ++o) acc[o] = 0.f;
    const float *xr = x + r * cin;
    for (int i = 0; i < cin; ++i) {
        const float v = xr[i];
#pragma unroll
        for (int o = 0; o < LIN_COUT; ++o) acc[o] = __builtin_fmaf(v, ws[i * LIN_COUT + o], acc[o]);
    }
    float *orow = out + r * LIN_COUT;
#pragma unroll
    for (int o = 0; o < LIN_COUT; o += 4) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = acc[o + u] + bs[o + u];
            v[u] = relu ? fmaxf(t, 0.f) : t;
        }
        *reinterpret_cast<f32x4 *>(orow + o) = v;
    }
}

}  // namespace sps

extern "C" int sps_dense_edge_conv(int b, int n, int d, int k, int growth, int relative_only, const float *x, const int *idx,
                                   const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                                   const float *b3, float *out, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0) return fail(SPS_ERR_INVALID, "dense_edge_conv: bad shape b=%d n=%d", b, n);
    if (d != DEC_D || k != DEC_K || growth != DEC_G)
        return fail(SPS_ERR_INVALID, "dense_edge_conv: built for d=%d, knn=%d, growth=%d (got %d, %d, %d)", DEC_D, DEC_K, DEC_G, d, k, growth);
    if (b == 0 || n == 0) return SPS_OK;
    if (!x || !idx || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out) return fail(SPS_ERR_INVALID, "dense_edge_conv: null pointer");
    DecArgs a;
    a.n = n; a.units = (long long)b * n; a.x = x; a.idx = idx;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.out = out;
    const long long want = ((a.units + DEC_TILE - 1) / DEC_TILE + 3) / 4;  // 4 waves per workgroup, one tile per wave and trip
    const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
    if (relative_only == 1) hipLaunchKernelGGL(dense_edge_conv_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream), a);
    else if (relative_only == 2) hipLaunchKernelGGL(dense_edge_conv_kernel<2>, dim3(grid), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(dense_edge_conv_kernel<0>, dim3(grid), dim3(256), 0, as_stream(stream), a);
    return check_launch("dense_edge_conv_kernel");
}

extern "C" int sps_linear_rows(long long rows, int cin, int cout, const float *x, const float *w, const float *bias, int relu,
                               float *out, sps_stream_t stream) {
    using namespace sps;
    if (rows < 0 || cin <= 0) return fail(SPS_ERR_INVALID, "linear_rows: bad shape rows=%lld cin=%d", rows, cin);
    if (cout != LIN_COUT || cin > LIN_MAX_CIN) return fail(SPS_ERR_INVALID, "linear_rows: built for cout=%d, cin<=%d (got %d, %d)", LIN_COUT, LIN_MAX_CIN, cout, cin);
    if (rows == 0) return SPS_OK;
    if (!x || !w || !out) return fail(SPS_ERR_INVALID, "linear_rows: null pointer");
    const long long grid = (rows + 255) / 256;
    if (grid > 0x7fffffffLL) return fail(SPS_ERR_INVALID, "linear_rows: too many rows");
    hipLaunchKernelGGL(linear_rows24_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), rows, cin, x, w, bias, relu, out);
    return check_launch("linear_rows24_kernel");
}
