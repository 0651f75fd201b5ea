// dense_edge_conv.hip -- the reference's DenseEdgeConv (surface_feature.py:45-116) as ONE kernel per convolution, plus
// the row-wise Linear ("transform", surface_feature.py:8-27,147-152) that precedes it.
//
// The reference materialises, per convolution, the grouped tensor (B, d, N, K), its (B, N, K, 3d) edge features and
// three concatenated activations (B, N, K, d + i*c) through ~15 cuBLAS/elementwise launches; at B x N = 8 x 16384,
// K = 16, d = 24 that is ~1.3 GB of traffic per convolution.  Here a wave owns one centre point: its K = 16
// neighbours are the 16 columns of v_mfma_f32_16x16x4_f32, the growth-rate (12) output channels are the rows (padded
// to 16), every activation stays in registers (a D fragment of layer i is directly the B operand of layer i+1: lane
// (q, col) holds rows 4q..4q+3 of column col, which is k-slot q of k-step r for r = 0..3), and the only HBM traffic
// is the 96-byte feature rows read (L2-resident) and the 240-byte output row written.
//
// Exact fp32 (MFMA fp32 = an fmaf chain per output), weights live in registers for the whole kernel (42 VGPRs).
// Channel -> k-slot mapping of the d = 24 input channels: lane q of k-step j carries channel 6q + j, so that a lane
// loads 24 contiguous bytes of a feature row.  The host packs the weights to match (fused.py: pack_dense_edge_conv).
//
// Output row (reference order, surface_feature.py:98-116): [max_k y3 (12) | max_k y2 (12) | max_k y1 (12) | x (24)].
#include "sps_common.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int DEC_D = 24, DEC_K = 16, DEC_G = 12, DEC_OUT = DEC_D + 3 * DEC_G;

struct DecArgs {
    int n;
    long long units;  // B * N centres
    const float *x;   // (B, N, 24)
    const int *idx;   // (B, N, 16)
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    float *out;       // (B, N, 60)
};

__device__ __forceinline__ f32x4 dec_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// max over the 16 lanes of a DPP row, valid in every lane (row rotations 8, 4, 2, 1)
__device__ __forceinline__ float row_allmax(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false)));
    return v;
}

template <bool REL>
__global__ __launch_bounds__(256) void dense_edge_conv_kernel(DecArgs a) {
    constexpr int KS1 = REL ? 6 : 18;
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    float w1r[KS1], w2r[10], w3r[14];
#pragma unroll
    for (int k = 0; k < KS1; ++k) w1r[k] = a.w1[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 10; ++k) w2r[k] = a.w2[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 14; ++k) w3r[k] = a.w3[k * 64 + lane];
    const f32x4 b1v = *reinterpret_cast<const f32x4 *>(a.b1 + 4 * q);
    const f32x4 b2v = *reinterpret_cast<const f32x4 *>(a.b2 + 4 * q);
    const f32x4 b3v = *reinterpret_cast<const f32x4 *>(a.b3 + 4 * q);

    const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long p = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); p < a.units; p += nwaves) {
        const long long scene = p / a.n;
        const int nb = a.idx[p * DEC_K + c];
        const f32x2 *xcp = reinterpret_cast<const f32x2 *>(a.x + p * DEC_D + 6 * q);
        const f32x2 *xnp = reinterpret_cast<const f32x2 *>(a.x + (scene * a.n + nb) * DEC_D + 6 * q);
        float xc[6], xn[6];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const f32x2 u = xcp[h], v = xnp[h];
            xc[2 * h] = u[0]; xc[2 * h + 1] = u[1];
            xn[2 * h] = v[0]; xn[2 * h + 1] = v[1];
        }
        // ---- layer_first: [centre | neighbour | neighbour - centre] (or the difference only) -> 12, ReLU ----
        f32x4 y1 = b1v;
        if (!REL) {
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = dec_mfma(w1r[j], xc[j], y1);
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = dec_mfma(w1r[6 + j], xn[j], y1);
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = dec_mfma(w1r[12 + j], xn[j] - xc[j], y1);
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) y1 = dec_mfma(w1r[j], xn[j] - xc[j], y1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) y1[r] = fmaxf(y1[r], 0.f);
        // ---- layers.0: [y1 | centre] -> 12, ReLU ----
        f32x4 y2 = b2v;
#pragma unroll
        for (int r = 0; r < 4; ++r) y2 = dec_mfma(w2r[r], y1[r], y2);
#pragma unroll
        for (int j = 0; j < 6; ++j) y2 = dec_mfma(w2r[4 + j], xc[j], y2);
#pragma unroll
        for (int r = 0; r < 4; ++r) y2[r] = fmaxf(y2[r], 0.f);
        // ---- layer_last: [y2 | y1 | centre] -> 12, no activation ----
        f32x4 y3 = b3v;
#pragma unroll
        for (int r = 0; r < 4; ++r) y3 = dec_mfma(w3r[r], y2[r], y3);
#pragma unroll
        for (int r = 0; r < 4; ++r) y3 = dec_mfma(w3r[4 + r], y1[r], y3);
#pragma unroll
        for (int j = 0; j < 6; ++j) y3 = dec_mfma(w3r[8 + j], xc[j], y3);
        // ---- max over the K columns; lane (q, c) with c < 3 stores segment c of rows 4q..4q+3 ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y1[r] = row_allmax(y1[r]);
            y2[r] = row_allmax(y2[r]);
            y3[r] = row_allmax(y3[r]);
        }
        float *o = a.out + p * DEC_OUT;
        if (q < 3) {
            if (c == 0) *reinterpret_cast<f32x4 *>(o + 4 * q) = y3;
            if (c == 1) *reinterpret_cast<f32x4 *>(o + DEC_G + 4 * q) = y2;
            if (c == 2) *reinterpret_cast<f32x4 *>(o + 2 * DEC_G + 4 * q) = y1;
        }
        if (c >= 3 && c < 6) {
            f32x2 v;
            v[0] = xc[2 * (c - 3)];
            v[1] = xc[2 * (c - 3) + 1];
            *reinterpret_cast<f32x2 *>(o + 3 * DEC_G + 6 * q + 2 * (c - 3)) = v;
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
    const long long want = (a.units + 3) / 4;
    const int grid = (int)(want < 256 * 16 ? want : 256 * 16);
    if (relative_only) hipLaunchKernelGGL(dense_edge_conv_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL(dense_edge_conv_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), a);
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
