// pw_mlp.hip -- the per-point ("1x1") layer stacks that follow the grouped MLPs of an SA layer, as ONE kernel:
//   aggregation  Conv1d(Ccat -> Cagg, no bias) + BatchNorm1d + ReLU          (pointnet2_modules.py:213-228, 449-450)
//   confidence   Conv1d(Cagg -> Cagg) + BatchNorm1d + ReLU, Conv1d(Cagg -> num_class, bias)   (:230-245, 454-455)
// The reference runs them as 3 + 4 separate cuDNN/elementwise launches per layer (here: hipBLASLt GEMM, MIOpen BN,
// clamp, ...: ~12 launches, 60-90 us per SA layer); BatchNorm is folded on the host (eval mode).
//
// Exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32 = an fmaf chain per output).  A workgroup owns 16 consecutive
// points (the 16 columns of an MFMA) of one scene; its 4..16 waves split the 16-row output tiles of each layer; the
// input columns and the activations of each layer sit in LDS ([channel][16 + 1 pad]).  Layer 1 writes the aggregated
// features (B, C1, M) and keeps them for layer 2;
// layer 3 (num_class <= 16 rows) writes the class scores point-major (B, M, num_class), which is the layout the
// samplers of the next SA layer read (the reference returns a transposed view and copies later).
// Weight fragments are packed by the host as [tile][k16][lane = 16 q + i][r]: W[16 tile + i][16 k16 + 4 r + q], i.e. one
// dwordx4 per lane feeds the four MFMAs of a 16-channel step.
#include "sps_common.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int PW_MAX_WAVES = 16;
constexpr int PW_PAD = 17;  // LDS row pitch of the activation images (floats)

struct PwArgs {
    int m, j0, cin, c1, c2, c3, c3_real;  // j0: first point of the launch (a multiple of 16)
    const float *x, *w1, *b1, *w2, *b2, *w3, *b3;
    float *y1, *y1t, *y3;
    int out_h16;   // y1 / y1t are written as halves (fp16 features in HBM, BASELINE configs[4]); arithmetic stays fp32
    int x_pm;      // x is point-major (B, M, cin) instead of (B, cin, M)
    const int *run_if;   // predicated launch: nothing happens when *run_if == 0
    const int *alt;      // *alt != 0: all m points of every scene instead of [j0, j0 + 16 gridDim.x) (self-repairing range launch)
};

// ReLU as an integer max: keeps +Inf / +NaN where v_max_f32 would drop a NaN (torch's ReLU propagates it)
__device__ __forceinline__ float pw_relu(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

__device__ __forceinline__ f32x4 pw_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// one 16-row output tile: acc += W[tile] * act (act = LDS image [k][17]).  A tile is a serial chain of MFMAs fed by weight
// fragments streamed from L2 (~1-2 us per round trip under load): the fragments of the NEXT eight 16-channel steps are
// requested before the current eight are multiplied, so that only the first round trip of a tile is exposed (four steps
// per trip and no overlap cost 8 + 4 + 4 exposed round trips per workgroup at IA-SSD layer 2: 26 us -> see DESIGN.md 4.4).
__device__ __forceinline__ f32x4 pw_tile(const f32x4 *__restrict__ wp, const float *__restrict__ act, int k16n, int q, int c,
                                         f32x4 acc) {
    constexpr int G = 8;
    const float *ap = act + q * PW_PAD + c;
    f32x4 cur[G], nxt[G];
    int k = 0;
    if (k16n >= G) {
#pragma unroll
        for (int u = 0; u < G; ++u) cur[u] = wp[(size_t)u * 64];
        for (; k + G <= k16n; k += G) {
            const bool more = k + 2 * G <= k16n;
            if (more) {
#pragma unroll
                for (int u = 0; u < G; ++u) nxt[u] = wp[(size_t)(k + G + u) * 64];
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = pw_mfma(cur[u][r], ap[(16 * (k + u) + 4 * r) * PW_PAD], acc);
            if (more) {
#pragma unroll
                for (int u = 0; u < G; ++u) cur[u] = nxt[u];
            }
        }
    }
    for (; k < k16n; ++k) {
        const f32x4 w = wp[(size_t)k * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = pw_mfma(w[r], ap[(16 * k + 4 * r) * PW_PAD], acc);
    }
    return acc;
}

__global__ __launch_bounds__(64 * PW_MAX_WAVES) void pw_mlp_kernel(PwArgs a) {
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ float pw_lds[];
    // [x tile | layer-2 output] share a region (the input tile is dead once layer 1 is done), then layer-1 output
    const int r0 = (a.cin > a.c2 ? a.cin : a.c2) * PW_PAD;
    float *xt = pw_lds, *act2 = pw_lds, *act1 = pw_lds + r0;
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int scene = blockIdx.y;
    const bool whole = a.alt && *a.alt != 0;
    const int first = whole ? 0 : a.j0, tiles = whole ? a.m / 16 : (int)gridDim.x;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int m0 = first + tile * 16;
    __syncthreads();   // the previous tile's LDS images are dead

    // ---- stage the 16 input columns: cin rows of 64 bytes (or, point-major, 16 rows of cin floats), coalesced ----
    if (a.x_pm) {
        const float *xs = a.x + ((size_t)scene * a.m + m0) * a.cin;
        for (int e = threadIdx.x; e < a.cin * 16; e += blockDim.x) {
            const int col = e / a.cin, ch = e - col * a.cin;
            xt[ch * PW_PAD + col] = xs[e];
        }
    } else {
        const float *xs = a.x + (size_t)scene * a.cin * a.m + m0;
        for (int e = threadIdx.x; e < a.cin * 16; e += blockDim.x) {
            const int ch = e >> 4, col = e & 15;
            xt[ch * PW_PAD + col] = xs[(size_t)ch * a.m + col];
        }
    }
    __syncthreads();
    // ---- layer 1 -> y1 (global) and act1 (LDS) ----
    {
        const int k16n = a.cin / 16;
        for (int t = wv; t < a.c1 / 16; t += nw) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(a.b1 + 16 * t + 4 * q);
            acc = pw_tile(reinterpret_cast<const f32x4 *>(a.w1) + (size_t)t * k16n * 64 + lane, xt, k16n, q, c, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = pw_relu(acc[r]);
                const int row = 16 * t + 4 * q + r;
                const size_t at = ((size_t)scene * a.c1 + row) * a.m + m0 + c;
                if (a.out_h16) {
                    // features live in HBM as halves; the head below keeps reading what was STORED (the rounded value), so
                    // that the class scores are a function of the returned features
                    const _Float16 hv = (_Float16)acc[r];
                    reinterpret_cast<_Float16 *>(a.y1)[at] = hv;
                    acc[r] = (float)hv;
                } else {
                    a.y1[at] = acc[r];
                }
                act1[row * PW_PAD + c] = acc[r];
            }
            // point-major twin (B, M, C1) for the next layer's grouped-MLP gathers: 16 (8) bytes per lane, 64 (32) per column
            if (a.y1t) {
                const size_t at = ((size_t)scene * a.m + m0 + c) * a.c1 + 16 * t + 4 * q;
                if (a.out_h16) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(a.y1t) + at) =
                        (h4){(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
                } else {
                    *reinterpret_cast<f32x4 *>(a.y1t + at) = acc;
                }
            }
        }
    }
    if (!a.w2) continue;
    __syncthreads();
    // ---- layer 2: act1 -> act2 ----
    {
        const int k16n = a.c1 / 16;
        for (int t = wv; t < a.c2 / 16; t += nw) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(a.b2 + 16 * t + 4 * q);
            acc = pw_tile(reinterpret_cast<const f32x4 *>(a.w2) + (size_t)t * k16n * 64 + lane, act1, k16n, q, c, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) act2[(16 * t + 4 * q + r) * PW_PAD + c] = pw_relu(acc[r]);
        }
    }
    __syncthreads();
    // ---- layer 3: one 16-row tile (num_class <= 16), no ReLU, point-major output ----
    if (wv == 0) {
        f32x4 acc = *reinterpret_cast<const f32x4 *>(a.b3 + 4 * q);
        acc = pw_tile(reinterpret_cast<const f32x4 *>(a.w3) + lane, act2, a.c2 / 16, q, c, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * q + r;
            if (row < a.c3_real) a.y3[((size_t)scene * a.m + m0 + c) * a.c3_real + row] = acc[r];
        }
    }
    }   // tile loop
}

// ---- PointnetFPModule (pointnet2_modules.py:539-587) in inference, as ONE kernel -------------------------------------------
//   three_interpolate(known_feats, idx, weight)  ->  cat([interpolated, unknow_feats], dim=1)  ->  [Conv2d 1x1 + BatchNorm2d +
//   ReLU] x (1 | 2)
// The reference runs an interpolation kernel, a concatenation copy and 3 launches per MLP layer over (B, C, n) tensors at the
// resolution of the FINER level (PointRCNN's last feature-propagation layer: 8 x 16 384 points x 257 -> 128 -> 128).  Here a
// workgroup owns 16 points: the [interpolated | skip] input columns are built straight in LDS (same contraction as
// three_interpolate_kernel: w0 f0, then fma, fma -- the interpolated values are bit-identical to the unfused op), the layers
// run as in pw_mlp_kernel (exact fp32 MFMA, BatchNorm folded), and only the last layer's output is written.
struct FpArgs {
    int n, m, c_known, c_skip, cin, c1, c2;      // cin = c_known + c_skip rounded up to 16; c2 = 0: a one-layer stack
    const float *known_feats, *skip;             // (b, c_known, m), (b, c_skip, n) | NULL
    const int *idx;                              // (b, n, 3)
    const float *weight;                         // (b, n, 3)
    const float *w1, *b1, *w2, *b2;
    float *y;                                    // (b, c2 ? c2 : c1, n), or point-major (b, n, c2 ? c2 : c1) with y_pm
    int from_dist;                               // `weight` holds three_nn's distances: the weights are formed here
    int y_pm;
};

// one 16-row output tile against NT column tiles of an LDS image [k][16 NT + 1]: the weight fragments of a 16-channel step
// feed NT times four MFMAs (at the finest level a layer has 131 072 points: per-16-point weight streams would be 1.6 GB)
template <int NT>
__device__ __forceinline__ void fp_tile(const f32x4 *__restrict__ wp, const float *__restrict__ act, int k16n, int q, int c,
                                        f32x4 (&acc)[NT]) {
    constexpr int PITCH = 16 * NT + 1;
    constexpr int G = 4;
    const float *ap = act + q * PITCH + c;
    f32x4 cur[G], nxt[G];
    int k = 0;
    if (k16n >= G) {
#pragma unroll
        for (int u = 0; u < G; ++u) cur[u] = wp[(size_t)u * 64];
        for (; k + G <= k16n; k += G) {
            const bool more = k + 2 * G <= k16n;
            if (more) {
#pragma unroll
                for (int u = 0; u < G; ++u) nxt[u] = wp[(size_t)(k + G + u) * 64];
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = pw_mfma(cur[u][r], ap[(16 * (k + u) + 4 * r) * PITCH + 16 * nt], acc[nt]);
            if (more) {
#pragma unroll
                for (int u = 0; u < G; ++u) cur[u] = nxt[u];
            }
        }
    }
    for (; k < k16n; ++k) {
        const f32x4 w = wp[(size_t)k * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = pw_mfma(w[r], ap[(16 * k + 4 * r) * PITCH + 16 * nt], acc[nt]);
    }
}

template <int NT>
__global__ __launch_bounds__(64 * PW_MAX_WAVES) void fp_mlp_kernel(FpArgs a) {
    extern __shared__ float pw_lds[];
    constexpr int PITCH = 16 * NT + 1, COLS = 16 * NT;
    const int r0 = (a.cin > a.c2 ? a.cin : a.c2) * PITCH;
    float *xt = pw_lds, *act1 = pw_lds + r0;
    __shared__ int s_idx[3 * COLS];
    __shared__ float s_w[3 * COLS];
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int scene = blockIdx.y;
    const int m0 = blockIdx.x * COLS;
    for (int e = threadIdx.x; e < 3 * COLS; e += blockDim.x) {
        const bool in = m0 + e / 3 < a.n;
        s_idx[e] = in ? a.idx[((size_t)scene * a.n + m0) * 3 + e] : 0;
        s_w[e] = in ? a.weight[((size_t)scene * a.n + m0) * 3 + e] : 0.f;
    }
    __syncthreads();
    if (a.from_dist) {
        // the module's own arithmetic (pointnet2_modules.py:572-574): 1 / (dist + 1e-8), normalised by the sum of the three
        for (int p = threadIdx.x; p < COLS; p += blockDim.x) {
            const float i0 = 1.0f / (s_w[3 * p] + 1e-8f), i1 = 1.0f / (s_w[3 * p + 1] + 1e-8f), i2 = 1.0f / (s_w[3 * p + 2] + 1e-8f);
            const float sum = (i0 + i1) + i2;
            s_w[3 * p] = i0 / sum; s_w[3 * p + 1] = i1 / sum; s_w[3 * p + 2] = i2 / sum;
        }
        __syncthreads();
    }
    // ---- the input columns: interpolated channels, then the skip channels, then zero padding ----
    const float *kf = a.known_feats + (size_t)scene * a.c_known * a.m;
    for (int e = threadIdx.x; e < a.cin * COLS; e += blockDim.x) {
        const int ch = e / COLS, col = e - ch * COLS;
        float v = 0.f;
        if (m0 + col < a.n) {
            if (ch < a.c_known) {
                const float *f = kf + (size_t)ch * a.m;
                // same three-term order as three_interpolate_kernel (interpolate.hip): w1*f1 first
                float t = s_w[3 * col + 1] * f[s_idx[3 * col + 1]];
                t = __builtin_fmaf(s_w[3 * col], f[s_idx[3 * col]], t);
                v = __builtin_fmaf(s_w[3 * col + 2], f[s_idx[3 * col + 2]], t);
            } else if (ch < a.c_known + a.c_skip) {
                v = a.skip[((size_t)scene * a.c_skip + (ch - a.c_known)) * a.n + m0 + col];
            }
        }
        xt[ch * PITCH + col] = v;
    }
    __syncthreads();
    {
        const int k16n = a.cin / 16;
        for (int t = wv; t < a.c1 / 16; t += nw) {
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b1 + 16 * t + 4 * q);
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
            fp_tile<NT>(reinterpret_cast<const f32x4 *>(a.w1) + (size_t)t * k16n * 64 + lane, xt, k16n, q, c, acc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * q + r, col = 16 * nt + c;
                    const float v = pw_relu(acc[nt][r]);
                    if (a.c2 != 0) act1[row * PITCH + col] = v;
                    else if (m0 + col < a.n && !a.y_pm) a.y[((size_t)scene * a.c1 + row) * a.n + m0 + col] = v;
                }
            if (a.c2 == 0 && a.y_pm) {   // point-major: a lane's four rows are one 16-byte store, a point's 16 channels 64 bytes
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = 16 * nt + c;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = pw_relu(acc[nt][r]);
                    if (m0 + col < a.n) *reinterpret_cast<f32x4 *>(a.y + ((size_t)scene * a.n + m0 + col) * a.c1 + 16 * t + 4 * q) = v;
                }
            }
        }
    }
    if (a.c2 == 0) return;
    __syncthreads();
    {
        const int k16n = a.c1 / 16;
        for (int t = wv; t < a.c2 / 16; t += nw) {
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.b2 + 16 * t + 4 * q);
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = bias;
            fp_tile<NT>(reinterpret_cast<const f32x4 *>(a.w2) + (size_t)t * k16n * 64 + lane, act1, k16n, q, c, acc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * nt + c;
                    if (m0 + col < a.n && !a.y_pm) a.y[((size_t)scene * a.c2 + 16 * t + 4 * q + r) * a.n + m0 + col] = pw_relu(acc[nt][r]);
                }
            if (a.y_pm) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = 16 * nt + c;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = pw_relu(acc[nt][r]);
                    if (m0 + col < a.n) *reinterpret_cast<f32x4 *>(a.y + ((size_t)scene * a.n + m0 + col) * a.c2 + 16 * t + 4 * q) = v;
                }
            }
        }
    }
}

// LDS a workgroup may take for its images, KiB: half a CU's, so that two workgroups share one and the input stage of one
// (scattered reads of the coarse level) overlaps the matrix stage of the other -- PointNet2MSG forward 3.96 ms at 140, 3.90 at
// 74, 3.99 at 48 (fewer points per weight stream).  SPS_FP_LDS_KB overrides (read per launch: experiments).
static int fp_lds_budget() {
    const char *e = getenv("SPS_FP_LDS_KB");
    const int v = e ? atoi(e) : 0;
    return v >= 16 && v <= 140 ? v : 74;
}

template <int NT>
static int fp_launch(const FpArgs &a, int b, size_t lds, int waves, hipStream_t st) {
    static LdsLimitOnce raised;
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit((const void *)fp_mlp_kernel<NT>, 148 * 1024, raised, "fp_module_mlp");
        if (rc != SPS_OK) return rc;
    }
    hipLaunchKernelGGL(fp_mlp_kernel<NT>, dim3(divup(a.n, 16 * NT), b), dim3(64 * waves), lds, st, a);
    return check_launch("fp_mlp_kernel");
}

}  // namespace sps

// cin = c_known + c_skip rounded up to 16 (the packed w1 has cin columns, zero beyond the real ones); c1, c2 multiples of 16,
// c2 = 0 (w2, b2 NULL) for a one-layer stack; y (b, c2 ? c2 : c1, n).  Weights packed as for
// sps_pointwise_mlp (BatchNorm folded, [tile][k16][lane][r]).
extern "C" int sps_fp_module_mlp(int b, int n, int m, int c_known, int c_skip, int c1, int c2, const float *known_feats,
                                 const float *skip, const int *idx, const float *weight, const float *w1, const float *b1,
                                 const float *w2, const float *b2, float *y, sps_stream_t stream) {
    return sps_fp_module_mlp_ex(b, n, m, c_known, c_skip, c1, c2, known_feats, skip, idx, weight, 0, w1, b1, w2, b2, y, 0, stream);
}

// weights_from_dist: `weight` holds three_nn's distances (b, n, 3) and the interpolation weights are formed in the kernel
// (pointnet2_modules.py:572-574); y_point_major: y is (b, n, c2 ? c2 : c1) -- what a backbone that hands out per-point rows
// (pcdet/models/backbones_3d/pointnet2_backbone.py:91) wants from its last module.
extern "C" int sps_fp_module_mlp_ex(int b, int n, int m, int c_known, int c_skip, int c1, int c2, const float *known_feats,
                                    const float *skip, const int *idx, const float *weight, int weights_from_dist,
                                    const float *w1, const float *b1, const float *w2, const float *b2, float *y,
                                    int y_point_major, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0 || m <= 0 || c_known <= 0 || c_skip < 0 || c1 <= 0 || c2 < 0)
        return fail(SPS_ERR_INVALID, "fp_module_mlp: bad shape b=%d n=%d m=%d c_known=%d c_skip=%d c1=%d c2=%d", b, n, m, c_known, c_skip, c1, c2);
    if (b == 0 || n == 0) return SPS_OK;
    if (c1 % 16 || c2 % 16) return fail(SPS_ERR_INVALID, "fp_module_mlp: c1, c2 (%d, %d) must be multiples of 16", c1, c2);
    if (!known_feats || !idx || !weight || !w1 || !b1 || !y || (c_skip > 0 && !skip) || (c2 > 0 && (!w2 || !b2)))
        return fail(SPS_ERR_INVALID, "fp_module_mlp: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "fp_module_mlp: batch %d exceeds the grid limit", b);
    FpArgs a;
    a.n = n; a.m = m; a.c_known = c_known; a.c_skip = c_skip; a.cin = 16 * divup(c_known + c_skip, 16); a.c1 = c1; a.c2 = c2;
    a.known_feats = known_feats; a.skip = skip; a.idx = idx; a.weight = weight; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = y;
    a.from_dist = weights_from_dist != 0; a.y_pm = y_point_major != 0;
    const int wide = c2 > a.cin ? c2 : a.cin;
    const size_t rows = (size_t)wide + (c2 ? c1 : 0);
    // as many 16-point column tiles per workgroup as the LDS images allow (the weights stream once per workgroup)
    int nt = 4;
    while (nt > 1 && (sizeof(float) * (16 * nt + 1) * rows > (size_t)fp_lds_budget() * 1024 || 16 * (nt / 2) >= n)) nt >>= 1;
    const size_t lds = sizeof(float) * (size_t)(16 * nt + 1) * rows;
    if (lds > 148 * 1024) return fail(SPS_ERR_INVALID, "fp_module_mlp: widths (%d, %d, %d) need more LDS than a workgroup has", a.cin, c1, c2);
    int tiles = c1 / 16;
    if (c2 / 16 > tiles) tiles = c2 / 16;
    const int waves = tiles < 4 ? 4 : (tiles > PW_MAX_WAVES ? PW_MAX_WAVES : tiles);
    hipStream_t st = as_stream(stream);
    if (nt == 4) return fp_launch<4>(a, b, lds, waves, st);
    if (nt == 2) return fp_launch<2>(a, b, lds, waves, st);
    return fp_launch<1>(a, b, lds, waves, st);
}

extern "C" int sps_pointwise_mlp_range(int b, int m, int j0, int jcount, int cin, int c1, int c2, int c3_real, const float *x,
                                       const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                                       const float *b3, float *y1, float *y1_point_major, float *y3, sps_stream_t stream) {
    return sps_pointwise_mlp_ex(b, m, j0, jcount, cin, c1, c2, c3_real, x, w1, b1, w2, b2, w3, b3, y1, y1_point_major, y3, 0,
                                nullptr, nullptr, stream);
}

// flags: 1 = y1 / y1_point_major are fp16 buffers (halves), 2 = x is point-major (B, M, cin)
extern "C" int sps_pointwise_mlp_ex(int b, int m, int j0, int jcount, int cin, int c1, int c2, int c3_real, const float *x,
                                    const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                                    const float *b3, void *y1, void *y1_point_major, float *y3, int flags,
                                    const int *run_if, const int *full_range_if, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || m < 0 || cin <= 0 || c1 <= 0) return fail(SPS_ERR_INVALID, "pointwise_mlp: bad shape b=%d m=%d cin=%d c1=%d", b, m, cin, c1);
    if (j0 < 0 || jcount < 0 || j0 + jcount > m || j0 % 16 || jcount % 16)
        return fail(SPS_ERR_INVALID, "pointwise_mlp: range [%d, +%d) must lie in [0, %d) on multiples of 16", j0, jcount, m);
    if (b == 0 || jcount == 0) return SPS_OK;
    if (m % 16 || cin % 16 || c1 % 16) return fail(SPS_ERR_INVALID, "pointwise_mlp: m, cin, c1 (%d, %d, %d) must be multiples of 16", m, cin, c1);
    if (!x || !w1 || !b1 || !y1) return fail(SPS_ERR_INVALID, "pointwise_mlp: null pointer");
    const bool deep = w2 != nullptr;
    if (deep) {
        if (c2 <= 0 || c2 % 16 || c3_real <= 0 || c3_real > 16 || !b2 || !w3 || !b3 || !y3)
            return fail(SPS_ERR_INVALID, "pointwise_mlp: head needs c2 %% 16 == 0 (got %d), 1 <= classes <= 16 (got %d), all pointers", c2, c3_real);
    }
    if (b > 65535) return fail(SPS_ERR_INVALID, "pointwise_mlp: batch %d exceeds the grid limit", b);
    PwArgs a;
    a.m = m; a.j0 = j0; a.cin = cin; a.c1 = c1; a.c2 = deep ? c2 : 0; a.c3 = 16; a.c3_real = deep ? c3_real : 0;
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = deep ? w2 : nullptr; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.y1 = (float *)y1; a.y1t = (float *)y1_point_major; a.y3 = y3;
    a.out_h16 = flags & 1; a.x_pm = (flags >> 1) & 1; a.run_if = run_if; a.alt = full_range_if;
    const int wide = (deep && c2 > cin) ? c2 : cin;
    const size_t lds = sizeof(float) * (size_t)PW_PAD * ((size_t)wide + c1);
    if (lds > 150 * 1024) return fail(SPS_ERR_INVALID, "pointwise_mlp: widths (%d, %d, %d) need more LDS than a workgroup has", cin, c1, c2);
    static LdsLimitOnce raised;
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit((const void *)pw_mlp_kernel, 150 * 1024, raised, "pointwise_mlp");
        if (rc != SPS_OK) return rc;
    }
    // one wave per output tile of the widest layer (4..16): the tiles of a layer are independent, a tile is a serial
    // chain of MFMAs fed by L2-latency weight loads, so waves are what hides that latency
    int tiles = c1 / 16;
    if (deep && c2 / 16 > tiles) tiles = c2 / 16;
    const int waves = tiles < 4 ? 4 : (tiles > PW_MAX_WAVES ? PW_MAX_WAVES : tiles);
    hipLaunchKernelGGL(pw_mlp_kernel, dim3(jcount / 16, b), dim3(64 * waves), lds, as_stream(stream), a);
    return check_launch("pw_mlp_kernel");
}

extern "C" int sps_pointwise_mlp(int b, int m, int cin, int c1, int c2, int c3_real, const float *x, const float *w1,
                                 const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                                 float *y1, float *y1_point_major, float *y3, sps_stream_t stream) {
    return sps_pointwise_mlp_range(b, m, 0, m, cin, c1, c2, c3_real, x, w1, b1, w2, b2, w3, b3, y1, y1_point_major, y3, stream);
}
