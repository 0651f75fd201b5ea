// bn_relu_train.hip -- BatchNorm2d (batch statistics) + ReLU of the grouped MLPs in TRAINING mode, forward and backward
// (pointnet2_modules.py:203-209 builds [Conv2d, BatchNorm2d, ReLU] stacks; in train() BatchNorm normalises with the
// statistics of the batch).  torch dispatches these to MIOpen's spatial BatchNorm kernels, which ran at ~0.6 TB/s on the
// (B, C, M, nsample) activations of the SA layers (3.2 + 3.0 ms of a 17.6 ms training step at the IA-SSD shapes) plus
// separate ReLU kernels.  Here: one coalesced pass for the statistics (fp64 accumulation, fixed-order two-stage
// reduction: bit-reproducible), one fused normalise + affine + ReLU pass; backward = one reduction pass (d gamma, d beta
// with the ReLU mask folded in) and one elementwise pass.  Semantics are torch's: biased variance for normalising,
// unbiased for running_var, running = (1 - momentum) running + momentum batch.
#include "sps_common.h"

namespace sps {

constexpr int BN_THREADS = 256;
constexpr int BN_CHUNK = 8192;  // elements of one (scene, channel) row per workgroup

struct BnShape {
    int b, c;
    long long l;        // elements per (scene, channel) row: M * nsample
    int chunks_per_row; // ceil(l / BN_CHUNK)
};

// elements [l0, l1) of a row, four per lane and trip when the row allows 16-byte accesses (vec), else one
#define BN_FOR_EACH(VEC, BODY4, BODY1)                                                       \
    if (VEC) {                                                                               \
        for (long long e = l0 + 4 * threadIdx.x; e < l1; e += 4 * BN_THREADS) { BODY4 }      \
    } else {                                                                                 \
        for (long long e = l0 + threadIdx.x; e < l1; e += BN_THREADS) { BODY1 }              \
    }

__device__ __forceinline__ double block_sum(double v, double *sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < BN_THREADS / 64; ++w) t += sh[w];  // fixed order
    return t;
}

// partial[(c * nchunks + chunk) * 2 + {0, 1}] = sum, sum of squares of one chunk
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(BnShape s, const float *__restrict__ x, double *__restrict__ partial) {
    __shared__ double sh[BN_THREADS / 64];
    const int ch = blockIdx.x, chunk = blockIdx.y;
    const int scene = chunk / s.chunks_per_row, part = chunk % s.chunks_per_row;
    const long long l0 = (long long)part * BN_CHUNK;
    const long long l1 = (l0 + BN_CHUNK < s.l) ? l0 + BN_CHUNK : s.l;
    const float *row = x + ((long long)scene * s.c + ch) * s.l;
    double a = 0.0, q = 0.0;
    const bool vec = (s.l & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    BN_FOR_EACH(vec,
                const float4 v4 = *reinterpret_cast<const float4 *>(row + e);
                const double v0 = v4.x; const double v1 = v4.y; const double v2 = v4.z; const double v3 = v4.w;
                a += (v0 + v1) + (v2 + v3);
                q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);,
                const double v = row[e];
                a += v;
                q += v * v;)
    const double ta = block_sum(a, sh), tq = block_sum(q, sh);
    if (threadIdx.x == 0) {
        double *o = partial + ((long long)ch * gridDim.y + chunk) * 2;
        o[0] = ta; o[1] = tq;
    }
}

// one thread per channel: mean, invstd, scale/shift of the fused pass, running statistics
__global__ void bn_finalize_kernel(int c, int nchunks, double count, const double *__restrict__ partial,
                                   const float *__restrict__ weight, const float *__restrict__ bias, float eps, float momentum,
                                   float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ mean,
                                   float *__restrict__ invstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double a = 0.0, q = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        a += partial[((long long)ch * nchunks + k) * 2];
        q += partial[((long long)ch * nchunks + k) * 2 + 1];
    }
    const double m = a / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[ch] = (float)m;
    invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
    (void)weight; (void)bias;
}

// y = relu(((x - mean) * invstd) * weight + bias), torch's operation order (the backward recomputes its sign from x)
__global__ __launch_bounds__(BN_THREADS) void bn_apply_relu_kernel(BnShape s, const float *__restrict__ x,
                                                                   const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                   const float *__restrict__ weight, const float *__restrict__ bias,
                                                                   float *__restrict__ y) {
    const int ch = blockIdx.x, chunk = blockIdx.y;
    const int scene = chunk / s.chunks_per_row, part = chunk % s.chunks_per_row;
    const long long l0 = (long long)part * BN_CHUNK;
    const long long l1 = (l0 + BN_CHUNK < s.l) ? l0 + BN_CHUNK : s.l;
    const long long base = ((long long)scene * s.c + ch) * s.l;
    const float m = mean[ch], is = invstd[ch], w = weight ? weight[ch] : 1.f, bb = bias ? bias[ch] : 0.f;
    const bool vec = (s.l & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    auto f = [&](float xv) { const float v = ((xv - m) * is) * w + bb; return v > 0.f ? v : 0.f; };
    BN_FOR_EACH(vec,
                const float4 v4 = *reinterpret_cast<const float4 *>(x + base + e);
                *reinterpret_cast<float4 *>(y + base + e) = make_float4(f(v4.x), f(v4.y), f(v4.z), f(v4.w));,
                y[base + e] = f(x[base + e]);)
}

// partial sums of dy' and dy' * xhat per chunk (dy' = dy where the output was positive)
// (the ReLU mask is recomputed from x with the forward's own expression, bit for bit, instead of reading y)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(BnShape s, const float *__restrict__ x,
                                                                   const float *__restrict__ dy, const float *__restrict__ mean,
                                                                   const float *__restrict__ invstd, const float *__restrict__ weight,
                                                                   const float *__restrict__ bias, double *__restrict__ partial) {
    __shared__ double sh[BN_THREADS / 64];
    const int ch = blockIdx.x, chunk = blockIdx.y;
    const int scene = chunk / s.chunks_per_row, part = chunk % s.chunks_per_row;
    const long long l0 = (long long)part * BN_CHUNK;
    const long long l1 = (l0 + BN_CHUNK < s.l) ? l0 + BN_CHUNK : s.l;
    const long long base = ((long long)scene * s.c + ch) * s.l;
    const float m = mean[ch], is = invstd[ch], w = weight ? weight[ch] : 1.f, bb = bias ? bias[ch] : 0.f;
    double a = 0.0, q = 0.0;
    const bool vec = (s.l & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    auto f = [&](float xv, float dv) {
        const float xh = (xv - m) * is;
        const float g = (xh * w + bb) > 0.f ? dv : 0.f;
        a += g;
        q += (double)g * xh;
    };
    BN_FOR_EACH(vec,
                const float4 xv = *reinterpret_cast<const float4 *>(x + base + e);
                const float4 dv = *reinterpret_cast<const float4 *>(dy + base + e);
                f(xv.x, dv.x); f(xv.y, dv.y); f(xv.z, dv.z); f(xv.w, dv.w);,
                f(x[base + e], dy[base + e]);)
    const double ta = block_sum(a, sh), tq = block_sum(q, sh);
    if (threadIdx.x == 0) {
        double *o = partial + ((long long)ch * gridDim.y + chunk) * 2;
        o[0] = ta; o[1] = tq;
    }
}

__global__ void bn_bwd_finalize_kernel(int c, int nchunks, const double *__restrict__ partial, float *__restrict__ dweight,
                                       float *__restrict__ dbias, float *__restrict__ sum_dy, float *__restrict__ sum_dy_xh) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double a = 0.0, q = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        a += partial[((long long)ch * nchunks + k) * 2];
        q += partial[((long long)ch * nchunks + k) * 2 + 1];
    }
    sum_dy[ch] = (float)a;
    sum_dy_xh[ch] = (float)q;
    if (dbias) dbias[ch] = (float)a;
    if (dweight) dweight[ch] = (float)q;
}

// dx = weight * invstd * (dy' - mean(dy') - xhat * mean(dy' * xhat))
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(BnShape s, float inv_count, const float *__restrict__ x,
                                                                  const float *__restrict__ dy,
                                                                  const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                  const float *__restrict__ weight, const float *__restrict__ bias,
                                                                  const float *__restrict__ sum_dy,
                                                                  const float *__restrict__ sum_dy_xh, float *__restrict__ dx) {
    const int ch = blockIdx.x, chunk = blockIdx.y;
    const int scene = chunk / s.chunks_per_row, part = chunk % s.chunks_per_row;
    const long long l0 = (long long)part * BN_CHUNK;
    const long long l1 = (l0 + BN_CHUNK < s.l) ? l0 + BN_CHUNK : s.l;
    const long long base = ((long long)scene * s.c + ch) * s.l;
    const float m = mean[ch], is = invstd[ch], w = weight ? weight[ch] : 1.f, bb = bias ? bias[ch] : 0.f;
    const float k1 = sum_dy[ch] * inv_count, k2 = sum_dy_xh[ch] * inv_count, sc = w * is;
    const bool vec = (s.l & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    auto f = [&](float xv, float dv) {
        const float xh = (xv - m) * is;
        const float g = (xh * w + bb) > 0.f ? dv : 0.f;
        return sc * (g - k1 - xh * k2);
    };
    BN_FOR_EACH(vec,
                const float4 xv = *reinterpret_cast<const float4 *>(x + base + e);
                const float4 dv = *reinterpret_cast<const float4 *>(dy + base + e);
                *reinterpret_cast<float4 *>(dx + base + e) = make_float4(f(xv.x, dv.x), f(xv.y, dv.y), f(xv.z, dv.z), f(xv.w, dv.w));,
                dx[base + e] = f(x[base + e], dy[base + e]);)
}

static int bn_shape(const char *what, int b, int c, long long l, BnShape *s) {
    if (b <= 0 || c <= 0 || l <= 0) return fail(SPS_ERR_INVALID, "%s: bad shape b=%d c=%d l=%lld", what, b, c, l);
    s->b = b; s->c = c; s->l = l;
    const long long cpr = (l + BN_CHUNK - 1) / BN_CHUNK;
    if (cpr * b > 65535 || c > 0x7fffffff) return fail(SPS_ERR_INVALID, "%s: tensor too large for the launch grid", what);
    s->chunks_per_row = (int)cpr;
    return SPS_OK;
}

}  // namespace sps

using namespace sps;

extern "C" long long sps_bn_train_workspace_doubles(int b, int c, long long l) {
    if (b <= 0 || c <= 0 || l <= 0) return 0;
    return 2LL * c * b * ((l + BN_CHUNK - 1) / BN_CHUNK);
}

extern "C" int sps_bn_relu_train_fwd(int b, int c, long long l, const float *x, const float *weight, const float *bias,
                                     float eps, float momentum, float *running_mean, float *running_var, float *mean,
                                     float *invstd, float *y, double *work, sps_stream_t stream) {
    BnShape s;
    if (int rc = bn_shape("bn_relu_train_fwd", b, c, l, &s)) return rc;
    if (!x || !mean || !invstd || !y || !work) return fail(SPS_ERR_INVALID, "bn_relu_train_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int nchunks = b * s.chunks_per_row;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(c, nchunks), dim3(BN_THREADS), 0, st, s, x, work);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(divup(c, 64)), dim3(64), 0, st, c, nchunks, (double)b * (double)l, work, weight, bias,
                       eps, momentum, running_mean, running_var, mean, invstd);
    hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(c, nchunks), dim3(BN_THREADS), 0, st, s, x, mean, invstd, weight, bias, y);
    return check_launch("bn_relu_train_fwd");
}

extern "C" int sps_bn_relu_train_bwd(int b, int c, long long l, const float *x, const float *dy, const float *mean,
                                     const float *invstd, const float *weight, const float *bias, float *dx, float *dweight,
                                     float *dbias, float *scratch2c, double *work, sps_stream_t stream) {
    BnShape s;
    if (int rc = bn_shape("bn_relu_train_bwd", b, c, l, &s)) return rc;
    if (!x || !dy || !mean || !invstd || !dx || !scratch2c || !work) return fail(SPS_ERR_INVALID, "bn_relu_train_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int nchunks = b * s.chunks_per_row;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(c, nchunks), dim3(BN_THREADS), 0, st, s, x, dy, mean, invstd, weight, bias, work);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(divup(c, 64)), dim3(64), 0, st, c, nchunks, work, dweight, dbias, scratch2c,
                       scratch2c + c);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(c, nchunks), dim3(BN_THREADS), 0, st, s, (float)(1.0 / ((double)b * (double)l)), x,
                       dy, mean, invstd, weight, bias, scratch2c, scratch2c + c, dx);
    return check_launch("bn_relu_train_bwd");
}
