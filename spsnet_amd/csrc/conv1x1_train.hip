// conv1x1_train.hip -- the 1x1 convolutions of the grouped MLPs in TRAINING (pointnet2_modules.py:203-209: Conv2d(k=1,
// bias=False) on (B, C, M, nsample) activations), forward, data gradient and weight gradient, in exact fp32 on the matrix
// cores (v_mfma_f32_16x16x4_f32 = an fmaf chain per output).  torch hands these to MIOpen, which picks NHWC implicit-GEMM
// kernels for them and pays for layout round trips: ~4.5 ms of a 12.6 ms training step of SA layers 0-2 at the IA-SSD
// shapes (0.9 ms of it batched transposes).  The activations stay channel-major (B, C, L), L = M * nsample:
//   * out = A in (forward: A = W; data gradient: A = W^T): a wave owns 64 columns; a lane reads 16 contiguous bytes of an
//     input row (4 rows x 256 B per instruction, fully coalesced) and element j of that float4 is column 4c + j of MFMA
//     column-tile j -- the same permutation on the way out makes the store a float4 again.
//   * dW = sum over all columns of dy (x) x: the reduction index is the column, so rows of dy / x are the MFMA rows and a
//     lane's float4 supplies k-slots (j, q) <-> column 4q + j of a 16-column block; a wave keeps a 64 x 64 block of dW
//     (16 accumulator tiles), the four waves of a workgroup take different column slices and meet in LDS; workgroup
//     partials are summed in a fixed order by a second kernel.
// Weight fragments ([row tile][k-step][lane] = A[16 t + i][4 ks + q], zero padded) are packed by the host.
#include "sps_common.h"

namespace sps {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 cv_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// out (b, co, l) = A (co x ci) in (b, ci, l); afrag = packed A; grid (ceil(l / 256), b, chunks), 4 waves of 64 columns.
// KS > 0: ci <= 4 KS input rows stay in registers and the workgroup loops over ALL 64-row chunks of the output (the input
// is read once); KS == 0: any ci, one chunk per workgroup (blockIdx.z), the input re-read per chunk through L2.
// (Tried: all row tiles of the output accumulating at once instead -- up to 256 accumulator registers, one wave per SIMD,
// nothing left to hide the loads: 13.7 vs 11.7 ms per training step.)
template <int KS>
__global__ __launch_bounds__(256) void conv1x1_apply_kernel(int ci, int co, long long l, const float *__restrict__ in,
                                                           const float *__restrict__ afrag, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
    const int wave = threadIdx.x >> 6;
    const long long col = ((long long)blockIdx.x * 4 + wave) * 64 + 4 * c;   // first of this lane's four columns
    const bool live = col < l;                                               // l % 4 == 0 (host)
    const int scene = blockIdx.y;
    const int ksteps = (ci + 3) >> 2;
    const float *src = in + (long long)scene * ci * l + col;
    float *dst = out + (long long)scene * co * l + col;
    f32x4 vin[KS > 0 ? KS : 1];
    if (KS > 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int row = 4 * ks + q;
            vin[ks] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (live && row < ci) vin[ks] = *reinterpret_cast<const f32x4 *>(src + (long long)row * l);
        }
    }
    const int chunk0 = KS > 0 ? 0 : blockIdx.z, chunk1 = KS > 0 ? (co + 63) >> 6 : blockIdx.z + 1;
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
        const int tiles = (co - 64 * chunk + 15) >> 4;                       // row tiles of this chunk that exist
        const float *af = afrag + ((long long)(4 * chunk) * ksteps) * 64 + lane;
        f32x4 acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto step = [&](int ks, const f32x4 v) {
            float a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = t < tiles ? af[((long long)t * ksteps + ks) * 64] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = cv_mfma(a[t], v[j], acc[t][j]);
        };
        if (KS > 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (ks < ksteps) step(ks, vin[ks]);
        } else {
            for (int ks = 0; ks < ksteps; ++ks) {
                const int row = 4 * ks + q;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (live && row < ci) v = *reinterpret_cast<const f32x4 *>(src + (long long)row * l);
                step(ks, v);
            }
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int orow = 64 * chunk + 16 * t + 4 * q + r;
                    if (orow < co)
                        *reinterpret_cast<f32x4 *>(dst + (long long)orow * l) =
                            (f32x4){acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
                }
        }
    }
}

// partial[blk][o][i] (64 x 64 block of dW) over this workgroup's column slice; grid (ceil(co / 64), ceil(ci / 64), b * splits)
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(int ci, int co, long long l, int splits, const float *__restrict__ x,
                                                            const float *__restrict__ dy, float *__restrict__ partial) {
    __shared__ float red[64 * 64];
    const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
    const int wave = threadIdx.x >> 6;
    const int ob = blockIdx.x, ib = blockIdx.y;
    const int scene = blockIdx.z / splits, split = blockIdx.z % splits;
    const long long blocks16 = l >> 4;                                       // l % 16 == 0 (host)
    const long long per = (blocks16 + (long long)splits * 4 - 1) / ((long long)splits * 4);
    const long long k0 = ((long long)split * 4 + wave) * per, k1 = (k0 + per < blocks16) ? k0 + per : blocks16;
    const float *dyb = dy + (long long)scene * co * l + 4 * q;
    const float *xb = x + (long long)scene * ci * l + 4 * q;
    int orow[4], irow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { orow[t] = 64 * ob + 16 * t + i; irow[t] = 64 * ib + 16 * t + i; }
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long long kb = k0; kb < k1; ++kb) {
        const long long col = kb << 4;
        f32x4 a[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bv[t] = a[t];
            if (orow[t] < co) a[t] = *reinterpret_cast<const f32x4 *>(dyb + (long long)orow[t] * l + col);
            if (irow[t] < ci) bv[t] = *reinterpret_cast<const f32x4 *>(xb + (long long)irow[t] * l + col);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = cv_mfma(a[t][j], bv[u][j], acc[t][u]);
    }
    // the four column slices meet in LDS, one wave at a time (fixed order)
    for (int e = threadIdx.x; e < 64 * 64; e += 256) red[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * t + 4 * q + r) * 64 + 16 * u + i] += acc[t][u][r];
        }
        __syncthreads();
    }
    float *out = partial + (((long long)blockIdx.z * gridDim.x + ob) * gridDim.y + ib) * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) out[e] = red[e];
}

// dW[o][i] = sum over the (scene, split) partial blocks: one wave per element, lane l adds blocks l, l + 64, ... and the
// lane sums meet in a butterfly -- a fixed order
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(int ci, int co, int nz, int nob, int nib,
                                                                   const float *__restrict__ partial, float *__restrict__ dw) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= co * ci) return;
    const int o = e / ci, i = e - o * ci;
    const int ob = o >> 6, ib = i >> 6;
    const float *p = partial + ((long long)ob * nib + ib) * 4096 + (o & 63) * 64 + (i & 63);
    float s = 0.f;
    for (int z = lane; z < nz; z += 64) s += p[(long long)z * nob * nib * 4096];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) dw[e] = s;
}

}  // namespace sps

using namespace sps;

extern "C" int sps_conv1x1_apply(int b, int ci, int co, long long l, const float *in, const float *afrag, float *out,
                                 sps_stream_t stream) {
    if (b < 0 || ci <= 0 || co <= 0 || l < 0) return fail(SPS_ERR_INVALID, "conv1x1_apply: bad shape b=%d ci=%d co=%d l=%lld", b, ci, co, l);
    if (b == 0 || l == 0) return SPS_OK;
    if (l % 4) return fail(SPS_ERR_INVALID, "conv1x1_apply: l = %lld must be a multiple of 4", l);
    if (!in || !afrag || !out) return fail(SPS_ERR_INVALID, "conv1x1_apply: null pointer");
    const long long gx = (l + 255) / 256;
    if (gx > 0x7fffffffLL || b > 65535 || divup(co, 64) > 65535) return fail(SPS_ERR_INVALID, "conv1x1_apply: grid too large");
    hipStream_t st = as_stream(stream);
    if (ci <= 16) hipLaunchKernelGGL(conv1x1_apply_kernel<4>, dim3((unsigned)gx, b, 1), dim3(256), 0, st, ci, co, l, in, afrag, out);
    else if (ci <= 32) hipLaunchKernelGGL(conv1x1_apply_kernel<8>, dim3((unsigned)gx, b, 1), dim3(256), 0, st, ci, co, l, in, afrag, out);
    else if (ci <= 68) hipLaunchKernelGGL(conv1x1_apply_kernel<17>, dim3((unsigned)gx, b, 1), dim3(256), 0, st, ci, co, l, in, afrag, out);
    else hipLaunchKernelGGL(conv1x1_apply_kernel<0>, dim3((unsigned)gx, b, divup(co, 64)), dim3(256), 0, st, ci, co, l, in, afrag, out);
    return check_launch("conv1x1_apply_kernel");
}

extern "C" long long sps_conv1x1_wgrad_workspace_floats(int b, int ci, int co, long long l) {
    if (b <= 0 || ci <= 0 || co <= 0 || l <= 0) return 0;
    const int nob = divup(co, 64), nib = divup(ci, 64);
    int splits = divup(512, nob * nib * b);
    splits = splits < 1 ? 1 : splits;
    while (splits > 1 && (long long)splits * 4 * 16 > l) --splits;
    return (long long)b * splits * nob * nib * 4096;
}

extern "C" int sps_conv1x1_wgrad(int b, int ci, int co, long long l, const float *x, const float *dy, float *dw, float *work,
                                 sps_stream_t stream) {
    if (b <= 0 || ci <= 0 || co <= 0 || l <= 0) return fail(SPS_ERR_INVALID, "conv1x1_wgrad: bad shape b=%d ci=%d co=%d l=%lld", b, ci, co, l);
    if (l % 16) return fail(SPS_ERR_INVALID, "conv1x1_wgrad: l = %lld must be a multiple of 16", l);
    if (!x || !dy || !dw || !work) return fail(SPS_ERR_INVALID, "conv1x1_wgrad: null pointer");
    const int nob = divup(co, 64), nib = divup(ci, 64);
    int splits = divup(512, nob * nib * b);
    splits = splits < 1 ? 1 : splits;
    while (splits > 1 && (long long)splits * 4 * 16 > l) --splits;
    if ((long long)b * splits > 65535) return fail(SPS_ERR_INVALID, "conv1x1_wgrad: grid too large");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(nob, nib, b * splits), dim3(256), 0, st, ci, co, l, splits, x, dy, work);
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(divup(co * ci, 4)), dim3(256), 0, st, ci, co, b * splits, nob, nib, work, dw);
    return check_launch("conv1x1_wgrad_kernel");
}
