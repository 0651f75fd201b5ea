// group_gather.hip -- index gathers of the SA layer and their gradients.
//
// Replaces gather_points(_grad)_kernel_fast (reference
// pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:8-24, 46-63) and
// group_points(_grad)_kernel_fast (group_points_gpu.cu:14-31, 53-71):
//   gather: out[b,c,j]   = points[b,c,idx[b,j]]
//   group : out[b,c,j,s] = points[b,c,idx[b,j,s]]
// A gather is a group with nsample = 1, so both share one kernel.  Each thread owns one
// (j,s) output column, reads its index ONCE and walks a chunk of channels (the reference
// launches one thread per (c,j,s) and re-reads idx C times); stores are coalesced along
// (j,s), the 4-byte source reads are the unavoidable random part and are served by L2.
#include "sps_common.h"

namespace sps {

constexpr int GG_THREADS = 256;
constexpr int GG_CCHUNK = 16;  // channels per workgroup (grid.z walks the chunks)

__global__ __launch_bounds__(GG_THREADS) void group_kernel(
    int c, int n, int cols, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out) {
    const int scene = blockIdx.y;
    const int e = blockIdx.x * GG_THREADS + threadIdx.x;
    if (e >= cols) return;
    const int src = idx[(size_t)scene * cols + e];
    const int c0 = blockIdx.z * GG_CCHUNK;
    const int c1 = (c0 + GG_CCHUNK < c) ? c0 + GG_CCHUNK : c;
    const float *p = points + ((size_t)scene * c + c0) * n + src;
    float *o = out + ((size_t)scene * c + c0) * cols + e;
    for (int ch = c0; ch < c1; ++ch, p += n, o += cols) *o = *p;
}

// (gstride: floats between two scenes of grad_out -- c * cols, or more when grad_out is a channel slice of a wider tensor)
__global__ __launch_bounds__(GG_THREADS) void group_grad_kernel(
    int c, int n, int cols, size_t gstride, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
    const int scene = blockIdx.y;
    const int e = blockIdx.x * GG_THREADS + threadIdx.x;
    if (e >= cols) return;
    const int dst = idx[(size_t)scene * cols + e];
    const int c0 = blockIdx.z * GG_CCHUNK;
    const int c1 = (c0 + GG_CCHUNK < c) ? c0 + GG_CCHUNK : c;
    const float *g = grad_out + (size_t)scene * gstride + (size_t)c0 * cols + e;
    float *p = grad_points + ((size_t)scene * c + c0) * n + dst;
    for (int ch = c0; ch < c1; ++ch, g += cols, p += n) atomicAdd(p, *g);
}

// The same sums through LDS: a workgroup owns one (scene, channel) row of grad_points (n floats: 64 KiB at 16 384 points),
// streams that row's grad_out columns coalesced (16 bytes per lane), merges runs of equal targets inside a lane (a ball
// with fewer than nsample points repeats its first hit) and adds with ds_add_f32; the finished row is written once.
// Global fp32 atomics ran at ~20 G/s here (885 us for the 16.8 M elements of IA-SSD layer 1).  Measured bound: ds_add_f32 retires
// about one lane per 3 cycles per CU whatever the address pattern (hot targets, uniform targets and 4x deeper load pipelining
// all give 340 us for B=8, C=24, 16 384 x 16 columns), i.e. ~200 G adds/s chip-wide -- 10x the global rate, 3x below the
// coalesced read of grad_out.  Summation order is unspecified, as with the reference's atomicAdd.
constexpr int GGL_THREADS = 512;
__global__ __launch_bounds__(GGL_THREADS) void group_grad_lds_kernel(int c, int n, int cols, size_t gstride,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx, float *__restrict__ grad_points) {
    extern __shared__ float gg_acc[];
    const int scene = blockIdx.y, ch = blockIdx.x;
    for (int k = threadIdx.x; k < n; k += GGL_THREADS) gg_acc[k] = 0.f;
    __syncthreads();
    const float *g = grad_out + (size_t)scene * gstride + (size_t)ch * cols;
    const int *ix = idx + (size_t)scene * cols;
    const bool vec = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ix)) & 15) == 0;
    const int cols4 = vec ? (cols & ~3) : 0;
    for (int e = threadIdx.x * 4; e < cols4; e += GGL_THREADS * 4) {
        const float4 v = *reinterpret_cast<const float4 *>(g + e);
        const int4 t = *reinterpret_cast<const int4 *>(ix + e);
        int cur = t.x;
        float sum = v.x;
        if (t.y == cur) sum += v.y; else { atomicAdd(&gg_acc[cur], sum); cur = t.y; sum = v.y; }
        if (t.z == cur) sum += v.z; else { atomicAdd(&gg_acc[cur], sum); cur = t.z; sum = v.z; }
        if (t.w == cur) sum += v.w; else { atomicAdd(&gg_acc[cur], sum); cur = t.w; sum = v.w; }
        atomicAdd(&gg_acc[cur], sum);
    }
    for (int e = cols4 + threadIdx.x; e < cols; e += GGL_THREADS) atomicAdd(&gg_acc[ix[e]], g[e]);
    __syncthreads();
    float *p = grad_points + ((size_t)scene * c + ch) * n;
    for (int k = threadIdx.x; k < n; k += GGL_THREADS) p[k] += gg_acc[k];
}

// new_xyz[b,j,:] = xyz[b,idx[b,j],:] on the native (B,N,3) layout: what the reference obtains with
// transpose + gather_operation + transpose (pointnet2_modules.py:261,423-424), without the two copies
// run_if: a device flag; the launch does nothing when it is zero (sa_stack's redo of a streamed layer whose bounded wait
// gave up).  The index is clamped into the cloud: a consumer that ran ahead of its producer reads garbage indices, which
// must not become wild addresses (the redo repairs the values).
__global__ __launch_bounds__(GG_THREADS) void gather_xyz_kernel(int n, int m, int j0, int jcount,
                                                                 const float *__restrict__ xyz,
                                                                 const int *__restrict__ idx, float *__restrict__ out,
                                                                 const int *__restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    const int scene = blockIdx.y;
    const int j = j0 + blockIdx.x * GG_THREADS + threadIdx.x;
    if (j >= j0 + jcount) return;
    int src = idx[(size_t)scene * m + j];
    src = src < 0 ? 0 : (src >= n ? n - 1 : src);
    const float *p = xyz + ((size_t)scene * n + src) * 3;
    float *o = out + ((size_t)scene * m + j) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

static int launch_group(bool grad, const char *what, int b, int c, int n, int npoints, int nsample,
                        const float *src, const int *idx, float *dst, hipStream_t st, long long grad_batch_stride = -1) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "%s: bad shape b=%d c=%d n=%d npoints=%d nsample=%d", what, b, c, n, npoints, nsample);
    const long long cols_ll = (long long)npoints * nsample;
    if (cols_ll > 0x7FFFFFFFLL) return fail(SPS_ERR_INVALID, "%s: npoints*nsample overflows int", what);
    const int cols = (int)cols_ll;
    if (b == 0 || c == 0 || cols == 0) return SPS_OK;
    if (n == 0) return fail(SPS_ERR_INVALID, "%s: n == 0 with a non-empty index", what);
    if (!src || !idx || !dst) return fail(SPS_ERR_INVALID, "%s: null pointer", what);
    if (b > 65535 || divup(c, GG_CCHUNK) > 65535) return fail(SPS_ERR_INVALID, "%s: grid too large", what);
    if (grad_batch_stride >= 0 && grad_batch_stride < (long long)c * cols)
        return fail(SPS_ERR_INVALID, "%s: batch stride %lld below c * npoints * nsample", what, grad_batch_stride);
    const size_t gstride = grad_batch_stride >= 0 ? (size_t)grad_batch_stride : (size_t)c * cols;
    if (grad && cols >= 1024 && (size_t)n * 4 <= 150 * 1024 && c <= 65535 && (long long)b * c >= 64) {
        const size_t lds = (size_t)n * 4;
        static LdsLimitOnce raised;
        if (lds > 64 * 1024) {
            const int rc = raise_lds_limit((const void *)group_grad_lds_kernel, 150 * 1024, raised, what);
            if (rc != SPS_OK) return rc;
        }
        hipLaunchKernelGGL(group_grad_lds_kernel, dim3(c, b), dim3(GGL_THREADS), lds, st, c, n, cols, gstride, src, idx, dst);
        return check_launch(what);
    }
    dim3 grid(divup(cols, GG_THREADS), b, divup(c, GG_CCHUNK)), block(GG_THREADS);
    if (grad) hipLaunchKernelGGL(group_grad_kernel, grid, block, 0, st, c, n, cols, gstride, src, idx, dst);
    else hipLaunchKernelGGL(group_kernel, grid, block, 0, st, c, n, cols, src, idx, dst);
    return check_launch(what);
}

}  // namespace sps

extern "C" int sps_gather_xyz(int b, int n, int m, const float *xyz, const int *idx, float *out, sps_stream_t stream) {
    return sps_gather_xyz_range(b, n, m, 0, m, xyz, idx, out, nullptr, stream);
}

extern "C" int sps_gather_xyz_range(int b, int n, int m, int j0, int jcount, const float *xyz, const int *idx, float *out,
                                    const int *run_if, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0 || m < 0 || j0 < 0 || jcount < 0 || j0 + jcount > m)
        return fail(SPS_ERR_INVALID, "gather_xyz: bad shape b=%d n=%d m=%d range [%d,+%d)", b, n, m, j0, jcount);
    if (b == 0 || jcount == 0) return SPS_OK;
    if (n == 0) return fail(SPS_ERR_INVALID, "gather_xyz: n == 0 with a non-empty index");
    if (!xyz || !idx || !out) return fail(SPS_ERR_INVALID, "gather_xyz: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "gather_xyz: grid too large");
    hipLaunchKernelGGL(gather_xyz_kernel, dim3(divup(jcount, GG_THREADS), b), dim3(GG_THREADS), 0, as_stream(stream), n, m, j0,
                       jcount, xyz, idx, out, run_if);
    return check_launch("gather_xyz_kernel");
}

extern "C" int sps_gather_points_kernel_launcher_fast(int b, int c, int n, int npoints, const float *points,
                                                      const int *idx, float *out, sps_stream_t stream) {
    return sps::launch_group(false, "gather_points", b, c, n, npoints, 1, points, idx, out, sps::as_stream(stream));
}

extern "C" int sps_gather_points_grad_kernel_launcher_fast(int b, int c, int n, int npoints,
                                                           const float *grad_out, const int *idx,
                                                           float *grad_points, sps_stream_t stream) {
    return sps::launch_group(true, "gather_points_grad", b, c, n, npoints, 1, grad_out, idx, grad_points,
                             sps::as_stream(stream));
}

extern "C" int sps_group_points_kernel_launcher_fast(int b, int c, int n, int npoints, int nsample,
                                                     const float *points, const int *idx, float *out,
                                                     sps_stream_t stream) {
    return sps::launch_group(false, "group_points", b, c, n, npoints, nsample, points, idx, out,
                             sps::as_stream(stream));
}

extern "C" int sps_group_points_grad_kernel_launcher_fast(int b, int c, int n, int npoints, int nsample,
                                                          const float *grad_out, const int *idx,
                                                          float *grad_points, sps_stream_t stream) {
    return sps::launch_group(true, "group_points_grad", b, c, n, npoints, nsample, grad_out, idx, grad_points,
                             sps::as_stream(stream));
}

extern "C" int sps_group_points_grad_strided(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                             long long grad_batch_stride, const int *idx, float *grad_points,
                                             sps_stream_t stream) {
    if (grad_batch_stride < 0) return sps::fail(SPS_ERR_INVALID, "group_points_grad_strided: negative batch stride");
    return sps::launch_group(true, "group_points_grad_strided", b, c, n, npoints, nsample, grad_out, idx, grad_points,
                             sps::as_stream(stream), grad_batch_stride);
}

// ---- deterministic gradients ---------------------------------------------------------------------------------
// group_points_grad / gather_points_grad scatter with atomicAdd (reference group_points_gpu.cu:53-71,
// sampling_gpu.cu:46-63): the fp32 summation order, hence the low bits of the gradient, change from run to run.
// sps_index_add_deterministic computes the same sums in a FIXED order -- ascending column e = (j, s), which is the
// order a sequential CPU loop (and the oracle) uses, so the result is bit-identical to the oracle:
//   1. count the columns per target point (integer atomics are order-independent), prefix-sum to segment offsets;
//   2. scatter the column numbers into their segments (arbitrary order), sort every segment ascending;
//   3. one thread per point walks its segment for every channel.
namespace sps {

constexpr int IA_THREADS = 256;

__global__ __launch_bounds__(IA_THREADS) void ia_count_kernel(int n, int cols, const int *__restrict__ idx, int *__restrict__ offs) {
    const int scene = blockIdx.y, e = blockIdx.x * IA_THREADS + threadIdx.x;
    if (e < cols) atomicAdd(&offs[(size_t)scene * (n + 1) + idx[(size_t)scene * cols + e] + 1], 1);
}

// in place: offs[scene][i] = number of columns whose target is < i (offs[scene][0] = 0 already)
__global__ __launch_bounds__(1024) void ia_scan_kernel(int n, int *__restrict__ offs, int *__restrict__ cursor) {
    __shared__ int wsum[16];
    __shared__ int carry;
    int *o = offs + (size_t)blockIdx.x * (n + 1) + 1;
    int *cur = cursor + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? o[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < n) {
            o[i] = before + incl;          // offs[i + 1] = end of segment i
            cur[i] = before + incl - v;    // start of segment i: the fill cursor
        }
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(IA_THREADS) void ia_fill_kernel(int n, int cols, const int *__restrict__ idx, int *__restrict__ cursor,
                                                             int *__restrict__ list) {
    const int scene = blockIdx.y, e = blockIdx.x * IA_THREADS + threadIdx.x;
    if (e >= cols) return;
    const int pos = atomicAdd(&cursor[(size_t)scene * n + idx[(size_t)scene * cols + e]], 1);
    list[(size_t)scene * cols + pos] = e;
}

__global__ __launch_bounds__(IA_THREADS) void ia_sort_kernel(int n, int cols, const int *__restrict__ offs, int *__restrict__ list) {
    const int scene = blockIdx.y, p = blockIdx.x * IA_THREADS + threadIdx.x;
    if (p >= n) return;
    const int *o = offs + (size_t)scene * (n + 1);
    int *seg = list + (size_t)scene * cols + o[p];
    const int len = o[p + 1] - o[p];
    for (int i = 1; i < len; ++i) {  // insertion sort: segments are short (columns per point ~ M*nsample/N)
        const int v = seg[i];
        int k = i - 1;
        while (k >= 0 && seg[k] > v) { seg[k + 1] = seg[k]; --k; }
        seg[k + 1] = v;
    }
}

__global__ __launch_bounds__(IA_THREADS) void ia_sum_kernel(int c, int n, int cols, const float *__restrict__ grad_out,
                                                            const int *__restrict__ offs, const int *__restrict__ list,
                                                            float *__restrict__ grad_points) {
    const int scene = blockIdx.y, p = blockIdx.x * IA_THREADS + threadIdx.x;
    if (p >= n) return;
    const int *o = offs + (size_t)scene * (n + 1);
    const int beg = o[p], end = o[p + 1];
    const int *seg = list + (size_t)scene * cols;
    const int c0 = blockIdx.z * GG_CCHUNK;
    const int c1 = (c0 + GG_CCHUNK < c) ? c0 + GG_CCHUNK : c;
    for (int ch = c0; ch < c1; ++ch) {
        const float *g = grad_out + ((size_t)scene * c + ch) * cols;
        float *dst = grad_points + ((size_t)scene * c + ch) * n + p;
        float acc = *dst;  // accumulate onto the caller's buffer, like the atomic kernels do
        for (int k = beg; k < end; ++k) acc += g[seg[k]];
        *dst = acc;
    }
}

}  // namespace sps

extern "C" long long sps_index_add_workspace_ints(int b, int n, int cols) {
    return (long long)b * ((long long)(n + 1) + n + cols);
}

// grad_points (b, c, n) += scatter of grad_out (b, c, cols) by idx (b, cols), summed per target in ascending column
// order (cols = npoints * nsample for group_points_grad, npoints for gather_points_grad).  work: device ints,
// sps_index_add_workspace_ints(b, n, cols) of them.
extern "C" int sps_index_add_deterministic(int b, int c, int n, int cols, const float *grad_out, const int *idx,
                                           float *grad_points, int *work, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || c < 0 || n < 0 || cols < 0) return fail(SPS_ERR_INVALID, "index_add: bad shape b=%d c=%d n=%d cols=%d", b, c, n, cols);
    if (b == 0 || c == 0 || cols == 0) return SPS_OK;
    if (n == 0) return fail(SPS_ERR_INVALID, "index_add: indices into an empty tensor");
    if (!grad_out || !idx || !grad_points || !work) return fail(SPS_ERR_INVALID, "index_add: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "index_add: batch %d exceeds the grid limit", b);
    hipStream_t st = as_stream(stream);
    int *offs = work, *cursor = work + (size_t)b * (n + 1), *list = cursor + (size_t)b * n;
    if (hipMemsetAsync(offs, 0, sizeof(int) * (size_t)b * (n + 1), st) != hipSuccess) return fail(SPS_ERR_LAUNCH, "index_add: memset failed");
    hipLaunchKernelGGL(ia_count_kernel, dim3(divup(cols, IA_THREADS), b), dim3(IA_THREADS), 0, st, n, cols, idx, offs);
    hipLaunchKernelGGL(ia_scan_kernel, dim3(b), dim3(1024), 0, st, n, offs, cursor);
    hipLaunchKernelGGL(ia_fill_kernel, dim3(divup(cols, IA_THREADS), b), dim3(IA_THREADS), 0, st, n, cols, idx, cursor, list);
    hipLaunchKernelGGL(ia_sort_kernel, dim3(divup(n, IA_THREADS), b), dim3(IA_THREADS), 0, st, n, cols, offs, list);
    hipLaunchKernelGGL(ia_sum_kernel, dim3(divup(n, IA_THREADS), b, divup(c, GG_CCHUNK)), dim3(IA_THREADS), 0, st, c, n, cols,
                       grad_out, offs, list, grad_points);
    return check_launch("index_add_deterministic");
}


// ---- max over the samples of a group, with its gradient -------------------------------------------------------------------
// F.max_pool2d(x, kernel_size=[1, nsample]) of the SA modules (pointnet2_modules.py:441-444) on a contiguous
// (rows = B*C*M, nsample) view: torch's generic NCHW pooling kernels took 2.2 + 0.7 ms of a 23 ms training step at the
// IA-SSD shapes.  Same values and the same gradient routing: the FIRST maximum of a row wins (strict '>'); a NaN wins and
// propagates, and of several NaNs the LAST one keeps the index, as in torch's max_pool2d kernels.
namespace sps {

// does (b, bi) replace (a, ai) in the serial scan's outcome?  A NaN beats numbers and an earlier NaN (torch's max_pool2d tests
// `val > max || isnan(val)`, so the LAST NaN keeps the index); among numbers the larger value, then the smaller index
__device__ __forceinline__ bool pool_takes(float b, int bi, float a, int ai) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return bn && (!an || bi > ai);
    return b > a || (b == a && bi < ai);
}

// G = nsample / 4 lanes share a row (16 bytes each: the wave reads 1 KiB contiguous), then a G-lane butterfly on
// (value, index) whose order is total, so every lane ends with the serial scan's answer
template <int G>
__global__ __launch_bounds__(256) void pool_max_fwd_coop_kernel(long long rows, const float *__restrict__ x,
                                                                float *__restrict__ out, unsigned char *__restrict__ arg) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = e / G;
    const int sub = (int)(e - r * G);
    const bool live = r < rows;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = *reinterpret_cast<const float4 *>(x + e * 4);
    const float q[4] = {v.x, v.y, v.z, v.w};
    float best = q[0];
    int bi = sub * 4;
#pragma unroll
    for (int u = 1; u < 4; ++u)
        if (q[u] > best || q[u] != q[u]) { best = q[u]; bi = sub * 4 + u; }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (pool_takes(ob, oi, best, bi)) { best = ob; bi = oi; }
    }
    if (live && sub == 0) {
        out[r] = best;
        arg[r] = (unsigned char)bi;
    }
}

__global__ __launch_bounds__(256) void pool_max_fwd_kernel(long long rows, int ns, const float *__restrict__ x,
                                                           float *__restrict__ out, unsigned char *__restrict__ arg) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = x + r * ns;
    float best = p[0];
    int bi = 0;
    for (int s = 1; s < ns; ++s) {
        const float e = p[s];
        if (e > best || e != e) { best = e; bi = s; }
    }
    out[r] = best;
    arg[r] = (unsigned char)bi;
}

// one float4 of grad_in per thread when nsample % 4 == 0
__global__ __launch_bounds__(256) void pool_max_bwd4_kernel(long long total4, int ns4, const float *__restrict__ grad_out,
                                                            const unsigned char *__restrict__ arg, float *__restrict__ grad_in) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    const long long r = e / ns4;
    const int s = (int)(e - r * ns4) * 4;
    const int a = (int)arg[r] - s;
    const float g = ((unsigned)a < 4u) ? grad_out[r] : 0.f;
    *reinterpret_cast<float4 *>(grad_in + e * 4) = make_float4(a == 0 ? g : 0.f, a == 1 ? g : 0.f, a == 2 ? g : 0.f, a == 3 ? g : 0.f);
}

__global__ __launch_bounds__(256) void pool_max_bwd_kernel(long long total, int ns, const float *__restrict__ grad_out,
                                                           const unsigned char *__restrict__ arg, float *__restrict__ grad_in) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one element of grad_in per thread: coalesced
    if (e >= total) return;
    const long long r = e / ns;
    const int s = (int)(e - r * ns);
    grad_in[e] = (s == (int)arg[r]) ? grad_out[r] : 0.f;
}

}  // namespace sps

extern "C" int sps_pool_max_fwd(long long rows, int nsample, const float *x, float *out, unsigned char *arg, sps_stream_t stream) {
    using namespace sps;
    if (rows < 0 || nsample <= 0 || nsample > 255) return fail(SPS_ERR_INVALID, "pool_max: bad shape rows=%lld nsample=%d", rows, nsample);
    if (rows == 0) return SPS_OK;
    if (!x || !out || !arg) return fail(SPS_ERR_INVALID, "pool_max: null pointer");
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (aligned && (nsample == 16 || nsample == 32 || nsample == 64)) {
        const int lanes = nsample / 4;
        const long long gc = (rows * lanes + 255) / 256;
        if (gc > 0x7fffffffLL) return fail(SPS_ERR_INVALID, "pool_max: too many rows");
        if (lanes == 4) hipLaunchKernelGGL(pool_max_fwd_coop_kernel<4>, dim3((unsigned)gc), dim3(256), 0, as_stream(stream), rows, x, out, arg);
        else if (lanes == 8) hipLaunchKernelGGL(pool_max_fwd_coop_kernel<8>, dim3((unsigned)gc), dim3(256), 0, as_stream(stream), rows, x, out, arg);
        else hipLaunchKernelGGL(pool_max_fwd_coop_kernel<16>, dim3((unsigned)gc), dim3(256), 0, as_stream(stream), rows, x, out, arg);
        return check_launch("pool_max_fwd_coop_kernel");
    }
    const long long g = (rows + 255) / 256;
    if (g > 0x7fffffffLL) return fail(SPS_ERR_INVALID, "pool_max: too many rows");
    hipLaunchKernelGGL(pool_max_fwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), rows, nsample, x, out, arg);
    return check_launch("pool_max_fwd_kernel");
}

extern "C" int sps_pool_max_bwd(long long rows, int nsample, const float *grad_out, const unsigned char *arg, float *grad_in,
                                sps_stream_t stream) {
    using namespace sps;
    if (rows < 0 || nsample <= 0 || nsample > 255) return fail(SPS_ERR_INVALID, "pool_max_grad: bad shape rows=%lld nsample=%d", rows, nsample);
    if (rows == 0) return SPS_OK;
    if (!grad_out || !arg || !grad_in) return fail(SPS_ERR_INVALID, "pool_max_grad: null pointer");
    if ((nsample & 3) == 0 && (reinterpret_cast<uintptr_t>(grad_in) & 15) == 0) {
        const long long total4 = rows * (nsample / 4), g4 = (total4 + 255) / 256;
        if (g4 > 0x7fffffffLL) return fail(SPS_ERR_INVALID, "pool_max_grad: too many elements");
        hipLaunchKernelGGL(pool_max_bwd4_kernel, dim3((unsigned)g4), dim3(256), 0, as_stream(stream), total4, nsample / 4, grad_out, arg, grad_in);
        return check_launch("pool_max_bwd4_kernel");
    }
    const long long total = rows * nsample, g = (total + 255) / 256;
    if (g > 0x7fffffffLL) return fail(SPS_ERR_INVALID, "pool_max_grad: too many elements");
    hipLaunchKernelGGL(pool_max_bwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), total, nsample, grad_out, arg, grad_in);
    return check_launch("pool_max_bwd_kernel");
}
