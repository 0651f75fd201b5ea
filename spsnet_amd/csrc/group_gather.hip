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

__global__ __launch_bounds__(GG_THREADS) void group_grad_kernel(
    int c, int n, int cols, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
    const int scene = blockIdx.y;
    const int e = blockIdx.x * GG_THREADS + threadIdx.x;
    if (e >= cols) return;
    const int dst = idx[(size_t)scene * cols + e];
    const int c0 = blockIdx.z * GG_CCHUNK;
    const int c1 = (c0 + GG_CCHUNK < c) ? c0 + GG_CCHUNK : c;
    const float *g = grad_out + ((size_t)scene * c + c0) * cols + e;
    float *p = grad_points + ((size_t)scene * c + c0) * n + dst;
    for (int ch = c0; ch < c1; ++ch, g += cols, p += n) atomicAdd(p, *g);
}

// new_xyz[b,j,:] = xyz[b,idx[b,j],:] on the native (B,N,3) layout: what the reference obtains with
// transpose + gather_operation + transpose (pointnet2_modules.py:261,423-424), without the two copies
__global__ __launch_bounds__(GG_THREADS) void gather_xyz_kernel(int n, int m, int j0, int jcount,
                                                                 const float *__restrict__ xyz,
                                                                 const int *__restrict__ idx, float *__restrict__ out) {
    const int scene = blockIdx.y;
    const int j = j0 + blockIdx.x * GG_THREADS + threadIdx.x;
    if (j >= j0 + jcount) return;
    const float *p = xyz + ((size_t)scene * n + idx[(size_t)scene * m + j]) * 3;
    float *o = out + ((size_t)scene * m + j) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

static int launch_group(bool grad, const char *what, int b, int c, int n, int npoints, int nsample,
                        const float *src, const int *idx, float *dst, hipStream_t st) {
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "%s: bad shape b=%d c=%d n=%d npoints=%d nsample=%d", what, b, c, n, npoints, nsample);
    const long long cols_ll = (long long)npoints * nsample;
    if (cols_ll > 0x7FFFFFFFLL) return fail(SPS_ERR_INVALID, "%s: npoints*nsample overflows int", what);
    const int cols = (int)cols_ll;
    if (b == 0 || c == 0 || cols == 0) return SPS_OK;
    if (n == 0) return fail(SPS_ERR_INVALID, "%s: n == 0 with a non-empty index", what);
    if (!src || !idx || !dst) return fail(SPS_ERR_INVALID, "%s: null pointer", what);
    if (b > 65535 || divup(c, GG_CCHUNK) > 65535) return fail(SPS_ERR_INVALID, "%s: grid too large", what);
    dim3 grid(divup(cols, GG_THREADS), b, divup(c, GG_CCHUNK)), block(GG_THREADS);
    if (grad) hipLaunchKernelGGL(group_grad_kernel, grid, block, 0, st, c, n, cols, src, idx, dst);
    else hipLaunchKernelGGL(group_kernel, grid, block, 0, st, c, n, cols, src, idx, dst);
    return check_launch(what);
}

}  // namespace sps

extern "C" int sps_gather_xyz(int b, int n, int m, const float *xyz, const int *idx, float *out, sps_stream_t stream) {
    return sps_gather_xyz_range(b, n, m, 0, m, xyz, idx, out, stream);
}

extern "C" int sps_gather_xyz_range(int b, int n, int m, int j0, int jcount, const float *xyz, const int *idx, float *out,
                                    sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0 || m < 0 || j0 < 0 || jcount < 0 || j0 + jcount > m)
        return fail(SPS_ERR_INVALID, "gather_xyz: bad shape b=%d n=%d m=%d range [%d,+%d)", b, n, m, j0, jcount);
    if (b == 0 || jcount == 0) return SPS_OK;
    if (n == 0) return fail(SPS_ERR_INVALID, "gather_xyz: n == 0 with a non-empty index");
    if (!xyz || !idx || !out) return fail(SPS_ERR_INVALID, "gather_xyz: null pointer");
    if (b > 65535) return fail(SPS_ERR_INVALID, "gather_xyz: grid too large");
    hipLaunchKernelGGL(gather_xyz_kernel, dim3(divup(jcount, GG_THREADS), b), dim3(GG_THREADS), 0, as_stream(stream), n, m, j0,
                       jcount, xyz, idx, out);
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
