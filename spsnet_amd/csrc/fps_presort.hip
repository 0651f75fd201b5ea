// fps_presort.hip -- the spatial sort of the register-resident FPS kernel (fps_pruned.hip) as a pre-pass of its own.
//
// Inside fps_pruned_kernel one workgroup per scene sorts its scene first: two latency-bound passes over the points for the
// bounding box and the cell histogram, a scan, the scatter of point indices and an indexed (uncoalesced) load of every
// point -- ~75 us of a 1.79 ms launch on 8 of 256 compute units.  Here K workgroups per scene do the same sort in
// parallel (fps_sort_split.h: boxes, histograms and offsets exchanged through tagged granules and the workspace) and leave
// {x, y, z, running distance, rank} sorted in a workspace; the FPS kernel's PRESORT instantiation then starts with 5 P
// coalesced loads per lane.  Same sort key, same buckets up to the (arbitrary, irrelevant) order inside a cell.
#include "fps_sort_split.h"

namespace sps {

// work: per scene `stride` floats: 5 arrays of npad elements (x, y, z, t, rank), then the exchange area (zeroed by the launcher)
__global__ __launch_bounds__(PF_THREADS) void fps_presort_kernel(int b, int K, int n, int bs, int l2, int rb, int npad,
                                                                 long long stride, const float *__restrict__ dataset,
                                                                 const float *__restrict__ temp, float *__restrict__ work,
                                                                 int spread) {
    __shared__ PcSortShared sh;
    // blocks s, s + 8, s + 16, ... share an XCD (observed dispatch order): a scene's K workgroups sit on one L2
    // (spread: DIAGNOSTIC mapping that puts a scene's workgroups on consecutive blocks = different XCDs; tests run the
    //  cross-XCD form of every exchange with it)
    const int scene = spread ? (int)blockIdx.x / K : (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * K));
    const int cu = spread ? (int)blockIdx.x % K : (blockIdx.x >> 3) % K;
    if (scene >= b) return;
    const float *xyz = dataset + (size_t)scene * n * 3;
    if (temp) temp += (size_t)scene * n;
    float *sx = work + (size_t)scene * stride, *sy = sx + npad, *sz = sy + npad, *st = sz + npad;
    int *srk = reinterpret_cast<int *>(st + npad);
    unsigned long long *xg = reinterpret_cast<unsigned long long *>(work + (size_t)scene * stride + (size_t)5 * npad);
    pc_sort_split(sh, cu, K, n, npad, bs, l2, rb, xyz, temp, xg, sx, sy, sz, st, srk, false);   // the launch's end hands over
}

size_t fps_cluster_exchange_floats();   // fps_pruned_cluster.hip
int fps_cluster_spread();

// sorted scenes -> work (b * stride floats); temp may be NULL (all running distances 1e10)
int launch_fps_presort(int b, int n, const float *dataset, const float *temp, float *work, long long stride, hipStream_t st) {
    int K = PC_MAXK;
    while (K > 1 && b * K > 64) K >>= 1;   // the K workgroups of a scene spin on each other: all of them resident
    if (b * K > 128) return -1;
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int npad = divup(n, 64) * 64;
    hipError_t e = hipMemset2DAsync(work + (size_t)5 * npad, (size_t)stride * sizeof(float), 0,
                                    (size_t)PC_GRANULES * 8, (size_t)b, st);   // (the granules; the histograms behind them are written before they are read)
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "fps(presort): hipMemset2DAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(fps_presort_kernel, dim3(8 * K * divup(b, 8)), dim3(PF_THREADS), 0, st, b, K, n, bs, l2, rb, npad, stride,
                       dataset, temp, work, fps_cluster_spread());
    return check_launch("fps_presort_kernel");
}

}  // namespace sps
