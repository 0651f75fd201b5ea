// ball_query_grid.hip -- the reference's ball query (ball_query_gpu.cu:9-45, 70-117: first `nsample` points in ascending
// index with d2 < r^2) WITHOUT the O(N x M) scan, for launches where both the cloud and the centroid set are large
// (DenseEdgeConv: M = N = 16384, r = 0.8; the SA layers' first level).
//
// The brute-force kernels (ball_query.hip) are already VALU-bound: 7.5 instructions per (centroid, point) pair.  The
// only way further down is to test fewer pairs -- but the result depends on the ORDER of the points (the first nsample
// hits by index), so a cell grid alone does not do: the candidates of a ball must be visited in ascending index.
//
//   1. setup/count/scan/scatter: a uniform grid over the cloud's bounding box (cell edge ~ radius, at most 65536
//      cells, x fastest), a counting sort of the point ids by cell (order inside a cell is arbitrary) and the same for
//      the centroids (which only need to be grouped spatially).
//   2. query: a wave owns 64 centroids that are adjacent in cell order.  The box that holds their balls selects a range
//      of cells per (y, z) row -- contiguous in the sorted id array because x runs fastest; every id in those ranges
//      sets its bit in an N-bit bitmap in LDS.  Reading the bitmap back in word order IS the ascending-index order, so
//      the wave walks the set bits, stages the candidates' coordinates in LDS 64 at a time and runs the same ordered
//      append loop as the brute-force kernel, lane per centroid, leaving as soon as all 64 rows are full.
//
// Exactness: the candidate set is a superset of every ball (cell coordinates are monotone in the coordinate and the
// box is padded beyond fp32 rounding), the hit test is the reference's fp32 expression, the visiting order is
// ascending index -- rows are bit-identical to the scan's.  Nothing depends on the arbitrary order inside a cell.
#include "sps_common.h"

#include <stdlib.h>

namespace sps {

constexpr int BQG_NCMAX = 65536;   // cells per scene
constexpr int BQG_HDR = 48;        // ints: lo[3], inv, dim[3], nc | shift[3], key bits, keys | 16 x (axis << 8 | bit)
constexpr int BQG_MAX_N = 262144;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct BqgLayout {  // offsets (ints) inside one scene's workspace
    long long pcell, ccell, pstart, pcur, cstart, ccur, sorted, perm, stride;
};
__host__ __device__ inline BqgLayout bqg_layout(int n, int m) {
    BqgLayout l;
    long long o = BQG_HDR;
    auto take = [&o](long long count) { const long long at = o; o = (o + count + 3) & ~3LL; return at; };  // 16-byte aligned arrays
    l.pcell = take(n);
    l.ccell = take(m);
    l.pstart = take(BQG_NCMAX + 1);
    l.pcur = take(BQG_NCMAX);
    l.cstart = take(BQG_NCMAX + 1);
    l.ccur = take(BQG_NCMAX);
    l.sorted = take(n);
    l.perm = take(m);
    l.stride = (o + 3) & ~3LL;
    return l;
}

struct BqgGrid {
    float lo[3], inv;
    int dim[3], nc;
};

__device__ __forceinline__ int bqg_axis(float p, float lo, float inv, int lim) {
    const float u = (p - lo) * inv;
    return (u > 0.f) ? (int)fminf(u, (float)lim) : 0;  // NaN / below the box -> 0, above / +inf -> lim: monotone in p
}
__device__ __forceinline__ BqgGrid bqg_load(const int *hdr) {
    BqgGrid g;
    g.lo[0] = __int_as_float(hdr[0]); g.lo[1] = __int_as_float(hdr[1]); g.lo[2] = __int_as_float(hdr[2]);
    g.inv = __int_as_float(hdr[3]);
    g.dim[0] = hdr[4]; g.dim[1] = hdr[5]; g.dim[2] = hdr[6]; g.nc = hdr[7];
    return g;
}
__device__ __forceinline__ int bqg_cell(const BqgGrid &g, float x, float y, float z) {
    const int cx = bqg_axis(x, g.lo[0], g.inv, g.dim[0] - 1), cy = bqg_axis(y, g.lo[1], g.inv, g.dim[1] - 1),
              cz = bqg_axis(z, g.lo[2], g.inv, g.dim[2] - 1);
    return (cz * g.dim[1] + cy) * g.dim[0] + cx;
}

// Centroids are grouped along a Z-order curve over (coarsened) cell coordinates -- 64 neighbours on it have a compact box
// in all three axes, which a run in the points' x-fastest order does not (it wraps around at the end of every row).
__device__ __forceinline__ int bqg_centroid_key(const int *hdr, const BqgGrid &g, float x, float y, float z) {
    const int q[3] = {bqg_axis(x, g.lo[0], g.inv, g.dim[0] - 1) >> hdr[8], bqg_axis(y, g.lo[1], g.inv, g.dim[1] - 1) >> hdr[9],
                      bqg_axis(z, g.lo[2], g.inv, g.dim[2] - 1) >> hdr[10]};
    const int nbits = hdr[11];
    int key = 0;
    for (int i = 0; i < nbits; ++i) {
        const int e = hdr[16 + i], a = e >> 8;
        const int qa = a == 0 ? q[0] : (a == 1 ? q[1] : q[2]);
        key = (key << 1) | ((qa >> (e & 255)) & 1);
    }
    return key;
}

// ---- 1a. bounding box of the finite coordinates, grid dimensions, zeroed histograms -----------------------------------
__global__ __launch_bounds__(1024) void bqg_setup_kernel(int n, const float *__restrict__ xyz, float radius, int *__restrict__ work,
                                                         long long stride, BqgLayout lay) {
    __shared__ float red[6][16];
    __shared__ int s_nc, s_keys;
    const int scene = blockIdx.x;
    const float *p = xyz + (size_t)scene * n * 3;
    int *w = work + (size_t)scene * stride;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[(size_t)k * 3 + a];
            if (v > -3.0e38f && v < 3.0e38f) {  // finite (NaN fails both)
                lo[a] = fminf(lo[a], v);
                hi[a] = fmaxf(hi[a], v);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    if (ln == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a][wv] = lo[a]; red[3 + a][wv] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ext[3];
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], h = red[3 + a][0];
            for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { l = fminf(l, red[a][i]); h = fmaxf(h, red[3 + a][i]); }
            if (!(h >= l)) { l = 0.f; h = 0.f; }  // no finite coordinate on this axis
            lo[a] = l;
            ext[a] = h - l;
            if (!(ext[a] < 3.0e38f)) ext[a] = 0.f;
        }
        float cell = radius;
        if (!(cell > 0.f) || !(cell < 3.0e38f)) cell = 1.f;
        // at most BQG_NCMAX cells: grow the edge until the box fits
        int d[3];
        for (int it = 0; it < 200; ++it) {
            long long prod = 1;
            for (int a = 0; a < 3; ++a) {
                const float q = ext[a] / cell;
                d[a] = (q < 60000.f) ? (int)q + 1 : 60001;
                prod *= d[a];
            }
            if (prod <= BQG_NCMAX) break;
            cell *= 1.25f;
        }
        if ((long long)d[0] * d[1] * d[2] > BQG_NCMAX) { d[0] = d[1] = d[2] = 1; }
        w[0] = __float_as_int(lo[0]); w[1] = __float_as_int(lo[1]); w[2] = __float_as_int(lo[2]);
        w[3] = __float_as_int(1.f / cell);
        w[4] = d[0]; w[5] = d[1]; w[6] = d[2]; w[7] = d[0] * d[1] * d[2];
        s_nc = w[7];
        // centroid keys: at most 16 bits, handed out MSB first to the axis with the most bits left
        int rem[3], shift[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) { rem[a] = 0; while ((1 << rem[a]) < d[a]) ++rem[a]; }
        while (rem[0] + rem[1] + rem[2] > 16) {
            const int a = (rem[0] >= rem[1] && rem[0] >= rem[2]) ? 0 : (rem[1] >= rem[2] ? 1 : 2);
            --rem[a]; ++shift[a];
        }
        const int nbits = rem[0] + rem[1] + rem[2];
        w[8] = shift[0]; w[9] = shift[1]; w[10] = shift[2]; w[11] = nbits; w[12] = 1 << nbits;
        for (int i = 0; i < nbits; ++i) {
            const int a = (rem[0] >= rem[1] && rem[0] >= rem[2]) ? 0 : (rem[1] >= rem[2] ? 1 : 2);
            --rem[a];
            w[16 + i] = (a << 8) | rem[a];
        }
        s_keys = 1 << nbits;
    }
    __syncthreads();
    const int nc = s_nc, keys = s_keys;
    for (int i = threadIdx.x; i <= nc; i += blockDim.x) w[lay.pstart + i] = 0;
    for (int i = threadIdx.x; i <= keys; i += blockDim.x) w[lay.cstart + i] = 0;
}

// Lanes of a wave that fall into the same cell are combined before they touch memory (feature-space "positions" put
// thousands of points into one cell: one atomic per lane would serialise on a single address): `leader` = first lane
// of my cell, `rank` = my place among its lanes, `cnt` = how many there are.
struct BqgPeers { int leader, rank, cnt; };
__device__ __forceinline__ BqgPeers bqg_peers(int cell, bool valid) {
    const int lane = threadIdx.x & 63;
    BqgPeers p = {lane, 0, 1};
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        const int c0 = __builtin_amdgcn_readlane(cell, l);
        const unsigned long long mask = __ballot(valid && cell == c0);
        if (valid && cell == c0) {
            p.leader = l;
            p.rank = __popcll(mask & ((1ull << lane) - 1ull));
            p.cnt = __popcll(mask);
        }
        todo &= ~mask;
    }
    return p;
}

// ---- 1b. cell of every point / centroid, histogram ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void bqg_count_kernel(int n, int m, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                        int *__restrict__ work, long long stride, BqgLayout lay) {
    const int scene = blockIdx.y;
    int *w = work + (size_t)scene * stride;
    const BqgGrid g = bqg_load(w);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    {
        int c = 0;
        if (i < n) {
            const float *p = xyz + ((size_t)scene * n + i) * 3;
            c = bqg_cell(g, p[0], p[1], p[2]);
            w[lay.pcell + i] = c;
        }
        const BqgPeers pr = bqg_peers(c, i < n);
        if (i < n && pr.leader == lane) atomicAdd(&w[lay.pstart + c], pr.cnt);
    }
    {
        int c = 0;
        if (i < m) {
            const float *p = new_xyz + ((size_t)scene * m + i) * 3;
            c = bqg_centroid_key(w, g, p[0], p[1], p[2]);
            w[lay.ccell + i] = c;
        }
        const BqgPeers pr = bqg_peers(c, i < m);
        if (i < m && pr.leader == lane) atomicAdd(&w[lay.cstart + c], pr.cnt);
    }
}

// ---- 1c. exclusive scan of a histogram (in place, total at [count]) and a copy as scatter cursors -----------------------
// One workgroup per histogram; all its entries are fetched first (4 consecutive per thread and 4096-entry tile, so the
// global-load latency is paid once), then the tiles are scanned from registers.
constexpr int BQG_SCAN_TILES = BQG_NCMAX / 4096 + 1;
__global__ __launch_bounds__(1024) void bqg_scan_kernel(int *__restrict__ work, long long stride, BqgLayout lay) {
    __shared__ int wsum[2][16];
    const int scene = blockIdx.y;
    int *w = work + (size_t)scene * stride;
    const int nc = blockIdx.x == 0 ? w[7] : w[12];
    int *start = w + (blockIdx.x == 0 ? lay.pstart : lay.cstart);
    int *cur = w + (blockIdx.x == 0 ? lay.pcur : lay.ccur);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    i32x4 v[BQG_SCAN_TILES];
#pragma unroll
    for (int t = 0; t < BQG_SCAN_TILES; ++t) {
        const int i = t * 4096 + 4 * threadIdx.x;
        v[t] = (i32x4){0, 0, 0, 0};
        if (i + 3 < nc) v[t] = *reinterpret_cast<const i32x4 *>(start + i);
        else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u < nc) v[t][u] = start[i + u];
        }
    }
    int carry = 0;
#pragma unroll
    for (int t = 0; t < BQG_SCAN_TILES; ++t) {
        if (t * 4096 >= nc) break;
        const int i = t * 4096 + 4 * threadIdx.x;
        const int tot = v[t][0] + v[t][1] + v[t][2] + v[t][3];
        int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[t & 1][wv] = incl;
        __syncthreads();  // one barrier per tile: the two halves of wsum alternate
        int woff = 0, tile = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int sv = wsum[t & 1][k];
            woff += (k < wv) ? sv : 0;
            tile += sv;
        }
        int run = carry + woff + incl - tot;
        i32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[u] = run; run += v[t][u]; }
        if (i + 3 < nc) {
            *reinterpret_cast<i32x4 *>(start + i) = o;
            *reinterpret_cast<i32x4 *>(cur + i) = o;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u < nc) { start[i + u] = o[u]; cur[i + u] = o[u]; }
        }
        carry += tile;
    }
    if (threadIdx.x == 0) start[nc] = carry;
}

// ---- 1d. ids grouped by cell (arbitrary order inside a cell) ------------------------------------------------------------
__global__ __launch_bounds__(256) void bqg_scatter_kernel(int n, int m, int *__restrict__ work, long long stride, BqgLayout lay) {
    const int scene = blockIdx.y;
    int *w = work + (size_t)scene * stride;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    {
        const int c = (i < n) ? w[lay.pcell + i] : 0;
        const BqgPeers pr = bqg_peers(c, i < n);
        int pos = (i < n && pr.leader == lane) ? atomicAdd(&w[lay.pcur + c], pr.cnt) : 0;
        pos = __shfl(pos, pr.leader);
        if (i < n) w[lay.sorted + pos + pr.rank] = i;
    }
    {
        const int c = (i < m) ? w[lay.ccell + i] : 0;
        const BqgPeers pr = bqg_peers(c, i < m);
        int pos = (i < m && pr.leader == lane) ? atomicAdd(&w[lay.ccur + c], pr.cnt) : 0;
        pos = __shfl(pos, pr.leader);
        if (i < m) w[lay.perm + pos + pr.rank] = i;
    }
}

// ---- 2. query --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i(int v) { return -wave_max_i32(-v); }  // cell coordinates: no INT_MIN
__device__ __forceinline__ int wave_max_i(int v) { return wave_max_i32(v); }

constexpr int BQG_MAX_SEG = 4;  // 8 measured slower at N = 16384: the per-wave set-up is replicated

struct BqgQuery {
    int n, m, fill_empty, words;
    float radius;          // largest radius: the box of a group
    float r2[2], r2min;    // squared radii (r2min: dilated form, single radius only)
    int ns[2], vec4[2];
    const float *new_xyz, *xyz;
    int *idx[2];
    const int *work;
    long long stride;
    BqgLayout lay;
};
// LDS of a workgroup (ints): bitmap[words] | cnts[R][S][64] | final[64][max nsample + 1] | jrow[64] |
//       per wave: list (u16 x 2048 = 1024 ints) | stage[2][64] float4 (512 ints) | hits per radius (u16 x nsample x 64)
__host__ __device__ inline size_t bqg_shared_ints(int words, int nsmax, int S, int R) {
    return (size_t)words + (size_t)R * S * 64 + (size_t)64 * (nsmax + 1) + 64;
}
__host__ __device__ inline size_t bqg_wave_ints(int ns_total) { return (size_t)1024 + 512 + (size_t)ns_total * 32; }

// One workgroup = 64 centroids adjacent on the Z curve; its S waves fill the candidate bitmap together, then each scans
// one S-th of the index range (lane = centroid) and the segments are concatenated in order, as in ball_query_seg_kernel.
// R = 2: the two grouping radii of an SA layer from ONE walk (distance evaluated once, two ordered hit lists).
template <bool DILATED, int R>
__global__ __launch_bounds__(64 * BQG_MAX_SEG) void bqg_query_kernel(BqgQuery a) {
    extern __shared__ __attribute__((aligned(16))) int bqg_lds[];
    const int S = blockDim.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int scene = blockIdx.y, group = blockIdx.x;
    const int n = a.n, m = a.m, words = a.words;
    const int nsmax = (R == 2 && a.ns[1] > a.ns[0]) ? a.ns[1] : a.ns[0];
    const int nstot = a.ns[0] + (R == 2 ? a.ns[1] : 0);
    unsigned *bitmap = reinterpret_cast<unsigned *>(bqg_lds);
    int *cnts = bqg_lds + words;
    int *final_img = cnts + R * S * 64;
    int *jrow = final_img + 64 * (nsmax + 1);
    int *mine_base = jrow + 64 + (size_t)wv * bqg_wave_ints(nstot);
    unsigned short *list = reinterpret_cast<unsigned short *>(mine_base);
    f32x4 *stage = reinterpret_cast<f32x4 *>(mine_base + 1024);
    unsigned short *hits[2];
    hits[0] = reinterpret_cast<unsigned short *>(mine_base + 1024 + 512);
    hits[1] = hits[0] + a.ns[0] * 64;

    const int *w = a.work + (size_t)scene * a.stride;
    const BqgGrid g = bqg_load(w);
    const int *sorted = w + a.lay.sorted;
    const int *pstart = w + a.lay.pstart;
    const int *perm = w + a.lay.perm;
    const float *xyz = a.xyz + (size_t)scene * n * 3;

    const int slot = group * 64 + lane;
    const bool active = slot < m;
    const int j = perm[active ? slot : group * 64];
    const float *ctr = a.new_xyz + ((size_t)scene * m + j) * 3;
    const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
    if (wv == 0) jrow[lane] = active ? j : -1;

    // ---- the cells that can hold a hit of any of the 64 balls (padded beyond the rounding of c -+ r) ----
    int lo_c[3], hi_c[3];
    {
        const float c3[3] = {cx, cy, cz};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float pad = a.radius * 1.0001f + fabsf(c3[ax]) * 1.0e-6f;
            const int l = bqg_axis(c3[ax] - pad, g.lo[ax], g.inv, g.dim[ax] - 1);
            const int h = bqg_axis(c3[ax] + pad, g.lo[ax], g.inv, g.dim[ax] - 1);
            lo_c[ax] = wave_min_i(l < h ? l : h);
            hi_c[ax] = wave_max_i(l < h ? h : l);
        }
    }
    // ---- bitmap of the candidate ids: the (y, z) rows of the box are dealt out to the lanes, a lane fetches its row's
    //      [start, end) in the sorted id array, then the waves set the bits row by row ----
    for (int i = threadIdx.x; i < words; i += blockDim.x) bitmap[i] = 0u;
    __syncthreads();
    {
        const int ny = hi_c[1] - lo_c[1] + 1, nz = hi_c[2] - lo_c[2] + 1, rows = ny * nz;
        int turn = 0;
        for (int r0 = 0; r0 < rows; r0 += 64) {
            const int rr = r0 + lane;
            int s = 0, e = 0;
            if (rr < rows) {
                const int z = lo_c[2] + rr / ny, y = lo_c[1] + rr % ny;
                const int row = (z * g.dim[1] + y) * g.dim[0];
                s = pstart[row + lo_c[0]];
                e = pstart[row + hi_c[0] + 1];
            }
            // short rows are dealt out to the waves whole (all the loads of a row are issued before its bits are set, so
            // a wave pays one memory latency per row it owns); long rows are sliced across all the waves
            for (unsigned long long left = __ballot(e > s); left; left &= left - 1) {
                const int l = __builtin_ctzll(left);
                const int sl = __builtin_amdgcn_readlane(s, l), el = __builtin_amdgcn_readlane(e, l);
                if (el - sl <= 512) {
                    const bool mine = (turn++ % S) == wv;
                    if (!mine) continue;
                    int id[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = sl + u * 64 + lane;
                        id[u] = (t < el) ? sorted[t] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (id[u] >= 0) atomicOr(&bitmap[id[u] >> 5], 1u << (id[u] & 31));
                } else {
                    int t = sl + wv * 64 + lane;
                    for (; t + 3 * 64 * S < el; t += 4 * 64 * S) {  // four loads in flight
                        int id[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) id[u] = sorted[t + u * 64 * S];
#pragma unroll
                        for (int u = 0; u < 4; ++u) atomicOr(&bitmap[id[u] >> 5], 1u << (id[u] & 31));
                    }
                    for (; t < el; t += 64 * S) {
                        const int id = sorted[t];
                        atomicOr(&bitmap[id >> 5], 1u << (id & 31));
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- this wave's share of the index range, 2048 points (one bitmap word per lane) at a time, in index order ----
    const int wps = words / S;  // multiple of 64 (host)
    const int segbase = wv * wps * 32;
    int cnt[2];
    float thr_r[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        cnt[r] = (active && r < R) ? 0 : (r < R ? a.ns[r] : 0);
        thr_r[r] = (r < R && cnt[r] < a.ns[r]) ? a.r2[r] : -1.f;
    }
    float thr = R == 2 ? fmaxf(thr_r[0], thr_r[1]) : thr_r[0];  // full / inactive lanes can no longer be hit
    const f32x4 nanp = {__builtin_nanf(""), 0.f, 0.f, 0.f};
    for (int w0 = wv * wps; w0 < (wv + 1) * wps; w0 += 64) {
        unsigned word = bitmap[w0 + lane];
        const int pc = __popc(word);
        int incl = pc;  // inclusive prefix over the lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) continue;
        {
            int pos = incl - pc;
            while (word) {
                const int bit = __ffs(word) - 1;
                word &= word - 1;
                list[pos++] = (unsigned short)(lane * 32 + bit);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int idbase = w0 * 32;
        // candidates of this chunk, 64 at a time: stage (x, y, z, id) in LDS, then the ordered append loop
        auto fetch = [&](int s) -> f32x4 {
            f32x4 v = nanp;
            if (s + lane < total) {
                const int id = idbase + list[s + lane];
                const float *p = xyz + (size_t)id * 3;
                v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
                v[3] = __int_as_float(id - segbase);
            }
            return v;
        };
        f32x4 nxt = fetch(0);
        int buf = 0;
        for (int s = 0; s < total; s += 64) {
            stage[buf * 64 + lane] = nxt;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (s + 64 < total) nxt = fetch(s + 64);
            const int here = (total - s < 64) ? total - s : 64;
            const f32x4 *st = stage + buf * 64;
            for (int t0 = 0; t0 < here; t0 += 8) {
                f32x4 p[8];
                float d2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = st[t0 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) d2[u] = sqdist(cx, cy, cz, p[u][0], p[u][1], p[u][2]);
                float dmin = d2[0];
#pragma unroll
                for (int u = 1; u < 8; ++u) dmin = fminf(dmin, d2[u]);  // fminf drops the NaN padding
                const bool maybe = DILATED ? (dmin < thr || (dmin == 0.f && thr >= 0.f)) : (dmin < thr);
                if (__any(maybe)) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned short k = (unsigned short)__float_as_int(p[u][3]);
                        if (DILATED) {
                            if (d2[u] == 0.f && cnt[0] < a.ns[0]) { hits[0][cnt[0] * 64 + lane] = k; ++cnt[0]; }
                            if (d2[u] >= a.r2min && d2[u] < a.r2[0] && cnt[0] < a.ns[0]) { hits[0][cnt[0] * 64 + lane] = k; ++cnt[0]; }
                        } else {
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                if (d2[u] < a.r2[r] && cnt[r] < a.ns[r]) { hits[r][cnt[r] * 64 + lane] = k; ++cnt[r]; }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) thr_r[r] = (cnt[r] < a.ns[r]) ? a.r2[r] : -1.f;
                    thr = R == 2 ? fmaxf(thr_r[0], thr_r[1]) : thr_r[0];
                }
            }
            buf ^= 1;
            if (__all(thr < 0.f)) break;
        }
        if (__all(thr < 0.f)) break;
    }
    // ---- per radius: ordered concatenation of the segments, padding with the first hit, coalesced rows ----
#pragma unroll
    for (int r = 0; r < R; ++r) cnts[(r * S + wv) * 64 + lane] = active ? cnt[r] : 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int nsample = a.ns[r];
        int before = 0, total_hits = 0;
        for (int s = 0; s < S; ++s) {
            const int c = cnts[(r * S + s) * 64 + lane];
            before += (s < wv) ? c : 0;
            total_hits += c;
        }
        const int mine = active ? cnt[r] : 0;
        for (int i = 0; i < mine && before + i < nsample; ++i)
            final_img[lane * (nsmax + 1) + before + i] = segbase + hits[r][i * 64 + lane];
        __syncthreads();
        if (wv == 0) {
            const int kept = total_hits < nsample ? total_hits : nsample;
            const int pad = kept > 0 ? final_img[lane * (nsmax + 1)] : 0;
            for (int i = kept; i < nsample; ++i) final_img[lane * (nsmax + 1) + i] = pad;
            jrow[lane] = (active && (a.fill_empty || total_hits > 0)) ? j : -1;  // rows of empty balls: the caller's zeros
        }
        __syncthreads();
        int *idx = a.idx[r];
        if (a.vec4[r]) {
            const int q4 = nsample >> 2;
            for (int e = threadIdx.x; e < 64 * q4; e += blockDim.x) {
                const int c = e / q4, i = (e - c * q4) * 4;
                const int jc = jrow[c];
                if (jc < 0) continue;
                i32x4 v;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = final_img[c * (nsmax + 1) + i + u];
                *reinterpret_cast<i32x4 *>(idx + ((size_t)scene * m + jc) * nsample + i) = v;
            }
        } else {
            for (int e = threadIdx.x; e < 64 * nsample; e += blockDim.x) {
                const int c = e / nsample, i = e - c * nsample;
                const int jc = jrow[c];
                if (jc < 0) continue;
                idx[((size_t)scene * m + jc) * nsample + i] = final_img[c * (nsmax + 1) + i];
            }
        }
        if (R == 2) __syncthreads();  // final_img / jrow are reused by the second radius
    }
}

// set-up launches + query; shared by the one- and two-radius entry points
static int bqg_run(int R, bool dilated, int b, int n, int m, float ra, float rb, float rmin, int nsa, int nsb, int fill_empty,
                   const float *new_xyz, const float *xyz, int *idx_a, int *idx_b, int *work, hipStream_t st) {
    const BqgLayout lay = bqg_layout(n, m);
    // S waves per group of 64 centroids (a power of two, one 2048-point bitmap chunk per wave at least)
    const int chunks = divup(divup(n, 32), 64);
    int S = 1;
    static const int seg_cap = [] { const char *e = getenv("SPS_BQG_SEG"); const int v = e ? atoi(e) : 0; return v > 0 ? v : BQG_MAX_SEG; }();
    while (S < BQG_MAX_SEG && S < seg_cap && 2 * S <= chunks) S <<= 1;
    const int words = divup(chunks, S) * S * 64;
    const int nsmax = (R == 2 && nsb > nsa) ? nsb : nsa, nstot = nsa + (R == 2 ? nsb : 0);
    const size_t lds = 4 * (bqg_shared_ints(words, nsmax, S, R) + (size_t)S * bqg_wave_ints(nstot));
    if (lds > 150 * 1024) return -1;  // caller falls back to the scan kernels
    const void *fn = R == 2 ? (const void *)bqg_query_kernel<false, 2>
                            : (dilated ? (const void *)bqg_query_kernel<true, 1> : (const void *)bqg_query_kernel<false, 1>);
    static LdsLimitOnce raised[3];
    const int which = R == 2 ? 2 : (dilated ? 1 : 0);
    if (lds > 64 * 1024) {
        const int rc = raise_lds_limit(fn, 150 * 1024, raised[which], "ball_query_grid");
        if (rc != SPS_OK) return rc;
    }
    const int big = n > m ? n : m;
    const float rbox = (R == 2 && rb > ra) ? rb : ra;
    hipLaunchKernelGGL(bqg_setup_kernel, dim3(b), dim3(1024), 0, st, n, xyz, rbox, work, lay.stride, lay);
    hipLaunchKernelGGL(bqg_count_kernel, dim3(divup(big, 256), b), dim3(256), 0, st, n, m, xyz, new_xyz, work, lay.stride, lay);
    hipLaunchKernelGGL(bqg_scan_kernel, dim3(2, b), dim3(1024), 0, st, work, lay.stride, lay);
    hipLaunchKernelGGL(bqg_scatter_kernel, dim3(divup(big, 256), b), dim3(256), 0, st, n, m, work, lay.stride, lay);
    BqgQuery q;
    q.n = n; q.m = m; q.fill_empty = fill_empty; q.words = words; q.radius = rbox;
    q.r2[0] = ra * ra; q.r2[1] = rb * rb; q.r2min = rmin * rmin;  // fp32 products, as the reference forms them (ball_query_gpu.cu:23, 84-85)
    q.ns[0] = nsa; q.ns[1] = nsb;
    q.vec4[0] = (nsa % 4 == 0 && (reinterpret_cast<uintptr_t>(idx_a) & 15) == 0) ? 1 : 0;
    q.vec4[1] = (R == 2 && nsb % 4 == 0 && (reinterpret_cast<uintptr_t>(idx_b) & 15) == 0) ? 1 : 0;
    q.new_xyz = new_xyz; q.xyz = xyz; q.idx[0] = idx_a; q.idx[1] = idx_b; q.work = work; q.stride = lay.stride; q.lay = lay;
    const dim3 grid(divup(m, 64), b), block(64 * S);
    if (R == 2) hipLaunchKernelGGL((bqg_query_kernel<false, 2>), grid, block, lds, st, q);
    else if (dilated) hipLaunchKernelGGL((bqg_query_kernel<true, 1>), grid, block, lds, st, q);
    else hipLaunchKernelGGL((bqg_query_kernel<false, 1>), grid, block, lds, st, q);
    return check_launch("bqg_query_kernel");
}

}  // namespace sps

extern "C" long long sps_ball_query_grid_workspace_ints(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0 || n > sps::BQG_MAX_N) return 0;
    return (long long)b * sps::bqg_layout(n, m).stride;
}

extern "C" int sps_ball_query_grid(int b, int n, int m, float max_radius, float min_radius, int dilated, int nsample,
                                   int fill_empty, const float *new_xyz, const float *xyz, int *idx, int *work,
                                   sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n < 0 || m < 0 || nsample < 0)
        return fail(SPS_ERR_INVALID, "ball_query_grid: bad shape b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (b == 0 || m == 0 || nsample == 0) return SPS_OK;
    hipStream_t st = as_stream(stream);
    const bool usable = work && n > 0 && n <= BQG_MAX_N && max_radius > 0.f && max_radius < 3.0e38f && b <= 65535;
    if (usable) {
        if (!new_xyz || !xyz || !idx) return fail(SPS_ERR_INVALID, "ball_query_grid: null pointer");
        const int rc = bqg_run(1, dilated != 0, b, n, m, max_radius, 0.f, min_radius, nsample, 0, fill_empty, new_xyz, xyz, idx,
                               nullptr, work, st);
        if (rc >= 0) return rc;
    }
    return launch_ball_query(dilated != 0, fill_empty != 0, b, n, m, max_radius, min_radius, nsample, new_xyz, xyz, idx, st);
}

extern "C" int sps_ball_query_grid2(int b, int n, int m, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                    const float *new_xyz, const float *xyz, int *idx_a, int *idx_b, int *work,
                                    sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0 || nsample_a <= 0 || nsample_b <= 0)
        return fail(SPS_ERR_INVALID, "ball_query_grid2: bad shape b=%d n=%d m=%d ns=(%d,%d)", b, n, m, nsample_a, nsample_b);
    if (b == 0 || m == 0) return SPS_OK;
    if (!new_xyz || !xyz || !idx_a || !idx_b) return fail(SPS_ERR_INVALID, "ball_query_grid2: null pointer");
    const bool usable = work && n <= BQG_MAX_N && radius_a > 0.f && radius_a < 3.0e38f && radius_b > 0.f && radius_b < 3.0e38f && b <= 65535;
    if (usable) {
        const int rc = bqg_run(2, false, b, n, m, radius_a, radius_b, 0.f, nsample_a, nsample_b, 1, new_xyz, xyz, idx_a, idx_b,
                               work, as_stream(stream));
        if (rc >= 0) return rc;
    }
    return sps_ball_query_full2(b, n, m, radius_a, nsample_a, radius_b, nsample_b, new_xyz, xyz, idx_a, idx_b, nullptr, stream);
}
