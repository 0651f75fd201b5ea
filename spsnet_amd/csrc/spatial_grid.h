// spatial_grid.h -- 12-bit Morton-like cell keys used to make groups of points spatially compact
// (fps_pruned.hip: buckets of points; ball_query.hip: groups of 64 centroids).  Device-only helpers.
#pragma once
#include "sps_common.h"

namespace sps {

constexpr int PF_KEY_BITS = 12;
constexpr int PF_BINS = 1 << PF_KEY_BITS;

struct PfGrid {       // wave-uniform description of the cell grid used for the spatial sort
    float lo[3], scale[3];
    int lim[3];              // cells per axis - 1
    int axis[PF_KEY_BITS];   // key bit i (MSB first) is bit shift[i] of the cell coordinate on axis[i]
    int shift[PF_KEY_BITS];
};

__device__ __forceinline__ int pf_cell_key(const PfGrid &g, float x, float y, float z) {
    const float p[3] = {x, y, z};
    int qv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = (p[a] - g.lo[a]) * g.scale[a];
        const int c = (u > 0.f) ? (int)fminf(u, (float)g.lim[a]) : 0;  // NaN / negative -> cell 0
        qv[a] = c > g.lim[a] ? g.lim[a] : c;
    }
    int key = 0;
#pragma unroll
    for (int i = 0; i < PF_KEY_BITS; ++i) {
        const int qa = g.axis[i] == 0 ? qv[0] : (g.axis[i] == 1 ? qv[1] : qv[2]);
        key = (key << 1) | ((qa >> g.shift[i]) & 1);
    }
    return key;
}

// Build the grid for the bounding box [lo, hi] (wave-uniform inputs): the 12 key bits are handed out one at a
// time to the axis whose cells are currently the longest, so cells end up roughly cubic whatever the extents.
__device__ __forceinline__ PfGrid pf_make_grid(const float lo[3], const float hi[3]) {
    PfGrid grid;
    float ext[3];
    int nb0 = 0, nb1 = 0, nb2 = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float e = hi[a] - lo[a];
        if (!(e > 0.f) || !(e < 3.0e38f)) e = 0.f;  // degenerate / infinite extent: one cell on this axis
        grid.lo[a] = (lo[a] > -3.0e38f && lo[a] < 3.0e38f) ? lo[a] : 0.f;
        ext[a] = e;
    }
    float c0 = ext[0], c1 = ext[1], c2 = ext[2];
#pragma unroll
    for (int i = 0; i < PF_KEY_BITS; ++i) {
        int a = 0;
        float cm = c0;
        if (c1 > cm) { a = 1; cm = c1; }
        if (c2 > cm) { a = 2; }
        grid.axis[i] = a;
        if (a == 0) { nb0 += 1; c0 *= 0.5f; } else if (a == 1) { nb1 += 1; c1 *= 0.5f; } else { nb2 += 1; c2 *= 0.5f; }
    }
    int u0 = 0, u1 = 0, u2 = 0;
#pragma unroll
    for (int i = 0; i < PF_KEY_BITS; ++i) {  // the j-th bit given to an axis is its j-th most significant cell bit
        const int a = grid.axis[i];
        if (a == 0) { u0 += 1; grid.shift[i] = nb0 - u0; } else if (a == 1) { u1 += 1; grid.shift[i] = nb1 - u1; } else { u2 += 1; grid.shift[i] = nb2 - u2; }
    }
    const int nbs[3] = {nb0, nb1, nb2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        grid.lim[a] = (1 << nbs[a]) - 1;
        grid.scale[a] = (ext[a] > 0.f) ? (float)(1 << nbs[a]) / ext[a] : 0.f;
    }
    return grid;
}

}  // namespace sps
