// pack_columns.hip -- drop the padded columns of a ball query before the grouped MLP.
//
// A ball-query row holds the first `nsample` hits in index order and REPEATS the first hit in the slots it could not fill
// (ball_query_gpu.cu:35-42); the reference then runs the shared MLP over all nsample columns and max-pools
// (pointnet2_modules.py:429-447).  A repeated column reproduces the first column's activations bit for bit, and max is
// idempotent, so only the distinct columns of a ball have to be computed.  On KITTI-shaped scenes a ball of the 16-sample
// scales holds 3-6 distinct neighbours and 20-40 % of the 32-sample balls hold <= 16.
//
// This kernel turns idx (B, M, nsample) into a stream of 16-column MFMA tiles:
//   * a centroid with cnt distinct columns gets a SLOT of 2^ceil(log2 cnt) columns, filled with its cnt columns and then
//     with repeats of its first column (exactly what the reference's padding does, only less of it);
//   * inside a window of 64 consecutive centroids the slots are laid out by decreasing size, so every slot is aligned to
//     its own size: the max over a slot is then a PREFIX of the xor butterfly the kernels already use for pooling
//     (steps 1, 2, 4, 8 with the lanes of smaller slots masked off) -- no segmented scan, no atomics;
//   * slots of 32 (64) columns are whole pairs (quads) of tiles; a window's tail is padded to a multiple of 4 tiles with
//     unused lanes (their columns repeat a valid point and are never written).
// Every column carries a meta word: centroid j within its scene [19:0], scene [27:20], log2(slot) [30:28], unused [31].
// Windows reserve their tiles with one atomic add, so the tile order across windows is arbitrary; results do not depend
// on it (a column's activations do not depend on its neighbours in the tile).
#include "sps_common.h"

namespace sps {

constexpr int PACK_WINDOW = 64;

// (the second argument set: a layer's other grouping scale, packed by the blocks with blockIdx.y == 1 of the same launch --
//  sps_pack_columns2; a launch costs ~6-9 us on the critical path behind the ball query, whatever it packs)
__global__ __launch_bounds__(256) void pack_columns_kernel(int m, int ns, int j0, int jcount, const int *__restrict__ idx,
                                                           int *__restrict__ cols, unsigned *__restrict__ meta,
                                                           int *__restrict__ ntiles, int tile_cap, int ns2 = 0,
                                                           const int *__restrict__ idx2 = nullptr, int *__restrict__ cols2 = nullptr,
                                                           unsigned *__restrict__ meta2 = nullptr, int *__restrict__ ntiles2 = nullptr,
                                                           int tile_cap2 = 0, int staged = 0, const int *__restrict__ prev = nullptr,
                                                           const int *__restrict__ prev2 = nullptr, int *__restrict__ taken = nullptr,
                                                           int *__restrict__ taken2 = nullptr, int k_late = 0,
                                                           const int *__restrict__ full_if = nullptr,
                                                           const int *__restrict__ full_if_any = nullptr, int any_count = 0) {
    // STAGED mode (sps_pack_columns2_late; staged = 1, or 2 for the last stage): `idx` holds the rows of a query over ONE point
    // range of every scene (-1 rows: no hit there); prev[scene][j] (NULL: zeros) = how many columns of centroid j the stages
    // over the lower ranges have put through the MLP already.  The complete row of the reference is the earlier hits followed by these, cut at nsample: a
    // centroid contributes its first min(hits here, nsample - prev) columns, possibly none (then it gets no slot at all);
    // taken[scene][j] (may be NULL) receives prev + that number for the next stage.  *full_if != 0 or any full_if_any != 0:
    // `idx` holds COMPLETE rows after all (the query was widened: a repair) and is packed as in the plain mode.
    extern __shared__ int s_idx[];   // [PACK_WINDOW][ns]
    if (blockIdx.y == 1) { ns = ns2; idx = idx2; cols = cols2; meta = meta2; ntiles = ntiles2; tile_cap = tile_cap2; prev = prev2; taken = taken2; }
    const bool late = staged && !flag_or_any(full_if, full_if_any, any_count);
    __shared__ int s_cnt[PACK_WINDOW], s_lg[PACK_WINDOW], s_pos[PACK_WINDOW];
    __shared__ int s_total, s_base;
    const int windows = (jcount + PACK_WINDOW - 1) / PACK_WINDOW;
    const int scene = blockIdx.x / windows, w = blockIdx.x - scene * windows;
    const int jw = j0 + w * PACK_WINDOW;                               // first centroid of the window
    const int nw = min(PACK_WINDOW, j0 + jcount - jw);                 // centroids in it
    const int t = threadIdx.x;
    const int *rows = idx + ((size_t)scene * m + jw) * ns;
    for (int e = t; e < nw * ns; e += blockDim.x) s_idx[e] = rows[e];
    __syncthreads();
    if (t < 64) {                                                      // wave 0: one lane per centroid of the window
        int cnt = 0, lg = -1;
        if (t < nw) {
            const int *r = s_idx + t * ns;
            const int first = r[0];
            cnt = 1;
            for (int s = ns - 1; s >= 1; --s)
                if (r[s] != first) { cnt = s + 1; break; }              // real entries form a prefix; the rest repeats r[0]
            if (late) {
                if (first < 0) cnt = 0;                                  // a stage's row without a hit is all -1
                const int before = prev ? prev[(size_t)scene * m + jw + t] : 0;
                cnt = cnt < ns - before ? cnt : ns - before;
                cnt = cnt < 0 ? 0 : cnt;
                // the reference's empty ball groups point 0 (its zeroed idx row): a centroid that no stage found a neighbour
                // for gets that one column in the LAST stage (k_late < 0 marks it; the column index is clamped below)
                if (staged == 2 && before + cnt == 0) cnt = 1;
                if (taken) taken[(size_t)scene * m + jw + t] = before + cnt;
            }
            lg = cnt <= 0 ? -1 : (cnt == 1 ? 0 : 32 - __builtin_clz(cnt - 1));
        }
        // slots by decreasing size: rank inside the class from a ballot, class bases from the class sizes
        int pos = 0, running = 0;
#pragma unroll
        for (int cls = 6; cls >= 0; --cls) {
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(lg == cls);
            if (lg == cls) pos = running + (__builtin_popcountll(mask & ((1ull << t) - 1ull)) << cls);
            running += __builtin_popcountll(mask) << cls;
        }
        s_cnt[t] = cnt; s_lg[t] = lg; s_pos[t] = pos;
        if (t == 0) {
            const int tiles = ((running + 15) / 16 + 3) & ~3;
            s_total = running;
            s_base = atomicAdd(ntiles, tiles);
        }
    }
    __syncthreads();
    const int base_tile = s_base;
    const int total = s_total;
    const int tiles = ((total + 15) / 16 + 3) & ~3;
    if (base_tile + tiles > tile_cap) return;                          // cannot happen with the documented capacity
    int *oc = cols + (size_t)base_tile * 16;
    unsigned *om = meta + (size_t)base_tile * 16;
    {
        const int cen = t & 63, part = t >> 6;                          // four threads share a slot
        if (cen < nw && s_lg[cen] >= 0) {
            const int cnt = s_cnt[cen], lg = s_lg[cen], pos = s_pos[cen];
            const unsigned word = (unsigned)(jw + cen) | ((unsigned)scene << 20) | ((unsigned)lg << 28);
            const int *r = s_idx + cen * ns;
            for (int s = part; s < (1 << lg); s += 4) {
                const int v = r[s < cnt ? s : 0];
                oc[pos + s] = v < 0 ? 0 : v;       // (-1: the empty-ball column of a last stage = point 0)
                om[pos + s] = word;
            }
        }
    }
    const unsigned idle = (unsigned)jw | ((unsigned)scene << 20) | 0x80000000u;
    for (int e = total + t; e < tiles * 16; e += blockDim.x) {         // the window's tail: valid addresses, never written
        oc[e] = s_idx[0] < 0 ? 0 : s_idx[0];
        om[e] = idle;
    }
}

}  // namespace sps

// Tiles needed in the worst case (no padded column anywhere) for b scenes x jcount centroids x nsample columns.
extern "C" long long sps_pack_columns_capacity(int b, int jcount, int nsample) {
    if (b <= 0 || jcount <= 0 || nsample <= 0) return 0;
    const long long windows = (jcount + sps::PACK_WINDOW - 1) / sps::PACK_WINDOW;
    int slot = 1;
    while (slot < nsample) slot <<= 1;
    const long long per_window = ((long long)sps::PACK_WINDOW * slot / 16 + 3) & ~3LL;
    return (long long)b * windows * per_window;
}

// idx (b, m, nsample) rows of centroids [j0, j0 + jcount) of every scene -> cols / meta (tile_cap x 16 each) and the number
// of tiles written (*ntiles, a multiple of 4; the caller zeroes it).  nsample <= 64, m < 2^20, b <= 256.
extern "C" int sps_pack_columns(int b, int m, int j0, int jcount, int nsample, const int *idx, int *cols, unsigned *meta,
                                int *ntiles, long long tile_cap, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || m <= 0 || j0 < 0 || jcount < 0 || j0 + jcount > m || nsample <= 0 || nsample > 64 || m >= (1 << 20) || b > 256)
        return fail(SPS_ERR_INVALID, "pack_columns: bad shape (b=%d m=%d range [%d, +%d) nsample=%d)", b, m, j0, jcount, nsample);
    if (b == 0 || jcount == 0) return SPS_OK;
    if (!idx || !cols || !meta || !ntiles) return fail(SPS_ERR_INVALID, "pack_columns: null pointer");
    if (tile_cap < sps_pack_columns_capacity(b, jcount, nsample) || tile_cap > 0x7FFFFFF)
        return fail(SPS_ERR_INVALID, "pack_columns: tile capacity %lld, need %lld", tile_cap,
                    sps_pack_columns_capacity(b, jcount, nsample));
    const int windows = (jcount + PACK_WINDOW - 1) / PACK_WINDOW;
    hipLaunchKernelGGL(pack_columns_kernel, dim3(b * windows), dim3(256), (size_t)PACK_WINDOW * nsample * sizeof(int),
                       as_stream(stream), m, nsample, j0, jcount, idx, cols, meta, ntiles, (int)tile_cap);
    return check_launch("pack_columns_kernel");
}

// Both grouping scales of a layer in ONE launch (same b, m and centroid range; each with its own nsample, rows and outputs).
extern "C" int sps_pack_columns2(int b, int m, int j0, int jcount, int nsample_a, const int *idx_a, int *cols_a, unsigned *meta_a,
                                 int *ntiles_a, long long tile_cap_a, int nsample_b, const int *idx_b, int *cols_b,
                                 unsigned *meta_b, int *ntiles_b, long long tile_cap_b, sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || m <= 0 || j0 < 0 || jcount < 0 || j0 + jcount > m || nsample_a <= 0 || nsample_a > 64 || nsample_b <= 0 ||
        nsample_b > 64 || m >= (1 << 20) || b > 256)
        return fail(SPS_ERR_INVALID, "pack_columns2: bad shape (b=%d m=%d range [%d, +%d) nsample=%d/%d)", b, m, j0, jcount,
                    nsample_a, nsample_b);
    if (b == 0 || jcount == 0) return SPS_OK;
    if (!idx_a || !cols_a || !meta_a || !ntiles_a || !idx_b || !cols_b || !meta_b || !ntiles_b)
        return fail(SPS_ERR_INVALID, "pack_columns2: null pointer");
    if (tile_cap_a < sps_pack_columns_capacity(b, jcount, nsample_a) || tile_cap_a > 0x7FFFFFF ||
        tile_cap_b < sps_pack_columns_capacity(b, jcount, nsample_b) || tile_cap_b > 0x7FFFFFF)
        return fail(SPS_ERR_INVALID, "pack_columns2: tile capacity too small");
    const int windows = (jcount + PACK_WINDOW - 1) / PACK_WINDOW;
    const int nsmax = nsample_a > nsample_b ? nsample_a : nsample_b;
    hipLaunchKernelGGL(pack_columns_kernel, dim3(b * windows, 2), dim3(256), (size_t)PACK_WINDOW * nsmax * sizeof(int),
                       as_stream(stream), m, nsample_a, j0, jcount, idx_a, cols_a, meta_a, ntiles_a, (int)tile_cap_a, nsample_b, idx_b,
                       cols_b, meta_b, ntiles_b, (int)tile_cap_b);
    return check_launch("pack_columns_kernel<2>");
}

// STAGED packing of both scales (see pack_columns_kernel): idx_* = the rows of sps_ball_query_full2_points over one point range
// (-1 rows = no hit in the range); prev_* (b, m) ints or NULL = columns of every centroid that the stages over the lower ranges
// took; taken_* (b, m) or NULL receives the count including this stage.  Only the columns the complete row would hold are
// packed; last_stage != 0: a centroid that no stage found a neighbour for gets the reference's empty-ball column (point 0).  *full_if != 0 or any of
// full_if_any[0 .. any_count) != 0 (device ints, may be NULL): idx_* hold complete rows and are packed whole.  All m centroids.
extern "C" int sps_pack_columns2_late(int b, int m, int last_stage, int nsample_a, const int *prev_a, const int *idx_a, int *taken_a,
                                      int *cols_a, unsigned *meta_a, int *ntiles_a, long long tile_cap_a, int nsample_b,
                                      const int *prev_b, const int *idx_b, int *taken_b, int *cols_b, unsigned *meta_b,
                                      int *ntiles_b, long long tile_cap_b, const int *full_if, const int *full_if_any,
                                      int any_count, sps_stream_t stream) {
    const int k_late = 0;
    using namespace sps;
    if (b < 0 || m <= 0 || k_late < 0 || nsample_a <= 0 || nsample_a > 64 || nsample_b <= 0 || nsample_b > 64 || m >= (1 << 20) || b > 256)
        return fail(SPS_ERR_INVALID, "pack_columns2_late: bad shape (b=%d m=%d k_late=%d nsample=%d/%d)", b, m, k_late, nsample_a, nsample_b);
    if (b == 0) return SPS_OK;
    if (!idx_a || !cols_a || !meta_a || !ntiles_a || !idx_b || !cols_b || !meta_b || !ntiles_b)
        return fail(SPS_ERR_INVALID, "pack_columns2_late: null pointer");
    if (tile_cap_a < sps_pack_columns_capacity(b, m, nsample_a) || tile_cap_a > 0x7FFFFFF ||
        tile_cap_b < sps_pack_columns_capacity(b, m, nsample_b) || tile_cap_b > 0x7FFFFFF)
        return fail(SPS_ERR_INVALID, "pack_columns2_late: tile capacity too small");
    const int windows = (m + PACK_WINDOW - 1) / PACK_WINDOW;
    const int nsmax = nsample_a > nsample_b ? nsample_a : nsample_b;
    hipLaunchKernelGGL(pack_columns_kernel, dim3(b * windows, 2), dim3(256), (size_t)PACK_WINDOW * nsmax * sizeof(int),
                       as_stream(stream), m, nsample_a, 0, m, idx_a, cols_a, meta_a, ntiles_a, (int)tile_cap_a, nsample_b,
                       idx_b, cols_b, meta_b, ntiles_b, (int)tile_cap_b, last_stage ? 2 : 1, prev_a, prev_b, taken_a, taken_b, k_late,
                       full_if, full_if_any, any_count);
    return check_launch("pack_columns_kernel<staged>");
}
