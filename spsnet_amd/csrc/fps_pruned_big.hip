// fps_pruned_big.hip -- exact farthest point sampling with spatial pruning for scenes that do not fit the registers
// of one CU (16 384 < N <= 262 144: Waymo-shaped clouds, BASELINE config 5).
//
// Same contract, same results (indices and final `temp`, bit for bit) and the same algorithm as fps_pruned.hip:
// 64-point buckets made spatially compact by a 12-bit cell-key counting sort, one metadata lane per bucket (box,
// largest running distance + that point's rank and coordinates, second largest distance), the monotone box test
// that proves most buckets unchanged, and several exact picks per barrier (two records per wave, 16 x 16 ordered
// pairs; see the comments there for the proof).  What differs is where things live:
//   * the sorted points {x, y, z, running distance, rank} sit in a caller-provided workspace (20 B per point, 3.6 MB
//     for 180 000 points -- L2-resident: one workgroup per scene, and a workgroup lives on one XCD);
//   * a wave owns up to 64 * ROWS buckets, their metadata in ROWS registers per lane (ROWS = 2, 4, 8);
//   * a bucket the box test cannot rule out is fetched (4 coalesced 256-byte loads), re-evaluated with exactly the
//     reference arithmetic and, if any lane changed, written back and its metadata refreshed;
//   * the runner-up inside a bucket, needed for the bound of a published record, is kept in the metadata (computed
//     while the bucket is in registers anyway) instead of being re-derived from the points.
// The brute-force streaming kernel (fps.hip) needs 337 ms for 8 x 65 536 -> 16 384 and 881 ms for 8 x 180 000 ->
// 16 384 on MI355X; this kernel's rounds cost a few microseconds for ~6 picks.
#include "fps_pruned_util.h"

#include <cstdio>
#include <cstdlib>

#include <type_traits>
#include <utility>

namespace sps {

namespace {

constexpr int IMIN = (int)0x80000000;

struct PbShared {
    int hist[PF_BINS];
    __attribute__((aligned(16))) int soa[2][6][2 * PF_WAVES];  // round parity x field x record (see fps_pruned.hip)
    float red[6][PF_WAVES];
    int wsum[PF_WAVES];
};


template <int R, int ROWS, class F>
__device__ __forceinline__ void rows_each(F &fn) {
    if constexpr (R < ROWS) {
        fn(std::integral_constant<int, R>{});
        rows_each<R + 1, ROWS>(fn);
    }
}

}  // namespace

// work: per scene `stride` floats, of which 5 arrays of npad elements: x, y, z, t (float), rank (int); npad = 64 * number of buckets
template <int ROWS>
__global__ __launch_bounds__(PF_THREADS) void fps_pruned_big_kernel(int n, int m, int bs, int l2, int rb, int npad,
                                                                    long long stride, const float *__restrict__ dataset,
                                                                    float *__restrict__ temp, int *__restrict__ idxs,
                                                                    float *__restrict__ work, int only_given_up = 0,
                                                                    int *__restrict__ progress = nullptr) {
    // only_given_up: the follow-up launch of the clustered kernel (fps_pruned_cluster.hip) -- it takes the scenes whose
    // give-up word (granule 0 of the exchange area behind the scene's 5 npad workspace floats) was raised by a bounded poll
    // that ran out; temp may then be NULL (all running distances start at 1e10, nothing handed back) and progress[scene],
    // when given, is set to m once every pick is visible: the consumers of a publishing launch wait for exactly that.
    if (m <= 0) return;
    __shared__ PbShared sh;
    const int scene = blockIdx.x;
    if (only_given_up) {
        const unsigned long long *xg = reinterpret_cast<const unsigned long long *>(work + (size_t)scene * stride + (size_t)5 * npad);
        if (__hip_atomic_load(xg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) return;
    }
    const float *xyz = dataset + (size_t)scene * n * 3;
    if (temp) temp += (size_t)scene * n;
    idxs += (size_t)scene * m;
    float *sx = work + (size_t)scene * stride, *sy = sx + npad, *sz = sy + npad, *st = sz + npad;
    int *srk = reinterpret_cast<int *>(st + npad);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = npad / 64;  // buckets

    // ------------------------------------------------------------------ spatial sort (once)
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k0 = tid; k0 < n; k0 += 8 * PF_THREADS) {
        float v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * PF_THREADS;
            const int kk = k < n ? k : k0;
#pragma unroll
            for (int a = 0; a < 3; ++a) v[u][a] = xyz[(size_t)kk * 3 + a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo3[a] = fminf(lo3[a], v[u][a]); hi3[a] = fmaxf(hi3[a], v[u][a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo3[a] = wave_allmin_f32(lo3[a]);
        hi3[a] = wave_allmax_f32(hi3[a]);
        if (lane == 0) { sh.red[a][wave] = lo3[a]; sh.red[3 + a][wave] = hi3[a]; }
    }
    for (int i = tid; i < PF_BINS; i += PF_THREADS) sh.hist[i] = 0;
    __syncthreads();
    float glo[3], ghi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = sh.red[a][0], h = sh.red[3 + a][0];
#pragma unroll
        for (int w = 1; w < PF_WAVES; ++w) { l = fminf(l, sh.red[a][w]); h = fmaxf(h, sh.red[3 + a][w]); }
        glo[a] = l; ghi[a] = h;
    }
    const PfGrid grid = pf_make_grid(glo, ghi);
    for (int k0 = tid; k0 < n; k0 += 8 * PF_THREADS) {
        float v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * PF_THREADS;
            const int kk = k < n ? k : k0;
#pragma unroll
            for (int a = 0; a < 3; ++a) v[u][a] = xyz[(size_t)kk * 3 + a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u * PF_THREADS < n) atomicAdd(&sh.hist[pf_cell_key(grid, v[u][0], v[u][1], v[u][2])], 1);
    }
    __syncthreads();
    {   // exclusive prefix sum of the histogram: 8 bins per thread, wave scan, cross-wave offsets
        constexpr int PER = PF_BINS / PF_THREADS;
        int loc[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { loc[i] = sh.hist[tid * PER + i]; sum += loc[i]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) sh.wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += sh.wsum[w];
        int run = base + incl - sum;
#pragma unroll
        for (int i = 0; i < PER; ++i) { sh.hist[tid * PER + i] = run; run += loc[i]; }
    }
    __syncthreads();
    for (int k0 = tid; k0 < n; k0 += 8 * PF_THREADS) {
        float v[8][3], tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * PF_THREADS;
            const int kk = k < n ? k : k0;
#pragma unroll
            for (int a = 0; a < 3; ++a) v[u][a] = xyz[(size_t)kk * 3 + a];
            tv[u] = temp ? temp[kk] : 1e10f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * PF_THREADS;
            if (k < n) {
                const int pos = atomicAdd(&sh.hist[pf_cell_key(grid, v[u][0], v[u][1], v[u][2])], 1);
                sx[pos] = v[u][0]; sy[pos] = v[u][1]; sz[pos] = v[u][2]; st[pos] = tv[u];
                srk[pos] = (int)pf_rank((unsigned)k, bs, l2, rb);
            }
        }
    }
    for (int p = n + tid; p < npad; p += PF_THREADS) {  // padding: never inside a box, distance stays -1, worst rank
        sx[p] = NAN; sy[p] = NAN; sz[p] = NAN; st[p] = -1.f; srk[p] = 0x0FFFFFFF;
    }
    __threadfence();
    __syncthreads();

    // ------------------------------------------------------------------ bucket metadata
    // bucket g = v * PF_WAVES + wave is slot v of this wave: row v / 64, lane v % 64
    typedef float vfR __attribute__((ext_vector_type(ROWS)));
    typedef int viR __attribute__((ext_vector_type(ROWS)));
    vfR blo_x, blo_y, blo_z, bhi_x, bhi_y, bhi_z, bpx, bpy, bpz;
    viR bmax, bsec, bkeylo;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {  // no bucket: a box the test always rules out
        blo_x[r] = blo_y[r] = blo_z[r] = bhi_x[r] = bhi_y[r] = bhi_z[r] = INFINITY;
        bpx[r] = bpy[r] = bpz[r] = 0.f;
        bmax[r] = __float_as_int(-1.f); bsec[r] = __float_as_int(-1.f); bkeylo[r] = 0;
    }
    // the bucket's points of the lanes, its maximum under the reference's tie rule and the runner-up distance
    int r_vmax = 0, r_sec = 0, r_keylo = 0, r_px = 0, r_py = 0, r_pz = 0;
    auto refresh = [&](float tv, int rv, float xv, float yv, float zv) {
        const int tb = __float_as_int(tv);
        const int vmax = wave_max_i32_id(tb);
        const unsigned long long eq = __ballot(tb == vmax);
        int wl = __builtin_ctzll(eq);
        if (__builtin_popcountll(eq) > 1) {  // equal distances: the reference's tie rule decides
            const int inv = (tb == vmax) ? (0x0FFFFFFF - rv) : -1;
            const int best = wave_max_i32_id(inv);
            wl = __builtin_ctzll(__ballot(inv == best));
        }
        r_sec = wave_max_i32_id(lane != wl ? tb : IMIN);
        const int rank = __builtin_amdgcn_readlane(rv, wl);
        r_px = __builtin_amdgcn_readlane(__float_as_int(xv), wl);
        r_py = __builtin_amdgcn_readlane(__float_as_int(yv), wl);
        r_pz = __builtin_amdgcn_readlane(__float_as_int(zv), wl);
        r_vmax = vmax;
        r_keylo = (int)((0x0FFFFFFFu - (unsigned)rank) << 4);
    };
    auto commit = [&](auto rc, int l) {  // metadata of (row R, lane l) <- the scalars refresh() left: six v_writelane, one M0 set-up
        constexpr int R = decltype(rc)::value;
        int a0 = bmax[R], a1 = bsec[R], a2 = bkeylo[R], a3 = __float_as_int(bpx[R]), a4 = __float_as_int(bpy[R]), a5 = __float_as_int(bpz[R]);
        unsigned keep;
        asm volatile("s_mov_b32 %6, m0\n\t"
                     "s_mov_b32 m0, %13\n\t"
                     "s_nop 3\n\t"
                     "v_writelane_b32 %0, %7, m0\n\t"
                     "v_writelane_b32 %1, %8, m0\n\t"
                     "v_writelane_b32 %2, %9, m0\n\t"
                     "v_writelane_b32 %3, %10, m0\n\t"
                     "v_writelane_b32 %4, %11, m0\n\t"
                     "v_writelane_b32 %5, %12, m0\n\t"
                     "s_mov_b32 m0, %6"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "=&s"(keep)
                     : "s"(r_vmax), "s"(r_sec), "s"(r_keylo), "s"(r_px), "s"(r_py), "s"(r_pz), "s"(__builtin_amdgcn_readfirstlane(l)));
        bmax[R] = a0; bsec[R] = a1; bkeylo[R] = a2; bpx[R] = __int_as_float(a3); bpy[R] = __int_as_float(a4); bpz[R] = __int_as_float(a5);
    };
    auto for_rows = [&](auto &&fn) { rows_each<0, ROWS>(fn); };  // fn(integral_constant row), unrolled
    for_rows([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        for (int l = 0; l < 64; ++l) {
            const int g = (R * 64 + l) * PF_WAVES + wave;
            if (g >= nb) break;
            const size_t p = (size_t)g * 64 + lane;
            const float xv = sx[p], yv = sy[p], zv = sz[p], tv = st[p];
            const int rv = srk[p];
            float lx = xv, ly = yv, lz = zv, hx = xv, hy = yv, hz = zv;
            wave_box6(lx, ly, lz, hx, hy, hz);
            if (lane == l) { blo_x[R] = lx; blo_y[R] = ly; blo_z[R] = lz; bhi_x[R] = hx; bhi_y[R] = hy; bhi_z[R] = hz; }
            refresh(tv, rv, xv, yv, zv);
            commit(rc, l);
        }
    });

    if (tid == 0) idxs[0] = 0;
    __syncthreads();

    // ------------------------------------------------------------------ sampling loop (see fps_pruned.hip)
    int crec = 0;
    bool cand_stale = true;
    int cand_e1 = -1, cand_e2 = -1;  // (row << 6 | lane) of the two published buckets
    float ax = xyz[0], ay = xyz[1], az = xyz[2];
    unsigned long long pend = m > 1 ? 1ull : 0ull;
    int j = 1, round = 0;
    const bool younger = wave >= PF_WAVES / 2;  // priority alternates by phase between the two waves of a SIMD (fps_pruned.hip)
    for (;;) {
        if (younger) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        while (pend) {
            const int rr = __builtin_ctzll(pend);
            pend &= pend - 1;
            const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), rr));
            const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), rr));
            const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), rr));
            for_rows([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                // 1. one lane per bucket: can the new centre lower any distance in the box?
                const float qx = __builtin_amdgcn_fmed3f(cx, blo_x[R], bhi_x[R]);
                const float qy = __builtin_amdgcn_fmed3f(cy, blo_y[R], bhi_y[R]);
                const float qz = __builtin_amdgcn_fmed3f(cz, blo_z[R], bhi_z[R]);
                const float lb = sqdist(qx, qy, qz, cx, cy, cz);
                const bool skip = lb >= __int_as_float(bmax[R]);  // NaN -> not skipped
                unsigned long long todo = __ballot(!skip);
                // 2. fetch and re-evaluate the surviving buckets (exactly the reference arithmetic, one point per lane)
                while (todo) {
                    const int l = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int g = (R * 64 + l) * PF_WAVES + wave;
                    const size_t p = (size_t)g * 64 + lane;
                    const float xv = sx[p], yv = sy[p], zv = sz[p];
                    const float tv = __hip_atomic_load(st + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float d = sqdist(xv, yv, zv, cx, cy, cz);
                    if (__ballot(d < tv) != 0ull) {  // some point moved closer to a sample
                        const float tn = fmin_raw(d, tv);
                        __hip_atomic_store(st + p, tn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int rv = srk[p];
                        refresh(tn, rv, xv, yv, zv);
                        commit(rc, l);
                        const int e = (R << 6) | l;
                        if (e == cand_e1 || e == cand_e2) cand_stale = true;
                    }
                }
            });
        }
        if (j >= m) break;
        // 3. the wave's two records: the maxima of its two best buckets and the bounds that take over once they are picked
        if (cand_stale) {
            // best entry over rows x lanes, entries in `excl1` / `excl2` left out: distance bits, lane, row
            auto best_entry = [&](int excl1, int excl2, int &vmax, int &row) -> int {
                int lv = IMIN, lk = -1, lr = 0;  // this lane's best row
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const int e = (r << 6) | lane;
                    const bool ok = e != excl1 && e != excl2;
                    const int v = bmax[r], k = (int)((unsigned)bkeylo[r] >> 4);
                    const bool better = ok && (v > lv || (v == lv && k > lk));
                    lv = better ? v : lv; lk = better ? k : lk; lr = better ? r : lr;
                }
                vmax = wave_max_i32_id(lv);
                const unsigned long long eq = __ballot(lv == vmax);
                int wl = __builtin_ctzll(eq);
                if (__builtin_popcountll(eq) > 1) {
                    const int kl = (lv == vmax) ? lk : -1;
                    const int kbest = wave_max_i32_id(kl);
                    wl = __builtin_ctzll(__ballot(kl == kbest));
                }
                row = __builtin_amdgcn_readlane(lr, wl);
                return wl;
            };
            int v1, v2, v3, r1, r2, r3;
            const int wl1 = best_entry(-1, -1, v1, r1);
            const int e1 = (r1 << 6) | wl1;
            const int wl2 = best_entry(e1, -1, v2, r2);
            const int e2 = (r2 << 6) | wl2;
            (void)best_entry(e1, e2, v3, r3);
            auto record = [&](int row, int wl, int vmax, int others, auto rcn) {
                constexpr int RN = decltype(rcn)::value;
                int klo = 0, px = 0, py = 0, pz = 0, sec = IMIN;
                for_rows([&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    if (row == R) {
                        klo = __builtin_amdgcn_readlane(bkeylo[R], wl);
                        px = __builtin_amdgcn_readlane(__float_as_int(bpx[R]), wl);
                        py = __builtin_amdgcn_readlane(__float_as_int(bpy[R]), wl);
                        pz = __builtin_amdgcn_readlane(__float_as_int(bpz[R]), wl);
                        sec = __builtin_amdgcn_readlane(bsec[R], wl);
                    }
                });
                const float fx = __int_as_float(px), fy = __int_as_float(py), fz = __int_as_float(pz);
                const float own = fmin_raw(sqdist(fx, fy, fz, fx, fy, fz), __int_as_float(vmax));
                const int bound = imax(imax(others, sec), __builtin_amdgcn_readfirstlane(__float_as_int(own)));
                put_lane<0 + RN>(crec, vmax);
                put_lane<2 + RN>(crec, klo);
                put_lane<4 + RN>(crec, px);
                put_lane<6 + RN>(crec, py);
                put_lane<8 + RN>(crec, pz);
                put_lane<10 + RN>(crec, bound);
            };
            record(r1, wl1, v1, IMIN, std::integral_constant<int, 0>{});
            record(r2, wl2, v2, v3, std::integral_constant<int, 1>{});
            cand_e1 = e1; cand_e2 = e2;
            cand_stale = false;
        }
        // 4. exchange + acceptance: identical to fps_pruned.hip
        const int buf = round & 1;
        if (lane < 12) sh.soa[buf][lane >> 1][2 * wave + (lane & 1)] = crec;
        __syncthreads();
        if (younger) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
        const int rj = lane >> 2, rb4 = lane & 3;
        const int4 id = *(const int4 *)&sh.soa[buf][0][rb4 * 4], ik = *(const int4 *)&sh.soa[buf][1][rb4 * 4];
        const int4 ixv = *(const int4 *)&sh.soa[buf][2][rb4 * 4], iyv = *(const int4 *)&sh.soa[buf][3][rb4 * 4];
        const int4 izv = *(const int4 *)&sh.soa[buf][4][rb4 * 4], ibv = *(const int4 *)&sh.soa[buf][5][rb4 * 4];
        const int jd = sh.soa[buf][0][rj], jk = sh.soa[buf][1][rj];
        const float jx = __int_as_float(sh.soa[buf][2][rj]), jy = __int_as_float(sh.soa[buf][3][rj]);
        const float jz = __int_as_float(sh.soa[buf][4][rj]);
        const float jt = __int_as_float(jd);
        int nbef = 0, nbad = 0;
        auto pair = [&](int idist, int iklo, int ixb, int iyb, int izb, int ibound) {
            const bool before = (idist > jd) | ((idist == jd) & ((unsigned)iklo > (unsigned)jk));
            const float dij = sqdist(jx, jy, jz, __int_as_float(ixb), __int_as_float(iyb), __int_as_float(izb));
            const bool lowered = !(dij >= jt);
            const bool hidden = !(jt > __int_as_float(ibound));
            nbef += before ? 1 : 0;
            nbad += (before & (lowered | hidden)) ? 1 : 0;
        };
        pair(id.x, ik.x, ixv.x, iyv.x, izv.x, ibv.x);
        pair(id.y, ik.y, ixv.y, iyv.y, izv.y, ibv.y);
        pair(id.z, ik.z, ixv.z, iyv.z, izv.z, ibv.z);
        pair(id.w, ik.w, ixv.w, iyv.w, izv.w, ibv.w);
        int cnt = nbef | (nbad << 8);
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
        const int pos = cnt & 0xFF;
        const int firstbad = -wave_max_i32_id((cnt >> 8) ? -pos : -2 * PF_WAVES);
        const int L = firstbad < m - j ? firstbad : m - j;
        const bool taken = rb4 == 0 && pos < L;
        if (tid < 64 && taken) {
            const unsigned rank = 0x0FFFFFFFu - ((unsigned)jk >> 4);
            idxs[j + pos] = (int)pf_unrank(rank, l2, rb);
        }
        pend = __ballot(taken && (j + pos) != m - 1);  // the reference never applies its last pick to `temp`
        ax = jx; ay = jy; az = jz;
        j += L;
        round += 1;
    }

    // the reference leaves the final running min-distances in `temp` (original order)
    __threadfence();
    __syncthreads();
    if (temp)
        for (int p = tid; p < n; p += PF_THREADS) {
            const float tv = __hip_atomic_load(st + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            temp[pf_unrank((unsigned)srk[p], l2, rb)] = tv;
        }
    if (progress && tid == 0)   // (behind the fence + barrier above: every pick of this workgroup is visible device-wide)
        __hip_atomic_store(progress + scene, m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

size_t fps_cluster_exchange_floats();
int launch_fps_pruned_cluster(int b, int K, int T, int n, int m, const float *dataset, float *temp, int *idxs, float *work,
                              long long stride, int *progress, hipStream_t st);

// workspace floats per scene for sps_fps_with_workspace (0: this size is served without one): the sorted points and,
// behind them, the record exchange area of fps_pruned_cluster.hip
size_t fps_big_workspace_elems(int n) {
    // (6144 .. 16 384 points: the register-resident kernel, whose scenes a pre-pass sorts into the workspace, fps_presort.hip)
    if (n < 6144 || n > 8 * 64 * 64 * PF_WAVES) return 0;
    return (size_t)5 * ((size_t)divup(n, 64) * 64) + fps_cluster_exchange_floats();
}

// Workgroups per scene (K <= 16) and records each publishes per round (T <= 8, K T <= 64); SPS_FPS_CLUSTER="K,T" overrides
// ("1" = one workgroup, the kernel below).  Measured on MI355X (tools/fps_cluster_probe.py; one workgroup = 1.00):
//   round 2-4 (K <= 8, K T <= 32):
//   180 000 -> 16 384, 1 scene    K,T = 2,8  0.71   4,8  0.41   8,3  0.33   8,4  0.30  (32.8 -> 9.95 ms; 8.19 ms by round 4)
//    65 536 -> 16 384, 2 scenes         2,8  0.97   4,8  0.56              8,4  0.50  (16.4 -> 8.27 ms)
//    32 768 ->  8 192, 8 scenes         2,8  1.04   4,8  0.64              8,4  0.59  ( 7.5 -> 4.43 ms)
//   round 5 (K <= 16, K T <= 64: one lane per record in the acceptance; profiles/round5/r5g_fps_cluster_64_records.txt):
//   180 000 -> 16 384, 1 scene    K,T = 8,4  8.56 ms   8,8  7.81   12,5  7.12   16,2  8.86   16,3  7.38   16,4  6.89
//   180 000 -> 16 384, 4 scenes         8,4  8.99      8,8  8.21                                          16,4  7.36
//    65 536 -> 16 384, 2 scenes         8,4  7.24      8,8  6.46                             16,3  6.65   16,4  6.04
//    32 768 ->  8 192, 8 scenes         8,4  3.93      8,6  3.68   8,8  3.68
// A round is the cross-workgroup hop (poll + barrier: 4.2 k of 12.5 k cycles at 8,4) plus an apply phase that grows with the
// buckets a workgroup holds (5.0 k at K = 8, 2.9 k at K = 16: tools/fps_cluster_profile.py), so more workgroups AND more
// records per round both pay: 16,4 runs 1237 rounds of 13.5 k cycles (13.2 picks each) where 8,4 ran 1482 of 12.5 k (11.1
// picks) -- although the acceptance over 64 record slots, one lane per record and the i-records dealt over the eight waves,
// costs 2.0 k cycles per round where the 32-slot form cost 1.1 k (profiles/round5/r5h_fps_cluster_phase_profile_64_records.txt).
// On top (round 5, same table re-measured: r5m_fps_cluster_xcd_local_exchange_ab.txt): the records of a scene whose workgroups share an
// XCD travel through that XCD's L2 (granule_store_xcd / granule_load_xcd: -0.8 k cycles of hop per round) and a wave reads and writes
// the running distances of its own buckets with plain accesses instead of agent-scope atomics:
//   180 000 x 1 at 16,4: 6.62 ms   65 536 x 2 at 16,4: 5.65 ms   32 768 x 8 at 8,8: 3.33 ms   (8.19 / 6.78 / 3.93 in round 4)
// (K = 2 columns of round 2 measured before the per-bucket application of a round's centres, the others with it.)
// The K workgroups of a scene spin on each other's records, so all b K must be resident at once: at most 64 (a CU each).
static void fps_cluster_shape(int b, int n, int &K, int &T) {
    K = 16;
    while (K > 1 && b * K > 64) K >>= 1;
    if (K == 2 && n < 65536) K = 1;
    T = K >= 16 ? 4 : 8;
    const char *env = getenv("SPS_FPS_CLUSTER");   // (read per launch: tests switch it)
    if (env && *env) {
        int k = 1, t = 8;
        if (sscanf(env, "%d,%d", &k, &t) >= 1) { K = k; T = t; }
        while (K > 1 && b * K > 64) K >>= 1;
    }
}

// SPS_FPS_CLUSTER_SMALL=1 (DIAGNOSTIC A/B, read per launch): scenes of 6144 .. 16 384 points that come with a workspace take the
// clustered kernel (K workgroups per scene, points in the L2-resident workspace) instead of the register-resident one
static bool fps_cluster_small() {
    const char *e = getenv("SPS_FPS_CLUSTER_SMALL");
    return e && *e && *e != '0';
}

int launch_fps_pruned_big(int b, int n, int m, const float *dataset, float *temp, int *idxs, float *work, hipStream_t st) {
    if (fps_big_workspace_elems(n) == 0 || !work) return -1;
    if (n <= 32 * PF_THREADS && !fps_cluster_small())   // register-resident kernel behind the sorting pre-pass (-1: shape not served, caller falls back)
        return launch_fps_pruned(b, n, m, dataset, temp, idxs, st, nullptr, nullptr, work, (long long)fps_big_workspace_elems(n));
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int npad = divup(n, 64) * 64;
    const long long stride = (long long)fps_big_workspace_elems(n);
    int K, T;
    fps_cluster_shape(b, n, K, T);
    if (K > 1) {
        const int rc = launch_fps_pruned_cluster(b, K, T, n, m, dataset, temp, idxs, work, stride, nullptr, st);
        if (rc >= 0) return rc;
    }
    const int rows = divup(npad / 64, 64 * PF_WAVES);
    dim3 grid(b), block(PF_THREADS);
#define SPS_PB_CASE(R)                                                                                               \
    if (rows <= R) {                                                                                                 \
        hipLaunchKernelGGL((fps_pruned_big_kernel<R>), grid, block, 0, st, n, m, bs, l2, rb, npad, stride, dataset, temp, idxs, work); \
        return check_launch("fps_pruned_big_kernel");                                                               \
    }
    SPS_PB_CASE(2)
    SPS_PB_CASE(4)
    SPS_PB_CASE(8)
#undef SPS_PB_CASE
    return -1;
}

// Behind a clustered launch: the one-workgroup kernel for the scenes whose give-up word is up (a bounded cross-workgroup poll
// ran out -- e.g. the K workgroups were not resident together); the other scenes' workgroups leave at once.
int launch_fps_big_redo_given_up(int b, int n, int m, const float *dataset, float *temp, int *idxs, float *work, int *progress,
                                 hipStream_t st) {
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int npad = divup(n, 64) * 64;
    const long long stride = (long long)fps_big_workspace_elems(n);
    const int rows = divup(npad / 64, 64 * PF_WAVES);
    dim3 grid(b), block(PF_THREADS);
#define SPS_PB_CASE(R)                                                                                               \
    if (rows <= R) {                                                                                                 \
        hipLaunchKernelGGL((fps_pruned_big_kernel<R>), grid, block, 0, st, n, m, bs, l2, rb, npad, stride, dataset, temp, idxs, work, \
                           1, progress);                                                                             \
        return check_launch("fps_pruned_big_kernel<given up>");                                                     \
    }
    SPS_PB_CASE(2)
    SPS_PB_CASE(4)
    SPS_PB_CASE(8)
#undef SPS_PB_CASE
    return fail(SPS_ERR_INVALID, "fps(cluster): no one-workgroup kernel for n=%d", n);
}

// publishing variant for chunked consumers (large scenes: the clustered kernel only); -1 if it does not apply
int launch_fps_big_publish(int b, int n, int m, const float *dataset, float *temp, int *idxs, int *progress, float *work,
                           hipStream_t st) {
    if (fps_big_workspace_elems(n) == 0 || !work || m < 2) return -1;
    if (n <= 32 * PF_THREADS && !fps_cluster_small()) return -1;
    int K, T;
    fps_cluster_shape(b, n, K, T);
    if (K < 2) return -1;
    return launch_fps_pruned_cluster(b, K, T, n, m, dataset, temp, idxs, work, (long long)fps_big_workspace_elems(n), progress, st);
}

}  // namespace sps

extern "C" long long sps_fps_workspace_floats(int n) { return (long long)sps::fps_big_workspace_elems(n); }

// farthest_point_sampling_kernel_launcher with an optional device workspace of b * sps_fps_workspace_floats(n) floats:
// with it, scenes of 16 385 .. 262 144 points take the pruned large-scene kernel; without (or for other sizes) this is
// sps_farthest_point_sampling_kernel_launcher.
extern "C" int sps_fps_with_workspace(int b, int n, int m, const float *dataset, float *temp, int *idxs, float *work,
                                      sps_stream_t stream) {
    using namespace sps;
    if (b < 0 || n <= 0 || m < 0) return fail(SPS_ERR_INVALID, "fps: bad shape b=%d n=%d m=%d", b, n, m);
    if (b == 0 || m == 0) return SPS_OK;
    if (!dataset || !temp || !idxs) return fail(SPS_ERR_INVALID, "fps: null pointer");
    if (work && m > 1 && fps_mode() == 0) {
        const int rc = launch_fps_pruned_big(b, n, m, dataset, temp, idxs, work, as_stream(stream));
        if (rc >= 0) return rc;
    }
    return sps_farthest_point_sampling_kernel_launcher(b, n, m, dataset, temp, idxs, stream);
}
