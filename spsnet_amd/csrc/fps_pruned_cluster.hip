// fps_pruned_cluster.hip -- the large-scene farthest point sampling of fps_pruned_big.hip with ONE SCENE SPREAD OVER K
// COMPUTE UNITS ("split a scene over a few CUs on the same GPU", BASELINE config 5: 180 000 points).
//
// Same contract and the same results (indices and final `temp`, bit for bit) as fps_pruned_big.hip / the reference
// (sampling_gpu.cu:93-253); same buckets, box test, records and acceptance proof.  What changes:
//   * the scene's buckets are dealt round-robin over the 8 K waves of K workgroups (a wave owns a quarter of the buckets it
//     owned at K = 4: fewer box tests per centre, fewer fetched buckets per round on the busiest wave); with so few
//     touched buckets per wave the apply phase is a chain of L2 round trips, so a round's centres are applied PER BUCKET
//     (fetched once, minimum over its centres in registers, refreshed once) with the next bucket's loads in flight;
//   * every round each workgroup ranks its 16 wave records and PUBLISHES ITS T BEST to the other workgroups; the last of
//     them carries the value of the best record held back in its bound, so the acceptance argument of fps_pruned.hip
//     holds unchanged over the K T published records (a candidate behind it must beat everything still hidden);
//   * the exchange is one hop per round: 8-byte {value, round tag} granules written and polled with agent-scope atomics
//     (MI355X_MICROARCH.md, hand-off price list: ~1 us; no fence, no separate flag), double-buffered by round parity --
//     a workgroup can publish round r+2 only after everybody has published r+1, i.e. finished reading r;
//   * a workgroup polls the K T records with one granule per lane (384 lanes at 64 records: one round trip; eight waves
//     each polling everything cost 1 ms of 13 more) into LDS, a second barrier hands them to its waves;
//   * the acceptance over the K T <= 64 records (round 5; 32 before) holds one record per lane, the i-records it is tested
//     against dealt over the eight waves (eight each), partial counts meeting in LDS behind one more barrier; each wave
//     then applies the accepted centres (a 64-bit mask per bucket) to its own buckets only.
// More records per round and less work per wave: 180 000 -> 16 384 in 6.9 ms at K = 16, T = 4 (round 5; 8.2 ms at 8 x 4,
// 9.95 ms when first built) against 32.8 ms on one CU
// (fps_pruned_big.hip's launcher holds the measured table and picks K, T).
// The K workgroups of a scene must be resident together: the launcher uses the cluster only for b K <= 64 workgroups
// (a CU each; the polls are bounded -- a poll that runs out raises the scene's give-up word, its workgroups leave and the
// launcher's follow-up launch samples that scene with the one-workgroup kernel: correct or redo, never a trap or a hang).
#include "fps_sort_split.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace sps {

namespace {

constexpr int IMIN_C = (int)0x80000000;
#ifdef SPS_PC_ATOMIC_ST
constexpr bool PLAIN_ST = false;     // (A/B build: the running distances through agent-scope atomics, as in rounds 2-4)
#else
constexpr bool PLAIN_ST = true;
#endif

struct PcShared {
    PcSortShared sort;
    __attribute__((aligned(16))) int soa[2][6][2 * PF_WAVES];  // the workgroup's own 16 records: round parity x field x record
    __attribute__((aligned(16))) int xr[6][PC_MAXR];           // the round's published records of all K workgroups
    int part[PF_WAVES][64];                                    // acceptance: per-wave partial counts of a lane's pairs
};

template <int R, int ROWS, class F>
__device__ __forceinline__ void rows_each_c(F &fn) {
    if constexpr (R < ROWS) {
        fn(std::integral_constant<int, R>{});
        rows_each_c<R + 1, ROWS>(fn);
    }
}

}  // namespace

// work: per scene `stride` floats: 5 arrays of npad elements (x, y, z, t, rank), then the exchange area (zeroed by the launcher)
template <int ROWS>
__global__ __launch_bounds__(PF_THREADS) void fps_pruned_cluster_kernel(int b, int K, int T, int n, int m, int bs, int l2, int rb,
                                                                        int npad, long long stride,
                                                                        const float *__restrict__ dataset,
                                                                        float *__restrict__ temp, int *__restrict__ idxs,
                                                                        float *__restrict__ work, int *__restrict__ progress,
                                                                        int spread, unsigned spin_limit, int xcd_local) {
    // progress != NULL: consumers on other CUs read idxs while this kernel runs (sa_stack's streamed first layer) -- picks
    // are stored write-through and progress[scene] counts the published ones (every 64 picks; fps_pruned.hip's protocol).
    // temp may then be NULL: all running distances start at 1e10 and are not handed back.
    if (m <= 0) return;
    __shared__ PcShared sh;
    // blocks s, s + 8, s + 16, ... share an XCD (observed dispatch order): a scene's K workgroups sit on one L2
    // (spread: DIAGNOSTIC mapping that puts a scene's workgroups on consecutive blocks = different XCDs; tests run the
    //  cross-XCD form of every exchange with it)
    const int scene = spread ? (int)blockIdx.x / K : (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * K));
    const int cu = spread ? (int)blockIdx.x % K : (blockIdx.x >> 3) % K;
    if (scene >= b) return;
    const float *xyz = dataset + (size_t)scene * n * 3;
    if (temp) temp += (size_t)scene * n;
    idxs += (size_t)scene * m;
    float *sx = work + (size_t)scene * stride, *sy = sx + npad, *sz = sy + npad, *st = sz + npad;
    int *srk = reinterpret_cast<int *>(st + npad);
    unsigned long long *xg = reinterpret_cast<unsigned long long *>(work + (size_t)scene * stride + (size_t)5 * npad);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gwave = cu * PF_WAVES + wave, nwaves = K * PF_WAVES;   // this wave among the scene's waves
    const int nb = npad / 64;  // buckets
    const int R = K * T;       // records per round

    // ------------------------------------------------------------------ spatial sort (once), split over the K workgroups
    // Every cross-workgroup poll is bounded; one that gives up raises the scene's give-up word (granule 0 of the zeroed
    // exchange area), all K workgroups leave, and the launcher's follow-up launch samples the scene with the one-workgroup
    // kernel (fps_pruned_big.hip) -- nothing written so far is used (temp is only written at the very end).
    const PcGiveUp gu{xg, 1u, spin_limit};
    // (every workgroup publishes its XCC_ID with its sort flag: the K workgroups of a scene that find themselves on ONE XCD
    //  exchange their records through that XCD's L2 instead of the memory side -- granule_store_xcd / granule_load_xcd)
    const int my_xcc = (int)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xFFu);   // HW_REG_XCC_ID
    if (!pc_sort_split<PC_MAXK>(sh.sort, cu, K, n, npad, bs, l2, rb, xyz, temp, xg, sx, sy, sz, st, srk, true, gu, nullptr, 2u,
                                1 + my_xcc))
        return;
    bool local = false;
    if (xcd_local) {
        const int theirs = lane < K ? (int)(unsigned)granule_load(xg + PC_FLAG_AT + lane) : 1 + my_xcc;
        local = __ballot(theirs != 1 + my_xcc) == 0ull;      // (every workgroup reads the same K values: the same verdict)
    }

    // ------------------------------------------------------------------ bucket metadata
    // bucket g = v * nwaves + gwave is slot v of this wave: row v / 64, lane v % 64
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
        r_sec = wave_max_i32_id(lane != wl ? tb : IMIN_C);
        const int rank = __builtin_amdgcn_readlane(rv, wl);
        r_px = __builtin_amdgcn_readlane(__float_as_int(xv), wl);
        r_py = __builtin_amdgcn_readlane(__float_as_int(yv), wl);
        r_pz = __builtin_amdgcn_readlane(__float_as_int(zv), wl);
        r_vmax = vmax;
        r_keylo = (int)((0x0FFFFFFFu - (unsigned)rank) << 4);
    };
    auto commit = [&](auto rc, int l) {  // metadata of (row, lane l) <- the scalars refresh() left
        constexpr int RW = decltype(rc)::value;
        int a0 = bmax[RW], a1 = bsec[RW], a2 = bkeylo[RW], a3 = __float_as_int(bpx[RW]), a4 = __float_as_int(bpy[RW]), a5 = __float_as_int(bpz[RW]);
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
        bmax[RW] = a0; bsec[RW] = a1; bkeylo[RW] = a2; bpx[RW] = __int_as_float(a3); bpy[RW] = __int_as_float(a4); bpz[RW] = __int_as_float(a5);
    };
    auto for_rows = [&](auto &&fn) { rows_each_c<0, ROWS>(fn); };
    for_rows([&](auto rc) {
        constexpr int RW = decltype(rc)::value;
        for (int l = 0; l < 64; ++l) {
            const int g = (RW * 64 + l) * nwaves + gwave;
            if (g >= nb) break;
            const size_t p = (size_t)g * 64 + lane;
            const float xv = sx[p], yv = sy[p], zv = sz[p], tv = st[p];
            const int rv = srk[p];
            float lx = xv, ly = yv, lz = zv, hx = xv, hy = yv, hz = zv;
            wave_box6(lx, ly, lz, hx, hy, hz);
            if (lane == l) { blo_x[RW] = lx; blo_y[RW] = ly; blo_z[RW] = lz; bhi_x[RW] = hx; bhi_y[RW] = hy; bhi_z[RW] = hz; }
            refresh(tv, rv, xv, yv, zv);
            commit(rc, l);
        }
    });

    if (cu == 0 && tid == 0) __hip_atomic_store(&idxs[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // record slots beyond R never come before anything (most negative distance, lowest key)
    for (int i = tid; i < 6 * PC_MAXR; i += PF_THREADS) sh.xr[i / PC_MAXR][i % PC_MAXR] = (i / PC_MAXR == 0) ? IMIN_C : 0;
    __syncthreads();

    // ------------------------------------------------------------------ sampling loop
    int crec = 0;
    bool cand_stale = true;
    int cand_e1 = -1, cand_e2 = -1;
    float ax = xyz[0], ay = xyz[1], az = xyz[2];
    unsigned long long pend = m > 1 ? 1ull : 0ull;
    int j = 1, round = 0;
    const bool younger = wave >= PF_WAVES / 2;
#ifdef SPS_PC_PROFILE
    unsigned long long pseg[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long {
        unsigned long long tt;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return tt;
    };
#define PC_STAMP(var) const unsigned long long var = stamp()
#else
#define PC_STAMP(var)
#endif
    for (;;) {
        if (younger) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        PC_STAMP(p0);
        // ---- apply the round's accepted centres (lanes `pend`: lane rr holds record rr, bit rr names a centre) to my buckets.
        // With ~1.5 touched buckets per wave and round the phase is a chain of L2 round trips, so: (1) all box tests first --
        // lane l collects, per row, the centres its bucket cannot rule out; (2) every touched bucket is fetched ONCE
        // (x, y, z, t and the ranks together), takes the minimum over its centres in registers and is refreshed once;
        // (3) the next touched bucket's loads are in flight meanwhile.  Same minima as one centre at a time: the box tests
        // use the bucket maxima of the round's start (a superset of what the sequential order would fetch).
        if (pend) {
            for_rows([&](auto rc) {
                constexpr int RW = decltype(rc)::value;
                unsigned long long cm = 0ull;
                for (unsigned long long pm = pend; pm; pm &= pm - 1) {
                    const int rr = __builtin_ctzll(pm);
                    const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), rr));
                    const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), rr));
                    const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), rr));
                    const float qx = __builtin_amdgcn_fmed3f(cx, blo_x[RW], bhi_x[RW]);
                    const float qy = __builtin_amdgcn_fmed3f(cy, blo_y[RW], bhi_y[RW]);
                    const float qz = __builtin_amdgcn_fmed3f(cz, blo_z[RW], bhi_z[RW]);
                    const float lb = sqdist(qx, qy, qz, cx, cy, cz);
                    const bool skip = lb >= __int_as_float(bmax[RW]);  // NaN -> not skipped
                    cm |= skip ? 0ull : (1ull << rr);
                }
                unsigned long long todo = __ballot(cm != 0ull);
                if (todo) {
                    struct Bucket { float x, y, z, t; int rk; };
                    auto fetch = [&](int l) -> Bucket {
                        const size_t p = (size_t)((RW * 64 + l) * nwaves + gwave) * 64 + lane;
                        Bucket bk;
                        bk.x = sx[p]; bk.y = sy[p]; bk.z = sz[p];
                        // (a bucket's running distances are read and written by the ONE wave that owns it: plain accesses, served
                        //  by this XCD's L2 -- round 2-4 used agent-scope atomics here, i.e. memory-side round trips)
                        bk.t = PLAIN_ST ? st[p] : __hip_atomic_load(st + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bk.rk = srk[p];
                        return bk;
                    };
                    int l = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    Bucket cur = fetch(l);
                    for (;;) {
                        const bool more = todo != 0ull;
                        const int ln = more ? __builtin_ctzll(todo) : l;
                        todo &= todo - 1;                       // (0 stays 0)
                        Bucket nxt = cur;
                        if (more) nxt = fetch(ln);              // in flight while this bucket is worked on
                        unsigned long long mc = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(cm >> 32), l) << 32) |
                                                (unsigned)__builtin_amdgcn_readlane((int)cm, l);
                        float tn = cur.t;
                        bool moved = false;
                        while (mc) {
                            const int rr = __builtin_ctzll(mc);
                            mc &= mc - 1;
                            const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax), rr));
                            const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay), rr));
                            const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az), rr));
                            const float d = sqdist(cur.x, cur.y, cur.z, cx, cy, cz);
                            moved |= __ballot(d < tn) != 0ull;   // some point moved closer to a sample
                            tn = fmin_raw(d, tn);
                        }
                        if (moved) {
                            const size_t p = (size_t)((RW * 64 + l) * nwaves + gwave) * 64 + lane;
                            if (PLAIN_ST) st[p] = tn;
                            else __hip_atomic_store(st + p, tn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            refresh(tn, cur.rk, cur.x, cur.y, cur.z);
                            commit(rc, l);
                            const int e = (RW << 6) | l;
                            if (e == cand_e1 || e == cand_e2) cand_stale = true;
                        }
                        if (!more) break;
                        l = ln;
                        cur = nxt;
                    }
                }
            });
            pend = 0ull;
        }
        if (j >= m) break;
        PC_STAMP(p1);
        // the wave's two records (fps_pruned_big.hip)
        if (cand_stale) {
            auto best_entry = [&](int excl1, int excl2, int &vmax, int &row) -> int {
                int lv = IMIN_C, lk = -1, lr = 0;
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
                int klo = 0, px = 0, py = 0, pz = 0, sec = IMIN_C;
                for_rows([&](auto rc) {
                    constexpr int RW = decltype(rc)::value;
                    if (row == RW) {
                        klo = __builtin_amdgcn_readlane(bkeylo[RW], wl);
                        px = __builtin_amdgcn_readlane(__float_as_int(bpx[RW]), wl);
                        py = __builtin_amdgcn_readlane(__float_as_int(bpy[RW]), wl);
                        pz = __builtin_amdgcn_readlane(__float_as_int(bpz[RW]), wl);
                        sec = __builtin_amdgcn_readlane(bsec[RW], wl);
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
            record(r1, wl1, v1, IMIN_C, std::integral_constant<int, 0>{});
            record(r2, wl2, v2, v3, std::integral_constant<int, 1>{});
            cand_e1 = e1; cand_e2 = e2;
            cand_stale = false;
        }
        // ---- the workgroup's 16 records meet in LDS; its first wave ranks them and publishes the T best
        PC_STAMP(p2);
        const int buf = round & 1;
        const unsigned tag = (unsigned)round + 2u;   // (1 is the "sorted" flag's tag, 0 the zeroed area)
        unsigned long long *xround = xg + 8 + (size_t)buf * (PC_MAXR * 6);
        if (lane < 12) sh.soa[buf][lane >> 1][2 * wave + (lane & 1)] = crec;
        __syncthreads();
        PC_STAMP(p3);
        if (younger) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
        if (wave == 0) {
            // position of record rj among the workgroup's 16: records before it under (distance, key), equal ones (only empty
            // records are) in index order -- a permutation of 0 .. 15
            const int rj = lane >> 2, rb4 = lane & 3;
            const int4 id = *(const int4 *)&sh.soa[buf][0][rb4 * 4], ik = *(const int4 *)&sh.soa[buf][1][rb4 * 4];
            const int jd = sh.soa[buf][0][rj], jk = sh.soa[buf][1][rj];
            int nbef = 0;
            auto before = [&](int idist, int iklo, int ii) {
                const bool same = (idist == jd) & (iklo == jk);
                nbef += ((idist > jd) | ((idist == jd) & ((unsigned)iklo > (unsigned)jk)) | (same & (ii < rj))) ? 1 : 0;
            };
            before(id.x, ik.x, rb4 * 4 + 0); before(id.y, ik.y, rb4 * 4 + 1);
            before(id.z, ik.z, rb4 * 4 + 2); before(id.w, ik.w, rb4 * 4 + 3);
            nbef += __builtin_amdgcn_update_dpp(0, nbef, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
            nbef += __builtin_amdgcn_update_dpp(0, nbef, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
            const int hid = wave_max_i32_id(nbef == T ? jd : IMIN_C);   // the best record held back (none: -0.0)
            if (rb4 == 0 && nbef < T) {
                int bound = sh.soa[buf][5][rj];
                if (nbef == T - 1) bound = imax(bound, hid);   // whoever comes behind my last record must beat what I hide
                unsigned long long *dst = xround + (size_t)(cu * T + nbef) * 6;
                const int f2 = sh.soa[buf][2][rj], f3 = sh.soa[buf][3][rj], f4 = sh.soa[buf][4][rj];
                if (local) {
                    granule_store_xcd(dst + 0, jd, tag); granule_store_xcd(dst + 1, jk, tag); granule_store_xcd(dst + 2, f2, tag);
                    granule_store_xcd(dst + 3, f3, tag); granule_store_xcd(dst + 4, f4, tag); granule_store_xcd(dst + 5, bound, tag);
                } else {
                    granule_store(dst + 0, jd, tag); granule_store(dst + 1, jk, tag); granule_store(dst + 2, f2, tag);
                    granule_store(dst + 3, f3, tag); granule_store(dst + 4, f4, tag); granule_store(dst + 5, bound, tag);
                }
            }
        }
        PC_STAMP(p4);
        // ---- the workgroup polls the K T published records, one granule per lane (granule = {value, tag}: nothing else to
        // order), into ONE copy in LDS; a second barrier hands it to every wave
        {
            const int ng = R * 6;
            if (tid < ng) {
                unsigned long long g = 0;
                if (!pc_bounded_poll(gu, 1, [&] {
                        g = local ? granule_load_xcd(xround + tid) : granule_load(xround + tid);
                        return (unsigned)(g >> 32) == tag;
                    })) {
                    sh.sort.giveup = 1;
                    gu.raise();
                }
                sh.xr[tid % 6][tid / 6] = (int)(unsigned)g;
            }
        }
        __syncthreads();
        if (sh.sort.giveup) return;
        PC_STAMP(p5);
        // ---- acceptance over the R <= 64 records: lane rj holds record rj
        const int rj = lane;
        const int jd = sh.xr[0][rj], jk = sh.xr[1][rj];
        const float jx = __int_as_float(sh.xr[2][rj]), jy = __int_as_float(sh.xr[3][rj]);
        const float jz = __int_as_float(sh.xr[4][rj]);
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
        {
            // the 64 i-records are dealt over the eight waves, eight each (every wave evaluating all of them was 3.0 k cycles
            // of a 14.9 k round at 32 records: two waves per SIMD issuing the same VALU instructions); the partial counts
            // meet in LDS behind one more barrier, which aligned waves cross in a few hundred cycles
            static_assert(PC_MAXR == 8 * PF_WAVES, "eight i-records per wave");
            const int i0 = wave * 8;
            // (slots beyond R hold records that are never "before" anything)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int4 id = *(const int4 *)&sh.xr[0][i0 + 4 * u], ik = *(const int4 *)&sh.xr[1][i0 + 4 * u];
                const int4 ixv = *(const int4 *)&sh.xr[2][i0 + 4 * u], iyv = *(const int4 *)&sh.xr[3][i0 + 4 * u];
                const int4 izv = *(const int4 *)&sh.xr[4][i0 + 4 * u], ibv = *(const int4 *)&sh.xr[5][i0 + 4 * u];
                pair(id.x, ik.x, ixv.x, iyv.x, izv.x, ibv.x);
                pair(id.y, ik.y, ixv.y, iyv.y, izv.y, ibv.y);
                pair(id.z, ik.z, ixv.z, iyv.z, izv.z, ibv.z);
                pair(id.w, ik.w, ixv.w, iyv.w, izv.w, ibv.w);
            }
            sh.part[wave][lane] = nbef | (nbad << 8);
            __syncthreads();
            int sum = 0;
#pragma unroll
            for (int w = 0; w < PF_WAVES; ++w) sum += sh.part[w][lane];
            nbef = sum & 0xFF;
            nbad = sum >> 8;
        }
        const int pos = nbef;
        const bool real = rj < R;
        const int firstbad = -wave_max_i32_id((real && nbad) ? -pos : -R);   // nobody bad: all R records
        const int L = firstbad < m - j ? firstbad : m - j;
        const bool taken = real && pos < L;
        if (cu == 0 && wave == 0) {
            if (taken) {
                const unsigned rank = 0x0FFFFFFFu - ((unsigned)jk >> 4);
                __hip_atomic_store(&idxs[j + pos], (int)pf_unrank(rank, l2, rb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (progress && lane == 0 && (((j + L) >> 6) != (j >> 6) || j + L == m)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left
                __hip_atomic_store(&progress[scene], j + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        pend = __ballot(taken && (j + pos) != m - 1);  // the reference never applies its last pick to `temp`
        ax = jx; ay = jy; az = jz;
        j += L;
        round += 1;
#ifdef SPS_PC_PROFILE
        {
            const unsigned long long p6 = stamp();
            pseg[0] += p1 - p0; pseg[1] += p2 - p1; pseg[2] += p3 - p2; pseg[3] += p4 - p3; pseg[4] += p5 - p4; pseg[5] += p6 - p5;
        }
#endif
    }
#ifdef SPS_PC_PROFILE
    // (diagnostic build only: s_memtime ticks = 10 ns; per wave: apply, records, barrier 1, rank + publish, poll + barrier 2, accept, rounds)
    if (lane == 0) {
        unsigned long long *o = xg + PC_GRANULES + (size_t)PC_MAXK * PF_BINS / 2 - 8 * (size_t)(nwaves - gwave);
        for (int i = 0; i < 6; ++i) o[i] = pseg[i];
        o[6] = (unsigned long long)round;
    }
#endif

    // the reference leaves the final running min-distances in `temp` (original order): every wave writes its own buckets
    if (temp)
    for (int v = 0; v < 64 * ROWS; ++v) {
        const int g = v * nwaves + gwave;
        if (g >= nb) break;
        const size_t p = (size_t)g * 64 + lane;
        if (p < (size_t)n) {
            const float tv = PLAIN_ST ? st[p] : __hip_atomic_load(st + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            temp[pf_unrank((unsigned)srk[p], l2, rb)] = tv;
        }
    }
}

// floats of the exchange area behind a scene's 5 npad workspace floats
size_t fps_cluster_exchange_floats() { return (size_t)2 * PC_GRANULES + (size_t)PC_MAXK * PF_BINS; }

// SPS_FPS_CLUSTER_SPREAD=1 (read per launch): the diagnostic block mapping
int fps_cluster_spread() {
    const char *e = getenv("SPS_FPS_CLUSTER_SPREAD");
    return e && *e && *e != '0';
}

// K workgroups per scene publishing T records each; -1: shape not served
int launch_fps_big_redo_given_up(int b, int n, int m, const float *dataset, float *temp, int *idxs, float *work, int *progress,
                                 hipStream_t st);   // fps_pruned_big.hip

int launch_fps_pruned_cluster(int b, int K, int T, int n, int m, const float *dataset, float *temp, int *idxs, float *work,
                              long long stride, int *progress, hipStream_t st) {
    if (K < 2 || K > PC_MAXK || T < 1 || T > PC_MAXT || K * T > PC_MAXR || b * K > 64 || !work) return -1;
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int npad = divup(n, 64) * 64;
    const int rows = divup(npad / 64, 64 * PF_WAVES * K);
    if (rows > 8) return -1;
    // the exchange areas start zeroed (tags of an earlier launch must not be mistaken for this one's)
    hipError_t e = hipMemset2DAsync(work + (size_t)5 * npad, (size_t)stride * sizeof(float), 0,
                                    (size_t)PC_GRANULES * 8, (size_t)b, st);   // (the granules; the histograms behind them are written before they are read)
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "fps(cluster): hipMemset2DAsync: %s", hipGetErrorString(e));
    dim3 grid(8 * K * divup(b, 8)), block(PF_THREADS);
    const int spread = fps_cluster_spread();
    // SPS_FPS_CLUSTER_XCD (read per launch; default 1): record exchange through the XCD's L2 where a scene's workgroups share one
    const char *xe = getenv("SPS_FPS_CLUSTER_XCD");
    const int xcd_local = !(xe && *xe == '0');
#define SPS_PC_CASE(RW)                                                                                                \
    if (rows <= RW) {                                                                                                  \
        hipLaunchKernelGGL((fps_pruned_cluster_kernel<RW>), grid, block, 0, st, b, K, T, n, m, bs, l2, rb, npad, stride, \
                           dataset, temp, idxs, work, progress, spread, pc_spin_limit(), xcd_local);                  \
        if (check_launch("fps_pruned_cluster_kernel") != SPS_OK) return SPS_ERR_LAUNCH;                               \
        /* the scenes a bounded poll gave up on (normally none: a launch whose workgroups leave at once) */           \
        return launch_fps_big_redo_given_up(b, n, m, dataset, temp, idxs, work, progress, st);                        \
    }
    SPS_PC_CASE(1)
    SPS_PC_CASE(2)
    SPS_PC_CASE(4)
    SPS_PC_CASE(8)
#undef SPS_PC_CASE
    return -1;
}

}  // namespace sps
