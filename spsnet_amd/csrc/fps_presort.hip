// fps_presort.hip -- the spatial sort of the register-resident FPS kernel (fps_pruned.hip) as a pre-pass of its own.
//
// Inside fps_pruned_kernel one workgroup per scene sorts its scene first: two latency-bound passes over the points for the
// bounding box and the cell histogram, a scan, the scatter of point indices and an indexed (uncoalesced) load of every
// point -- ~75 us of a 1.79 ms launch on 8 of 256 compute units.  Here K workgroups per scene do the same sort in
// parallel (fps_sort_split.h: boxes, histograms and offsets exchanged through tagged granules and the workspace) and leave
// {x, y, z, running distance, rank} sorted in a workspace; the FPS kernel's PRESORT instantiation then starts with 5 P
// coalesced loads per lane.  Same sort key, same buckets up to the (arbitrary, irrelevant) order inside a cell.
#include "fps_sort_split.h"

#include <atomic>
#include <mutex>

namespace sps {

// work: per scene `stride` floats: 5 arrays of npad elements (x, y, z, t, rank), then the exchange area (zeroed by the launcher)
__global__ __launch_bounds__(PF_THREADS) void fps_presort_kernel(int b, int K, int n, int bs, int l2, int rb, int npad,
                                                                 long long stride, const float *__restrict__ dataset,
                                                                 const float *__restrict__ temp, float *__restrict__ work,
                                                                 int spread, unsigned long long *flags, unsigned epoch,
                                                                 unsigned spin_limit) {
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
    // (the launch's end hands the sorted scene over; the flags of the histogram exchange: this launch's slot of the pool.
    //  The second round of flags is not used by the pre-pass: its first granule is the scene's give-up word -- a poll that gave
    //  up leaves it tagged with this launch's epoch, the FPS kernel behind skips the scene and the launcher's predicated
    //  follow-up launch samples it with the in-kernel sort, see presort_gate)
    unsigned long long *fl = flags + (size_t)scene * 2 * PC_MAXK;
    const PcGiveUp gu{fl + PC_MAXK, epoch, spin_limit};
    (void)pc_sort_split(sh, cu, K, n, npad, bs, l2, rb, xyz, temp, xg, sx, sy, sz, st, srk, false, gu, fl, epoch);
}

static std::atomic<unsigned> g_spin_limit{PC_SPIN_LIMIT};
unsigned pc_spin_limit() { return g_spin_limit.load(std::memory_order_relaxed); }

size_t fps_cluster_exchange_floats();   // fps_pruned_cluster.hip
int fps_cluster_spread();

// The flags of the pre-pass's one exchange live in a pool the library owns (per device, zeroed once): a launch takes the next of
// PS_SLOTS slots and a fresh 32-bit epoch as its tag, so nothing has to be zeroed per launch (a memset in front of the
// producer cost ~4 us + a launch gap of every pass).  A stale flag carries an older epoch and never matches; more than
// PS_SLOTS pre-passes in flight on one device would share a slot -- far beyond the 64-workgroup residency rule anyway.
constexpr int PS_SLOTS = 64, PS_SCENES = 64;
static unsigned long long *presort_flag_pool(int dev) {
    static std::mutex mu;
    static unsigned long long *pool[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) return nullptr;
    if (!pool[dev]) {
        const size_t bytes = (size_t)PS_SLOTS * PS_SCENES * 2 * PC_MAXK * sizeof(unsigned long long);
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        // (null-stream memset, then a device-wide wait: the first pre-pass is launched on a non-blocking stream and must not
        //  overtake the zeroes -- once per device and process)
        if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
        pool[dev] = static_cast<unsigned long long *>(p);
    }
    return pool[dev];
}

// sorted scenes -> work (b * stride floats); temp may be NULL (all running distances 1e10).  *gate: where the FPS kernels
// behind this launch find out which scenes the pre-pass gave up on (scene s: gate.word + s * gate.stride, raised = its tag
// equals gate.tag).
int launch_fps_presort(int b, int n, const float *dataset, const float *temp, float *work, long long stride, hipStream_t st,
                       PresortGate *gate) {
    int K = PC_MAXK;
    while (K > 1 && b * K > 64) K >>= 1;   // the K workgroups of a scene spin on each other: all of them resident
    if (b * K > 128 || b > PS_SCENES) return -1;
    const int bs = sps_opt_n_threads(n);
    int l2 = 0;
    while ((1 << (l2 + 1)) <= bs) ++l2;
    int rb = 0;
    while ((1 << rb) < divup(n, bs)) ++rb;
    const int npad = divup(n, 64) * 64;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    unsigned long long *pool = presort_flag_pool(dev);
    if (!pool) return -1;   // (the caller falls back to the in-kernel sort)
    static std::atomic<unsigned> counter{0};
    unsigned ticket = counter.fetch_add(1, std::memory_order_relaxed) + 1u;
    if ((ticket << 1) == 0u) ticket = counter.fetch_add(1, std::memory_order_relaxed) + 1u;   // tag 0 = the zeroed pool
    unsigned long long *flags = pool + (size_t)(ticket % PS_SLOTS) * PS_SCENES * 2 * PC_MAXK;
    hipLaunchKernelGGL(fps_presort_kernel, dim3(8 * K * divup(b, 8)), dim3(PF_THREADS), 0, st, b, K, n, bs, l2, rb, npad, stride,
                       dataset, temp, work, fps_cluster_spread(), flags, ticket << 1, pc_spin_limit());
    // (SPS_FPS_PRESORT_GATE=0, DIAGNOSTIC / A-B timing only: no gate, no follow-up launch -- a pre-pass that gave up would then
    //  leave its scenes unsampled)
    static const bool gate_on = [] { const char *e = getenv("SPS_FPS_PRESORT_GATE"); return !(e && *e == '0'); }();
    if (gate) *gate = gate_on ? PresortGate{flags + PC_MAXK, 2 * PC_MAXK, ticket << 1} : PresortGate{nullptr, 0, 0u};
    return check_launch("fps_presort_kernel");
}

}  // namespace sps

// DIAGNOSTIC: the spin bound of every cross-workgroup poll of the FPS kernels (the split sort of fps_presort.hip /
// fps_pruned_cluster.hip and the clustered kernel's record exchange); 0 restores the default, 0xFFFFFFFF makes every poll
// give up without looking -- how the tests drive the give-up -> redo path.  Returns the previous bound.
extern "C" unsigned sps_debug_set_exchange_spins(unsigned spins) {
    return sps::g_spin_limit.exchange(spins ? spins : sps::PC_SPIN_LIMIT, std::memory_order_relaxed);
}
