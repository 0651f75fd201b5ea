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
    unsigned long long *fl = flags + (size_t)scene * 2 * PS_MAXK;
    const PcGiveUp gu{fl + PS_MAXK, epoch, spin_limit};
    (void)pc_sort_split<PS_MAXK>(sh, cu, K, n, npad, bs, l2, rb, xyz, temp, xg, sx, sy, sz, st, srk, false, gu, fl, epoch);
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
// The pool is created by sps_init() -- the ONE entry point of the library that allocates and synchronises -- never by a
// launcher: a launch that finds no pool for its device declines (the caller then takes the kernel that sorts for itself),
// so that every launcher stays free of allocations and synchronisation (include/spsnet_sa.h: legal under stream capture).
static std::mutex g_pool_mu;
static unsigned long long *g_pool[64] = {};
static unsigned long long *presort_flag_pool(int dev) {
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_pool_mu);
    return g_pool[dev];
}
int presort_pool_create(int dev, hipStream_t st) {
    if (dev < 0 || dev >= 64) return fail(SPS_ERR_INVALID, "sps_init: device %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (g_pool[dev]) return SPS_OK;
    const size_t bytes = (size_t)PS_SLOTS * PS_SCENES * 2 * PS_MAXK * sizeof(unsigned long long);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(p, 0, bytes, st);
    // (the first pre-pass may be launched on any stream and must not overtake the zeroes: wait for them here, once)
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        if (p) (void)hipFree(p);
        return fail(SPS_ERR_LAUNCH, "sps_init: presort flag pool: %s", hipGetErrorString(e));
    }
    g_pool[dev] = static_cast<unsigned long long *>(p);
    return SPS_OK;
}
bool presort_pool_exists(int dev) { return presort_flag_pool(dev) != nullptr; }

// sorted scenes -> work (b * stride floats); temp may be NULL (all running distances 1e10).  *gate: where the FPS kernels
// behind this launch find out which scenes the pre-pass gave up on (scene s: gate.word + s * gate.stride, raised = its tag
// equals gate.tag).
int launch_fps_presort(int b, int n, const float *dataset, const float *temp, float *work, long long stride, hipStream_t st,
                       PresortGate *gate) {
    int K = PS_MAXK;
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
    // a launch being CAPTURED into a graph declines as well: its flags are told apart by a per-launch epoch, and a replayed
    // launch would meet its own flags of the previous replay (the kernel that sorts for itself has no such state)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return -1;
    unsigned long long *pool = presort_flag_pool(dev);
    if (!pool) return -1;   // (sps_init() has not run for this device: the caller takes the kernel that sorts for itself)
    static std::atomic<unsigned> counter{0};
    unsigned ticket = counter.fetch_add(1, std::memory_order_relaxed) + 1u;
    if ((ticket << 1) == 0u) ticket = counter.fetch_add(1, std::memory_order_relaxed) + 1u;   // tag 0 = the zeroed pool
    unsigned long long *flags = pool + (size_t)(ticket % PS_SLOTS) * PS_SCENES * 2 * PS_MAXK;
    hipLaunchKernelGGL(fps_presort_kernel, dim3(8 * K * divup(b, 8)), dim3(PF_THREADS), 0, st, b, K, n, bs, l2, rb, npad, stride,
                       dataset, temp, work, fps_cluster_spread(), flags, ticket << 1, pc_spin_limit());
    // (SPS_FPS_PRESORT_GATE=0, DIAGNOSTIC / A-B timing only: no gate, no follow-up launch -- a pre-pass that gave up would then
    //  leave its scenes unsampled)
    static const bool gate_on = [] { const char *e = getenv("SPS_FPS_PRESORT_GATE"); return !(e && *e == '0'); }();
    if (gate) *gate = gate_on ? PresortGate{flags + PS_MAXK, 2 * PS_MAXK, ticket << 1} : PresortGate{nullptr, 0, 0u};
    return check_launch("fps_presort_kernel");
}

}  // namespace sps

// One-time set-up of the library's per-device state on `device` (made current for the call): today the flag pool of the FPS
// sorting pre-pass.  The ONLY entry point that allocates device memory or synchronises (`stream` is synchronised); idempotent
// and thread-safe.  A process that never calls it still gets correct results everywhere -- the FPS launchers then use the
// kernel that sorts for itself (~25 us slower at 8 x 16 384 points) -- so it must be called BEFORE a stream capture that is
// meant to record the fast path.  No reference counterpart (the reference's launchers own no device state).
extern "C" int sps_init(int device, sps_stream_t stream) {
    using namespace sps;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(SPS_ERR_INVALID, "sps_init: no device %d", device);
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) prev = device;
    if (prev != device && hipSetDevice(device) != hipSuccess) return fail(SPS_ERR_LAUNCH, "sps_init: cannot select device %d", device);
    const int rc = presort_pool_create(device, as_stream(stream));
    if (prev != device) (void)hipSetDevice(prev);
    return rc;
}

// 1 once sps_init(device, ...) has succeeded in this process
extern "C" int sps_is_initialized(int device) { return sps::presort_pool_exists(device) ? 1 : 0; }

// DIAGNOSTIC: the spin bound of every cross-workgroup poll of the FPS kernels (the split sort of fps_presort.hip /
// fps_pruned_cluster.hip and the clustered kernel's record exchange); 0 restores the default, 0xFFFFFFFF makes every poll
// give up without looking -- how the tests drive the give-up -> redo path.  Returns the previous bound.
extern "C" unsigned sps_debug_set_exchange_spins(unsigned spins) {
    return sps::g_spin_limit.exchange(spins ? spins : sps::PC_SPIN_LIMIT, std::memory_order_relaxed);
}
