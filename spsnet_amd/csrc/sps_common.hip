// sps_common.hip -- error reporting and library identity of libspsnet_sa.
#include "sps_common.h"

#include <stdarg.h>
#include <stdio.h>

namespace sps {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SPS_OK;
}

}  // namespace sps

extern "C" int sps_abi_version(void) { return SPS_ABI_VERSION; }
extern "C" const char *sps_last_error(void) { return sps::g_err; }

// ---- diagnostics and stream plumbing for the CU-fenced schedule (sa_stack.pipelined_bench) --------------------------
namespace sps {
// One record per workgroup: {HW_REG_XCC_ID, HW_REG_HW_ID} as the hardware reports them (gfx950: XCC_ID[3:0] = XCD,
// HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]); spins `spin` s_sleep steps so that the grid has to spread.
__global__ void where_kernel(unsigned *out, int spin) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID, bits 31:0
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);
}
}  // namespace sps

extern "C" int sps_debug_where(int blocks, int threads, int spin, unsigned *out, sps_stream_t stream) {
    using namespace sps;
    if (blocks <= 0 || threads <= 0 || threads > 1024 || !out) return fail(SPS_ERR_INVALID, "debug_where: bad arguments");
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(threads), 0, as_stream(stream), out, spin);
    return check_launch("where_kernel");
}

// A HIP stream whose kernels may only run on the compute units whose bits are set in mask[0 .. words) (bit i of word w =
// CU 32 w + i in the runtime's numbering; tools/cumask_probe.py prints which physical CUs that is).  The caller owns the
// stream and releases it with sps_stream_destroy.  Used to fence the serial FPS chain of a pass onto CUs of its own.
extern "C" int sps_stream_create_cu_mask(int words, const unsigned *mask, sps_stream_t *stream) {
    using namespace sps;
    if (words <= 0 || !mask || !stream) return fail(SPS_ERR_INVALID, "stream_create_cu_mask: bad arguments");
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    *stream = reinterpret_cast<sps_stream_t>(st);
    return SPS_OK;
}

extern "C" int sps_stream_destroy(sps_stream_t stream) {
    using namespace sps;
    const hipError_t e = hipStreamDestroy(as_stream(stream));
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "hipStreamDestroy: %s", hipGetErrorString(e));
    return SPS_OK;
}

// ---- do two streams really run side by side? ------------------------------------------------------------------------
// HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and kernels of two streams that
// share one run strictly one after the other -- which stream lands on which queue depends on everything else in the process
// that created streams (RCCL alone is enough to put the FPS producer and its consumers on one queue: the pass then takes
// FPS + everything else instead of their maximum).  The probe: a one-lane kernel on `a` waits (bounded by wall clock) for a
// word that a kernel on `b`, launched right behind it, sets.  On different queues it sees the word within microseconds;
// on one queue the setter cannot start before the waiter has given up.
namespace sps {
__global__ void hq_wait_kernel(int *word, int *seen, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    int v = 0;
    do {
        v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0) __builtin_amdgcn_s_sleep(16);
    } while (v == 0 && wall_clock64() - t0 < ticks);
    *seen = v;
}
__global__ void hq_set_kernel(int *word) { __hip_atomic_store(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace sps

// scratch: two device ints.  *concurrent = 1 when a kernel on `b` ran while a kernel on `a` was still running.  Synchronises
// both streams (a set-up call, not for the data path); at most ~limit_us of GPU time when the streams do share a queue.
extern "C" int sps_streams_run_concurrently(sps_stream_t a, sps_stream_t b, int *scratch, int limit_us, int *concurrent) {
    using namespace sps;
    if (!scratch || !concurrent || limit_us <= 0) return fail(SPS_ERR_INVALID, "streams_run_concurrently: bad arguments");
    hipStream_t sa = as_stream(a), sb = as_stream(b);
    *concurrent = 0;
    if (sa == sb) return SPS_OK;
    // a stream gets its hardware queue at its FIRST launch (milliseconds the first time a queue is created): one throw-away
    // launch on each, so that the timed pair below measures queue sharing and nothing else
    hipLaunchKernelGGL(hq_set_kernel, dim3(1), dim3(1), 0, sb, scratch);
    hipError_t e = hipStreamSynchronize(sb);
    if (e == hipSuccess) e = hipMemsetAsync(scratch, 0, 2 * sizeof(int), sa);
    if (e == hipSuccess) e = hipStreamSynchronize(sa);
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "streams_run_concurrently: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(hq_wait_kernel, dim3(1), dim3(1), 0, sa, scratch, scratch + 1, 100ull * (unsigned long long)limit_us);  // 100 MHz
    hipLaunchKernelGGL(hq_set_kernel, dim3(1), dim3(1), 0, sb, scratch);
    const int rc = check_launch("hq_wait_kernel / hq_set_kernel");
    if (rc != SPS_OK) return rc;
    int seen = 0;
    e = hipStreamSynchronize(sa);
    if (e == hipSuccess) e = hipStreamSynchronize(sb);
    if (e == hipSuccess) e = hipMemcpy(&seen, scratch + 1, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(SPS_ERR_LAUNCH, "streams_run_concurrently: %s", hipGetErrorString(e));
    *concurrent = seen != 0;
    return SPS_OK;
}
