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
