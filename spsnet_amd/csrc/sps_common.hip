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
