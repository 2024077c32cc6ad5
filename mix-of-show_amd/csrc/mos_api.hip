// mos_api.hip — error plumbing and version of libmos_hip.so (see include/mos_hip.h).
#include <cstdarg>
#include <cstdio>
#include "mos_common.h"

namespace {
thread_local char g_err[512] = "ok";
}

int mos_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mos_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mos_set_error(MOS_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return MOS_OK;
}

extern "C" {
int mos_version(void) { return 100; }  // 0.1.0
const char* mos_last_error_string(void) { return g_err; }
}
