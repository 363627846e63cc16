// Library-wide plumbing: per-thread error string, launch check, version.
#include "hd_common.cuh"
#include "../../include/hypdiff.h"
#include <string.h>

namespace hd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return HD_EHIP;
    }
    return HD_OK;
}

}  // namespace hd

extern "C" int hd_version(void) { return 100; }

extern "C" size_t hd_last_error(char* buf, size_t cap) {
    if (!buf || cap == 0) return strlen(hd::g_err);
    strncpy(buf, hd::g_err, cap - 1);
    buf[cap - 1] = '\0';
    return strlen(buf);
}
