// host_util.h - error reporting shared by the extern "C" entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

char* sw_errbuf();                       // thread-local message buffer (capi.hip)
#define SW_ERRBUF_LEN 512

static inline int sw_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(sw_errbuf(), SW_ERRBUF_LEN, fmt, ap);
    va_end(ap);
    return code;
}

static inline int sw_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    snprintf(sw_errbuf(), SW_ERRBUF_LEN, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
