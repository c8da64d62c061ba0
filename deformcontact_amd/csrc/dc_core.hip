// dc_core.hip -- version + thread-local error string of libdeformcontact_hip.so.
#include <stdarg.h>

#include "dc_common.h"

namespace dc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dc

extern "C" int dc_version(void) { return 200; }   // 0.2.0
extern "C" const char *dc_last_error(void) { return dc::g_err; }

extern "C" int64_t dc_stream_capture_id(dc_stream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return st == hipStreamCaptureStatusActive ? (int64_t)id : 0;
}
