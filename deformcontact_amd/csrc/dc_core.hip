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

extern "C" int dc_version(void) { return 100; }   // 0.1.0
extern "C" const char *dc_last_error(void) { return dc::g_err; }
