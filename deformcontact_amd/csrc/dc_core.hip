// dc_core.hip -- version + thread-local error string of libdeformcontact_hip.so.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "dc_common.h"

namespace dc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_trace_on = 0;
static std::mutex g_trace_mu;
static std::map<std::string, int64_t> g_trace_counts;
void trace_kernel_slow(const char *name) {
    std::lock_guard<std::mutex> lock(g_trace_mu);
    // "(k_foo<A, B>)" as the launch macro sees it -> "k_foo<A, B>"
    std::string n(name);
    while (!n.empty() && (n.front() == '(' || n.front() == ' ')) n.erase(n.begin());
    while (!n.empty() && (n.back() == ')' || n.back() == ' ')) n.pop_back();
    if (n.rfind("dc::", 0) == 0) n.erase(0, 4);
    ++g_trace_counts[n];
}
}  // namespace dc

// Launch log: dc_kernel_trace(1) clears it and starts counting every kernel launch of the library by kernel name (as
// written at the launch site, e.g. "k_fwd_h2w<true, false>"; template PARAMETERS of the host function stay
// symbolic, e.g. "k_hop_chain_gcn<STEPS>"), dc_kernel_trace(0) stops; dc_kernel_trace_dump writes "name count\n"
// lines (NUL-terminated, truncated to cap) and returns the bytes the full text needs.
extern "C" void dc_kernel_trace(int on) {
    std::lock_guard<std::mutex> lock(dc::g_trace_mu);
    if (on) dc::g_trace_counts.clear();
    dc::g_trace_on = on ? 1 : 0;
}
extern "C" int64_t dc_kernel_trace_dump(char *buf, int64_t cap) {
    std::lock_guard<std::mutex> lock(dc::g_trace_mu);
    std::string out;
    for (const auto &kv : dc::g_trace_counts) out += kv.first + " " + std::to_string(kv.second) + "\n";
    if (buf && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

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
