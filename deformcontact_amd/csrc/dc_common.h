// dc_common.h -- shared helpers for the gfx950 kernels (internal; the public
// surface is include/deformcontact.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deformcontact.h"

namespace dc {

constexpr int kWave = 64;      // CDNA4 wavefront
constexpr int kXcds = 8;       // MI355X: 8 XCDs, each with a private 4 MiB L2

void set_error(const char *fmt, ...);

// launch log (dc_kernel_trace / dc_kernel_trace_dump, dc_core.hip): off unless a test or tool switches it on
extern int g_trace_on;
void trace_kernel_slow(const char *name);
inline void trace_kernel(const char *name) {
    if (__builtin_expect(g_trace_on, 0)) trace_kernel_slow(name);
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DC_ELAUNCH;
    }
    return DC_OK;
}

// XCD-aware block remap (bijective for any grid size).  Workgroups are dealt
// round-robin over the 8 XCDs, so hardware blocks b and b+8 share an L2.  The
// remap hands XCD k the k-th CONTIGUOUS chunk of logical blocks: rows of one
// mesh (block-diagonal batch => its neighbours too) are then gathered through
// one L2 instead of all eight.  Speed only; any placement is correct.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk / kXcds, r = nblk % kXcds;
    const unsigned xcd = bid % kXcds, idx = bid / kXcds;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// gcn_norm's deg.pow(-0.5) with inf -> 0 (PyG gcn_conv.py: gcn_norm): ONE definition, so that every kernel that
// forms a weight dis[source] * dis[destination] (dc_csr.hip writes them, dc_hopchain.hip re-forms them) gets the same bits
__device__ __forceinline__ float inv_sqrt_count(int d) {
    return d > 0 ? 1.0f / sqrtf((float)d) : 0.0f;
}

}  // namespace dc

// every kernel launch of the library: the kernel expression as written (template arguments included) goes to the
// launch log when it is on
#define DC_LAUNCH(kernel, ...)                     \
    do {                                           \
        dc::trace_kernel(#kernel);                 \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);   \
    } while (0)

#define DC_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            dc::set_error(__VA_ARGS__);  \
            return DC_EINVAL;            \
        }                                \
    } while (0)
