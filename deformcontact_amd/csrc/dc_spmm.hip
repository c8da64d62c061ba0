// dc_spmm.hip -- the hop: fused gather - scale - segment-sum on gfx950.
//
// Replaces PyG's MessagePassing.propagate (index_select -> mul -> scatter_add_,
// two [E,F] temporaries) called 3x per TAGConv.forward from
// /root/reference/models/model.py:71,77.  HBM/L2-bound fp32 row gather:
//
//   y[i,:] = addend[i,:] + sum_{p in [ptr[i],ptr[i+1])} w[p] * x[other[p],:]
//
// Design (MI355X):
//   * edges pre-sorted by destination (dc_csr.hip) -> no atomics, one plain
//     coalesced store per output row, deterministic summation order;
//   * F = 256 fp32 = 1 KiB = one global_load_dwordx4 per 64-lane wave: one wave
//     owns one destination row, its row index / segment bounds / neighbour ids
//     / weights are wave-uniform (SGPRs, scalar loads), U neighbour rows are in
//     flight per wave before the first is consumed;
//   * narrow rows (F = 21/25/32...) pack 64/L rows into a wave (L lanes each);
//   * logical blocks are handed to XCDs in contiguous chunks (xcd_remap) so the
//     neighbour rows of a mesh - block-diagonal batch - are served by ONE 4 MiB
//     L2 instead of being re-fetched by all eight;
//   * multiply and add are rounded separately (no FMA contraction) and summed
//     in p order: bit-identical to a serial scatter_add_ over the stable order.
#include "dc_common.h"

#pragma clang fp contract(off)

namespace dc {

template <int VEC> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<4> { using T = float4; };

__device__ __forceinline__ float vzero(float) { return 0.0f; }
__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ void vaxpy(float &acc, float w, float v) {
    const float m = w * v;
    acc = acc + m;
}
__device__ __forceinline__ void vaxpy(float4 &acc, float w, const float4 &v) {
    const float mx = w * v.x, my = w * v.y, mz = w * v.z, mw = w * v.w;
    acc.x = acc.x + mx;
    acc.y = acc.y + my;
    acc.z = acc.z + mz;
    acc.w = acc.w + mw;
}

// ---- one wave per destination row (F >= 64*VEC/2 ... up to any F) ----------
template <int VEC, int U>
__global__ void __launch_bounds__(256)
k_spmm_wave(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
            const float *__restrict__ w, const float *__restrict__ x, int64_t ldx,
            const float *addend, int64_t ldadd, float *y, int64_t ldy,
            int64_t N, int F) {
    using V = typename Vec<VEC>::T;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    // wave-uniform row: force into an SGPR so bounds / ids / weights use scalar loads
    const int64_t row =
        __builtin_amdgcn_readfirstlane((int)(lb * 4u + (threadIdx.x >> 6)));
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int beg = ptr[row], end = ptr[row + 1];

    for (int c = lane * VEC; c < F; c += kWave * VEC) {
        V acc = addend ? *reinterpret_cast<const V *>(addend + row * ldadd + c) : vzero(V{});
        for (int p = beg; p < end; p += U) {
            const int n = end - p;   // wave-uniform
            int s[U];
            float ww[U];
            V v[U];
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) {
                    s[j] = other[p + j];
                    ww[j] = w ? w[p + j] : 1.0f;
                }
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) v[j] = *reinterpret_cast<const V *>(x + (int64_t)s[j] * ldx + c);
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) vaxpy(acc, ww[j], v[j]);
        }
        *reinterpret_cast<V *>(y + row * ldy + c) = acc;
    }
}

// ---- L lanes per row, 64/L rows per wave (narrow feature rows) -------------
template <int VEC, int L, int U>
__global__ void __launch_bounds__(256)
k_spmm_sub(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
           const float *__restrict__ w, const float *__restrict__ x, int64_t ldx,
           const float *addend, int64_t ldadd, float *y, int64_t ldy,
           int64_t N, int F) {
    using V = typename Vec<VEC>::T;
    constexpr int kRows = 256 / L;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = (int64_t)lb * kRows + threadIdx.x / L;
    if (row >= N) return;
    const int sub = threadIdx.x % L;
    const int beg = ptr[row], end = ptr[row + 1];

    for (int c = sub * VEC; c < F; c += L * VEC) {
        V acc = addend ? *reinterpret_cast<const V *>(addend + row * ldadd + c) : vzero(V{});
        for (int p = beg; p < end; p += U) {
            int s[U];
            float ww[U];
            V v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool ok = p + j < end;
                s[j] = ok ? other[p + j] : 0;
                ww[j] = ok ? (w ? w[p + j] : 1.0f) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool ok = p + j < end;
                v[j] = ok ? *reinterpret_cast<const V *>(x + (int64_t)s[j] * ldx + c) : vzero(V{});
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (p + j < end) vaxpy(acc, ww[j], v[j]);
        }
        *reinterpret_cast<V *>(y + row * ldy + c) = acc;
    }
}

template <int VEC, int L, int U>
static void launch_sub(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                       int64_t ldx, const float *addend, int64_t ldadd, float *y, int64_t ldy,
                       int64_t N, int F, hipStream_t stream) {
    constexpr int kRows = 256 / L;
    const unsigned grid = (unsigned)((N + kRows - 1) / kRows);
    hipLaunchKernelGGL((k_spmm_sub<VEC, L, U>), dim3(grid), dim3(256), 0, stream, ptr, other, w, x,
                       ldx, addend, ldadd, y, ldy, N, F);
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace dc

using namespace dc;

extern "C" int dc_spmm_f32(const int32_t *ptr, const int32_t *other, const float *w,
                           const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                           float *y, int64_t ldy, int64_t N, int64_t F, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 0, "dc_spmm_f32: negative size N=%lld F=%lld", (long long)N,
               (long long)F);
    if (N == 0 || F == 0) return DC_OK;
    DC_REQUIRE(ptr && x && y, "dc_spmm_f32: null ptr/x/y");
    DC_REQUIRE(N < (int64_t)INT32_MAX / 4 && F < (1 << 24), "dc_spmm_f32: size out of range");
    DC_REQUIRE(ldx >= F && ldy >= F && (!addend || ldadd >= F),
               "dc_spmm_f32: leading dimension smaller than F");
    DC_REQUIRE(x != y, "dc_spmm_f32: y must not alias x");

    const bool vec4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) &&
                      aligned16(y) && (!addend || ((ldadd % 4 == 0) && aligned16(addend)));
    const int Fi = (int)F;
    if (vec4) {
        const int64_t v = F / 4;
        if (v > 32) {
            const unsigned grid = (unsigned)((N + 3) / 4);
            hipLaunchKernelGGL((k_spmm_wave<4, 8>), dim3(grid), dim3(256), 0, stream, ptr, other,
                               w, x, ldx, addend, ldadd, y, ldy, N, Fi);
        } else if (v > 16)
            launch_sub<4, 32, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
        else if (v > 8)
            launch_sub<4, 16, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
        else if (v > 4)
            launch_sub<4, 8, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
        else
            launch_sub<4, 4, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
    } else {
        if (F > 32) {
            const unsigned grid = (unsigned)((N + 3) / 4);
            hipLaunchKernelGGL((k_spmm_wave<1, 8>), dim3(grid), dim3(256), 0, stream, ptr, other,
                               w, x, ldx, addend, ldadd, y, ldy, N, Fi);
        } else if (F > 16)
            launch_sub<1, 32, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
        else if (F > 8)
            launch_sub<1, 16, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
        else
            launch_sub<1, 8, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, stream);
    }
    return check_launch("dc_spmm_f32");
}
