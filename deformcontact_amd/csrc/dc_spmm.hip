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
#include <stdlib.h>

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

__device__ __forceinline__ void vadd(float &a, float b) { a = a + b; }
__device__ __forceinline__ void vadd(float4 &a, const float4 &b) { a.x = a.x + b.x, a.y = a.y + b.y, a.z = a.z + b.z, a.w = a.w + b.w; }
__device__ __forceinline__ void vrelu(float &a) { a = fmaxf(a, 0.f); }
__device__ __forceinline__ void vrelu(float4 &a) { a.x = fmaxf(a.x, 0.f), a.y = fmaxf(a.y, 0.f), a.z = fmaxf(a.z, 0.f), a.w = fmaxf(a.w, 0.f); }

// ---- one wave per destination row (F >= 64*VEC/2 ... up to any F) ----------
__device__ __forceinline__ float vabsmax(float v) { return fabsf(v); }
__device__ __forceinline__ float vabsmax(const float4 &v) {
    return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// RM: also write rowmax[row] = max |y[row,:]| (bit 0 of rm_mode: include |x[row,:]|, the
// wave's OWN input row; bit 1: keep the larger of the new value and what rowmax[row] holds) -
// the row scales of the fp16x2 dense block (dc_dense_split.hip) at no extra pass.
// SGPR budget: a 256-thread block is admitted 8 per CU only up to 80 SGPRs (82-96 -> 7,
// MI355X_MICROARCH.md "Residency"); the row-maxima variant needed 85 and ran with one block per CU
// fewer - the whole +1.3 us it cost over the plain hop (r02 hop_exp).  The attribute caps the
// allocation (the few extra scalars live in VGPR lanes).
// max over the 64 lanes of a wave of non-negative values (every lane gets it; DPP + readlane, no LDS)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    const int i = __float_as_int(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = dpp_max<0xB1>(v);                    // quad_perm [1,0,3,2]
    v = dpp_max<0x4E>(v);                    // quad_perm [2,3,0,1]
    v = dpp_max<0x141>(v);                   // row_half_mirror
    v = dpp_max<0x140>(v);                   // row_mirror: every lane = max of its 16-lane row
    const int i = __float_as_int(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(i, 0)), b = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(i, 32)), d = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// EPI: y = act(sum + bias) - the "+ bias" and the ReLU that follow the aggregation of a GCNConv / GATConv layer
// (PyG gcn_conv.py / gat_conv.py: out = propagate(...); out = out + bias; models/model.py:71,77: relu) in the row's
// epilogue instead of two elementwise passes; same values (the sum is complete before the bias is added)
template <int VEC, int U, bool RM = false, bool EPI = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
k_spmm_wave(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
            const float *__restrict__ w, const float *__restrict__ x, int64_t ldx,
            const float *addend, int64_t ldadd, float *y, int64_t ldy,
            int64_t N, int F, float *rowmax = nullptr, int rm_mode = 0, int src_off = 0,
            const float *__restrict__ bias = nullptr, int relu = 0) {
    // src_off: the adjacency is a ROW WINDOW of a larger (merged, block-diagonal) one - `ptr` points at the
    // window's first row, neighbour ids are ids of the larger node space and x / y / addend / rowmax hold only
    // the window's rows: neighbour row = other[p] - src_off
    using V = typename Vec<VEC>::T;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    // wave-uniform row: force into an SGPR so bounds / ids / weights use scalar loads
    const int64_t row =
        __builtin_amdgcn_readfirstlane((int)(lb * 4u + (threadIdx.x >> 6)));
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int beg = ptr[row], end = ptr[row + 1];
    // the stored maximum this launch joins with (mode bit 1) is fetched NOW, with the segment
    // bounds, not after the row has been reduced: at the end of the wave it was one more exposed
    // memory latency per row (+1.3 us on the 17 us launch, r02 hop_exp)
    float rmax = 0.f;
    if (RM && (rm_mode & 2)) rmax = rowmax[row];

    for (int c = lane * VEC; c < F; c += kWave * VEC) {
        V acc = addend ? *reinterpret_cast<const V *>(addend + row * ldadd + c) : vzero(V{});
        V self = vzero(V{});
        if (RM && (rm_mode & 1)) self = *reinterpret_cast<const V *>(x + row * ldx + c);
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
                if (j < n) v[j] = *reinterpret_cast<const V *>(x + (int64_t)(s[j] - src_off) * ldx + c);
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) vaxpy(acc, ww[j], v[j]);
        }
        if (EPI) {
            if (bias) vadd(acc, *reinterpret_cast<const V *>(bias + c));
            if (relu) vrelu(acc);
        }
        *reinterpret_cast<V *>(y + row * ldy + c) = acc;
        if (RM) rmax = fmaxf(rmax, fmaxf(vabsmax(acc), vabsmax(self)));
    }
    if (RM) {
        // wave maximum without the LDS pipe: four DPP steps give every lane the maximum of its 16-lane row, the
        // four row values are read as scalars.  (__shfl_xor compiles to six dependent ds_bpermute_b32, each
        // behind an lgkmcnt(0) wait: ~0.5 us at the end of every row's wave, +1.3 us per launch - r02 hop_exp.)
        rmax = wave_max_nonneg(rmax);
        if (lane == 0) rowmax[row] = rmax;
    }
}


// Broadcast of the value held by lane J of each 16-lane DPP row to all lanes of that row: a VALU move with
// the row_share modifier - no LDS round trip (ds_bpermute, which __shfl compiles to, has ~100 cycles of
// latency behind an lgkmcnt wait, in front of every chunk's gathers).
// (`old` = the source itself: every lane of the row is written, so no separate v_mov to initialise the destination)
template <int J>
__device__ __forceinline__ int row_share_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, false);
}
template <int J>
__device__ __forceinline__ float row_share_f(float v) {
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x150 + J, 0xF, 0xF, false));
}
// ids / weights of a chunk of 8 edges for a lane group of L >= 8 lanes (sub = lane index inside the group):
// L >= 16: the first 8 lanes of EVERY 16-lane row load them (one coalesced 32-byte load each), row_share
// hands them round; L == 8: the group's 8 lanes load, __shfl hands round
template <int L>
__device__ __forceinline__ void chunk_ids(const int32_t *__restrict__ other, const float *__restrict__ w,
                                          int p, int end, int sub, int (&s)[8], float (&ww)[8]) {
    if (L >= 16) {
        const int q = sub & 15;
        const bool mine = q < 8 && p + q < end;
        const int my_s = mine ? other[p + q] : 0;
        const float my_w = mine ? (w ? w[p + q] : 1.0f) : 0.0f;
        s[0] = row_share_i<0>(my_s), s[1] = row_share_i<1>(my_s), s[2] = row_share_i<2>(my_s);
        s[3] = row_share_i<3>(my_s), s[4] = row_share_i<4>(my_s), s[5] = row_share_i<5>(my_s);
        s[6] = row_share_i<6>(my_s), s[7] = row_share_i<7>(my_s);
        ww[0] = row_share_f<0>(my_w), ww[1] = row_share_f<1>(my_w), ww[2] = row_share_f<2>(my_w);
        ww[3] = row_share_f<3>(my_w), ww[4] = row_share_f<4>(my_w), ww[5] = row_share_f<5>(my_w);
        ww[6] = row_share_f<6>(my_w), ww[7] = row_share_f<7>(my_w);
    } else {
        const bool mine = sub < 8 && p + sub < end;
        const int my_s = mine ? other[p + sub] : 0;
        const float my_w = mine ? (w ? w[p + sub] : 1.0f) : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s[j] = __shfl(my_s, j, L);
            ww[j] = __shfl(my_w, j, L);
        }
    }
}

// ---- L lanes per row, 64/L rows per wave (narrow feature rows) -------------
template <int VEC, int L, int U>
__global__ void __launch_bounds__(256)
k_spmm_sub(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
           const float *__restrict__ w, const float *__restrict__ x, int64_t ldx,
           const float *addend, int64_t ldadd, float *y, int64_t ldy,
           int64_t N, int F, int src_off, float *self_dst = nullptr, int pad_beg = 0, int pad_end = 0) {
    using V = typename Vec<VEC>::T;
    constexpr int kRows = 256 / L;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = (int64_t)lb * kRows + threadIdx.x / L;
    if (row >= N) return;
    const int sub = threadIdx.x % L;
    const int beg = ptr[row], end = ptr[row + 1];
    if (self_dst) {
        // fused dc_tag_pack_input (first hop of a layer's own input): the row itself goes into column block 0 of the
        // slab row (self_dst has y's leading dimension), the slab's K padding [pad_beg, pad_end) is zeroed
        for (int c = sub; c < F; c += L) self_dst[row * ldy + c] = x[row * ldx + c];
        for (int c = pad_beg + sub; c < pad_end; c += L) self_dst[row * ldy + c] = 0.f;
    }

    // the column loop is uniform across the row's lane group (lanes past F idle inside it): the
    // shuffles below need their source lanes - the first U of the group - in the loop
    for (int c0 = 0; c0 < F; c0 += L * VEC) {
        const int c = c0 + sub * VEC;
        const bool act = c < F;
        V acc = (addend && act) ? *reinterpret_cast<const V *>(addend + row * ldadd + c) : vzero(V{});
        for (int p = beg; p < end; p += U) {
            int s[U];
            float ww[U];
            V v[U];
            if (L >= U && U == 8) {
                // ids / weights of the chunk: one coalesced load each, handed round inside the lane
                // group (2 vector-memory instructions per chunk instead of 2 U)
                chunk_ids<L>(other, w, p, end, sub, s, ww);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const bool ok = p + j < end;
                    s[j] = ok ? other[p + j] : 0;
                    ww[j] = ok ? (w ? w[p + j] : 1.0f) : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool ok = act && p + j < end;
                v[j] = ok ? *reinterpret_cast<const V *>(x + (int64_t)(s[j] - src_off) * ldx + c) : vzero(V{});
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (p + j < end) vaxpy(acc, ww[j], v[j]);
        }
        if (act) *reinterpret_cast<V *>(y + row * ldy + c) = acc;
    }
}

template <int VEC, int L, int U>
static void launch_sub(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                       int64_t ldx, const float *addend, int64_t ldadd, float *y, int64_t ldy,
                       int64_t N, int F, int src_off, hipStream_t stream, float *self_dst = nullptr,
                       int pad_beg = 0, int pad_end = 0) {
    constexpr int kRows = 256 / L;
    const unsigned grid = (unsigned)((N + kRows - 1) / kRows);
    DC_LAUNCH((k_spmm_sub<VEC, L, U>), dim3(grid), dim3(256), 0, stream, ptr, other, w, x,
                       ldx, addend, ldadd, y, ldy, N, F, src_off, self_dst, pad_beg, pad_end);
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// ---- bf16 feature rows, fp32 accumulation -------------------------
// x is stored as bf16 (half the gather bytes of the fp32 hop); products and the running sum are
// fp32 exactly as in the fp32 kernels (bf16 -> fp32 is exact, multiply and add rounded
// separately, p order), and the result is stored as fp32 or rounded once (nearest-even) to bf16.
// L lanes share one row, 8 columns (one 16-byte load) per lane and step: F = 256 is 32 lanes,
// two destination rows per wave.
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct Bf16x8 { uint4 q; };
__device__ __forceinline__ void unpack8(const uint4 &q, float (&v)[8]) {
    const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(d[i] << 16);
        v[2 * i + 1] = __uint_as_float(d[i] & 0xffff0000u);
    }
}

template <int L, int U, bool OUT_F32>
__global__ void __launch_bounds__(256)
k_spmm_bf16x8(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
              const float *__restrict__ w, const uint16_t *__restrict__ x, int64_t ldx,
              const void *addend, int64_t ldadd, void *y, int64_t ldy, int64_t N, int F) {
    constexpr int kRows = 256 / L;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = (int64_t)lb * kRows + threadIdx.x / L;
    if (row >= N) return;
    const int sub = threadIdx.x % L;
    const int beg = ptr[row], end = ptr[row + 1];

    for (int c0 = 0; c0 < F; c0 += L * 8) {      // uniform across the lane group (see k_spmm_sub)
        const int c = c0 + sub * 8;
        const bool act = c < F;
        float acc[8];
        if (addend && act) {
            if (OUT_F32) {
                const float4 *a = reinterpret_cast<const float4 *>((const float *)addend + row * ldadd + c);
                const float4 a0 = a[0], a1 = a[1];
                acc[0] = a0.x; acc[1] = a0.y; acc[2] = a0.z; acc[3] = a0.w;
                acc[4] = a1.x; acc[5] = a1.y; acc[6] = a1.z; acc[7] = a1.w;
            } else {
                unpack8(*reinterpret_cast<const uint4 *>((const uint16_t *)addend + row * ldadd + c), acc);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        }
        for (int p = beg; p < end; p += U) {
            int s[U];
            float ww[U];
            uint4 q[U];
            if (L >= U && U == 8) {
                // the 8 neighbour ids / weights of the chunk: ONE coalesced load each, handed round inside
                // the lane group - two vector-memory instructions per chunk instead of 16 (with bf16 rows a
                // wave serves two rows, so every instruction saved counts twice: 1.5 -> 0.63 per edge)
                chunk_ids<L>(other, w, p, end, sub, s, ww);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const bool ok = p + j < end;
                    s[j] = ok ? other[p + j] : 0;
                    ww[j] = ok ? (w ? w[p + j] : 1.0f) : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
                q[j] = (act && p + j < end) ? *reinterpret_cast<const uint4 *>(x + (int64_t)s[j] * ldx + c)
                                            : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (p + j < end) {
                    float v[8];
                    unpack8(q[j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float m = ww[j] * v[i];
                        acc[i] = acc[i] + m;
                    }
                }
        }
        if (!act) continue;
        if (OUT_F32) {
            float4 *o = reinterpret_cast<float4 *>((float *)y + row * ldy + c);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
            uint4 o;
            o.x = f32_to_bf16_rne(acc[0]) | ((uint32_t)f32_to_bf16_rne(acc[1]) << 16);
            o.y = f32_to_bf16_rne(acc[2]) | ((uint32_t)f32_to_bf16_rne(acc[3]) << 16);
            o.z = f32_to_bf16_rne(acc[4]) | ((uint32_t)f32_to_bf16_rne(acc[5]) << 16);
            o.w = f32_to_bf16_rne(acc[6]) | ((uint32_t)f32_to_bf16_rne(acc[7]) << 16);
            *reinterpret_cast<uint4 *>((uint16_t *)y + row * ldy + c) = o;
        }
    }
}


// one wave per PAIR of destination rows, 32 lanes x 8 bf16 (16 bytes) each: every gather
// instruction still moves 1 KiB, and the bounds / neighbour ids / weights of BOTH rows are
// wave-uniform scalar loads (the half-wave picks its own with a select), so a pair of edges costs
// one vector-memory instruction instead of the three of k_spmm_bf16x8.
template <int U, bool OUT_F32>
__global__ void __launch_bounds__(256)
k_spmm_bf16_pair(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
                 const float *__restrict__ w, const uint16_t *__restrict__ x, int64_t ldx,
                 const void *addend, int64_t ldadd, void *y, int64_t ldy, int64_t N, int F) {
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = 2 * (int64_t)__builtin_amdgcn_readfirstlane((int)(lb * 4u + (threadIdx.x >> 6)));
    if (row0 >= N) return;
    const bool has1 = row0 + 1 < N;
    const int lane = threadIdx.x & 63;
    const bool hi = lane >= 32;
    const int beg0 = ptr[row0], beg1 = ptr[row0 + 1];
    const int len0 = beg1 - beg0, len1 = has1 ? ptr[row0 + 2] - beg1 : 0;
    const int lmax = len0 > len1 ? len0 : len1;
    const int64_t row = row0 + (hi ? 1 : 0);
    const bool live = !hi || has1;

    for (int c = (lane & 31) * 8; c < F; c += 256) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        if (addend && live) {
            if (OUT_F32) {
                const float4 *a = reinterpret_cast<const float4 *>((const float *)addend + row * ldadd + c);
                const float4 a0 = a[0], a1 = a[1];
                acc[0] = a0.x; acc[1] = a0.y; acc[2] = a0.z; acc[3] = a0.w;
                acc[4] = a1.x; acc[5] = a1.y; acc[6] = a1.z; acc[7] = a1.w;
            } else {
                unpack8(*reinterpret_cast<const uint4 *>((const uint16_t *)addend + row * ldadd + c), acc);
            }
        }
        for (int t = 0; t < lmax; t += U) {
            int s[U];
            float ww[U];
            bool ok[U];
            uint4 q[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                int sa = 0, sb = 0;
                float wa = 0.0f, wb = 0.0f;
                if (t + j < len0) {              // wave-uniform: scalar loads
                    sa = other[beg0 + t + j];
                    wa = w ? w[beg0 + t + j] : 1.0f;
                }
                if (t + j < len1) {
                    sb = other[beg1 + t + j];
                    wb = w ? w[beg1 + t + j] : 1.0f;
                }
                s[j] = hi ? sb : sa;
                ww[j] = hi ? wb : wa;
                ok[j] = hi ? (t + j < len1) : (t + j < len0);
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (t + j < lmax)
                    q[j] = ok[j] ? *reinterpret_cast<const uint4 *>(x + (int64_t)s[j] * ldx + c)
                                 : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (t + j < lmax && ok[j]) {
                    float v[8];
                    unpack8(q[j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float m = ww[j] * v[i];
                        acc[i] = acc[i] + m;
                    }
                }
        }
        if (!live) continue;
        if (OUT_F32) {
            float4 *o = reinterpret_cast<float4 *>((float *)y + row * ldy + c);
            o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
            uint4 o;
            o.x = f32_to_bf16_rne(acc[0]) | ((uint32_t)f32_to_bf16_rne(acc[1]) << 16);
            o.y = f32_to_bf16_rne(acc[2]) | ((uint32_t)f32_to_bf16_rne(acc[3]) << 16);
            o.z = f32_to_bf16_rne(acc[4]) | ((uint32_t)f32_to_bf16_rne(acc[5]) << 16);
            o.w = f32_to_bf16_rne(acc[6]) | ((uint32_t)f32_to_bf16_rne(acc[7]) << 16);
            *reinterpret_cast<uint4 *>((uint16_t *)y + row * ldy + c) = o;
        }
    }
}

// any F / alignment: one column per lane and step
template <int L, int U, bool OUT_F32>
__global__ void __launch_bounds__(256)
k_spmm_bf16x1(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
              const float *__restrict__ w, const uint16_t *__restrict__ x, int64_t ldx,
              const void *addend, int64_t ldadd, void *y, int64_t ldy, int64_t N, int F) {
    constexpr int kRows = 256 / L;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = (int64_t)lb * kRows + threadIdx.x / L;
    if (row >= N) return;
    const int sub = threadIdx.x % L;
    const int beg = ptr[row], end = ptr[row + 1];
    for (int c = sub; c < F; c += L) {
        float acc = 0.0f;
        if (addend)
            acc = OUT_F32 ? ((const float *)addend)[row * ldadd + c]
                          : bf16_to_f32(((const uint16_t *)addend)[row * ldadd + c]);
        for (int p = beg; p < end; p += U) {
            int s[U];
            float ww[U], v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool ok = p + j < end;
                s[j] = ok ? other[p + j] : 0;
                ww[j] = ok ? (w ? w[p + j] : 1.0f) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
                v[j] = (p + j < end) ? bf16_to_f32(x[(int64_t)s[j] * ldx + c]) : 0.0f;
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (p + j < end) {
                    const float m = ww[j] * v[j];
                    acc = acc + m;
                }
        }
        if (OUT_F32) ((float *)y)[row * ldy + c] = acc;
        else ((uint16_t *)y)[row * ldy + c] = f32_to_bf16_rne(acc);
    }
}

template <int L, bool X8, bool OUT_F32>
static void launch_bf16(const int32_t *ptr, const int32_t *other, const float *w, const uint16_t *x,
                        int64_t ldx, const void *addend, int64_t ldadd, void *y, int64_t ldy,
                        int64_t N, int F, hipStream_t stream) {
    constexpr int kRows = 256 / L;
    const unsigned grid = (unsigned)((N + kRows - 1) / kRows);
    if (X8)
        DC_LAUNCH((k_spmm_bf16x8<L, 8, OUT_F32>), dim3(grid), dim3(256), 0, stream, ptr,
                           other, w, x, ldx, addend, ldadd, y, ldy, N, F);
    else
        DC_LAUNCH((k_spmm_bf16x1<L, 8, OUT_F32>), dim3(grid), dim3(256), 0, stream, ptr,
                           other, w, x, ldx, addend, ldadd, y, ldy, N, F);
}

template <bool OUT_F32>
static void dispatch_bf16(const int32_t *ptr, const int32_t *other, const float *w,
                          const uint16_t *x, int64_t ldx, const void *addend, int64_t ldadd,
                          void *y, int64_t ldy, int64_t N, int F, bool x8, hipStream_t stream) {
    // measured (tools/hop_stress.py, r01): while x sits in L2 / the Infinity Cache the plain
    // half-wave kernel is as fast or faster (14.3 vs 16.3 us at the everyday shape); once x spills
    // to HBM the scalar-index row-pair kernel wins (842 vs 990 us at 2.1 M rows)
    // the rule looks at what ONE hop gathers from - N rows of F values - not at the slab the block sits in (round 4: on the
    // 100k-point radius graph the slab's leading dimension made the rule pick the row-pair kernel for a 51 MB block that the
    // Infinity Cache holds: 67 vs 45 us per hop, forward 0.64 vs 0.51 ms)
    const bool pair = N * (int64_t)F * 2 > (int64_t)128 << 20;
    if (x8 && F >= 256 && pair) {
        const unsigned grid = (unsigned)((N + 7) / 8);
        DC_LAUNCH((k_spmm_bf16_pair<8, OUT_F32>), dim3(grid), dim3(256), 0, stream, ptr,
                           other, w, x, ldx, addend, ldadd, y, ldy, N, F);
    } else if (x8) {
        const int v = F / 8;
        if (v > 16) launch_bf16<32, true, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
        else if (v > 8) launch_bf16<16, true, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
        else if (v > 4) launch_bf16<8, true, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
        else launch_bf16<4, true, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
    } else {
        if (F > 32) launch_bf16<64, false, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
        else if (F > 16) launch_bf16<32, false, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
        else launch_bf16<16, false, OUT_F32>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, stream);
    }
}

}  // namespace dc

using namespace dc;

static int spmm_f32_impl(const int32_t *ptr, const int32_t *other, const float *w,
                         const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                         float *y, int64_t ldy, int64_t N, int64_t F, int src_off, dc_stream_t stream_) {
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
            DC_LAUNCH((k_spmm_wave<4, 8>), dim3(grid), dim3(256), 0, stream, ptr, other,
                               w, x, ldx, addend, ldadd, y, ldy, N, Fi, nullptr, 0, src_off);
        } else if (v > 16)
            launch_sub<4, 32, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
        else if (v > 8)
            launch_sub<4, 16, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
        else if (v > 4)
            launch_sub<4, 8, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
        else
            launch_sub<4, 4, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
    } else {
        if (F > 32) {
            const unsigned grid = (unsigned)((N + 3) / 4);
            DC_LAUNCH((k_spmm_wave<1, 8>), dim3(grid), dim3(256), 0, stream, ptr, other,
                               w, x, ldx, addend, ldadd, y, ldy, N, Fi, nullptr, 0, src_off);
        } else if (F > 16)
            launch_sub<1, 32, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
        else if (F > 8)
            launch_sub<1, 16, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
        else
            launch_sub<1, 8, 8>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, Fi, src_off, stream);
    }
    return check_launch("dc_spmm_f32");
}


extern "C" int dc_spmm_f32(const int32_t *ptr, const int32_t *other, const float *w,
                           const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                           float *y, int64_t ldy, int64_t N, int64_t F, dc_stream_t stream) {
    return spmm_f32_impl(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, 0, stream);
}

// y = act(A x + bias) in one launch (GCNConv / GATConv aggregation + bias + the encoder's ReLU); one wave per row for
// every width (these layers aggregate the hidden width, 256)
extern "C" int dc_spmm_f32_bias_act(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                                    int64_t ldx, const float *bias, int relu, float *y, int64_t ldy, int64_t N,
                                    int64_t F, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 0, "dc_spmm_f32_bias_act: negative size");
    if (N == 0 || F == 0) return DC_OK;
    DC_REQUIRE(ptr && x && y, "dc_spmm_f32_bias_act: null ptr/x/y");
    DC_REQUIRE(N < (int64_t)INT32_MAX / 4 && F < (1 << 24) && ldx >= F && ldy >= F && x != y,
               "dc_spmm_f32_bias_act: bad sizes / aliasing");
    const bool vec4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y) &&
                      (!bias || aligned16(bias));
    const unsigned grid = (unsigned)((N + 3) / 4);
    if (vec4)
        DC_LAUNCH((k_spmm_wave<4, 8, false, true>), dim3(grid), dim3(256), 0, stream, ptr, other, w, x, ldx, nullptr,
                  (int64_t)0, y, ldy, N, (int)F, nullptr, 0, 0, bias, relu);
    else
        DC_LAUNCH((k_spmm_wave<1, 8, false, true>), dim3(grid), dim3(256), 0, stream, ptr, other, w, x, ldx, nullptr,
                  (int64_t)0, y, ldy, N, (int)F, nullptr, 0, 0, bias, relu);
    return check_launch("dc_spmm_f32_bias_act");
}

extern "C" int dc_spmm_f32_pack(const int32_t *ptr, const int32_t *other, const float *w, const float *x,
                                int64_t ldx, float *slab, int64_t lds, int64_t N, int64_t F, int64_t width,
                                int64_t wpad, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 1 && F <= 32, "dc_spmm_f32_pack: rows of 1..32 floats (got F=%lld)", (long long)F);
    if (N == 0) return DC_OK;
    DC_REQUIRE(ptr && x && slab && (other || true), "dc_spmm_f32_pack: null pointer");
    DC_REQUIRE(N < (int64_t)INT32_MAX / 8 && ldx >= F && width >= 2 * F && wpad >= width && lds >= wpad,
               "dc_spmm_f32_pack: needs ldx >= F, width >= 2 F (block 1 exists), wpad >= width, lds >= wpad");
    DC_REQUIRE((const void *)x != (const void *)slab, "dc_spmm_f32_pack: the slab must not alias x");
    float *y = slab + F;
    if (F > 16)
        launch_sub<1, 32, 8>(ptr, other, w, x, ldx, nullptr, 0, y, lds, N, (int)F, 0, stream, slab, (int)width, (int)wpad);
    else if (F > 8)
        launch_sub<1, 16, 8>(ptr, other, w, x, ldx, nullptr, 0, y, lds, N, (int)F, 0, stream, slab, (int)width, (int)wpad);
    else
        launch_sub<1, 8, 8>(ptr, other, w, x, ldx, nullptr, 0, y, lds, N, (int)F, 0, stream, slab, (int)width, (int)wpad);
    return check_launch("dc_spmm_f32_pack");
}

extern "C" int dc_spmm_f32_window(const int32_t *ptr, const int32_t *other, const float *w,
                                  const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                                  float *y, int64_t ldy, int64_t N, int64_t F, int64_t row_offset,
                                  dc_stream_t stream) {
    DC_REQUIRE(row_offset >= 0 && row_offset < (int64_t)INT32_MAX, "dc_spmm_f32_window: bad row_offset");
    return spmm_f32_impl(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, (int)row_offset, stream);
}

extern "C" int dc_spmm_bf16(const int32_t *ptr, const int32_t *other, const float *w,
                            const uint16_t *x, int64_t ldx, const void *addend, int64_t ldadd,
                            void *y, int64_t ldy, int64_t N, int64_t F, int y_is_f32,
                            dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 0, "dc_spmm_bf16: negative size N=%lld F=%lld", (long long)N,
               (long long)F);
    if (N == 0 || F == 0) return DC_OK;
    DC_REQUIRE(ptr && x && y, "dc_spmm_bf16: null ptr/x/y");
    DC_REQUIRE(N < (int64_t)INT32_MAX / 8 && F < (1 << 24), "dc_spmm_bf16: size out of range");
    DC_REQUIRE(ldx >= F && ldy >= F && (!addend || ldadd >= F),
               "dc_spmm_bf16: leading dimension smaller than F");
    DC_REQUIRE((const void *)x != (const void *)y, "dc_spmm_bf16: y must not alias x");
    // 16-byte accesses: 8 bf16 of x; 8 bf16 or 2 x 4 fp32 of y / addend
    const int64_t ymod = y_is_f32 ? 4 : 8;
    const bool x8 = (F % 8 == 0) && (ldx % 8 == 0) && aligned16(x) && (ldy % ymod == 0) &&
                    aligned16(y) && (!addend || ((ldadd % ymod == 0) && aligned16(addend)));
    if (y_is_f32)
        dispatch_bf16<true>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, (int)F, x8, stream);
    else
        dispatch_bf16<false>(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, (int)F, x8, stream);
    return check_launch("dc_spmm_bf16");
}


static int spmm_f32_rowmax_impl(const int32_t *ptr, const int32_t *other, const float *w,
                                const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                                float *y, int64_t ldy, int64_t N, int64_t F, float *rowmax,
                                int mode, int src_off, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 0, "dc_spmm_f32_rowmax: negative size");
    if (N == 0) return DC_OK;
    DC_REQUIRE(F >= 1 && ptr && x && y && rowmax, "dc_spmm_f32_rowmax: null ptr/x/y/rowmax or F == 0");
    DC_REQUIRE(N < (int64_t)INT32_MAX / 4 && F < (1 << 24), "dc_spmm_f32_rowmax: size out of range");
    DC_REQUIRE(ldx >= F && ldy >= F && (!addend || ldadd >= F),
               "dc_spmm_f32_rowmax: leading dimension smaller than F");
    DC_REQUIRE(x != y, "dc_spmm_f32_rowmax: y must not alias x");
    DC_REQUIRE((mode & ~3) == 0, "dc_spmm_f32_rowmax: mode is a 2-bit mask");
    const bool vec4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) &&
                      aligned16(y) && (!addend || ((ldadd % 4 == 0) && aligned16(addend)));
    const unsigned grid = (unsigned)((N + 3) / 4);
    if (vec4)
        DC_LAUNCH((k_spmm_wave<4, 8, true>), dim3(grid), dim3(256), 0, stream, ptr, other, w,
                           x, ldx, addend, ldadd, y, ldy, N, (int)F, rowmax, mode, src_off);
    else
        DC_LAUNCH((k_spmm_wave<1, 8, true>), dim3(grid), dim3(256), 0, stream, ptr, other, w,
                           x, ldx, addend, ldadd, y, ldy, N, (int)F, rowmax, mode, src_off);
    return check_launch("dc_spmm_f32_rowmax");
}

extern "C" int dc_spmm_f32_rowmax(const int32_t *ptr, const int32_t *other, const float *w,
                                  const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                                  float *y, int64_t ldy, int64_t N, int64_t F, float *rowmax,
                                  int mode, dc_stream_t stream) {
    return spmm_f32_rowmax_impl(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, rowmax, mode, 0, stream);
}

extern "C" int dc_spmm_f32_rowmax_window(const int32_t *ptr, const int32_t *other, const float *w,
                                         const float *x, int64_t ldx, const float *addend, int64_t ldadd,
                                         float *y, int64_t ldy, int64_t N, int64_t F, float *rowmax,
                                         int mode, int64_t row_offset, dc_stream_t stream) {
    DC_REQUIRE(row_offset >= 0 && row_offset < (int64_t)INT32_MAX, "dc_spmm_f32_rowmax_window: bad row_offset");
    return spmm_f32_rowmax_impl(ptr, other, w, x, ldx, addend, ldadd, y, ldy, N, F, rowmax, mode, (int)row_offset,
                                stream);
}
