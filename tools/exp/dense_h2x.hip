// tools/exp/dense_h2x.hip -- EXPERIMENT (not part of the product library): the forward-shaped fp16x2 dense block
// with 256 x 256 tiles, one wave per SIMD (gfx950).  Built into tools/exp/libdenseh2x.so by tools/exp/dense_h2x.py.
//
// Same arithmetic and operand formats as k_fwd_h2 / k_fwd_h2w (results bit-identical).  Why a third shape
// (r02 ablations of k_fwd_h2w, tools/exp/dense_abl.py + astream.py): with 128-row tiles every workgroup
// streams the whole 1 MB weight image from L2 for 0.5 MB of activations - 390 MB of L2 -> CU traffic per
// launch, 39 us when the loop does nothing but load, against ~30 us of matrix work at the sustained clock; and
// the two do not overlap well.  A 256 x 256 tile halves the weight traffic per row and the fragment reads per
// MFMA (wave tile 128 x 128: 16 ds_read_b128 feed 48 MFMAs per k-step).  256 accumulator registers per lane
// force ONE wave per SIMD (4 waves, 512 registers each); N / 256 tiles fill only half of the chip for one
// branch of the everyday batch (128 + 96 tiles), so this shape pays when the soft and the rigid branch run
// side by side on their two streams (224 tiles), and costs nothing when they do not (the tile does twice the
// work of a 128-row tile in about the time two of them take today).
#include "../../deformcontact_amd/csrc/dc_dense.h"

namespace dc {

using hx_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using hx_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hx_f32x4 = __attribute__((ext_vector_type(4))) float;
using hx_u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kXBM = 256, kXBN = 256, kXBK = 32;
constexpr int kXRow = 128;                              // bytes per LDS row: 8 pieces of 16 B (see dc_dense_h2w.hip)
constexpr int kXSzA = kXBM * kXRow, kXSzB = kXBN * kXRow;

__device__ __forceinline__ int hx_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }

template <bool FULL>
__global__ void __launch_bounds__(256)
k_fwd_h2x(FwdParams p) {
    __shared__ __attribute__((aligned(16))) char sA[2 * kXSzA];      // 2 x 32 KB
    __shared__ __attribute__((aligned(16))) char sB[2 * kXSzB];      // 2 x 32 KB
    __shared__ __attribute__((aligned(16))) float s_inv[kXBM];
    const unsigned ntn = (unsigned)((p.Fo + kXBN - 1) / kXBN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * kXBM, col0 = (int64_t)(lb % ntn) * kXBN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
    const int lane = threadIdx.x & 63;
    const int k8 = threadIdx.x & 7, r = threadIdx.x >> 3;            // staging: 8 threads per 128-byte row, 32 rows per pass
    const int64_t lda = p.x[0].ld;

    // per-thread constants.  Rows r + 32 j (j < 8) of both tiles: the LDS positions of pass j are those of
    // pass 0 plus j * 32 rows (32 rows do not change the swizzle), global offsets grow by 32 rows per pass
    const int fsw_st = hx_swz(r);
    const int qa = 4 * (k8 >> 2) + ((k8 >> 1) & 1);
    const int ldsAh = r * kXRow + 16 * (qa ^ fsw_st) + 8 * (k8 & 1);
    const int ldsAl = r * kXRow + 16 * ((qa + 2) ^ fsw_st) + 8 * (k8 & 1);
    const int ldsB = r * kXRow + 16 * (k8 ^ fsw_st);
    // global offsets of pass j: offX0 + j * strideX, clamped to the last valid row of the operand (rows past
    // N / Fo re-read it; their results are never stored) - computed per load, not held in registers
    const unsigned offA0 = (unsigned)(r * lda + 4 * k8), strideA = (unsigned)(32 * lda);
    const unsigned offB0 = (unsigned)(r * p.Fi + 4 * k8), strideB = (unsigned)(32 * p.Fi);
    const int64_t lastA = p.N - 1 - row0, lastB = p.Fo - 1 - col0;      // >= 0: the tile has at least one row / column
    const unsigned maxA = (unsigned)((lastA < kXBM - 1 ? lastA : kXBM - 1) * lda + 4 * k8);
    const unsigned maxB = (unsigned)((lastB < kXBN - 1 ? lastB : kXBN - 1) * p.Fi + 4 * k8);
    float scA[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rl = r + 32 * j;
        int64_t row = row0 + rl;
        row = (FULL || row < p.N) ? row : p.N - 1;
        const float m = p.h2.a_rowmax[row];
        scA[j] = h2_scale(m);
        if (k8 == 0) s_inv[rl] = h2_unscale(m);
    }
    auto offA = [&](int j) {
        const unsigned o = offA0 + (unsigned)j * strideA;
        return FULL ? o : (o < maxA ? o : maxA);
    };
    auto offB = [&](int j) {
        const unsigned o = offB0 + (unsigned)j * strideB;
        return FULL ? o : (o < maxB ? o : maxB);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
    const int nst = (int)(p.Fi / kXBK);
    const float *baseA = p.x[0].p + row0 * lda;
    const float *baseB = p.w[0].p + col0 * p.Fi;
    hx_f32x4 va0[8], va1[8];                           // x: two register sets (HBM latency), weights: one (L2)
    hx_u32x4 vb[8];

    // issue order matters (vmcnt counts in order): the weights of the NEXT stage first, then x two stages ahead -
    // the wait for the weights at the end of this stage then leaves the 8 x loads in flight
    auto gload_b = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) vb[j] = *reinterpret_cast<const hx_u32x4 *>(baseB + offB(j));
        baseB += kXBK;
    };
    auto gload_a = [&](hx_f32x4 (&va)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) va[j] = *reinterpret_cast<const hx_f32x4 *>(baseA + offA(j));
        baseA += kXBK;
    };
    auto lstore_b = [&](int b) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<hx_u32x4 *>(sB + b * kXSzB + ldsB + j * 32 * kXRow) = vb[j];
    };
    auto lstore_a = [&](const hx_f32x4 (&va)[8], int b, int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const hx_f32x4 v = va[j] * scA[j];
            hx_f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 a = (_Float16)v[i];
                h[i] = a;
                l[i] = (_Float16)(v[i] - (float)a);
            }
            *reinterpret_cast<hx_f16x4 *>(sA + b * kXSzA + ldsAh + j * 32 * kXRow) = h;
            *reinterpret_cast<hx_f16x4 *>(sA + b * kXSzA + ldsAl + j * 32 * kXRow) = l;
        }
    };
    const int fr = lane & 31, fh = lane >> 5, fsw = hx_swz(fr);
    const int fragA = (wm * 128 + fr) * kXRow, fragB = (wn * 128 + fr) * kXRow;
    hx_f16x8 fa[4][2], fb[4][2];
    auto frags = [&](int b, int ks) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fa[mb][pl] = *reinterpret_cast<const hx_f16x8 *>(sA + b * kXSzA + fragA + mb * 32 * kXRow +
                                                                 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fb[nb][pl] = *reinterpret_cast<const hx_f16x8 *>(sB + b * kXSzB + fragB + nb * 32 * kXRow +
                                                                 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
    };
    auto mma = [&]() {
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};       // smallest terms first (as k_fwd_h2)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]],
                                                                         acc[mb][nb], 0, 0, 0);
    };

    // Stage s lives in LDS buffer s & 1.  At the top of stage it: weights of stage it+1, x of stage it+2 (register
    // set it & 1); under the MFMAs: split + store of x of stage it+1 (set (it+1) & 1, loaded a stage ago), at the
    // end the weights of stage it+1.
    gload_b();
    gload_a(va0);
    if (nst > 1) gload_a(va1);
    lstore_b(0);
    lstore_a(va0, 0, 0, 8);
    __syncthreads();
    int it = 0;
#define DC_H2X_STAGE(CUR, VA_L, VA_S)                                                                 \
    gload_b();                                         /* stage it+1 */                               \
    gload_a(VA_L);                                     /* stage it+2 */                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    frags(CUR, 0);                                                                                    \
    mma();                                                                                            \
    lstore_a(VA_S, CUR ^ 1, 0, 4);                     /* stage it+1 */                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    frags(CUR, 1);                                                                                    \
    mma();                                                                                            \
    lstore_a(VA_S, CUR ^ 1, 4, 8);                                                                    \
    lstore_b(CUR ^ 1);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    __syncthreads();
    for (; it + 3 < nst; it += 2) {                    // steady state: two stages per trip (static register sets)
        DC_H2X_STAGE(0, va0, va1)
        DC_H2X_STAGE(1, va1, va0)
    }
#undef DC_H2X_STAGE
#define DC_H2X_TAIL(K, CUR, VA_L, VA_S)                                                               \
    if (it + K < nst) {                                                                               \
        if (it + K + 1 < nst) gload_b();                                                              \
        if (it + K + 2 < nst) gload_a(VA_L);                                                          \
        frags(CUR, 0);                                                                                \
        mma();                                                                                        \
        frags(CUR, 1);                                                                                \
        mma();                                                                                        \
        if (it + K + 1 < nst) {                                                                       \
            lstore_a(VA_S, CUR ^ 1, 0, 8);                                                            \
            lstore_b(CUR ^ 1);                                                                        \
        }                                                                                             \
        __syncthreads();                                                                              \
    }
    for (; it < nst; it += 2) {                        // last stages (at most three)
        DC_H2X_TAIL(0, 0, va0, va1)
        DC_H2X_TAIL(1, 1, va1, va0)
    }
#undef DC_H2X_TAIL

    // epilogue: C/D fragment (reg, lane) -> row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col lane & 31
    const bool relu = p.relu != 0;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int64_t col = col0 + wn * 128 + nb * 32 + c;
        const bool cok = FULL || col < p.Fo;
        const int64_t colc = cok ? col : p.Fo - 1;
        const float bcol = p.bias ? p.bias[colc] : 0.f;
        const float icol = h2_unscale(p.h2.b_rowmax[colc]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = wm * 128 + mb * 32 + 8 * g + 4 * h;
                const float4 si = *reinterpret_cast<const float4 *>(&s_inv[rl]);
                const float sv[4] = {si.x, si.y, si.z, si.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t row = row0 + rl + i;
                    float v = (acc[mb][nb][4 * g + i] * sv[i]) * icol;
                    v += bcol;
                    if (relu) v = fmaxf(v, 0.f);
                    if (FULL || (cok && row < p.N)) p.out[row * p.ldo + col] = v;
                }
            }
    }
}

}  // namespace dc

// out[N,Fo] = act(x[N,K] . W^T + b) with W as dc_tag_weight_prep's image; same contract as dc_tag_linear_fwd_h2p
extern "C" int h2x_run(const float *x, int64_t ldx, const void *w_image, const float *bias, int relu, float *out,
                       int64_t ldo, int64_t N, int64_t K, int64_t Fo, const float *x_rowmax, const float *w_rowmax,
                       void *stream) {
    using namespace dc;
    if (K % kXBK != 0 || K < kXBK || ldx % 4 != 0 || ldx * kXBM >= ((int64_t)1 << 30) || K * kXBN >= ((int64_t)1 << 30))
        return 1;
    FwdParams p{};
    p.x[0] = Mat{x, ldx};
    p.w[0] = Mat{(const float *)w_image, K};
    p.bias = bias, p.out = out, p.ldo = ldo, p.N = N, p.Fi = K, p.Fo = Fo, p.nseg = 1, p.relu = relu;
    p.h2.a_rowmax = x_rowmax, p.h2.b_rowmax = w_rowmax, p.h2.b_presplit = 1;
    const int64_t tiles = ((N + kXBM - 1) / kXBM) * ((Fo + kXBN - 1) / kXBN);
    const dim3 gd((unsigned)tiles), bd(256);
    if (N % kXBM == 0 && Fo % kXBN == 0)
        hipLaunchKernelGGL((k_fwd_h2x<true>), gd, bd, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_fwd_h2x<false>), gd, bd, 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
