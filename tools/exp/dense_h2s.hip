// tools/exp/dense_h2s.hip -- EXPERIMENT (not part of the product library): the forward-shaped fp16x2 dense block,
// 128 x 256 tiles, with the workgroup's eight waves split by ROLE (gfx950).  Built by tools/exp/dense_h2s.py.
//
// Same arithmetic and operand formats as k_fwd_h2w (results bit-identical).  Why: in k_fwd_h2w every wave loads,
// splits, stores, reads fragments and issues MFMAs, two waves per SIMD, all in step behind one barrier per stage -
// the matrix pipe is busy 31 % of the launch although no single unit is saturated (profiles/r04: dense_pmc.json).
// Here waves 0-3 (one per SIMD) only read fragments and issue MFMAs (wave tile 64 x 128: 48 MFMAs per stage against
// 24 fragment reads), waves 4-7 (the other wave of each SIMD) only move data: global -> registers -> split -> LDS,
// loads two stages ahead.  Their VALU / LDS / VMEM instructions issue beside the other wave's MFMAs; the MFMA
// waves never wait on vmcnt.
//   -DH2S_ABL=<bits>  timing-only builds: 1 no MFMAs, 2 producers idle, 4 no fragment reads, 8 no barrier in the loop,
//                     16 weight image read as if stored stage-major (32 KB contiguous per stage)
#include "../../deformcontact_amd/csrc/dc_dense.h"

#ifndef H2S_ABL
#define H2S_ABL 0
#endif
#ifndef H2S_SGB
#define H2S_SGB 1          // fragment reads dealt out between the MFMAs (1 per gap) instead of in clumps
#endif

namespace dc { __device__ long long *g_h2s_dbg = nullptr; __device__ long long *g_h2s_trace = nullptr; }
#ifndef H2S_TRACE
#define H2S_TRACE 0
#endif
#define H2S_MARK(role, slot) do { if (H2S_TRACE && g_h2s_trace && lane == 0 && (wid & 3) == 0) g_h2s_trace[(blockIdx.x * 2 + (role)) * 160 + (slot)] = wall_clock64(); } while (0)   // per workgroup: {core-clock cycles, 100 MHz ticks} of the main loop

namespace dc {

using hs_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using hs_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hs_f32x4 = __attribute__((ext_vector_type(4))) float;
using hs_u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kSBM = 128, kSBN = 256, kSBK = 32;
constexpr int kSRow = 128;                              // bytes per LDS row: 8 pieces of 16 B (see dc_dense_h2w.hip)
constexpr int kSSzA = kSBM * kSRow, kSSzB = kSBN * kSRow;

__device__ __forceinline__ int hs_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }

template <bool FULL>
__global__ void __launch_bounds__(512)
k_fwd_h2s(FwdParams p) {
    __shared__ __attribute__((aligned(16))) char sA[3 * kSSzA];      // ring of three stages: 48 + 96 KB
    __shared__ __attribute__((aligned(16))) char sB[3 * kSSzB];
    __shared__ __attribute__((aligned(16))) float s_inv[kSBM];
    const unsigned ntn = (unsigned)((p.Fo + kSBN - 1) / kSBN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * kSBM, col0 = (int64_t)(lb % ntn) * kSBN;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nst = (int)(p.Fi / kSBK);
    auto nextb = [](int b) { return b == 2 ? 0 : b + 1; };

    if (wid >= 4) {
        // ------------------------------------------------------------------ producers: 256 threads
        const int t = threadIdx.x - 256, k8 = t & 7, r = t >> 3;     // 8 threads per 128-byte row, 32 rows per pass
        const int64_t lda = p.x[0].ld;
        const int f = hs_swz(r);                                      // rows r + 32 j share it
        const int qa = 4 * (k8 >> 2) + ((k8 >> 1) & 1);
        const int ldsAh = r * kSRow + 16 * (qa ^ f) + 8 * (k8 & 1);
        const int ldsAl = r * kSRow + 16 * ((qa + 2) ^ f) + 8 * (k8 & 1);
        const int ldsB = r * kSRow + 16 * (k8 ^ f);
        unsigned offA[4], offB[8];
        float scA[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = r + 32 * j;
            int64_t row = row0 + rl;
            row = (FULL || row < p.N) ? row : p.N - 1;
            offA[j] = (unsigned)((row - row0) * lda + 4 * k8);
            const float m = p.h2.a_rowmax[row];
            scA[j] = h2_scale(m);
            if (k8 == 0) s_inv[rl] = h2_unscale(m);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int64_t col = col0 + r + 32 * j;
            col = (FULL || col < p.Fo) ? col : p.Fo - 1;
            offB[j] = (unsigned)((col - col0) * p.Fi + 4 * k8);
            if (H2S_ABL & 16) offB[j] = (unsigned)((j * 256 + t) * 4);
        }
        const float *baseA = p.x[0].p + row0 * lda;
        const float *baseB = p.w[0].p + col0 * p.Fi;
        hs_f32x4 va0[4], va1[4];
        hs_u32x4 vb0[8], vb1[8];
        auto gload_set = [&](hs_f32x4 (&va)[4], hs_u32x4 (&vb)[8]) {
            if (H2S_ABL & 2) return;
#pragma unroll
            for (int j = 0; j < 8; ++j) vb[j] = *reinterpret_cast<const hs_u32x4 *>(baseB + offB[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) va[j] = *reinterpret_cast<const hs_f32x4 *>(baseA + offA[j]);
            baseA += kSBK;
            baseB += (H2S_ABL & 16) ? 8192 : kSBK;
        };
        auto lstore_set = [&](const hs_f32x4 (&va)[4], const hs_u32x4 (&vb)[8], int b) {
            if (H2S_ABL & 2) return;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<hs_u32x4 *>(sB + b * kSSzB + ldsB + j * 32 * kSRow) = vb[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const hs_f32x4 v = va[j] * scA[j];
                hs_f16x4 h, l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 a = (_Float16)v[i];
                    h[i] = a;
                    l[i] = (_Float16)(v[i] - (float)a);
                }
                *reinterpret_cast<hs_f16x4 *>(sA + b * kSSzA + ldsAh + j * 32 * kSRow) = h;
                *reinterpret_cast<hs_f16x4 *>(sA + b * kSSzA + ldsAl + j * 32 * kSRow) = l;
            }
        };
        auto gload = [&](int set) {
            if (set == 0) gload_set(va0, vb0); else gload_set(va1, vb1);
        };
        auto lstore = [&](int set, int b) {
            if (set == 0) lstore_set(va0, vb0, b); else lstore_set(va1, vb1, b);
        };
        // stage s: register set s & 1, LDS buffer s % 3.  Iteration `it` (the consumers compute stage it): loads of
        // stage it+3 issued, stage it+2 (loaded during iteration it-1) split and written, barrier.
        H2S_MARK(1, 0);
        gload(0);                                      // stage 0
        if (nst > 1) gload(1);                         // stage 1
        lstore(0, 0);
        if (nst > 2) gload(0);                         // stage 2
        if (nst > 1) lstore(1, 1);
        H2S_MARK(1, 1);
        __syncthreads();                               // P: stages 0 and 1 readable
        H2S_MARK(1, 2);
        int it = 0, b2 = 2;                            // b2 = (it + 2) % 3
        for (; it + 4 < nst; it += 2) {
            gload(1);                                  // stage it+3
            if (H2S_TRACE == 2) {
                H2S_MARK(1, 80 + 2 * it);
                __builtin_amdgcn_s_waitcnt(0x0F7C);    // vmcnt(12): the loads of stage it+2 have landed
                H2S_MARK(1, 81 + 2 * it);
            }
            lstore(0, b2);                             // stage it+2
            b2 = nextb(b2);
            if (H2S_TRACE == 2) {
                __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): LDS stores done
                H2S_MARK(1, 82 + 2 * it);
            }
            H2S_MARK(1, 4 + 2 * it);
            if (!(H2S_ABL & 8)) __syncthreads();
            H2S_MARK(1, 5 + 2 * it);
            gload(0);                                  // stage it+4
            lstore(1, b2);                             // stage it+3
            b2 = nextb(b2);
            H2S_MARK(1, 6 + 2 * it);
            if (!(H2S_ABL & 8)) __syncthreads();
            H2S_MARK(1, 7 + 2 * it);
        }
        for (; it < nst; ++it) {
            const bool even = (it & 1) == 0;
            if (it + 3 < nst) {
                if (even) gload(1); else gload(0);
            }
            if (it + 2 < nst) {
                if (even) lstore(0, b2); else lstore(1, b2);
            }
            b2 = nextb(b2);
            H2S_MARK(1, 4 + 2 * it);
            if (!(H2S_ABL & 8)) __syncthreads();
            H2S_MARK(1, 5 + 2 * it);
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers: waves 0-3, 64 x 128 each
    H2S_MARK(0, 0);
    __builtin_amdgcn_s_setprio(3);
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 31, fh = lane >> 5, fsw = hs_swz(fr);
    const int fragA = (wm * 64 + fr) * kSRow, fragB = (wn * 128 + fr) * kSRow;
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
    hs_f16x8 fa0[2][2], fb0[4][2], fa1[2][2], fb1[4][2];
    auto frags = [&](hs_f16x8 (&fa)[2][2], hs_f16x8 (&fb)[4][2], int b, int ks) {
        if (H2S_ABL & 4) return;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fa[mb][pl] = *reinterpret_cast<const hs_f16x8 *>(sA + b * kSSzA + fragA + mb * 32 * kSRow +
                                                                 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fb[nb][pl] = *reinterpret_cast<const hs_f16x8 *>(sB + b * kSSzB + fragB + nb * 32 * kSRow +
                                                                 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
    };
    auto mma = [&](const hs_f16x8 (&fa)[2][2], const hs_f16x8 (&fb)[4][2]) {
        if (H2S_ABL & 1) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb][0] += (float)fa[mb][0][0] + (float)fb[nb][1][1] +
                                                                 (float)fa[mb][1][2] + (float)fb[nb][0][3];
            return;
        }
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};       // smallest terms first (as k_fwd_h2)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]],
                                                                         acc[mb][nb], 0, 0, 0);
    };
    if (H2S_ABL & 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) fa0[i][j][e] = fa1[i][j][e] = (_Float16)(lane + e);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) fb0[i][j][e] = fb1[i][j][e] = (_Float16)(lane - e);
    }
    H2S_MARK(0, 1);
    __syncthreads();                                   // P
    H2S_MARK(0, 2);
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    frags(fa0, fb0, 0, 0);
    int cur = 0;
    auto deal = [&]() {                                // 24 MFMAs, 12 fragment reads: one read per gap, then the rest
        if (!H2S_SGB || (H2S_ABL & 5)) return;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    };
    for (int it = 0; it < nst; ++it) {
        const int b1 = nextb(cur);
        frags(fa1, fb1, cur, 1);
        mma(fa0, fb0);
        deal();
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nst) frags(fa0, fb0, b1, 0);
        mma(fa1, fb1);
        deal();
        __builtin_amdgcn_sched_barrier(0);
        H2S_MARK(0, 4 + 2 * it);
        if (!(H2S_ABL & 8)) __syncthreads();
        H2S_MARK(0, 5 + 2 * it);
        cur = b1;
    }
    __builtin_amdgcn_s_setprio(0);
    if (g_h2s_dbg && threadIdx.x == 0) {
        g_h2s_dbg[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
        g_h2s_dbg[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }

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
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = wm * 64 + mb * 32 + 8 * g + 4 * h;
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
    H2S_MARK(0, 70);
    if (H2S_TRACE) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        H2S_MARK(0, 71);
    }
}

}  // namespace dc

extern "C" int h2s_set_trace(long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(dc::g_h2s_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
extern "C" int h2s_set_dbg(long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(dc::g_h2s_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}

// out[N,Fo] = act(x[N,K] . W^T + b) with W as dc_tag_weight_prep's image; same contract as dc_tag_linear_fwd_h2p
extern "C" int h2s_run(const float *x, int64_t ldx, const void *w_image, const float *bias, int relu, float *out,
                       int64_t ldo, int64_t N, int64_t K, int64_t Fo, const float *x_rowmax, const float *w_rowmax,
                       void *stream) {
    using namespace dc;
    if (K % kSBK != 0 || K < kSBK || ldx % 4 != 0 || ldx * kSBM >= ((int64_t)1 << 30) || K * kSBN >= ((int64_t)1 << 30))
        return 1;
    FwdParams p{};
    p.x[0] = Mat{x, ldx};
    p.w[0] = Mat{(const float *)w_image, K};
    p.bias = bias, p.out = out, p.ldo = ldo, p.N = N, p.Fi = K, p.Fo = Fo, p.nseg = 1, p.relu = relu;
    p.h2.a_rowmax = x_rowmax, p.h2.b_rowmax = w_rowmax, p.h2.b_presplit = 1;
    const int64_t tiles = ((N + kSBM - 1) / kSBM) * ((Fo + kSBN - 1) / kSBN);
    const dim3 gd((unsigned)tiles), bd(512);
    if (N % kSBM == 0 && Fo % kSBN == 0)
        hipLaunchKernelGGL((k_fwd_h2s<true>), gd, bd, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_fwd_h2s<false>), gd, bd, 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
