// tools/exp/dense_h2r.hip -- EXPERIMENT (not part of the product library): the forward-shaped fp16x2 dense block,
// 128 x 256 tiles, the workgroup's eight waves split into THREE roles (gfx950).  Built by tools/exp/dense_h2s.py
// (H2S_SRC=dense_h2r.hip).
//
// Same arithmetic and operand formats as k_fwd_h2w (results bit-identical).  What tools/exp/dense_h2s.hip showed
// (profiles/r04): with MFMA waves and data-moving waves apart, the data-moving waves are what every stage waits for -
// 1.1 us to get 12 loads per thread issued behind the previous stage's, 0.8 us more until those have landed, 0.5 us
// for the split and the 48 KB of LDS stores.  Here:
//   waves 0-3  read fragments and issue MFMAs (wave tile 64 x 128, 48 MFMAs and 24 ds_read_b128 per stage);
//   waves 4-5  bring the weight tile in by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write; the
//              swizzle is applied to the per-lane source address), one stage ahead - the image is L2-resident;
//   waves 6-7  stream x: buffer loads NSA stages ahead (their own vmcnt queue: nothing L2-resident waits behind an
//              HBM miss), scale + split + ds_write_b64.
//   -DH2S_ABL=<bits>  timing-only builds: 1 no MFMAs, 2 x waves idle, 4 no fragment reads, 8 no barrier in the loop,
//                     32 weight waves idle
#include "../../deformcontact_amd/csrc/dc_dense.h"

#ifndef H2S_ABL
#define H2S_ABL 0
#endif
#ifndef H2S_SGB
#define H2S_SGB 1          // fragment reads dealt out between the MFMAs (1 per gap) instead of in clumps
#endif
#ifndef H2S_NSA
#define H2S_NSA 4          // register sets (= stages in flight) of the x waves
#endif
#ifndef H2S_PRIO
#define H2S_PRIO 3
#endif
#ifndef H2S_TRACE
#define H2S_TRACE 0
#endif

namespace dc {
__device__ long long *g_h2s_dbg = nullptr;     // per workgroup: {core-clock cycles, 100 MHz ticks} of the main loop
__device__ long long *g_h2s_trace = nullptr;
}
#define H2S_MARK(role, slot)                                                                          \
    do {                                                                                              \
        if (H2S_TRACE && g_h2s_trace && lane == 0 && (wid == 0 || wid == 4 || wid == 6))              \
            g_h2s_trace[(blockIdx.x * 3 + (role)) * 160 + (slot)] = wall_clock64();                   \
    } while (0)

namespace dc {

using hs_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using hs_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hs_f32x4 = __attribute__((ext_vector_type(4))) float;
using hs_u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kSBM = 128, kSBN = 256, kSBK = 32;
constexpr int kSRow = 128;                              // bytes per LDS row: 8 pieces of 16 B (see dc_dense_h2w.hip)
constexpr int kSSzA = kSBM * kSRow, kSSzB = kSBN * kSRow;
constexpr int kNSA = H2S_NSA;

__device__ __forceinline__ int hs_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }

template <bool FULL>
__global__ void __launch_bounds__(512)
k_fwd_h2r(FwdParams p) {
    __shared__ __attribute__((aligned(1024))) char sB[3 * kSSzB];    // ring of three stages: 96 + 48 KB
    __shared__ __attribute__((aligned(16))) char sA[3 * kSSzA];
    __shared__ __attribute__((aligned(16))) float s_inv[kSBM];
    const unsigned ntn = (unsigned)((p.Fo + kSBN - 1) / kSBN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * kSBM, col0 = (int64_t)(lb % ntn) * kSBN;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nst = (int)(p.Fi / kSBK);
    auto nextb = [](int b) { return b == 2 ? 0 : b + 1; };
    auto bar = [&]() {
        if (!(H2S_ABL & 8)) __syncthreads();
    };

    if (wid >= 6) {
        // ------------------------------------------------------------------ x waves: 128 threads
        const int t = threadIdx.x - 384, k8 = t & 7, r = t >> 3;     // 8 threads per 128-byte row piece, 16 rows per pass
        const int64_t lda = p.x[0].ld;
        const int f = hs_swz(r);                                      // rows r + 16 j share it
        const int qa = 4 * (k8 >> 2) + ((k8 >> 1) & 1);
        const int ldsAh = r * kSRow + 16 * (qa ^ f) + 8 * (k8 & 1);
        const int ldsAl = r * kSRow + 16 * ((qa + 2) ^ f) + 8 * (k8 & 1);
        float scA[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = r + 16 * j;
            int64_t row = row0 + rl;
            row = (FULL || row < p.N) ? row : p.N - 1;
            const float m = p.h2.a_rowmax[row];
            scA[j] = h2_scale(m);
            if (k8 == 0) s_inv[rl] = h2_unscale(m);
        }
        // the tile's rows as a buffer: rows past N fall out of range and load zeros (their results are never stored)
        const int64_t trows = p.N - row0 < kSBM ? p.N - row0 : kSBM;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.x[0].p + row0 * lda), 0, (int)(((trows - 1) * lda + p.Fi) * 4), 0x00020000);
        const int voff = (int)((r * lda + 4 * k8) * 4), vstep = (int)(16 * lda * 4);
        hs_f32x4 va[kNSA][8];
        auto gload = [&](hs_f32x4 (&v)[8], int stage) {
            if (H2S_ABL & 2) return;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const hs_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * vstep, stage * (kSBK * 4), 0);
                v[j] = __builtin_bit_cast(hs_f32x4, w);
            }
        };
        auto lstore = [&](const hs_f32x4 (&v)[8], int b) {
            if (H2S_ABL & 2) return;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const hs_f32x4 x = v[j] * scA[j];
                hs_f16x4 h, l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 a = (_Float16)x[i];
                    h[i] = a;
                    l[i] = (_Float16)(x[i] - (float)a);
                }
                *reinterpret_cast<hs_f16x4 *>(sA + b * kSSzA + ldsAh + j * 16 * kSRow) = h;
                *reinterpret_cast<hs_f16x4 *>(sA + b * kSSzA + ldsAl + j * 16 * kSRow) = l;
            }
        };
        if (H2S_ABL & 64) {
            // timing only: x by LDS-DMA, raw fp32 (16 KB per stage, the size of the split image), one stage ahead
            const int w = wid - 6;
            const int xo = (int)(((int64_t)(lane >> 3) * lda) * 4 + 16 * (lane & 7));
            const int cst = (int)(8 * lda * 4);
            auto xstage = [&](int s, int b) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 8 * w + j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (void __attribute__((address_space(3))) *)(sA + b * kSSzA + c * 1024),
                                                             16, xo, c * cst + s * (kSBK * 4), 0, 0);
                }
            };
            xstage(0, 0);
            xstage(1, 1);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            int b2 = 2;
            for (int it = 0; it < nst; ++it) {
                if (it + 2 < nst) xstage(it + 2, b2);
                b2 = nextb(b2);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                bar();
            }
            return;
        }
        // stage s: register set s % NSA, LDS buffer s % 3.  Iteration `it` (the MFMA waves compute stage it): loads of
        // stage it+1+NSA into the set stage it+1 left, stage it+2 split and written, barrier.
        H2S_MARK(2, 0);
#pragma unroll
        for (int s = 0; s < kNSA; ++s)
            if (s < nst) gload(va[s], s);
        lstore(va[0], 0);
        if (kNSA < nst) gload(va[0], kNSA);
        if (nst > 1) lstore(va[1 % kNSA], 1);
        H2S_MARK(2, 1);
        __syncthreads();                               // P: stages 0 and 1 readable
        H2S_MARK(2, 2);
        int b2 = 2;                                    // (it + 2) % 3
        int it0 = 0;
        // steady state: every iteration of a trip loads and stores (no branch around a load: behind one hipcc has
        // to assume the fewest loads in flight and waits for the newest ones - the whole latency, every stage)
        for (; it0 + 2 * kNSA <= nst; it0 += kNSA) {
#pragma unroll
            for (int u = 0; u < kNSA; ++u) {
                const int it = it0 + u;
                gload(va[(u + 1) % kNSA], it + 1 + kNSA);
                __builtin_amdgcn_sched_barrier(0);
                if (H2S_TRACE == 2) {
                    H2S_MARK(2, 80 + 2 * it);
                    constexpr int n = 8 * (kNSA - 1);
                    __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));
                    H2S_MARK(2, 81 + 2 * it);
                }
                lstore(va[(u + 2) % kNSA], b2);
                b2 = nextb(b2);
                if (H2S_TRACE == 2) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    H2S_MARK(2, 150);
                    if (lane == 0 && wid == 6 && g_h2s_trace)
                        g_h2s_trace[(blockIdx.x * 3 + 2) * 160 + 151 + 0] = 0;
                }
                H2S_MARK(2, 4 + 2 * it);
                bar();
                H2S_MARK(2, 5 + 2 * it);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; it0 < nst; it0 += kNSA) {               // last stages
#pragma unroll
            for (int u = 0; u < kNSA; ++u) {
                const int it = it0 + u;
                if (it < nst) {
                    if (it + 1 + kNSA < nst) gload(va[(u + 1) % kNSA], it + 1 + kNSA);
                    if (it + 2 < nst) lstore(va[(u + 2) % kNSA], b2);
                    b2 = nextb(b2);
                    H2S_MARK(2, 4 + 2 * it);
                    bar();
                    H2S_MARK(2, 5 + 2 * it);
                }
            }
        }
        return;
    }

    if (wid >= 4) {
        // ------------------------------------------------------------------ weight waves: 128 threads, LDS-DMA
        // instruction j of wave w fills rows 8 (16 w + j) .. + 7 of the tile (1 KiB, lane-linear): lane l is row
        // 8 c + (l >> 3), position l & 7, and fetches the piece that belongs there: q = position ^ swz(row)
        const int w = wid - 4;
        const int64_t tcols = p.Fo - col0 < kSBN ? p.Fo - col0 : kSBN;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.w[0].p + col0 * p.Fi), 0, (int)(tcols * p.Fi * 4), 0x00020000);
        int voff[2];                                   // chunks of even / odd index (8 rows: the swizzle repeats every 16)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rl = 8 * e + (lane >> 3);
            voff[e] = (int)(((int64_t)(lane >> 3) * p.Fi) * 4 + 16 * ((lane & 7) ^ hs_swz(rl)));
        }
        const int cstep = (int)(8 * p.Fi * 4);         // bytes from one chunk's rows to the next chunk's
        auto stage = [&](int s, int b) {
            if (H2S_ABL & 32) return;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = 16 * w + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (void __attribute__((address_space(3))) *)(sB + b * kSSzB + c * 1024),
                                                         16, voff[j & 1], c * cstep + s * (kSBK * 4), 0, 0);
            }
        };
        H2S_MARK(1, 0);
        stage(0, 0);
        if (nst > 1) stage(1, 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
        H2S_MARK(1, 1);
        __syncthreads();                               // P
        H2S_MARK(1, 2);
        int b2 = 2;
        for (int it = 0; it < nst; ++it) {
            if (it + 2 < nst) stage(it + 2, b2);
            b2 = nextb(b2);
            if (H2S_TRACE == 2) H2S_MARK(1, 80 + it);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            H2S_MARK(1, 4 + 2 * it);
            bar();
            H2S_MARK(1, 5 + 2 * it);
        }
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves 0-3, 64 x 128 each
    H2S_MARK(0, 0);
    if (H2S_PRIO) __builtin_amdgcn_s_setprio(H2S_PRIO);
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
        bar();
        H2S_MARK(0, 5 + 2 * it);
        cur = b1;
    }
    if (H2S_PRIO) __builtin_amdgcn_s_setprio(0);
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
    if (K % kSBK != 0 || K < kSBK || ldx % 4 != 0 || ldx * kSBM >= ((int64_t)1 << 29) || K * kSBN >= ((int64_t)1 << 29))
        return 1;
    FwdParams p{};
    p.x[0] = Mat{x, ldx};
    p.w[0] = Mat{(const float *)w_image, K};
    p.bias = bias, p.out = out, p.ldo = ldo, p.N = N, p.Fi = K, p.Fo = Fo, p.nseg = 1, p.relu = relu;
    p.h2.a_rowmax = x_rowmax, p.h2.b_rowmax = w_rowmax, p.h2.b_presplit = 1;
    const int64_t tiles = ((N + kSBM - 1) / kSBM) * ((Fo + kSBN - 1) / kSBN);
    const dim3 gd((unsigned)tiles), bd(512);
    if (N % kSBM == 0 && Fo % kSBN == 0)
        hipLaunchKernelGGL((k_fwd_h2r<true>), gd, bd, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_fwd_h2r<false>), gd, bd, 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
