// tools/exp/dense_h2d.hip -- EXPERIMENT (not part of the product library): the forward-shaped fp16x2 dense block,
// 128 x 256 tiles, BOTH operands by LDS-DMA, x split into its fp16 planes by the MFMA waves themselves (gfx950).
// Built by tools/exp/dense_h2s.py (H2S_SRC=dense_h2d.hip).
//
// Same arithmetic and operand formats as k_fwd_h2w (results bit-identical).  What tools/exp/dense_h2s.hip / dense_h2r.hip
// showed (profiles/r04): a wave that stages x through REGISTERS beside MFMA waves gets its loads issued at a third of
// the rate (1.5 us for 8 loads per stage), with or without barriers - x alone streams in 24 us, the MFMAs alone take
// 34 us, together 65; with x brought in by LDS-DMA (no VGPR write port involved) the same loop takes 39 us.  Here:
//   waves 0-3  (one per SIMD) read the fp32 x fragments and the weight fragments from LDS, scale + split x into its
//              two fp16 planes in registers (VALU under the MFMAs) and issue the MFMAs (wave tile 64 x 128);
//   waves 4-5  weight tile by LDS-DMA, ring of 3 stages (one stage ahead: the image is L2-resident);
//   waves 6-7  x tile by LDS-DMA as it is (fp32), ring of RX stages (RX - 2 ahead: HBM / Infinity Cache).
//   -DH2S_ABL=<bits>  timing-only builds: 1 no MFMAs, 2 x waves idle, 4 no fragment reads, 8 no barrier in the loop,
//                     32 weight waves idle, 128 no split (fragments used as read)
#include "../../deformcontact_amd/csrc/dc_dense.h"

#ifndef H2S_ABL
#define H2S_ABL 0
#endif
#ifndef H2S_RX
#define H2S_RX 4
#endif
#ifndef H2S_PRIO
#define H2S_PRIO 0
#endif
#ifndef H2S_TRACE
#define H2S_TRACE 0
#endif
// launch `id` (relu >> 8), workgroup, slot: 0 start, 1 P released, 2 loop done, 3 epilogue issued, 4 stores drained (MFMA wave 0);
// 5 / 6: weight / x wave at P
#define H2S_MARK(slot)                                                                                \
    do {                                                                                              \
        if (H2S_TRACE && g_h2s_trace && lane == 0)                                                    \
            g_h2s_trace[(((int64_t)(p.relu >> 8) * gridDim.x + blockIdx.x) * 8) + (slot)] = wall_clock64();   \
    } while (0)

namespace dc {
__device__ long long *g_h2s_dbg = nullptr;     // per workgroup: {core-clock cycles, 100 MHz ticks} of the main loop
__device__ long long *g_h2s_trace = nullptr;

using hs_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using hs_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hs_f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kSBM = 128, kSBN = 256, kSBK = 32;
constexpr int kSRow = 128;                              // bytes per LDS row (both operands): 8 pieces of 16 B
constexpr int kSSzA = kSBM * kSRow, kSSzB = kSBN * kSRow;
constexpr int kRX = H2S_RX, kRW = 3;
static_assert(kRW * kSSzB + kRX * kSSzA <= 160 * 1024, "LDS");

__device__ __forceinline__ int hs_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }
#define hs_waitvm(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))   // s_waitcnt vmcnt(n) only

template <bool FULL>
__global__ void __launch_bounds__(512)
k_fwd_h2d(FwdParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[kRW * kSSzB + kRX * kSSzA];
    char *const sB = lds, *const sA = lds + kRW * kSSzB;
    const unsigned ntn = (unsigned)((p.Fo + kSBN - 1) / kSBN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * kSBM, col0 = (int64_t)(lb % ntn) * kSBN;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nst = (int)(p.Fi / kSBK);
    const int64_t lda = p.x[0].ld;

    // epilogue, second half (all eight waves): the accumulators of the 128 x 256 tile sit in LDS row-major (1 KiB rows),
    // behind them the row factors [128] and the column factors / bias [256] each (written by the x waves, which loaded
    // them at the start of the launch); wave w finishes rows 16 w .. 16 w + 15: (acc * row factor) * column factor + bias,
    // ReLU, one 1-KiB global_store_dwordx4 wave-instruction per row
    constexpr int kOffRowF = kSBM * 1024, kOffColF = kOffRowF + kSBM * 4, kOffBias = kOffColF + kSBN * 4;
    static_assert(kOffBias + kSBN * 4 <= kRW * kSSzB + kRX * kSSzA, "epilogue tables");
    auto store_rows = [&]() {
        __syncthreads();
        const bool relu = (p.relu & 1) != 0;
        const hs_f32x4 icol = *reinterpret_cast<const hs_f32x4 *>(lds + kOffColF + 16 * lane);
        const hs_f32x4 bcol = *reinterpret_cast<const hs_f32x4 *>(lds + kOffBias + 16 * lane);
        const int64_t col = col0 + 4 * lane;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int rl = wid * 16 + j;
            const int64_t row = row0 + rl;
            const float sv = *reinterpret_cast<const float *>(lds + kOffRowF + 4 * rl);
            hs_f32x4 v = *reinterpret_cast<const hs_f32x4 *>(lds + rl * 1024 + 16 * lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = (v[i] * sv) * icol[i];
                t += bcol[i];
                v[i] = relu ? fmaxf(t, 0.f) : t;
            }
            if (FULL || (row < p.N && col < p.Fo)) *reinterpret_cast<hs_f32x4 *>(p.out + row * p.ldo + col) = v;
        }
    };

    if (wid >= 4) {
        // ------------------------------------------------------------------ loader waves: LDS-DMA only
        // one instruction fills 8 rows x 128 B of a tile (1 KiB, lane-linear): lane l is row 8 c + (l >> 3), position
        // l & 7, and fetches the piece that belongs there: q = position ^ swz(row) (the swizzle repeats every 16 rows:
        // one per-lane offset for the even chunks, one for the odd ones).  Rows past the operand's end fall out of
        // the buffer's range and arrive as zeros.
        const bool isx = wid >= 6;
        const int w = wid & 1;
        const int64_t ld = isx ? lda : p.Fi;
        const int64_t trows = isx ? (p.N - row0 < kSBM ? p.N - row0 : kSBM) : (p.Fo - col0 < kSBN ? p.Fo - col0 : kSBN);
        const float *base = isx ? p.x[0].p + row0 * lda : p.w[0].p + col0 * p.Fi;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(base), 0, (int)(((trows - 1) * ld + p.Fi) * 4), 0x00020000);
        int voff[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
            voff[e] = (int)(((int64_t)(lane >> 3) * ld) * 4 + 16 * ((lane & 7) ^ hs_swz(8 * e + (lane >> 3))));
        const int cstep = (int)(8 * ld * 4);           // bytes from one chunk's rows to the next chunk's
        if (isx) {
            // wave 6 also owns the epilogue's factors: 2 rows and 4 columns per lane, loaded now, written to LDS when the
            // rings are free
            float rowf[2] = {0.f, 0.f};
            hs_f32x4 colf = {0.f, 0.f, 0.f, 0.f}, biasv = {0.f, 0.f, 0.f, 0.f};
            if (w == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int64_t row = row0 + 2 * lane + i;
                    rowf[i] = h2_unscale(p.h2.a_rowmax[(FULL || row < p.N) ? row : p.N - 1]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t col = col0 + 4 * lane + i;
                    const int64_t colc = (FULL || col < p.Fo) ? col : p.Fo - 1;
                    colf[i] = h2_unscale(p.h2.b_rowmax[colc]);
                    biasv[i] = p.bias ? p.bias[colc] : 0.f;
                }
            }
            auto stage = [&](int s) {                  // 8 instructions per wave: chunks 8 w .. 8 w + 7 of 16
                if (H2S_ABL & 2) return;
                char *dst = sA + (s % kRX) * kSSzA;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 8 * w + j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3))) *)(dst + c * 1024), 16,
                                                             voff[j & 1], c * cstep + s * (kSBK * 4), 0, 0);
                }
            };
            // stage s sits in slot s % RX; it is issued at the top of iteration s - (RX - 1) and has landed at the end
            // of iteration s - 2: RX - 2 iterations in flight
#pragma unroll
            for (int s = 0; s < kRX - 1; ++s)
                if (s < nst) stage(s);
            if (nst > kRX - 1) hs_waitvm(8 * (kRX - 3)); else hs_waitvm(0);      // stages 0 and 1
            if (w == 0) H2S_MARK(6);
            __builtin_amdgcn_s_barrier();              // P
            int it = 0;
            for (; it + kRX - 1 < nst; ++it) {
                stage(it + kRX - 1);
                hs_waitvm(8 * (kRX - 3));              // stage it + 2
                if (!(H2S_ABL & 8)) __builtin_amdgcn_s_barrier();
            }
            for (; it < nst; ++it) {
                hs_waitvm(0);
                if (!(H2S_ABL & 8)) __builtin_amdgcn_s_barrier();
            }
            if (w == 0) {
                *reinterpret_cast<float2 *>(lds + kOffRowF + 8 * lane) = make_float2(rowf[0], rowf[1]);
                *reinterpret_cast<hs_f32x4 *>(lds + kOffColF + 16 * lane) = colf;
                *reinterpret_cast<hs_f32x4 *>(lds + kOffBias + 16 * lane) = biasv;
            }
        } else {
            auto stage = [&](int s) {                  // 16 instructions per wave: chunks 16 w .. 16 w + 15 of 32
                if (H2S_ABL & 32) return;
                char *dst = sB + (s % kRW) * kSSzB;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = 16 * w + j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3))) *)(dst + c * 1024), 16,
                                                             voff[j & 1], c * cstep + s * (kSBK * 4), 0, 0);
                }
            };
            stage(0);
            if (nst > 1) stage(1);
            hs_waitvm(0);
            if (w == 0) H2S_MARK(5);
            __builtin_amdgcn_s_barrier();              // P
            for (int it = 0; it < nst; ++it) {
                if (it + 2 < nst) stage(it + 2);
                hs_waitvm(0);
                if (!(H2S_ABL & 8)) __builtin_amdgcn_s_barrier();
            }
        }
        store_rows();
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves 0-3, 64 x 128 each
    if (wid == 0) H2S_MARK(0);
    if (H2S_PRIO) __builtin_amdgcn_s_setprio(H2S_PRIO);
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 31, fh = lane >> 5, fsw = hs_swz(fr);
    const int fragA = (wm * 64 + fr) * kSRow, fragB0 = (wn * 128 + fr) * kSRow;
    float scA[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int64_t row = row0 + wm * 64 + mb * 32 + fr;
        row = (FULL || row < p.N) ? row : p.N - 1;
        scA[mb] = h2_scale(p.h2.a_rowmax[row]);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
    // x fragment of k-step ks, row block mb: k = 16 ks + 8 fh .. + 7 of row fr = pieces 4 ks + 2 fh and + 1 (fp32)
    hs_f32x4 ra[2][2];                                 // raw, one k-step: [mb][piece]
    hs_f16x8 fa0[2][2], fa1[2][2], fb[4][2];           // x planes: two sets (k-steps alternate); weights: ONE set
    auto rawA = [&](int slot, int ks) {
        if (H2S_ABL & 4) return;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                ra[mb][h] = *reinterpret_cast<const hs_f32x4 *>(sA + slot * kSSzA + fragA + mb * 32 * kSRow +
                                                                16 * ((4 * ks + 2 * fh + h) ^ fsw));
    };
    auto fragB = [&](int nb, int slot, int ks) {
        if (H2S_ABL & 4) return;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            fb[nb][pl] = *reinterpret_cast<const hs_f16x8 *>(sB + slot * kSSzB + fragB0 + nb * 32 * kSRow +
                                                             16 * ((4 * ks + 2 * pl + fh) ^ fsw));
    };
    auto split = [&](hs_f16x8 (&fa)[2][2], int mb) {   // the scaled value's two fp16 planes, as dc_dense_h2w.hip's staging
        if (H2S_ABL & 4) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (H2S_ABL & 128) {
                const hs_f16x8 t = __builtin_bit_cast(hs_f16x8, ra[mb][h]);
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[mb][0][4 * h + i] = t[i], fa[mb][1][4 * h + i] = t[4 + i];
                continue;
            }
            const hs_f32x4 v = ra[mb][h] * scA[mb];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 a = (_Float16)v[i];
                fa[mb][0][4 * h + i] = a;
                fa[mb][1][4 * h + i] = (_Float16)(v[i] - (float)a);
            }
        }
    };
    // the 6 MFMAs of one 32-column block of a k-step: products h2*h1, h1*h2, h1*h1 (smallest terms first, as k_fwd_h2),
    // each on both row blocks - every accumulator sees its terms in k_fwd_h2w's order
    auto mma_nb = [&](const hs_f16x8 (&fa)[2][2], int nb) {
        if (H2S_ABL & 1) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[mb][nb][0] += (float)fa[mb][0][0] + (float)fb[nb][1][1] +
                                                             (float)fa[mb][1][2] + (float)fb[nb][0][3];
            return;
        }
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]], acc[mb][nb], 0, 0, 0);
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
                for (int e = 0; e < 8; ++e) fb[i][j][e] = (_Float16)(lane - e);
    }
    __syncthreads();                                   // P
    if (wid == 0) H2S_MARK(1);
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    rawA(0, 0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) fragB(nb, 0, 0);
    split(fa0, 0);
    split(fa0, 1);
    int sx = 0, sw = 0;                                // slots of stage it: it % RX, it % 3
    // One k-step = four blocks of 6 MFMAs.  The weight fragments of a column block are re-read for the NEXT k-step
    // right behind the block's MFMAs (one register set: 32 VGPRs instead of 64), the raw x fragments of the next
    // k-step are read in block 0 and split under blocks 2 and 3.
    auto sgb_mfma_valu = [&]() {                       // 6 MFMAs, about 5 VALU instructions of the split behind each
        if (H2S_ABL & 5) return;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    };
#define H2D_KSTEP(FA_CUR, FA_NEXT, SX_NEXT, SW_NEXT, KS_NEXT)                                         \
    {                                                                                                 \
        rawA(SX_NEXT, KS_NEXT);                                                                       \
        mma_nb(FA_CUR, 0);                                                                            \
        fragB(0, SW_NEXT, KS_NEXT);                                                                   \
        if (!(H2S_ABL & 5)) {                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                        \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        mma_nb(FA_CUR, 1);                                                                            \
        fragB(1, SW_NEXT, KS_NEXT);                                                                   \
        if (!(H2S_ABL & 5)) {                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                        \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        mma_nb(FA_CUR, 2);                                                                            \
        split(FA_NEXT, 0);                                                                            \
        fragB(2, SW_NEXT, KS_NEXT);                                                                   \
        sgb_mfma_valu();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        mma_nb(FA_CUR, 3);                                                                            \
        split(FA_NEXT, 1);                                                                            \
        fragB(3, SW_NEXT, KS_NEXT);                                                                   \
        sgb_mfma_valu();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    for (int it = 0; it + 1 < nst; ++it) {
        const int sx1 = sx + 1 == kRX ? 0 : sx + 1, sw1 = sw + 1 == kRW ? 0 : sw + 1;
        H2D_KSTEP(fa0, fa1, sx, sw, 1)
        H2D_KSTEP(fa1, fa0, sx1, sw1, 0)
        if (!(H2S_ABL & 8)) __syncthreads();
        sx = sx1, sw = sw1;
    }
    H2D_KSTEP(fa0, fa1, sx, sw, 1)                     // last stage
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) mma_nb(fa1, nb);
    if (!(H2S_ABL & 8)) __syncthreads();
#undef H2D_KSTEP
    if (wid == 0) H2S_MARK(2);
    if (H2S_PRIO) __builtin_amdgcn_s_setprio(0);
    if (g_h2s_dbg && threadIdx.x == 0) {
        g_h2s_dbg[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
        g_h2s_dbg[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }

    // epilogue, first half: C/D fragment (reg, lane) -> row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col lane & 31: the
    // accumulators as they are into LDS (the rings are free: the loop's last barrier is behind every wave)
    const int c = lane & 31, h = lane >> 5;
    float *const so = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                so[(wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 256 + wn * 128 + nb * 32 + c] = acc[mb][nb][r];
    store_rows();
    if (H2S_TRACE && wid == 0) {
        H2S_MARK(3);
        hs_waitvm(0);
        H2S_MARK(4);
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
        hipLaunchKernelGGL((k_fwd_h2d<true>), gd, bd, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_fwd_h2d<false>), gd, bd, 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
