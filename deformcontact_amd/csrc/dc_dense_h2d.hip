// dc_dense_h2d.hip -- the forward-shaped fp16x2 dense block for the wide layers, 128 x 256 tiles, both operands
// by LDS-DMA and the workgroup's eight waves split by role (gfx950).
//
//   out[N,Fo] = act(x[N,K] . W[Fo,K]^T + b),  W handed over as the scaled, split image of dc_tag_weight_prep,
//   x fp32 (hop slab).  Products, their order and the k order are those of k_fwd_h2w (dc_dense_h2w.hip): results
//   are bit-identical to it (tests/test_wide_dense.py).
//
// Why a third form (round 4, tools/exp/dense_h2s.hip -> dense_h2r.hip -> dense_h2d.hip, profiles/r04):
//   * k_fwd_h2w has every wave load, split, store, read fragments and issue MFMAs, two waves per SIMD in step
//     behind one barrier per stage; its matrix pipe is busy 31 % of a launch.  With MFMA waves and loading waves
//     apart, the MFMA waves run at 86 % of back-to-back issue - but a wave that stages x through REGISTERS beside
//     them gets its loads issued at a third of the usual rate (1.5 us for 8 loads per stage, with or without
//     barriers: x alone streams in 24 us, the MFMAs alone take 34 us, together 65).  Brought in by LDS-DMA (no VGPR
//     write port involved) the same bytes arrive in the MFMAs' shadow.
//   * So: waves 0-3 (one per SIMD) read fp32 x fragments and weight fragments from LDS, scale and split x into its two
//     fp16 planes in registers (VALU dealt out between the MFMAs) and issue the MFMAs, wave tile 64 x 128;
//     waves 4-5 bring the weight tile (32 KB per stage, L2-resident) in by LDS-DMA, ring of 3 stages;
//     waves 6-7 bring the x tile (16 KB per stage, fp32 as it is) in by LDS-DMA, ring of 4 stages = two stages of
//     HBM / Infinity Cache latency in flight.  160 KB of LDS, one workgroup per CU.
//   * The epilogue costs 7.6 us as 4-byte stores from the accumulator layout (two 128-byte runs per instruction);
//     here the accumulators go through the (now free) LDS and leave as one 1-KiB row per store instruction, the row
//     / column factors and the bias staged beside them by a loading wave that fetched them at the start: 3.2 us.
// LDS images: rows of 128 bytes = 8 pieces of 16 bytes, piece q of row r at position q ^ F(r) (F as in
// dc_dense_h2w.hip: conflict-free ds_read_b128 fragment reads); an LDS-DMA instruction writes 1 KiB lane-linear
// (8 rows), so the swizzle is applied to the per-lane SOURCE address.  x pieces: q = k / 4 (fp32); weight pieces:
// q = 4 * kstep + 2 * plane + half (dc_tag_weight_prep's record order).
#include "dc_dense.h"

namespace dc {

using hd_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hd_f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kDBM = 128, kDBN = 256, kDBK = 32;
constexpr int kDRow = 128;                              // bytes per LDS row (both operands)
constexpr int kDSzA = kDBM * kDRow, kDSzB = kDBN * kDRow;
constexpr int kDRX = 4, kDRW = 3;                       // ring depths: x, weights
constexpr int kDLds = kDRW * kDSzB + kDRX * kDSzA;      // 160 KB
// epilogue image: [128][256] accumulators, then row factors [128], column factors [256], bias [256]
constexpr int kDOffRowF = kDBM * 1024, kDOffColF = kDOffRowF + kDBM * 4, kDOffBias = kDOffColF + kDBN * 4;
static_assert(kDLds <= 160 * 1024 && kDOffBias + kDBN * 4 <= kDLds, "LDS budget");

__device__ __forceinline__ int hd_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }
#define DC_HD_WAITVM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))   // s_waitcnt vmcnt(n) only

template <bool FULL>
__global__ void __launch_bounds__(512)
k_fwd_h2d(FwdParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[kDLds];
    char *const sB = lds, *const sA = lds + kDRW * kDSzB;
    const unsigned ntn = (unsigned)((p.Fo + kDBN - 1) / kDBN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * kDBM, col0 = (int64_t)(lb % ntn) * kDBN;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nst = (int)(p.Fi / kDBK);
    const int64_t lda = p.x[0].ld;
    // grouped launch (block-diagonal union, dc_tag_grouped_fwd_h2p): the tile's group picks weights / bias / scales
    const float *wimg = p.w[0].p, *bias = p.bias, *brm = p.h2.b_rowmax;
    int64_t data_end = p.N;                                   // rows at and behind it are a group's zero padding
    if (p.grp.n >= 1) {
        const int g = group_of_row(p.grp.row_beg, p.grp.n, row0);
        wimg = p.grp.w[g], bias = p.grp.bias[g], brm = p.grp.b_rowmax[g];
        data_end = p.grp.row_end[g];
    }

    // epilogue, second half (all eight waves): the accumulators of the tile sit in LDS row-major (1 KiB rows), behind
    // them the row factors, the column factors and the bias; wave w finishes rows 16 w .. 16 w + 15 - (acc * row
    // factor) * column factor + bias, ReLU: k_fwd_h2w's operations in its order - one 1-KiB store instruction per row
    auto store_rows = [&]() {
        __syncthreads();
        const bool relu = p.relu != 0;
        const hd_f32x4 icol = *reinterpret_cast<const hd_f32x4 *>(lds + kDOffColF + 16 * lane);
        const hd_f32x4 bcol = *reinterpret_cast<const hd_f32x4 *>(lds + kDOffBias + 16 * lane);
        const int64_t col = col0 + 4 * lane;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int rl = wid * 16 + j;
            const int64_t row = row0 + rl;
            const float sv = *reinterpret_cast<const float *>(lds + kDOffRowF + 4 * rl);
            hd_f32x4 v = *reinterpret_cast<const hd_f32x4 *>(lds + rl * 1024 + 16 * lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = (v[i] * sv) * icol[i];
                t += bcol[i];
                if (relu) t = fmaxf(t, 0.f);
                v[i] = row >= data_end ? 0.f : t;             // padding rows of a grouped launch stay zero rows
            }
            if (FULL || (row < p.N && col < p.Fo)) *reinterpret_cast<hd_f32x4 *>(p.out + row * p.ldo + col) = v;
        }
    };

    if (wid >= 4) {
        // ------------------------------------------------------------------ loading waves: LDS-DMA only
        // one instruction fills 8 rows x 128 B of a tile: lane l is row 8 c + (l >> 3), position l & 7, and fetches the
        // piece that belongs there, q = position ^ F(row) (F repeats every 16 rows: one per-lane offset for the even
        // chunks, one for the odd ones).  Rows past the operand's end fall out of the buffer's range: zeros.
        const bool isx = wid >= 6;
        const int w = wid & 1;
        const int64_t ld = isx ? lda : p.Fi;
        const int64_t trows = isx ? (p.N - row0 < kDBM ? p.N - row0 : kDBM) : (p.Fo - col0 < kDBN ? p.Fo - col0 : kDBN);
        const float *base = isx ? p.x[0].p + row0 * lda : wimg + col0 * p.Fi;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(base), 0, (int)(((trows - 1) * ld + p.Fi) * 4), 0x00020000);
        int voff[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
            voff[e] = (int)(((int64_t)(lane >> 3) * ld) * 4 + 16 * ((lane & 7) ^ hd_swz(8 * e + (lane >> 3))));
        const int cstep = (int)(8 * ld * 4);           // bytes from one chunk's rows to the next chunk's
        if (isx) {
            // wave 6 also owns the epilogue's factors: 2 rows and 4 columns per lane, fetched now, written to LDS once
            // the rings are free
            float rowf[2] = {0.f, 0.f};
            hd_f32x4 colf = {0.f, 0.f, 0.f, 0.f}, biasv = {0.f, 0.f, 0.f, 0.f};
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
                    colf[i] = h2_unscale(brm[colc]);
                    biasv[i] = bias ? bias[colc] : 0.f;
                }
            }
            auto stage = [&](int s) {                  // 8 instructions per wave: chunks 8 w .. 8 w + 7 of 16
                char *dst = sA + (s % kDRX) * kDSzA;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 8 * w + j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3))) *)(dst + c * 1024), 16,
                                                             voff[j & 1], c * cstep + s * (kDBK * 4), 0, 0);
                }
            };
            // stage s sits in slot s % 4; it is issued at the top of iteration s - 3 and has landed at the end of
            // iteration s - 2 (vmcnt retires in order: all but the newest stage's 8 instructions)
#pragma unroll
            for (int s = 0; s < kDRX - 1; ++s)
                if (s < nst) stage(s);
            if (nst > kDRX - 1) DC_HD_WAITVM(8 * (kDRX - 3)); else DC_HD_WAITVM(0);      // stages 0 and 1
            __builtin_amdgcn_s_barrier();              // P
            int it = 0;
            for (; it + kDRX - 1 < nst; ++it) {
                stage(it + kDRX - 1);
                DC_HD_WAITVM(8 * (kDRX - 3));          // stage it + 2
                __builtin_amdgcn_s_barrier();
            }
            for (; it < nst; ++it) {
                DC_HD_WAITVM(0);
                __builtin_amdgcn_s_barrier();
            }
            if (w == 0) {
                *reinterpret_cast<float2 *>(lds + kDOffRowF + 8 * lane) = make_float2(rowf[0], rowf[1]);
                *reinterpret_cast<hd_f32x4 *>(lds + kDOffColF + 16 * lane) = colf;
                *reinterpret_cast<hd_f32x4 *>(lds + kDOffBias + 16 * lane) = biasv;
            }
        } else {
            auto stage = [&](int s) {                  // 16 instructions per wave: chunks 16 w .. 16 w + 15 of 32
                char *dst = sB + (s % kDRW) * kDSzB;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int c = 16 * w + j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3))) *)(dst + c * 1024), 16,
                                                             voff[j & 1], c * cstep + s * (kDBK * 4), 0, 0);
                }
            };
            stage(0);
            if (nst > 1) stage(1);
            DC_HD_WAITVM(0);
            __builtin_amdgcn_s_barrier();              // P
            for (int it = 0; it < nst; ++it) {
                if (it + 2 < nst) stage(it + 2);       // into the slot stage it - 1 left at the last barrier
                DC_HD_WAITVM(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        store_rows();
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves 0-3, 64 x 128 each
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 31, fh = lane >> 5, fsw = hd_swz(fr);
    const int fragA = (wm * 64 + fr) * kDRow, fragB0 = (wn * 128 + fr) * kDRow;
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
    hd_f32x4 ra[2][2];                                 // raw, one k-step: [mb][piece]
    hd_f16x8 fa0[2][2], fa1[2][2], fb[4][2];           // x planes: two sets (k-steps alternate); weights: ONE set
    auto rawA = [&](int slot, int ks) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                ra[mb][h] = *reinterpret_cast<const hd_f32x4 *>(sA + slot * kDSzA + fragA + mb * 32 * kDRow +
                                                                16 * ((4 * ks + 2 * fh + h) ^ fsw));
    };
    auto fragB = [&](int nb, int slot, int ks) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            fb[nb][pl] = *reinterpret_cast<const hd_f16x8 *>(sB + slot * kDSzB + fragB0 + nb * 32 * kDRow +
                                                             16 * ((4 * ks + 2 * pl + fh) ^ fsw));
    };
    auto split = [&](hd_f16x8 (&fa)[2][2], int mb) {   // the scaled value's two fp16 planes, as k_fwd_h2w's staging
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const hd_f32x4 v = ra[mb][h] * scA[mb];
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
#ifdef DC_H2D_ABL_MFMA16
    // timing-only ablation (wrong results by construction; tools/r06/mfma16_abl.sh): every 32x32x16 MFMA replaced by two
    // 16x16x32 ones on the same operand registers and two quarters of the same accumulator - same FLOPs, same LDS reads, same
    // VALU: does the chip hold a higher clock on the smaller MFMA shape inside THIS loop (MI355X_MICROARCH.md, DVFS item 7)?
    hd_f32x4 acc4[2][4][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc4[mb][nb][q] = hd_f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma_nb = [&](const hd_f16x8 (&fa)[2][2], int nb) {
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                acc4[mb][nb][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mb][pa[t]], fb[nb][pb[t]], acc4[mb][nb][t], 0, 0, 0);
                acc4[mb][nb][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mb][pb[t]], fb[nb][pa[t]], acc4[mb][nb][3], 0, 0, 0);
            }
    };
#else
    auto mma_nb = [&](const hd_f16x8 (&fa)[2][2], int nb) {
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]], acc[mb][nb], 0, 0, 0);
    };
#endif
    __syncthreads();                                   // P: stages 0 and 1 of both operands have landed
    rawA(0, 0);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) fragB(nb, 0, 0);
    split(fa0, 0);
    split(fa0, 1);
    int sx = 0, sw = 0;                                // slots of stage it: it % 4, it % 3
    // One k-step = four blocks of 6 MFMAs.  The weight fragments of a column block are re-read for the NEXT k-step
    // right behind the block's MFMAs (one register set: 32 VGPRs, not 64), the raw x fragments of the next k-step are
    // read in block 0 and split under blocks 2 and 3, about 5 VALU instructions behind each MFMA.  hipcc is held to
    // this order (sched_group_barrier); left alone it clumps the 55 VALU instructions of a split with no MFMA between.
    auto sgb_mfma_valu = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    };
#define DC_H2D_KSTEP(FA_CUR, FA_NEXT, SX_NEXT, SW_NEXT, KS_NEXT)                                      \
    {                                                                                                 \
        rawA(SX_NEXT, KS_NEXT);                                                                       \
        mma_nb(FA_CUR, 0);                                                                            \
        fragB(0, SW_NEXT, KS_NEXT);                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        mma_nb(FA_CUR, 1);                                                                            \
        fragB(1, SW_NEXT, KS_NEXT);                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                            \
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
        const int sx1 = sx + 1 == kDRX ? 0 : sx + 1, sw1 = sw + 1 == kDRW ? 0 : sw + 1;
        DC_H2D_KSTEP(fa0, fa1, sx, sw, 1)
        DC_H2D_KSTEP(fa1, fa0, sx1, sw1, 0)            // first fragments of stage it + 1: there since the last barrier
        __syncthreads();
        sx = sx1, sw = sw1;
    }
    DC_H2D_KSTEP(fa0, fa1, sx, sw, 1)                  // last stage
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) mma_nb(fa1, nb);
    __syncthreads();
#undef DC_H2D_KSTEP

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
#ifdef DC_H2D_ABL_MFMA16
                so[(wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 256 + wn * 128 + nb * 32 + c] = acc4[mb][nb][r >> 2][r & 3];
#else
                so[(wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 256 + wn * 128 + nb * 32 + c] = acc[mb][nb][r];
#endif
    store_rows();
}

static inline bool hd_al16(const void *q) { return ((uintptr_t)q & 15) == 0; }

// eligible: what fwd_h2w_launch takes, minus the split reduction, the exp epilogue and the correction operand (those stay
// with k_fwd_h2w), plus 16-byte rows on the output side
bool fwd_h2d_launch(const FwdParams &p, hipStream_t hs) {
    static const int on = [] {
        const char *v = getenv("DC_H2_DMA");
        return (v && *v) ? atoi(v) : 1;
    }();
    if (!on || p.ksplit > 1 || p.exp_lse || p.x2) return false;
    if (!p.h2.a_rowmax || !p.h2.b_rowmax || !p.h2.b_presplit || p.nseg != 1) return false;
    // (a short reduction does not amortise the two-stage prologue and the LDS epilogue: K = 96 ran 17.4 us against 14.3 on
    // k_fwd_h2w; from eight stages on this form is at least as fast - unless forced for the tests' small shapes)
    static const int min_k = [] {
        const char *v = getenv("DC_H2_DMA_MIN_K");
        return (v && *v) ? atoi(v) : 8 * kDBK;
    }();
    if (p.Fi % kDBK != 0 || p.Fi < kDBK || p.Fi < min_k || p.Fo % 4 != 0 || p.ldo % 4 != 0 || p.x[0].ld % 4 != 0) return false;
    if (!hd_al16(p.x[0].p) || !hd_al16(p.w[0].p) || !hd_al16(p.out)) return false;
    for (int g = 0; g < p.grp.n; ++g)
        if (!hd_al16(p.grp.w[g])) return false;
    // buffer descriptors: 32-bit byte counts per tile
    if (p.x[0].ld * kDBM >= ((int64_t)1 << 28) || p.Fi * kDBN >= ((int64_t)1 << 28)) return false;
    const int64_t tiles = ((p.N + kDBM - 1) / kDBM) * ((p.Fo + kDBN - 1) / kDBN);
    if (tiles >= (int64_t)INT32_MAX) return false;
    const dim3 gd((unsigned)tiles), bd(512);
    if (p.N % kDBM == 0 && p.Fo % kDBN == 0)
        DC_LAUNCH((k_fwd_h2d<true>), gd, bd, 0, hs, p);
    else
        DC_LAUNCH((k_fwd_h2d<false>), gd, bd, 0, hs, p);
    return true;
}

}  // namespace dc
