// dc_dense_h2w.hip -- the forward-shaped fp16x2 dense block for the wide layers, 128 x 256 tiles (gfx950).
//
//   out[N,Fo] = act(x[N,K] . W[Fo,K]^T + b),  W handed over as the scaled, split image of
//   dc_tag_weight_prep (one 64-byte record {h1[16], h2[16]} per row and 16 k), x fp32 (hop slab).
//   Same arithmetic as k_fwd_h2 (dc_dense_h2.hip): products h2*h1 + h1*h2 + h1*h1 in that order, fp32
//   accumulate, k ascending in steps of 16 - the results are bit-identical to it.
//
// Why a second kernel (r02 ablations of k_fwd_h2 on cold operands, tools/exp/dense_abl.py): that kernel
// takes 81 us for the soft layer-2 block; with its MFMAs removed it still takes 67 us, with all global
// traffic removed 52 us.  Neither half is near the 20.6 us the matrix pipe needs:
//   * data movement: 64-byte row pieces (half cache lines), one 16-wide stage of prefetch that the
//     per-stage vmcnt(0) drains - 16 KB in flight per CU, i.e. a latency-bound 2 TB/s stream of x;
//   * the loop itself: 12 MFMAs (384 cycles) per wave between barriers against ~600 cycles of fragment-read
//     latency and barrier skew.
// A third cause remains after both are fixed (profiles/r02/l_dense_ablation.txt): every 128-row tile streams the
// whole 1 MB weight image from L2 - 390 MB of L2 -> CU traffic per launch, 39 us when the loop only loads.
// Here: BK = 32 (x rows move as full 128-byte lines, 24 MFMAs per wave and barrier), one 128 x 256 tile
// per 512-thread workgroup (x is staged and split ONCE for all 256 output columns instead of once per
// 128), both operands staged through REGISTERS - no LDS-DMA in the kernel, so hipcc's waits stay counted
// and __syncthreads() is a bare s_barrier: the loads of stage it+3 are issued at the top of stage it (two
// register sets) and written during stage it+1 into a ring of three LDS stage buffers, so that stage it+1 is
// complete one barrier before it is computed and its first fragments can be read under the last MFMAs of
// stage it (two fragment register sets).  LDS rows are 128 bytes = 8 pieces of 16 bytes (piece q = 4*kstep
// + 2*plane + half); piece q of row r sits at position q ^ F(r), F(r) = ((r >> 1) & 7) ^ 2*(r & 1):
// conflict-free ds_read_b128 fragment reads (each 16-lane group sees 8 even and 8 odd rows with 8 distinct
// F each) and conflict-free ds_write_b64 of the split x (two adjacent rows per 16-lane group land on
// disjoint pieces).
#include "dc_dense.h"

namespace dc {

using hw_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using hw_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using hw_f32x4 = __attribute__((ext_vector_type(4))) float;
using hw_u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kWBM = 128, kWBN = 256, kWBK = 32;
constexpr int kWRow = 128;                              // bytes per LDS row (both operands)
constexpr int kWSzA = kWBM * kWRow, kWSzB = kWBN * kWRow;

__device__ __forceinline__ int hw_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }

// CORR: the left operand is x - x2_coef[row] * x2 (dc_tag_linear_fwd_h2p_corr); a template parameter so that the plain
// kernel keeps its register allocation
template <bool FULL, bool CORR = false>
__global__ void __launch_bounds__(512)
k_fwd_h2w(FwdParams p) {
    __shared__ __attribute__((aligned(16))) char sA[3 * kWSzA];      // ring of three stages: 144 KB
    __shared__ __attribute__((aligned(16))) char sB[3 * kWSzB];
    __shared__ __attribute__((aligned(16))) float s_inv[kWBM];
    const unsigned ntn = (unsigned)((p.Fo + kWBN - 1) / kWBN);
    // split reduction (p.ksplit > 1: a long K with too few output tiles for the chip, e.g. attention weights x
    // values): block = (tile, range kz); every range writes its own partial, summed in range order afterwards
    const unsigned ks = p.ksplit > 1 ? (unsigned)p.ksplit : 1u, ntiles = gridDim.x / ks;
    const unsigned lbb = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned lb = lbb % ntiles, kz = lbb / ntiles;
    const int64_t row0 = (int64_t)(lb / ntn) * kWBM, col0 = (int64_t)(lb % ntn) * kWBN;
    const int wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
    const int lane = threadIdx.x & 63;
    const int k8 = threadIdx.x & 7, r = threadIdx.x >> 3;          // staging: 8 threads per 128-byte row
    const int64_t lda = p.x[0].ld;
    // grouped launch (block-diagonal union, dc_tag_grouped_fwd_h2p): the tile's group picks weights / bias / scales
    const float *wimg = p.w[0].p, *bias = p.bias, *brm = p.h2.b_rowmax;
    int64_t data_end = p.N;                                   // rows at and behind it are a group's zero padding
    if (p.grp.n >= 1) {
        const int g = group_of_row(p.grp.row_beg, p.grp.n, row0);
        wimg = p.grp.w[g], bias = p.grp.bias[g], brm = p.grp.b_rowmax[g];
        data_end = p.grp.row_end[g];
    }

    unsigned offA[2], offB[4];
    int ldsAh[2], ldsAl[2], ldsB[4];
    float scA[2], cA[2] = {0.f, 0.f};               // cA: coefficient of the correction operand x2 for this thread's rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = r + 64 * j;
        int64_t row = row0 + rl;
        row = (FULL || row < p.N) ? row : p.N - 1;
        offA[j] = (unsigned)((row - row0) * lda + 4 * k8);
        // this thread's 4 k: k-step k8 >> 2, half (k8 >> 1) & 1, 8-byte slot k8 & 1 of the piece
        const int q = 4 * (k8 >> 2) + ((k8 >> 1) & 1), f = hw_swz(rl);
        ldsAh[j] = rl * kWRow + 16 * (q ^ f) + 8 * (k8 & 1);
        ldsAl[j] = rl * kWRow + 16 * ((q + 2) ^ f) + 8 * (k8 & 1);
        const float m = p.h2.a_rowmax[row];
        scA[j] = h2_scale(m);
        if (k8 == 0) s_inv[rl] = h2_unscale(m);
        if (CORR) cA[j] = p.x2_coef[row];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = r + 64 * j;
        int64_t col = col0 + rl;
        col = (FULL || col < p.Fo) ? col : p.Fo - 1;
        offB[j] = (unsigned)((col - col0) * p.Fi + 4 * k8);           // 4-byte units, 16 bytes per piece
        ldsB[j] = rl * kWRow + 16 * (k8 ^ hw_swz(rl));
    }

    f32x16 acc[2][2];
    zero_acc<2>(acc);
    const int nst_all = (int)(p.Fi / kWBK), per = (nst_all + (int)ks - 1) / (int)ks;
    const int st_beg = (int)kz * per;
    const int nst = st_beg >= nst_all ? 0 : (nst_all - st_beg < per ? nst_all - st_beg : per);
    const float *baseA = p.x[0].p + row0 * lda + (int64_t)st_beg * kWBK;      // wave-uniform running bases
    const float *baseB = wimg + col0 * p.Fi + (int64_t)st_beg * kWBK;
    const float *baseA2 = CORR ? p.x2 + row0 * lda + (int64_t)st_beg * kWBK : nullptr;
    hw_f32x4 va0[2], va1[2];                                          // two register sets, named: no runtime index
    hw_f32x4 vc0[2], vc1[2];                                          // (CORR) the correction operand's pieces
    hw_u32x4 vb0[4], vb1[4];

    auto gload_set = [&](hw_f32x4 (&va)[2], hw_f32x4 (&vc)[2], hw_u32x4 (&vb)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            va[j] = *reinterpret_cast<const hw_f32x4 *>(baseA + offA[j]);
        if (CORR) {                                                       // (combined at LDS-store time: stays in flight)
#pragma unroll
            for (int j = 0; j < 2; ++j) vc[j] = *reinterpret_cast<const hw_f32x4 *>(baseA2 + offA[j]);
            baseA2 += kWBK;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j] = *reinterpret_cast<const hw_u32x4 *>(baseB + offB[j]);
        baseA += kWBK;
        baseB += kWBK;
    };
    auto lstore_b = [&](const hw_u32x4 (&vb)[4], int b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<hw_u32x4 *>(sB + b * kWSzB + ldsB[j]) = vb[j];
    };
    auto lstore_a = [&](const hw_f32x4 (&va)[2], const hw_f32x4 (&vc)[2], int b) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const hw_f32x4 v = (CORR ? va[j] - vc[j] * cA[j] : va[j]) * scA[j];
            hw_f16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 a = (_Float16)v[i];
                h[i] = a;
                l[i] = (_Float16)(v[i] - (float)a);
            }
            *reinterpret_cast<hw_f16x4 *>(sA + b * kWSzA + ldsAh[j]) = h;
            *reinterpret_cast<hw_f16x4 *>(sA + b * kWSzA + ldsAl[j]) = l;
        }
    };
    auto gload = [&](int set) {
        if (set == 0) gload_set(va0, vc0, vb0); else gload_set(va1, vc1, vb1);
    };
    auto lstoreB = [&](int set, int b) {
        if (set == 0) lstore_b(vb0, b); else lstore_b(vb1, b);
    };
    auto lstoreA = [&](int set, int b) {
        if (set == 0) lstore_a(va0, vc0, b); else lstore_a(va1, vc1, b);
    };
    auto lstore = [&](int set, int b) {
        lstoreB(set, b);
        lstoreA(set, b);
    };
    const int fr = lane & 31, fh = lane >> 5, fsw = hw_swz(fr);
    const int fragA = (wm * 64 + fr) * kWRow, fragB = (wn * 64 + fr) * kWRow;
    // fragment registers: two sets (one k-step each), so that the reads of the next k-step - also the first
    // one of the NEXT stage, whose buffer has been complete since the previous barrier - are in flight
    // under the MFMAs of the current one and no MFMA waits on LDS latency behind a barrier
    hw_f16x8 fa0[2][2], fb0[2][2], fa1[2][2], fb1[2][2];
    auto frags = [&](hw_f16x8 (&fa)[2][2], hw_f16x8 (&fb)[2][2], int b, int ks) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fa[mb][pl] = *reinterpret_cast<const hw_f16x8 *>(sA + b * kWSzA + fragA + mb * 32 * kWRow +
                                                                 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fb[nb][pl] = *reinterpret_cast<const hw_f16x8 *>(sB + b * kWSzB + fragB + nb * 32 * kWRow +
                                                                 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
    };
    auto mma = [&](const hw_f16x8 (&fa)[2][2], const hw_f16x8 (&fb)[2][2]) {
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};       // smallest terms first (as k_fwd_h2)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]],
                                                                         acc[mb][nb], 0, 0, 0);
    };
    auto nextb = [](int b) { return b == 2 ? 0 : b + 1; };

    // Stage s lives in LDS buffer s % 3.  During stage it: the loads of stage it+3 are issued (register set
    // it & 1 ... see below), stage it+2 is written to LDS (its loads were issued at the top of stage it-1),
    // the barrier at the end of stage it makes it readable from the end of stage it+1 on.
    //   register set of stage s: s & 1; loads(s) at the top of stage s-3, store(s) at the end of stage s-2
    if (nst > 0) gload(0);                             // stage 0 (nst == 0: an empty range of a split reduction)
    if (nst > 1) gload(1);                             // stage 1
    lstore(0, 0);
    if (nst > 2) gload(0);                             // stage 2
    if (nst > 1) lstore(1, 1);
    __syncthreads();
    frags(fa0, fb0, 0, 0);
    int it = 0, cur = 0;                               // cur = it % 3
    // steady state: two stages per trip (static register sets); needs stages it+3 and it+4 to exist
    // Inside a stage the order is pinned in three regions (hipcc schedules within a region only; left to itself
    // it hoists the split's first multiplies - and with them the vmcnt wait for the staged tile - to the top
    // of the stage and bunches the fragment reads right in front of the MFMAs that need them):
    //   loads of stage it+3 | reads k-step 1, 12 MFMAs k-step 0, B tile of stage it+2 -> LDS |
    //   reads k-step 0 of stage it+1, 12 MFMAs k-step 1, split + A tile of stage it+2 -> LDS | barrier
#define DC_H2W_STAGE(SET_LOAD, SET_STORE)                                                             \
    {                                                                                                 \
        const int b1 = nextb(cur), b2 = nextb(b1);                                                    \
        gload(SET_LOAD);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        frags(fa1, fb1, cur, 1);                                                                      \
        mma(fa0, fb0);                                                                                \
        lstoreB(SET_STORE, b2);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        frags(fa0, fb0, b1, 0);                                                                       \
        mma(fa1, fb1);                                                                                \
        lstoreA(SET_STORE, b2);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        __syncthreads();                                                                              \
        cur = b1;                                                                                     \
    }
    for (; it + 4 < nst; it += 2) {
        DC_H2W_STAGE(1, 0)                             // loads stage it+3, stores stage it+2
        DC_H2W_STAGE(0, 1)                             // loads stage it+4, stores stage it+3
    }
#undef DC_H2W_STAGE
    for (; it < nst; ++it) {                           // last stages (at most four)
        const int b1 = nextb(cur), b2 = nextb(b1);
        const bool even = (it & 1) == 0;
        if (it + 3 < nst) {
            if (even) gload(1); else gload(0);
        }
        frags(fa1, fb1, cur, 1);
        mma(fa0, fb0);
        if (it + 1 < nst) frags(fa0, fb0, b1, 0);
        mma(fa1, fb1);
        if (it + 2 < nst) {
            if (even) lstore(0, b2); else lstore(1, b2);
        }
        __syncthreads();
        cur = b1;
    }

    // epilogue: C/D fragment (reg, lane) -> row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col lane & 31
    const bool relu = p.relu != 0;
    const int c = lane & 31, h = lane >> 5;
    if (ks > 1) {                                       // split reduction: plain partial of this range
        float *part = p.kpartial + (int64_t)kz * p.N * p.Fo;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int64_t col = col0 + wn * 64 + nb * 32 + c;
            const bool cok = FULL || col < p.Fo;
            const float icol = h2_unscale(brm[cok ? col : p.Fo - 1]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rl = wm * 64 + mb * 32 + 8 * g + 4 * h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t row = row0 + rl + i;
                        if (FULL || (cok && row < p.N))
                            part[row * p.Fo + col] = (acc[mb][nb][4 * g + i] * s_inv[rl + i]) * icol;
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + wn * 64 + nb * 32 + c;
        const bool cok = FULL || col < p.Fo;
        const int64_t colc = cok ? col : p.Fo - 1;
        const float bcol = bias ? bias[colc] : 0.f;
        const float icol = h2_unscale(brm[colc]);
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
                    if (p.exp_lse)                  // attention recompute: the block's weights from the saved row lse
                        v = col < p.exp_ncols ? expf(v - p.exp_lse[(FULL || row < p.N) ? row : p.N - 1]) : 0.f;
                    if (row >= data_end) v = 0.f;            // padding rows of a grouped launch stay zero rows
                    if (FULL || (cok && row < p.N)) p.out[row * p.ldo + col] = v;
                }
            }
    }
}

static inline bool hw_al16(const void *q) { return ((uintptr_t)q & 15) == 0; }
static inline int hw_env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// eligible: one K segment, pre-split weights, K % 32 == 0, no split reduction, enough row tiles to give
// every CU one (a 128 x 256 tile is a whole CU's work: small N stays with the 64/128 x 128 kernel)
bool fwd_h2w_launch(const FwdParams &p, hipStream_t hs) {
    static const int wide = hw_env_int("DC_H2_WIDE", 1);
    if (!p.h2.a_rowmax || !p.h2.b_rowmax || !p.h2.b_presplit || p.nseg != 1) return false;
    if (!wide && p.grp.n < 1 && !p.x2) return false;
    const int64_t ks = p.ksplit > 1 ? p.ksplit : 1;
    if (ks > 1 && (!p.kpartial || p.bias || p.relu || p.exp_lse)) return false;
    if (p.Fi % kWBK != 0 || p.Fi < kWBK) return false;
    if (p.x[0].ld * kWBM >= ((int64_t)1 << 30) || p.Fi * kWBN >= ((int64_t)1 << 30)) return false;
    if (!hw_al16(p.x[0].p) || !hw_al16(p.w[0].p) || p.x[0].ld % 4 != 0) return false;
    const int64_t tiles = ((p.N + kWBM - 1) / kWBM) * ((p.Fo + kWBN - 1) / kWBN);
    static const int min_tiles = hw_env_int("DC_H2_WIDE_MIN_TILES", 128);
    if ((tiles * ks < min_tiles && p.grp.n < 1 && !p.x2) || tiles * ks >= (int64_t)INT32_MAX) return false;
    if (fwd_h2d_launch(p, hs)) return true;            // both operands by LDS-DMA, waves split by role (dc_dense_h2d.hip)
    const dim3 gd((unsigned)(tiles * ks)), bd(512);
    if (p.x2) {
        if (!p.x2_coef || ks > 1 || !hw_al16(p.x2)) return false;
        if (p.N % kWBM == 0 && p.Fo % kWBN == 0)
            DC_LAUNCH((k_fwd_h2w<true, true>), gd, bd, 0, hs, p);
        else
            DC_LAUNCH((k_fwd_h2w<false, true>), gd, bd, 0, hs, p);
        return true;
    }
    if (p.N % kWBM == 0 && p.Fo % kWBN == 0)
        DC_LAUNCH((k_fwd_h2w<true>), gd, bd, 0, hs, p);
    else
        DC_LAUNCH((k_fwd_h2w<false>), gd, bd, 0, hs, p);
    return true;
}

}  // namespace dc
