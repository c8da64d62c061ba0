// dc_dense_h2.hip -- the forward-shaped fp16x2 dense block, tuned main loop (gfx950).
//
//   out[N,Fo] = act(sum_s xs[s][N,Fi] . ws[s][Fo,Fi]^T + b)      (also the dX block of the backward:
//   xs = hop slab of the masked gradient, ws = transposed weights, see dc_dense_split.hip)
//
// Same arithmetic, LDS layout and epilogue as k_fwd_split<MB, 2> (two power-of-two-scaled fp16
// planes per operand, products h1*h1 + h1*h2 + h2*h1, 80-byte LDS rows); what differs is the
// instruction stream of the steady-state loop, which PMC showed to be the limiter (matrix pipe
// 31 % busy, VALU and MFMA co-executing in only 6 % of the MFMA-busy cycles, ~100 VALU per stage
// and wave):
//   * ONE K segment (the host hands over the hop slab whole and the weights concatenated along K by
//     dc_tag_weight_prep), so operand addresses are a wave-uniform base (SGPR pair, bumped by scalar
//     adds) plus a constant 32-bit per-thread offset, instead of per-thread 64-bit pointers bumped
//     with 64-bit VALU adds and re-based at segment switches;
//   * the steady-state body has no branches (the last two stages are peeled), so the whole stage is
//     ONE basic block and hipcc interleaves the split VALU of stage it+1 with the MFMAs of stage it
//     instead of emitting them as two phases.
#include "dc_dense.h"

namespace dc {

using h2_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using h2_f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kH2Plane = 32;                      // bytes per plane inside an LDS row (16 fp16)
constexpr int kH2Row = 2 * kH2Plane + 16;         // 80 B: conflict-free b64 stores / b128 reads

__device__ __forceinline__ void h2_split_store(char *dst, float4 x, float s) {
    x = make_float4(x.x * s, x.y * s, x.z * s, x.w * s);
    h2_f16x4 h, l;
    const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 a = (_Float16)v[i];
        h[i] = a;
        l[i] = (_Float16)(v[i] - (float)a);
    }
    *reinterpret_cast<h2_f16x4 *>(dst) = h;
    *reinterpret_cast<h2_f16x4 *>(dst + kH2Plane) = l;
}

template <int MB, bool BDMA>
__global__ void __launch_bounds__(256)
k_fwd_h2(FwdParams p) {
    constexpr int BM = 64 * MB, NVA = BM / 64, NVB = BN / 64;
    // BDMA: the weights arrive already scaled and split (dc_tag_weight_prep: one 64-byte record
    // {h1[16], h2[16]} per row and stage) and go global -> LDS by LDS-DMA, bypassing registers and
    // the VALU.  The DMA writes lane i's 16 bytes at base + 16*i, so the B image is dense (64-byte
    // rows) and the bank spread of the b128 fragment reads comes from an XOR swizzle instead of
    // padding: piece q of row r sits at position q ^ ((r >> 2) & 3); each lane simply FETCHES the
    // piece that belongs at its position.
    constexpr int kRowB = BDMA ? 64 : kH2Row;
    // A and B live in SEPARATE LDS objects: hipcc guards every DS access that may alias an
    // outstanding LDS-DMA with s_waitcnt vmcnt(0); with one array the ds_writes of the A tile
    // (issued right after the DMA of the B tile) waited for that DMA - a global-memory latency in
    // the middle of every stage
    constexpr int kSzA = BM * kH2Row, kSzB = BN * kRowB, kOffB = 0;
    __shared__ __attribute__((aligned(16))) char sA[2 * kSzA];
    __shared__ __attribute__((aligned(16))) char sB[2 * kSzB];
    __shared__ float s_inv[BM];
    const unsigned ntn = (unsigned)((p.Fo + BN - 1) / BN);
    const unsigned ks = p.ksplit > 1 ? (unsigned)p.ksplit : 1u, ntiles = gridDim.x / ks;
    const unsigned lbb = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned lb = lbb % ntiles, kz = lbb / ntiles;           // tile, reduction range
    const int64_t row0 = (int64_t)(lb / ntn) * BM, col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
    const int lane = threadIdx.x & 63;
    const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
    const int64_t lda = p.x[0].ld;

    // per-thread constant offsets (elements) from the block's uniform bases; rows clamped once
    unsigned offA[NVA], offB[NVB];
    int ldsA[NVA], ldsB[NVB];
    float scA[NVA], scB[NVB];
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
        int64_t row = row0 + r + 64 * j;
        row = row < p.N ? row : p.N - 1;
        offA[j] = (unsigned)((row - row0) * lda + 4 * k4);
        ldsA[j] = (r + 64 * j) * kH2Row + 8 * k4;
        const float m = p.h2.a_rowmax[row];
        scA[j] = h2_scale(m);
        if (k4 == 0) s_inv[r + 64 * j] = h2_unscale(m);
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
        int64_t col = col0 + r + 64 * j;
        col = col < p.Fo ? col : p.Fo - 1;
        offB[j] = (unsigned)((col - col0) * p.Fi + 4 * k4);
        ldsB[j] = kOffB + (r + 64 * j) * kH2Row + 8 * k4;
        scB[j] = BDMA ? 1.f : h2_scale(p.h2.b_rowmax[col]);
    }
    // BDMA: wave `wid` fills rows [32*wid, 32*wid+32) of the B image with two 1 KiB DMA pieces
    unsigned dmaOff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rowl = wid * 32 + q * 16 + (lane >> 2);
        int64_t col = col0 + rowl;
        col = col < p.Fo ? col : p.Fo - 1;
        dmaOff[q] = (unsigned)((col - col0) * p.Fi + 4 * ((lane & 3) ^ ((rowl >> 2) & 3)));
    }

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);
    // ONE segment (launcher): plain K loop over this block's range of 16-wide stages
    const int nst_all = (int)(p.Fi / BK), per = (nst_all + (int)ks - 1) / (int)ks;
    const int st_beg = (int)kz * per;
    const int nst = st_beg >= nst_all ? 0 : (nst_all - st_beg < per ? nst_all - st_beg : per);
    // wave-uniform running bases (scalar registers)
    const float *baseA = p.x[0].p + row0 * lda + (int64_t)st_beg * BK;
    const float *baseB = p.w[0].p + col0 * p.Fi + (int64_t)st_beg * BK;
    float4 va[NVA], vb[NVB];

    auto load = [&]() {
#pragma unroll
        for (int j = 0; j < NVA; ++j) va[j] = *reinterpret_cast<const float4 *>(baseA + offA[j]);
        baseA += BK;                                  // scalar adds
        if (!BDMA) {
#pragma unroll
            for (int j = 0; j < NVB; ++j) vb[j] = *reinterpret_cast<const float4 *>(baseB + offB[j]);
            baseB += BK;
        }
    };
    auto dma_b = [&](int b) {                         // B of the NEXT stage straight into buffer b
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1))) *)(baseB + dmaOff[q]),
                (void __attribute__((address_space(3))) *)(sB + b * kSzB + (wid * 32 + q * 16) * 64), 16, 0, 0);
        baseB += BK;
    };
    auto store = [&](int b) {
#pragma unroll
        for (int j = 0; j < NVA; ++j) h2_split_store(sA + b * kSzA + ldsA[j], va[j], scA[j]);
        if (!BDMA) {
#pragma unroll
            for (int j = 0; j < NVB; ++j) h2_split_store(sB + b * kSzB + ldsB[j], vb[j], scB[j]);
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    const int fragA = (wm * 32 * MB + fr) * kH2Row + 16 * fh;
    const int fragB = kOffB + (wn * 64 + fr) * kH2Row + 16 * fh;
    int fragBd[2][2];                                 // BDMA: swizzled positions of (nb, plane)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int rowl = wn * 64 + nb * 32 + fr;
            fragBd[nb][pl] = kOffB + rowl * 64 + 16 * ((2 * pl + fh) ^ ((rowl >> 2) & 3));
        }
    h2_f16x8 fa[MB][2], fb[2][2];
    auto frags = [&](int b) {
        const char *buf = sB + b * kSzB;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fa[mb][pl] = *reinterpret_cast<const h2_f16x8 *>(sA + b * kSzA + fragA + mb * 32 * kH2Row +
                                                                 pl * kH2Plane);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fb[nb][pl] = BDMA ? *reinterpret_cast<const h2_f16x8 *>(buf + fragBd[nb][pl])
                                  : *reinterpret_cast<const h2_f16x8 *>(buf + fragB + nb * 32 * kH2Row +
                                                                          pl * kH2Plane);
    };
    auto mma = [&]() {
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};       // smallest terms first
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]],
                                                                         acc[mb][nb], 0, 0, 0);
    };

    if (nst > 0) {
        if (BDMA) dma_b(0);
        load();
        store(0);
        if (nst > 1) load();
        if (BDMA) __builtin_amdgcn_s_waitcnt(0xF70 | 0);      // vmcnt(0): the DMA piece has landed
        __syncthreads();
        int it = 0;
        for (; it + 2 < nst; ++it) {                 // steady state: one basic block
            const int cur = it & 1, nxt = cur ^ 1;
            frags(cur);
            if (BDMA) dma_b(nxt);                     // lands during this stage's MFMAs
            store(nxt);
            load();
            mma();
            // the DMA pieces must be in LDS before the barrier.  vmcnt(0) also drains the register
            // loads of stage it+2 (issued at the top of this stage, a whole MFMA phase ago): a counted
            // vmcnt(NVA) would let them stay in flight but relies on hipcc keeping the DMAs ahead of
            // those loads in program order - measured no faster (61.4 vs 61.3 us), so the robust form
            if (BDMA) __builtin_amdgcn_s_waitcnt(0xF70 | 0);
            // (explicit sched_group_barrier orders - 1 MFMA : 6-8 VALU, VALU first, early loads, 2 MFMA
            // groups - all measured 1-3 % slower than hipcc's own order of this single block)
            __syncthreads();
        }
        for (; it < nst; ++it) {                     // last two stages
            const int cur = it & 1, nxt = cur ^ 1;
            frags(cur);
            if (it + 1 < nst) {
                if (BDMA) dma_b(nxt);
                store(nxt);
            }
            mma();
            if (BDMA) __builtin_amdgcn_s_waitcnt(0xF70 | 0);
            __syncthreads();
        }
    }

    float bcol[2], icol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + wn * 64 + nb * 32 + (threadIdx.x & 31);
        bcol[nb] = (p.bias && col < p.Fo) ? p.bias[col] : 0.f;
        icol[nb] = h2_unscale(p.h2.b_rowmax[col < p.Fo ? col : p.Fo - 1]);
    }
    const bool relu = p.relu != 0;
    if (ks > 1) {                                     // split reduction: plain partial, summed later
        float *part = p.kpartial + (int64_t)kz * p.N * p.Fo;
        for_each_acc<MB>(acc, wm, wn, [&](int rr, int c, float v) {
            const int64_t row = row0 + rr, col = col0 + c;
            if (row < p.N && col < p.Fo) part[row * p.Fo + col] = (v * s_inv[rr]) * icol[(c >> 5) & 1];
        });
        return;
    }
    for_each_acc<MB>(acc, wm, wn, [&](int rr, int c, float v) {
        const int64_t row = row0 + rr, col = col0 + c;
        if (row < p.N && col < p.Fo) {
            v = (v * s_inv[rr]) * icol[(c >> 5) & 1];
            v += bcol[(c >> 5) & 1];
            if (relu) v = fmaxf(v, 0.f);
            if (p.exp_lse) v = col < p.exp_ncols ? expf(v - p.exp_lse[row]) : 0.f;
            p.out[row * p.ldo + col] = v;
        }
    });
}



static inline bool h2_al16(const void *q) { return ((uintptr_t)q & 15) == 0; }

bool fwd_h2_launch(const FwdParams &p, int mb, hipStream_t hs) {
    if (!p.h2.a_rowmax || !p.h2.b_rowmax || p.nseg != 1) return false;
    if (fwd_h2w_launch(p, hs)) return true;
    if (p.Fi % BK != 0 || p.Fi < BK) return false;
    // 32-bit per-thread offsets: a block touches 128 rows of each operand
    if (p.x[0].ld * 128 >= ((int64_t)1 << 30) || p.Fi * 128 >= ((int64_t)1 << 30)) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (!h2_al16(p.x[s].p) || !h2_al16(p.w[s].p) || p.x[s].ld % 4 != 0 || p.x[s].ld != p.x[0].ld)
            return false;
    const int64_t grid = ((p.N + 64 * mb - 1) / (64 * mb)) * ((p.Fo + BN - 1) / BN) *
                         (p.ksplit > 1 ? p.ksplit : 1);
    const dim3 gd((unsigned)grid), bd(256);
    if (p.h2.b_presplit) {
        if (mb == 2) DC_LAUNCH((k_fwd_h2<2, true>), gd, bd, 0, hs, p);
        else DC_LAUNCH((k_fwd_h2<1, true>), gd, bd, 0, hs, p);
    } else {
        if (mb == 2) DC_LAUNCH((k_fwd_h2<2, false>), gd, bd, 0, hs, p);
        else DC_LAUNCH((k_fwd_h2<1, false>), gd, bd, 0, hs, p);
    }
    return true;
}

}  // namespace dc
