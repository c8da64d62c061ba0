// dc_dense.h -- shared pieces of the dense-block kernels (internal).
#pragma once
#include "dc_common.h"

namespace dc {

constexpr int BK = 16;
constexpr int LDK = BK + 4;
constexpr int BN = 128;           // block tile = (64*MB) x 128, 2x2 waves of (32*MB) x 64
constexpr int kMaxSeg = DC_MAX_SEG;

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Mat {
    const float *p;
    int64_t ld;
};

// ---- LDS -> register fragments (one 8-wide k chunk) and the 8 MFMAs that consume them ----
// The fragments of the NEXT chunk are read while the MFMAs of the current chunk issue (the
// next chunk may belong to the next stage: 3-deep LDS ring), so the matrix pipe never waits
// on LDS latency, also right after a barrier.
template <int MB>
struct Frag {
    float a[MB][4];
    float b[2][4];
};

template <int MB, bool A_KC, bool B_KC>
__device__ __forceinline__ void load_frag(Frag<MB> &f, const float *As, const float *Bs, int c,
                                          int wm, int wn) {
    constexpr int BM = 64 * MB;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int kb = 8 * c + 4 * h;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int row = wm * 32 * MB + mb * 32 + r;
        if (A_KC) {
            const float4 t = *reinterpret_cast<const float4 *>(As + row * LDK + kb);
            f.a[mb][0] = t.x, f.a[mb][1] = t.y, f.a[mb][2] = t.z, f.a[mb][3] = t.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) f.a[mb][t] = As[(kb + t) * BM + row];
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (B_KC) {
            const float4 t =
                *reinterpret_cast<const float4 *>(Bs + (wn * 64 + nb * 32 + r) * LDK + kb);
            f.b[nb][0] = t.x, f.b[nb][1] = t.y, f.b[nb][2] = t.z, f.b[nb][3] = t.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) f.b[nb][t] = Bs[(kb + t) * BN + wn * 64 + nb * 32 + r];
        }
    }
}

template <int MB>
__device__ __forceinline__ void mma_frag(const Frag<MB> &f, f32x16 (&acc)[MB][2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mb][t], f.b[0][t], acc[mb][0], 0, 0, 0);
            acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mb][t], f.b[1][t], acc[mb][1], 0, 0, 0);
        }
}

template <int MB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MB][2]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][0][i] = 0.f, acc[mb][1][i] = 0.f;
}

template <int MB> struct Tile {
    static constexpr int BM = 64 * MB;
    static constexpr int A_KC = BM * LDK, B_KC = BN * LDK, A_RC = BK * BM, B_RC = BK * BN;
};

// C/D fragment -> (row, col) of the wave's 32x32 blocks: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
template <int MB, typename F>
__device__ __forceinline__ void for_each_acc(const f32x16 (&acc)[MB][2], int wm, int wn, F &&f) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = wm * 32 * MB + mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int col = wn * 64 + nb * 32 + c;
                f(row, col, acc[mb][nb][reg]);
            }
}

// Power-of-two operand scaling of the fp16x2 ("h2") kernels: an operand row / tensor whose
// largest magnitude is m is multiplied by 2^(141 - e(m)) (e = biased fp32 exponent, clamped to
// >= 15) so that it lands in [2^14, 2^15) - inside the fp16 range with 2x headroom - and the
// accumulator is multiplied back by 2^(e(m) - 141); both factors are exact.
struct H2Scales {
    const float *a_rowmax;   // fwd: [N] max |x[row, :]| over all segments; dX, dW: [N] max |g[row, :]|
    const float *b_rowmax;   // fwd / dX: [Fo] max_s,f |W_s[o, f]|;  dW: [N] max |x[row, :]|
    int b_presplit;          // fwd: w[0] is the scaled fp16x2 image written by dc_tag_weight_prep
};

__device__ __forceinline__ unsigned h2_exp(float m) {
    unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
    e = e < 15u ? 15u : e;
    return e > 254u ? 254u : e;
}
__device__ __forceinline__ float h2_scale(float m) { return __uint_as_float((268u - h2_exp(m)) << 23); }
__device__ __forceinline__ float h2_unscale(float m) { return __uint_as_float((h2_exp(m) - 14u) << 23); }

// max of v[beg..end) over a 256-thread block (every thread gets the result); `red` = 4 floats of LDS
__device__ __forceinline__ float h2_block_max(const float *v, int64_t beg, int64_t end, float *red) {
    float m = 0.f;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) m = fmaxf(m, v[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return m;
}

// Row groups of a block-diagonal union (e.g. the soft and the rigid branch of the encoder in one node space,
// models/model.py:69-78): rows [row_beg[g], row_beg[g+1]) use group g's weights.  Group starts are multiples
// of kGroupAlign rows (the host pads every group with zero rows), so no tile / node chunk straddles two groups.
constexpr int kMaxGroups = DC_MAX_GROUPS;
constexpr int kGroupAlign = DC_GROUP_ALIGN;
struct FwdGroups {
    int n;                              // 0 = ungrouped (p.w[0] / p.bias / p.h2.b_rowmax apply to every row)
    int64_t row_beg[kMaxGroups];
    int64_t row_end[kMaxGroups];        // row_beg[g] + data rows of group g: the rows behind are padding and are stored as 0
    const float *w[kMaxGroups];         // pre-split weight image of group g
    const float *bias[kMaxGroups];
    const float *b_rowmax[kMaxGroups];
};
__device__ __forceinline__ int group_of_row(const int64_t (&row_beg)[kMaxGroups], int n, int64_t row) {
    int g = 0;
#pragma unroll
    for (int q = 1; q < kMaxGroups; ++q)
        if (q < n && row >= row_beg[q]) g = q;
    return g;
}

struct FwdParams {
    Mat x[kMaxSeg];
    Mat w[kMaxSeg];
    const float *bias;
    float *out;
    int64_t ldo, N, Fi, Fo;
    int nseg, relu;
    H2Scales h2;
    int ksplit;              // k_fwd_h2: > 1 = the reduction is cut into ksplit ranges, each block
    float *kpartial;         //   writes its [N, Fo] partial to kpartial[z] (no bias / relu), summed later
    // attention recompute (dc_tag_linear_fwd_h2p_exp): out[i, j] = j < exp_ncols ? exp(acc - exp_lse[i]) : 0
    const float *exp_lse;
    int64_t exp_ncols;
    FwdGroups grp;
    // k_fwd_h2w only: the left operand is x[i, :] - x2_coef[i] * x2[i, :] (x2 with x's leading dimension) - the attention
    // backward's dQ = (dS' - eps o P) K (dc_tag_linear_fwd_h2p_corr); null = plain x
    const float *x2, *x2_coef;
};

struct DxParams {
    Mat g, mask;
    int has_mask;
    Mat w[kMaxSeg];
    float *gx[kMaxSeg];
    int64_t ldgx[kMaxSeg];
    int64_t N, Fi, Fo;
    int nseg;
    H2Scales h2;
};

// node chunks of a grouped dW: group g owns chunks [chunk_beg[g], chunk_beg[g+1]) of chunk_rows[g] nodes each
// over rows [row_beg[g], row_end[g]) - the plan a separate launch over that group alone would use, so the
// partial slabs (and their sums) are bit-identical to it
struct DwGroups {
    int n;                              // 0 = ungrouped
    int chunk_beg[kMaxGroups + 1];
    int64_t row_beg[kMaxGroups], row_end[kMaxGroups], chunk_rows[kMaxGroups];
};

struct DwParams {
    Mat g, mask;
    int has_mask;
    Mat x[kMaxSeg];
    float *partial;        // [nchunks][nseg][Fo][Fi]
    float *bias_partial;   // [nchunks][Fo] or null
    int64_t N, Fi, Fo, chunk_rows;
    int nseg, nchunks;
    H2Scales h2;
    DwGroups grp;
    // k_dw_h2w only: the gradient operand is g[i, :] - g2_coef[i] * g2[i, :] (g2 with g's leading dimension) - the
    // attention backward's row-sum correction of dS (dc_tag_linear_bwd_dw_h2_corr); null = plain g
    const float *g2, *g2_coef;
};


// dW over bf16-stored operands (dc_dense_bf16.hip): gm [N, Fo] and the hop slab x [N, nseg * Fi], both bf16
struct DwBf16Params {
    const uint16_t *g, *x;
    int64_t ldg, ldx;
    float *partial;        // [nchunks][nseg][Fo][Fi]
    float *bias_partial;   // [nchunks][Fo] or null
    int64_t N, Fi, Fo, chunk_rows;
    int nseg, nchunks;
};
bool dw_bf16_launch(const DwBf16Params &p, hipStream_t hs);

// lean fast-path launchers (dc_dense_fast.hip); return false when the shape is not eligible
bool fwd_fast_launch(const FwdParams &p, int mb, hipStream_t hs);
bool dx_fast_launch(const DxParams &p, int mb, hipStream_t hs);
bool dw_fast_launch(const DwParams &p, int mb, hipStream_t hs);
// split launchers (dc_dense_split.hip): products = 6 / 3 / 1 bf16 planes, or 2 = the scaled
// fp16x2 mode (two fp16 planes, 3 MFMA products, fp32-accurate; needs p.h2)
bool fwd_split_launch(const FwdParams &p, int mb, int products, hipStream_t hs);
bool dx_split_launch(DxParams p, float *wt, int mb, int products, hipStream_t hs);
bool dw_split_launch(const DwParams &p, int mb, int products, hipStream_t hs);
// dW, fp16x2, 128 x 256 tiles over node chunks (dc_dense_split.hip); the only dW kernel that takes p.grp
bool dw_h2w_launch(const DwParams &p, hipStream_t hs);
// tuned forward-shaped fp16x2 kernel (dc_dense_h2.hip)
bool fwd_h2_launch(const FwdParams &p, int mb, hipStream_t hs);
// 128 x 256 tiles, BK = 32, for the wide layers (dc_dense_h2w.hip); tried first by fwd_h2_launch
bool fwd_h2w_launch(const FwdParams &p, hipStream_t hs);
// the same tiles with both operands by LDS-DMA and the waves split by role (dc_dense_h2d.hip); tried first by fwd_h2w_launch
bool fwd_h2d_launch(const FwdParams &p, hipStream_t hs);
// wt[s][f][o] = ws[s][o][f]
void transpose_weights_launch(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *wt,
                              hipStream_t hs);

}  // namespace dc
