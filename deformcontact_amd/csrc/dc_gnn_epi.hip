// dc_gnn_epi.hip -- the row-wise passes around the aggregation of a GCNConv / GATConv layer, fused.
//
// PyG gcn_conv.py / gat_conv.py (reached from /root/reference/models/model.py:39,71,77 with backbone = "GCNConv" /
// "GATConv"): out = propagate(...) + bias, then relu in the encoder loop; GAT also alpha_src = (h * att_src).sum(-1),
// alpha_dst likewise.  On stock PyTorch those are elementwise / reduce kernels of their own (add, threshold,
// threshold_backward, two mul + two sum, and in backward sum(0) for the bias and the attention vectors).  Here:
//   dc_spmm_f32_bias_act (dc_spmm.hip)  bias + ReLU in the aggregation's epilogue
//   dc_mask_colsum_f32                  gm = g * (y > 0) and the bias gradient sum_i gm[i, :] in one pass
//   dc_gat_alpha_fwd / _bwd             both attention dot products in one pass over h; their backward (rank-one
//                                       updates of dh, the two attention-vector gradients) in one pass
// Column sums are deterministic: per-block partials combined in block order (no float atomics).
#include "dc_common.h"

#pragma clang fp contract(off)

namespace dc {

constexpr int kEpiRows = 32;                        // rows per block of the column-sum passes (1,024 blocks at N = 32,768)

// block = kEpiRows rows; thread t owns 4 columns (c = 4 (t % (F/4))) of the row group t / (F/4): for F = 256, 64 threads
// span a row and 4 rows are walked side by side
__global__ void __launch_bounds__(256)
k_mask_colsum(const float *__restrict__ g, int64_t ldg, const float *__restrict__ y, int64_t ldy, float *gm,
              int64_t ldgm, int64_t N, int F, float *__restrict__ partial) {
    __shared__ float red[256 * 4];
    const int tpr = F / 4, groups = 256 / tpr;            // threads per row, rows side by side
    const int gidx = threadIdx.x / tpr, c = 4 * (threadIdx.x % tpr);
    const int64_t r0 = (int64_t)blockIdx.x * kEpiRows;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gidx < groups)
        for (int64_t r = r0 + gidx; r < r0 + kEpiRows && r < N; r += groups) {
            float4 v = *reinterpret_cast<const float4 *>(g + r * ldg + c);
            if (y) {
                const float4 m = *reinterpret_cast<const float4 *>(y + r * ldy + c);
                v.x = m.x > 0.f ? v.x : 0.f, v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f, v.w = m.w > 0.f ? v.w : 0.f;
            }
            if (gm) *reinterpret_cast<float4 *>(gm + r * ldgm + c) = v;
            s.x = s.x + v.x, s.y = s.y + v.y, s.z = s.z + v.z, s.w = s.w + v.w;
        }
    *reinterpret_cast<float4 *>(&red[4 * threadIdx.x]) = s;
    __syncthreads();
    if (threadIdx.x < tpr) {                               // the row groups of a column quad, in group order
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < groups; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(&red[4 * (q * tpr + threadIdx.x)]);
            t.x = t.x + v.x, t.y = t.y + v.y, t.z = t.z + v.z, t.w = t.w + v.w;
        }
        *reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * F + c) = t;
    }
}

// out[c] (+)= sum over blocks of partial[b * stride + c]: one wave per column, lane l sums the blocks l, l + 64, ...
// in order, the 64 lane sums meet in a fixed butterfly - the same association every run (deterministic), and 64 loads
// in flight per column instead of one thread walking hundreds of partials one after the other
__global__ void __launch_bounds__(256)
k_colsum_final(const float *__restrict__ partial, int64_t nblocks, int64_t stride, int F, float *out, int accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= F) return;
    float s = 0.f;
    for (int64_t b = lane; b < nblocks; b += 64) s = s + partial[b * stride + c];
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) s = s + __shfl_xor(s, q);
    if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

// one wave per row: a_src[i] = sum_c h[i, c] att_src[c], a_dst likewise
__global__ void __launch_bounds__(256)
k_gat_alpha_fwd(const float *__restrict__ h, int64_t ldh, const float *__restrict__ att_src,
                const float *__restrict__ att_dst, float *a_src, float *a_dst, int64_t N, int F) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f, d = 0.f;
    for (int c = 4 * lane; c < F; c += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(h + row * ldh + c);
        const float4 as = *reinterpret_cast<const float4 *>(att_src + c), ad = *reinterpret_cast<const float4 *>(att_dst + c);
        s += v.x * as.x + v.y * as.y + v.z * as.z + v.w * as.w;
        d += v.x * ad.x + v.y * ad.y + v.z * ad.z + v.w * ad.w;
    }
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) {
        s += __shfl_xor(s, q);
        d += __shfl_xor(d, q);
    }
    if (lane == 0) a_src[row] = s, a_dst[row] = d;
}

// backward of the two dot products, one pass over the rows of a block:
//   gh[i, :] += ga_src[i] att_src + ga_dst[i] att_dst          (gh already holds the aggregation's gradient)
//   partial[block][0:F]   = sum_i ga_src[i] h[i, :],  partial[block][F:2F] = sum_i ga_dst[i] h[i, :]
__global__ void __launch_bounds__(256)
k_gat_alpha_bwd(const float *__restrict__ h, int64_t ldh, const float *__restrict__ ga_src,
                const float *__restrict__ ga_dst, const float *__restrict__ att_src, const float *__restrict__ att_dst,
                float *gh, int64_t ldgh, int64_t N, int F, float *__restrict__ partial) {
    __shared__ float red[256 * 8];
    const int tpr = F / 4, groups = 256 / tpr;
    const int gidx = threadIdx.x / tpr, c = 4 * (threadIdx.x % tpr);
    const int64_t r0 = (int64_t)blockIdx.x * kEpiRows;
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f), sd = ss;
    if (gidx < groups) {
        const float4 as = *reinterpret_cast<const float4 *>(att_src + c), ad = *reinterpret_cast<const float4 *>(att_dst + c);
        for (int64_t r = r0 + gidx; r < r0 + kEpiRows && r < N; r += groups) {
            const float gs = ga_src[r], gd = ga_dst[r];
            const float4 v = *reinterpret_cast<const float4 *>(h + r * ldh + c);
            float4 o = *reinterpret_cast<const float4 *>(gh + r * ldgh + c);
            o.x += gs * as.x + gd * ad.x, o.y += gs * as.y + gd * ad.y;
            o.z += gs * as.z + gd * ad.z, o.w += gs * as.w + gd * ad.w;
            *reinterpret_cast<float4 *>(gh + r * ldgh + c) = o;
            ss.x += gs * v.x, ss.y += gs * v.y, ss.z += gs * v.z, ss.w += gs * v.w;
            sd.x += gd * v.x, sd.y += gd * v.y, sd.z += gd * v.z, sd.w += gd * v.w;
        }
    }
    *reinterpret_cast<float4 *>(&red[8 * threadIdx.x]) = ss;
    *reinterpret_cast<float4 *>(&red[8 * threadIdx.x + 4]) = sd;
    __syncthreads();
    if (threadIdx.x < tpr) {
        float4 ts = make_float4(0.f, 0.f, 0.f, 0.f), td = ts;
        for (int q = 0; q < groups; ++q) {
            const float4 a = *reinterpret_cast<const float4 *>(&red[8 * (q * tpr + threadIdx.x)]);
            const float4 b = *reinterpret_cast<const float4 *>(&red[8 * (q * tpr + threadIdx.x) + 4]);
            ts.x += a.x, ts.y += a.y, ts.z += a.z, ts.w += a.w;
            td.x += b.x, td.y += b.y, td.z += b.z, td.w += b.w;
        }
        *reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * 2 * F + c) = ts;
        *reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * 2 * F + F + c) = td;
    }
}

static inline bool epi_al16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static inline bool epi_width_ok(int64_t F) { return F >= 4 && F <= 1024 && F % 4 == 0 && 256 % (F / 4) == 0; }

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_colsum_workspace_bytes(int64_t N, int64_t F, int nvec) {
    if (N < 0 || F < 1 || nvec < 1) return -1;
    return ((N + kEpiRows - 1) / kEpiRows) * F * nvec * (int64_t)sizeof(float);
}

extern "C" int dc_mask_colsum_f32(const float *g, int64_t ldg, const float *y_mask, int64_t ldy, float *gm,
                                  int64_t ldgm, int64_t N, int64_t F, void *workspace, int64_t workspace_bytes,
                                  float *colsum, int accumulate, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && epi_width_ok(F), "dc_mask_colsum_f32: F must be a multiple of 4 that divides 1024 (F=%lld)",
               (long long)F);
    DC_REQUIRE(g && colsum && ldg >= F && ldg % 4 == 0 && epi_al16(g) && (!y_mask || (ldy >= F && ldy % 4 == 0 && epi_al16(y_mask))) &&
                   (!gm || (ldgm >= F && ldgm % 4 == 0 && epi_al16(gm))),
               "dc_mask_colsum_f32: null / misaligned operand");
    const int64_t nb = (N + kEpiRows - 1) / kEpiRows;
    DC_REQUIRE(workspace_bytes >= dc_colsum_workspace_bytes(N, F, 1) && (workspace || nb == 0),
               "dc_mask_colsum_f32: workspace too small");
    if (nb > 0)
        DC_LAUNCH(k_mask_colsum, dim3((unsigned)nb), dim3(256), 0, stream, g, ldg, y_mask, ldy, gm, ldgm, N, (int)F,
                  (float *)workspace);
    DC_LAUNCH(k_colsum_final, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, stream, (const float *)workspace, nb, F,
              (int)F, colsum, accumulate);
    return check_launch("dc_mask_colsum_f32");
}

extern "C" int dc_gat_alpha_fwd(const float *h, int64_t ldh, const float *att_src, const float *att_dst, float *a_src,
                                float *a_dst, int64_t N, int64_t F, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 4 && F % 4 == 0 && ldh >= F && ldh % 4 == 0, "dc_gat_alpha_fwd: F %% 4 == 0, ldh %% 4 == 0");
    if (N == 0) return DC_OK;
    DC_REQUIRE(h && att_src && att_dst && a_src && a_dst && epi_al16(h) && epi_al16(att_src) && epi_al16(att_dst),
               "dc_gat_alpha_fwd: null / misaligned operand");
    DC_LAUNCH(k_gat_alpha_fwd, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, h, ldh, att_src, att_dst, a_src,
              a_dst, N, (int)F);
    return check_launch("dc_gat_alpha_fwd");
}

extern "C" int dc_gat_alpha_bwd(const float *h, int64_t ldh, const float *ga_src, const float *ga_dst,
                                const float *att_src, const float *att_dst, float *gh, int64_t ldgh, int64_t N,
                                int64_t F, void *workspace, int64_t workspace_bytes, float *g_att_src,
                                float *g_att_dst, int accumulate, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && epi_width_ok(F), "dc_gat_alpha_bwd: F must be a multiple of 4 that divides 1024 (F=%lld)",
               (long long)F);
    DC_REQUIRE(h && ga_src && ga_dst && att_src && att_dst && gh && g_att_src && g_att_dst && ldh >= F && ldgh >= F &&
                   ldh % 4 == 0 && ldgh % 4 == 0 && epi_al16(h) && epi_al16(gh) && epi_al16(att_src) && epi_al16(att_dst),
               "dc_gat_alpha_bwd: null / misaligned operand");
    const int64_t nb = (N + kEpiRows - 1) / kEpiRows;
    DC_REQUIRE(workspace_bytes >= dc_colsum_workspace_bytes(N, F, 2) && (workspace || nb == 0),
               "dc_gat_alpha_bwd: workspace too small");
    if (nb > 0)
        DC_LAUNCH(k_gat_alpha_bwd, dim3((unsigned)nb), dim3(256), 0, stream, h, ldh, ga_src, ga_dst, att_src, att_dst,
                  gh, ldgh, N, (int)F, (float *)workspace);
    // every block's partial row holds the two column vectors [src | dst]
    const float *ws = (const float *)workspace;
    DC_LAUNCH(k_colsum_final, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, stream, ws, nb, 2 * F, (int)F, g_att_src,
              accumulate);
    DC_LAUNCH(k_colsum_final, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, stream, ws + F, nb, 2 * F, (int)F,
              g_att_dst, accumulate);
    return check_launch("dc_gat_alpha_bwd");
}
