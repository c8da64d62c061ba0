// dc_attn.hip -- row kernels of the blocked cross-attention.
//
// The reference's MultiHeadAttention (/root/reference/models/model.py:7-21) materialises, per head,
// scores = head(x_soft) . head(x_rigid)^T  [N_s, N_r], softmax over the rigid nodes, and
// weights . x_rigid - 3.2 GB of scores + 3.2 GB of saved weights per head at batch 32.  Here the
// score matrix only ever exists for a block of Bq soft rows: the three GEMMs of a block run on the
// fp16x2 dense kernels (dc_dense_h2.hip / dc_dense_split.hip) and the kernels below do the row-wise
// parts in place; the backward recomputes the block's weights from the saved row log-sum-exp.
//
//   dc_attn_softmax_rows : S[i, 0:n] <- softmax(S[i, 0:n]),  S[i, n:npad] <- 0,  lse[i] = log sum exp
//   dc_attn_exp_rows     : S[i, 0:n] <- exp(S[i, 0:n] - lse[i]),  S[i, n:npad] <- 0      (recompute)
//   dc_attn_ds_rows      : dP[i, j] <- P[i, j] * (dP[i, j] - delta[i]),  rowmax[i] = max_j |.|
// One 256-thread workgroup per row; rows of up to 32 K (softmax) / 24 K (ds) floats are held in registers (one global
// read per input row), longer ones are re-read from L2 on the later passes.
#include "dc_common.h"

namespace dc {

__device__ __forceinline__ float block_reduce_max(float v, float *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
}

__device__ __forceinline__ float block_reduce_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return v;
}

__global__ void __launch_bounds__(256)
k_attn_softmax_rows(float *__restrict__ s, int64_t ld, int64_t n, int64_t npad, float *__restrict__ lse) {
    __shared__ float red[4];
    float *row = s + (int64_t)blockIdx.x * ld;
    float m = -INFINITY;
    for (int64_t j = threadIdx.x; j < n; j += 256) m = fmaxf(m, row[j]);
    m = block_reduce_max(m, red);
    float acc = 0.f;
    for (int64_t j = threadIdx.x; j < n; j += 256) acc += expf(row[j] - m);
    acc = block_reduce_sum(acc, red);
    const float l = m + logf(acc);
    for (int64_t j = threadIdx.x; j < npad; j += 256) row[j] = j < n ? expf(row[j] - l) : 0.f;
    if (threadIdx.x == 0) lse[blockIdx.x] = l;
}

__global__ void __launch_bounds__(256)
k_attn_exp_rows(float *__restrict__ s, int64_t ld, int64_t n, int64_t npad, const float *__restrict__ lse) {
    float *row = s + (int64_t)blockIdx.x * ld;
    const float l = lse[blockIdx.x];
    for (int64_t j = threadIdx.x; j < npad; j += 256) row[j] = j < n ? expf(row[j] - l) : 0.f;
}

// dS = P * (dP - delta) in place of dP, + the row maxima of |dS|.  delta == nullptr: the kernel forms
// delta_i = sum_j P_ij dP_ij itself (a first pass over the row, fixed-order block sum) - what the
// softmax backward of autograd does.  With a delta handed in from rowsum(dO * O) the row sums of dS
// are off zero by O's own rounding (~1e-6 of |delta|); that bias survives into sum_j dK_j, which is
// mathematically zero, and is amplified wherever the keys share a large common component (after a
// ReLU encoder: 1e-5 .. 4e-5 in the shared head weights against float64, r02 attn_diag).
__global__ void __launch_bounds__(256)
k_attn_ds_rows(const float *__restrict__ p, float *__restrict__ dp, int64_t ld, int64_t npad,
               const float *__restrict__ delta, float *__restrict__ rowmax) {
    __shared__ float red[4];
    const float *pr = p + (int64_t)blockIdx.x * ld;
    float *dr = dp + (int64_t)blockIdx.x * ld;
    float d;
    if (delta) {
        d = delta[blockIdx.x];
    } else {
        // the recomputed weights exp(S - lse) sum to 1 + O(|lse| 2^-24), not to 1: delta is taken
        // relative to their actual sum, so that sum_j dS_ij = sum P dP - delta sum P vanishes
        float acc = 0.f, sp = 0.f;
        for (int64_t j = threadIdx.x; j < npad; j += 256) {
            const float pj = pr[j];
            acc += pj * dr[j];
            sp += pj;
        }
        acc = block_reduce_sum(acc, red);
        sp = block_reduce_sum(sp, red);
        d = sp > 0.f ? acc / sp : acc;
    }
    float m = 0.f;
    for (int64_t j = threadIdx.x; j < npad; j += 256) {
        const float v = pr[j] * (dr[j] - d);
        dr[j] = v;
        m = fmaxf(m, fabsf(v));
    }
    m = block_reduce_max(m, red);
    if (threadIdx.x == 0) rowmax[blockIdx.x] = m;
}

// ---- register-resident rows ---------------------------------------------------------------------------
// The kernels above read a row two or three times (the later passes from L2).  For rows of up to NV x 1024
// floats a 256-thread workgroup can hold the whole row in registers (NV float4 per thread): ONE global read per
// input row, one write.  Same formulas; the partition of a row over the threads differs from the strided
// kernels, so sums agree to rounding, not bit for bit.
using af32x4 = __attribute__((ext_vector_type(4))) float;

template <int NV>
__global__ void __launch_bounds__(256)
k_attn_softmax_rows_reg(float *__restrict__ s, int64_t ld, int64_t n64, int64_t npad64, float *__restrict__ lse) {
    __shared__ float red[4];
    float *row = s + (int64_t)blockIdx.x * ld;
    const int n = (int)n64, npad = (int)npad64;
    af32x4 v[NV];
    // columns >= n (key padding) are loaded as -inf: they drop out of the maximum, add exp(-inf) = 0 to the
    // sum and come out of the last pass as exactly 0 - no per-element tests after the load
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 256 + (int)threadIdx.x) * 4;
        if (i < npad) {
            v[j] = *reinterpret_cast<const af32x4 *>(row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] = i + e < n ? v[j][e] : -INFINITY;
                m = fmaxf(m, v[j][e]);
            }
        } else {
            v[j] = af32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    m = block_reduce_max(m, red);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += expf(v[j][e] - m);
    acc = block_reduce_sum(acc, red);
    const float l = m + logf(acc);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (j * 256 + (int)threadIdx.x) * 4;
        if (i < npad) {
            af32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = expf(v[j][e] - l);
            *reinterpret_cast<af32x4 *>(row + i) = o;
        }
    }
    if (threadIdx.x == 0) lse[blockIdx.x] = l;
}

// dS = P * (dP - delta_i) in place of dP with delta_i = sum_j P dP / sum_j P formed in the kernel (see
// k_attn_ds_rows), + the row maximum of |dS|; P and dP rows held in registers
template <int NV>
__global__ void __launch_bounds__(256)
k_attn_ds_rows_reg(const float *__restrict__ p, float *__restrict__ dp, int64_t ld, int64_t npad,
                   float *__restrict__ rowmax) {
    __shared__ float red[4];
    const float *pr = p + (int64_t)blockIdx.x * ld;
    float *dr = dp + (int64_t)blockIdx.x * ld;
    af32x4 vp[NV], vd[NV];
    float acc = 0.f, sp = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t i = ((int64_t)j * 256 + threadIdx.x) * 4;
        if (i < npad) {
            vp[j] = *reinterpret_cast<const af32x4 *>(pr + i);
            vd[j] = *reinterpret_cast<const af32x4 *>(dr + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += vp[j][e] * vd[j][e];
                sp += vp[j][e];
            }
        }
    }
    acc = block_reduce_sum(acc, red);
    sp = block_reduce_sum(sp, red);
    const float d = sp > 0.f ? acc / sp : acc;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t i = ((int64_t)j * 256 + threadIdx.x) * 4;
        if (i < npad) {
            af32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = vp[j][e] * (vd[j][e] - d);
                m = fmaxf(m, fabsf(o[e]));
            }
            *reinterpret_cast<af32x4 *>(dr + i) = o;
        }
    }
    m = block_reduce_max(m, red);
    if (threadIdx.x == 0) rowmax[blockIdx.x] = m;
}

}  // namespace dc

using namespace dc;

static inline bool attn_vec_ok(const void *a, const void *b, int64_t ld, int64_t npad) {
    return npad % 4 == 0 && ld % 4 == 0 && (((uintptr_t)a) & 15) == 0 && (!b || (((uintptr_t)b) & 15) == 0);
}

static int attn_check(const char *what, const void *s, int64_t ld, int64_t rows, int64_t n, int64_t npad) {
    DC_REQUIRE(rows >= 0 && n >= 1 && npad >= n && ld >= npad, "%s: bad sizes (rows=%lld n=%lld npad=%lld ld=%lld)",
               what, (long long)rows, (long long)n, (long long)npad, (long long)ld);
    DC_REQUIRE(rows == 0 || s, "%s: null pointer", what);
    DC_REQUIRE(rows < (int64_t)INT32_MAX, "%s: too many rows", what);
    return DC_OK;
}

extern "C" int dc_attn_softmax_rows(float *s, int64_t ld, int64_t rows, int64_t n, int64_t npad,
                                    float *lse, dc_stream_t stream) {
    if (int rc = attn_check("dc_attn_softmax_rows", s, ld, rows, n, npad)) return rc;
    if (rows == 0) return DC_OK;
    DC_REQUIRE(lse, "dc_attn_softmax_rows: null lse");
    const dim3 gr((unsigned)rows), bl(256);
    hipStream_t hs = (hipStream_t)stream;
    if (attn_vec_ok(s, nullptr, ld, npad) && npad <= 32 * 1024) {
        if (npad <= 8 * 1024) DC_LAUNCH((k_attn_softmax_rows_reg<8>), gr, bl, 0, hs, s, ld, n, npad, lse);
        else if (npad <= 16 * 1024) DC_LAUNCH((k_attn_softmax_rows_reg<16>), gr, bl, 0, hs, s, ld, n, npad, lse);
        else if (npad <= 24 * 1024) DC_LAUNCH((k_attn_softmax_rows_reg<24>), gr, bl, 0, hs, s, ld, n, npad, lse);
        else DC_LAUNCH((k_attn_softmax_rows_reg<32>), gr, bl, 0, hs, s, ld, n, npad, lse);
        return check_launch("dc_attn_softmax_rows");
    }
    DC_LAUNCH(k_attn_softmax_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, ld,
                       n, npad, lse);
    return check_launch("dc_attn_softmax_rows");
}

extern "C" int dc_attn_exp_rows(float *s, int64_t ld, int64_t rows, int64_t n, int64_t npad,
                                const float *lse, dc_stream_t stream) {
    if (int rc = attn_check("dc_attn_exp_rows", s, ld, rows, n, npad)) return rc;
    if (rows == 0) return DC_OK;
    DC_REQUIRE(lse, "dc_attn_exp_rows: null lse");
    DC_LAUNCH(k_attn_exp_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, ld, n,
                       npad, lse);
    return check_launch("dc_attn_exp_rows");
}

extern "C" int dc_attn_ds_rows(const float *p, float *dp, int64_t ld, int64_t rows, int64_t npad,
                               const float *delta, float *rowmax, dc_stream_t stream) {
    if (int rc = attn_check("dc_attn_ds_rows", p, ld, rows, npad, npad)) return rc;
    if (rows == 0) return DC_OK;
    DC_REQUIRE(dp && rowmax, "dc_attn_ds_rows: null pointer");
    if (!delta && attn_vec_ok(p, dp, ld, npad) && npad <= 24 * 1024) {
        const dim3 gr((unsigned)rows), bl(256);
        hipStream_t hs = (hipStream_t)stream;
        if (npad <= 8 * 1024) DC_LAUNCH((k_attn_ds_rows_reg<8>), gr, bl, 0, hs, p, dp, ld, npad, rowmax);
        else if (npad <= 16 * 1024) DC_LAUNCH((k_attn_ds_rows_reg<16>), gr, bl, 0, hs, p, dp, ld, npad, rowmax);
        else DC_LAUNCH((k_attn_ds_rows_reg<24>), gr, bl, 0, hs, p, dp, ld, npad, rowmax);
        return check_launch("dc_attn_ds_rows");
    }
    DC_LAUNCH(k_attn_ds_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, ld,
                       npad, delta, rowmax);
    return check_launch("dc_attn_ds_rows");
}
