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
// One 256-thread workgroup per row; a row (up to ~100 KB) is read from L2 on the second pass.
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

}  // namespace dc

using namespace dc;

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
    hipLaunchKernelGGL(k_attn_softmax_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, ld,
                       n, npad, lse);
    return check_launch("dc_attn_softmax_rows");
}

extern "C" int dc_attn_exp_rows(float *s, int64_t ld, int64_t rows, int64_t n, int64_t npad,
                                const float *lse, dc_stream_t stream) {
    if (int rc = attn_check("dc_attn_exp_rows", s, ld, rows, n, npad)) return rc;
    if (rows == 0) return DC_OK;
    DC_REQUIRE(lse, "dc_attn_exp_rows: null lse");
    hipLaunchKernelGGL(k_attn_exp_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, ld, n,
                       npad, lse);
    return check_launch("dc_attn_exp_rows");
}

extern "C" int dc_attn_ds_rows(const float *p, float *dp, int64_t ld, int64_t rows, int64_t npad,
                               const float *delta, float *rowmax, dc_stream_t stream) {
    if (int rc = attn_check("dc_attn_ds_rows", p, ld, rows, npad, npad)) return rc;
    if (rows == 0) return DC_OK;
    DC_REQUIRE(dp && rowmax, "dc_attn_ds_rows: null pointer");
    hipLaunchKernelGGL(k_attn_ds_rows, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, ld,
                       npad, delta, rowmax);
    return check_launch("dc_attn_ds_rows");
}
