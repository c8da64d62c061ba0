// dc_dense.hip -- the dense block of TAGConv on fp32 MFMA (gfx950).
//
// Replaces PyG tag_conv.py's `out = lins[0](x); out = out + lins[k](x_k); out + bias`
// (K+1 bias-free F.linear calls + adds) and the ReLU the reference applies right
// after (/root/reference/models/model.py:71,77), forward and backward:
//
//   fwd : out[N,Fo]   = act( sum_s xs[s][N,Fi] . ws[s][Fo,Fi]^T + bias )
//   dX  : gxs[s][N,Fi] = (g * relu')[N,Fo] . ws[s][Fo,Fi]
//   dW  : gws[s][Fo,Fi] = (g * relu')^T[Fo,N] . xs[s][N,Fi],  gbias = colsum(g * relu')
//
// Exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain (no TF32/xf32 on gfx950),
// 64 FLOP/clk/SIMD = 157 TF peak.  At that rate the kernel is MFMA-bound by a wide margin
// (12 B/clk/CU of tile traffic), so the structure is the simple one: 64x128 block tile,
// 4 waves (2x2) of 32x64, BK = 16, register-staged double-buffered LDS, one barrier per
// stage, 4-5 blocks per CU so every SIMD always has an MFMA-ready wave.
//
// LDS layouts.  An operand whose reduction index is contiguous in memory is kept
// "KC" ([row][k], stride 20 floats: conflict-free ds_read_b128); one whose row index is
// contiguous is kept "RC" ([k][row]: conflict-free ds_read_b32) - no transposes while
// staging.  Inside each 8-wide k chunk the k order is permuted identically for A and B
// (lane half h takes k = 4h+t for MFMA t) so a KC operand needs ONE b128 per 4 MFMAs.
#include "dc_common.h"

namespace dc {

constexpr int BK = 16;
constexpr int LDK = BK + 4;
constexpr int BM = 64, BN = 128;
constexpr int kMaxSeg = DC_MAX_SEG;

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Mat {
    const float *p;
    int64_t ld;
};

// VEC kernels are only launched when every operand is 16-byte aligned with ld % 4 == 0 and
// extents % 4 == 0, so a float4 is either wholly inside or wholly outside: one predicated
// global_load_dwordx4, no tail code in the hot loop.  The scalar variant handles F = 21 / 25.
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float *p, int nvalid) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
        if (nvalid >= 4) r = *reinterpret_cast<const float4 *>(p);
    } else {
        if (nvalid > 0) r.x = p[0];
        if (nvalid > 1) r.y = p[1];
        if (nvalid > 2) r.z = p[2];
        if (nvalid > 3) r.w = p[3];
    }
    return r;
}

__device__ __forceinline__ int nvalid4(int64_t remaining) {
    return remaining >= 4 ? 4 : (remaining > 0 ? (int)remaining : 0);
}

__device__ __forceinline__ float4 relu_mask(float4 v, float4 m) {
    return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f,
                       m.w > 0.f ? v.w : 0.f);
}

// ---- global -> registers for one BK-deep stage --------------------------------
// KC source: element (row, k) at p[row*ld + k]; tile = ROWS x BK
template <int ROWS, bool VEC>
struct StageKC {
    float4 v[ROWS / 64];
    __device__ __forceinline__ void load(const Mat &m, const Mat *mask, int64_t row0,
                                         int64_t nrows, int64_t k0, int64_t kmax) {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) {
            const int64_t row = row0 + r + 64 * j, k = k0 + 4 * k4;
            const int nv = row < nrows ? nvalid4(kmax - k) : 0;
            v[j] = ld4<VEC>(m.p + row * m.ld + k, nv);
            if (mask) v[j] = relu_mask(v[j], ld4<VEC>(mask->p + row * mask->ld + k, nv));
        }
    }
    __device__ __forceinline__ void store(float *lds) const {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j)
            *reinterpret_cast<float4 *>(lds + (r + 64 * j) * LDK + 4 * k4) = v[j];
    }
};

// RC source: element (k, col) at p[k*ld + col]; tile = BK x COLS
template <int COLS, bool VEC>
struct StageRC {
    static constexpr int PER = COLS / 4;        // float4 per k row
    static constexpr int KPER = 256 / PER;      // k rows covered per pass
    static constexpr int NLD = BK / KPER;
    float4 v[NLD];
    __device__ __forceinline__ void load(const Mat &m, const Mat *mask, int64_t k0, int64_t kmax,
                                         int64_t col0, int64_t ncols) {
        const int c4 = threadIdx.x % PER, kk = threadIdx.x / PER;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int64_t k = k0 + kk + KPER * j, col = col0 + 4 * c4;
            const int nv = k < kmax ? nvalid4(ncols - col) : 0;
            v[j] = ld4<VEC>(m.p + k * m.ld + col, nv);
            if (mask) v[j] = relu_mask(v[j], ld4<VEC>(mask->p + k * mask->ld + col, nv));
        }
    }
    __device__ __forceinline__ void store(float *lds) const {
        const int c4 = threadIdx.x % PER, kk = threadIdx.x / PER;
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            *reinterpret_cast<float4 *>(lds + (kk + KPER * j) * COLS + 4 * c4) = v[j];
    }
};

// ---- one BK stage of MFMAs for this wave's 32 x 64 sub-tile --------------------
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void mma_stage(const float *As, const float *Bs, f32x16 (&acc)[2],
                                          int wm, int wn) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
        float a[4], b[2][4];
        const int kb = 8 * c + 4 * h;
        if (A_KC) {
            const float4 t = *reinterpret_cast<const float4 *>(As + (wm * 32 + r) * LDK + kb);
            a[0] = t.x, a[1] = t.y, a[2] = t.z, a[3] = t.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = As[(kb + t) * BM + wm * 32 + r];
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if (B_KC) {
                const float4 t =
                    *reinterpret_cast<const float4 *>(Bs + (wn * 64 + nb * 32 + r) * LDK + kb);
                b[nb][0] = t.x, b[nb][1] = t.y, b[nb][2] = t.z, b[nb][3] = t.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[nb][t] = Bs[(kb + t) * BN + wn * 64 + nb * 32 + r];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[0][t], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[1][t], acc[1], 0, 0, 0);
        }
    }
}

constexpr int kLdsA_KC = BM * LDK, kLdsB_KC = BN * LDK;
constexpr int kLdsA_RC = BK * BM, kLdsB_RC = BK * BN;

// C/D fragment -> (row, col) of the wave's 32x32 block: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
template <typename F>
__device__ __forceinline__ void for_each_acc(const f32x16 (&acc)[2], int wm, int wn, F &&f) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int col = wn * 64 + nb * 32 + c;
            f(row, col, acc[nb][reg]);
        }
}

// =============================== forward ========================================
struct FwdParams {
    Mat x[kMaxSeg];
    Mat w[kMaxSeg];
    const float *bias;
    float *out;
    int64_t ldo, N, Fi, Fo;
    int nseg, relu;
};

template <bool VEC>
__global__ void __launch_bounds__(256)
k_tag_linear_fwd(FwdParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (kLdsA_KC + kLdsB_KC)];
    const unsigned ntn = (unsigned)((p.Fo + BN - 1) / BN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * BM, col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
    constexpr int kStage = kLdsA_KC + kLdsB_KC, kOffB = kLdsA_KC;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = 0.f, acc[1][i] = 0.f;

    const int kst = (int)((p.Fi + BK - 1) / BK), nst = kst * p.nseg;
    StageKC<BM, VEC> ra;
    StageKC<BN, VEC> rb;
    ra.load(p.x[0], nullptr, row0, p.N, 0, p.Fi);
    rb.load(p.w[0], nullptr, col0, p.Fo, 0, p.Fi);
    ra.store(lds);
    rb.store(lds + kOffB);
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        const int nx = it + 1;
        if (nx < nst) {
            const int s = nx / kst;
            const int64_t k0 = (int64_t)(nx % kst) * BK;
            ra.load(p.x[s], nullptr, row0, p.N, k0, p.Fi);
            rb.load(p.w[s], nullptr, col0, p.Fo, k0, p.Fi);
        }
        mma_stage<true, true>(lds + (it & 1) * kStage, lds + (it & 1) * kStage + kOffB, acc, wm, wn);
        if (nx < nst) {
            ra.store(lds + (nx & 1) * kStage);
            rb.store(lds + (nx & 1) * kStage + kOffB);
        }
        __syncthreads();
    }
    // each lane owns two output columns (one per 32-wide block): fetch their bias once
    float bcol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + wn * 64 + nb * 32 + (threadIdx.x & 31);
        bcol[nb] = (p.bias && col < p.Fo) ? p.bias[col] : 0.f;
    }
    const bool relu = p.relu != 0;
    for_each_acc(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (row < p.N && col < p.Fo) {
            v += bcol[(c >> 5) & 1];
            if (relu) v = fmaxf(v, 0.f);
            p.out[row * p.ldo + col] = v;
        }
    });
}

// =============================== backward: dX ===================================
struct DxParams {
    Mat g, mask;
    int has_mask;
    Mat w[kMaxSeg];
    float *gx[kMaxSeg];
    int64_t ldgx[kMaxSeg];
    int64_t N, Fi, Fo;
    int nseg;
};

template <bool VEC>
__global__ void __launch_bounds__(256)
k_tag_linear_bwd_dx(DxParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (kLdsA_KC + kLdsB_RC)];
    const unsigned ntn = (unsigned)((p.Fi + BN - 1) / BN), per_row = ntn * p.nseg;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / per_row) * BM;
    const int s = (int)((lb % per_row) / ntn);
    const int64_t col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
    constexpr int kStage = kLdsA_KC + kLdsB_RC, kOffB = kLdsA_KC;
    const Mat *mk = p.has_mask ? &p.mask : nullptr;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = 0.f, acc[1][i] = 0.f;

    const int nst = (int)((p.Fo + BK - 1) / BK);
    StageKC<BM, VEC> ra;
    StageRC<BN, VEC> rb;
    ra.load(p.g, mk, row0, p.N, 0, p.Fo);
    rb.load(p.w[s], nullptr, 0, p.Fo, col0, p.Fi);
    ra.store(lds);
    rb.store(lds + kOffB);
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        const int nx = it + 1;
        if (nx < nst) {
            ra.load(p.g, mk, row0, p.N, (int64_t)nx * BK, p.Fo);
            rb.load(p.w[s], nullptr, (int64_t)nx * BK, p.Fo, col0, p.Fi);
        }
        mma_stage<true, false>(lds + (it & 1) * kStage, lds + (it & 1) * kStage + kOffB, acc, wm, wn);
        if (nx < nst) {
            ra.store(lds + (nx & 1) * kStage);
            rb.store(lds + (nx & 1) * kStage + kOffB);
        }
        __syncthreads();
    }
    float *out = p.gx[s];
    const int64_t ldo = p.ldgx[s];
    for_each_acc(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (row < p.N && col < p.Fi) out[row * ldo + col] = v;
    });
}

// =============================== backward: dW ===================================
struct DwParams {
    Mat g, mask;
    int has_mask;
    Mat x[kMaxSeg];
    float *partial;        // [nchunks][nseg][Fo][Fi]
    float *bias_partial;   // [nchunks][Fo] or null
    int64_t N, Fi, Fo, chunk_rows;
    int nseg, nchunks;
};

template <bool VEC>
__global__ void __launch_bounds__(256)
k_tag_linear_bwd_dw(DwParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (kLdsA_RC + kLdsB_RC)];
    const unsigned ntm = (unsigned)((p.Fo + BM - 1) / BM), ntn = (unsigned)((p.Fi + BN - 1) / BN);
    const unsigned tiles = ntm * ntn, per_chunk = tiles * p.nseg;
    const unsigned lb = blockIdx.x;
    const unsigned chunk = lb / per_chunk, rem = lb % per_chunk;
    const int s = (int)(rem / tiles);
    const int64_t o0 = (int64_t)((rem % tiles) / ntn) * BM, f0 = (int64_t)((rem % tiles) % ntn) * BN;
    const int64_t n_beg = (int64_t)chunk * p.chunk_rows;
    const int64_t n_end = (n_beg + p.chunk_rows < p.N) ? n_beg + p.chunk_rows : p.N;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
    constexpr int kStage = kLdsA_RC + kLdsB_RC, kOffB = kLdsA_RC;
    const Mat *mk = p.has_mask ? &p.mask : nullptr;
    const bool do_bias = p.bias_partial && s == 0 && f0 == 0;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = 0.f, acc[1][i] = 0.f;
    float bsum = 0.f;

    const int nst = (int)((n_end - n_beg + BK - 1) / BK);
    StageRC<BM, VEC> ra;
    StageRC<BN, VEC> rb;
    if (nst > 0) {
        ra.load(p.g, mk, n_beg, n_end, o0, p.Fo);
        rb.load(p.x[s], nullptr, n_beg, n_end, f0, p.Fi);
        ra.store(lds);
        rb.store(lds + kOffB);
    }
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        const int nx = it + 1;
        if (nx < nst) {
            ra.load(p.g, mk, n_beg + (int64_t)nx * BK, n_end, o0, p.Fo);
            rb.load(p.x[s], nullptr, n_beg + (int64_t)nx * BK, n_end, f0, p.Fi);
        }
        mma_stage<false, false>(lds + (it & 1) * kStage, lds + (it & 1) * kStage + kOffB, acc, wm, wn);
        if (do_bias && threadIdx.x < BM) {
            const float *a = lds + (it & 1) * kStage;
#pragma unroll
            for (int k = 0; k < BK; ++k) bsum += a[k * BM + threadIdx.x];
        }
        if (nx < nst) {
            ra.store(lds + (nx & 1) * kStage);
            rb.store(lds + (nx & 1) * kStage + kOffB);
        }
        __syncthreads();
    }
    float *out = p.partial + ((int64_t)chunk * p.nseg + s) * p.Fo * p.Fi;
    for_each_acc(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t o = o0 + r, f = f0 + c;
        if (o < p.Fo && f < p.Fi) out[o * p.Fi + f] = v;
    });
    if (do_bias && threadIdx.x < BM && o0 + threadIdx.x < p.Fo)
        p.bias_partial[(int64_t)chunk * p.Fo + o0 + threadIdx.x] = bsum;
}

// sum the per-chunk slabs in chunk order (deterministic), scatter into the per-segment outputs
struct ReduceParams {
    const float *partial, *bias_partial;
    float *gw[kMaxSeg];
    float *gbias;
    int64_t Fi, Fo;
    int nseg, nchunks;
};

__global__ void __launch_bounds__(256)
k_dw_reduce(ReduceParams p) {
    const int64_t per_seg = p.Fo * p.Fi, total = per_seg * p.nseg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        float s = 0.f;
        for (int c = 0; c < p.nchunks; ++c) s += p.partial[(int64_t)c * total + i];
        p.gw[i / per_seg][i % per_seg] = s;
    } else if (p.gbias && i < total + p.Fo) {
        const int64_t o = i - total;
        float s = 0.f;
        for (int c = 0; c < p.nchunks; ++c) s += p.bias_partial[(int64_t)c * p.Fo + o];
        p.gbias[o] = s;
    }
}

static inline Mat make_mat(const float *p, int64_t ld, bool *vec) {
    if (((uintptr_t)p & 15) != 0 || (ld % 4) != 0) *vec = false;
    return Mat{p, ld};
}

static void dw_plan(int64_t N, int64_t Fi, int64_t Fo, int nseg, int64_t *chunk_rows,
                    int *nchunks) {
    const int64_t tiles = ((Fo + BM - 1) / BM) * ((Fi + BN - 1) / BN) * nseg;
    int64_t want = 1024 / (tiles > 0 ? tiles : 1);
    if (want < 1) want = 1;
    int64_t rows = (N + want - 1) / want;
    if (rows < 256) rows = 256;
    rows = (rows + BK - 1) / BK * BK;
    *chunk_rows = rows;
    *nchunks = (int)((N + rows - 1) / rows);
    if (*nchunks < 1) *nchunks = 1;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_tag_linear_fwd(const float *const *xs, const int64_t *ldxs,
                                 const float *const *ws, int nseg, const float *bias, int relu,
                                 float *out, int64_t ldo, int64_t N, int64_t Fi, int64_t Fo,
                                 dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, "dc_tag_linear_fwd: nseg must be 1..%d", kMaxSeg);
    DC_REQUIRE(N >= 0 && Fi >= 1 && Fo >= 1, "dc_tag_linear_fwd: bad sizes");
    if (N == 0) return DC_OK;
    DC_REQUIRE(xs && ldxs && ws && out && ldo >= Fo, "dc_tag_linear_fwd: null pointer / ldo < Fo");
    FwdParams p{};
    bool vec = (Fi % 4 == 0);
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(xs[s] && ws[s] && ldxs[s] >= Fi, "dc_tag_linear_fwd: bad segment %d", s);
        p.x[s] = make_mat(xs[s], ldxs[s], &vec);
        p.w[s] = make_mat(ws[s], Fi, &vec);
    }
    p.bias = bias, p.out = out, p.ldo = ldo, p.N = N, p.Fi = Fi, p.Fo = Fo;
    p.nseg = nseg, p.relu = relu;
    const int64_t grid = ((N + BM - 1) / BM) * ((Fo + BN - 1) / BN);
    DC_REQUIRE(grid < (int64_t)INT32_MAX, "dc_tag_linear_fwd: grid too large");
    if (vec)
        hipLaunchKernelGGL(k_tag_linear_fwd<true>, dim3((unsigned)grid), dim3(256), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_tag_linear_fwd<false>, dim3((unsigned)grid), dim3(256), 0,
                           (hipStream_t)stream, p);
    return check_launch("dc_tag_linear_fwd");
}

extern "C" int dc_tag_linear_bwd_dx(const float *g, int64_t ldg, const float *out_for_mask,
                                    int64_t ldo, const float *const *ws, int nseg,
                                    float *const *gxs, const int64_t *ldgxs, int64_t N, int64_t Fi,
                                    int64_t Fo, dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, "dc_tag_linear_bwd_dx: nseg must be 1..%d", kMaxSeg);
    DC_REQUIRE(N >= 0 && Fi >= 1 && Fo >= 1, "dc_tag_linear_bwd_dx: bad sizes");
    if (N == 0) return DC_OK;
    DC_REQUIRE(g && ws && gxs && ldgxs && ldg >= Fo, "dc_tag_linear_bwd_dx: null pointer / ldg < Fo");
    DC_REQUIRE(!out_for_mask || ldo >= Fo, "dc_tag_linear_bwd_dx: ldo < Fo");
    DxParams p{};
    bool vec = (Fi % 4 == 0) && (Fo % 4 == 0);
    p.g = make_mat(g, ldg, &vec);
    p.has_mask = out_for_mask != nullptr;
    if (out_for_mask) p.mask = make_mat(out_for_mask, ldo, &vec);
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(ws[s] && gxs[s] && ldgxs[s] >= Fi, "dc_tag_linear_bwd_dx: bad segment %d", s);
        p.w[s] = make_mat(ws[s], Fi, &vec);
        p.gx[s] = gxs[s];
        p.ldgx[s] = ldgxs[s];
    }
    p.N = N, p.Fi = Fi, p.Fo = Fo, p.nseg = nseg;
    const int64_t grid = ((N + BM - 1) / BM) * ((Fi + BN - 1) / BN) * nseg;
    DC_REQUIRE(grid < (int64_t)INT32_MAX, "dc_tag_linear_bwd_dx: grid too large");
    if (vec)
        hipLaunchKernelGGL(k_tag_linear_bwd_dx<true>, dim3((unsigned)grid), dim3(256), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_tag_linear_bwd_dx<false>, dim3((unsigned)grid), dim3(256), 0,
                           (hipStream_t)stream, p);
    return check_launch("dc_tag_linear_bwd_dx");
}

extern "C" int64_t dc_tag_linear_bwd_dw_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo,
                                                        int nseg) {
    if (N < 0 || Fi < 1 || Fo < 1 || nseg < 1 || nseg > kMaxSeg) return DC_EINVAL;
    int64_t rows;
    int nchunks;
    dw_plan(N, Fi, Fo, nseg, &rows, &nchunks);
    return (int64_t)sizeof(float) * nchunks * (nseg * Fo * Fi + Fo) + 16;
}

extern "C" int dc_tag_linear_bwd_dw(const float *g, int64_t ldg, const float *out_for_mask,
                                    int64_t ldo, const float *const *xs, const int64_t *ldxs,
                                    int nseg, float *const *gws, float *gbias, void *partials,
                                    int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                    dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, "dc_tag_linear_bwd_dw: nseg must be 1..%d", kMaxSeg);
    DC_REQUIRE(N >= 0 && Fi >= 1 && Fo >= 1, "dc_tag_linear_bwd_dw: bad sizes");
    DC_REQUIRE(g && xs && ldxs && gws && partials && ldg >= Fo,
               "dc_tag_linear_bwd_dw: null pointer / ldg < Fo");
    DC_REQUIRE(!out_for_mask || ldo >= Fo, "dc_tag_linear_bwd_dw: ldo < Fo");
    DC_REQUIRE(partials_bytes >= dc_tag_linear_bwd_dw_workspace_bytes(N, Fi, Fo, nseg),
               "dc_tag_linear_bwd_dw: workspace too small");
    DC_REQUIRE(((uintptr_t)partials & 15) == 0, "dc_tag_linear_bwd_dw: workspace not 16-byte aligned");
    DwParams p{};
    bool vec = (Fi % 4 == 0) && (Fo % 4 == 0);
    p.g = make_mat(g, ldg, &vec);
    p.has_mask = out_for_mask != nullptr;
    if (out_for_mask) p.mask = make_mat(out_for_mask, ldo, &vec);
    ReduceParams r{};
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(xs[s] && gws[s] && ldxs[s] >= Fi, "dc_tag_linear_bwd_dw: bad segment %d", s);
        p.x[s] = make_mat(xs[s], ldxs[s], &vec);
        r.gw[s] = gws[s];
    }
    dw_plan(N, Fi, Fo, nseg, &p.chunk_rows, &p.nchunks);
    p.N = N, p.Fi = Fi, p.Fo = Fo, p.nseg = nseg;
    p.partial = (float *)partials;
    p.bias_partial = gbias ? p.partial + (int64_t)p.nchunks * nseg * Fo * Fi : nullptr;
    const int64_t tiles = ((Fo + BM - 1) / BM) * ((Fi + BN - 1) / BN);
    const int64_t grid = tiles * nseg * p.nchunks;
    if (vec)
        hipLaunchKernelGGL(k_tag_linear_bwd_dw<true>, dim3((unsigned)grid), dim3(256), 0,
                           (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_tag_linear_bwd_dw<false>, dim3((unsigned)grid), dim3(256), 0,
                           (hipStream_t)stream, p);
    r.partial = p.partial, r.bias_partial = p.bias_partial, r.gbias = gbias;
    r.Fi = Fi, r.Fo = Fo, r.nseg = nseg, r.nchunks = p.nchunks;
    const int64_t total = (int64_t)nseg * Fo * Fi + (gbias ? Fo : 0);
    hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, r);
    return check_launch("dc_tag_linear_bwd_dw");
}
