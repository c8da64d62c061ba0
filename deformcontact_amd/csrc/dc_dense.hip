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
#include <stdlib.h>

#include "dc_dense.h"

#include <atomic>
// launches of the GENERIC (bounds-checked, scalar-load) dense kernels since the library was loaded / the counter was
// reset: the default step must never reach them (dc_generic_dense_launches; tests pin that)
static std::atomic<long long> g_generic_dense_launches{0};

namespace dc {

// Loads are UNCONDITIONAL with clamped (always in-bounds) addresses and the out-of-range /
// ReLU-masked lanes are zeroed when the registers are written to LDS: no branches around
// loads, so the prefetch stays in flight behind the MFMAs.
// VEC kernels are only launched when every operand is 16-byte aligned with ld % 4 == 0 and
// extents % 4 == 0, so a float4 is either wholly inside or wholly outside.  The scalar
// variant handles F = 21 / 25 (four dword loads, per-element validity).
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float *row_ptr, int64_t k, int64_t kmax) {
    if (VEC) {
        const int64_t kc = (k + 4 <= kmax) ? k : 0;
        return *reinterpret_cast<const float4 *>(row_ptr + kc);
    } else {
        const int64_t last = kmax - 1;
        return make_float4(row_ptr[k + 0 <= last ? k + 0 : last], row_ptr[k + 1 <= last ? k + 1 : last],
                           row_ptr[k + 2 <= last ? k + 2 : last], row_ptr[k + 3 <= last ? k + 3 : last]);
    }
}

__device__ __forceinline__ float4 zero_invalid(float4 v, bool row_ok, int64_t k, int64_t kmax) {
    return make_float4((row_ok && k + 0 < kmax) ? v.x : 0.f, (row_ok && k + 1 < kmax) ? v.y : 0.f,
                       (row_ok && k + 2 < kmax) ? v.z : 0.f, (row_ok && k + 3 < kmax) ? v.w : 0.f);
}

__device__ __forceinline__ float4 relu_mask(float4 v, float4 m) {
    return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f,
                       m.w > 0.f ? v.w : 0.f);
}

// ---- global -> registers for one BK-deep stage --------------------------------
// The register sets hold ONLY the loaded float4s; tile coordinates are recomputed from the
// (wave-uniform) stage arguments when the set is written to LDS, so a second set in flight
// costs 4 VGPRs per float4 and nothing else.
struct TileArgs {          // what load() and store() both need; all wave-uniform
    int64_t r0, rmax;      // first row / row bound of the tile's slow index
    int64_t c0, cmax;      // first element / bound of the tile's contiguous index
};

// KC source: element (row, k) at p[row*ld + k]; tile = ROWS x BK (requires rmax, cmax >= 1)
template <int ROWS, bool VEC, bool MASK>
struct StageKC {
    float4 v[ROWS / 64];
    float4 m[MASK ? ROWS / 64 : 1];
    __device__ __forceinline__ void load(const Mat &mat, const Mat &mask, const TileArgs &t) {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
        const int64_t k = t.c0 + 4 * k4;
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j) {
            const int64_t row = t.r0 + r + 64 * j;
            const int64_t rc = row < t.rmax ? row : t.rmax - 1;
            v[j] = ld4<VEC>(mat.p + rc * mat.ld, k, t.cmax);
            if (MASK) m[j] = ld4<VEC>(mask.p + rc * mask.ld, k, t.cmax);
        }
    }
    __device__ __forceinline__ float4 value(int j, const TileArgs &t) const {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
        float4 x = zero_invalid(v[j], t.r0 + r + 64 * j < t.rmax, t.c0 + 4 * k4, t.cmax);
        if (MASK) x = relu_mask(x, m[j]);
        return x;
    }
    __device__ __forceinline__ void store(float *lds, const TileArgs &t) const {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < ROWS / 64; ++j)
            *reinterpret_cast<float4 *>(lds + (r + 64 * j) * LDK + 4 * k4) = value(j, t);
    }
};

// RC source: element (k, col) at p[k*ld + col]; tile = BK x COLS: r0/rmax index k, c0/cmax
// index the contiguous column
template <int COLS, bool VEC, bool MASK>
struct StageRC {
    static constexpr int PER = COLS / 4;        // float4 per k row
    static constexpr int KPER = 256 / PER;      // k rows covered per pass
    static constexpr int NLD = BK / KPER;
    float4 v[NLD];
    float4 m[MASK ? NLD : 1];
    __device__ __forceinline__ void load(const Mat &mat, const Mat &mask, const TileArgs &t) {
        const int c4 = threadIdx.x % PER, kr = threadIdx.x / PER;
        const int64_t col = t.c0 + 4 * c4;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int64_t k = t.r0 + kr + KPER * j;
            const int64_t kc = k < t.rmax ? k : t.rmax - 1;
            v[j] = ld4<VEC>(mat.p + kc * mat.ld, col, t.cmax);
            if (MASK) m[j] = ld4<VEC>(mask.p + kc * mask.ld, col, t.cmax);
        }
    }
    __device__ __forceinline__ float4 value(int j, const TileArgs &t) const {
        const int c4 = threadIdx.x % PER, kr = threadIdx.x / PER;
        float4 x = zero_invalid(v[j], t.r0 + kr + KPER * j < t.rmax, t.c0 + 4 * c4, t.cmax);
        if (MASK) x = relu_mask(x, m[j]);
        return x;
    }
    __device__ __forceinline__ void store(float *lds, const TileArgs &t) const {
        const int c4 = threadIdx.x % PER, kr = threadIdx.x / PER;
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            *reinterpret_cast<float4 *>(lds + (kr + KPER * j) * COLS + 4 * c4) = value(j, t);
    }
};

constexpr int kRing = 3;   // LDS stages
static_assert(BK == 16, "the pipeline below is written for two 8-wide chunks per stage");

// Software pipeline shared by the three kernels.  Per stage: [read chunk 1 | MFMA chunk 0]
// [read chunk 0 of stage it+1 | MFMA chunk 1], then refill the ring: registers (global data
// of stage it+2, loaded one step ago) -> LDS, issue the global loads of stage it+3, barrier.
// Software pipeline shared by the three kernels.
//   global -> registers : issued FOUR stages ahead into one of two named register sets, so a
//                         load has two full MFMA phases to land (L2/HBM latency under load is
//                         longer than one phase: with a one-stage prefetch the kernel ran at
//                         ~60 % of the MFMA rate, serialising memory and matrix time);
//   registers -> LDS    : two stages ahead, into a 3-deep ring;
//   LDS -> fragments    : one 8-wide chunk ahead of the MFMAs that consume it.
struct NoHook {
    template <typename S>
    __device__ __forceinline__ void operator()(const S &, const TileArgs &) const {}
};

// Software pipeline shared by the three kernels.
//   global -> registers : issued FOUR stages ahead into one of two named register sets, so a
//                         load has two full MFMA phases to land (L2/HBM latency under load is
//                         longer than one phase: with a one-stage prefetch the kernel ran at
//                         ~60 % of the MFMA rate, serialising memory and matrix time);
//   registers -> LDS    : two stages ahead, into a 3-deep ring;
//   LDS -> fragments    : one 8-wide chunk ahead of the MFMAs that consume it.
// `tile(st, ta, tb)` fills the wave-uniform tile coordinates of stage st; `ldA/ldB(set, t)`
// issue the loads.
template <int MB, bool A_KC, bool B_KC, int STAGE, int OFFB, typename SA, typename SB,
          typename TileFn, typename LdA, typename LdB, typename Hook = NoHook>
__device__ __forceinline__ void gemm_pipeline(float *lds, int nst, TileFn &&tile, LdA &&ldA,
                                              LdB &&ldB, f32x16 (&acc)[MB][2], int wm, int wn,
                                              Hook &&on_store_a = NoHook{}) {
    SA a0, a1;
    SB b0, b1;
    auto gload = [&](int st, SA &ra, SB &rb) {
        TileArgs ta, tb;
        tile(st, ta, tb);
        ldA(ra, ta);
        ldB(rb, tb);
    };
    auto lstore = [&](int st, const SA &ra, const SB &rb) {
        TileArgs ta, tb;
        tile(st, ta, tb);
        ra.store(lds + (st % kRing) * STAGE, ta);
        rb.store(lds + (st % kRing) * STAGE + OFFB, tb);
        on_store_a(ra, ta);             // every stage passes here exactly once
    };
    if (nst > 0) gload(0, a0, b0);
    if (nst > 1) gload(1, a1, b1);
    if (nst > 0) lstore(0, a0, b0);
    if (nst > 2) gload(2, a0, b0);
    if (nst > 1) lstore(1, a1, b1);
    if (nst > 3) gload(3, a1, b1);
    __syncthreads();
    Frag<MB> f0, f1;
    if (nst > 0) load_frag<MB, A_KC, B_KC>(f0, lds, lds + OFFB, 0, wm, wn);
    auto step = [&](int it, SA &ra, SB &rb) {
        if (it + 2 < nst) lstore(it + 2, ra, rb);      // loaded two steps ago
        if (it + 4 < nst) gload(it + 4, ra, rb);
        const float *cur = lds + (it % kRing) * STAGE;
        load_frag<MB, A_KC, B_KC>(f1, cur, cur + OFFB, 1, wm, wn);
        mma_frag<MB>(f0, acc);
        if (it + 1 < nst) {
            const float *nxt = lds + ((it + 1) % kRing) * STAGE;
            load_frag<MB, A_KC, B_KC>(f0, nxt, nxt + OFFB, 0, wm, wn);
        }
        mma_frag<MB>(f1, acc);
        __syncthreads();
    };
    for (int it = 0; it < nst; it += 2) {     // two named register sets: nothing runtime-indexed
        step(it, a0, b0);
        if (it + 1 >= nst) break;
        step(it + 1, a1, b1);
    }
}

// =============================== forward ========================================

template <int MB, bool VEC>
__global__ void __launch_bounds__(256)
k_tag_linear_fwd(FwdParams p) {
    using T = Tile<MB>;
    constexpr int BM = T::BM, kStage = T::A_KC + T::B_KC, kOffB = T::A_KC;
    __shared__ __attribute__((aligned(16))) float lds[kRing * kStage];
    const unsigned ntn = (unsigned)((p.Fo + BN - 1) / BN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * BM, col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);

    const int kst = (int)((p.Fi + BK - 1) / BK), nst = kst * p.nseg;
    using SA = StageKC<BM, VEC, false>;
    using SB = StageKC<BN, VEC, false>;
    int seg = 0;   // segment of the stage most recently described by tile()
    auto tile = [&](int st, TileArgs &ta, TileArgs &tb) {
        seg = st / kst;
        const int64_t k0 = (int64_t)(st % kst) * BK;
        ta = TileArgs{row0, p.N, k0, p.Fi};
        tb = TileArgs{col0, p.Fo, k0, p.Fi};
    };
    auto ldA = [&](SA &ra, const TileArgs &t) { ra.load(p.x[seg], p.x[seg], t); };
    auto ldB = [&](SB &rb, const TileArgs &t) { rb.load(p.w[seg], p.w[seg], t); };
    gemm_pipeline<MB, true, true, kStage, kOffB, SA, SB>(lds, nst, tile, ldA, ldB, acc, wm, wn);
    // each lane owns two output columns (one per 32-wide block): fetch their bias once
    float bcol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + wn * 64 + nb * 32 + (threadIdx.x & 31);
        bcol[nb] = (p.bias && col < p.Fo) ? p.bias[col] : 0.f;
    }
    const bool relu = p.relu != 0;
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (row < p.N && col < p.Fo) {
            v += bcol[(c >> 5) & 1];
            if (relu) v = fmaxf(v, 0.f);
            p.out[row * p.ldo + col] = v;
        }
    });
}

// =============================== backward: dX ===================================

template <int MB, bool VEC, bool MASK>
__global__ void __launch_bounds__(256)
k_tag_linear_bwd_dx(DxParams p) {
    using T = Tile<MB>;
    constexpr int BM = T::BM, kStage = T::A_KC + T::B_RC, kOffB = T::A_KC;
    __shared__ __attribute__((aligned(16))) float lds[kRing * kStage];
    const unsigned ntn = (unsigned)((p.Fi + BN - 1) / BN), per_row = ntn * p.nseg;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / per_row) * BM;
    const int s = (int)((lb % per_row) / ntn);
    const int64_t col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);

    const int nst = (int)((p.Fo + BK - 1) / BK);
    using SA = StageKC<BM, VEC, MASK>;
    using SB = StageRC<BN, VEC, false>;
    auto tile = [&](int st, TileArgs &ta, TileArgs &tb) {
        ta = TileArgs{row0, p.N, (int64_t)st * BK, p.Fo};
        tb = TileArgs{(int64_t)st * BK, p.Fo, col0, p.Fi};
    };
    auto ldA = [&](SA &ra, const TileArgs &t) { ra.load(p.g, p.mask, t); };
    auto ldB = [&](SB &rb, const TileArgs &t) { rb.load(p.w[s], p.w[s], t); };
    gemm_pipeline<MB, true, false, kStage, kOffB, SA, SB>(lds, nst, tile, ldA, ldB, acc, wm, wn);
    float *out = p.gx[s];
    const int64_t ldo = p.ldgx[s];
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (row < p.N && col < p.Fi) out[row * ldo + col] = v;
    });
}

// =============================== backward: dW ===================================

template <int MB, bool VEC, bool MASK>
__global__ void __launch_bounds__(256)
k_tag_linear_bwd_dw(DwParams p) {
    using T = Tile<MB>;
    constexpr int BM = T::BM, kStage = T::A_RC + T::B_RC, kOffB = T::A_RC;
    __shared__ __attribute__((aligned(16))) float lds[kRing * kStage];
    const unsigned ntm = (unsigned)((p.Fo + BM - 1) / BM), ntn = (unsigned)((p.Fi + BN - 1) / BN);
    const unsigned tiles = ntm * ntn, per_chunk = tiles * p.nseg;
    const unsigned lb = blockIdx.x;
    const unsigned chunk = lb / per_chunk, rem = lb % per_chunk;
    const int s = (int)(rem / tiles);
    const int64_t o0 = (int64_t)((rem % tiles) / ntn) * BM, f0 = (int64_t)((rem % tiles) % ntn) * BN;
    const int64_t n_beg = (int64_t)chunk * p.chunk_rows;
    const int64_t n_end = (n_beg + p.chunk_rows < p.N) ? n_beg + p.chunk_rows : p.N;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;
    const bool do_bias = p.bias_partial && s == 0 && f0 == 0;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);
    float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);

    const int nst = (int)((n_end - n_beg + BK - 1) / BK);
    using SA = StageRC<BM, VEC, MASK>;
    using SB = StageRC<BN, VEC, false>;
    auto tile = [&](int st, TileArgs &ta, TileArgs &tb) {
        ta = TileArgs{n_beg + (int64_t)st * BK, n_end, o0, p.Fo};
        tb = TileArgs{n_beg + (int64_t)st * BK, n_end, f0, p.Fi};
    };
    auto ldA = [&](SA &ra, const TileArgs &t) { ra.load(p.g, p.mask, t); };
    auto ldB = [&](SB &rb, const TileArgs &t) { rb.load(p.x[s], p.x[s], t); };
    // bias gradient = column sums of the (masked) g tile: each thread adds up the float4s it
    // stages (4 columns x NLD node rows per stage); reduced across threads after the loop
    auto bias_hook = [&](const SA &ra, const TileArgs &t) {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < SA::NLD; ++j) {
                const float4 x = ra.value(j, t);
                bsum4.x += x.x, bsum4.y += x.y, bsum4.z += x.z, bsum4.w += x.w;
            }
        }
    };
    gemm_pipeline<MB, false, false, kStage, kOffB, SA, SB>(lds, nst, tile, ldA, ldB, acc, wm, wn,
                                                          bias_hook);
    float *out = p.partial + ((int64_t)chunk * p.nseg + s) * p.Fo * p.Fi;
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t o = o0 + r, f = f0 + c;
        if (o < p.Fo && f < p.Fi) out[o * p.Fi + f] = v;
    });
    if (do_bias) {
        // thread t holds columns 4*(t % PER)..+3 of k-rows t / PER: reduce the 256/PER row
        // groups through LDS (all stages are consumed; the ring is free)
        __syncthreads();
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[threadIdx.x] = bsum4;
        __syncthreads();
        if (threadIdx.x < SA::PER) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < 256 / SA::PER; ++g) {
                const float4 v = red[g * SA::PER + threadIdx.x];
                t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
            }
            float *bp = p.bias_partial + (int64_t)chunk * p.Fo;
            const int64_t o = o0 + 4 * threadIdx.x;
            if (o + 0 < p.Fo) bp[o + 0] = t.x;
            if (o + 1 < p.Fo) bp[o + 1] = t.y;
            if (o + 2 < p.Fo) bp[o + 2] = t.z;
            if (o + 3 < p.Fo) bp[o + 3] = t.w;
        }
    }
}

// Sum the per-chunk slabs in chunk order (deterministic) and write each output block:
// block j = columns [(j % bps)*cols, +cols) of segment j / bps, as a contiguous [Fo, cols]
// matrix; `accumulate` adds into the destination (gradient accumulation fused here instead of
// one elementwise add per parameter afterwards).
struct ReduceParams {
    const float *partial, *bias_partial;
    float *gw[2 * kMaxSeg];
    float *gbias;
    int64_t Fi, Fo, cols;
    int nseg, nchunks, ngw, bps, accumulate;
};

// Round 6: FOUR threads per output (one wave each: a block is 64 outputs x 4 chunk quarters).  One thread per output walked up to
// 128 chunks - 16 dependent rounds of 8 strided loads, 8 us for 13 MB; a quarter sums its consecutive chunks in order, the four
// partial sums meet in LDS and are added in quarter order: still one fixed summation order per element, a quarter of the latency.
constexpr int kReduceOutputs = 64;                 // outputs per 256-thread block
__device__ __forceinline__ void dw_reduce_body(const ReduceParams &p) {
    __shared__ float red[4][kReduceOutputs];
    const int64_t per_out = p.Fo * p.cols, total = per_out * p.ngw;
    const int64_t seg_elems = p.Fo * p.Fi, slab = seg_elems * p.nseg;
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kReduceOutputs + el;
    const float *src = nullptr;
    float *dst = nullptr;
    int64_t stride = 0;
    if (i < total) {
        const int j = (int)(i / per_out);
        const int64_t rem = i % per_out, o = rem / p.cols, c = rem % p.cols;
        src = p.partial + (int64_t)(j / p.bps) * seg_elems + o * p.Fi + (int64_t)(j % p.bps) * p.cols + c;
        dst = p.gw[j] + rem;
        stride = slab;
    } else if (p.gbias && i < total + p.Fo) {
        src = p.bias_partial + (i - total);
        dst = p.gbias + (i - total);
        stride = p.Fo;
    }
    const int per = (p.nchunks + 3) / 4;
    const int c_beg = q * per < p.nchunks ? q * per : p.nchunks;
    const int c_end = c_beg + per < p.nchunks ? c_beg + per : p.nchunks;
    float s = 0.f;
    if (src) {
        int c = c_beg;
        for (; c + 8 <= c_end; c += 8) {          // 8 independent loads in flight, fixed add order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(c + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < c_end; ++c) s += src[(int64_t)c * stride];
    }
    red[q][el] = s;
    __syncthreads();
    if (q == 0 && dst) {
        const float t = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
        *dst = p.accumulate ? *dst + t : t;
    }
}

// one thread per output, 256 outputs per block: few chunks (the wide layers' 32) - the walk is short and the wider loads win
__device__ __forceinline__ void dw_reduce_body1(const ReduceParams &p) {
    const int64_t per_out = p.Fo * p.cols, total = per_out * p.ngw;
    const int64_t seg_elems = p.Fo * p.Fi, slab = seg_elems * p.nseg;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float *src;
    float *dst;
    int64_t stride;
    if (i < total) {
        const int j = (int)(i / per_out);
        const int64_t rem = i % per_out, o = rem / p.cols, c = rem % p.cols;
        src = p.partial + (int64_t)(j / p.bps) * seg_elems + o * p.Fi + (int64_t)(j % p.bps) * p.cols + c;
        dst = p.gw[j] + rem;
        stride = slab;
    } else if (p.gbias && i < total + p.Fo) {
        src = p.bias_partial + (i - total);
        dst = p.gbias + (i - total);
        stride = p.Fo;
    } else {
        return;
    }
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= p.nchunks; c += 8) {          // 8 independent loads in flight, fixed add order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < p.nchunks; ++c) s += src[(int64_t)c * stride];
    *dst = p.accumulate ? *dst + s : s;
}

__global__ void __launch_bounds__(256)
k_dw_reduce(ReduceParams p) { dw_reduce_body1(p); }
__global__ void __launch_bounds__(256)
k_dw_reduce4(ReduceParams p) { dw_reduce_body(p); }
// the form by chunk count: from 64 chunks on (the first layers' 128) four threads share an output
constexpr int kReduce4From = 64;
static inline void launch_reduce(const ReduceParams &r, int64_t total, hipStream_t hs) {
    if (r.nchunks >= kReduce4From)
        DC_LAUNCH(k_dw_reduce4, dim3((unsigned)((total + kReduceOutputs - 1) / kReduceOutputs)), dim3(256), 0, hs, r);
    else
        DC_LAUNCH(k_dw_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs, r);
}

// one launch for the groups of a grouped dW (blockIdx.y = group): each group sums its own chunk range
struct ReduceGroups {
    ReduceParams g[kMaxGroups];
};
__global__ void __launch_bounds__(256)
k_dw_reduce_grouped(ReduceGroups rg) { dw_reduce_body1(rg.g[blockIdx.y]); }

static inline Mat make_mat(const float *p, int64_t ld, bool *vec) {
    if (((uintptr_t)p & 15) != 0 || (ld % 4) != 0) *vec = false;
    return Mat{p, ld};
}

// tile height: 128-row tiles halve the L2->LDS traffic per FLOP (the kernel is only MFMA-bound
// with that slack); 64-row tiles for small problems keep enough blocks in flight.
static inline int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

static inline int pick_mb(int64_t rows, int64_t col_tiles) {
    return (((rows + 127) / 128) * col_tiles >= 512) ? 2 : 1;
}

// 128-row tiles move 1/3 less L2 traffic per FLOP, 64-row tiles balance better when the tile
// count is not a multiple of the 256 CUs (the rigid branch: 381 tiles of 64 rows): compare the
// blocks-per-CU makespans, with a 12 % efficiency handicap on the small tile.
static inline int split_mb(int64_t rows, int64_t col_tiles) {
    const int64_t nb2 = ((rows + 127) / 128) * col_tiles, nb1 = ((rows + 63) / 64) * col_tiles;
    const double t2 = (double)((nb2 + 255) / 256);
    const double t1 = (double)((nb1 + 255) / 256) * 0.5 * 1.12;
    return t1 < t2 ? 1 : 2;
}

static inline int dw_mb(int64_t Fo) {
    return Fo >= 128 ? 2 : 1;
}

static void dw_plan(int64_t N, int64_t Fi, int64_t Fo, int nseg, int64_t *chunk_rows,
                    int *nchunks) {
    const int64_t BM = 64 * dw_mb(Fo);
    const int64_t tiles = ((Fo + BM - 1) / BM) * ((Fi + BN - 1) / BN) * nseg;
    constexpr int target = 512, maxchunks = 128;
    int64_t want = target / (tiles > 0 ? tiles : 1);
    if (want < 1) want = 1;
    if (want > maxchunks) want = maxchunks;   // keeps the slab-reduce pass short
    if (want == 1 && N >= 8192) {
        // a wide output (the attention's dK / dV: Fo = 24,384 keys -> 191 tiles of 128 x 256 on 256 CUs, 75 % of one
        // wave of workgroups): cut the rows into the chunk count whose tiles fill whole waves best (4 -> 764 of 768)
        const int64_t t2 = ((Fo + 127) / 128) * ((Fi + 255) / 256) * nseg;
        double best = 0.0;
        for (int64_t c = 1; c <= 8 && N / c >= 2048; ++c) {
            const int64_t w = (t2 * c + 255) / 256;
            const double eff = (double)(t2 * c) / (double)(w * 256);
            if (eff > best + 0.02) best = eff, want = c;
        }
    }
    int64_t rows = (N + want - 1) / want;
    if (rows < 256) rows = 256;
    // whole 32-node stages: k_dw_h2w (the only kernel that forms the corrected gradient operand of the attention
    // backward) walks a chunk 32 nodes at a time; 16 was enough for the other kernels and sent e.g. N = 8,192 in 5
    // chunks (1,648 rows) to the fallback - an error for dc_tag_linear_bwd_dw_h2_corr
    rows = (rows + 2 * BK - 1) / (2 * BK) * (2 * BK);
    *chunk_rows = rows;
    *nchunks = (int)((N + rows - 1) / rows);
    if (*nchunks < 1) *nchunks = 1;
}

}  // namespace dc

using namespace dc;

static int fwd_impl(const float *const *xs, const int64_t *ldxs, const float *const *ws, int nseg,
                    const float *bias, int relu, float *out, int64_t ldo, int64_t N, int64_t Fi,
                    int64_t Fo, dc_stream_t stream, int products, H2Scales h2 = H2Scales{},
                    float *ksplit_ws = nullptr, int64_t ksplit_ws_bytes = 0, const float *exp_lse = nullptr,
                    int64_t exp_ncols = 0) {
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
    p.h2 = h2;
    p.exp_lse = exp_lse, p.exp_ncols = exp_ncols;
    DC_REQUIRE(!exp_lse || h2.b_presplit, "dc_tag_linear_fwd: the exp epilogue exists on the pre-split h2 kernels only");
    const int64_t ntn = (Fo + BN - 1) / BN;
    const int mb = pick_mb(N, ntn);
    const int64_t grid = ((N + 64 * mb - 1) / (64 * mb)) * ntn;
    DC_REQUIRE(grid < (int64_t)INT32_MAX, "dc_tag_linear_fwd: grid too large");
    const dim3 gd((unsigned)grid), bd(256);
    hipStream_t hs = (hipStream_t)stream;
    if (products == 2 && vec) {
        int smb = split_mb(N, ntn);
        // a long reduction with a small output (e.g. attention weights x values: [2048, 24384] x
        // [24384, 256]) has too few tiles for 256 CUs: cut the reduction, sum the partials
        const int64_t nst = Fi / BK;
        if (ksplit_ws && !bias && !relu && ldo == Fo && nseg == 1) {
            smb = 2;
            int64_t tiles = ((N + 127) / 128) * ntn;
            if (tiles < 128) smb = 1, tiles = ((N + 63) / 64) * ntn;
            int64_t ks = tiles < 256 ? (512 + tiles - 1) / tiles : 1;
            if (ks > nst / 8) ks = nst / 8;
            if (ks > 32) ks = 32;
            while (ks > 1 && ks * N * Fo * (int64_t)sizeof(float) > ksplit_ws_bytes) --ks;
            if (ks > 1) p.ksplit = (int)ks, p.kpartial = ksplit_ws;
        }
        if (fwd_h2_launch(p, smb, hs)) {
            if (p.ksplit > 1) {
                ReduceParams r{};
                r.partial = p.kpartial, r.gw[0] = out, r.Fi = Fo, r.Fo = N, r.cols = Fo;
                r.nseg = 1, r.nchunks = p.ksplit, r.ngw = 1, r.bps = 1, r.accumulate = 0;
                const int64_t total = N * Fo;
                launch_reduce(r, total, hs);
            }
            return check_launch("dc_tag_linear_fwd_h2");
        }
        p.ksplit = 0, p.kpartial = nullptr;
    }
    DC_REQUIRE(!h2.b_presplit, "dc_tag_linear_fwd_h2p: shape not eligible for the pre-split kernel");
    if (products && vec && fwd_split_launch(p, split_mb(N, ntn), products, hs))
        return check_launch("dc_tag_linear_fwd_split");
    DC_REQUIRE(products != 2, "dc_tag_linear_fwd_h2: needs Fi %% 16 == 0, 16-byte aligned operands, "
                              "equal leading dimensions (Fi=%lld)", (long long)Fi);
    if (vec && fwd_fast_launch(p, mb, hs)) return check_launch("dc_tag_linear_fwd");
    static const int trace = env_int("DC_DENSE_TRACE", 0);
    ++g_generic_dense_launches;
    if (trace)
        fprintf(stderr, "[dc] generic fwd dense kernel: N=%lld Fi=%lld Fo=%lld nseg=%d vec=%d ldx=%lld ldo=%lld products=%d\n",
                (long long)N, (long long)Fi, (long long)Fo, nseg, (int)vec, (long long)ldxs[0], (long long)ldo, products);
    if (mb == 2 && vec)
        DC_LAUNCH((k_tag_linear_fwd<2, true>), gd, bd, 0, hs, p);
    else if (mb == 2)
        DC_LAUNCH((k_tag_linear_fwd<2, false>), gd, bd, 0, hs, p);
    else if (vec)
        DC_LAUNCH((k_tag_linear_fwd<1, true>), gd, bd, 0, hs, p);
    else
        DC_LAUNCH((k_tag_linear_fwd<1, false>), gd, bd, 0, hs, p);
    return check_launch("dc_tag_linear_fwd");
}

static int dx_impl(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                   const float *const *ws, int nseg, float *const *gxs, const int64_t *ldgxs,
                   int64_t N, int64_t Fi, int64_t Fo, dc_stream_t stream, float *split_ws,
                   int products, H2Scales h2 = H2Scales{}) {
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
    p.h2 = h2;
    const int64_t ntn = ((Fi + BN - 1) / BN) * nseg;
    const int mb = pick_mb(N, ntn);
    const int64_t grid = ((N + 64 * mb - 1) / (64 * mb)) * ntn;
    DC_REQUIRE(grid < (int64_t)INT32_MAX, "dc_tag_linear_bwd_dx: grid too large");
    if (!p.has_mask) p.mask = p.g;
    const dim3 gd((unsigned)grid), bd(256);
    hipStream_t hs = (hipStream_t)stream;
    if (split_ws && vec && dx_split_launch(p, split_ws, split_mb(N, ntn), products, hs))
        return check_launch("dc_tag_linear_bwd_dx_split");
    DC_REQUIRE(products != 2, "dc_tag_linear_bwd_dx_h2: needs Fo %% 16 == 0, Fi %% 4 == 0 and 16-byte "
                              "aligned operands (Fi=%lld Fo=%lld)", (long long)Fi, (long long)Fo);
    if (vec && dx_fast_launch(p, mb, hs)) return check_launch("dc_tag_linear_bwd_dx");
    ++g_generic_dense_launches;
#define DC_DX(MB_, V_, M_) DC_LAUNCH((k_tag_linear_bwd_dx<MB_, V_, M_>), gd, bd, 0, hs, p)
    if (mb == 2) {
        if (vec && p.has_mask) DC_DX(2, true, true);
        else if (vec) DC_DX(2, true, false);
        else if (p.has_mask) DC_DX(2, false, true);
        else DC_DX(2, false, false);
    } else {
        if (vec && p.has_mask) DC_DX(1, true, true);
        else if (vec) DC_DX(1, true, false);
        else if (p.has_mask) DC_DX(1, false, true);
        else DC_DX(1, false, false);
    }
#undef DC_DX
    return check_launch("dc_tag_linear_bwd_dx");
}

extern "C" int dc_tag_linear_fwd(const float *const *xs, const int64_t *ldxs,
                                 const float *const *ws, int nseg, const float *bias, int relu,
                                 float *out, int64_t ldo, int64_t N, int64_t Fi, int64_t Fo,
                                 dc_stream_t stream) {
    return fwd_impl(xs, ldxs, ws, nseg, bias, relu, out, ldo, N, Fi, Fo, stream, 0);
}

extern "C" int dc_tag_linear_fwd_split(const float *const *xs, const int64_t *ldxs,
                                       const float *const *ws, int nseg, const float *bias,
                                       int relu, float *out, int64_t ldo, int64_t N, int64_t Fi,
                                       int64_t Fo, int products, dc_stream_t stream) {
    DC_REQUIRE(products == 6 || products == 3 || products == 1,
               "dc_tag_linear_fwd_split: products must be 6, 3 or 1");
    return fwd_impl(xs, ldxs, ws, nseg, bias, relu, out, ldo, N, Fi, Fo, stream, products);
}

extern "C" int dc_tag_linear_bwd_dx(const float *g, int64_t ldg, const float *out_for_mask,
                                    int64_t ldo, const float *const *ws, int nseg,
                                    float *const *gxs, const int64_t *ldgxs, int64_t N, int64_t Fi,
                                    int64_t Fo, dc_stream_t stream) {
    return dx_impl(g, ldg, out_for_mask, ldo, ws, nseg, gxs, ldgxs, N, Fi, Fo, stream, nullptr, 0);
}

extern "C" int dc_tag_linear_bwd_dw(const float *g, int64_t ldg, const float *out_for_mask,
                                    int64_t ldo, const float *const *xs, const int64_t *ldxs,
                                    int nseg, float *const *gws, int ngw, int64_t gw_cols,
                                    float *gbias, int accumulate, void *partials,
                                    int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                    dc_stream_t stream);
extern "C" int dc_tag_linear_bwd_dw_split(const float *g, int64_t ldg, const float *out_for_mask,
                                          int64_t ldo, const float *const *xs, const int64_t *ldxs,
                                          int nseg, float *const *gws, int ngw, int64_t gw_cols,
                                          float *gbias, int accumulate, void *partials,
                                          int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                          int products, dc_stream_t stream);

extern "C" int64_t dc_tag_linear_bwd_dx_split_workspace_bytes(int64_t Fi, int64_t Fo, int nseg) {
    if (Fi < 1 || Fo < 1 || nseg < 1 || nseg > kMaxSeg) return DC_EINVAL;
    return (int64_t)sizeof(float) * nseg * Fi * Fo + 16;
}

extern "C" int dc_tag_linear_bwd_dx_split(const float *g, int64_t ldg, const float *out_for_mask,
                                          int64_t ldo, const float *const *ws, int nseg,
                                          float *const *gxs, const int64_t *ldgxs, void *workspace,
                                          int64_t workspace_bytes, int64_t N, int64_t Fi,
                                          int64_t Fo, int products, dc_stream_t stream) {
    DC_REQUIRE(products == 6 || products == 3 || products == 1,
               "dc_tag_linear_bwd_dx_split: products must be 6, 3 or 1");
    DC_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 &&
                   workspace_bytes >= dc_tag_linear_bwd_dx_split_workspace_bytes(Fi, Fo, nseg),
               "dc_tag_linear_bwd_dx_split: workspace missing, misaligned or too small");
    return dx_impl(g, ldg, out_for_mask, ldo, ws, nseg, gxs, ldgxs, N, Fi, Fo, stream,
                   (float *)workspace, products);
}

extern "C" int64_t dc_tag_linear_bwd_dw_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo,
                                                        int nseg) {
    if (N < 0 || Fi < 1 || Fo < 1 || nseg < 1 || nseg > kMaxSeg) return DC_EINVAL;
    int64_t rows;
    int nchunks;
    dw_plan(N, Fi, Fo, nseg, &rows, &nchunks);
    // + 1 slot: the < 16 trailing rows of an N that is not a multiple of the 16-row stage get their own partial
    return (int64_t)sizeof(float) * (nchunks + 1) * (nseg * Fo * Fi + Fo) + 16;
}

static int dw_impl(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                   const float *const *xs, const int64_t *ldxs, int nseg, float *const *gws, int ngw,
                   int64_t gw_cols, float *gbias, int accumulate, void *partials,
                   int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo, dc_stream_t stream,
                   int products, H2Scales h2 = H2Scales{}, const float *g2 = nullptr,
                   const float *g2_coef = nullptr) {
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
        DC_REQUIRE(xs[s] && ldxs[s] >= Fi, "dc_tag_linear_bwd_dw: bad segment %d", s);
        p.x[s] = make_mat(xs[s], ldxs[s], &vec);
    }
    DC_REQUIRE(ngw >= nseg && ngw <= 2 * kMaxSeg && ngw % nseg == 0 && gw_cols >= 1 &&
                   (ngw / nseg) * gw_cols <= Fi,
               "dc_tag_linear_bwd_dw: %d output blocks of %lld columns do not tile %d segments of %lld",
               ngw, (long long)gw_cols, nseg, (long long)Fi);
    for (int j = 0; j < ngw; ++j) {
        DC_REQUIRE(gws[j], "dc_tag_linear_bwd_dw: null output block %d", j);
        r.gw[j] = gws[j];
    }
    dw_plan(N, Fi, Fo, nseg, &p.chunk_rows, &p.nchunks);
    p.N = N, p.Fi = Fi, p.Fo = Fo, p.nseg = nseg;
    p.h2 = h2;
    p.partial = (float *)partials;
    // The split / fp16x2 / lean kernels walk the nodes in whole 16-row stages.  An N that is not a multiple of 16
    // (the reference's shipped batch of 4 rigid spheres: 3,048 rows) used to send the whole block to the generic
    // kernel; now the first N - N % 16 rows take the fast kernels and the < 16 trailing rows go through the generic
    // one into a partial slot of their own (slot nchunks), summed with the others by the slab reduce.
    const int64_t tail = vec ? N % BK : 0;
    const bool ragged = tail != 0 && N > tail;
    const int nslots = p.nchunks + (ragged ? 1 : 0);
    p.bias_partial = gbias ? p.partial + (int64_t)nslots * nseg * Fo * Fi : nullptr;
    const int mb = dw_mb(Fo);
    const int64_t tiles = ((Fo + 64 * mb - 1) / (64 * mb)) * ((Fi + BN - 1) / BN);
    const int64_t grid = tiles * nseg * p.nchunks;
    if (!p.has_mask) p.mask = p.g;
    const dim3 gd((unsigned)grid), bd(256);
    hipStream_t hs = (hipStream_t)stream;
    if (g2) {
        // corrected gradient operand: only the 128 x 256 tile kernel forms it
        DC_REQUIRE(g2_coef && products == 2 && vec && !ragged && (((uintptr_t)g2) & 15) == 0,
                   "dc_tag_linear_bwd_dw_h2_corr: needs the coefficient vector, aligned operands and N %% 16 == 0");
        p.g2 = g2, p.g2_coef = g2_coef;
        DC_REQUIRE(dw_h2w_launch(p, hs), "dc_tag_linear_bwd_dw_h2_corr: shape not eligible for the 128 x 256 tile kernel "
                   "(needs Fo %% 128 == 0, Fi == 256, N %% 32 == 0; N=%lld Fi=%lld Fo=%lld)", (long long)N,
                   (long long)Fi, (long long)Fo);
        r.partial = p.partial, r.bias_partial = p.bias_partial, r.gbias = gbias;
        r.Fi = Fi, r.Fo = Fo, r.nseg = nseg, r.nchunks = nslots;
        r.cols = gw_cols, r.ngw = ngw, r.bps = ngw / nseg, r.accumulate = accumulate;
        const int64_t total_c = (int64_t)ngw * Fo * gw_cols + (gbias ? Fo : 0);
        launch_reduce(r, total_c, hs);
        return check_launch("dc_tag_linear_bwd_dw_h2_corr");
    }
    if (ragged) {
        DwParams t = p;                                  // the trailing rows: one chunk of the generic kernel
        const int64_t n0 = N - tail;
        t.g.p += n0 * t.g.ld;
        t.mask.p += n0 * t.mask.ld;
        for (int s = 0; s < nseg; ++s) t.x[s].p += n0 * t.x[s].ld;
        t.N = tail, t.chunk_rows = BK, t.nchunks = 1;
        t.partial = p.partial + (int64_t)p.nchunks * nseg * Fo * Fi;
        t.bias_partial = p.bias_partial ? p.bias_partial + (int64_t)p.nchunks * Fo : nullptr;
        t.h2 = H2Scales{};
        const dim3 gt((unsigned)(tiles * nseg));
        if (mb == 2) {
            if (p.has_mask) DC_LAUNCH((k_tag_linear_bwd_dw<2, true, true>), gt, bd, 0, hs, t);
            else DC_LAUNCH((k_tag_linear_bwd_dw<2, true, false>), gt, bd, 0, hs, t);
        } else {
            if (p.has_mask) DC_LAUNCH((k_tag_linear_bwd_dw<1, true, true>), gt, bd, 0, hs, t);
            else DC_LAUNCH((k_tag_linear_bwd_dw<1, true, false>), gt, bd, 0, hs, t);
        }
        p.N = n0;                                        // chunks past n0 become empty (zero partials)
    }
    bool fast_done = products && vec && dw_split_launch(p, mb, products, hs);
    DC_REQUIRE(fast_done || products != 2, "dc_tag_linear_bwd_dw_h2: needs N %% 16 == 0, Fi %% 4 == 0, "
               "Fo %% 4 == 0 and 16-byte aligned operands (N=%lld)", (long long)N);
    fast_done = fast_done || (vec && dw_fast_launch(p, mb, hs));
#define DC_DW(MB_, V_, M_) DC_LAUNCH((k_tag_linear_bwd_dw<MB_, V_, M_>), gd, bd, 0, hs, p)
    if (!fast_done) ++g_generic_dense_launches;
    if (fast_done) {
    } else
    if (mb == 2) {
        if (vec && p.has_mask) DC_DW(2, true, true);
        else if (vec) DC_DW(2, true, false);
        else if (p.has_mask) DC_DW(2, false, true);
        else DC_DW(2, false, false);
    } else {
        if (vec && p.has_mask) DC_DW(1, true, true);
        else if (vec) DC_DW(1, true, false);
        else if (p.has_mask) DC_DW(1, false, true);
        else DC_DW(1, false, false);
    }
#undef DC_DW
    r.partial = p.partial, r.bias_partial = p.bias_partial, r.gbias = gbias;
    r.Fi = Fi, r.Fo = Fo, r.nseg = nseg, r.nchunks = nslots;
    r.cols = gw_cols, r.ngw = ngw, r.bps = ngw / nseg, r.accumulate = accumulate;
    const int64_t total = (int64_t)ngw * Fo * gw_cols + (gbias ? Fo : 0);
    launch_reduce(r, total, (hipStream_t)stream);
    return check_launch("dc_tag_linear_bwd_dw");
}

extern "C" int dc_tag_linear_bwd_dw(const float *g, int64_t ldg, const float *out_for_mask,
                                    int64_t ldo, const float *const *xs, const int64_t *ldxs,
                                    int nseg, float *const *gws, int ngw, int64_t gw_cols,
                                    float *gbias, int accumulate, void *partials,
                                    int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                    dc_stream_t stream) {
    return dw_impl(g, ldg, out_for_mask, ldo, xs, ldxs, nseg, gws, ngw, gw_cols, gbias, accumulate,
                   partials, partials_bytes, N, Fi, Fo, stream, 0);
}

extern "C" int dc_tag_linear_bwd_dw_split(const float *g, int64_t ldg, const float *out_for_mask,
                                          int64_t ldo, const float *const *xs, const int64_t *ldxs,
                                          int nseg, float *const *gws, int ngw, int64_t gw_cols,
                                          float *gbias, int accumulate, void *partials,
                                          int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                          int products, dc_stream_t stream) {
    DC_REQUIRE(products == 6 || products == 3 || products == 1,
               "dc_tag_linear_bwd_dw_split: products must be 6, 3 or 1");
    return dw_impl(g, ldg, out_for_mask, ldo, xs, ldxs, nseg, gws, ngw, gw_cols, gbias, accumulate,
                   partials, partials_bytes, N, Fi, Fo, stream, products);
}


// ---- fp16x2 ("h2") entry points: two scaled fp16 planes, 3 MFMA products -------------------
extern "C" int dc_tag_linear_fwd_h2(const float *const *xs, const int64_t *ldxs,
                                    const float *const *ws, int nseg, const float *bias, int relu,
                                    float *out, int64_t ldo, int64_t N, int64_t Fi, int64_t Fo,
                                    const float *x_rowmax, const float *w_rowmax,
                                    dc_stream_t stream) {
    DC_REQUIRE(x_rowmax && w_rowmax, "dc_tag_linear_fwd_h2: x_rowmax / w_rowmax missing");
    H2Scales h{};
    h.a_rowmax = x_rowmax, h.b_rowmax = w_rowmax;
    return fwd_impl(xs, ldxs, ws, nseg, bias, relu, out, ldo, N, Fi, Fo, stream, 2, h);
}

extern "C" int64_t dc_tag_linear_fwd_h2p_workspace_bytes(int64_t N, int64_t K, int64_t Fo) {
    if (N < 0 || K < 1 || Fo < 1) return DC_EINVAL;
    const int64_t tiles = ((N + 63) / 64) * ((Fo + BN - 1) / BN);
    if (tiles >= 256 || K / BK < 16) return 0;                 // enough tiles / too short to cut
    return 32 * N * Fo * (int64_t)sizeof(float);
}

extern "C" int dc_tag_linear_fwd_h2p(const float *x, int64_t ldx, const void *w_image,
                                     const float *bias, int relu, float *out, int64_t ldo, int64_t N,
                                     int64_t K, int64_t Fo, const float *x_rowmax,
                                     const float *w_rowmax, void *workspace,
                                     int64_t workspace_bytes, dc_stream_t stream) {
    DC_REQUIRE(x && w_image && x_rowmax && w_rowmax, "dc_tag_linear_fwd_h2p: null pointer");
    DC_REQUIRE(K >= 16 && K % 16 == 0 && ldx >= K && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 &&
                   (((uintptr_t)w_image) & 15) == 0,
               "dc_tag_linear_fwd_h2p: K %% 16 == 0 and 16-byte aligned operands required (K=%lld)",
               (long long)K);
    H2Scales h{};
    h.a_rowmax = x_rowmax, h.b_rowmax = w_rowmax, h.b_presplit = 1;
    const float *xs[1] = {x};
    const float *ws[1] = {(const float *)w_image};   // 4 bytes per element, fp32 addressing
    const int64_t ld[1] = {ldx};
    DC_REQUIRE(!workspace || (((uintptr_t)workspace) & 15) == 0, "dc_tag_linear_fwd_h2p: workspace misaligned");
    return fwd_impl(xs, ld, ws, 1, bias, relu, out, ldo, N, K, Fo, stream, 2, h, (float *)workspace,
                    workspace ? workspace_bytes : 0);
}

extern "C" int dc_tag_linear_fwd_h2p_corr(const float *x, int64_t ldx, const float *x2, const float *x2_coef,
                                          const void *w_image, float *out, int64_t ldo, int64_t N, int64_t K,
                                          int64_t Fo, const float *x_rowmax, const float *w_rowmax,
                                          dc_stream_t stream) {
    DC_REQUIRE(N >= 0 && K >= 32 && K % 32 == 0 && Fo >= 1, "dc_tag_linear_fwd_h2p_corr: K must be a multiple of 32");
    if (N == 0) return DC_OK;
    DC_REQUIRE(x && x2 && x2_coef && w_image && out && x_rowmax && w_rowmax && ldx >= K && ldo >= Fo &&
                   (((uintptr_t)x2) & 15) == 0, "dc_tag_linear_fwd_h2p_corr: null / short / misaligned operand");
    FwdParams p{};
    bool vec = true;
    p.x[0] = make_mat(x, ldx, &vec);
    p.w[0] = make_mat((const float *)w_image, K, &vec);
    p.out = out, p.ldo = ldo, p.N = N, p.Fi = K, p.Fo = Fo, p.nseg = 1;
    p.h2.a_rowmax = x_rowmax, p.h2.b_rowmax = w_rowmax, p.h2.b_presplit = 1;
    p.x2 = x2, p.x2_coef = x2_coef;
    DC_REQUIRE(vec && fwd_h2w_launch(p, (hipStream_t)stream),
               "dc_tag_linear_fwd_h2p_corr: shape not eligible for the 128 x 256 tile kernel (N=%lld K=%lld Fo=%lld: "
               "needs 16-byte aligned rows)", (long long)N, (long long)K, (long long)Fo);
    return check_launch("dc_tag_linear_fwd_h2p_corr");
}

extern "C" int dc_tag_linear_fwd_h2p_exp(const float *x, int64_t ldx, const void *w_image, float *out,
                                         int64_t ldo, int64_t N, int64_t K, int64_t Fo, const float *x_rowmax,
                                         const float *w_rowmax, const float *row_lse, int64_t ncols_valid,
                                         dc_stream_t stream) {
    DC_REQUIRE(x && w_image && x_rowmax && w_rowmax && row_lse, "dc_tag_linear_fwd_h2p_exp: null pointer");
    DC_REQUIRE(K >= 16 && K % 16 == 0 && ldx >= K && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 &&
                   (((uintptr_t)w_image) & 15) == 0 && ncols_valid >= 0 && ncols_valid <= Fo,
               "dc_tag_linear_fwd_h2p_exp: K %% 16 == 0, 16-byte aligned operands, 0 <= ncols_valid <= Fo required");
    H2Scales h{};
    h.a_rowmax = x_rowmax, h.b_rowmax = w_rowmax, h.b_presplit = 1;
    const float *xs[1] = {x};
    const float *ws[1] = {(const float *)w_image};
    const int64_t ld[1] = {ldx};
    return fwd_impl(xs, ld, ws, 1, nullptr, 0, out, ldo, N, K, Fo, stream, 2, h, nullptr, 0, row_lse, ncols_valid);
}

extern "C" int dc_tag_linear_bwd_dx_h2(const float *g, int64_t ldg, const float *out_for_mask,
                                       int64_t ldo, const float *const *ws, int nseg,
                                       float *const *gxs, const int64_t *ldgxs, void *workspace,
                                       int64_t workspace_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                       const float *g_rowmax, const float *w_rowmax,
                                       dc_stream_t stream) {
    DC_REQUIRE(w_rowmax && g_rowmax, "dc_tag_linear_bwd_dx_h2: g_rowmax / w_rowmax missing");
    DC_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 &&
                   workspace_bytes >= dc_tag_linear_bwd_dx_split_workspace_bytes(Fi, Fo, nseg),
               "dc_tag_linear_bwd_dx_h2: workspace missing, misaligned or too small");
    H2Scales h{};
    h.a_rowmax = g_rowmax, h.b_rowmax = w_rowmax;
    return dx_impl(g, ldg, out_for_mask, ldo, ws, nseg, gxs, ldgxs, N, Fi, Fo, stream,
                   (float *)workspace, 2, h);
}

extern "C" int dc_tag_linear_bwd_dw_h2(const float *g, int64_t ldg, const float *out_for_mask,
                                       int64_t ldo, const float *const *xs, const int64_t *ldxs,
                                       int nseg, float *const *gws, int ngw, int64_t gw_cols,
                                       float *gbias, int accumulate, void *partials,
                                       int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                       const float *g_rowmax, const float *x_rowmax,
                                       dc_stream_t stream) {
    DC_REQUIRE(g_rowmax && x_rowmax, "dc_tag_linear_bwd_dw_h2: g_rowmax / x_rowmax missing");
    H2Scales h{};
    h.a_rowmax = g_rowmax, h.b_rowmax = x_rowmax;
    return dw_impl(g, ldg, out_for_mask, ldo, xs, ldxs, nseg, gws, ngw, gw_cols, gbias, accumulate,
                   partials, partials_bytes, N, Fi, Fo, stream, 2, h);
}

extern "C" int dc_tag_linear_bwd_dw_h2_corr(const float *g, int64_t ldg, const float *g2, const float *g2_coef,
                                            const float *const *xs, const int64_t *ldxs, int nseg,
                                            float *const *gws, int ngw, int64_t gw_cols, int accumulate,
                                            void *partials, int64_t partials_bytes, int64_t N, int64_t Fi,
                                            int64_t Fo, const float *g_rowmax, const float *x_rowmax,
                                            dc_stream_t stream) {
    DC_REQUIRE(g2 && g2_coef && g_rowmax && x_rowmax, "dc_tag_linear_bwd_dw_h2_corr: null g2 / coefficients / row maxima");
    H2Scales h2{};
    h2.a_rowmax = g_rowmax, h2.b_rowmax = x_rowmax;
    return dw_impl(g, ldg, nullptr, 0, xs, ldxs, nseg, gws, ngw, gw_cols, nullptr, accumulate, partials,
                   partials_bytes, N, Fi, Fo, stream, 2, h2, g2, g2_coef);
}

namespace dc {
// rowmax[i] = max |x[i, 0:F]| : one wave per row
__global__ void __launch_bounds__(256)
k_rowabsmax(const float *__restrict__ x, int64_t ld, int64_t N, int F, float *__restrict__ rowmax,
            bool vec4) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const float *xr = x + row * ld;
    float m = 0.f;
    if (vec4) {
        for (int c = lane * 4; c < F; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    } else {
        for (int c = lane; c < F; c += 64) m = fmaxf(m, fabsf(xr[c]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) rowmax[row] = m;
}

// out[o] = max over segments s and columns f of |W_s[o, f]| : one wave per output row
struct WRowmaxParams {
    const float *w[kMaxSeg];
    int nseg;
    int64_t Fo, Fi;
    float *out;
};
__global__ void __launch_bounds__(256) k_w_rowmax(WRowmaxParams p) {
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= p.Fo) return;
    const int lane = threadIdx.x & 63;
    float m = 0.f;
    for (int s = 0; s < p.nseg; ++s) {
        const float *wr = p.w[s] + o * p.Fi;
        for (int64_t c = lane; c < p.Fi; c += 64) m = fmaxf(m, fabsf(wr[c]));
    }
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q));
    if (lane == 0) p.out[o] = m;
}
}  // namespace dc

namespace dc {
// gm[i,:] = g[i,:] * (out[i,:] > 0), rowmax[i] = max |gm[i,:]| : one wave per row
__global__ void __launch_bounds__(256)
k_mask_grad(const float *__restrict__ g, int64_t ldg, const float *__restrict__ mask, int64_t ldm,
            float *__restrict__ gm, int64_t ldgm, int64_t N, int F, float *__restrict__ rowmax_a,
            float *__restrict__ rowmax_b, bool vec4) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const float *gr = g + row * ldg, *mr = mask ? mask + row * ldm : nullptr;
    float *o = gm + row * ldgm;
    float m = 0.f;
    if (vec4) {
        for (int c = lane * 4; c < F; c += 256) {
            float4 v = *reinterpret_cast<const float4 *>(gr + c);
            if (mr) {
                const float4 k = *reinterpret_cast<const float4 *>(mr + c);
                v = make_float4(k.x > 0.f ? v.x : 0.f, k.y > 0.f ? v.y : 0.f, k.z > 0.f ? v.z : 0.f,
                                k.w > 0.f ? v.w : 0.f);
            }
            *reinterpret_cast<float4 *>(o + c) = v;
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    } else {
        for (int c = lane; c < F; c += 64) {
            float v = gr[c];
            if (mr && !(mr[c] > 0.f)) v = 0.f;
            o[c] = v;
            m = fmaxf(m, fabsf(v));
        }
    }
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q));
    if (lane == 0) {
        rowmax_a[row] = m;
        if (rowmax_b) rowmax_b[row] = m;
    }
}
}  // namespace dc

extern "C" int dc_tag_mask_grad(const float *g, int64_t ldg, const float *out_for_mask, int64_t ldo,
                                float *gm, int64_t ldgm, int64_t N, int64_t F, float *rowmax_a,
                                float *rowmax_b, dc_stream_t stream) {
    DC_REQUIRE(N >= 0 && F >= 1 && F < (1 << 24) && ldg >= F && ldgm >= F && (!out_for_mask || ldo >= F),
               "dc_tag_mask_grad: bad sizes");
    if (N == 0) return DC_OK;
    DC_REQUIRE(g && gm && rowmax_a, "dc_tag_mask_grad: null pointer");
    const bool vec4 = (F % 4 == 0) && (ldg % 4 == 0) && (ldgm % 4 == 0) && (((uintptr_t)g) & 15) == 0 &&
                      (((uintptr_t)gm) & 15) == 0 &&
                      (!out_for_mask || ((ldo % 4 == 0) && (((uintptr_t)out_for_mask) & 15) == 0));
    DC_LAUNCH(k_mask_grad, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g,
                       ldg, out_for_mask, ldo, gm, ldgm, N, (int)F, rowmax_a, rowmax_b, vec4);
    return check_launch("dc_tag_mask_grad");
}

namespace dc {
// Everything the h2 dense blocks of one TAGConv layer need from its weights, in ONE launch:
//   blocks [0, ceil(Fo/4))          : w_rowmax[o] = max_s,f |W_s[o,f]| and (optional) the scaled
//                                     fp16x2 image of row o over the concatenated reduction
//                                     k = s*Fi + f  (wave per row o, two passes over its 4 KiB)
//   blocks [ceil(Fo/4), +ceil(Fi/4)): the same for the transposed weights: wt_rowmax[f] =
//                                     max_s,o |W_s[o,f]| and the image of row f over k = s*Fo + o
//                                     (wave per column f: strided reads of the L2-resident weights)
// Image of a row: per 16-wide stage one 64-byte record {h1[16], h2[16]} (fp16), x * 2^e = h1 + h2
// with the row's power-of-two scale (dc_dense.h) - the bytes k_fwd_h2 wants in LDS, so that kernel
// stages the weights by LDS-DMA without touching registers or the VALU.
struct WPrepParams {
    const float *w[kMaxSeg];
    int nseg;
    int64_t Fo, Fi;
    float *w_rowmax, *wt_rowmax;
    _Float16 *wimg, *wtimg;
    float *zero;                         // optional: a float buffer this launch also clears (dc_tag_weight_prep_zero)
    int64_t zero_n;
    int tall;                            // the transposed half is done by k_wt_colmax + k_wt_image (tall matrices): this
                                         // launch only clears wt_rowmax for their atomic maxima
};
__device__ __forceinline__ void wprep_put(_Float16 *img_row, int64_t k, float v, float scale) {
    const float x = v * scale;
    const _Float16 h = (_Float16)x;
    _Float16 *rec = img_row + (k >> 4) * 32 + (k & 15);
    rec[0] = h;
    rec[16] = (_Float16)(x - (float)h);
}
__device__ __forceinline__ void weight_prep_body(const WPrepParams &p) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.zero_n; i += (int64_t)gridDim.x * blockDim.x)
        p.zero[i] = 0.f;
    if (p.tall)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.Fi; i += (int64_t)gridDim.x * blockDim.x)
            p.wt_rowmax[i] = 0.f;
    const int64_t rb = (p.Fo + 3) / 4;
    float m = 0.f;
    // up to 256 x 256 per segment (the encoder's layers) a wave keeps its row / column in registers between the maximum and the
    // image pass - one trip to L2 per wave instead of two dependent ones (round 6: the launch sits on each branch's critical path)
    const bool small = p.Fi <= 256 && p.Fo <= 256;
    if ((int64_t)blockIdx.x < rb) {
        const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (o >= p.Fo) return;
        float v[kMaxSeg][4];
        if (small) {
#pragma unroll
            for (int s = 0; s < kMaxSeg; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t c = lane + 64 * j;
                    v[s][j] = (s < p.nseg && c < p.Fi) ? p.w[s][o * p.Fi + c] : 0.f;
                    m = fmaxf(m, fabsf(v[s][j]));
                }
        } else {
            for (int s = 0; s < p.nseg; ++s) {
                const float *wr = p.w[s] + o * p.Fi;
                for (int64_t c = lane; c < p.Fi; c += 64) m = fmaxf(m, fabsf(wr[c]));
            }
        }
#pragma unroll
        for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q));
        if (lane == 0) p.w_rowmax[o] = m;
        if (p.wimg) {
            const float sc = h2_scale(m);
            _Float16 *row = p.wimg + o * (2 * p.nseg * p.Fi);
            if (small) {
#pragma unroll
                for (int s = 0; s < kMaxSeg; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t c = lane + 64 * j;
                        if (s < p.nseg && c < p.Fi) wprep_put(row, s * p.Fi + c, v[s][j], sc);
                    }
            } else {
                for (int s = 0; s < p.nseg; ++s) {
                    const float *wr = p.w[s] + o * p.Fi;
                    for (int64_t c = lane; c < p.Fi; c += 64) wprep_put(row, s * p.Fi + c, wr[c], sc);
                }
            }
        }
    } else {
        const int64_t f = ((int64_t)blockIdx.x - rb) * 4 + (threadIdx.x >> 6);
        if (f >= p.Fi) return;
        float v[kMaxSeg][4];
        if (small) {
#pragma unroll
            for (int s = 0; s < kMaxSeg; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t o = lane + 64 * j;
                    v[s][j] = (s < p.nseg && o < p.Fo) ? p.w[s][o * p.Fi + f] : 0.f;
                    m = fmaxf(m, fabsf(v[s][j]));
                }
        } else {
            for (int s = 0; s < p.nseg; ++s) {
                const float *wc = p.w[s] + f;
                for (int64_t o = lane; o < p.Fo; o += 64) m = fmaxf(m, fabsf(wc[o * p.Fi]));
            }
        }
#pragma unroll
        for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q));
        if (lane == 0) p.wt_rowmax[f] = m;
        const float sc = h2_scale(m);
        _Float16 *row = p.wtimg + f * (2 * p.nseg * p.Fo);
        if (small) {
#pragma unroll
            for (int s = 0; s < kMaxSeg; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t o = lane + 64 * j;
                    if (s < p.nseg && o < p.Fo) wprep_put(row, s * p.Fo + o, v[s][j], sc);
                }
        } else {
            for (int s = 0; s < p.nseg; ++s) {
                const float *wc = p.w[s] + f;
                for (int64_t o = lane; o < p.Fo; o += 64) wprep_put(row, s * p.Fo + o, wc[o * p.Fi], sc);
            }
        }
    }
}
__global__ void __launch_bounds__(256) k_weight_prep(WPrepParams p) { weight_prep_body(p); }

// ---- the transposed half for TALL matrices (the attention's keys / values as "weights": Fo = 24,384 rows) ----------
// weight_prep_body gives a wave one column f and lets its lanes walk the rows: every lane of a load sits in another cache
// line - 0.8 GB of L2 requests and 193 us for a 25 MB matrix (6 launches, 1.2 ms of a batch-32 step).  Here the column
// maxima come from coalesced row reads (a thread per column, a block per 256-row chunk, joined with integer atomicMax:
// the values are non-negative floats) and the image from 64 x 64 tiles transposed through LDS; same scale, same
// rounding: wt_rowmax and the image are bit-identical to weight_prep_body's.
constexpr int kWtChunk = 256;
__global__ void __launch_bounds__(256)
k_wt_colmax(WPrepParams p) {
    const int64_t f = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int64_t o0 = (int64_t)blockIdx.x * kWtChunk, o1 = o0 + kWtChunk < p.Fo ? o0 + kWtChunk : p.Fo;
    if (f >= p.Fi) return;
    float m = 0.f;
    for (int s = 0; s < p.nseg; ++s) {
        const float *wc = p.w[s] + f;
        for (int64_t o = o0; o < o1; ++o) m = fmaxf(m, fabsf(wc[o * p.Fi]));
    }
    atomicMax(reinterpret_cast<int *>(p.wt_rowmax + f), __float_as_int(m));
}

// tile: 64 rows o x 64 columns f of segment s (blockIdx.z); needs Fo % 16 == 0 (records do not straddle segments)
__global__ void __launch_bounds__(256)
k_wt_image(WPrepParams p) {
    __shared__ _Float16 sh[64][64 + 8], sl[64][64 + 8];                  // [f][o]: the two planes of the tile, transposed
    __shared__ float ssc[64];
    const int s = blockIdx.z;
    const int64_t o0 = (int64_t)blockIdx.x * 64, f0 = (int64_t)blockIdx.y * 64;
    if (threadIdx.x < 64) {
        const int64_t f = f0 + threadIdx.x;
        ssc[threadIdx.x] = h2_scale(f < p.Fi ? p.wt_rowmax[f] : 0.f);
    }
    __syncthreads();
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;               // column of the tile, row phase
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int r = r4 + 4 * j;
        const int64_t o = o0 + r, f = f0 + c;
        float v = 0.f;
        if (o < p.Fo && f < p.Fi) v = p.w[s][o * p.Fi + f];             // a wave reads 256 contiguous bytes of a row
        const float x = v * ssc[c];
        const _Float16 h = (_Float16)x;
        sh[c][r] = h;
        sl[c][r] = (_Float16)(x - (float)h);
    }
    __syncthreads();
    // one 64-byte record {h1[16], h2[16]} per thread: column f0 + t / 4, rows o0 + 16 (t % 4) .. + 15
    const int fc = threadIdx.x >> 2, rec = threadIdx.x & 3;
    const int64_t f = f0 + fc, ob = o0 + 16 * rec;
    if (f < p.Fi && ob < p.Fo) {
        _Float16 *dst = p.wtimg + f * (2 * p.nseg * p.Fo) + (((int64_t)s * p.Fo + ob) >> 4) * 32;
        using h8 = __attribute__((ext_vector_type(8))) _Float16;
        const h8 *ph = reinterpret_cast<const h8 *>(&sh[fc][16 * rec]), *pl = reinterpret_cast<const h8 *>(&sl[fc][16 * rec]);
        h8 *d = reinterpret_cast<h8 *>(dst);
        d[0] = ph[0], d[1] = ph[1], d[2] = pl[0], d[3] = pl[1];
    }
}
}  // namespace dc

extern "C" int dc_tag_weight_prep_zero(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *w_rowmax,
                                       void *w_image, void *wt_image, float *wt_rowmax, float *zero, int64_t zero_n,
                                       dc_stream_t stream);
extern "C" int dc_tag_weight_prep(const float *const *ws, int nseg, int64_t Fo, int64_t Fi,
                                  float *w_rowmax, void *w_image, void *wt_image, float *wt_rowmax,
                                  dc_stream_t stream) {
    return dc_tag_weight_prep_zero(ws, nseg, Fo, Fi, w_rowmax, w_image, wt_image, wt_rowmax, nullptr, 0, stream);
}

extern "C" int dc_tag_weight_prep_zero(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *w_rowmax,
                                       void *w_image, void *wt_image, float *wt_rowmax, float *zero, int64_t zero_n,
                                       dc_stream_t stream) {
    DC_REQUIRE(zero_n >= 0 && (zero_n == 0 || zero), "dc_tag_weight_prep_zero: bad zero buffer");
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg && Fo >= 1 && Fi >= 1 && ws && w_rowmax,
               "dc_tag_weight_prep: bad arguments");
    DC_REQUIRE((wt_image == nullptr) == (wt_rowmax == nullptr),
               "dc_tag_weight_prep: wt_image and wt_rowmax go together");
    DC_REQUIRE((!w_image || (nseg * Fi) % 16 == 0) && (!wt_image || (nseg * Fo) % 16 == 0),
               "dc_tag_weight_prep: images need a reduction extent that is a multiple of 16");
    DC_REQUIRE((((uintptr_t)w_image) & 15) == 0 && (((uintptr_t)wt_image) & 15) == 0,
               "dc_tag_weight_prep: images must be 16-byte aligned");
    WPrepParams p{};
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(ws[s], "dc_tag_weight_prep: null segment %d", s);
        p.w[s] = ws[s];
    }
    p.nseg = nseg, p.Fo = Fo, p.Fi = Fi, p.w_rowmax = w_rowmax, p.wt_rowmax = wt_rowmax;
    p.wimg = (_Float16 *)w_image, p.wtimg = (_Float16 *)wt_image;
    p.zero = zero, p.zero_n = zero_n;
    constexpr int tall_min = 2048;
    const bool tall = wt_image && Fo >= tall_min && Fo % 16 == 0;
    p.tall = tall ? 1 : 0;
    const int64_t blocks = (Fo + 3) / 4 + ((wt_image && !tall) ? (Fi + 3) / 4 : 0);
    DC_LAUNCH(k_weight_prep, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    if (tall) {
        DC_LAUNCH(k_wt_colmax, dim3((unsigned)((Fo + kWtChunk - 1) / kWtChunk), (unsigned)((Fi + 255) / 256)), dim3(256), 0,
                  (hipStream_t)stream, p);
        DC_LAUNCH(k_wt_image, dim3((unsigned)((Fo + 63) / 64), (unsigned)((Fi + 63) / 64), (unsigned)nseg), dim3(256), 0,
                  (hipStream_t)stream, p);
    }
    return check_launch("dc_tag_weight_prep");
}

extern "C" int dc_rowabsmax_f32(const float *x, int64_t ld, int64_t N, int64_t F, float *rowmax,
                                dc_stream_t stream) {
    DC_REQUIRE(N >= 0 && F >= 1 && F < (1 << 24) && ld >= F, "dc_rowabsmax_f32: bad sizes");
    if (N == 0) return DC_OK;
    DC_REQUIRE(x && rowmax, "dc_rowabsmax_f32: null pointer");
    const bool vec4 = (F % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)x) & 15) == 0;
    DC_LAUNCH(k_rowabsmax, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, ld, N, (int)F, rowmax, vec4);
    return check_launch("dc_rowabsmax_f32");
}

extern "C" int dc_tag_transpose_weights(const float *const *ws, int nseg, int64_t Fo, int64_t Fi,
                                        float *wt, dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg && Fo >= 1 && Fi >= 1 && ws && wt,
               "dc_tag_transpose_weights: bad arguments");
    for (int s = 0; s < nseg; ++s) DC_REQUIRE(ws[s], "dc_tag_transpose_weights: null segment %d", s);
    transpose_weights_launch(ws, nseg, Fo, Fi, wt, (hipStream_t)stream);
    return check_launch("dc_tag_transpose_weights");
}

extern "C" int dc_tag_weight_rowmax(const float *const *ws, int nseg, int64_t Fo, int64_t Fi,
                                    float *w_rowmax, dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg && Fo >= 1 && Fi >= 1 && ws && w_rowmax,
               "dc_tag_weight_rowmax: bad arguments");
    WRowmaxParams p{};
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(ws[s], "dc_tag_weight_rowmax: null segment %d", s);
        p.w[s] = ws[s];
    }
    p.nseg = nseg, p.Fo = Fo, p.Fi = Fi, p.out = w_rowmax;
    DC_LAUNCH(k_w_rowmax, dim3((unsigned)((Fo + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dc_tag_weight_rowmax");
}


// ======================================================================================================
// Grouped launches: the layer-2 blocks of BOTH encoder branches (models/model.py:69-78) as one launch over
// a merged, block-diagonal node space.  Group g = rows [row_beg[g], row_beg[g] + rows[g]) with its own
// weights; row_beg[g] is a multiple of DC_GROUP_ALIGN and the rows between the groups (and behind the last
// one, up to N_total - itself a multiple of DC_GROUP_ALIGN) are ZERO rows the host pads with (isolated nodes
// of the merged adjacency: the hops keep them zero).  Per row the arithmetic is that of the ungrouped
// entry points on that group alone - outputs and gradients are bit-identical to separate launches.
// ======================================================================================================
namespace dc {
static int check_groups(const char *what, int ngroups, const int64_t *row_beg, const int64_t *rows,
                        int64_t N_total) {
    DC_REQUIRE(ngroups >= 1 && ngroups <= kMaxGroups && row_beg && rows, "%s: 1..%d groups required", what,
               kMaxGroups);
    DC_REQUIRE(N_total >= 0 && N_total % kGroupAlign == 0, "%s: N_total must be a multiple of %d", what,
               kGroupAlign);
    for (int g = 0; g < ngroups; ++g) {
        DC_REQUIRE(row_beg[g] >= 0 && row_beg[g] % kGroupAlign == 0 && rows[g] >= 0,
                   "%s: group %d must start at a multiple of %d rows", what, g, kGroupAlign);
        const int64_t next = g + 1 < ngroups ? row_beg[g + 1] : N_total;
        DC_REQUIRE(row_beg[g] + rows[g] <= next, "%s: group %d overlaps the next one", what, g);
    }
    return DC_OK;
}

struct MaskGradGroups {
    const float *g[kMaxGroups];
    int64_t ldg[kMaxGroups], row_beg[kMaxGroups], rows[kMaxGroups];
    int n;
};
// gm[i,:] = g_group(i)[i - row_beg,:] * (out[i,:] > 0) for the rows of a group, 0 for padding rows;
// rowmax[i] = max |gm[i,:]| : one wave per merged row (k_mask_grad over a merged node space)
__global__ void __launch_bounds__(256)
k_mask_grad_grouped(MaskGradGroups q, const float *__restrict__ mask, int64_t ldm, float *__restrict__ gm,
                    int64_t ldgm, int64_t N, int F, float *__restrict__ rowmax_a, float *__restrict__ rowmax_b) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int grp = group_of_row(q.row_beg, q.n, row);
    const int64_t local = row - q.row_beg[grp];
    const bool live = local < q.rows[grp];
    const float *gr = q.g[grp] + (live ? local : 0) * q.ldg[grp], *mr = mask ? mask + row * ldm : nullptr;
    float *o = gm + row * ldgm;
    float m = 0.f;
    for (int c = lane * 4; c < F; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = *reinterpret_cast<const float4 *>(gr + c);
            if (mr) {
                const float4 k = *reinterpret_cast<const float4 *>(mr + c);
                v = make_float4(k.x > 0.f ? v.x : 0.f, k.y > 0.f ? v.y : 0.f, k.z > 0.f ? v.z : 0.f,
                                k.w > 0.f ? v.w : 0.f);
            }
        }
        *reinterpret_cast<float4 *>(o + c) = v;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if (lane == 0) {
        rowmax_a[row] = m;
        if (rowmax_b) rowmax_b[row] = m;
    }
}

struct WPrepGroups {
    WPrepParams g[kMaxGroups];
};
__global__ void __launch_bounds__(256) k_weight_prep_grouped(WPrepGroups q) { weight_prep_body(q.g[blockIdx.y]); }
}  // namespace dc

extern "C" int dc_tag_grouped_weight_prep(const float *const *ws, int ngroups, int nseg, int64_t Fo, int64_t Fi,
                                          float *const *w_rowmax, void *const *w_image, void *const *wt_image,
                                          float *const *wt_rowmax, dc_stream_t stream) {
    DC_REQUIRE(ngroups >= 1 && ngroups <= kMaxGroups && nseg >= 1 && nseg <= kMaxSeg && Fo >= 1 && Fi >= 1 && ws &&
                   w_rowmax && w_image,
               "dc_tag_grouped_weight_prep: bad arguments");
    DC_REQUIRE((nseg * Fi) % 16 == 0 && (!wt_image || (nseg * Fo) % 16 == 0),
               "dc_tag_grouped_weight_prep: images need a reduction extent that is a multiple of 16");
    DC_REQUIRE((wt_image == nullptr) == (wt_rowmax == nullptr),
               "dc_tag_grouped_weight_prep: wt_image and wt_rowmax go together");
    WPrepGroups q{};
    for (int g = 0; g < ngroups; ++g) {
        WPrepParams &p = q.g[g];
        for (int s = 0; s < nseg; ++s) {
            DC_REQUIRE(ws[g * nseg + s], "dc_tag_grouped_weight_prep: null segment %d of group %d", s, g);
            p.w[s] = ws[g * nseg + s];
        }
        DC_REQUIRE(w_rowmax[g] && w_image[g] && (((uintptr_t)w_image[g]) & 15) == 0 &&
                       (!wt_image || (wt_image[g] && wt_rowmax[g] && (((uintptr_t)wt_image[g]) & 15) == 0)),
                   "dc_tag_grouped_weight_prep: null / misaligned output of group %d", g);
        p.nseg = nseg, p.Fo = Fo, p.Fi = Fi, p.w_rowmax = w_rowmax[g];
        p.wt_rowmax = wt_rowmax ? wt_rowmax[g] : nullptr;
        p.wimg = (_Float16 *)w_image[g], p.wtimg = wt_image ? (_Float16 *)wt_image[g] : nullptr;
    }
    const int64_t blocks = (Fo + 3) / 4 + (wt_image ? (Fi + 3) / 4 : 0);
    DC_LAUNCH(k_weight_prep_grouped, dim3((unsigned)blocks, (unsigned)ngroups), dim3(256), 0,
                       (hipStream_t)stream, q);
    return check_launch("dc_tag_grouped_weight_prep");
}

extern "C" int dc_tag_grouped_fwd_h2p(const float *x, int64_t ldx, int ngroups, const int64_t *row_beg,
                                      const int64_t *rows, int64_t N_total, const void *const *w_images,
                                      const float *const *biases, int relu, float *out, int64_t ldo, int64_t K,
                                      int64_t Fo, const float *x_rowmax, const float *const *w_rowmaxes,
                                      dc_stream_t stream) {
    if (int rc = check_groups("dc_tag_grouped_fwd_h2p", ngroups, row_beg, rows, N_total)) return rc;
    if (N_total == 0) return DC_OK;
    DC_REQUIRE(x && w_images && out && x_rowmax && w_rowmaxes && ldo >= Fo, "dc_tag_grouped_fwd_h2p: null pointer");
    DC_REQUIRE(K >= 32 && K % 32 == 0 && ldx >= K && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0,
               "dc_tag_grouped_fwd_h2p: K %% 32 == 0 and a 16-byte aligned x required (K=%lld)", (long long)K);
    FwdParams p{};
    p.x[0] = Mat{x, ldx};
    p.out = out, p.ldo = ldo, p.N = N_total, p.Fi = K, p.Fo = Fo, p.nseg = 1, p.relu = relu;
    p.h2.a_rowmax = x_rowmax, p.h2.b_presplit = 1;
    p.grp.n = ngroups;
    for (int g = 0; g < ngroups; ++g) {
        DC_REQUIRE(w_images[g] && w_rowmaxes[g] && (((uintptr_t)w_images[g]) & 15) == 0,
                   "dc_tag_grouped_fwd_h2p: null / misaligned weights of group %d", g);
        p.grp.row_beg[g] = row_beg[g], p.grp.row_end[g] = row_beg[g] + rows[g];
        p.grp.w[g] = (const float *)w_images[g];
        p.grp.bias[g] = biases ? biases[g] : nullptr, p.grp.b_rowmax[g] = w_rowmaxes[g];
    }
    // a single group is the ungrouped launch
    p.w[0] = Mat{p.grp.w[0], K}, p.bias = p.grp.bias[0], p.h2.b_rowmax = p.grp.b_rowmax[0];
    DC_REQUIRE(fwd_h2w_launch(p, (hipStream_t)stream),
               "dc_tag_grouped_fwd_h2p: shape not eligible for the 128 x 256 tile kernel (K=%lld Fo=%lld)",
               (long long)K, (long long)Fo);
    return check_launch("dc_tag_grouped_fwd_h2p");
}

extern "C" int dc_tag_grouped_mask_grad(const float *const *g, const int64_t *ldg, int ngroups,
                                        const int64_t *row_beg, const int64_t *rows, int64_t N_total,
                                        const float *out_for_mask, int64_t ldo, float *gm, int64_t ldgm, int64_t F,
                                        float *rowmax_a, float *rowmax_b, dc_stream_t stream) {
    if (int rc = check_groups("dc_tag_grouped_mask_grad", ngroups, row_beg, rows, N_total)) return rc;
    if (N_total == 0) return DC_OK;
    DC_REQUIRE(g && ldg && gm && rowmax_a && F >= 4 && F % 4 == 0 && F < (1 << 24) && ldgm >= F && ldgm % 4 == 0 &&
                   (((uintptr_t)gm) & 15) == 0,
               "dc_tag_grouped_mask_grad: null pointer, F %% 4 != 0 or misaligned gm");
    DC_REQUIRE(!out_for_mask || (ldo >= F && ldo % 4 == 0 && (((uintptr_t)out_for_mask) & 15) == 0),
               "dc_tag_grouped_mask_grad: misaligned mask");
    MaskGradGroups q{};
    q.n = ngroups;
    for (int k = 0; k < ngroups; ++k) {
        DC_REQUIRE(rows[k] == 0 || (g[k] && ldg[k] >= F && ldg[k] % 4 == 0 && (((uintptr_t)g[k]) & 15) == 0),
                   "dc_tag_grouped_mask_grad: null / misaligned gradient of group %d", k);
        q.g[k] = g[k], q.ldg[k] = ldg[k], q.row_beg[k] = row_beg[k], q.rows[k] = rows[k];
    }
    DC_LAUNCH(k_mask_grad_grouped, dim3((unsigned)((N_total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       q, out_for_mask, ldo, gm, ldgm, N_total, (int)F, rowmax_a, rowmax_b);
    return check_launch("dc_tag_grouped_mask_grad");
}

static void grouped_dw_plan(const int64_t *rows, int ngroups, int64_t Fi, int64_t Fo, int nseg, DwGroups *dg) {
    dg->n = ngroups;
    int c = 0;
    for (int g = 0; g < ngroups; ++g) {
        int64_t cr;
        int nch;
        dw_plan(rows[g], Fi, Fo, nseg, &cr, &nch);      // the plan of a separate launch over this group
        cr = (cr + 31) / 32 * 32;
        const int64_t padded = (rows[g] + 31) / 32 * 32;
        nch = (int)((padded + cr - 1) / cr);
        if (nch < 1) nch = 1;
        dg->chunk_beg[g] = c, dg->chunk_rows[g] = cr;
        c += nch;
    }
    for (int g = ngroups; g <= kMaxGroups; ++g) dg->chunk_beg[g] = c;
}

extern "C" int64_t dc_tag_grouped_bwd_dw_workspace_bytes(const int64_t *rows, int ngroups, int64_t Fi, int64_t Fo,
                                                         int nseg) {
    if (!rows || ngroups < 1 || ngroups > kMaxGroups || Fi < 1 || Fo < 1 || nseg < 1 || nseg > kMaxSeg) return DC_EINVAL;
    DwGroups dg{};
    grouped_dw_plan(rows, ngroups, Fi, Fo, nseg, &dg);
    return (int64_t)sizeof(float) * dg.chunk_beg[ngroups] * (nseg * Fo * Fi + Fo) + 16;
}

extern "C" int dc_tag_grouped_bwd_dw_h2(const float *g, int64_t ldg, const float *const *xs, const int64_t *ldxs,
                                        int nseg, int ngroups, const int64_t *row_beg, const int64_t *rows,
                                        int64_t N_total, float *const *gws, float *const *gbias, int accumulate,
                                        void *partials, int64_t partials_bytes, int64_t Fi, int64_t Fo,
                                        const float *g_rowmax, const float *x_rowmax, dc_stream_t stream) {
    if (int rc = check_groups("dc_tag_grouped_bwd_dw_h2", ngroups, row_beg, rows, N_total)) return rc;
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg && Fi == 256 && Fo >= 128 && Fo % 128 == 0,
               "dc_tag_grouped_bwd_dw_h2: needs Fi == 256 and Fo %% 128 == 0 (Fi=%lld Fo=%lld)", (long long)Fi,
               (long long)Fo);
    DC_REQUIRE(g && xs && ldxs && gws && partials && g_rowmax && x_rowmax && ldg >= Fo && ldg % 4 == 0 &&
                   (((uintptr_t)g) & 15) == 0 && (((uintptr_t)partials) & 15) == 0,
               "dc_tag_grouped_bwd_dw_h2: null / misaligned pointer");
    DC_REQUIRE(partials_bytes >= dc_tag_grouped_bwd_dw_workspace_bytes(rows, ngroups, Fi, Fo, nseg),
               "dc_tag_grouped_bwd_dw_h2: workspace too small");
    if (N_total == 0) return DC_OK;
    DwParams p{};
    p.g = Mat{g, ldg}, p.mask = p.g, p.has_mask = 0;
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(xs[s] && ldxs[s] >= Fi && ldxs[s] % 4 == 0 && (((uintptr_t)xs[s]) & 15) == 0,
                   "dc_tag_grouped_bwd_dw_h2: bad segment %d", s);
        p.x[s] = Mat{xs[s], ldxs[s]};
    }
    grouped_dw_plan(rows, ngroups, Fi, Fo, nseg, &p.grp);
    for (int k = 0; k < ngroups; ++k) {
        p.grp.row_beg[k] = row_beg[k];
        p.grp.row_end[k] = row_beg[k] + (rows[k] + 31) / 32 * 32;     // zero rows behind the group's data
    }
    for (int k = 0; k < ngroups; ++k)
        DC_REQUIRE(p.grp.row_end[k] <= (k + 1 < ngroups ? row_beg[k + 1] : N_total),
                   "dc_tag_grouped_bwd_dw_h2: group %d has no room for its zero padding rows", k);
    p.N = N_total, p.Fi = Fi, p.Fo = Fo, p.nseg = nseg, p.chunk_rows = kGroupAlign;
    p.nchunks = p.grp.chunk_beg[kMaxGroups];
    p.h2.a_rowmax = g_rowmax, p.h2.b_rowmax = x_rowmax;
    p.partial = (float *)partials;
    bool any_bias = false;
    for (int k = 0; k < ngroups; ++k) any_bias = any_bias || (gbias && gbias[k]);
    p.bias_partial = any_bias ? p.partial + (int64_t)p.nchunks * nseg * Fo * Fi : nullptr;
    hipStream_t hs = (hipStream_t)stream;
    DC_REQUIRE(dw_h2w_launch(p, hs), "dc_tag_grouped_bwd_dw_h2: shape not eligible for the 128 x 256 tile kernel");
    ReduceGroups rg{};
    for (int k = 0; k < ngroups; ++k) {
        ReduceParams &r = rg.g[k];
        const int c0 = p.grp.chunk_beg[k];
        r.partial = p.partial + (int64_t)c0 * nseg * Fo * Fi;
        r.bias_partial = p.bias_partial ? p.bias_partial + (int64_t)c0 * Fo : nullptr;
        r.gbias = (gbias && gbias[k]) ? gbias[k] : nullptr;
        for (int s = 0; s < nseg; ++s) {
            DC_REQUIRE(gws[k * nseg + s], "dc_tag_grouped_bwd_dw_h2: null output block %d of group %d", s, k);
            r.gw[s] = gws[k * nseg + s];
        }
        r.Fi = Fi, r.Fo = Fo, r.cols = Fi, r.nseg = nseg, r.nchunks = p.grp.chunk_beg[k + 1] - c0;
        r.ngw = nseg, r.bps = 1, r.accumulate = accumulate;
    }
    const int64_t total = (int64_t)nseg * Fo * Fi + (any_bias ? Fo : 0);
    DC_LAUNCH(k_dw_reduce_grouped, dim3((unsigned)((total + 255) / 256), (unsigned)ngroups), dim3(256), 0,
                       hs, rg);
    return check_launch("dc_tag_grouped_bwd_dw_h2");
}


extern "C" int64_t dc_generic_dense_launches(int reset) {
    return reset ? (int64_t)g_generic_dense_launches.exchange(0) : (int64_t)g_generic_dense_launches.load();
}

// ---- dW of the bf16-storage layer (BASELINE.json configs[4] backward): partial slabs per node chunk + slab reduce ----
static void dw_bf16_plan(int64_t N, int64_t Fi, int64_t Fo, int nseg, int64_t *chunk_rows, int *nchunks) {
    const int64_t tiles = (Fo / 128) * (Fi / 256) * nseg;
    int64_t want = 1024 / (tiles > 0 ? tiles : 1);                  // ~4 workgroups per CU: short tiles, good balance
    if (want < 1) want = 1;
    if (want > 128) want = 128;
    int64_t rows = (N + want - 1) / want;
    if (rows < 256) rows = 256;
    rows = (rows + 31) / 32 * 32;
    *chunk_rows = rows;
    *nchunks = (int)((N + rows - 1) / rows);
    if (*nchunks < 1) *nchunks = 1;
}

extern "C" int64_t dc_tag_linear_bwd_dw_bf16_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo, int nseg) {
    if (N < 0 || Fi < 1 || Fo < 1 || nseg < 1 || nseg > kMaxSeg) return DC_EINVAL;
    int64_t rows;
    int nch;
    dw_bf16_plan(N, Fi, Fo, nseg, &rows, &nch);
    return (int64_t)sizeof(float) * nch * (nseg * Fo * Fi + Fo) + 16;
}

extern "C" int dc_tag_linear_bwd_dw_bf16(const uint16_t *g, int64_t ldg, const uint16_t *x, int64_t ldx, int nseg,
                                         float *const *gws, float *gbias, int accumulate, void *partials,
                                         int64_t partials_bytes, int64_t N, int64_t Fi, int64_t Fo,
                                         dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg && N >= 0 && Fi >= 1 && Fo >= 1, "dc_tag_linear_bwd_dw_bf16: bad sizes");
    DC_REQUIRE(Fo % 128 == 0 && Fi % 256 == 0, "dc_tag_linear_bwd_dw_bf16: needs Fo %% 128 == 0 and Fi %% 256 == 0 "
               "(Fi=%lld Fo=%lld)", (long long)Fi, (long long)Fo);
    DC_REQUIRE(g && x && gws && partials && ldg >= Fo && ldx >= nseg * Fi && (((uintptr_t)partials) & 15) == 0,
               "dc_tag_linear_bwd_dw_bf16: null pointer / leading dimension too small");
    DC_REQUIRE(partials_bytes >= dc_tag_linear_bwd_dw_bf16_workspace_bytes(N, Fi, Fo, nseg),
               "dc_tag_linear_bwd_dw_bf16: workspace too small");
    DwBf16Params p{};
    p.g = g, p.x = x, p.ldg = ldg, p.ldx = ldx, p.N = N, p.Fi = Fi, p.Fo = Fo, p.nseg = nseg;
    dw_bf16_plan(N, Fi, Fo, nseg, &p.chunk_rows, &p.nchunks);
    p.partial = (float *)partials;
    p.bias_partial = gbias ? p.partial + (int64_t)p.nchunks * nseg * Fo * Fi : nullptr;
    hipStream_t hs = (hipStream_t)stream;
    if (N > 0)
        DC_REQUIRE(dw_bf16_launch(p, hs), "dc_tag_linear_bwd_dw_bf16: operands must be 16-byte aligned with "
                                          "leading dimensions that are multiples of 8");
    ReduceParams r{};
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(gws[s], "dc_tag_linear_bwd_dw_bf16: null output block %d", s);
        r.gw[s] = gws[s];
    }
    r.partial = p.partial, r.bias_partial = p.bias_partial, r.gbias = gbias;
    r.Fi = Fi, r.Fo = Fo, r.cols = Fi, r.nseg = nseg, r.nchunks = N > 0 ? p.nchunks : 0;
    r.ngw = nseg, r.bps = 1, r.accumulate = accumulate;
    const int64_t total = (int64_t)nseg * Fo * Fi + (gbias ? Fo : 0);
    launch_reduce(r, total, hs);
    return check_launch("dc_tag_linear_bwd_dw_bf16");
}
