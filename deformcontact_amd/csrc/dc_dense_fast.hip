// dc_dense_fast.hip -- lean hot-loop variants of the TAGConv dense block (gfx950).
//
// Same maths as dc_dense.hip (which stays as the any-shape path), specialised for the
// shapes the encoder actually runs: every operand 16-byte aligned, leading dimensions
// multiples of 4, reduction extents multiples of BK = 16.  Under those preconditions the
// main loop needs no bounds logic at all:
//   * partial edge tiles are handled ONCE, before the loop, by clamping the per-thread row /
//     column pointers into range (duplicated rows/columns only feed accumulator elements
//     that are never stored);
//   * every staged float4 has one precomputed global pointer that advances by a constant
//     per stage, and one precomputed LDS offset;
// so a stage is: 2-3 global_load_dwordx4, 2-3 ds_write_b128, 3-4 ds_read_b128 (or 12-16
// ds_read_b32), 16-32 v_mfma_f32_32x32x2_f32, one barrier and a handful of scalar ops.
// (The generic kernel spends ~7 VALU instructions per MFMA on 64-bit bounds arithmetic,
// which put a single wave at 44 % MFMA duty; see DESIGN.md "dense block".)
#include "dc_dense.h"

namespace dc {

template <int ROWS, bool KC, bool MASK>
struct FastOp {
    static constexpr int NV = ROWS / 64;            // float4 per thread per stage
    static constexpr int PER = ROWS / 4;            // RC: float4 per k row
    static constexpr int KPER = 256 / PER;          // RC: k rows covered per pass
    const float *p[NV];
    const float *pm[MASK ? NV : 1];
    float4 v[NV];
    float4 m[MASK ? NV : 1];
    int off[NV];                                    // LDS offsets (floats)
    int64_t step;                                   // floats per stage

    // KC: tile rows [row0, row0+ROWS) x k [k0, k0+BK) of a [nrows, K] matrix (k contiguous)
    __device__ __forceinline__ void init_kc(const float *base, const float *mbase, int64_t ld,
                                            int64_t row0, int64_t nrows, int64_t k0) {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            int64_t row = row0 + r + 64 * j;
            row = row < nrows ? row : nrows - 1;
            p[j] = base + row * ld + k0 + 4 * k4;
            if (MASK) pm[j] = mbase + row * ld + k0 + 4 * k4;
            off[j] = (r + 64 * j) * LDK + 4 * k4;
        }
        step = BK;
    }
    // RC: tile k [k0, k0+BK) x cols [col0, col0+ROWS) of a [K, ncols] matrix (col contiguous)
    __device__ __forceinline__ void init_rc(const float *base, const float *mbase, int64_t ld,
                                            int64_t k0, int64_t col0, int64_t ncols) {
        const int c4 = threadIdx.x % PER, kr = threadIdx.x / PER;
        int64_t col = col0 + 4 * c4;
        col = col + 4 <= ncols ? col : ncols - 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t k = k0 + kr + KPER * j;
            p[j] = base + k * ld + col;
            if (MASK) pm[j] = mbase + k * ld + col;
            off[j] = (kr + KPER * j) * ROWS + 4 * c4;
        }
        step = BK * ld;
    }
    __device__ __forceinline__ void rebase(int64_t delta) {
#pragma unroll
        for (int j = 0; j < NV; ++j) p[j] += delta;
    }
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = *reinterpret_cast<const float4 *>(p[j]);
            p[j] += step;
            if (MASK) {
                m[j] = *reinterpret_cast<const float4 *>(pm[j]);
                pm[j] += step;
            }
        }
    }
    __device__ __forceinline__ float4 value(int j) const {
        if (!MASK) return v[j];
        return make_float4(m[j].x > 0.f ? v[j].x : 0.f, m[j].y > 0.f ? v[j].y : 0.f,
                           m[j].z > 0.f ? v[j].z : 0.f, m[j].w > 0.f ? v[j].w : 0.f);
    }
    __device__ __forceinline__ void store(float *lds) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) *reinterpret_cast<float4 *>(lds + off[j]) = value(j);
    }
};

// 2-deep LDS ring, register prefetch one stage ahead (a full MFMA phase to land), the
// fragments of both chunks of the current stage read up front.
template <int MB, bool A_KC, bool B_KC, int STAGE, int OFFB, typename OA, typename OB,
          typename Next, typename Hook>
__device__ __forceinline__ void fast_loop(float *lds, int nst, OA &A, OB &B, Next &&next_stage,
                                          Hook &&on_store_a, f32x16 (&acc)[MB][2], int wm, int wn) {
    if (nst <= 0) return;
    A.load();
    B.load();
    next_stage();
    A.store(lds);
    B.store(lds + OFFB);
    on_store_a(A);
    if (nst > 1) {
        A.load();
        B.load();
        next_stage();
    }
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        float *cur = lds + (it & 1) * STAGE, *nxt = lds + ((it + 1) & 1) * STAGE;
        Frag<MB> f0, f1;
        load_frag<MB, A_KC, B_KC>(f0, cur, cur + OFFB, 0, wm, wn);
        load_frag<MB, A_KC, B_KC>(f1, cur, cur + OFFB, 1, wm, wn);
        if (it + 1 < nst) {                // registers hold stage it+1 (loaded one phase ago)
            A.store(nxt);
            B.store(nxt + OFFB);
            on_store_a(A);
        }
        if (it + 2 < nst) {
            A.load();
            B.load();
            next_stage();
        }
        mma_frag<MB>(f0, acc);
        mma_frag<MB>(f1, acc);
        __syncthreads();
    }
}

struct NoOp {
    template <typename T> __device__ __forceinline__ void operator()(const T &) const {}
    __device__ __forceinline__ void operator()() const {}
};

// ------------------------------- forward -------------------------------------------
template <int MB>
__global__ void __launch_bounds__(256)
k_fwd_fast(FwdParams p) {
    using T = Tile<MB>;
    constexpr int BM = T::BM, kStage = T::A_KC + T::B_KC, kOffB = T::A_KC;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
    const unsigned ntn = (unsigned)((p.Fo + BN - 1) / BN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * BM, col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);
    FastOp<BM, true, false> A;
    FastOp<BN, true, false> B;
    A.init_kc(p.x[0].p, nullptr, p.x[0].ld, row0, p.N, 0);
    B.init_kc(p.w[0].p, nullptr, p.Fi, col0, p.Fo, 0);
    const int kst = (int)(p.Fi / BK), nst = kst * p.nseg;
    int kk = 0, seg = 0;                      // stage-within-segment of the NEXT load
    auto next_stage = [&]() {
        if (++kk == kst && seg + 1 < p.nseg) {    // wave-uniform: switch to the next K segment
            kk = 0;
            A.rebase((p.x[seg + 1].p - p.x[seg].p) - p.Fi);
            B.rebase((p.w[seg + 1].p - p.w[seg].p) - p.Fi);
            ++seg;
        }
    };
    fast_loop<MB, true, true, kStage, kOffB>(lds, nst, A, B, next_stage, NoOp{}, acc, wm, wn);

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

// ------------------------------- backward: dX --------------------------------------
template <int MB, bool MASK>
__global__ void __launch_bounds__(256)
k_dx_fast(DxParams p) {
    using T = Tile<MB>;
    constexpr int BM = T::BM, kStage = T::A_KC + T::B_RC, kOffB = T::A_KC;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
    const unsigned ntn = (unsigned)((p.Fi + BN - 1) / BN), per_row = ntn * p.nseg;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / per_row) * BM;
    const int s = (int)((lb % per_row) / ntn);
    const int64_t col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);
    FastOp<BM, true, MASK> A;
    FastOp<BN, false, false> B;
    A.init_kc(p.g.p, p.mask.p, p.g.ld, row0, p.N, 0);
    B.init_rc(p.w[s].p, nullptr, p.Fi, 0, col0, p.Fi);
    fast_loop<MB, true, false, kStage, kOffB>(lds, (int)(p.Fo / BK), A, B, NoOp{}, NoOp{}, acc, wm,
                                              wn);
    float *out = p.gx[s];
    const int64_t ldo = p.ldgx[s];
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (row < p.N && col < p.Fi) out[row * ldo + col] = v;
    });
}

// ------------------------------- backward: dW --------------------------------------
template <int MB, bool MASK>
__global__ void __launch_bounds__(256)
k_dw_fast(DwParams p) {
    using T = Tile<MB>;
    constexpr int BM = T::BM, kStage = T::A_RC + T::B_RC, kOffB = T::A_RC;
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
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
    using OA = FastOp<BM, false, MASK>;
    OA A;
    FastOp<BN, false, false> B;
    A.init_rc(p.g.p, p.mask.p, p.g.ld, n_beg, o0, p.Fo);
    B.init_rc(p.x[s].p, nullptr, p.x[s].ld, n_beg, f0, p.Fi);
    auto bias_hook = [&](const OA &a) {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < OA::NV; ++j) {
                const float4 x = a.value(j);
                bsum4.x += x.x, bsum4.y += x.y, bsum4.z += x.z, bsum4.w += x.w;
            }
        }
    };
    fast_loop<MB, false, false, kStage, kOffB>(lds, (int)((n_end - n_beg) / BK), A, B, NoOp{},
                                               bias_hook, acc, wm, wn);
    float *out = p.partial + ((int64_t)chunk * p.nseg + s) * p.Fo * p.Fi;
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t o = o0 + r, f = f0 + c;
        if (o < p.Fo && f < p.Fi) out[o * p.Fi + f] = v;
    });
    if (do_bias) {
        __syncthreads();
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[threadIdx.x] = bsum4;
        __syncthreads();
        if (threadIdx.x < OA::PER) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < 256 / OA::PER; ++g) {
                const float4 v = red[g * OA::PER + threadIdx.x];
                t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
            }
            float *bp = p.bias_partial + (int64_t)chunk * p.Fo;
            // the column clamp in init_rc duplicates columns past Fo: store only real ones
            const int64_t o = o0 + 4 * threadIdx.x;
            if (o + 4 <= p.Fo) {
                bp[o + 0] = t.x, bp[o + 1] = t.y, bp[o + 2] = t.z, bp[o + 3] = t.w;
            }
        }
    }
}

static inline bool al16(const void *q) { return ((uintptr_t)q & 15) == 0; }

bool fwd_fast_launch(const FwdParams &p, int mb, hipStream_t hs) {
    if (p.Fi % BK != 0 || p.Fi < BK) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (!al16(p.x[s].p) || !al16(p.w[s].p) || p.x[s].ld % 4 != 0 || p.x[s].ld != p.x[0].ld)
            return false;
    const int64_t grid = ((p.N + 64 * mb - 1) / (64 * mb)) * ((p.Fo + BN - 1) / BN);
    const dim3 gd((unsigned)grid), bd(256);
    if (mb == 2)
        DC_LAUNCH((k_fwd_fast<2>), gd, bd, 0, hs, p);
    else
        DC_LAUNCH((k_fwd_fast<1>), gd, bd, 0, hs, p);
    return true;
}

bool dx_fast_launch(const DxParams &p, int mb, hipStream_t hs) {
    if (p.Fo % BK != 0 || p.Fi % 4 != 0 || p.Fi < 4 || !al16(p.g.p) || p.g.ld % 4 != 0) return false;
    if (p.has_mask && (!al16(p.mask.p) || p.mask.ld != p.g.ld)) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (!al16(p.w[s].p)) return false;
    const int64_t grid = ((p.N + 64 * mb - 1) / (64 * mb)) * ((p.Fi + BN - 1) / BN) * p.nseg;
    const dim3 gd((unsigned)grid), bd(256);
#define DC_L(MB_, M_) DC_LAUNCH((k_dx_fast<MB_, M_>), gd, bd, 0, hs, p)
    if (mb == 2) { if (p.has_mask) DC_L(2, true); else DC_L(2, false); }
    else { if (p.has_mask) DC_L(1, true); else DC_L(1, false); }
#undef DC_L
    return true;
}

bool dw_fast_launch(const DwParams &p, int mb, hipStream_t hs) {
    if (p.N % BK != 0 || p.chunk_rows % BK != 0 || p.Fi % 4 != 0 || p.Fo % 4 != 0 || p.Fi < 4 ||
        p.Fo < 4 || !al16(p.g.p) || p.g.ld % 4 != 0)
        return false;
    if (p.has_mask && (!al16(p.mask.p) || p.mask.ld != p.g.ld)) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (!al16(p.x[s].p) || p.x[s].ld % 4 != 0) return false;
    const int64_t tiles = ((p.Fo + 64 * mb - 1) / (64 * mb)) * ((p.Fi + BN - 1) / BN);
    const dim3 gd((unsigned)(tiles * p.nseg * p.nchunks)), bd(256);
#define DC_L(MB_, M_) DC_LAUNCH((k_dw_fast<MB_, M_>), gd, bd, 0, hs, p)
    if (mb == 2) { if (p.has_mask) DC_L(2, true); else DC_L(2, false); }
    else { if (p.has_mask) DC_L(1, true); else DC_L(1, false); }
#undef DC_L
    return true;
}

}  // namespace dc
