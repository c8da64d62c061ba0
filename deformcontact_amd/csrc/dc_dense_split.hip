// dc_dense_split.hip -- fp32-accurate dense block on the bf16 matrix cores (gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 operand is the exact sum
// of three bf16 numbers, x = hi + mid + lo (8 + 8 + 8 significant bits, each remainder exactly
// representable), and a bf16 x bf16 product is exact in the fp32 accumulator, so
//
//   a*b = hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid) + O(2^-24 |a*b|)
//
// i.e. SIX v_mfma_f32_32x32x16_bf16 reproduce the fp32 product to fp32 rounding level (simulated
// error of the split: 6e-9 relative, below the 5e-7 of the fp32 accumulation itself) at
// 16/6 = 2.7x the fp32 MFMA peak.  Range is not an issue (bf16 has the fp32 exponent).
//
// This file: the two dense kernels whose operands are both reduction-contiguous ("KC/KC"):
//   fwd : out[N,Fo]    = act(sum_s xs[s][N,Fi] . ws[s][Fo,Fi]^T + b)
//   dX  : gxs[s][N,Fi] = (g*relu')[N,Fo] . wt[s][Fi,Fo]^T      with wt[s] = ws[s]^T (tiny
//                                                               transposes done per call)
// fp32 stays the storage format everywhere: tiles are split into the three planes on their way
// from registers into LDS (v_cvt_pk_bf16_f32 + exact subtractions), 112-byte LDS rows
// (3 x 32 B planes + 16 B pad) keep both the b64 plane stores and the b128 fragment reads
// conflict-free.  dW (both operands node-major) stays on the fp32 kernel.
#include "dc_dense.h"

// timing-only ablations of k_dw_split (tools/r06/dw_abl.sh builds this file with -DDC_DWS_ABL=<bits>; results are wrong by
// construction): 1 no MFMAs, 2 no mask loads, 4 no partial stores, 8 no gradient loads, 16 no x loads
#ifndef DC_DWS_ABL
#define DC_DWS_ABL 0
#endif

namespace dc {

using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kPlane = 32;                // bytes per plane inside a row (16 bf16)

// NP = number of bf16 MFMA products per tile: 6 = fp32-accurate (hi, mid, lo planes),
// 3 = hi/mid planes, products hi*hi + hi*mid + mid*hi (~4e-6 relative), 1 = plain bf16.
//
// NP = 2 is the fp16x2 ("h2") mode: each operand is first multiplied by an exact power of two
// that puts its row / tensor maximum in [2^14, 2^15) (dc_dense.h: H2Scales) and then written as
// TWO fp16 planes, x*s = h1 + h2 + O(2^-23 |x*s|)  (11 + 11 significant bits + the sign of the
// remainder; v_mfma_f32_32x32x16_f16 honours fp16 denormals, so elements down to 2^-39 of the
// maximum keep absolute accuracy).  THREE products h1*h1 + h1*h2 + h2*h1 then carry the fp32
// product to 2^-22 relative (simulated: 8e-8 on the dense block, below the 4e-7 of fp32
// accumulation) at half the matrix work of the 6-product bf16 form.
template <int NP> struct Planes {
    static_assert(NP == 6 || NP == 3 || NP == 1 || NP == 2, "products per tile: 6, 3, 1 or 2 (= fp16x2)");
    static constexpr bool F16 = NP == 2;
    static constexpr int P = NP == 6 ? 3 : ((NP == 3 || NP == 2) ? 2 : 1);
    static constexpr int NPROD = NP == 2 ? 3 : NP;
    static constexpr int SROW = P * kPlane + 16;     // 112 / 80 / 48 B: conflict-free b128 reads
};

using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ void split4_h2(const float4 &v, f16x4 &h, f16x4 &l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 a = (_Float16)x[i];
        h[i] = a;
        l[i] = (_Float16)(x[i] - (float)a);          // remainder is exact in fp32
    }
}

__device__ __forceinline__ void split1(float x, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
    hi = (__bf16)x;
    const float r = x - (float)hi;        // exact
    mid = (__bf16)r;
    const float r2 = r - (float)mid;      // exact
    lo = (__bf16)r2;
}

__device__ __forceinline__ void split4(const float4 &v, bf16x4 &hi, bf16x4 &mid, bf16x4 &lo) {
    __bf16 h, m, l;
    split1(v.x, h, m, l); hi[0] = h, mid[0] = m, lo[0] = l;
    split1(v.y, h, m, l); hi[1] = h, mid[1] = m, lo[1] = l;
    split1(v.z, h, m, l); hi[2] = h, mid[2] = m, lo[2] = l;
    split1(v.w, h, m, l); hi[3] = h, mid[3] = m, lo[3] = l;
}

// fp32 [rows][K] (k contiguous) operand: per-thread float4 pointers, split at LDS-store time
template <int ROWS, bool MASK, int NP>
struct SplitOp {
    static constexpr int SROW = Planes<NP>::SROW, P = Planes<NP>::P;
    static constexpr int NV = ROWS / 64;
    const float *p[NV];
    const float *pm[MASK ? NV : 1];
    float4 v[NV];
    float4 m[MASK ? NV : 1];
    int off[NV];                                     // LDS byte offsets of the hi plane
    float sc[NV];                                    // fp16x2 mode: power-of-two scale of row j

    __device__ __forceinline__ void init(const float *base, const float *mbase, int64_t ld,
                                         int64_t row0, int64_t nrows, int64_t mld = 0) {
        const int k4 = threadIdx.x & 3, r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            int64_t row = row0 + r + 64 * j;
            row = row < nrows ? row : nrows - 1;
            p[j] = base + row * ld + 4 * k4;
            if (MASK) pm[j] = mbase + row * (mld ? mld : ld) + 4 * k4;
            off[j] = (r + 64 * j) * SROW + 8 * k4;
        }
    }
    __device__ __forceinline__ void rebase(int64_t delta) {
#pragma unroll
        for (int j = 0; j < NV; ++j) p[j] += delta;
    }
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = *reinterpret_cast<const float4 *>(p[j]);
            p[j] += BK;
            if (MASK) {
                m[j] = *reinterpret_cast<const float4 *>(pm[j]);
                pm[j] += BK;
            }
        }
    }
    __device__ __forceinline__ void store(char *lds) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 x = v[j];
            if (MASK)
                x = make_float4(m[j].x > 0.f ? x.x : 0.f, m[j].y > 0.f ? x.y : 0.f,
                                m[j].z > 0.f ? x.z : 0.f, m[j].w > 0.f ? x.w : 0.f);
            if (Planes<NP>::F16) {
                x = make_float4(x.x * sc[j], x.y * sc[j], x.z * sc[j], x.w * sc[j]);
                f16x4 h, l;
                split4_h2(x, h, l);
                *reinterpret_cast<f16x4 *>(lds + off[j]) = h;
                *reinterpret_cast<f16x4 *>(lds + off[j] + kPlane) = l;
                continue;
            }
            bf16x4 hi, mid, lo;
            split4(x, hi, mid, lo);
            *reinterpret_cast<bf16x4 *>(lds + off[j]) = hi;
            if (P > 1) *reinterpret_cast<bf16x4 *>(lds + off[j] + kPlane) = mid;
            if (P > 2) *reinterpret_cast<bf16x4 *>(lds + off[j] + 2 * kPlane) = lo;
        }
    }
    __device__ __forceinline__ void set_scale(float s) {
#pragma unroll
        for (int j = 0; j < NV; ++j) sc[j] = s;
    }
};

template <int MB, int NP>
struct SplitFrag {
    bf16x8 a[MB][Planes<NP>::P];
    bf16x8 b[2][Planes<NP>::P];
};

template <int MB, int NP>
__device__ __forceinline__ void load_split_frag(SplitFrag<MB, NP> &f, const char *As,
                                                const char *Bs, int wm, int wn) {
    constexpr int SROW = Planes<NP>::SROW, P = Planes<NP>::P;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int pl = 0; pl < P; ++pl)
            f.a[mb][pl] = *reinterpret_cast<const bf16x8 *>(
                As + (wm * 32 * MB + mb * 32 + r) * SROW + pl * kPlane + 16 * h);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int pl = 0; pl < P; ++pl)
            f.b[nb][pl] = *reinterpret_cast<const bf16x8 *>(
                Bs + (wn * 64 + nb * 32 + r) * SROW + pl * kPlane + 16 * h);
}

// NP products, smallest terms first
template <int MB, int NP>
__device__ __forceinline__ void mma_split(const SplitFrag<MB, NP> &f, f32x16 (&acc)[MB][2]) {
    constexpr int HI = 0, MID = 1, LO = 2;
    constexpr int pa6[6] = {LO, HI, MID, MID, HI, HI}, pb6[6] = {HI, LO, MID, HI, MID, HI};
    constexpr int pa3[3] = {MID, HI, HI}, pb3[3] = {HI, MID, HI};
    constexpr int NPROD = Planes<NP>::NPROD;
#pragma unroll
    for (int t = 0; t < NPROD; ++t) {
        const int ia = NPROD == 6 ? pa6[t] : (NPROD == 3 ? pa3[t] : HI);
        const int ib = NPROD == 6 ? pb6[t] : (NPROD == 3 ? pb3[t] : HI);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (Planes<NP>::F16)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(f16x8, f.a[mb][ia]), __builtin_bit_cast(f16x8, f.b[nb][ib]),
                        acc[mb][nb], 0, 0, 0);
                else
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mb][ia], f.b[nb][ib],
                                                                          acc[mb][nb], 0, 0, 0);
            }
    }
}

template <int MB, int NP, typename OA, typename OB, typename Next>
__device__ __forceinline__ void split_loop(char *lds, int nst, OA &A, OB &B, Next &&next_stage,
                                           f32x16 (&acc)[MB][2], int wm, int wn) {
    constexpr int SROW = Planes<NP>::SROW;
    constexpr int BM = 64 * MB, kOffB = BM * SROW, kStage = (BM + BN) * SROW;
    if (nst <= 0) return;
    A.load();
    B.load();
    next_stage();
    A.store(lds);
    B.store(lds + kOffB);
    if (nst > 1) {
        A.load();
        B.load();
        next_stage();
    }
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        char *cur = lds + (it & 1) * kStage, *nxt = lds + ((it + 1) & 1) * kStage;
        SplitFrag<MB, NP> f;
        load_split_frag<MB, NP>(f, cur, cur + kOffB, wm, wn);
        if (it + 1 < nst) {
            A.store(nxt);
            B.store(nxt + kOffB);
        }
        if (it + 2 < nst) {
            A.load();
            B.load();
            next_stage();
        }
        mma_split<MB, NP>(f, acc);
        __syncthreads();
    }
}

// ------------------------------- forward -------------------------------------------
template <int MB, int NP>
__global__ void __launch_bounds__(256)
k_fwd_split(FwdParams p) {
    constexpr int BM = 64 * MB, SROW = Planes<NP>::SROW;
    __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * SROW];
    const unsigned ntn = (unsigned)((p.Fo + BN - 1) / BN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / ntn) * BM, col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);
    SplitOp<BM, false, NP> A;
    SplitOp<BN, false, NP> B;
    A.init(p.x[0].p, nullptr, p.x[0].ld, row0, p.N);
    B.init(p.w[0].p, nullptr, p.Fi, col0, p.Fo);
    constexpr bool F16 = Planes<NP>::F16;
    __shared__ float s_inv[F16 ? BM : 1];           // fp16x2: per-row unscale factors
    if (F16) {
        const int r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < BM / 64; ++j) {
            int64_t row = row0 + r + 64 * j;
            row = row < p.N ? row : p.N - 1;
            const float m = p.h2.a_rowmax[row];
            A.sc[j] = h2_scale(m);
            if ((threadIdx.x & 3) == 0) s_inv[r + 64 * j] = h2_unscale(m);
        }
#pragma unroll
        for (int j = 0; j < BN / 64; ++j) {          // B rows = output columns: one scale each
            int64_t col = col0 + r + 64 * j;
            col = col < p.Fo ? col : p.Fo - 1;
            B.sc[j] = h2_scale(p.h2.b_rowmax[col]);
        }
    }
    const int kst = (int)(p.Fi / BK), nst = kst * p.nseg;
    int kk = 0, seg = 0;
    auto next_stage = [&]() {
        if (++kk == kst && seg + 1 < p.nseg) {
            kk = 0;
            A.rebase((p.x[seg + 1].p - p.x[seg].p) - p.Fi);
            B.rebase((p.w[seg + 1].p - p.w[seg].p) - p.Fi);
            ++seg;
        }
    };
    split_loop<MB, NP>(lds, nst, A, B, next_stage, acc, wm, wn);

    float bcol[2], icol[2] = {1.f, 1.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + wn * 64 + nb * 32 + (threadIdx.x & 31);
        bcol[nb] = (p.bias && col < p.Fo) ? p.bias[col] : 0.f;
        if (F16) icol[nb] = h2_unscale(p.h2.b_rowmax[col < p.Fo ? col : p.Fo - 1]);
    }
    const bool relu = p.relu != 0;
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (row < p.N && col < p.Fo) {
            if (F16) v = (v * s_inv[r]) * icol[(c >> 5) & 1];
            v += bcol[(c >> 5) & 1];
            if (relu) v = fmaxf(v, 0.f);
            p.out[row * p.ldo + col] = v;
        }
    });
}

// ------------------------------- backward: dX --------------------------------------
// p.w[s] here are the TRANSPOSED weights wt[s] [Fi, Fo] (ld = Fo)
template <int MB, bool MASK, int NP>
__global__ void __launch_bounds__(256)
k_dx_split(DxParams p) {
    constexpr int BM = 64 * MB, SROW = Planes<NP>::SROW;
    __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * SROW];
    const unsigned ntn = (unsigned)((p.Fi + BN - 1) / BN), per_row = ntn * p.nseg;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)(lb / per_row) * BM;
    const int s = (int)((lb % per_row) / ntn);
    const int64_t col0 = (int64_t)(lb % ntn) * BN;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    f32x16 acc[MB][2];
    zero_acc<MB>(acc);
    SplitOp<BM, MASK, NP> A;
    SplitOp<BN, false, NP> B;
    A.init(p.g.p, p.mask.p, p.g.ld, row0, p.N, p.mask.ld);
    B.init(p.w[s].p, nullptr, p.Fo, col0, p.Fi);
    constexpr bool F16 = Planes<NP>::F16;
    __shared__ float s_inv[F16 ? BM : 1];
    float invb = 1.f;
    if (F16) {
        const int r = threadIdx.x >> 2;
#pragma unroll
        for (int j = 0; j < BM / 64; ++j) {
            int64_t row = row0 + r + 64 * j;
            row = row < p.N ? row : p.N - 1;
            const float m = p.h2.a_rowmax[row];      // >= max |(g * relu')[row, :]|
            A.sc[j] = h2_scale(m);
            if ((threadIdx.x & 3) == 0) s_inv[r + 64 * j] = h2_unscale(m);
        }
        // B = W^T: one scale for the whole weight tensor, the max of the [Fo] row maxima
        const float bm = h2_block_max(p.h2.b_rowmax, 0, p.Fo, reinterpret_cast<float *>(lds));
        B.set_scale(h2_scale(bm));
        invb = h2_unscale(bm);
    }
    split_loop<MB, NP>(lds, (int)(p.Fo / BK), A, B, []() {}, acc, wm, wn);
    float *out = p.gx[s];
    const int64_t ldo = p.ldgx[s];
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t row = row0 + r, col = col0 + c;
        if (F16) v = (v * s_inv[r]) * invb;
        if (row < p.N && col < p.Fi) out[row * ldo + col] = v;
    });
}

// ------------------------------- backward: dW --------------------------------------
// Both operands are node-major ([n][o] and [n][f], the reduction index n is the SLOW one), so
// the split planes are kept as [k][m] images (k = node within the stage) and the MFMA operands
// (8 consecutive k for one m per lane) come out of gfx950's transposing LDS read
// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of (row q, columns
// 4p..4p+3) of a 4 x 16 block and lane i receives column i of the 4 rows.  Rows are padded by
// 64 B so the four rows of a block land on disjoint bank quarters.
template <int COLS, int NP> struct TrImage {
    static constexpr int ROWB = COLS * 2 + 64;          // bytes per k row of one plane
    static constexpr int PLANE = BK * ROWB;             // bytes per plane
    static constexpr int BYTES = Planes<NP>::P * PLANE;
};

// fp32 [K][cols] (col contiguous) operand tile BK x COLS: split at LDS-store time into 3 images
template <int COLS, bool MASK, int NP>
struct SplitOpRC {
    static constexpr int P = Planes<NP>::P;
    static constexpr int NV = COLS / 64;
    static constexpr int PER = COLS / 4, KPER = 256 / PER;
    using Img = TrImage<COLS, NP>;
    const float *p[NV];
    const float *pm[MASK ? NV : 1];
    float4 v[NV];
    float4 m[MASK ? NV : 1];
    int off[NV];
    int64_t step, mstep;
    float sc;                                        // fp16x2 mode: power-of-two tensor scale

    __device__ __forceinline__ void init(const float *base, const float *mbase, int64_t ld,
                                         int64_t k0, int64_t col0, int64_t ncols, int64_t mld = 0) {
        const int c4 = threadIdx.x % PER, kr = threadIdx.x / PER;
        int64_t col = col0 + 4 * c4;
        col = col + 4 <= ncols ? col : ncols - 4;
        mld = mld ? mld : ld;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t k = k0 + kr + KPER * j;
            p[j] = base + k * ld + col;
            if (MASK) pm[j] = mbase + k * mld + col;
            off[j] = (kr + KPER * j) * Img::ROWB + 8 * c4;
        }
        step = BK * ld;
        mstep = BK * mld;
    }
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if ((DC_DWS_ABL & 8) && MASK) v[j] = make_float4(1.f, 2.f, 3.f, (float)j);
            else if ((DC_DWS_ABL & 16) && !MASK) v[j] = make_float4(1.f, 2.f, 3.f, (float)j);
            else v[j] = *reinterpret_cast<const float4 *>(p[j]);
            p[j] += step;
            if (MASK) {
                if (DC_DWS_ABL & 2) m[j] = make_float4(1.f, -1.f, 1.f, 1.f);
                else m[j] = *reinterpret_cast<const float4 *>(pm[j]);
                pm[j] += mstep;
            }
        }
    }
    __device__ __forceinline__ float4 value(int j) const {
        if (!MASK) return v[j];
        return make_float4(m[j].x > 0.f ? v[j].x : 0.f, m[j].y > 0.f ? v[j].y : 0.f,
                           m[j].z > 0.f ? v[j].z : 0.f, m[j].w > 0.f ? v[j].w : 0.f);
    }
    __device__ __forceinline__ void store(char *lds) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (Planes<NP>::F16) {
                float4 x = value(j);
                x = make_float4(x.x * sc, x.y * sc, x.z * sc, x.w * sc);
                f16x4 h, l;
                split4_h2(x, h, l);
                *reinterpret_cast<f16x4 *>(lds + off[j]) = h;
                *reinterpret_cast<f16x4 *>(lds + off[j] + Img::PLANE) = l;
                continue;
            }
            bf16x4 hi, mid, lo;
            split4(value(j), hi, mid, lo);
            *reinterpret_cast<bf16x4 *>(lds + off[j]) = hi;
            if (P > 1) *reinterpret_cast<bf16x4 *>(lds + off[j] + Img::PLANE) = mid;
            if (P > 2) *reinterpret_cast<bf16x4 *>(lds + off[j] + 2 * Img::PLANE) = lo;
        }
    }
};

using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// one MFMA operand (8 consecutive k for column m0 + (lane & 31)) from a [k][m] plane image
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_operand(const char *plane, int m0) {
    const int lane = threadIdx.x & 63, g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int h = g >> 1;
    const char *a = plane + (8 * h + q) * ROWB + (m0 + 16 * (g & 1) + 4 * pp) * 2;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(uintptr_t)(a));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_s16x4 *)(uintptr_t)(a + 4 * ROWB));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo4, u.s[1] = hi4;
    return u.b;
}

template <int MB, bool MASK, int NP>
__global__ void __launch_bounds__(256)
k_dw_split(DwParams p) {
    constexpr int BM = 64 * MB, P = Planes<NP>::P;
    using IA = TrImage<BM, NP>;
    using IB = TrImage<BN, NP>;
    constexpr int kStage = IA::BYTES + IB::BYTES, kOffB = IA::BYTES;
    __shared__ __attribute__((aligned(16))) char lds[2 * kStage];
    const unsigned ntm = (unsigned)((p.Fo + BM - 1) / BM), ntn = (unsigned)((p.Fi + BN - 1) / BN);
    const unsigned tiles = ntm * ntn, per_chunk = tiles * p.nseg;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);   // a chunk's tiles share one L2
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
    using OA = SplitOpRC<BM, MASK, NP>;
    OA A;
    SplitOpRC<BN, false, NP> B;
    A.init(p.g.p, p.mask.p, p.g.ld, n_beg, o0, p.Fo, p.mask.ld);
    B.init(p.x[s].p, nullptr, p.x[s].ld, n_beg, f0, p.Fi);
    float inva = 1.f, invb = 1.f;
    if (Planes<NP>::F16) {
        // the contraction runs over this block's node chunk: one scale per operand and chunk
        // (each chunk's partial product is stored separately and summed in fp32 afterwards)
        const float am = h2_block_max(p.h2.a_rowmax, n_beg, n_end, reinterpret_cast<float *>(lds));
        const float bm = h2_block_max(p.h2.b_rowmax, n_beg, n_end, reinterpret_cast<float *>(lds));
        A.sc = h2_scale(am), B.sc = h2_scale(bm);
        inva = h2_unscale(am), invb = h2_unscale(bm);
    }
    auto bias_acc = [&]() {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < OA::NV; ++j) {
                const float4 x = A.value(j);
                bsum4.x += x.x, bsum4.y += x.y, bsum4.z += x.z, bsum4.w += x.w;
            }
        }
    };
    const int nst = (int)((n_end - n_beg) / BK);
    if (nst > 0) {
        A.load();
        B.load();
        A.store(lds);
        B.store(lds + kOffB);
        bias_acc();
        if (nst > 1) {
            A.load();
            B.load();
        }
    }
    __syncthreads();
    for (int it = 0; it < nst; ++it) {
        const char *cur = lds + (it & 1) * kStage;
        char *nxt = lds + ((it + 1) & 1) * kStage;
        SplitFrag<MB, NP> f;
#pragma unroll
        for (int pl = 0; pl < P; ++pl) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                f.a[mb][pl] = tr_operand<IA::ROWB>(cur + pl * IA::PLANE, wm * 32 * MB + mb * 32);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                f.b[nb][pl] = tr_operand<IB::ROWB>(cur + kOffB + pl * IB::PLANE, wn * 64 + nb * 32);
        }
        if (it + 1 < nst) {
            A.store(nxt);
            B.store(nxt + kOffB);
            bias_acc();
        }
        if (it + 2 < nst) {
            A.load();
            B.load();
        }
        if (!(DC_DWS_ABL & 1)) mma_split<MB, NP>(f, acc);
        else acc[0][0][it & 15] += (float)f.a[0][0][0] * (float)f.b[0][0][0];
        __syncthreads();
    }
    float *out = p.partial + ((int64_t)chunk * p.nseg + s) * p.Fo * p.Fi;
    for_each_acc<MB>(acc, wm, wn, [&](int r, int c, float v) {
        const int64_t o = o0 + r, ff = f0 + c;
        if (Planes<NP>::F16) v = (v * inva) * invb;
        if (o < p.Fo && ff < p.Fi && (!(DC_DWS_ABL & 4) || v == 12345.678f)) out[o * p.Fi + ff] = v;
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
            const int64_t o = o0 + 4 * threadIdx.x;
            if (o + 4 <= p.Fo) bp[o + 0] = t.x, bp[o + 1] = t.y, bp[o + 2] = t.z, bp[o + 3] = t.w;
        }
    }
}

// ---- dW, fp16x2, 128 (Fo) x 256 (Fi) tiles, 32 nodes per stage (r02) ------------------------------
// Same arithmetic as k_dw_split<2, false, 2> over the same node chunks (partials bit-identical); what changes
// is the shape of the loop, for the reasons measured on the forward-shaped block (dc_dense_h2w.hip): with
// 128 x 128 tiles a launch moves 536 MB from L2 to the CUs (g is re-read by 8 column tiles, x by 2 row
// tiles) and runs 12 MFMAs per wave between barriers.  Here a workgroup of 8 waves owns a 128 x 256 tile
// (one whole segment wide: 402 MB per launch), a stage is 32 nodes = two 16-deep k-steps (24 MFMAs per wave
// and barrier), and the loads of stage it+2 are issued at the top of stage it (two register sets).
constexpr int kDwK = 32;                                   // nodes per stage
template <int COLS> struct TrImage32 {
    static constexpr int ROWB = COLS * 2 + 64;             // bytes per node row of one plane (see TrImage)
    static constexpr int PLANE = kDwK * ROWB;
    static constexpr int BYTES = 2 * PLANE;
};

// CORR: the gradient operand is g - g2_coef[row] * g2 (dc_tag_linear_bwd_dw_h2_corr); a template parameter so that the
// plain kernel keeps its register allocation
template <bool CORR = false>
__global__ void __launch_bounds__(512)
k_dw_h2w(DwParams p) {
    using IA = TrImage32<128>;
    using IB = TrImage32<256>;
    constexpr int kStage = IA::BYTES + IB::BYTES, kOffB = IA::BYTES;      // 20,480 + 36,864 B
    __shared__ __attribute__((aligned(16))) char lds[2 * kStage];
    const unsigned nto = (unsigned)(p.Fo / 128);                          // 128-row blocks of Fo
    const unsigned per_chunk = nto * (unsigned)p.nseg;                    // (Fo block, segment) tiles
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);                 // a chunk's tiles share one L2
    const unsigned chunk = lb / per_chunk, rem = lb % per_chunk;
    const int s = (int)(rem / nto);
    const int64_t o0 = (int64_t)(rem % nto) * 128;
    int64_t n_beg = (int64_t)chunk * p.chunk_rows;
    int64_t n_end = (n_beg + p.chunk_rows < p.N) ? n_beg + p.chunk_rows : p.N;
    if (p.grp.n >= 1) {                                                  // grouped: the chunk's group owns its rows
        int g = 0;
#pragma unroll
        for (int q = 1; q < kMaxGroups; ++q)
            if (q < p.grp.n && (int)chunk >= p.grp.chunk_beg[q]) g = q;
        n_beg = p.grp.row_beg[g] + (int64_t)((int)chunk - p.grp.chunk_beg[g]) * p.grp.chunk_rows[g];
        n_end = n_beg + p.grp.chunk_rows[g] < p.grp.row_end[g] ? n_beg + p.grp.chunk_rows[g] : p.grp.row_end[g];
        if (n_end < n_beg) n_end = n_beg;
    }
    const int wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
    const bool do_bias = p.bias_partial && s == 0;

    // one power-of-two scale per operand and chunk: maxima of the row maxima over the chunk's nodes
    float am = 0.f, bm = 0.f;
    for (int64_t i = n_beg + threadIdx.x; i < n_end; i += 512) {
        am = fmaxf(am, p.h2.a_rowmax[i]);
        bm = fmaxf(bm, p.h2.b_rowmax[i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        am = fmaxf(am, __shfl_xor(am, o));
        bm = fmaxf(bm, __shfl_xor(bm, o));
    }
    {
        float *red = reinterpret_cast<float *>(lds);
        if ((threadIdx.x & 63) == 0) red[wid] = am, red[8 + wid] = bm;
        __syncthreads();
        am = bm = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) am = fmaxf(am, red[i]), bm = fmaxf(bm, red[8 + i]);
        __syncthreads();
    }
    const float sca = h2_scale(am), scb = h2_scale(bm), inva = h2_unscale(am), invb = h2_unscale(bm);

    // staging: g tile 32 x 128 (32 float4 per node row, 16 rows per pass), x tile 32 x 256 (64 per row, 8 rows)
    const int c4g = threadIdx.x & 31, krg = threadIdx.x >> 5;
    const int c4x = threadIdx.x & 63, krx = threadIdx.x >> 6;
    const int64_t ldg = p.g.ld, ldx = p.x[s].ld;
    unsigned offg[2], offx[4];
    int ldsg[2], ldsx[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        offg[j] = (unsigned)((krg + 16 * j) * ldg + 4 * c4g);
        ldsg[j] = (krg + 16 * j) * IA::ROWB + 8 * c4g;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        offx[j] = (unsigned)((krx + 8 * j) * ldx + 4 * c4x);
        ldsx[j] = kOffB + (krx + 8 * j) * IB::ROWB + 8 * c4x;
    }
    const float *baseg = p.g.p + n_beg * ldg + o0;                        // wave-uniform running bases
    const float *basex = p.x[s].p + n_beg * ldx;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 vg0[2], vg1[2], vx0[4], vx1[4];
    f32x4 vc0[2], vc1[2];                                                 // (CORR) pieces of the correction operand
    float cc0[2], cc1[2];                                                 //        and their rows' coefficients
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const float *baseg2 = CORR ? p.g2 + n_beg * ldg + o0 : nullptr;       // correction operand (same layout as g)
    const float *basec = CORR ? p.g2_coef + n_beg + krg : nullptr;

    auto gload_c = [&](f32x4 (&vg)[2], f32x4 (&vx)[4], f32x4 (&vc)[2], float (&cc)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) vg[j] = *reinterpret_cast<const f32x4 *>(baseg + offg[j]);
        if (CORR) {                                                       // (combined at LDS-store time: stays in flight)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                vc[j] = *reinterpret_cast<const f32x4 *>(baseg2 + offg[j]);
                cc[j] = basec[16 * j];
            }
            baseg2 += kDwK * ldg;
            basec += kDwK;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) vx[j] = *reinterpret_cast<const f32x4 *>(basex + offx[j]);
        baseg += kDwK * ldg;
        basex += kDwK * ldx;
    };
    // the register set of a stage is identified by its g array (two sets, named)
    auto gload = [&](f32x4 (&vg)[2], f32x4 (&vx)[4]) {
        if (&vg == &vg0) gload_c(vg, vx, vc0, cc0); else gload_c(vg, vx, vc1, cc1);
    };
    auto split_store = [&](char *dst, int plane, f32x4 v, float sc) {
        v = v * sc;
        f16x4 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const _Float16 a = (_Float16)v[i];
            h[i] = a;
            l[i] = (_Float16)(v[i] - (float)a);
        }
        *reinterpret_cast<f16x4 *>(dst) = h;
        *reinterpret_cast<f16x4 *>(dst + plane) = l;
    };
    auto lstore = [&](const f32x4 (&vg)[2], const f32x4 (&vx)[4], int b) {
        char *buf = lds + b * kStage;
        const bool set0 = &vg == &vg0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 gv = vg[j];
            if (CORR) gv = set0 ? gv - vc0[j] * cc0[j] : gv - vc1[j] * cc1[j];
            split_store(buf + ldsg[j], IA::PLANE, gv, sca);
            if (do_bias) bsum += vg[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) split_store(buf + ldsx[j], IB::PLANE, vx[j], scb);
    };
    f32x16 acc[2][2];
    zero_acc<2>(acc);
    auto compute = [&](int b) {
        const char *buf = lds + b * kStage;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            SplitFrag<2, 2> f;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    f.a[mb][pl] = tr_operand<IA::ROWB>(buf + pl * IA::PLANE + ks * 16 * IA::ROWB, wm * 64 + mb * 32);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    f.b[nb][pl] = tr_operand<IB::ROWB>(buf + kOffB + pl * IB::PLANE + ks * 16 * IB::ROWB,
                                                       wn * 64 + nb * 32);
            }
            mma_split<2, 2>(f, acc);
        }
    };

    const int nst = (int)((n_end - n_beg) / kDwK);
    if (nst > 0) {
        gload(vg0, vx0);
        if (nst > 1) gload(vg1, vx1);
        lstore(vg0, vx0, 0);
    }
    __syncthreads();
    int it = 0;
#define DC_DW_STAGE(CUR, VG_L, VX_L, VG_S, VX_S)                                                      \
    gload(VG_L, VX_L);                                 /* stage it+2 */                               \
    __builtin_amdgcn_sched_barrier(0);                 /* keep the loads at the top of the stage */   \
    compute(CUR);                                                                                     \
    lstore(VG_S, VX_S, CUR ^ 1);                       /* stage it+1 */                               \
    __syncthreads();
    for (; it + 3 < nst; it += 2) {
        DC_DW_STAGE(0, vg0, vx0, vg1, vx1)
        DC_DW_STAGE(1, vg1, vx1, vg0, vx0)
    }
#undef DC_DW_STAGE
#define DC_DW_TAIL(K, CUR, VG_L, VX_L, VG_S, VX_S)                                                    \
    if (it + K < nst) {                                                                               \
        if (it + K + 2 < nst) gload(VG_L, VX_L);                                                      \
        compute(CUR);                                                                                 \
        if (it + K + 1 < nst) lstore(VG_S, VX_S, CUR ^ 1);                                            \
        __syncthreads();                                                                              \
    }
    for (; it < nst; it += 2) {
        DC_DW_TAIL(0, 0, vg0, vx0, vg1, vx1)
        DC_DW_TAIL(1, 1, vg1, vx1, vg0, vx0)
    }
#undef DC_DW_TAIL

    float *out = p.partial + ((int64_t)chunk * p.nseg + s) * p.Fo * p.Fi;
    for_each_acc<2>(acc, wm, wn, [&](int r, int c, float v) {
        out[(o0 + r) * p.Fi + c] = (v * inva) * invb;
    });
    if (do_bias) {                                      // column sums of g over the chunk's nodes
        f32x4 *red = reinterpret_cast<f32x4 *>(lds);
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < 32) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < 16; ++g) t += red[g * 32 + threadIdx.x];
            float *bp = p.bias_partial + (int64_t)chunk * p.Fo + o0 + 4 * threadIdx.x;
            bp[0] = t[0], bp[1] = t[1], bp[2] = t[2], bp[3] = t[3];
        }
    }
}

// eligible: fp16x2, pre-masked g, Fo a multiple of 128, every segment 256 wide, node chunks that are whole stages
bool dw_h2w_launch(const DwParams &p, hipStream_t hs) {
    static const int wide = getenv("DC_DW_WIDE") ? atoi(getenv("DC_DW_WIDE")) : 1;
    if ((!wide && p.grp.n < 1) || p.has_mask || p.Fo % 128 != 0 || p.Fo < 128 || p.Fi != 256 || p.N % kDwK != 0 ||
        p.chunk_rows % kDwK != 0)
        return false;
    if (p.g.ld * kDwK >= ((int64_t)1 << 30)) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (p.x[s].ld * kDwK >= ((int64_t)1 << 30)) return false;
    const int64_t grid = (p.Fo / 128) * p.nseg * p.nchunks;
    if (grid >= (int64_t)INT32_MAX) return false;
    const dim3 gd((unsigned)grid), bd(512);
    if (p.g2) DC_LAUNCH((k_dw_h2w<true>), gd, bd, 0, hs, p);
    else DC_LAUNCH((k_dw_h2w<false>), gd, bd, 0, hs, p);
    return true;
}

// wt[s][f][o] = w[s][o][f] for up to kMaxSeg segments in one launch (32x32 LDS tiles)
struct TransposeParams {
    const float *w[kMaxSeg];
    float *wt;
    int64_t Fi, Fo;
    int nseg;
};

__global__ void __launch_bounds__(256)
k_transpose_w(TransposeParams p) {
    __shared__ float tile[32][33];
    const int64_t tf = (p.Fi + 31) / 32, to = (p.Fo + 31) / 32;
    const int64_t b = blockIdx.x;
    const int s = (int)(b / (tf * to));
    const int64_t o0 = ((b % (tf * to)) / tf) * 32, f0 = (b % tf) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int64_t o = o0 + i, f = f0 + tx;
        tile[i][tx] = (o < p.Fo && f < p.Fi) ? p.w[s][o * p.Fi + f] : 0.f;
    }
    __syncthreads();
    float *dst = p.wt + (int64_t)s * p.Fi * p.Fo;
    for (int i = ty; i < 32; i += 8) {
        const int64_t f = f0 + i, o = o0 + tx;
        if (f < p.Fi && o < p.Fo) dst[f * p.Fo + o] = tile[tx][i];
    }
}

void transpose_weights_launch(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, float *wt,
                              hipStream_t hs) {
    TransposeParams t{};
    for (int s = 0; s < nseg; ++s) t.w[s] = ws[s];
    t.wt = wt, t.Fi = Fi, t.Fo = Fo, t.nseg = nseg;
    const int64_t tiles = ((Fi + 31) / 32) * ((Fo + 31) / 32) * nseg;
    DC_LAUNCH(k_transpose_w, dim3((unsigned)tiles), dim3(256), 0, hs, t);
}

static inline bool al16(const void *q) { return ((uintptr_t)q & 15) == 0; }

bool fwd_split_launch(const FwdParams &p, int mb, int np, hipStream_t hs) {
    if (np != 6 && np != 3 && np != 1 && np != 2) return false;
    if (np == 2 && (!p.h2.a_rowmax || !p.h2.b_rowmax)) return false;
    if (p.Fi % BK != 0 || p.Fi < BK) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (!al16(p.x[s].p) || !al16(p.w[s].p) || p.x[s].ld % 4 != 0 || p.x[s].ld != p.x[0].ld)
            return false;
    const int64_t grid = ((p.N + 64 * mb - 1) / (64 * mb)) * ((p.Fo + BN - 1) / BN);
    const dim3 gd((unsigned)grid), bd(256);
#define DC_L(MB_, NP_) DC_LAUNCH((k_fwd_split<MB_, NP_>), gd, bd, 0, hs, p)
    if (mb == 2) { if (np == 6) DC_L(2, 6); else if (np == 3) DC_L(2, 3); else if (np == 2) DC_L(2, 2); else DC_L(2, 1); }
    else { if (np == 6) DC_L(1, 6); else if (np == 3) DC_L(1, 3); else if (np == 2) DC_L(1, 2); else DC_L(1, 1); }
#undef DC_L
    return true;
}

bool dx_split_eligible(const DxParams &p) {
    if (p.Fo % BK != 0 || p.Fo < BK || p.Fi % 4 != 0 || !al16(p.g.p) || p.g.ld % 4 != 0) return false;
    if (p.has_mask && (!al16(p.mask.p) || p.mask.ld % 4 != 0)) return false;
    return true;
}

// `p.w[s]` = original weights [Fo,Fi]; `wt` = workspace of nseg*Fi*Fo floats (16-B aligned)
bool dx_split_launch(DxParams p, float *wt, int mb, int np, hipStream_t hs) {
    if (!dx_split_eligible(p) || !al16(wt) || (np != 6 && np != 3 && np != 1 && np != 2)) return false;
    if (np == 2 && (!p.h2.b_rowmax || !p.h2.a_rowmax)) return false;
    TransposeParams t{};
    for (int s = 0; s < p.nseg; ++s) t.w[s] = p.w[s].p;
    t.wt = wt, t.Fi = p.Fi, t.Fo = p.Fo, t.nseg = p.nseg;
    const int64_t tiles = ((p.Fi + 31) / 32) * ((p.Fo + 31) / 32) * p.nseg;
    DC_LAUNCH(k_transpose_w, dim3((unsigned)tiles), dim3(256), 0, hs, t);
    for (int s = 0; s < p.nseg; ++s) p.w[s] = Mat{wt + (int64_t)s * p.Fi * p.Fo, p.Fo};
    const int64_t grid = ((p.N + 64 * mb - 1) / (64 * mb)) * ((p.Fi + BN - 1) / BN) * p.nseg;
    const dim3 gd((unsigned)grid), bd(256);
#define DC_L(MB_, M_)                                                                 \
    do {                                                                              \
        if (np == 6) DC_LAUNCH((k_dx_split<MB_, M_, 6>), gd, bd, 0, hs, p);       \
        else if (np == 3) DC_LAUNCH((k_dx_split<MB_, M_, 3>), gd, bd, 0, hs, p);  \
        else if (np == 2) DC_LAUNCH((k_dx_split<MB_, M_, 2>), gd, bd, 0, hs, p);  \
        else DC_LAUNCH((k_dx_split<MB_, M_, 1>), gd, bd, 0, hs, p);               \
    } while (0)
    if (mb == 2) { if (p.has_mask) DC_L(2, true); else DC_L(2, false); }
    else { if (p.has_mask) DC_L(1, true); else DC_L(1, false); }
#undef DC_L
    return true;
}

bool dw_split_launch(const DwParams &p, int mb, int np, hipStream_t hs) {
    if (np != 6 && np != 3 && np != 1 && np != 2) return false;
    if (np == 2 && (!p.h2.a_rowmax || !p.h2.b_rowmax)) return false;
    if (p.N % BK != 0 || p.chunk_rows % BK != 0 || p.Fi % 4 != 0 || p.Fo % 4 != 0 || p.Fi < 4 ||
        p.Fo < 4 || !al16(p.g.p) || p.g.ld % 4 != 0)
        return false;
    if (p.has_mask && (!al16(p.mask.p) || p.mask.ld % 4 != 0)) return false;
    for (int s = 0; s < p.nseg; ++s)
        if (!al16(p.x[s].p) || p.x[s].ld % 4 != 0) return false;
    if (np == 2 && dw_h2w_launch(p, hs)) return true;
    const int64_t tiles = ((p.Fo + 64 * mb - 1) / (64 * mb)) * ((p.Fi + BN - 1) / BN);
    const dim3 gd((unsigned)(tiles * p.nseg * p.nchunks)), bd(256);
#define DC_L(MB_, M_)                                                                 \
    do {                                                                              \
        if (np == 6) DC_LAUNCH((k_dw_split<MB_, M_, 6>), gd, bd, 0, hs, p);       \
        else if (np == 3) DC_LAUNCH((k_dw_split<MB_, M_, 3>), gd, bd, 0, hs, p);  \
        else if (np == 2) DC_LAUNCH((k_dw_split<MB_, M_, 2>), gd, bd, 0, hs, p);  \
        else DC_LAUNCH((k_dw_split<MB_, M_, 1>), gd, bd, 0, hs, p);               \
    } while (0)
    if (mb == 2) { if (p.has_mask) DC_L(2, true); else DC_L(2, false); }
    else { if (p.has_mask) DC_L(1, true); else DC_L(1, false); }
#undef DC_L
    return true;
}

}  // namespace dc
