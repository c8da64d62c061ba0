// dc_dense_bf16.hip -- the TAGConv dense block over bf16-STORED features (gfx950).
//
//   out[N,Fo] = act( A[N,K] . W[Fo,K]^T + bias )     A, W bf16 (K-contiguous), fp32 accumulate
//
// BASELINE.json configs[4] ("bf16 node/edge MLPs on MFMA"): A is the hop slab [x | A x | A^2 x | A^3 x]
// as dc_spmm_bf16 leaves it (K = (K_hops+1) * Fi), W the layer's lins[k].weight concatenated along K
// and rounded to bf16 once per call (dc_weights_to_bf16), out fp32 or bf16 (the next layer's slab).
// Plain bf16 operands on v_mfma_f32_32x32x16_bf16: one MFMA product per tile - the precision of the
// reference under torch.autocast(bfloat16), NOT the fp32-accurate split forms of dc_dense_split.hip.
//
// Structure: 128 x 128 block tile, 4 MFMA waves (2 x 2) + 4 LOADER waves, stage = 32 k (64 B per
// row).  BOTH operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU,
// no ds_write), issued by the loader waves into a ring of kSlots stage buffers kAhead stages ahead
// and published with a COUNTED vmcnt + one s_barrier per stage (the pieces of the stages still in
// flight stay in flight across the barrier; barriers do not drain VMEM).  The DMA writes lane i's
// 16 bytes at base + 16 i, so a tile image is dense (64-byte rows); bank spread of the b128 fragment
// reads comes from an XOR swizzle of the four 16-byte pieces of a row (piece q of row r sits at
// position q ^ ((r >> 2) & 3)) applied on the GLOBAL side of the DMA.
#include "dc_dense.h"

namespace dc {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kBfBK = 32;        // k per stage (bf16 elements): 64 bytes per row
constexpr int kBfSlots = 4;      // ring depth
constexpr int kBfAhead = 3;      // stages of DMA in flight behind the one being multiplied

struct Bf16Params {
    const uint16_t *a;     // [N, K] bf16, leading dimension lda (elements)
    const uint16_t *w;     // [Fo, K] bf16, leading dimension K
    const float *bias;     // [Fo] or null
    void *out;             // [N, Fo] fp32 or bf16, leading dimension ldo (elements)
    int64_t lda, ldo, N, K, Fo;
    int relu, out_bf16;
};

__device__ __forceinline__ uint16_t f32_to_bf16_rne_d(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__global__ void __launch_bounds__(512)
k_fwd_bf16(Bf16Params p) {
    constexpr int BM = 128, kTile = BM * 64;                 // bytes of one operand tile of a stage
    __shared__ __attribute__((aligned(16))) char sA[kBfSlots][kTile];
    __shared__ __attribute__((aligned(16))) char sB[kBfSlots][kTile];

    const unsigned ntn = (unsigned)((p.Fo + BN - 1) / BN);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    // column tiles of one row tile are neighbours in the logical order (same XCD, launched together):
    // the second read of the A rows is an L2 hit
    const int64_t row0 = (int64_t)(lb / ntn) * BM, col0 = (int64_t)(lb % ntn) * BN;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int nst = (int)(p.K / kBfBK);

    if (wid >= 4) {
        // ---------------- four loader waves: LDS-DMA only, never read LDS ----------------
        // (hipcc guards every ds_read of a wave that has LDS-DMAs in flight with s_waitcnt vmcnt(0) -
        // it cannot tell the slots apart - so the waves that issue the DMAs are not the ones that
        // read.  FOUR of them: one wave sustains only ~25 GB/s of LDS-DMA (MI355X_MICROARCH.md,
        // ldsdma-fill), a 128 x 128 x 32 bf16 stage needs 16 KiB per ~0.12 us of MFMA time; with a
        // single loader wave this kernel ran at 0.19 of the MFMA peak.)
        const int lw = wid - 4;                  // loader wave lw fills rows [32 lw, 32 lw + 32) of both tiles
        unsigned offA[2], offB[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rowl = lw * 32 + q * 16 + (lane >> 2);
            const int piece = (lane & 3) ^ ((rowl >> 2) & 3);
            int64_t row = row0 + rowl;
            row = row < p.N ? row : p.N - 1;
            int64_t col = col0 + rowl;
            col = col < p.Fo ? col : p.Fo - 1;
            offA[q] = (unsigned)((row - row0) * p.lda + 8 * piece);       // elements
            offB[q] = (unsigned)((col - col0) * p.K + 8 * piece);
        }
        const uint16_t *baseA = p.a + row0 * p.lda;
        const uint16_t *baseB = p.w + col0 * p.K;
        auto dma = [&](int slot) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1))) *)(baseA + offA[q]),
                    (void __attribute__((address_space(3))) *)(sA[slot] + (lw * 32 + q * 16) * 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1))) *)(baseB + offB[q]),
                    (void __attribute__((address_space(3))) *)(sB[slot] + (lw * 32 + q * 16) * 64), 16, 0, 0);
            }
            baseA += kBfBK;
            baseB += kBfBK;
        };
        for (int s = 0; s < kBfAhead && s < nst; ++s) dma(s);
        for (int it = 0; it < nst; ++it) {
            // wait until only the stages AFTER `it` (4 DMA instructions per wave and stage; up to two of
            // them pending here) are still in flight; the barrier publishes stage `it` and tells the
            // loaders that the MFMA waves are done with stage it - 1, whose slot then takes stage it + 3
            const int later = nst - 1 - it;
            if (later >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);        // vmcnt(8)
            else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);   // vmcnt(4)
            else __builtin_amdgcn_s_waitcnt(0x0F70 | 0);                   // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            if (it + kBfAhead < nst) dma((it + kBfAhead) & (kBfSlots - 1));
        }
        return;
    }

    // ---------------- four MFMA waves (2 x 2), 64 x 64 of the tile each ----------------
    const int wm = wid >> 1, wn = wid & 1;
    // fragment positions: lane (fr = lane & 31, fh = lane >> 5) of k-step t reads the 8 bf16
    // k = 16 t + 8 fh .. +7 of its row = piece 2 t + fh, stored at (2 t + fh) ^ ((row >> 2) & 3)
    const int fr = lane & 31, fh = lane >> 5;
    int fragA[2][2], fragB[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = wm * 64 + b * 32 + fr, rb = wn * 64 + b * 32 + fr;
            fragA[b][t] = ra * 64 + 16 * ((2 * t + fh) ^ ((ra >> 2) & 3));
            fragB[b][t] = rb * 64 + 16 * ((2 * t + fh) ^ ((rb >> 2) & 3));
        }

    f32x16 acc[2][2];
    zero_acc<2>(acc);
    for (int it = 0; it < nst; ++it) {
        __builtin_amdgcn_s_barrier();
        const char *ca = sA[it & (kBfSlots - 1)], *cb = sB[it & (kBfSlots - 1)];
        bf16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[b][t] = *reinterpret_cast<const bf16x8 *>(ca + fragA[b][t]);
                fb[b][t] = *reinterpret_cast<const bf16x8 *>(cb + fragB[b][t]);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mb][t], fb[nb][t], acc[mb][nb], 0, 0, 0);
    }

    const bool relu = p.relu != 0;
    float bcol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + wn * 64 + nb * 32 + (threadIdx.x & 31);
        bcol[nb] = (p.bias && col < p.Fo) ? p.bias[col] : 0.f;
    }
    for_each_acc<2>(acc, wm, wn, [&](int rr, int c, float v) {
        const int64_t row = row0 + rr, col = col0 + c;
        if (row < p.N && col < p.Fo) {
            v += bcol[(c >> 5) & 1];
            if (relu) v = fmaxf(v, 0.f);
            if (p.out_bf16) ((uint16_t *)p.out)[row * p.ldo + col] = f32_to_bf16_rne_d(v);
            else ((float *)p.out)[row * p.ldo + col] = v;
        }
    });
}

// fp32 [rows, cols] (ld) -> bf16 (round to nearest even), optionally K-concatenating nseg matrices:
// dst[r, s * cols + c] = bf16(src[s][r, c])
__global__ void __launch_bounds__(256)
k_to_bf16(const float *const *src, int nseg, int64_t rows, int64_t cols, int64_t ld_src,
          uint16_t *dst, int64_t ld_dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = rows * cols;
    if (i >= per * nseg) return;
    const int s = (int)(i / per);
    const int64_t r = (i % per) / cols, c = (i % per) % cols;
    dst[r * ld_dst + s * cols + c] = f32_to_bf16_rne_d(src[s][r * ld_src + c]);
}

struct PtrPack { const float *p[kMaxSeg]; };

__global__ void __launch_bounds__(256)
k_to_bf16_pack(PtrPack src, int nseg, int64_t rows, int64_t cols, int64_t ld_src, uint16_t *dst,
               int64_t ld_dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = rows * cols;
    if (i >= per * nseg) return;
    const int s = (int)(i / per);
    const int64_t r = (i % per) / cols, c = (i % per) % cols;
    dst[r * ld_dst + s * cols + c] = f32_to_bf16_rne_d(src.p[s][r * ld_src + c]);
}

}  // namespace dc

using namespace dc;

extern "C" int dc_tag_linear_fwd_bf16(const uint16_t *a, int64_t lda, const uint16_t *w,
                                      const float *bias, int relu, void *out, int64_t ldo,
                                      int out_is_bf16, int64_t N, int64_t K, int64_t Fo,
                                      dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && K >= 0 && Fo >= 0, "dc_tag_linear_fwd_bf16: negative size");
    if (N == 0 || Fo == 0) return DC_OK;
    DC_REQUIRE(a && w && out, "dc_tag_linear_fwd_bf16: null a/w/out");
    DC_REQUIRE(K >= kBfBK && K % kBfBK == 0, "dc_tag_linear_fwd_bf16: K=%lld must be a positive multiple of %d",
               (long long)K, kBfBK);
    DC_REQUIRE(lda >= K && lda % 8 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0,
               "dc_tag_linear_fwd_bf16: a / w must be 16-byte aligned with lda %% 8 == 0");
    DC_REQUIRE(ldo >= Fo, "dc_tag_linear_fwd_bf16: ldo < Fo");
    DC_REQUIRE(lda * 128 < ((int64_t)1 << 31) && K * 128 < ((int64_t)1 << 31),
               "dc_tag_linear_fwd_bf16: leading dimension too large for 32-bit tile offsets");
    Bf16Params p{a, w, bias, out, lda, ldo, N, K, Fo, relu, out_is_bf16};
    const int64_t grid = ((N + 127) / 128) * ((Fo + BN - 1) / BN);
    hipLaunchKernelGGL(k_fwd_bf16, dim3((unsigned)grid), dim3(512), 0, stream, p);
    return check_launch("dc_tag_linear_fwd_bf16");
}

extern "C" int dc_to_bf16(const float *const *srcs, int nseg, int64_t rows, int64_t cols,
                          int64_t ld_src, uint16_t *dst, int64_t ld_dst, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, "dc_to_bf16: nseg=%d out of range", nseg);
    DC_REQUIRE(rows >= 0 && cols >= 0, "dc_to_bf16: negative size");
    if (rows == 0 || cols == 0) return DC_OK;
    DC_REQUIRE(srcs && dst && ld_src >= cols && ld_dst >= cols * nseg, "dc_to_bf16: bad pointers / leading dimensions");
    PtrPack pk{};
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(srcs[s], "dc_to_bf16: null source %d", s);
        pk.p[s] = srcs[s];
    }
    const int64_t total = rows * cols * nseg;
    hipLaunchKernelGGL(k_to_bf16_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pk,
                       nseg, rows, cols, ld_src, dst, ld_dst);
    return check_launch("dc_to_bf16");
}
