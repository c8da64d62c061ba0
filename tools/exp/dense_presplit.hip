// tools/exp/dense_presplit.hip -- EXPERIMENT (r03, not part of the product library): how fast is the wide forward block if
// NEITHER operand is split inside it?  out = act(A . W^T + b) with BOTH operands handed over as dc_tag_weight_prep's
// scaled fp16x2 image (one 64-byte record {h1[16], h2[16]} per row and 16 k): everything goes global -> LDS by LDS-DMA
// into a ring, the loop has no staging registers, no split VALU and no ds_write - three fp16 MFMA products per
// fragment pair, the product kernel's LDS row image (so the fragment reads are k_fwd_h2w's), 128 x 256 tiles, 8 waves.
// Same arithmetic as k_fwd_h2w: bit-identical outputs.  Built and driven by tools/exp/dense_presplit.py.
// The activation image would have to come from the kernels that PRODUCE the hop slab; here a
// separate pass (dc_tag_weight_prep over the slab) makes it, outside the timed region.
#include "../../deformcontact_amd/csrc/dc_dense.h"

namespace dc {
void set_error(const char *, ...) {}

using px_f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kPBM = 128, kPBN = 256, kPBK = 32;
constexpr int kPRow = 128;                                  // bytes per LDS row: 2 records x {h1[16 B x 2], h2[16 B x 2]}
constexpr int kPSzA = kPBM * kPRow, kPSzB = kPBN * kPRow;   // 16 KB + 32 KB per stage
constexpr int kPSlots = 3, kPAhead = 2;

__device__ __forceinline__ int px_swz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }

struct PreParams {
    const char *a, *w;          // images: row stride K * 4 bytes
    const float *a_rowmax, *w_rowmax, *bias;
    float *out;
    int64_t ldo, N, K, Fo;
    int relu;
};

__global__ void __launch_bounds__(512)
k_fwd_presplit(PreParams p) {
    __shared__ __attribute__((aligned(16))) char lds[kPSlots * (kPSzA + kPSzB)];      // 144 KB, one array
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row0 = (int64_t)lb * kPBM;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wm = wid >> 2, wn = wid & 3;
    const int nst = (int)(p.K / kPBK);
    const int64_t rowbytes = p.K * 4;
    // staging: a wave-instruction fills 8 rows x 128 B; A tile = 16 instructions (2 per wave), B tile = 32 (4 per wave).
    // lane l: row 8 c + (l >> 3), position l & 7 holds piece q = position ^ swz(row): record q >> 2, plane (q >> 1) & 1,
    // half q & 1 -> source byte offset 64 (q >> 2) + 32 ((q >> 1) & 1) + 16 (q & 1) inside the stage's 128 bytes of the row
    auto src_off = [&](int rl) {
        const int q = (lane & 7) ^ px_swz(rl);
        return 64 * (q >> 2) + 32 * ((q >> 1) & 1) + 16 * (q & 1);
    };
    unsigned offA[2], offB[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = 8 * (2 * wid + j) + (lane >> 3);
        int64_t row = row0 + rl;
        row = row < p.N ? row : p.N - 1;
        offA[j] = (unsigned)((row - row0) * rowbytes + src_off(rl));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = 8 * (4 * wid + j) + (lane >> 3);
        offB[j] = (unsigned)(rl * rowbytes + src_off(rl));
    }
    const char *baseA = p.a + row0 * rowbytes;
    const char *baseB = p.w;
    auto stage = [&](int slot) {                                            // 6 DMA instructions per wave and stage
        char *ta = lds + slot * (kPSzA + kPSzB), *tb = ta + kPSzA;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(baseA + offA[j]),
                                             (void __attribute__((address_space(3))) *)(ta + (2 * wid + j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(baseB + offB[j]),
                                             (void __attribute__((address_space(3))) *)(tb + (4 * wid + j) * 1024), 16, 0, 0);
        baseA += 128;
        baseB += 128;
    };
    const int fr = lane & 31, fh = lane >> 5, fsw = px_swz(fr);
    const int fragA = (wm * 64 + fr) * kPRow, fragB = (wn * 64 + fr) * kPRow;
    f32x16 acc[2][2];
    zero_acc<2>(acc);

    for (int s = 0; s < kPAhead && s < nst; ++s) stage(s);
    int slot = 0;
    for (int it = 0; it < nst; ++it) {
        const int later = nst - 1 - it;
        if (later >= 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 6);             // vmcnt(6): one later stage in flight
        else __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        __builtin_amdgcn_s_barrier();
        if (it + kPAhead < nst) stage(slot >= 1 ? slot - 1 : kPSlots - 1);   // the slot stage it - 1 lived in
        const char *ta = lds + slot * (kPSzA + kPSzB), *tb = ta + kPSzA;
        slot = slot + 1 == kPSlots ? 0 : slot + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            px_f16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fa[mb][pl] = *reinterpret_cast<const px_f16x8 *>(ta + fragA + mb * 32 * kPRow + 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fb[nb][pl] = *reinterpret_cast<const px_f16x8 *>(tb + fragB + nb * 32 * kPRow + 16 * ((4 * ks + 2 * pl + fh) ^ fsw));
            constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};             // smallest terms first (as k_fwd_h2w)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mb][pa[t]], fb[nb][pb[t]], acc[mb][nb], 0, 0, 0);
        }
    }
    // epilogue as k_fwd_h2w
    const bool relu = p.relu != 0;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = wn * 64 + nb * 32 + c;
        const float bcol = p.bias ? p.bias[col] : 0.f;
        const float icol = h2_unscale(p.w_rowmax[col]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t row = row0 + wm * 64 + mb * 32 + 8 * g + 4 * h + i;
                    if (row < p.N) {
                        float v = (acc[mb][nb][4 * g + i] * h2_unscale(p.a_rowmax[row])) * icol;
                        v += bcol;
                        if (relu) v = fmaxf(v, 0.f);
                        p.out[row * p.ldo + col] = v;
                    }
                }
    }
}
}  // namespace dc

extern "C" int presplit_run(const void *a_image, const void *w_image, const float *a_rowmax, const float *w_rowmax,
                            const float *bias, int relu, float *out, int64_t ldo, int64_t N, int64_t K, int64_t Fo,
                            void *stream) {
    using namespace dc;
    if (Fo != 256 || K % 32 != 0) return 1;
    PreParams p{(const char *)a_image, (const char *)w_image, a_rowmax, w_rowmax, bias, out, ldo, N, K, Fo, relu};
    hipLaunchKernelGGL(k_fwd_presplit, dim3((unsigned)((N + kPBM - 1) / kPBM)), dim3(512), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
