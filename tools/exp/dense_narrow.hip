// tools/exp/dense_narrow.hip -- EXPERIMENT, not built into the library (r03): a resident-reduction forward block for the
// narrow layers.  Bit-identical to dc_tag_linear_fwd_h2 with exact row maxima (tested on the GPU when it was wired in);
// measured 16.6 us (x tile through LDS, one workgroup per CU) / 18.1 us (x fragments straight from global memory,
// 2 - 3 workgroups per CU) against 26.7 us for the six-product split the product uses and 19.7 us for the generic
// fp16x2 entry: a gain of ~15 us per step for a sixth dense implementation and a change of the layer-1 arithmetic -
// not adopted.
// dc_dense_narrow.hip -- the forward dense block of the NARROW layers (the encoder's first layer: one reduction
// segment over the zero-padded concatenated hop slab, K = (K_hops + 1) * F_in padded to 96 / 112, Fo = 256), gfx950.
//
//   out[N, Fo] = act(x[N, K] . W[Fo, K]^T + b),   16 <= K <= 128, K % 16 == 0, Fo % 128 == 0
//   (/root/reference/models/model.py:69-78: the first TAGConv of either branch, F_in = 21 / 25)
//
// The generic kernels stream K in 16-wide stages through a double-buffered pipeline; with 6 - 8 stages in all, that
// is prologue and epilogue and nothing in between: 26.7 us (soft, six-product bf16 split) against a floor of ~8 us for
// writing the 33.5 MB of output (tools/exp/layer1_dense.py).  Here the whole reduction of a 128 x 128 output tile is
// resident: the x rows are staged ONCE (their maxima taken on the way - whole rows are in the tile, so the fp16x2
// scheme needs no row-maximum pass), the weights arrive as the pre-split image of dc_tag_narrow_weight_prep, one barrier,
// 6 - 8 k-steps of MFMAs straight out of LDS, store.  Arithmetic = dc_tag_linear_fwd_h2's (scaled fp16 pairs,
// l*h + h*l + h*h, k ascending in steps of 16): results are bit-identical to that entry given the exact row maxima.
#include "../../deformcontact_amd/csrc/dc_dense.h"

namespace dc {

using nr_f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using nr_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using nr_f32x4 = __attribute__((ext_vector_type(4))) float;
using nr_u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kNrBM = 128, kNrBN = 128;

struct NarrowParams {
    const float *x;          // [N, ldx] (the hop slab; columns >= K are never read)
    int64_t ldx;
    const char *wimg;        // [Fo, K] fp16x2 image (64-byte records), rows scaled by w_rowmax
    const float *w_rowmax;   // [Fo]
    const float *bias;       // [Fo] or null
    float *out;              // [N, ldo]
    int64_t ldo, N, Fo;
    int relu;
    float *x_rowmax;         // [N] or null: max |x[i, 0:K]| written out (the layer's dW scales with it)
};

// LDS holds the weight tile only: rows of KS records of 64 bytes + 16 bytes of padding (row stride = odd multiple of 16
// bytes: the 16 rows of a ds_read_b128 lane group fall into 16 distinct 16-byte slots of the 256-byte bank row).  The x
// rows go from global memory straight into A-operand fragments: lane (row lane & 31, half lane >> 5) reads the 8 floats
// 16 ks + 8 half of its row for every k-step, takes the row maximum with its partner lane, scales and splits in
// registers.  51 - 68 KB of LDS per workgroup: two or three workgroups per CU, so the loads of one overlap the stores
// of another (with the x tile staged through LDS as well - one workgroup per CU, everything in lock step - the block
// took 16.6 instead of 26.7 us; this form: see tools/exp/layer1_dense.py).
template <int KS>
__global__ void __launch_bounds__(512)
k_fwd_narrow(NarrowParams p) {
    constexpr int RS = KS * 64 + 16;
    __shared__ __attribute__((aligned(16))) char sB[kNrBN * RS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * kNrBM, col0 = (int64_t)blockIdx.y * kNrBN;
    const int wm = wid >> 1, wn = wid & 1, fr = lane & 31, fh = lane >> 5;     // 8 waves: 4 (rows) x 2 (columns)

    // ---- weight tile: rows col0 .. col0 + 127 of the image, KS * 64 contiguous bytes each
    {
        const char *src = p.wimg + col0 * (KS * 64);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int pi = j * 512 + tid, row = pi / (KS * 4), pc = pi - row * (KS * 4);
            *reinterpret_cast<nr_u32x4 *>(sB + row * RS + 16 * pc) =
                *reinterpret_cast<const nr_u32x4 *>(src + (int64_t)pi * 16);
        }
    }
    // ---- this lane's x row as fragments
    int64_t row = row0 + 32 * wm + fr;
    const bool rok = row < p.N;
    row = rok ? row : p.N - 1;
    nr_f16x8 ah[KS], al[KS];
    float inv;
    {
        const float *xp = p.x + row * p.ldx + 8 * fh;
        nr_f32x4 v[KS][2];
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            v[j][0] = *reinterpret_cast<const nr_f32x4 *>(xp + 16 * j);
            v[j][1] = *reinterpret_cast<const nr_f32x4 *>(xp + 16 * j + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fabsf(v[j][0][i]), fabsf(v[j][1][i])));
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        const float sc = h2_scale(m);
        inv = h2_unscale(m);
        if (p.x_rowmax && rok && fh == 0 && wn == 0 && blockIdx.y == 0) p.x_rowmax[row] = m;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const nr_f32x4 s = v[j][e] * sc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 a = (_Float16)s[i];
                    ah[j][4 * e + i] = a;
                    al[j][4 * e + i] = (_Float16)(s[i] - (float)a);
                }
            }
    }
    __syncthreads();

    const char *pb = sB + (64 * wn + fr) * RS + 16 * fh;
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        nr_f16x8 bh[2], bl[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            bh[nb] = *reinterpret_cast<const nr_f16x8 *>(pb + nb * 32 * RS + ks * 64);
            bl[nb] = *reinterpret_cast<const nr_f16x8 *>(pb + nb * 32 * RS + ks * 64 + 32);
        }
        // as k_fwd_h2 / k_fwd_h2w: x_l w_h, x_h w_l, x_h w_h
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[nb], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[nb], acc[nb], 0, 0, 0);
    }

    // ---- epilogue: C/D fragment (reg, lane) -> row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col lane & 31; the row's
    // unscale factor sits in the lane that holds the row as an A fragment (lane index = row within the 32)
    const bool relu = p.relu != 0;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t col = col0 + 64 * wn + 32 * nb + fr;
        const float bcol = p.bias ? p.bias[col] : 0.f;
        const float icol = h2_unscale(p.w_rowmax[col]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 8 * g + 4 * fh + i;                       // row within the wave's 32
                const float si = __shfl(inv, rl);
                const int64_t orow = row0 + 32 * wm + rl;
                float v = (acc[nb][4 * g + i] * si) * icol;
                v += bcol;
                if (relu) v = fmaxf(v, 0.f);
                if (orow < p.N) p.out[orow * p.ldo + col] = v;
            }
    }
}

// dc_tag_weight_prep for the narrow layers: the K_hops + 1 weight blocks W_s [Fo, Fi] as ONE image over the
// concatenated reduction k = s * Fi + f, zero-padded to Kp (what dc_tag_pack_weights + dc_tag_weight_prep would
// produce in two launches), + the row maxima
struct NarrowPrepParams {
    const float *w[kMaxSeg];
    int nseg;
    int64_t Fo, Fi, Kp;
    float *w_rowmax;
    _Float16 *wimg;
};

__global__ void __launch_bounds__(256)
k_narrow_weight_prep(NarrowPrepParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= p.Fo) return;
    float m = 0.f;
    for (int s = 0; s < p.nseg; ++s) {
        const float *wr = p.w[s] + o * p.Fi;
        for (int64_t c = lane; c < p.Fi; c += 64) m = fmaxf(m, fabsf(wr[c]));
    }
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) m = fmaxf(m, __shfl_xor(m, q));
    if (lane == 0) p.w_rowmax[o] = m;
    const float sc = h2_scale(m);
    _Float16 *row = p.wimg + o * (2 * p.Kp);
    const int64_t K = p.nseg * p.Fi;
    for (int64_t k = lane; k < p.Kp; k += 64) {
        float v = 0.f;
        if (k < K) {
            const int s = (int)(k / p.Fi);
            v = p.w[s][o * p.Fi + (k - (int64_t)s * p.Fi)];
        }
        const float x = v * sc;
        const _Float16 h = (_Float16)x;
        _Float16 *rec = row + (k >> 4) * 32 + (k & 15);
        rec[0] = h;
        rec[16] = (_Float16)(x - (float)h);
    }
}

}  // namespace dc

using namespace dc;

extern "C" int dc_tag_narrow_weight_prep(const float *const *ws, int nseg, int64_t Fo, int64_t Fi, int64_t Kp,
                                         float *w_rowmax, void *w_image, dc_stream_t stream) {
    DC_REQUIRE(nseg >= 1 && nseg <= kMaxSeg && Fo >= 1 && Fi >= 1 && ws && w_rowmax && w_image,
               "dc_tag_narrow_weight_prep: bad arguments");
    DC_REQUIRE(Kp >= nseg * Fi && Kp % 16 == 0 && (((uintptr_t)w_image) & 15) == 0,
               "dc_tag_narrow_weight_prep: Kp must be a multiple of 16 >= nseg * Fi, the image 16-byte aligned");
    NarrowPrepParams p{};
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(ws[s], "dc_tag_narrow_weight_prep: null segment %d", s);
        p.w[s] = ws[s];
    }
    p.nseg = nseg, p.Fo = Fo, p.Fi = Fi, p.Kp = Kp, p.w_rowmax = w_rowmax, p.wimg = (_Float16 *)w_image;
    hipLaunchKernelGGL(k_narrow_weight_prep, dim3((unsigned)((Fo + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dc_tag_narrow_weight_prep");
}

extern "C" int dc_tag_linear_fwd_narrow(const float *x, int64_t ldx, const void *w_image, const float *w_rowmax,
                                        const float *bias, int relu, float *out, int64_t ldo, int64_t N, int64_t K,
                                        int64_t Fo, float *x_rowmax_out, dc_stream_t stream) {
    DC_REQUIRE(N >= 0 && K >= 16 && K <= 128 && K % 16 == 0 && Fo >= kNrBN && Fo % kNrBN == 0,
               "dc_tag_linear_fwd_narrow: K must be a multiple of 16 in [16, 128], Fo a multiple of %d", kNrBN);
    if (N == 0) return DC_OK;
    DC_REQUIRE(x && w_image && w_rowmax && out && ldx >= K && ldo >= Fo && ldx % 4 == 0 &&
                   (((uintptr_t)x) & 15) == 0 && (((uintptr_t)w_image) & 15) == 0,
               "dc_tag_linear_fwd_narrow: null pointer, short or misaligned rows");
    const int64_t tiles = (N + kNrBM - 1) / kNrBM;
    DC_REQUIRE(tiles < (int64_t)INT32_MAX && Fo / kNrBN < 65536, "dc_tag_linear_fwd_narrow: grid too large");
    NarrowParams p{x, ldx, (const char *)w_image, w_rowmax, bias, out, ldo, N, Fo, relu, x_rowmax_out};
    const dim3 gd((unsigned)tiles, (unsigned)(Fo / kNrBN)), bd(512);
    hipStream_t hs = (hipStream_t)stream;
    switch (K / 16) {
    case 1: hipLaunchKernelGGL((k_fwd_narrow<1>), gd, bd, 0, hs, p); break;
    case 2: hipLaunchKernelGGL((k_fwd_narrow<2>), gd, bd, 0, hs, p); break;
    case 3: hipLaunchKernelGGL((k_fwd_narrow<3>), gd, bd, 0, hs, p); break;
    case 4: hipLaunchKernelGGL((k_fwd_narrow<4>), gd, bd, 0, hs, p); break;
    case 5: hipLaunchKernelGGL((k_fwd_narrow<5>), gd, bd, 0, hs, p); break;
    case 6: hipLaunchKernelGGL((k_fwd_narrow<6>), gd, bd, 0, hs, p); break;
    case 7: hipLaunchKernelGGL((k_fwd_narrow<7>), gd, bd, 0, hs, p); break;
    default: hipLaunchKernelGGL((k_fwd_narrow<8>), gd, bd, 0, hs, p); break;
    }
    return check_launch("dc_tag_linear_fwd_narrow");
}
