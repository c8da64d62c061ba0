// dc_dense_narrow.hip -- the dense block of a TAGConv layer with a SHORT reduction: the first layer of each encoder branch
// (/root/reference/models/model.py:44-50: TAGConv(21, 256) / TAGConv(25, 256), K = 3 -> reduction (K + 1) * F_in = 84 / 100,
// zero-padded to 96 / 112 in the hop slab).
//
//   out[N, 256] = act(slab[N, Kp] . Wcat[256, Kp]^T + b),   Wcat = [W_0 | W_1 | W_2 | W_3 | 0]  (lins[k].weight of PyG tag_conv.py)
//
// Round 6 measured what these layers cost the two-stream headline step by making them free (tools/r06/skip_probe.py): forward
// blocks 53 us, weight-gradient blocks 64 us, hops 36 us of a 617 us step - all of it on the critical path, none hidden by the
// second stream.  The generic split kernel (dc_dense_split.hip: k_fwd_split, 128 x 128 tiles, both operands split per tile and
// stage, 4-byte epilogue stores) took 25.7 / 20.7 us for 46 / 35 MB of compulsory traffic (0.22 of the HBM roofline) and
// needed a packing launch for Wcat in front of it.
//
// This kernel is built for the shape: the whole reduction is ONE stage.
//   * persistent 256-thread workgroups, one per CU; wave w owns output columns [64 w, 64 w + 64) and keeps ITS slice of the
//     weights - read from the K + 1 lins[k].weight matrices themselves (no packing launch: coalesced reads into an fp32 LDS
//     image, fragments picked out of it), split into the three bf16 planes once per launch - in registers for all row tiles
//     (24 * Kp / 16 VGPRs; one wave per SIMD, 512 registers each);
//   * per row tile (64 or 32 rows): fp32 rows global -> registers (issued one tile ahead) -> three bf16 planes -> LDS (rows
//     padded to Kp * 6 + 16 bytes: conflict-free ds_read_b128 fragment reads), one barrier, 6 products x Kp / 16 k-steps of
//     v_mfma_f32_32x32x16_bf16 per accumulator IN THE ORDER of k_fwd_split<., 6> (k-steps ascending, products smallest terms
//     first) - the outputs are bit-identical to it (tests/test_narrow_dense.py);
//   * epilogue through LDS: bias + ReLU in registers, accumulators into a [rows][256] fp32 image, one barrier, one 1-KiB row per
//     global_store_dwordx4 wave-instruction (the 4-byte stores of the accumulator layout cost 7.6 us per 33.5 MB).
// Bytes per launch (B = 32): 12.6 + 33.5 MB soft, 10.9 + 25.0 MB rigid; MFMA work 9.7 / 8.4 GFLOP: HBM-bound.
#include "dc_dense.h"

// timing-only ablations (tools/r06/narrow_abl.sh builds this file with -DDC_NARROW_ABL=<bits>; results are wrong by
// construction): 1 no MFMAs, 2 no weight image (fragments made up), 4 no row stores, 8 no row loads / plane image, 16 no
// accumulator staging
#ifndef DC_NARROW_ABL
#define DC_NARROW_ABL 0
#endif

namespace dc {

using nb_bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using nb_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using nb_f32x4 = __attribute__((ext_vector_type(4))) float;

struct NarrowParams {
    const float *x;                 // hop slab [N, ld], Kp valid (zero-padded) columns
    int64_t ld;
    const float *w[kMaxSeg];        // lins[k].weight [Fo, fi]
    int nseg, fi;
    const float *bias;
    float *out;
    int64_t ldo, N;
    int relu, ntiles;
};

constexpr int kNarrowFo = 256;

// this wave's LDS operations have completed, then the workgroup meets.  NOT __syncthreads(): its fence waits for vmcnt(0) -
// for the next tile's rows (issued one tile ahead on purpose) and for the previous tile's 64 KB of row stores - which put two
// memory round trips per tile on the critical path (30 us per launch instead of 12)
__device__ __forceinline__ void nb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void nb_split1(float x, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
    hi = (__bf16)x;
    const float r = x - (float)hi;        // exact
    mid = (__bf16)r;
    const float r2 = r - (float)mid;      // exact
    lo = (__bf16)r2;
}

// KS: k-steps of 16 (Kp = 16 KS); MB: 32-row blocks per tile
template <int KS, int MB>
__global__ void __launch_bounds__(256)
k_fwd_narrow(NarrowParams p) {
    constexpr int TR = 32 * MB;                         // rows per tile
    constexpr int SROWA = KS * 96 + 16;                 // bytes per row of the plane image: [ks][plane][half][8 bf16] + pad
    constexpr int PPR = 4 * KS;                         // float4 pieces per row
    constexpr int TP = TR * PPR, NV = (TP + 255) / 256;
    constexpr int KP = 16 * KS;                         // padded reduction
    constexpr int kOffStage = ((TR * SROWA + 1023) / 1024) * 1024;
    constexpr int kLdsTiles = kOffStage + TR * 1024, kLdsW = kNarrowFo * (KP + 4) * 4;      // tile images / prologue weight image
    __shared__ __attribute__((aligned(1024))) char lds[kLdsTiles > kLdsW ? kLdsTiles : kLdsW];
    char *const sA = lds;
    float *const so = reinterpret_cast<float *>(lds + kOffStage);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;

    // ---- x tile staging: piece q = threadIdx.x + 256 j of the tile = float4 c4 of row r ----
    nb_f32x4 xv[NV];
    auto load_tile = [&](int t) {
        const int64_t row0 = (int64_t)t * TR;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = (int)threadIdx.x + 256 * j;
            if (TP % 256 != 0 && q >= TP) continue;
            const int r = q / PPR, c4 = q - r * PPR;
            int64_t row = row0 + r;
            row = row < p.N ? row : p.N - 1;
            if (DC_NARROW_ABL & 8) xv[j] = nb_f32x4{(float)row, 1.f, 2.f, 3.f};
            else xv[j] = *reinterpret_cast<const nb_f32x4 *>(p.x + row * p.ld + 4 * c4);
        }
    };
    int t = blockIdx.x;
    if (t < p.ntiles) load_tile(t);                     // the first tile's rows travel while the weights are prepared

    // ---- this wave's slice of the weights: fragments of all k-steps, three planes, in registers for the whole launch ----
    // First Wcat = [W_0 | ... | W_{nseg-1} | 0] as an fp32 image [256][Kp] in LDS, gathered with COALESCED reads of the
    // lins[k].weight matrices (element e of segment s = row e / fi, column e % fi; a per-lane gather of the fragments straight
    // from memory - 8 strided 4-byte loads per fragment - cost 20 us per launch: every workgroup made 7 M cache-line requests);
    // then every lane picks its fragments out of the image: column 64 wid + 32 nb + c, k = 16 ks + 8 h .. + 7.
    constexpr int WROW = KP + 4;                        // image row in floats: +16 B keeps the fragment picks conflict-free
    {
        // ALL loads (every segment) are issued before the first is used: a load per loop trip waited for each one in turn
        // (84 round trips to L2, 25 us per launch); a batch per segment still made four round trips
        float *wi = reinterpret_cast<float *>(lds);
        const int width = p.nseg * p.fi, per4 = kNarrowFo * p.fi / 4;         // float4 pieces per segment (256 fi / 4)
        constexpr int MAXQ = 8;                                              // fi <= 32
        nb_f32x4 wv[kMaxSeg][MAXQ];
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s)
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const int q = (int)threadIdx.x + 256 * j;
                if (!(DC_NARROW_ABL & 2) && s < p.nseg && q < per4) wv[s][j] = *reinterpret_cast<const nb_f32x4 *>(p.w[s] + 4 * q);
            }
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s)
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const int q = (int)threadIdx.x + 256 * j;
                if (!(DC_NARROW_ABL & 2) && s < p.nseg && q < per4) {
                    int o = (4 * q) / p.fi, f = 4 * q - o * p.fi;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        wi[o * WROW + s * p.fi + f] = wv[s][j][i];
                        if (++f == p.fi) f = 0, ++o;
                    }
                }
            }
        const int padw = KP - width;
        for (int e = threadIdx.x; e < kNarrowFo * padw; e += 256) {
            const int o = e / padw, f = e - o * padw;
            wi[o * WROW + width + f] = 0.f;
        }
    }
    __syncthreads();
    nb_bf16x8 fb[2][KS][3];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float *wr = reinterpret_cast<const float *>(lds) + (64 * wid + 32 * nb + c) * WROW + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const nb_f32x4 v0 = *reinterpret_cast<const nb_f32x4 *>(wr + 16 * ks);
            const nb_f32x4 v1 = *reinterpret_cast<const nb_f32x4 *>(wr + 16 * ks + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 hi, mid, lo;
                nb_split1(j < 4 ? v0[j] : v1[j - 4], hi, mid, lo);
                fb[nb][ks][0][j] = hi, fb[nb][ks][1][j] = mid, fb[nb][ks][2][j] = lo;
            }
        }
    }
    __syncthreads();                                    // the image's LDS becomes the plane / staging images
    float bcol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bcol[nb] = p.bias ? p.bias[64 * wid + 32 * nb + c] : 0.f;
    const bool relu = p.relu != 0;

    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = (int)threadIdx.x + 256 * j;
            if (TP % 256 != 0 && q >= TP) continue;
            const int r = q / PPR, c4 = q - r * PPR;
            nb_bf16x4 hi, mid, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __bf16 a, b, d;
                nb_split1(xv[j][i], a, b, d);
                hi[i] = a, mid[i] = b, lo[i] = d;
            }
            // k = 4 c4 .. + 3: k-step c4 / 4, half (c4 & 3) / 2, position 4 (c4 & 1) inside the half's 8
            char *dst = sA + r * SROWA + (c4 >> 2) * 96 + ((c4 >> 1) & 1) * 16 + (c4 & 1) * 8;
            *reinterpret_cast<nb_bf16x4 *>(dst) = hi;
            *reinterpret_cast<nb_bf16x4 *>(dst + 32) = mid;
            *reinterpret_cast<nb_bf16x4 *>(dst + 64) = lo;
        }
    };

    // one 1-KiB row per store instruction: wave w stores rows w, w + 4, ... of the staged tile
    auto store_rows = [&](int64_t row0) {
#pragma unroll 4
        for (int r = wid; r < TR; r += 4) {
            const int64_t row = row0 + r;
            const nb_f32x4 v = *reinterpret_cast<const nb_f32x4 *>(so + r * 256 + 4 * lane);
            if (row < p.N && (!(DC_NARROW_ABL & 4) || v[0] == 12345.678f)) *reinterpret_cast<nb_f32x4 *>(p.out + row * p.ldo + 4 * lane) = v;
        }
    };
    // Order inside an iteration (vector-memory operations retire in order, and hipcc's wait before the first use of the staged
    // rows is the conservative merge over both loop entries, i.e. vmcnt(0)): USE the rows of tile t (loaded one iteration ago,
    // the youngest operations in flight) -> row stores of tile t - 1 -> loads of tile t + 1 -> MFMAs of tile t.  Stores and
    // loads then travel under the MFMAs, and the wait at the top of the next iteration finds them a whole MFMA phase old.
    int64_t prev_row0 = -1;
    for (; t < p.ntiles; t += gridDim.x) {
        const int64_t row0 = (int64_t)t * TR;
        if (!(DC_NARROW_ABL & 8)) store_tile();         // (every wave is past the MFMAs of the previous tile: barrier B)
        if (prev_row0 >= 0) store_rows(prev_row0);
        if (t + (int)gridDim.x < p.ntiles) load_tile(t + gridDim.x);
        nb_lds_barrier();                               // A: the plane image is complete; the staging image has been read
        f32x16 acc[MB][2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
        constexpr int HI = 0, MID = 1, LO = 2;
        constexpr int pa6[6] = {LO, HI, MID, MID, HI, HI}, pb6[6] = {HI, LO, MID, HI, MID, HI};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            nb_bf16x8 fa[MB][3];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fa[mb][pl] = *reinterpret_cast<const nb_bf16x8 *>(sA + (mb * 32 + c) * SROWA + ks * 96 + pl * 32 + h * 16);
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        if (!(DC_NARROW_ABL & 1))
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mb][pa6[tt]], fb[nb][ks][pb6[tt]], acc[mb][nb], 0, 0, 0);
                        else
                            acc[mb][nb][tt] += (float)fa[mb][pa6[tt]][0] * (float)fb[nb][ks][pb6[tt]][0];
        }
        // epilogue: bias + ReLU, accumulators -> [TR][256] image (C/D layout: row (reg & 3) + 8 (reg >> 2) + 4 h, column c)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mb][nb][r] + bcol[nb];
                    if (relu) v = fmaxf(v, 0.f);
                    if (!(DC_NARROW_ABL & 16) || r == 0)
                        so[(mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 256 + 64 * wid + 32 * nb + c] = v;
                }
        nb_lds_barrier();                               // B: image complete; every wave has read its plane fragments
        prev_row0 = row0;
    }
    if (prev_row0 >= 0) store_rows(prev_row0);
}

// 32-row tiles: four (soft) / three (rigid) per persistent workgroup at B = 32 - measured 23.1 / 20.8 us against 25.1 / 22.6 us with
// 64-row tiles (MB = 2: two / one and a half per workgroup; profiles/r06/e_narrow_time.txt)
template <int KS>
static void narrow_launch(const NarrowParams &p0, hipStream_t hs) {
    NarrowParams p = p0;
    p.ntiles = (int)((p.N + 31) / 32);
    const unsigned grid = (unsigned)(p.ntiles < 256 ? p.ntiles : 256);
    DC_LAUNCH((k_fwd_narrow<KS, 1>), dim3(grid), dim3(256), 0, hs, p);
}

}  // namespace dc

using namespace dc;

// Whether dc_tag_linear_fwd_narrow takes this shape (host-side query; the entry itself returns DC_EINVAL otherwise)
extern "C" int dc_tag_linear_fwd_narrow_ok(int64_t fi, int nseg, int64_t wpad, int64_t Fo) {
    return (Fo == kNarrowFo && nseg >= 1 && nseg <= kMaxSeg && fi >= 1 && fi <= 32 && nseg * fi <= wpad &&
            (wpad == 96 || wpad == 112 || wpad == 128)) ? 1 : 0;
}

extern "C" int dc_tag_linear_fwd_narrow(const float *slab, int64_t ld, const float *const *ws, int nseg, int64_t fi,
                                        const float *bias, int relu, float *out, int64_t ldo, int64_t N, int64_t wpad,
                                        int64_t Fo, dc_stream_t stream) {
    DC_REQUIRE(dc_tag_linear_fwd_narrow_ok(fi, nseg, wpad, Fo),
               "dc_tag_linear_fwd_narrow: needs Fo = 256, a padded reduction of 96 / 112 / 128 >= nseg * fi (got Fo=%lld wpad=%lld "
               "nseg=%d fi=%lld)", (long long)Fo, (long long)wpad, nseg, (long long)fi);
    DC_REQUIRE(N >= 0, "dc_tag_linear_fwd_narrow: negative N");
    if (N == 0) return DC_OK;
    DC_REQUIRE(slab && ws && out && ld >= wpad && ldo >= Fo, "dc_tag_linear_fwd_narrow: null pointer / short leading dimension");
    DC_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "dc_tag_linear_fwd_narrow: rows of the slab and of the output must be 16-byte aligned");
    DC_REQUIRE(N < ((int64_t)1 << 31) - 64, "dc_tag_linear_fwd_narrow: too many rows");
    NarrowParams p{};
    p.x = slab, p.ld = ld, p.nseg = nseg, p.fi = (int)fi, p.bias = bias, p.out = out, p.ldo = ldo, p.N = N, p.relu = relu;
    for (int s = 0; s < nseg; ++s) {
        DC_REQUIRE(ws[s], "dc_tag_linear_fwd_narrow: null weight segment %d", s);
        p.w[s] = ws[s];
    }
    hipStream_t hs = (hipStream_t)stream;
    switch (wpad) {
    case 96: narrow_launch<6>(p, hs); break;
    case 112: narrow_launch<7>(p, hs); break;
    default: narrow_launch<8>(p, hs); break;
    }
    return check_launch("dc_tag_linear_fwd_narrow");
}
