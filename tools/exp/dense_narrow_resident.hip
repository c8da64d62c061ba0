// tools/exp/dense_narrow_resident.hip -- round-5 experiment, NOT part of the library (fragment of dc_dense_split.hip: needs its
// Planes / SplitFrag / load_split_frag / mma_split / split4_h2 and dc_dense.h).  The first TAGConv layers' forward block (one
// concatenated segment of 96 / 112 columns) with the WHOLE reduction resident in LDS: 128 rows x all output columns per
// workgroup, fp16x2 with the row scales formed in the kernel.  Correct (4.2e-7 per row vs float64 at the layer's shapes) and
// NOT faster: ~25 us per launch against ~23 us for k_fwd_split<., 6>, headline 0.662 - 0.667 ms against 0.651 - 0.653 ms on the
// same box (profiles/r05/h_narrow_resident_forward.txt).  143 KB of LDS means one 4-wave workgroup per CU, and its phases -
// read x, split, read weights, split, 168 MFMAs per wave, store, weights again - run one after the other with nothing to hide
// them; k_fwd_split's 57 KB tiles run two to a CU, 512 of them, and overlap each other.
// ------------------------------- forward, narrow reduction (round 5) ------------------------
// The encoder's first TAGConv layers (models/model.py:44-50: 21 -> 256 and 25 -> 256, K = 3) run their dense block over ONE
// concatenated, zero-padded segment of 96 / 112 columns.  On k_fwd_split that is 6 - 7 stages of 16 columns with a barrier
// each, half cache lines per row and stage, weights and activations re-split by every column tile: 23 / 21 us for 1.4 GFLOP
// and 46 MB (8 us of memory traffic).  Here the WHOLE reduction is resident: one workgroup takes 128 rows and all output
// columns; its x tile is read once (whole 384 / 448-byte rows), scaled by the row's own maximum - the row is complete, so no
// row-maxima pass is needed - split into two fp16 planes (the fp16x2 arithmetic of the wide layers: three products, fp32-accurate)
// and kept in LDS as KS stages of k_fwd_split's row image; per 128 output columns the weight rows get the same treatment, then
// KS x 24 MFMAs per wave run without a barrier in between.
template <int KS>
__global__ void __launch_bounds__(256)
k_fwd_narrow(FwdParams p) {
    constexpr int SROW = Planes<2>::SROW, kStage = 128 * SROW;   // 80-byte rows, 10,240 bytes per 16-column stage
    extern __shared__ __attribute__((aligned(16))) char nlds[];
    char *As = nlds, *Bs = nlds + KS * kStage;
    float *s_inv = reinterpret_cast<float *>(Bs + KS * kStage), *s_icol = s_inv + 128;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int r = threadIdx.x >> 2, k4 = threadIdx.x & 3;
    const int wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1;

    auto stage_rows = [&](const float *base, int64_t ld, int64_t first, int64_t limit, char *dst, float *unscale) {
        float4 v[2][KS];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int64_t row = first + r + 64 * j;
            row = row < limit ? row : limit - 1;
            const float *src = base + row * ld + 4 * k4;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) v[j][ks] = *reinterpret_cast<const float4 *>(src + 16 * ks);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float m = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j][ks].x), fabsf(v[j][ks].y)), fmaxf(fabsf(v[j][ks].z), fabsf(v[j][ks].w))));
            m = fmaxf(m, __shfl_xor(m, 1));                      // the four lanes of a row are neighbours
            m = fmaxf(m, __shfl_xor(m, 2));
            const float sc = h2_scale(m);
            if (k4 == 0) unscale[r + 64 * j] = h2_unscale(m);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float4 x = make_float4(v[j][ks].x * sc, v[j][ks].y * sc, v[j][ks].z * sc, v[j][ks].w * sc);
                f16x4 h, l;
                split4_h2(x, h, l);
                char *at = dst + ks * kStage + (r + 64 * j) * SROW + 8 * k4;
                *reinterpret_cast<f16x4 *>(at) = h;
                *reinterpret_cast<f16x4 *>(at + kPlane) = l;
            }
        }
    };

    stage_rows(p.x[0].p, p.x[0].ld, row0, p.N, As, s_inv);
    const bool relu = p.relu != 0;
    for (int64_t col0 = 0; col0 < p.Fo; col0 += 128) {
        stage_rows(p.w[0].p, p.Fi, col0, p.Fo, Bs, s_icol);
        __syncthreads();
        f32x16 acc[2][2];
        zero_acc<2>(acc);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            SplitFrag<2, 2> f;
            load_split_frag<2, 2>(f, As + ks * kStage, Bs + ks * kStage, wm, wn);
            mma_split<2, 2>(f, acc);
        }
        float bcol[2], icol[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int c = wn * 64 + nb * 32 + (threadIdx.x & 31);
            bcol[nb] = (p.bias && col0 + c < p.Fo) ? p.bias[col0 + c] : 0.f;
            icol[nb] = s_icol[c];
        }
        for_each_acc<2>(acc, wm, wn, [&](int rr, int c, float v) {
            const int64_t row = row0 + rr, col = col0 + c;
            if (row < p.N && col < p.Fo) {
                v = (v * s_inv[rr]) * icol[(c >> 5) & 1];
                v += bcol[(c >> 5) & 1];
                if (relu) v = fmaxf(v, 0.f);
                p.out[row * p.ldo + col] = v;
            }
        });
        __syncthreads();                                         // the next 128 columns overwrite the weight image
    }
}

template <int KS>
static bool fwd_narrow_launch_ks(const FwdParams &p, hipStream_t hs) {
    constexpr size_t lds = (size_t)2 * KS * 128 * Planes<2>::SROW + 256 * sizeof(float);
    static_assert(lds <= 160 * 1024, "k_fwd_narrow: the resident reduction does not fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd_narrow<KS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return false;
        attr_set = true;
    }
    DC_LAUNCH((k_fwd_narrow<KS>), dim3((unsigned)((p.N + 127) / 128)), dim3(256), lds, hs, p);
    return true;
}

// one segment, 16 <= Fi <= 112 a multiple of 16, 16-byte aligned operands; false = not eligible
bool fwd_narrow_launch(const FwdParams &p, hipStream_t hs) {
    if (p.nseg != 1 || p.Fi % 16 != 0 || p.Fi < 16 || p.Fi > 112 || !al16(p.x[0].p) || !al16(p.w[0].p) || p.x[0].ld % 4 != 0)
        return false;
    switch ((int)(p.Fi / 16)) {
    case 1: return fwd_narrow_launch_ks<1>(p, hs);
    case 2: return fwd_narrow_launch_ks<2>(p, hs);
    case 3: return fwd_narrow_launch_ks<3>(p, hs);
    case 4: return fwd_narrow_launch_ks<4>(p, hs);
    case 5: return fwd_narrow_launch_ks<5>(p, hs);
    case 6: return fwd_narrow_launch_ks<6>(p, hs);
    default: return fwd_narrow_launch_ks<7>(p, hs);
    }
}

