// tools/exp/dw_h2d_kernel.hip -- EXPERIMENT RECORD (round 4, last day), not compiled into the library.
// k_dw_h2w (the fp16x2 dW block of the wide layers) with both operands by LDS-DMA and the scale + split done in the MFMA
// waves on fragments gathered from the raw fp32 image.  It was built inside dc_dense_split.hip (it uses that file's
// DwParams / SplitFrag / mma_split / h2_scale / for_each_acc / kDwK), launched from dw_h2w_launch with 768 threads under
// DC_DW_H2_DMA=1, and checked by tests/test_wide_dense.py: weight AND bias partials bit-identical to k_dw_h2w and to
// k_dw_split<2, false, 2>.  133 VGPRs, no scratch, 147.5 KB LDS.  Measured in the step (bench.py roofline_mfma,
// in-sequence, same box): dW + slab reduce 76.9 / 61.5 us (soft / rigid) with k_dw_h2w, 89.7 / 71.9 us with this kernel;
// headline 525.0 -> 509.3 M edges/s.  Unlike its bf16 sibling (k_dw_bf16 -> k_dw_bf16d: 212 -> 86 us) the fp16x2 block
// is not waiting on its register-staged prefetch: ~350 VALU instructions per wave and stage for the gathered split
// (every value split by four (g) / two (x) waves instead of once by its loading thread) cost more than the loads hid.
// DESIGN.md section 8 has the conclusion.

// ---- k_dw_h2w with both operands by LDS-DMA (r04, last day; OPT-IN: DC_DW_H2_DMA=1) ---------------------------------
// k_dw_h2w's loop is k_dw_bf16's (dc_dense_bf16.hip): two register sets, one __syncthreads() per 32-node stage, and in
// the step it takes 1.9 us per stage whatever the stage computes (60 us for 32 stages; 0.37 us of MFMA work each) -
// the signature the bf16 kernel lost when loading waves took over its operands (212 -> 86 us).  The operands here are
// fp32 and are scaled and split on the way: the DMA'd image is RAW fp32 [node][column] (512 B / 1024 B rows, dense, no
// swizzle needed: a fragment read is 32 consecutive columns of one node), the eight MFMA waves gather their k-major
// fragments from it - 8 x 4-byte LDS reads per fragment in place of one transposing 8-byte read - and scale + split
// them in registers (same arithmetic per value as k_dw_h2w's store path, same products in the same order: bit-identical
// partials), four loading waves keep two 48 KiB stages in flight behind the one being multiplied.  Not for the
// corrected operand of the attention backward (k_dw_h2w<true>).  Off by default until a whole-suite run has seen it.
constexpr int kDwdSlots = 3, kDwdAhead = 2;
constexpr int kDwdRowG = 128 * 4, kDwdRowX = 256 * 4;                    // raw fp32 node rows
constexpr int kDwdOffX = kDwK * kDwdRowG;                                // 16 KiB of g, then 32 KiB of x
constexpr int kDwdStage = kDwK * (kDwdRowG + kDwdRowX);                  // 49,152 B
#define DC_DWDH_WAITVM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))   // s_waitcnt vmcnt(n) only

__global__ void __launch_bounds__(768)
k_dw_h2d(DwParams p) {
    __shared__ __attribute__((aligned(16))) char lds[kDwdSlots * kDwdStage];
    __shared__ float smax[24];
    const unsigned nto = (unsigned)(p.Fo / 128);
    const unsigned per_chunk = nto * (unsigned)p.nseg;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
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
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool do_bias = p.bias_partial && s == 0;
    const int nst = (int)((n_end - n_beg) / kDwK);
    const int64_t rows = n_end - n_beg, ldg = p.g.ld, ldx = p.x[s].ld;

    // the loading waves start their first stages before anything else happens
    __amdgpu_buffer_rsrc_t rg_, rx_;
    int voffg = 0, voffx = 0, gstep = 0, xstep = 0, gstage = 0, xstage = 0;
    const int w = wid - 8;
    auto stage = [&](int st) {                             // 12 instructions per wave: g pieces 4 w .. 4 w + 3 (two node
        char *dst = lds + (st % kDwdSlots) * kDwdStage;    // rows each), x rows 8 w .. 8 w + 7
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * w + j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg_, (void __attribute__((address_space(3))) *)(dst + c * 1024), 16,
                                                     voffg, c * gstep + st * gstage, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * w + j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (void __attribute__((address_space(3))) *)(dst + kDwdOffX + c * 1024),
                                                     16, voffx, c * xstep + st * xstage, 0, 0);
        }
    };
    if (wid >= 8 && rows > 0) {
        rg_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.g.p + n_beg * ldg + o0), 0,
                                                (int)(((rows - 1) * ldg + 128) * 4), 0x00020000);
        rx_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x[s].p + n_beg * ldx), 0,
                                                (int)(((rows - 1) * ldx + 256) * 4), 0x00020000);
        voffg = (int)((int64_t)(lane >> 5) * ldg * 4) + 16 * (lane & 31);
        voffx = 16 * lane;
        gstep = (int)(2 * ldg * 4), xstep = (int)(ldx * 4);
        gstage = (int)(kDwK * ldg * 4), xstage = (int)(kDwK * ldx * 4);
#pragma unroll
        for (int st = 0; st < kDwdAhead; ++st)
            if (st < nst) stage(st);
    }

    // one power-of-two scale per operand and chunk: maxima of the row maxima over the chunk's nodes (all 12 waves)
    float am = 0.f, bm = 0.f;
    for (int64_t i = n_beg + threadIdx.x; i < n_end; i += 768) {
        am = fmaxf(am, p.h2.a_rowmax[i]);
        bm = fmaxf(bm, p.h2.b_rowmax[i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        am = fmaxf(am, __shfl_xor(am, o));
        bm = fmaxf(bm, __shfl_xor(bm, o));
    }
    if (lane == 0) smax[wid] = am, smax[12 + wid] = bm;
    __syncthreads();
    am = bm = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) am = fmaxf(am, smax[i]), bm = fmaxf(bm, smax[12 + i]);
    const float sca = h2_scale(am), scb = h2_scale(bm), inva = h2_unscale(am), invb = h2_unscale(bm);

    if (wid >= 8) {
        // ------------------------------------------------------------------ loading waves
        for (int it = 0; it < nst; ++it) {
            if (nst - 1 - it >= 1) DC_DWDH_WAITVM(12);       // all but the newest stage's 12 instructions have landed
            else DC_DWDH_WAITVM(0);
            __builtin_amdgcn_s_barrier();                    // publishes stage it; the MFMA waves have left stage it - 1
            if (it + kDwdAhead < nst) stage(it + kDwdAhead);
        }
        if (do_bias) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ------------------------------------------------------------------ eight MFMA waves (2 x 4), 64 (o) x 64 (f) each
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int c4g = threadIdx.x & 31, krg = threadIdx.x >> 5;               // bias sums: k_dw_h2w's staging map
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc[2][2];
    zero_acc<2>(acc);
    // fragment of 8 consecutive nodes (k) of one column: gathered, scaled, split into the hi / lo fp16 planes
    auto frag = [&](const char *col, int rowb, float sc, bf16x8 &hi, bf16x8 &lo) {
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = *reinterpret_cast<const float *>(col + j * rowb) * sc;
            const _Float16 a = (_Float16)v;
            h[j] = a;
            l[j] = (_Float16)(v - (float)a);
        }
        hi = __builtin_bit_cast(bf16x8, h);
        lo = __builtin_bit_cast(bf16x8, l);
    };
    for (int it = 0; it < nst; ++it) {
        __builtin_amdgcn_s_barrier();
        const char *buf = lds + (it % kDwdSlots) * kDwdStage;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            SplitFrag<2, 2> f;
            const int k0 = ks * 16 + 8 * fh;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                frag(buf + k0 * kDwdRowG + (wm * 64 + mb * 32 + fr) * 4, kDwdRowG, sca, f.a[mb][0], f.a[mb][1]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                frag(buf + kDwdOffX + k0 * kDwdRowX + (wn * 64 + nb * 32 + fr) * 4, kDwdRowX, scb, f.b[nb][0], f.b[nb][1]);
            mma_split<2, 2>(f, acc);
        }
        if (do_bias) {
            bsum += *reinterpret_cast<const f32x4 *>(buf + krg * kDwdRowG + 16 * c4g);
            bsum += *reinterpret_cast<const f32x4 *>(buf + (krg + 16) * kDwdRowG + 16 * c4g);
        }
    }

    float *out = p.partial + ((int64_t)chunk * p.nseg + s) * p.Fo * p.Fi;
    for_each_acc<2>(acc, wm, wn, [&](int r, int c, float v) {
        out[(o0 + r) * p.Fi + c] = (v * inva) * invb;
    });
    if (do_bias) {                                      // column sums of g over the chunk's nodes
        __builtin_amdgcn_s_barrier();                   // every wave has left the ring
        f32x4 *red = reinterpret_cast<f32x4 *>(lds);
        red[threadIdx.x] = bsum;
        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): the stores have reached LDS
        __builtin_amdgcn_s_barrier();
        if (threadIdx.x < 32) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < 16; ++g) t += red[g * 32 + threadIdx.x];
            float *bp = p.bias_partial + (int64_t)chunk * p.Fo + o0 + 4 * threadIdx.x;
            bp[0] = t[0], bp[1] = t[1], bp[2] = t[2], bp[3] = t[3];
        }
    }
}

