// dc_attn_flash.hip -- flash-style forward of the reference's cross-attention head (gfx950).
//
//   O[i,:] = sum_j softmax_j(q_i . k_j) v_j,  lse[i] = log sum_j exp(q_i . k_j)
//   (/root/reference/models/model.py:13-21: per head softmax(head(x_soft) head(x_rigid)^T, dim=-1) x_rigid,
//    unmasked over the whole batch, no 1/sqrt(d)); d = dv = 256 (the shipped hidden width).
//
// The blocked form (attention.py) writes a [2048, N_r] score block (200 MB at batch 32), normalises it in place and
// reads it back for the weights x values product: three launches and ~1 GB of traffic per block and head.  Here the
// scores of a 128-query tile never leave the CU: one workgroup (4 waves, one per SIMD, 32 queries each) streams
// the keys in tiles of 32 and keeps a running max / sum / output per query (online softmax).
//
// Arithmetic = the blocked form's: both products on the fp16x2 scheme of dc_dense_h2w.hip (operands as scaled
// fp16 pairs h + l, products l*h + h*l + h*h in that order, fp32 accumulate, k ascending in steps of 16), the keys
// and V^T handed over as the pre-split images of dc_tag_weight_prep, q and the softmax weights split in
// registers - so the score of (i, j) is BIT-IDENTICAL to the one dc_tag_linear_fwd_h2p(_exp) computes, and the
// backward's recompute exp(s - lse) sees the same s the forward normalised.
//
// Layout trick (no transposes anywhere): the score tile is computed TRANSPOSED, S^T = K_tile Q^T (keys = MFMA rows,
// queries = MFMA columns).  In the 32x32 accumulator layout a lane then holds 16 keys of ONE query (column =
// lane & 31): max and sum over the keys are 16 in-lane operations + one exchange between the lane halves, the
// running statistics are one scalar per lane, and the weights a lane holds - keys {4h..4h+3, 8+4h..11+4h} (+16) of
// its query, h = lane >> 5 - are exactly a B-operand fragment of O^T += V^T P^T (contraction over the keys) once
// the A operand (the V^T image) lists its keys in the same order; the staging pass writes the V^T tile into LDS
// with the 4-key chunks of every 16 keys in the order (0, 2 | 1, 3) for that.
//
// Per 32-key tile and wave: 48 MFMAs for S^T (16 k-steps x 3 products), 48 for O^T (8 row tiles of V^T x 2 k-steps x
// 3 products), 64 ds_read_b128 fragment reads; q fragments (128 VGPRs) and the O^T accumulators (128) stay in
// registers for the whole kernel (512 registers per lane at one wave per SIMD).  K / V^T tiles are staged through
// registers into double-buffered LDS (2 x (32 + 32) KB), the loads of tile t+1 issued before the products of tile t.
#include "dc_dense.h"

namespace dc {

using fl_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using fl_f32x4 = __attribute__((ext_vector_type(4))) float;
using fl_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using fl_u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int kFlD = DC_ATTN_FLASH_D;          // d = dv
constexpr int kFlQ = 128;                      // queries per workgroup (4 waves x 32)
constexpr int kFlT = 32;                       // keys per tile
constexpr int kFlKRow = kFlD * 4;              // bytes per key row of the K image (16 records x 64 B)
constexpr int kFlKSz = kFlT * kFlKRow;         // 32 KB
constexpr int kFlVRow = kFlT * 4;              // bytes per V^T row and tile (2 records x 64 B)
constexpr int kFlVSz = kFlD * kFlVRow;         // 32 KB
constexpr int kFlORow = kFlD + 4;              // floats per query row of the epilogue transpose buffer
constexpr int kFlSmem = 4 * 32 * kFlORow * 4;  // 133,120 B >= 2 * (kFlKSz + kFlVSz) + 256

struct FlashParams {
    const float *q;        // [ns, ldq] fp32
    int64_t ldq;
    const float *qmax;     // [ns] row maxima of |q|
    const char *kimg;      // [nrp, d] fp16x2 image of the keys (rows scaled by kmax)
    const float *kmax;     // [nrp]
    const char *vtimg;     // [dv, nrp] fp16x2 image of V^T (rows scaled by vtmax)
    const float *vtmax;    // [dv]
    int64_t ns, nr, nrp;
    float *o;              // [ns, ldo]
    int64_t ldo;
    float *lse;            // [ns]
};

__device__ __forceinline__ int fl_vswz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }   // as hw_swz

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_attn_flash_fwd(FlashParams p) {
    __shared__ __attribute__((aligned(16))) char smem[kFlSmem];
    char *sK = smem, *sV = smem + 2 * kFlKSz;
    float *sUk = reinterpret_cast<float *>(smem + 2 * kFlKSz + 2 * kFlVSz);       // 2 x 32 key unscale factors
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int64_t q0 = (int64_t)blockIdx.x * kFlQ + wid * 32;
    int64_t qrow = q0 + fr;
    qrow = qrow < p.ns ? qrow : p.ns - 1;

    // ---- this lane's query as B-operand fragments: k = 16 ks + 8 fh + 0..7, scaled and split as dc_dense_h2w does
    const float qm = p.qmax[qrow];
    const float sq = h2_scale(qm), uq = h2_unscale(qm);
    fl_f16x8 qh[16], ql[16];
    {
        const float *qp = p.q + qrow * p.ldq + 8 * fh;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const fl_f32x4 a = *reinterpret_cast<const fl_f32x4 *>(qp + 16 * ks) * sq;
            const fl_f32x4 b = *reinterpret_cast<const fl_f32x4 *>(qp + 16 * ks + 4) * sq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 ha = (_Float16)a[i], hb = (_Float16)b[i];
                qh[ks][i] = ha, qh[ks][4 + i] = hb;
                ql[ks][i] = (_Float16)(a[i] - (float)ha), ql[ks][4 + i] = (_Float16)(b[i] - (float)hb);
            }
        }
    }

    // ---- staging: K tile = 32 consecutive 1 KB rows of the image (piece pc of row r at pc ^ (r & 15): each 16-lane
    // group of a ds_read_b128 fragment read sees 16 distinct 16-byte slots of the 256-byte bank row)
    const int nt = (int)((p.nr + kFlT - 1) / kFlT);
    fl_u32x4 rg[8];                                   // one staging register set, used for K then for V^T
    float ruk = 0.f;
    auto gloadK = [&](int t) {
        const char *kb = p.kimg + (int64_t)t * kFlKSz;
#pragma unroll
        for (int j = 0; j < 8; ++j) rg[j] = *reinterpret_cast<const fl_u32x4 *>(kb + (j * 256 + tid) * 16);
        if (tid < 32) ruk = h2_unscale(p.kmax[(int64_t)t * kFlT + tid]);
    };
    // piece j * 256 + tid of the tile: row 4 j + wid, piece tid & 63 -> slot piece ^ (row & 15); row & 15 = 4 (j & 3) + wid,
    // so four lane offsets (j & 3) + an immediate (j >> 2) * 16 KB address all eight stores
    int kst[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) kst[jj] = (4 * jj + wid) * kFlKRow + 16 * ((tid & 63) ^ (4 * jj + wid));
    auto lstoreK = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *reinterpret_cast<fl_u32x4 *>(sK + buf * kFlKSz + (j >> 2) * 16 * kFlKRow + kst[j & 3]) = rg[j];
        if (tid < 32) sUk[buf * 32 + tid] = ruk;
    };
    // V^T tile: 8 threads per 128-byte row piece, 32 rows per pass; 32-bit lane offsets off a wave-uniform base
    unsigned vgo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vgo[j] = (unsigned)(((tid >> 3) + 32 * j) * (p.nrp * 4) + (tid & 7) * 16);
    auto gloadV = [&](int t) {
        const char *vb = p.vtimg + (int64_t)t * kFlVRow;
#pragma unroll
        for (int j = 0; j < 8; ++j) rg[j] = *reinterpret_cast<const fl_u32x4 *>(vb + vgo[j]);
    };
    auto lstoreV = [&](int buf) {
        // the 16 bytes a thread holds are chunks (2 cp, 2 cp + 1) of one plane of one record (4 keys each); chunk c goes
        // to half (c & 1), 8-byte slot (c >> 1): half h of a plane then lists keys {4h..4h+3, 8+4h..11+4h}
        const int k8 = tid & 7, m = k8 >> 2, pl = (k8 >> 1) & 1, cp = k8 & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (tid >> 3) + 32 * j, f = fl_vswz(row);
            char *base = sV + buf * kFlVSz + row * kFlVRow + 8 * cp;
            const int qp0 = 4 * m + 2 * pl;
            *reinterpret_cast<fl_u32x2 *>(base + 16 * (qp0 ^ f)) = fl_u32x2{rg[j][0], rg[j][1]};
            *reinterpret_cast<fl_u32x2 *>(base + 16 * ((qp0 + 1) ^ f)) = fl_u32x2{rg[j][2], rg[j][3]};
        }
    };

    // fragment read offsets: the swizzle only touches the low 4 bits of a piece index, so 8 (K: k-step & 3, plane) and
    // 4 (V^T: k-step, plane) lane offsets + immediates address every read (the rows of V^T tile mt are 32 mt + fr and
    // its swizzle only depends on the low 4 row bits)
    int kfo[8], vfo[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) kfo[c] = fr * kFlKRow + 16 * ((4 * (c >> 1) + 2 * (c & 1) + fh) ^ (fr & 15));
#pragma unroll
    for (int c = 0; c < 4; ++c) vfo[c] = fr * kFlVRow + 16 * ((4 * (c >> 1) + 2 * (c & 1) + fh) ^ fl_vswz(fr));
    f32x16 oacc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[mt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float kLog2e = 1.4426950408889634f;

    gloadK(0);
    lstoreK(0);
    gloadV(0);
    lstoreV(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gloadK(t + 1);
        // ---- S^T tile: rows = the tile's 32 keys, columns = this wave's 32 queries
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
        {
            // fragments of two k-steps are read while the six MFMAs of the previous two run (sched_barrier: hipcc
            // otherwise hoists all 32 reads of the tile to the top - 128 registers - and spills)
            const char *kb = sK + buf * kFlKSz;
            fl_f16x8 kh0[2], kl0[2], kh1[2], kl1[2];
            auto kfrags = [&](fl_f16x8 (&kh)[2], fl_f16x8 (&kl)[2], int g) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ks = 2 * g + u;
                    kh[u] = *reinterpret_cast<const fl_f16x8 *>(kb + (ks >> 2) * 256 + kfo[2 * (ks & 3)]);
                    kl[u] = *reinterpret_cast<const fl_f16x8 *>(kb + (ks >> 2) * 256 + kfo[2 * (ks & 3) + 1]);
                }
            };
            auto kmma = [&](const fl_f16x8 (&kh)[2], const fl_f16x8 (&kl)[2], int g) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ks = 2 * g + u;
                    // same term order as k_fwd_h2w with x = q (A there) and W = k (B there): x_l w_h, x_h w_l, x_h w_h
                    st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[u], ql[ks], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[u], qh[ks], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[u], qh[ks], st, 0, 0, 0);
                }
            };
            kfrags(kh0, kl0, 0);
#pragma unroll
            for (int g = 0; g < 8; g += 2) {
                kfrags(kh1, kl1, g + 1);
                kmma(kh0, kl0, g);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < 8) kfrags(kh0, kl0, g + 2);
                kmma(kh1, kl1, g + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t + 1 < nt) {                       // (the other K buffer was last read before the previous barrier)
            lstoreK(buf ^ 1);
            gloadV(t + 1);
        }
        // ---- online softmax over the keys of this lane's query: reg i <-> key (i & 3) + 8 (i >> 2) + 4 fh
        float s[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const fl_f32x4 uk = *reinterpret_cast<const fl_f32x4 *>(sUk + buf * 32 + 8 * g + 4 * fh);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[4 * g + i] = (st[4 * g + i] * uq) * uk[i];     // as the h2w epilogue: (acc * s_row) * s_col
        }
        if (t == nt - 1) {
            const int64_t kbase = (int64_t)t * kFlT + 4 * fh;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kbase + (i & 3) + 8 * (i >> 2) >= p.nr) s[i] = -INFINITY;
        }
        float tm = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tm = fmaxf(tm, s[i]);
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        const float mn = fmaxf(m_run, tm);
        if (__builtin_amdgcn_ballot_w64(mn > m_run) != 0) {        // rare once the running maxima have settled
            const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * kLog2e);    // 1 exactly where the max stays
            l_run *= alpha;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[mt][i] *= alpha;
            m_run = mn;
        }
        float ps = 0.f;
        fl_f16x8 ph[2], pl[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f((s[i] - m_run) * kLog2e);
            ps += e;
            const _Float16 h = (_Float16)e;
            ph[i >> 3][i & 7] = h;
            pl[i >> 3][i & 7] = (_Float16)(e - (float)h);
        }
        ps += __shfl_xor(ps, 32);
        l_run += ps;
        // ---- O^T += V^T_tile P^T: rows = dv (8 tiles of 32), contraction over the tile's keys (2 k-steps of 16)
        {
            const char *vb = sV + buf * kFlVSz;
            fl_f16x8 vh0[2], vl0[2], vh1[2], vl1[2];
            auto vfrags = [&](fl_f16x8 (&vh)[2], fl_f16x8 (&vl)[2], int mt) {
                const char *vr = vb + mt * 32 * kFlVRow;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    vh[m] = *reinterpret_cast<const fl_f16x8 *>(vr + vfo[2 * m]);
                    vl[m] = *reinterpret_cast<const fl_f16x8 *>(vr + vfo[2 * m + 1]);
                }
            };
            auto vmma = [&](const fl_f16x8 (&vh)[2], const fl_f16x8 (&vl)[2], int mt) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    // blocked form: x = P (A there), W = V^T (B there): x_l w_h, x_h w_l, x_h w_h
                    oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[m], pl[m], oacc[mt], 0, 0, 0);
                    oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[m], ph[m], oacc[mt], 0, 0, 0);
                    oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[m], ph[m], oacc[mt], 0, 0, 0);
                }
            };
            vfrags(vh0, vl0, 0);
#pragma unroll
            for (int mt = 0; mt < 8; mt += 2) {
                vfrags(vh1, vl1, mt + 1);
                vmma(vh0, vl0, mt);
                __builtin_amdgcn_sched_barrier(0);
                if (mt + 2 < 8) vfrags(vh0, vl0, mt + 2);
                vmma(vh1, vl1, mt + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t + 1 < nt) lstoreV(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O[q, dv] = O^T acc / l * unscale(vtmax[dv]) through a per-wave LDS transpose, lse = m + log l
    const float inv_l = 1.0f / l_run;
    float *so = reinterpret_cast<float *>(smem) + wid * 32 * kFlORow;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dv = 32 * mt + 8 * g + 4 * fh;
            const fl_f32x4 uv = {h2_unscale(p.vtmax[dv]), h2_unscale(p.vtmax[dv + 1]), h2_unscale(p.vtmax[dv + 2]),
                                 h2_unscale(p.vtmax[dv + 3])};
            fl_f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (oacc[mt][4 * g + i] * uv[i]) * inv_l;
            *reinterpret_cast<fl_f32x4 *>(so + fr * kFlORow + dv) = v;
        }
    if (fh == 0 && q0 + fr < p.ns) p.lse[q0 + fr] = m_run + logf(l_run);
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
        if (q0 + r < p.ns)
            *reinterpret_cast<fl_f32x4 *>(p.o + (q0 + r) * p.ldo + 4 * lane) =
                *reinterpret_cast<const fl_f32x4 *>(so + r * kFlORow + 4 * lane);
    }
}

}  // namespace dc

using namespace dc;

extern "C" int dc_attn_flash_fwd(const float *q, int64_t ldq, const float *q_rowmax, const void *k_image,
                                 const float *k_rowmax, const void *vt_image, const float *vt_rowmax, int64_t ns,
                                 int64_t nr, int64_t nr_padded, int64_t d, float *o, int64_t ldo, float *lse,
                                 dc_stream_t stream) {
    DC_REQUIRE(ns >= 0 && nr >= 1 && nr_padded >= nr, "dc_attn_flash_fwd: needs ns >= 0, 1 <= nr <= nr_padded");
    if (ns == 0) return DC_OK;
    DC_REQUIRE(d == kFlD, "dc_attn_flash_fwd: d = dv = %d only (got %lld); use the blocked form", kFlD, (long long)d);
    DC_REQUIRE(nr_padded % kFlT == 0, "dc_attn_flash_fwd: nr_padded must be a multiple of %d", kFlT);
    DC_REQUIRE(q && q_rowmax && k_image && k_rowmax && vt_image && vt_rowmax && o && lse,
               "dc_attn_flash_fwd: null pointer");
    DC_REQUIRE(ldq >= d && ldo >= d && ldq % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)q & 15) == 0 &&
                   ((uintptr_t)o & 15) == 0 && ((uintptr_t)k_image & 15) == 0 && ((uintptr_t)vt_image & 15) == 0,
               "dc_attn_flash_fwd: rows must be 16-byte aligned");
    DC_REQUIRE((ns + kFlQ - 1) / kFlQ < (int64_t)INT32_MAX, "dc_attn_flash_fwd: too many query tiles");
    FlashParams p{q, ldq, q_rowmax, (const char *)k_image, k_rowmax, (const char *)vt_image, vt_rowmax,
                  ns, nr, nr_padded, o, ldo, lse};
    hipLaunchKernelGGL(k_attn_flash_fwd, dim3((unsigned)((ns + kFlQ - 1) / kFlQ)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return check_launch("dc_attn_flash_fwd");
}
