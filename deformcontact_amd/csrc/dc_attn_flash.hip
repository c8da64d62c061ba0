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
// the A operand (the V^T image) lists its keys in the same order: dc_attn_flash_prep rewrites the image with the
// 4-key chunks of every 16 keys in the order (0, 2 | 1, 3) for that.
//
// Per 32-key tile and wave: 48 MFMAs for S^T (16 k-steps x 3 products), 48 for O^T (8 row tiles of V^T x 2 k-steps x
// 3 products), 64 ds_read_b128 fragment reads; q fragments (128 VGPRs) and the O^T accumulators (128) stay in
// registers for the whole kernel (512 registers per lane at one wave per SIMD).  Nothing is left for staging
// registers (a first version that staged K / V^T through 32 registers exposed two L2 latencies per tile: 4.0 ms per
// head at batch 32, 2.3 ms with the staging removed), so the tiles come in by LDS-DMA (global_load_lds, 16 bytes per
// lane, 1 KB per instruction, no registers): three K buffers and two V^T buffers (160 KB, all of the LDS), the 8 + 8
// instructions of a wave for V^T(t+1) and K(t+2) issued between the MFMA groups of tile t's score product, one
// counted s_waitcnt vmcnt + raw s_barrier per tile.  LDS-DMA writes lane-linear, so the bank swizzle is applied on the
// SOURCE side (a lane fetches the piece that belongs in its slot), and the key order the weights need is baked into
// the V^T image once by dc_attn_flash_prep.  The keys' power-of-two scales come in through scalar loads.
//
// Register file by hand (all MFMAs are inline asm): the q fragments are pinned to 128 accumulation registers and read
// from there as B operands, four of the eight output tiles live in the other accumulation registers, four in
// architectural VGPRs.  With the builtins hipcc kept q in the 256 architectural registers, ran out, and moved ~140
// registers per tile through v_accvgpr_read / write (or gathered scattered dwords in front of every MFMA).
// Measured at batch 32 (32,768 queries x 24,384 keys, one head; tools/exp/attn_flash.py): 2.56 ms = 0.96 PFLOP/s of
// fp16 products (0.38 of the nominal 2.5 PF, ~0.56 of what the chip sustains under DVFS); the blocked form's three
// launches per 2,048-row block take 5.6 - 6.0 ms.  Timing ablations of this loop: no staging 2.30, no softmax
// arithmetic 2.27, neither 2.10 ms - the two MFMA chains themselves set the time.
#include "dc_dense.h"

namespace dc {

using fl_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using fl_f32x4 = __attribute__((ext_vector_type(4))) float;
using fl_u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int kFlD = DC_ATTN_FLASH_D;          // d = dv
constexpr int kFlQ = 128;                      // queries per workgroup (4 waves x 32)
constexpr int kFlT = 32;                       // keys per tile
constexpr int kFlKRow = kFlD * 4;              // bytes per key row of the K image (16 records x 64 B)
constexpr int kFlKSz = kFlT * kFlKRow;         // 32 KB
constexpr int kFlVRow = kFlT * 4;              // bytes per V^T row and tile (2 records x 64 B)
constexpr int kFlVSz = kFlD * kFlVRow;         // 32 KB
constexpr int kFlORow = kFlD + 4;              // floats per query row of the epilogue transpose buffer
constexpr int kFlSmem = 3 * kFlKSz + 2 * kFlVSz;   // 160 KB (the epilogue's 4 x 32 x 260 floats fit inside)

struct FlashParams {
    const float *q;        // [ns, ldq] fp32
    int64_t ldq;
    const float *qmax;     // [ns] row maxima of |q|
    const char *kimg;      // [nrp, d] fp16x2 image of the keys (rows scaled by kmax)
    const float *kuns;     // [nrp] h2_unscale(row maximum) of every key (dc_attn_flash_prep)
    const char *vtimg;     // [dv, nrp] fp16x2 image of V^T (rows scaled by vtmax), chunks reordered (flash_vt_image)
    const float *vtmax;    // [dv]
    int64_t ns, nr, nrp;
    float *o;              // [ns, ldo]
    int64_t ldo;
    float *lse;            // [ns]
};

__device__ __forceinline__ int fl_vswz(int row) { return ((row >> 1) & 7) ^ ((row & 1) << 1); }   // as hw_swz

#define DC_FL_GPTR(p) ((const void __attribute__((address_space(1))) *)(p))
#define DC_FL_LPTR(p) ((void __attribute__((address_space(3))) *)(p))

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_attn_flash_fwd(FlashParams p) {
    __shared__ __attribute__((aligned(16))) char smem[kFlSmem];
    char *sK = smem, *sV = smem + 3 * kFlKSz;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: DMA bases stay in SGPRs / M0
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
            // every fragment as ONE 128-bit accumulation-register tuple from here on (left alone hipcc keeps the four
            // dwords apart and gathers them in front of every MFMA)
            asm volatile("" : "+a"(qh[ks]), "+a"(ql[ks]));
        }
    }

    // ---- LDS-DMA staging.  An instruction moves 64 x 16 bytes into 1 KB of consecutive LDS: lane l fills slot l.
    // K tile (32 rows of 1 KB, piece pc of row r in slot pc ^ (r & 15)): instruction (jj, hi) of wave w fills row
    // r = 4 w + jj + 16 hi, whose swizzle only depends on 4 w + jj - four lane offsets.  V^T tile (256 rows of 128 B,
    // piece q of row r in slot q ^ vswz(r)): instruction I = 8 w + j fills rows 8 I .. 8 I + 7 (lane l: row 8 I + (l >> 3),
    // slot l & 7); vswz only depends on the low 4 row bits = 8 (I & 1) + (l >> 3) - two lane offsets.
    const int nt = (int)((p.nr + kFlT - 1) / kFlT);
    unsigned kso[4], vso[2];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) kso[jj] = (unsigned)((4 * wid + jj) * kFlKRow + 16 * (lane ^ (4 * wid + jj)));
#pragma unroll
    for (int e = 0; e < 2; ++e)
        vso[e] = (unsigned)((lane >> 3) * (p.nrp * 4) + 16 * ((lane & 7) ^ fl_vswz(8 * e + (lane >> 3))));
    const int64_t v8rows = 8 * p.nrp * 4;               // bytes between the row groups of consecutive V^T instructions
    auto dma_k = [&](int t, int slot, int c) {           // instruction c (0..7) of this wave for K tile t
        const int jj = c & 3, hi = c >> 2;
        const char *src = p.kimg + (int64_t)t * kFlKSz + hi * 16 * kFlKRow;
        char *dst = sK + slot * kFlKSz + (4 * wid + jj + 16 * hi) * kFlKRow;
        __builtin_amdgcn_global_load_lds(DC_FL_GPTR(src + kso[jj]), DC_FL_LPTR(dst), 16, 0, 0);
    };
    auto dma_v = [&](int t, int slot, int c) {           // instruction c (0..7) of this wave for V^T tile t
        const int I = 8 * wid + c;
        const char *src = p.vtimg + (int64_t)t * kFlVRow + I * v8rows;
        char *dst = sV + slot * kFlVSz + I * 1024;
        __builtin_amdgcn_global_load_lds(DC_FL_GPTR(src + vso[c & 1]), DC_FL_LPTR(dst), 16, 0, 0);
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

#pragma unroll
    for (int c = 0; c < 8; ++c) dma_k(0, 0, c);
#pragma unroll
    for (int c = 0; c < 8; ++c) dma_v(0, 0, c);
    if (nt > 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) dma_k(1, 1, c);
    }
    // the keys' power-of-two scales come in through SCALAR loads (constant address space: wave-uniform address, data
    // written before the launch): a vector load inside the loop would make hipcc wait for vmcnt(0) - i.e. for the DMA
    // queue - at its first use.  The next tile's 32 values are fetched right before the barrier that ends a tile.
    const float __attribute__((address_space(4))) *kmaxc =
        (const float __attribute__((address_space(4))) *)(uintptr_t)p.kuns;
    // uk0[i] / uk1[i]: the scale of the key that accumulator register i holds in the lower / upper lane half - SSA
    // vectors, not arrays (hipcc turns "fh ? uk[k + 4] : uk[k]" on an array into a lane-indexed load from a scratch
    // copy: VMEM inside the loop, and a vmcnt(0) with it)
    f32x16 uk0, uk1;
    auto load_uk = [&](int64_t base) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = (i & 3) + 8 * (i >> 2);
            uk0[i] = kmaxc[base + kk], uk1[i] = kmaxc[base + kk + 4];
        }
    };
    load_uk(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    int kslot = 0;                                       // t % 3
    for (int t = 0; t < nt; ++t) {
        const int vbuf = t & 1;
        const int kslot2 = kslot == 0 ? 2 : kslot - 1;   // (t + 2) % 3
        const bool more_v = t + 1 < nt, more_k = t + 2 < nt;
        // ---- S^T tile: rows = the tile's 32 keys, columns = this wave's 32 queries
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
        {
            // the fragments of a k-step are read while the three MFMAs of the previous one run (sched_barrier: hipcc
            // otherwise hoists all 32 reads of the tile to the top - 128 registers - and spills); the DMA instructions
            // of the next tiles go one per MFMA group: V^T(t+1) first (it is waited for first), then K(t+2)
            // Fragment reads and their waits are written out (inline asm): with LDS-DMA in the loop hipcc's own counter
            // bookkeeping falls back to s_waitcnt lgkmcnt(0) - a full drain that exposes the LDS latency every fourth
            // k-step.  Ring of four fragment sets, reads three k-steps ahead, counted waits (LDS returns in order).
            unsigned ka[8];
            {
                const unsigned kbase = (unsigned)(uintptr_t)DC_FL_LPTR(sK + kslot * kFlKSz);
#pragma unroll
                for (int c = 0; c < 8; ++c) ka[c] = kbase + (unsigned)kfo[c];
            }
            fl_f16x8 kh[4], kl[4];
            auto kfrags = [&](int ks) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kh[ks & 3]) : "v"(ka[2 * (ks & 3)]), "i"((ks >> 2) * 256));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kl[ks & 3]) : "v"(ka[2 * (ks & 3) + 1]), "i"((ks >> 2) * 256));
            };
            auto kmma = [&](int ks) {
                // same term order as k_fwd_h2w with x = q (A there) and W = k (B there): x_l w_h, x_h w_l, x_h w_h;
                // the q fragments are read straight from the accumulation registers ("a")
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(st) : "v"(kh[ks & 3]), "a"(ql[ks]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(st) : "v"(kl[ks & 3]), "a"(qh[ks]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(st) : "v"(kh[ks & 3]), "a"(qh[ks]));
            };
            auto dma1 = [&](int c) {                     // instruction c of this iteration's 16
                if (c < 8) {
                    if (more_v) dma_v(t + 1, vbuf ^ 1, c);
                } else if (more_k) {
                    dma_k(t + 2, kslot2, c - 8);
                }
            };
            kfrags(0);
            kfrags(1);
            kfrags(2);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + 3 < 16) kfrags(ks + 3);
                // the 2 x (sets read after set ks) youngest reads may still be in flight
                if (ks < 13) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else if (ks == 13) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if (ks == 14) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                kmma(ks);
                dma1(ks);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // (the hazard recogniser does not see through the asm: a 32x32 MFMA result needs 18 idle issue slots before
        // another instruction may read it)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(st));
        // ---- online softmax over the keys of this lane's query: reg i <-> key (i & 3) + 8 (i >> 2) + 4 fh
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = (st[i] * uq) * (fh ? uk1[i] : uk0[i]);            // as the h2w epilogue: (acc * s_row) * s_col
        }
        if (t == nt - 1) {
            const int64_t kbase = (int64_t)t * kFlT + 4 * fh;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kbase + (i & 3) + 8 * (i >> 2) >= p.nr) s[i] = -INFINITY;
        }
        fl_f16x8 ph[2], pl[2];
        {
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
        }
        // ---- O^T += V^T_tile P^T: rows = dv (8 tiles of 32), contraction over the tile's keys (2 k-steps of 16)
        {
            unsigned va[4];
            {
                const unsigned vbase = (unsigned)(uintptr_t)DC_FL_LPTR(sV + vbuf * kFlVSz);
#pragma unroll
                for (int c = 0; c < 4; ++c) va[c] = vbase + (unsigned)vfo[c];
            }
            fl_f16x8 vh[4], vl[4];                       // ring of four fragment sets, reads three steps ahead
            auto vfrags = [&](int c) {                   // c = 2 mt + m
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vh[c & 3]) : "v"(va[2 * (c & 1)]), "i"((c >> 1) * 32 * kFlVRow));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vl[c & 3]) : "v"(va[2 * (c & 1) + 1]), "i"((c >> 1) * 32 * kFlVRow));
            };
            auto vmma = [&](int c) {
                const int mt = c >> 1, m = c & 1;
                // blocked form: x = P (A there), W = V^T (B there): x_l w_h, x_h w_l, x_h w_h.  Register file by hand:
                // the q fragments fill half of the accumulation registers, so only four of the eight output tiles
                // live there ("+a"), the other four in architectural VGPRs ("+v") - 64 accumulation registers stay
                // free and hipcc stops shuffling tuples (v_accvgpr_mov) in front of every MFMA
                if (mt < 4) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(oacc[mt]) : "v"(vh[c & 3]), "v"(pl[m]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(oacc[mt]) : "v"(vl[c & 3]), "v"(ph[m]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(oacc[mt]) : "v"(vh[c & 3]), "v"(ph[m]));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(oacc[mt]) : "v"(vh[c & 3]), "v"(pl[m]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(oacc[mt]) : "v"(vl[c & 3]), "v"(ph[m]));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(oacc[mt]) : "v"(vh[c & 3]), "v"(ph[m]));
                }
            };
            asm volatile("s_nop 1" : "+v"(ph[0]), "+v"(pl[0]), "+v"(ph[1]), "+v"(pl[1]));   // VALU write -> MFMA read
            vfrags(0);
            vfrags(1);
            vfrags(2);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c + 3 < 16) vfrags(c + 3);
                if (c < 13) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else if (c == 13) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if (c == 14) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                vmma(c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t + 1 < nt) load_uk((int64_t)(t + 1) * kFlT);   // the next tile's key scales (arrive under the barrier)
        // V^T(t+1) (and K(t+1), older) must have landed; the 8 instructions of K(t+2) may stay in flight
        if (more_k) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        kslot = kslot == 2 ? 0 : kslot + 1;

    }

    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");          // last MFMA results -> VALU reads (asm MFMAs: no automatic nops)
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

// ---- backward producer: P = exp(S - lse), dS = P o (dP - delta) for ALL rows in one launch ---------------------------
// The blocked backward makes them per 2,048-row block with three launches - score GEMM + exp epilogue (writes P), dP GEMM
// (writes dP), dc_attn_ds_rows (reads both, writes dS) - each bound by its 200 MB of stores: 442 us per block and head.
// Here one workgroup owns 128 queries (the forward's skeleton: transposed tiles, q AND dO fragments resident in the 256
// accumulation registers, K / V row tiles by LDS-DMA) and sweeps the keys twice:
//   sweep 1: S^T = K_t Q^T, dP^T = V_t dO^T  ->  delta_i = sum_j P_ij dP_ij / sum_j P_ij   (nothing written)
//   sweep 2: the same products again (bit for bit), dS = P (dP - delta); P and dS tiles go out through a per-wave LDS
//            transpose as full 128-byte row segments, the row maxima of |dS| at the end.
// delta comes from the SAME recomputed P and dP that form dS (row sums of dS vanish to rounding - the precision fix of
// round 2), which is what the second sweep is for.  4 GEMM-equivalents instead of 2, but no score-sized
// round trip: P and dS are written once.  Downstream (dQ = dS K, dK = dS^T Q, dV = P^T dO) the existing kernels run as
// three launches over all rows.
constexpr int kDsTRow = 36;                                   // floats per row of the transpose region (32 + pad, 16-byte rows)
constexpr int kDsTSz = 32 * kDsTRow * 4;                      // 4,608 B per wave
constexpr int kDsSmem = 4 * kFlKSz + 4 * kDsTSz;              // 2 K + 2 V row tiles + 4 transpose regions = 149,504 B

struct FlashDsParams {
    const float *q;        // [ns, ldq]
    int64_t ldq;
    const float *qmax;     // [ns]
    const float *go;       // [ns, ldgo] upstream gradient of the head's output
    int64_t ldgo;
    const float *gomax;    // [ns]
    const char *kimg;      // [nrp, d] image of the (centred) keys
    const float *kuns;     // [nrp] unscale factors of its rows
    const char *vimg;      // [nrp, dv] image of the values' ROWS
    const float *vuns;     // [nrp]
    const float *lse;      // [ns] row log-sum-exp of the forward
    int64_t ns, nr, nrp;
    float *pmat, *dsmat;   // [ns, ldp] each
    int64_t ldp;
    float *dsmax;          // [ns] max_j |dS_ij|
    // single-sweep mode (delta_in != null): dS' = P (dP - delta_in) with the caller's delta (rowsum(dO o O)), and
    // eps_out[i] = sum_j dS'_ij / sum_j P_ij - the amount by which the consistent delta differs from delta_in; the
    // consumers that need rows of dS to sum to zero take dS' - eps_i P (dc_tag_linear_bwd_dw_h2_corr)
    const float *delta_in;
    float *eps_out;
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_attn_flash_ds(FlashDsParams p) {
    __shared__ __attribute__((aligned(16))) char smem[kDsSmem];
    char *sK = smem, *sV = smem + 2 * kFlKSz;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sT = reinterpret_cast<float *>(smem + 4 * kFlKSz + wid * kDsTSz);
    const int64_t q0 = (int64_t)blockIdx.x * kFlQ + wid * 32;
    int64_t qrow = q0 + fr;
    qrow = qrow < p.ns ? qrow : p.ns - 1;

    // ---- q and dO rows of this lane's query as B-operand fragments, pinned to the accumulation registers
    const float uq = h2_unscale(p.qmax[qrow]), ugo = h2_unscale(p.gomax[qrow]);
    const float lse_q = p.lse[qrow];
    fl_f16x8 qh[16], ql[16], gh[16], gl[16];
    {
        const float sq = h2_scale(p.qmax[qrow]), sg = h2_scale(p.gomax[qrow]);
        const float *qp = p.q + qrow * p.ldq + 8 * fh, *gp = p.go + qrow * p.ldgo + 8 * fh;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const fl_f32x4 a = *reinterpret_cast<const fl_f32x4 *>(qp + 16 * ks) * sq;
            const fl_f32x4 b = *reinterpret_cast<const fl_f32x4 *>(qp + 16 * ks + 4) * sq;
            const fl_f32x4 c = *reinterpret_cast<const fl_f32x4 *>(gp + 16 * ks) * sg;
            const fl_f32x4 e = *reinterpret_cast<const fl_f32x4 *>(gp + 16 * ks + 4) * sg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 ha = (_Float16)a[i], hb = (_Float16)b[i], hc = (_Float16)c[i], he = (_Float16)e[i];
                qh[ks][i] = ha, qh[ks][4 + i] = hb;
                ql[ks][i] = (_Float16)(a[i] - (float)ha), ql[ks][4 + i] = (_Float16)(b[i] - (float)hb);
                gh[ks][i] = hc, gh[ks][4 + i] = he;
                gl[ks][i] = (_Float16)(c[i] - (float)hc), gl[ks][4 + i] = (_Float16)(e[i] - (float)he);
            }
            asm volatile("" : "+a"(qh[ks]), "+a"(ql[ks]), "+a"(gh[ks]), "+a"(gl[ks]));
        }
    }

    // ---- staging (as the forward's K tiles; the V image of ROWS has the same 1 KB-per-key layout)
    unsigned kso[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) kso[jj] = (unsigned)((4 * wid + jj) * kFlKRow + 16 * (lane ^ (4 * wid + jj)));
    auto dma_rows = [&](const char *img, char *buf, int t, int slot, int c) {   // instruction c (0..7) of this wave
        const int jj = c & 3, hi = c >> 2;
        const char *src = img + (int64_t)t * kFlKSz + hi * 16 * kFlKRow;
        char *dst = buf + slot * kFlKSz + (4 * wid + jj + 16 * hi) * kFlKRow;
        __builtin_amdgcn_global_load_lds(DC_FL_GPTR(src + kso[jj]), DC_FL_LPTR(dst), 16, 0, 0);
    };
    int kfo[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kfo[c] = fr * kFlKRow + 16 * ((4 * (c >> 1) + 2 * (c & 1) + fh) ^ (fr & 15));
    const float __attribute__((address_space(4))) *kunsc = (const float __attribute__((address_space(4))) *)(uintptr_t)p.kuns;
    const float __attribute__((address_space(4))) *vunsc = (const float __attribute__((address_space(4))) *)(uintptr_t)p.vuns;
    auto load_us = [&](const float __attribute__((address_space(4))) *src, int64_t base, f32x16 &u0, f32x16 &u1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = (i & 3) + 8 * (i >> 2);
            u0[i] = src[base + kk], u1[i] = src[base + kk + 4];
        }
    };
    // one product chain: acc = tile rows (A, from LDS) x resident fragments (B, accumulation registers); ring of four
    // fragment sets, reads three k-steps ahead, counted waits; hook(ks) issues this step's DMA instruction
    auto chain = [&](f32x16 &acc, const char *tile, fl_f16x8 (&bh)[16], fl_f16x8 (&bl)[16], auto hook) {
        unsigned ka[8];
        const unsigned base = (unsigned)(uintptr_t)DC_FL_LPTR(tile);
#pragma unroll
        for (int c = 0; c < 8; ++c) ka[c] = base + (unsigned)kfo[c];
        fl_f16x8 ah[4], al[4];
        auto frags = [&](int ks) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[ks & 3]) : "v"(ka[2 * (ks & 3)]), "i"((ks >> 2) * 256));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[ks & 3]) : "v"(ka[2 * (ks & 3) + 1]), "i"((ks >> 2) * 256));
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        frags(0);
        frags(1);
        frags(2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + 3 < 16) frags(ks + 3);
            if (ks < 13) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else if (ks == 13) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (ks == 14) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // x = the resident operand (A in k_fwd_h2w), W = the tile: x_l w_h, x_h w_l, x_h w_h
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[ks & 3]), "a"(bl[ks]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(al[ks & 3]), "a"(bh[ks]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[ks & 3]), "a"(bh[ks]));
            hook(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));   // MFMA result -> VALU read (asm: no automatic nops)
    };

    constexpr float kLog2e = 1.4426950408889634f;
    const int nt1 = (int)((p.nr + kFlT - 1) / kFlT), nt2 = (int)(p.nrp / kFlT);
    float acc_pd = 0.f, acc_p = 0.f, delta = p.delta_in ? p.delta_in[qrow] : 0.f, dsm = 0.f;
    const bool full_tile = q0 + 32 <= p.ns;              // all 32 queries of this wave exist (q0: blockIdx + readfirstlane)
    for (int sweep = p.delta_in ? 1 : 0; sweep < 2; ++sweep) {
        const int nt = sweep ? nt2 : nt1;
#pragma unroll
        for (int c = 0; c < 8; ++c) dma_rows(p.kimg, sK, 0, 0, c);
#pragma unroll
        for (int c = 0; c < 8; ++c) dma_rows(p.vimg, sV, 0, 0, c);
        f32x16 uk0, uk1;
        load_us(kunsc, 0, uk0, uk1);
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            const bool more = t + 1 < nt;
            f32x16 st, dpt;
            chain(st, sK + buf * kFlKSz, qh, ql, [&](int ks) {
                if (more && ks < 8) dma_rows(p.kimg, sK, t + 1, buf ^ 1, ks);
            });
            float pv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float sv = (st[i] * uq) * (fh ? uk1[i] : uk0[i]);
                pv[i] = __builtin_amdgcn_exp2f((sv - lse_q) * kLog2e);
            }
            if (t >= nt1 - 1) {                          // key padding: weights exactly 0
                const int64_t kbase = (int64_t)t * kFlT + 4 * fh;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kbase + (i & 3) + 8 * (i >> 2) >= p.nr) pv[i] = 0.f;
            }
            f32x16 uv0, uv1;
            load_us(vunsc, (int64_t)t * kFlT, uv0, uv1);
            chain(dpt, sV + buf * kFlKSz, gh, gl, [&](int ks) {
                if (more && ks < 8) dma_rows(p.vimg, sV, t + 1, buf ^ 1, ks);
            });
            float dp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = (dpt[i] * ugo) * (fh ? uv1[i] : uv0[i]);
            if (!sweep) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc_pd += pv[i] * dp[i];
                    acc_p += pv[i];
                }
            } else {
                float dsv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    dsv[i] = pv[i] * (dp[i] - delta);
                    dsm = fmaxf(dsm, fabsf(dsv[i]));
                    acc_pd += dsv[i];                    // (single-sweep mode: row sums of dS' and of P)
                    acc_p += pv[i];
                }
                // the two tiles out through the wave's transpose region: [query][key] rows of 32 floats (+ pad), then
                // 8 lanes per row write one 128-byte segment per query
                const int64_t col0 = (int64_t)t * kFlT + 4 * (lane & 7);
                const bool full = full_tile;            // wave-uniform: no per-row predicates on whole tiles
#pragma unroll
                for (int mtx = 0; mtx < 2; ++mtx) {
                    const float *v = mtx ? dsv : pv;
                    float *mat = (mtx ? p.dsmat : p.pmat) + (q0 + (lane >> 3)) * p.ldp + col0;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<fl_f32x4 *>(sT + fr * kDsTRow + 8 * g + 4 * fh) =
                            fl_f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                    asm volatile("" ::: "memory");      // (LDS operations of one wave complete in order)
                    fl_f32x4 o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[j] = *reinterpret_cast<const fl_f32x4 *>(sT + (8 * j + (lane >> 3)) * kDsTRow + 4 * (lane & 7));
                    if (full) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<fl_f32x4 *>(mat + 8 * j * p.ldp) = o[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (q0 + 8 * j + (lane >> 3) < p.ns) *reinterpret_cast<fl_f32x4 *>(mat + 8 * j * p.ldp) = o[j];
                    }
                    asm volatile("" ::: "memory");
                }
            }
            if (more) load_us(kunsc, (int64_t)(t + 1) * kFlT, uk0, uk1);
            // the 16 DMA instructions of this iteration must have landed; in sweep 2 the 8 store instructions issued
            // after them may stay in flight (VMEM operations retire in order) - but only a wave whose 32 queries all
            // exist issues exactly 8: in a ragged last tile a predicated store with no live lane is branched around,
            // and vmcnt(8) would then leave DMA pieces of V(t+1) / K(t+1) in flight past the barrier
            if (sweep && full_tile) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
        }
        if (!sweep) {
            acc_pd += __shfl_xor(acc_pd, 32);
            acc_p += __shfl_xor(acc_p, 32);
            delta = acc_p > 0.f ? acc_pd / acc_p : acc_pd;         // as dc_attn_ds_rows
        }
    }
    dsm = fmaxf(dsm, __shfl_xor(dsm, 32));
    if (fh == 0 && q0 + fr < p.ns) p.dsmax[q0 + fr] = dsm;
    if (p.delta_in) {
        acc_pd += __shfl_xor(acc_pd, 32);
        acc_p += __shfl_xor(acc_p, 32);
        if (fh == 0 && q0 + fr < p.ns) p.eps_out[q0 + fr] = acc_p > 0.f ? acc_pd / acc_p : 0.f;
    }
}

// the V^T image with the 4-key chunks (8 bytes) of every plane of every 64-byte record in the order (c0, c2, c1, c3)
__global__ void __launch_bounds__(256)
k_attn_flash_vt_image(fl_u32x2 *img, int64_t nchunks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 32-byte plane
    if (4 * i >= nchunks) return;
    fl_u32x2 *c = img + 4 * i;
    const fl_u32x2 c1 = c[1], c2 = c[2];
    c[1] = c2, c[2] = c1;
}

__global__ void __launch_bounds__(256)
k_attn_flash_unscale(const float *rowmax, float *uns, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) uns[i] = h2_unscale(rowmax[i]);
}

}  // namespace dc

using namespace dc;

extern "C" int dc_attn_flash_prep(void *vt_image, int64_t dv, int64_t nr_padded, const float *k_rowmax,
                                  float *k_unscale, dc_stream_t stream) {
    DC_REQUIRE(dv >= 0 && nr_padded >= 0 && nr_padded % 16 == 0, "dc_attn_flash_prep: nr_padded must be a multiple of 16");
    const int64_t nchunks = dv * nr_padded / 2;          // 8-byte chunks: 4 bytes per element
    if (nchunks > 0) {
        DC_REQUIRE(vt_image && ((uintptr_t)vt_image & 15) == 0, "dc_attn_flash_prep: null or misaligned image");
        const int64_t nthreads = nchunks / 4;
        DC_LAUNCH(k_attn_flash_vt_image, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, (fl_u32x2 *)vt_image, nchunks);
    }
    if (nr_padded > 0) {
        DC_REQUIRE(k_rowmax && k_unscale, "dc_attn_flash_prep: null key maxima / output");
        DC_LAUNCH(k_attn_flash_unscale, dim3((unsigned)((nr_padded + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, k_rowmax, k_unscale, nr_padded);
    }
    return check_launch("dc_attn_flash_prep");
}

extern "C" int dc_attn_flash_fwd(const float *q, int64_t ldq, const float *q_rowmax, const void *k_image,
                                 const float *k_unscale, const void *vt_image, const float *vt_rowmax, int64_t ns,
                                 int64_t nr, int64_t nr_padded, int64_t d, float *o, int64_t ldo, float *lse,
                                 dc_stream_t stream) {
    DC_REQUIRE(ns >= 0 && nr >= 1 && nr_padded >= nr, "dc_attn_flash_fwd: needs ns >= 0, 1 <= nr <= nr_padded");
    if (ns == 0) return DC_OK;
    DC_REQUIRE(d == kFlD, "dc_attn_flash_fwd: d = dv = %d only (got %lld); use the blocked form", kFlD, (long long)d);
    DC_REQUIRE(nr_padded % kFlT == 0, "dc_attn_flash_fwd: nr_padded must be a multiple of %d", kFlT);
    DC_REQUIRE(q && q_rowmax && k_image && k_unscale && vt_image && vt_rowmax && o && lse,
               "dc_attn_flash_fwd: null pointer");
    DC_REQUIRE(ldq >= d && ldo >= d && ldq % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)q & 15) == 0 &&
                   ((uintptr_t)o & 15) == 0 && ((uintptr_t)k_image & 15) == 0 && ((uintptr_t)vt_image & 15) == 0,
               "dc_attn_flash_fwd: rows must be 16-byte aligned");
    DC_REQUIRE((ns + kFlQ - 1) / kFlQ < (int64_t)INT32_MAX, "dc_attn_flash_fwd: too many query tiles");
    FlashParams p{q, ldq, q_rowmax, (const char *)k_image, k_unscale, (const char *)vt_image, vt_rowmax,
                  ns, nr, nr_padded, o, ldo, lse};
    DC_LAUNCH(k_attn_flash_fwd, dim3((unsigned)((ns + kFlQ - 1) / kFlQ)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return check_launch("dc_attn_flash_fwd");
}

extern "C" int dc_attn_flash_ds(const float *q, int64_t ldq, const float *q_rowmax, const float *go, int64_t ldgo,
                                const float *go_rowmax, const void *k_image, const float *k_unscale,
                                const void *v_image, const float *v_unscale, const float *lse, int64_t ns, int64_t nr,
                                int64_t nr_padded, int64_t d, float *p_out, float *ds_out, int64_t ldp, float *ds_rowmax,
                                const float *delta_in, float *eps_out, dc_stream_t stream) {
    DC_REQUIRE((delta_in == nullptr) == (eps_out == nullptr), "dc_attn_flash_ds: delta_in and eps_out go together");
    DC_REQUIRE(ns >= 0 && nr >= 1 && nr_padded >= nr, "dc_attn_flash_ds: needs ns >= 0, 1 <= nr <= nr_padded");
    if (ns == 0) return DC_OK;
    DC_REQUIRE(d == kFlD, "dc_attn_flash_ds: d = dv = %d only (got %lld); use the blocked form", kFlD, (long long)d);
    DC_REQUIRE(nr_padded % kFlT == 0 && ldp >= nr_padded && ldp % 4 == 0,
               "dc_attn_flash_ds: nr_padded must be a multiple of %d, ldp >= nr_padded and a multiple of 4", kFlT);
    DC_REQUIRE(q && q_rowmax && go && go_rowmax && k_image && k_unscale && v_image && v_unscale && lse && p_out &&
                   ds_out && ds_rowmax, "dc_attn_flash_ds: null pointer");
    DC_REQUIRE(ldq >= d && ldgo >= d && ldq % 4 == 0 && ldgo % 4 == 0 && ((uintptr_t)q & 15) == 0 &&
                   ((uintptr_t)go & 15) == 0 && ((uintptr_t)k_image & 15) == 0 && ((uintptr_t)v_image & 15) == 0 &&
                   ((uintptr_t)p_out & 15) == 0 && ((uintptr_t)ds_out & 15) == 0,
               "dc_attn_flash_ds: rows must be 16-byte aligned");
    DC_REQUIRE((ns + kFlQ - 1) / kFlQ < (int64_t)INT32_MAX, "dc_attn_flash_ds: too many query tiles");
    FlashDsParams p{q, ldq, q_rowmax, go, ldgo, go_rowmax, (const char *)k_image, k_unscale, (const char *)v_image,
                    v_unscale, lse, ns, nr, nr_padded, p_out, ds_out, ldp, ds_rowmax, delta_in, eps_out};
    DC_LAUNCH(k_attn_flash_ds, dim3((unsigned)((ns + kFlQ - 1) / kFlQ)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("dc_attn_flash_ds");
}
