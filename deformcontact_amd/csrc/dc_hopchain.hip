// dc_hopchain.hip -- K chained hops of a batch of small graphs in ONE launch, each graph's features resident in LDS.
//
// TAGConv needs x_k = A_hat x_{k-1}, k = 1..K (/root/reference/models/model.py:71,77 -> PyG tag_conv.py: K = 3
// dependent propagate calls), its backward the same chain over the transposed adjacency.  Hop by hop (dc_spmm_f32)
// every x_k goes out to memory and comes back as E x 1 KiB of L2 -> CU gathers: 6 x the compulsory read bytes, and
// the launch sits at 0.41 of the HBM roofline.  A PyG batch is a block-diagonal union of meshes
// (Batch.from_data_list, train.py:36-38) and no edge leaves its mesh, so here one 1024-thread workgroup owns
// (graph, 32-column slice): the slice of the source block - <= 1024 nodes x 128 B - goes global -> LDS ONCE (LDS-DMA),
// every hop gathers its neighbour pieces from LDS (ds_read_b128, 8 lanes per 128-byte piece, 8 destination rows per
// wave-instruction), keeps the produced rows in registers until every wave has finished reading, then overwrites the
// LDS slice and streams the block out.  Memory traffic per chain: read 1 block + write K blocks (+ the adjacency from
// L2), instead of K x (6 gathers + 1 write) per row.
//
// Arithmetic and order are those of dc_spmm_f32 (multiply and add rounded separately, neighbours in p order, sum
// started at +0), so the blocks are bit-identical to K single hops.  A slot past the end of a row reads a row of
// zeros with weight 0: the running sum started at +0 and can never be -0, so adding +0 leaves it unchanged.
#include <stdlib.h>

#include <string>

#include "dc_common.h"

#pragma clang fp contract(off)

// timing-only ablations (tools/exp/hop_chain_abl.py builds this file with -DDC_CHAIN_ABL=<bits>; results are wrong by
// construction): 1 no block stores, 2 no neighbour slots at all, 4 no LDS write-back / barriers between hops,
// 8 no source-block staging, 16 no id / weight loads, 32 LDS gathers but no arithmetic, 64 block stores to a
// contiguous 128 KiB region per workgroup and hop (is the row-strided 128-byte store pattern what costs?), 128 ids /
// weights made up in registers instead of loaded (do the in-loop VMEM loads couple the compute to the stores?),
// 256 row maxima formed but not published (what do the atomics cost?  ~2 us per launch, profiles/r05/o_chain_rowmax_atomics.txt)
#ifndef DC_CHAIN_ABL
#define DC_CHAIN_ABL 0
#endif

namespace dc {

constexpr int kChainCols = 32;        // columns per slice: 128-byte row pieces, one full cache line per store
constexpr int kChainSteps = 8;        // 16 waves x 8 row groups x 8 steps = DC_CHAIN_MAX_NODES nodes per graph
constexpr int kChainGraphs = 96;      // graphs per launch (their node offsets travel as kernel arguments)
// Every chain workgroup asks for the WHOLE LDS of its compute unit (160 KB), whatever its slice needs: no other LDS-using
// workgroup can then be resident beside it.  Round 5 traced the rare run-to-run difference of profiles/r04/
// e_chain_rerun_difference.txt to exactly that co-residency (profiles/r05/README.md): with correct inputs in memory a chain
// workgroup of 20 - 100 KB returned a few wrong elements in 0.3 - 1.75 % of two-stream train steps - only when the other
// encoder branch's stream shared compute units with it (35 of 2,000 steps; 0 of 2,000 with disjoint CU masks), only with stock
// PyTorch attention kernels (rocBLAS GEMMs / ATen softmax) in the step (0 of 2,000 with this library's own), and never once
// it held the full 160 KB (0 of 2,000; 10 of 900 at 100 KB in round 4).  At the benchmark shape the slices are 131 - 150 KB
// already: one workgroup per CU either way.
constexpr size_t kChainLdsRequest = 160 * 1024;

struct ChainParams {
    const int32_t *ptr, *other;
    const float *w;
    const int32_t *deg_ptr;           // k_hop_chain_gcn: in-degree offsets, w[p] = d(dst)^-1/2 * d(src)^-1/2
    float *slab;
    int64_t ld;
    float *rowmax;
    int32_t cap;                      // elements of other / w (range check of the 16-byte id / weight loads)
    int F, nslices, K, src0, dir, rm_mode, nseg;
    int32_t node_ptr[kChainGraphs + 1];
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float chain_absmax(const float4 &v) {
    return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// this wave's LDS operations have completed (reads returned, writes landed), then the workgroup meets: no vmcnt wait -
// the block stores of a hop keep draining while the next hop computes (__syncthreads() would wait for them)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS through plain 32-bit byte addresses (base + offsets folded by hand: one v_lshl_add_u32 per neighbour piece)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_read4(unsigned addr) {
    const f32x4 v = *(const __attribute__((address_space(3))) f32x4 *)(uintptr_t)addr;
    return make_float4(v.x, v.y, v.z, v.w);
}

// a produced 16-byte piece leaves through a buffer descriptor over the graph's own rows: ONE unconditional instruction
// (rows past the graph's end fall outside the descriptor and are dropped by the range check), which hipcc can count -
// behind a store under a branch it waits for vmcnt(0), i.e. for the store itself, before every later use of a load
__device__ __forceinline__ void store_piece(const float4 &a, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    u32x4 av;
    av.x = __float_as_uint(a.x), av.y = __float_as_uint(a.y), av.z = __float_as_uint(a.z), av.w = __float_as_uint(a.w);
    if (DC_CHAIN_ABL & 1)
        asm volatile("" ::"v"(av.x), "v"(av.y), "v"(av.z), "v"(av.w));
    else
        __builtin_amdgcn_raw_buffer_store_b128(av, rs, off, 0, 0);
}

// 8 lanes x 8 rows of per-lane maxima -> lane `sub` holds the maximum of the row of step `sub` (a transposing
// butterfly); the slices of a row then meet in rowmax[row]: non-negative floats order like their bit patterns
__device__ __forceinline__ float rowmax_transpose(const float (&pm)[8], int sub) {
    float q[4], r[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = (sub & 1) ? pm[2 * i + 1] : pm[2 * i], send = (sub & 1) ? pm[2 * i] : pm[2 * i + 1];
        q[i] = fmaxf(keep, __shfl_xor(send, 1));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = (sub & 2) ? q[2 * i + 1] : q[2 * i], send = (sub & 2) ? q[2 * i] : q[2 * i + 1];
        r[i] = fmaxf(keep, __shfl_xor(send, 2));
    }
    const float keep = (sub & 4) ? r[1] : r[0], send = (sub & 4) ? r[0] : r[1];
    return fmaxf(keep, __shfl_xor(send, 4));
}

__device__ __forceinline__ void publish_rowmax(const float (&pm)[8], int sub, int steps, int row, int nn, float *rowmax) {
    const float rm = rowmax_transpose(pm, sub);
    if (DC_CHAIN_ABL & 256) return;
    if (sub < steps && row < nn) atomicMax(reinterpret_cast<int *>(rowmax + row), __float_as_int(rm));
}

// the same for LPR (2 or 4) lanes per row piece: one xor-butterfly over the row's lanes per step, lane sub == 0 publishes
template <int STEPS, int LPR>
__device__ __forceinline__ void publish_rowmax_narrow(const float (&pm)[8], int sub, int row0, int nn, float *rowmax) {
    constexpr int RPW = 64 / LPR;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        float m = pm[s];
#pragma unroll
        for (int d = 1; d < LPR; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
        const int row = row0 + RPW * s;
        if (sub == 0 && row < nn) atomicMax(reinterpret_cast<int *>(rowmax + row), __float_as_int(m));
    }
}

struct Chunk {                        // ids and weights of 8 consecutive edges
    u32x4 i0, i1, w0, w1;
};

template <bool W>
__device__ __forceinline__ void load_chunk(Chunk &c, __amdgpu_buffer_rsrc_t ro, __amdgpu_buffer_rsrc_t rw, int p) {
    // 4-byte aligned 16-byte loads, range-checked per dword against the arrays' size (past the end: 0)
    if (DC_CHAIN_ABL & 16) {
        c.i0 = c.i1 = c.w0 = c.w1 = u32x4{(unsigned)p, (unsigned)p, (unsigned)p, (unsigned)p} & 0u;
        return;
    }
    if (DC_CHAIN_ABL & 128) {                                    // ids made up from the edge offset: no VMEM load, spread rows
        const unsigned q = (unsigned)p / 6u;
        c.i0 = u32x4{(q + 1) & 511u, (q + 14) & 511u, (q + 27) & 511u, (q + 40) & 511u};
        c.i1 = u32x4{(q + 53) & 511u, (q + 66) & 511u, (q + 79) & 511u, (q + 92) & 511u};
        c.w0 = c.w1 = u32x4{0x3e000000u, 0x3e000000u, 0x3e000000u, 0x3e000000u};
        return;
    }
    c.i0 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * p, 0, 0);
    c.i1 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * p + 16, 0, 0);
    if (W) {
        c.w0 = __builtin_amdgcn_raw_buffer_load_b128(rw, 4 * p, 0, 0);
        c.w1 = __builtin_amdgcn_raw_buffer_load_b128(rw, 4 * p + 16, 0, 0);
    }
}

// NS neighbour slots of a chunk for the wave's 8 rows: every piece is requested from LDS before the first is used
// (one exposed LDS latency per step, not one per slot).  Slots >= MFROM are tested against `rem`, the neighbours the
// lane's row still has: past its end a slot reads the row of zeros with weight 0.  Slots < MFROM are known - by a
// wave-wide vote of the caller - to exist in all 8 rows: no test, one address instruction per piece.
//
// A neighbour id outside the workgroup's graph - only possible when the caller's layout is wrong (Batch.assume_segments on a
// batch that is not block-diagonal; the build flags such edges in its status word) - must not read another row of the slice or
// LDS that was never written: `off - win.lo < win.span` (unsigned, ids below the graph wrap around) sends it to the row
// of zeros with weight 0, like a slot past the end of the row.
struct LdsWindow {
    unsigned lo, span;               // byte address of this lane's piece of row 0, bytes of the graph's rows (nn * 128)
};

template <bool W, int NS, int MFROM, int J0 = 0, int RB = 128>
__device__ __forceinline__ void chunk_slots(float4 &acc, const Chunk &c, int rem, unsigned lbase, unsigned zsub,
                                            const LdsWindow win) {
    const unsigned id[8] = {c.i0.x, c.i0.y, c.i0.z, c.i0.w, c.i1.x, c.i1.y, c.i1.z, c.i1.w};
    const unsigned wb[8] = {c.w0.x, c.w0.y, c.w0.z, c.w0.w, c.w1.x, c.w1.y, c.w1.z, c.w1.w};
    float4 v[NS];
    float ww[NS];
    if (DC_CHAIN_ABL & 2) return;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        unsigned off = id[J0 + j] * (unsigned)RB + lbase;
        float wt = W ? __uint_as_float(wb[J0 + j]) : 1.0f;
        bool valid = off - win.lo < win.span;
        if (J0 + j >= MFROM) valid = valid && J0 + j < rem;
        off = valid ? off : zsub;
        wt = valid ? wt : 0.0f;
        v[j] = lds_read4(off);
        ww[j] = wt;
    }
    if (DC_CHAIN_ABL & 32) {
#pragma unroll
        for (int j = 0; j < NS; ++j) asm volatile("" ::"v"(v[j].x), "v"(v[j].y), "v"(v[j].z), "v"(v[j].w), "v"(ww[j]));
        return;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float mx = ww[j] * v[j].x, my = ww[j] * v[j].y, mz = ww[j] * v[j].z, mw = ww[j] * v[j].w;
        acc.x = acc.x + mx;
        acc.y = acc.y + my;
        acc.z = acc.z + mz;
        acc.w = acc.w + mw;
    }
}

// one step: the wave's 8 rows against the chunk of their first 8 neighbours, specialised by a wave-wide vote on the
// rows' degrees (triangle meshes: 5, 6 or 7 almost everywhere), and the rare longer rows chunk by chunk
template <bool W, int RB = 128>
__device__ __forceinline__ void step_rows(float4 &a, const Chunk &c, int pbeg, int rem, unsigned lbase, unsigned zsub,
                                          const LdsWindow win, __amdgpu_buffer_rsrc_t ro, __amdgpu_buffer_rsrc_t rw) {
    if (__all(rem >= 5) && !__any(rem > 7)) {
        if (__any(rem > 6))
            chunk_slots<W, 7, 5, 0, RB>(a, c, rem, lbase, zsub, win);
        else
            chunk_slots<W, 6, 5, 0, RB>(a, c, rem, lbase, zsub, win);
        return;
    }
    // the general path (padding rows, short or long rows) runs rarely: four pieces in flight keep its registers
    // below what the common paths need
    chunk_slots<W, 4, 0, 0, RB>(a, c, rem, lbase, zsub, win);
    if (__any(rem > 4)) chunk_slots<W, 4, 0, 4, RB>(a, c, rem, lbase, zsub, win);
    while (__any(rem > 8)) {                                     // rows with more than 8 neighbours (mesh poles, hubs)
        pbeg += 8, rem -= 8;
        Chunk n;
        load_chunk<W>(n, ro, rw, pbeg);
        chunk_slots<W, 4, 0, 0, RB>(a, n, rem, lbase, zsub, win);
        if (__any(rem > 4)) chunk_slots<W, 4, 0, 4, RB>(a, n, rem, lbase, zsub, win);
    }
}

// STEPS: row groups per wave = ceil(largest graph of the launch / rows per workgroup step), a compile-time constant: the
// step sequence is then straight-line code (no "s < steps" exits), and only in straight-line code does hipcc count the
// in-flight id / weight loads and block stores exactly (counted vmcnt(N) instead of vmcnt(0) - which would wait for the stores).
// LPR: lanes per row piece = slice width / 4 columns.  8 (32 columns, 128-byte pieces: whole cache lines) holds graphs of up
// to 1,024 nodes in the 160 KB of LDS; 4 (16 columns) up to 2,048, 2 (8 columns) up to 4,096 - the reference feeds whole
// meshes (configs/everyday.json:6 n_points = -1, loaders/everyday_deform.py:60-65), not only the ~1k-vertex ones of the
// benchmark shape.  A wave-instruction covers 64 / LPR rows; everything else is the same arithmetic in the same order.
template <bool W, int STEPS, int LPR>
__global__ void __launch_bounds__(1024)
k_hop_chain(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPW = 64 / LPR, RB = 16 * LPR, COLS = 4 * LPR;   // rows per wave-instruction, bytes / columns per row piece
    constexpr int R = 16 * RPW * STEPS;                          // rows of the LDS slice
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);      // the slices of a graph run side by side on one XCD
    const int seg = (int)(lb / (unsigned)p.nslices), slice = (int)(lb - (unsigned)seg * (unsigned)p.nslices);
    const int n0 = p.node_ptr[seg], nn = p.node_ptr[seg + 1] - n0;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, grp = lane / LPR, sub = lane % LPR;
    const int rwave = wid * RPW * STEPS;                        // the wave's rows: [rwave, rwave + RPW * STEPS)
    constexpr int zoff = R * RB;                                // a row of zeros behind the slice
    float *blk = p.slab + (int64_t)n0 * p.ld + slice * COLS + 4 * sub;

    // ---- source block slice -> LDS (one wave-instruction = RPW rows x RB bytes = 1 KiB, contiguous in LDS) ----
    {
        const float *src = blk + (int64_t)p.src0 * p.F;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int row = rwave + RPW * s + grp;
            if (row < nn && !(DC_CHAIN_ABL & 8))
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1))) *)(src + (int64_t)row * p.ld),
                    (void __attribute__((address_space(3))) *)(smem + (rwave + RPW * s) * RB), 16, 0, 0);
        }
    }
    if (threadIdx.x < LPR) *reinterpret_cast<float4 *>(smem + zoff + 16 * threadIdx.x) = make_float4(0.f, 0.f, 0.f, 0.f);
    // the graph's slice of `ptr` (first edge of every row; degree = the next entry minus it): the same for every hop, kept
    // in LDS behind the zero row (registers otherwise; a row's lanes read one address - a broadcast)
    int *ptrl = reinterpret_cast<int *>(smem + zoff + 128);
    for (int r = threadIdx.x; r <= R; r += 1024) ptrl[r] = p.ptr[n0 + (r < nn ? r : nn)];
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.other), 0, p.cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W ? p.w : reinterpret_cast<const float *>(p.other)), 0,
                                          p.cap * 4, 0x00020000);
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): the slice has landed (a wait hipcc can see:
    lds_barrier();                                               // behind an asm wait it drains vmcnt before every ds_read)

    // row maxima: every lane keeps the running maximum of ITS four columns of each of its rows; the lanes of a row piece
    // are joined once, at the end
    const bool want_rm = p.rowmax != nullptr;
    float pm[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        pm[s] = 0.f;
        if (s < STEPS && want_rm && (p.rm_mode & 1))             // the row's own piece of the source block
            pm[s] = chain_absmax(*reinterpret_cast<const float4 *>(smem + (rwave + RPW * s + grp) * RB + 16 * sub));
    }

    const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lbase = sbase + 16u * sub - ((DC_CHAIN_ABL & 128) ? 0u : (unsigned)n0 * (unsigned)RB);   // LDS address of a neighbour's piece: id * RB + lbase
    const unsigned zsub = sbase + zoff + 16u * sub;
    const LdsWindow win{sbase + 16u * sub, (unsigned)nn * (unsigned)RB};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.slab + (int64_t)n0 * p.ld, 0, (int)((unsigned)nn * (unsigned)p.ld * 4u), 0x00020000);
    const unsigned ldb = (unsigned)p.ld * 4u;
    for (int h = 0; h < p.K; ++h) {
        const unsigned dcol = (unsigned)((p.src0 + (h + 1) * p.dir) * p.F + slice * COLS + 4 * sub) * 4u;
        float4 acc[STEPS];
        Chunk ck[2];                                             // ids / weights: this step's and the next one's
        int2 bd[2];
        {
            const int b = ptrl[rwave + grp], e = ptrl[rwave + grp + 1];
            bd[0] = make_int2(b, e - b);
        }
        load_chunk<W>(ck[0], ro, rw, bd[0].x);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int pbeg = bd[s & 1].x, rem = bd[s & 1].y;
            if (s + 1 < STEPS) {                                 // one step ahead
                const int r1 = rwave + RPW * (s + 1) + grp;
                const int b = ptrl[r1], e = ptrl[r1 + 1];
                bd[(s + 1) & 1] = make_int2(b, e - b);
                load_chunk<W>(ck[(s + 1) & 1], ro, rw, b);
            }
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            step_rows<W, RB>(a, ck[s & 1], pbeg, rem, lbase, zsub, win, ro, rw);
            acc[s] = a;
            pm[s] = fmaxf(pm[s], chain_absmax(a));
            store_piece(a, rs, (unsigned)(rwave + RPW * s + grp) * ldb + dcol);             // streams out meanwhile
        }
        if (h + 1 == p.K) break;
        if (DC_CHAIN_ABL & 4) continue;
        lds_barrier();                                           // every wave has read what it needs of block h
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            *reinterpret_cast<float4 *>(smem + (rwave + RPW * s + grp) * RB + 16 * sub) = acc[s];
        lds_barrier();
    }
    if (want_rm) {
        if constexpr (LPR == 8)
            publish_rowmax(pm, sub, STEPS, rwave + 8 * sub + grp, nn, p.rowmax + n0);
        else
            publish_rowmax_narrow<STEPS, LPR>(pm, sub, rwave + grp, nn, p.rowmax + n0);
    }
}

// ---- gcn_norm weights: adjacency resident in LDS too -------------------------------------------------------------------
// With w[p] = dis[source] * dis[destination], dis = in-degree^-1/2 (gcn_norm without self loops: what TAGConv hops
// with, PyG tag_conv.py), the weights need not be loaded: dis (4 B per node) and the first 8 neighbour ids of every row
// (2 B each, local to the graph, rows shorter than 8 padded with the index of the row of zeros, whose dis is 0) fit
// behind the slice.  The hop loop then issues NO vector-memory load: vmcnt retires in order, so with the ids arriving
// by buffer loads every step's id wait also waited for the older block stores, and the compute ran in series with
// the write path (r04 ablation: stores and slots cost 54 + 47 us of a 148 us step and did not overlap; without
// in-loop loads 108 us).  Rows with more than 8 neighbours (mesh poles, hubs) take their tail from global memory.
template <int NS, int J0 = 0, int RB = 128>
__device__ __forceinline__ void gcn_slots(float4 &acc, const uint4 &iv, float di, unsigned lbase, unsigned dbase) {
    const unsigned pk[4] = {iv.x, iv.y, iv.z, iv.w};
    float4 v[NS];
    float ww[NS];
    if (DC_CHAIN_ABL & 2) return;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const unsigned id = ((J0 + j) & 1) ? pk[(J0 + j) >> 1] >> 16 : pk[(J0 + j) >> 1] & 0xffffu;
        v[j] = lds_read4(id * (unsigned)RB + lbase);
        const float dj = *(const __attribute__((address_space(3))) float *)(uintptr_t)(id * 4u + dbase);
        ww[j] = dj * di;                                         // dis[source] * dis[destination] (a row's ids are its sources
    }                                                            // in the forward set, its destinations in the transposed one)
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const float mx = ww[j] * v[j].x, my = ww[j] * v[j].y, mz = ww[j] * v[j].z, mw = ww[j] * v[j].w;
        acc.x = acc.x + mx;
        acc.y = acc.y + my;
        acc.z = acc.z + mz;
        acc.w = acc.w + mw;
    }
}

template <int STEPS>
__global__ void __launch_bounds__(1024)
k_hop_chain_gcn(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = 128 * STEPS;                              // rows of the LDS slice; row R is the row of zeros
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = (int)(lb / (unsigned)p.nslices), slice = (int)(lb - (unsigned)seg * (unsigned)p.nslices);
    const int n0 = p.node_ptr[seg], nn = p.node_ptr[seg + 1] - n0;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    const int rwave = wid * 8 * STEPS;
    float *blk = p.slab + (int64_t)n0 * p.ld + slice * kChainCols + 4 * sub;
    // LDS: slice [R][128 B] | zeros [128 B] | ids [R][8] u16 | dis [R + 1] f32 (16-byte padded) | {first edge, degree} [R]
    constexpr int kIds = (R + 1) * 128, kDis = kIds + R * 16, kBounds = kDis + ((R + 1) * 4 + 15) / 16 * 16;
#ifdef DC_CHAIN_POISON
    // diagnostic build (tools/r05): every byte of the workgroup's LDS starts as a quiet NaN, so that a read of LDS this
    // launch has not written yet cannot pass for data
    for (int i = threadIdx.x; i < (kBounds + R * 8) / 16; i += 1024)
        reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u);
    __syncthreads();
#endif
    // prologue, by wave role: waves 8-15 issue ALL the LDS-DMA of the slice (one instruction = 8 rows x 128 B, contiguous
    // in LDS), waves 0-7 build the tables.  hipcc makes a wave with LDS-DMA in flight wait for vmcnt(0) before each of
    // its own LDS accesses: with both jobs in every wave the tables' loads were only issued once the slice had landed.
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.other), 0, p.cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.cap * 4, 0x00020000);
    if (wid >= 8) {
        const float *src = blk + (int64_t)p.src0 * p.F;
#pragma unroll
        for (int s = 0; s < 2 * STEPS; ++s) {
            const int r0 = (wid - 8) * 16 * STEPS + 8 * s;      // the rows of waves 2 (wid - 8) and 2 (wid - 8) + 1
#ifdef DC_CHAIN_REGSTAGE
            // diagnostic build (tools/r06/chain_variants.sh): the same bytes through registers (global_load_dwordx4 +
            // ds_write_b128) instead of LDS-DMA - does the rare difference need the DMA path?
            if (r0 + grp < nn)
                *reinterpret_cast<float4 *>(smem + r0 * 128 + 16 * lane) =
                    *reinterpret_cast<const float4 *>(src + (int64_t)(r0 + grp) * p.ld);
#else
            if (r0 + grp < nn && !(DC_CHAIN_ABL & 8))
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1))) *)(src + (int64_t)(r0 + grp) * p.ld),
                    (void __attribute__((address_space(3))) *)(smem + r0 * 128), 16, 0, 0);
#endif
        }
#ifdef DC_CHAIN_DMA_READBACK
        // diagnostic build: behind its own vmcnt(0) every DMA wave reads back the last piece it brought in, before the
        // barrier releases the other waves' reads
        __builtin_amdgcn_s_waitcnt(0x0F70);
        {
            const int r_last = (wid - 8) * 16 * STEPS + 8 * (2 * STEPS - 1);
            const float4 v = *reinterpret_cast<const float4 *>(smem + r_last * 128 + 16 * lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
        }
#endif
    } else {
        if (threadIdx.x < 8)
            *reinterpret_cast<float4 *>(smem + R * 128 + 16 * threadIdx.x) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = threadIdx.x; r <= R; r += 512) {           // bounds, dis, padded local ids of one row
            const bool live = r < nn;
            const int b = live ? p.ptr[n0 + r] : 0, d = (live ? p.ptr[n0 + r + 1] : 0) - b;
            const int din = live ? p.deg_ptr[n0 + r + 1] - p.deg_ptr[n0 + r] : 0;
            *reinterpret_cast<float *>(smem + kDis + 4 * r) = inv_sqrt_count(din);
            if (r == R) break;
            *reinterpret_cast<int2 *>(smem + kBounds + 8 * r) = make_int2(b, d);
            const u32x4 i0 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * b, 0, 0);
            const u32x4 i1 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * b + 16, 0, 0);
            const unsigned g[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
            unsigned l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {                        // a neighbour outside the graph (flagged by the build)
                const unsigned loc = g[j] - (unsigned)n0;       // must not leave the LDS tables: it reads the zeros
                l[j] = (j < d && loc < (unsigned)nn) ? loc : (unsigned)R;
            }
            *reinterpret_cast<uint4 *>(smem + kIds + 16 * r) =
                make_uint4(l[0] | l[1] << 16, l[2] | l[3] << 16, l[4] | l[5] << 16, l[6] | l[7] << 16);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): slice and tables have landed
    lds_barrier();

    const bool want_rm = p.rowmax != nullptr;
    float pm[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        pm[s] = 0.f;
        if (s < STEPS && want_rm && (p.rm_mode & 1))
            pm[s] = chain_absmax(*reinterpret_cast<const float4 *>(smem + (rwave + 8 * s + grp) * 128 + 16 * sub));
    }
    const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned lbase = sbase + 16u * sub, dbase = sbase + kDis;
    const unsigned gbase = lbase - (unsigned)n0 * 128u, zsub = lbase + R * 128u;    // global-id addressing of the tail path
    const LdsWindow win{lbase, (unsigned)nn * 128u};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.slab + (int64_t)n0 * p.ld, 0, (int)((unsigned)nn * (unsigned)p.ld * 4u), 0x00020000);
    const unsigned ldb = (unsigned)p.ld * 4u;
    for (int h = 0; h < p.K; ++h) {
        const unsigned dcol = (unsigned)((p.src0 + (h + 1) * p.dir) * p.F + slice * kChainCols + 4 * sub) * 4u;
        float4 acc[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            int row = rwave + 8 * s + grp;
            // (the tables never change, and hipcc knows: it would keep all 8 steps' entries - 56 registers - live across
            // the hops and spill; an opaque row index keeps the three reads where they are)
            asm volatile("" : "+v"(row));
            const uint4 iv = *reinterpret_cast<const uint4 *>(smem + kIds + 16 * row);
            const int2 bd = *reinterpret_cast<const int2 *>(smem + kBounds + 8 * row);
            const float di = *reinterpret_cast<const float *>(smem + kDis + 4 * row);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!__any(bd.y > 6))
                gcn_slots<6>(a, iv, di, lbase, dbase);
            else if (!__any(bd.y > 7))
                gcn_slots<7>(a, iv, di, lbase, dbase);
            else {
                gcn_slots<4, 0>(a, iv, di, lbase, dbase);        // (rare: two halves keep the registers of this path
                gcn_slots<4, 4>(a, iv, di, lbase, dbase);        // below those of the common ones)
                int pbeg = bd.x, rem = bd.y;
                while (__any(rem > 8)) {                         // the tail of long rows: ids and weights from memory
                    pbeg += 8, rem -= 8;
                    Chunk n;
                    load_chunk<true>(n, ro, rw, pbeg);
                    chunk_slots<true, 4, 0, 0>(a, n, rem, gbase, zsub, win);
                    if (__any(rem > 4)) chunk_slots<true, 4, 0, 4>(a, n, rem, gbase, zsub, win);
                }
            }
            acc[s] = a;
            pm[s] = fmaxf(pm[s], chain_absmax(a));
            store_piece(a, rs, (unsigned)row * ldb + dcol);      // streams out while the next steps compute
        }
        if (h + 1 == p.K) break;
        if (DC_CHAIN_ABL & 4) continue;
        lds_barrier();
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            *reinterpret_cast<float4 *>(smem + (rwave + 8 * s + grp) * 128 + 16 * sub) = acc[s];
        lds_barrier();
    }
    if (want_rm) publish_rowmax(pm, sub, STEPS, rwave + 8 * sub + grp, nn, p.rowmax + n0);
#ifdef DC_CHAIN_POISON_END
    // diagnostic build (tools/r05): leave quiet NaNs behind in the slice, so that a LATER workgroup on this CU that reads LDS
    // it has not (yet) written cannot pass for data of the right magnitude; the prologue's timing stays what it is
    lds_barrier();
    for (int i = threadIdx.x; i < (R + 1) * 8; i += 1024)
        reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u);
#endif
}

template <int STEPS>
static bool launch_chain_gcn_steps(unsigned grid, hipStream_t stream, const ChainParams &p) {
    constexpr int R = 128 * STEPS;
    constexpr size_t lds = (size_t)(R + 1) * 128 + (size_t)R * 16 + ((size_t)(R + 1) * 4 + 15) / 16 * 16 + (size_t)R * 8;
    static_assert(lds <= 160 * 1024, "k_hop_chain_gcn: tables do not fit the LDS");
#ifdef DC_CHAIN_LDS_TIGHT
    constexpr size_t lds_req = lds;              // diagnostic build (tools/r05): only what the slice needs, as up to round 4
#else
    constexpr size_t lds_req = kChainLdsRequest;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hop_chain_gcn<STEPS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req) != hipSuccess)
            return false;
        attr_set = true;
    }
    static const std::string name = "k_hop_chain_gcn<" + std::to_string(STEPS) + ">";
    trace_kernel(name.c_str());
    hipLaunchKernelGGL((k_hop_chain_gcn<STEPS>), dim3(grid), dim3(1024), lds_req, stream, p);
    return true;
}

static bool launch_chain_gcn(int steps, unsigned grid, hipStream_t stream, const ChainParams &p) {
    switch (steps) {
    case 1: return launch_chain_gcn_steps<1>(grid, stream, p);
    case 2: return launch_chain_gcn_steps<2>(grid, stream, p);
    case 3: return launch_chain_gcn_steps<3>(grid, stream, p);
    case 4: return launch_chain_gcn_steps<4>(grid, stream, p);
    case 5: return launch_chain_gcn_steps<5>(grid, stream, p);
    case 6: return launch_chain_gcn_steps<6>(grid, stream, p);
    case 7: return launch_chain_gcn_steps<7>(grid, stream, p);
    default: return launch_chain_gcn_steps<8>(grid, stream, p);
    }
}

template <bool W, int STEPS, int LPR>
static bool launch_chain_steps(unsigned grid, hipStream_t stream, const ChainParams &p) {
    // slice + zero row (in a 128-byte slot) + the graph's slice of ptr
    constexpr int R = 16 * (64 / LPR) * STEPS;
    constexpr size_t lds = (size_t)R * 16 * LPR + 128 + ((size_t)(R + 1) * 4 + 15) / 16 * 16;
    static_assert(lds <= kChainLdsRequest, "k_hop_chain: slice does not fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hop_chain<W, STEPS, LPR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLdsRequest) != hipSuccess)
            return false;
        attr_set = true;
    }
    static const std::string name = std::string("k_hop_chain<") + (W ? "true, " : "false, ") + std::to_string(STEPS) + ", " +
                                    std::to_string(LPR) + ">";
    trace_kernel(name.c_str());
    hipLaunchKernelGGL((k_hop_chain<W, STEPS, LPR>), dim3(grid), dim3(1024), kChainLdsRequest, stream, p);
    return true;
}

template <bool W, int LPR>
static bool launch_chain(int steps, unsigned grid, hipStream_t stream, const ChainParams &p) {
    switch (steps) {
    case 1: return launch_chain_steps<W, 1, LPR>(grid, stream, p);
    case 2: return launch_chain_steps<W, 2, LPR>(grid, stream, p);
    case 3: return launch_chain_steps<W, 3, LPR>(grid, stream, p);
    case 4: return launch_chain_steps<W, 4, LPR>(grid, stream, p);
    case 5: return launch_chain_steps<W, 5, LPR>(grid, stream, p);
    case 6: return launch_chain_steps<W, 6, LPR>(grid, stream, p);
    case 7: return launch_chain_steps<W, 7, LPR>(grid, stream, p);
    default: return launch_chain_steps<W, 8, LPR>(grid, stream, p);
    }
}

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_hop_chain_lds_request(void) { return (int64_t)kChainLdsRequest; }

extern "C" int64_t dc_hop_chain_max_nodes(void) { return 4 * 128 * kChainSteps; }      // 8-column slices: 4,096 nodes

extern "C" int dc_hop_chain_f32(const int32_t *ptr, const int32_t *other, const float *w, const int32_t *deg_ptr,
                                int64_t cap, const int64_t *node_ptr_host, int nseg, float *slab, int64_t ld, int64_t N,
                                int64_t F, int K, int src_block, int dir, float *rowmax, int mode,
                                dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && F >= 1 && K >= 0 && nseg >= 0, "dc_hop_chain_f32: negative size");
    if (N == 0 || K == 0 || nseg == 0) return DC_OK;
    DC_REQUIRE(ptr && other && slab && node_ptr_host, "dc_hop_chain_f32: null pointer");
    DC_REQUIRE(dir == 1 || dir == -1, "dc_hop_chain_f32: dir must be +1 or -1");
    DC_REQUIRE((mode & ~3) == 0, "dc_hop_chain_f32: mode is a 2-bit mask");
    DC_REQUIRE(F % kChainCols == 0 && ld % 4 == 0 && ((uintptr_t)slab & 15) == 0,
               "dc_hop_chain_f32: needs F %% 32 == 0 and 16-byte aligned rows (F=%lld, ld=%lld)", (long long)F,
               (long long)ld);
    DC_REQUIRE(src_block >= 0 && src_block + K * dir >= 0 && (int64_t)(src_block + 1) * F <= ld &&
                   (int64_t)(src_block + K * dir + 1) * F <= ld,
               "dc_hop_chain_f32: column blocks outside the slab");
    DC_REQUIRE(cap >= 0 && cap < (int64_t)1 << 29 && N < (int64_t)1 << 24,
               "dc_hop_chain_f32: adjacency of %lld edges / %lld nodes exceeds the 32-bit LDS / buffer offsets",
               (long long)cap, (long long)N);
    DC_REQUIRE(node_ptr_host[0] == 0 && node_ptr_host[nseg] == N, "dc_hop_chain_f32: the graphs' offsets must cover [0, N]");
    for (int i = 0; i < nseg; ++i) {
        const int64_t dn = node_ptr_host[i + 1] - node_ptr_host[i];
        DC_REQUIRE(dn >= 0 && dn <= 4 * 128 * kChainSteps,
                   "dc_hop_chain_f32: graph %d has %lld nodes (cap %d): use dc_spmm_f32 hop by hop", i, (long long)dn,
                   4 * 128 * kChainSteps);
    }
    if (rowmax && !(mode & 2)) {
        if (hipMemsetAsync(rowmax, 0, (size_t)N * sizeof(float), stream) != hipSuccess)
            return check_launch("dc_hop_chain_f32 (memset)");
    }
    ChainParams p{};
    DC_REQUIRE(!deg_ptr || w, "dc_hop_chain_f32: deg_ptr describes the weights w - it cannot come without them");
    p.ptr = ptr, p.other = other, p.w = w, p.deg_ptr = deg_ptr, p.slab = slab, p.ld = ld, p.rowmax = rowmax, p.cap = (int32_t)cap;
    p.F = (int)F, p.nslices = (int)(F / kChainCols), p.K = K, p.src0 = src_block, p.dir = dir, p.rm_mode = mode;
    for (int s0 = 0; s0 < nseg; s0 += kChainGraphs) {
        const int cnt = nseg - s0 < kChainGraphs ? nseg - s0 : kChainGraphs;
        int64_t big = 0;
        for (int i = 0; i <= cnt; ++i) p.node_ptr[i] = (int32_t)node_ptr_host[s0 + i];
        for (int i = 0; i < cnt; ++i) {
            const int64_t dn = node_ptr_host[s0 + i + 1] - node_ptr_host[s0 + i];
            big = dn > big ? dn : big;
        }
        if (big == 0) continue;
        // a graph's rows are addressed through one buffer resource with 32-bit byte offsets (row * ld * 4 + column): the
        // resource size and every store offset must stay below 2^31, or the bounds-checked stores are dropped silently
        DC_REQUIRE(big * ld * 4 < ((int64_t)1 << 31),
                   "dc_hop_chain_f32: a graph of %lld nodes with leading dimension %lld exceeds the 32-bit buffer offsets "
                   "(nodes * ld * 4 must stay below 2^31)", (long long)big, (long long)ld);
        p.nseg = cnt;
        // slice width by the largest graph of the launch: 32 columns up to 1,024 nodes, 16 up to 2,048, 8 up to 4,096
        const int lpr = big <= 128 * kChainSteps ? 8 : big <= 2 * 128 * kChainSteps ? 4 : 2;
        const int rows_per_step = 16 * (64 / lpr);
        const int smax = (int)((big + rows_per_step - 1) / rows_per_step);
        p.nslices = (int)(F / (4 * lpr));
        const unsigned grid = (unsigned)cnt * (unsigned)p.nslices;
        const bool ok = (deg_ptr && lpr == 8) ? launch_chain_gcn(smax, grid, stream, p)
                        : lpr == 8 ? (w ? launch_chain<true, 8>(smax, grid, stream, p) : launch_chain<false, 8>(smax, grid, stream, p))
                        : lpr == 4 ? (w ? launch_chain<true, 4>(smax, grid, stream, p) : launch_chain<false, 4>(smax, grid, stream, p))
                                   : (w ? launch_chain<true, 2>(smax, grid, stream, p) : launch_chain<false, 2>(smax, grid, stream, p));
        DC_REQUIRE(ok, "dc_hop_chain_f32: cannot reserve the kernel's LDS");
    }
    return check_launch("dc_hop_chain_f32");
}
