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

#include "dc_common.h"

#pragma clang fp contract(off)

namespace dc {

constexpr int kChainCols = 32;        // columns per slice: 128-byte row pieces, one full cache line per store
constexpr int kChainSteps = 8;        // 16 waves x 8 row groups x 8 steps = DC_CHAIN_MAX_NODES nodes per graph
constexpr int kChainGraphs = 96;      // graphs per launch (their node offsets travel as kernel arguments)

struct ChainParams {
    const int32_t *ptr, *other;
    const float *w;
    float *slab;
    int64_t ld;
    float *rowmax;
    int32_t cap;                      // elements of other / w (range check of the 16-byte id / weight loads)
    int F, nslices, K, src0, dir, rm_mode, nseg;
    int32_t node_ptr[kChainGraphs + 1];
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float chain_dpp_max(float v) {
    const int i = __float_as_int(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float chain_absmax(const float4 &v) {
    return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// this wave's LDS operations have completed (reads returned, writes landed), then the workgroup meets: no vmcnt wait -
// the block stores of a hop keep draining while the next hop computes (__syncthreads() would wait for them)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// maximum of m over the 8 lanes of a row piece, joined into the lane that `keep` selects
__device__ __forceinline__ float group_max_into(float run, float m, bool keep) {
    m = chain_dpp_max<0xB1>(m);                                  // quad_perm [1,0,3,2]
    m = chain_dpp_max<0x4E>(m);                                  // quad_perm [2,3,0,1]
    m = chain_dpp_max<0x141>(m);                                 // row_half_mirror: all 8 lanes hold the maximum
    return keep ? fmaxf(run, m) : run;
}

struct Chunk {                        // ids and weights of 8 consecutive edges
    u32x4 i0, i1, w0, w1;
};

template <bool W>
__device__ __forceinline__ void load_chunk(Chunk &c, __amdgpu_buffer_rsrc_t ro, __amdgpu_buffer_rsrc_t rw, int p) {
    // 4-byte aligned 16-byte loads, range-checked per dword against the arrays' size (past the end: 0)
    c.i0 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * p, 0, 0);
    c.i1 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * p + 16, 0, 0);
    if (W) {
        c.w0 = __builtin_amdgcn_raw_buffer_load_b128(rw, 4 * p, 0, 0);
        c.w1 = __builtin_amdgcn_raw_buffer_load_b128(rw, 4 * p + 16, 0, 0);
    }
}

template <bool W>
__device__ __forceinline__ void slot(float4 &acc, unsigned id, unsigned wbits, bool valid, int lbase, int zsub,
                                     const char *smem) {
    const int off = valid ? (int)(id * 128u) + lbase : zsub;
    const float ww = valid ? (W ? __uint_as_float(wbits) : 1.0f) : 0.0f;
    const float4 v = *reinterpret_cast<const float4 *>(smem + off);
    const float mx = ww * v.x, my = ww * v.y, mz = ww * v.z, mw = ww * v.w;
    acc.x = acc.x + mx;
    acc.y = acc.y + my;
    acc.z = acc.z + mz;
    acc.w = acc.w + mw;
}

// the up-to-8 neighbours of a chunk for the wave's 8 rows (rem = neighbours the lane's row still has)
template <bool W>
__device__ __forceinline__ void chunk_slots(float4 &acc, const Chunk &c, int rem, int lbase, int zsub,
                                            const char *smem) {
    slot<W>(acc, c.i0.x, c.w0.x, 0 < rem, lbase, zsub, smem);
    slot<W>(acc, c.i0.y, c.w0.y, 1 < rem, lbase, zsub, smem);
    slot<W>(acc, c.i0.z, c.w0.z, 2 < rem, lbase, zsub, smem);
    slot<W>(acc, c.i0.w, c.w0.w, 3 < rem, lbase, zsub, smem);
    slot<W>(acc, c.i1.x, c.w1.x, 4 < rem, lbase, zsub, smem);
    if (__any(rem > 5)) {
        slot<W>(acc, c.i1.y, c.w1.y, 5 < rem, lbase, zsub, smem);
        if (__any(rem > 6)) {
            slot<W>(acc, c.i1.z, c.w1.z, 6 < rem, lbase, zsub, smem);
            if (__any(rem > 7)) slot<W>(acc, c.i1.w, c.w1.w, 7 < rem, lbase, zsub, smem);
        }
    }
}

// STEPS: row groups per wave = ceil(largest graph of the launch / 128), a compile-time constant: the step sequence
// is then straight-line code (no "s < steps" exits), and only in straight-line code does hipcc count the in-flight
// id / weight loads and block stores exactly (counted vmcnt(N) instead of vmcnt(0) - which would wait for the stores)
template <bool W, int STEPS>
__global__ void __launch_bounds__(1024)
k_hop_chain(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);      // the slices of a graph run side by side on one XCD
    const int seg = (int)(lb / (unsigned)p.nslices), slice = (int)(lb - (unsigned)seg * (unsigned)p.nslices);
    const int n0 = p.node_ptr[seg], nn = p.node_ptr[seg + 1] - n0;
    constexpr int steps = STEPS;                                // nn <= 128 * STEPS (host-checked)
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    const int rwave = wid * 8 * steps;                          // the wave's rows: [rwave, rwave + 8 steps)
    const int zoff = steps * 128 * 128;                         // a row of zeros behind the slice
    float *blk = p.slab + (int64_t)n0 * p.ld + slice * kChainCols + 4 * sub;

    // ---- source block slice -> LDS (one wave-instruction = 8 rows x 128 B, contiguous in LDS) ----
    {
        const float *src = blk + (int64_t)p.src0 * p.F;
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            if (s < steps) {
                const int row = rwave + 8 * s + grp;
                if (row < nn)
                    __builtin_amdgcn_global_load_lds(
                        (const void __attribute__((address_space(1))) *)(src + (int64_t)row * p.ld),
                        (void __attribute__((address_space(3))) *)(smem + (rwave + 8 * s) * 128), 16, 0, 0);
            }
    }
    if (threadIdx.x < 8) *reinterpret_cast<float4 *>(smem + zoff + 16 * threadIdx.x) = make_float4(0.f, 0.f, 0.f, 0.f);
    // segment bounds {first edge, degree} of every row: the same for every hop, kept in LDS behind the zero row
    // (16 registers per lane otherwise; a row's 8 lanes read one address - a broadcast)
    int2 *bounds = reinterpret_cast<int2 *>(smem + zoff + 128);
    for (int r = threadIdx.x; r < 128 * steps; r += 1024) {
        const int b = r < nn ? p.ptr[n0 + r] : 0, e = r < nn ? p.ptr[n0 + r + 1] : 0;
        bounds[r] = make_int2(b, e - b);
    }
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.other), 0, p.cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W ? p.w : reinterpret_cast<const float *>(p.other)), 0,
                                          p.cap * 4, 0x00020000);
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): the slice has landed (a wait hipcc can see:
    lds_barrier();                                               // behind an asm wait it drains vmcnt before every ds_read)

    // row maxima: lane `sub` of a row's lane group keeps the running maximum of the group's row of step `sub`
    const bool want_rm = p.rowmax != nullptr;
    float rm = 0.f;
    if (want_rm && (p.rm_mode & 1)) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            if (s < steps)                                       // the row's own piece of the source block
                rm = group_max_into(rm, chain_absmax(*reinterpret_cast<const float4 *>(
                                            smem + (rwave + 8 * s + grp) * 128 + 16 * sub)), sub == s);
    }

    const int lbase = 16 * sub - n0 * 128;                      // LDS byte offset of neighbour id: id * 128 + lbase
    const int zsub = zoff + 16 * sub;
    // the produced rows leave through a buffer descriptor over the graph's own rows: a store is ONE unconditional
    // instruction (rows past the graph's end fall outside the descriptor and are dropped by the range check), so
    // hipcc can count it: behind a store under a branch it waits for vmcnt(0) - i.e. for the store - before every
    // use of the next step's ids
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.slab + (int64_t)n0 * p.ld, 0, (int)((unsigned)nn * (unsigned)p.ld * 4u), 0x00020000);
    const unsigned ldb = (unsigned)p.ld * 4u;
    for (int h = 0; h < p.K; ++h) {
        const unsigned dcol = (unsigned)((p.src0 + (h + 1) * p.dir) * p.F + slice * kChainCols + 4 * sub) * 4u;
        float4 acc[STEPS];
        Chunk ck[2];                                             // ids / weights: this step's and the next one's
        int2 bd[2];
        bd[0] = bounds[rwave + grp];
        load_chunk<W>(ck[0], ro, rw, bd[0].x);
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            if (s < steps) {
                const int pbeg = bd[s & 1].x;
                int rem = bd[s & 1].y;
                if (s + 1 < STEPS) {                             // one step ahead
                    bd[(s + 1) & 1] = bounds[rwave + 8 * (s + 1) + grp];
                    load_chunk<W>(ck[(s + 1) & 1], ro, rw, bd[(s + 1) & 1].x);
                }
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                chunk_slots<W>(a, ck[s & 1], rem, lbase, zsub, smem);
                if (__any(rem > 8)) {                            // rows with more than 8 neighbours (mesh poles, hubs)
                    int pb = pbeg;
                    do {
                        pb += 8, rem -= 8;
                        Chunk c;
                        load_chunk<W>(c, ro, rw, pb);
                        chunk_slots<W>(a, c, rem, lbase, zsub, smem);
                    } while (__any(rem > 8));
                }
                acc[s] = a;
                const unsigned row = (unsigned)(rwave + 8 * s + grp);
                u32x4 av;
                av.x = __float_as_uint(a.x), av.y = __float_as_uint(a.y), av.z = __float_as_uint(a.z), av.w = __float_as_uint(a.w);
                __builtin_amdgcn_raw_buffer_store_b128(av, rs, row * ldb + dcol, 0, 0);      // streams out meanwhile
                if (want_rm) rm = group_max_into(rm, chain_absmax(a), sub == s);
            }
        if (h + 1 == p.K) break;
        lds_barrier();                                           // every wave has read what it needs of block h
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            if (s < steps) *reinterpret_cast<float4 *>(smem + (rwave + 8 * s + grp) * 128 + 16 * sub) = acc[s];
        lds_barrier();
    }
    if (want_rm) {
        // the slices of a row meet in rowmax[row]: non-negative floats order like their bit patterns
        const int row = rwave + 8 * sub + grp;
        if (sub < steps && row < nn) atomicMax(reinterpret_cast<int *>(p.rowmax + n0 + row), __float_as_int(rm));
    }
}

template <bool W, int STEPS>
static bool launch_chain_steps(unsigned grid, hipStream_t stream, const ChainParams &p) {
    // slice + zero row + {first edge, degree} per row
    constexpr size_t lds = ((size_t)128 * STEPS + 1) * 128 + (size_t)128 * STEPS * 8;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hop_chain<W, STEPS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_hop_chain<W, STEPS>), dim3(grid), dim3(1024), lds, stream, p);
    return true;
}

template <bool W>
static bool launch_chain(int steps, unsigned grid, hipStream_t stream, const ChainParams &p) {
    switch (steps) {
    case 1: return launch_chain_steps<W, 1>(grid, stream, p);
    case 2: return launch_chain_steps<W, 2>(grid, stream, p);
    case 3: return launch_chain_steps<W, 3>(grid, stream, p);
    case 4: return launch_chain_steps<W, 4>(grid, stream, p);
    case 5: return launch_chain_steps<W, 5>(grid, stream, p);
    case 6: return launch_chain_steps<W, 6>(grid, stream, p);
    case 7: return launch_chain_steps<W, 7>(grid, stream, p);
    default: return launch_chain_steps<W, 8>(grid, stream, p);
    }
}

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_hop_chain_max_nodes(void) { return 128 * kChainSteps; }

extern "C" int dc_hop_chain_f32(const int32_t *ptr, const int32_t *other, const float *w, int64_t cap,
                                const int64_t *node_ptr_host, int nseg, float *slab, int64_t ld, int64_t N,
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
        DC_REQUIRE(dn >= 0 && dn <= 128 * kChainSteps,
                   "dc_hop_chain_f32: graph %d has %lld nodes (cap %d): use dc_spmm_f32 hop by hop", i, (long long)dn,
                   128 * kChainSteps);
    }
    if (rowmax && !(mode & 2)) {
        if (hipMemsetAsync(rowmax, 0, (size_t)N * sizeof(float), stream) != hipSuccess)
            return check_launch("dc_hop_chain_f32 (memset)");
    }
    ChainParams p{};
    p.ptr = ptr, p.other = other, p.w = w, p.slab = slab, p.ld = ld, p.rowmax = rowmax, p.cap = (int32_t)cap;
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
        p.nseg = cnt;
        const int smax = (int)((big + 127) / 128);
        const unsigned grid = (unsigned)cnt * (unsigned)p.nslices;
        const bool ok = w ? launch_chain<true>(smax, grid, stream, p) : launch_chain<false>(smax, grid, stream, p);
        DC_REQUIRE(ok, "dc_hop_chain_f32: cannot reserve the kernel's LDS");
    }
    return check_launch("dc_hop_chain_f32");
}
