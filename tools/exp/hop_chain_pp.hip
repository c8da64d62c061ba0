// tools/exp/hop_chain_pp.hip -- round-5 experiment, NOT part of the library (fragment of dc_hopchain.hip: needs its ChainParams,
// gcn_slots, chunk_slots, store_piece, lds_barrier, ... to compile; it was built and run inside the library, DC_HOP_CHAIN_PP=1).
// Bit-identical to k_hop_chain_gcn (tests/test_hop_chain.py, all row-maxima modes, both directions) and SLOWER:
// the four chain launches of a step 169.9 us against 130.2 us (soft forward 39.2 vs 31.2 us), headline 0.640 - 0.644 ms against
// 0.622 - 0.628 ms on the same box (profiles/r05/g_chain_pingpong_store_waves.txt).  Taking the block stores off the gathering
// waves does not buy the overlap r04's ablation suggested: 16-column slices mean 512 workgroups in two rounds on 256 CUs, 64-byte
// store pieces, and 12 instead of 16 gathering waves.
// ---- the LDS-table form with the block stores taken OFF the computing waves (round 5) -------------------------------------------
// r04's ablation of k_hop_chain_gcn (profiles/r04/d_hop_chain_ablation.txt, soft forward chain): staging + tables 12 us, the
// three hops' LDS gathers alone +12.5 us, the three blocks' stores alone +12 us - and together +22 us: a wave issues in order,
// so a block store that waits for room in the memory pipeline holds up the gathers behind it.  Here the two are different
// waves.  A workgroup owns (graph, 16-column slice); the slice lives in LDS TWICE (64-byte row pieces: 2 x 64 KB for 1,024
// nodes): 12 waves gather hop h from one buffer and write the new rows straight into the other (no rows held in registers, ONE
// barrier per hop instead of two), while 4 waves stream the block the previous hop produced - the buffer being read - out to
// memory.  The last block leaves through all 16 waves.  Arithmetic and order are k_hop_chain_gcn's: bit-identical blocks and
// row maxima.
constexpr int kPpCompute = 12, kPpStore = 4;                     // waves by role

struct PpLayout {                                                // byte offsets inside the workgroup's LDS
    int buf[2], ids, dis, bounds, total;
};
__host__ __device__ inline PpLayout pp_layout(int rp) {
    PpLayout l;
    l.buf[0] = 0;
    l.buf[1] = (rp + 1) * 64;                                    // each buffer: rp rows + a row of zeros
    l.ids = 2 * (rp + 1) * 64;
    l.dis = l.ids + rp * 16;
    l.bounds = l.dis + ((rp + 1) * 4 + 15) / 16 * 16;
    l.total = l.bounds + rp * 8;
    return l;
}

__global__ void __launch_bounds__(1024)
k_hop_chain_pp(ChainParams p, int rp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RB = 64, COLS = 16;
    const PpLayout L = pp_layout(rp);
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = (int)(lb / (unsigned)p.nslices), slice = (int)(lb - (unsigned)seg * (unsigned)p.nslices);
    const int n0 = p.node_ptr[seg], nn = p.node_ptr[seg + 1] - n0;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, grp = lane >> 2, sub = lane & 3;
    const bool computes = wid < kPpCompute;
    float *blk = p.slab + (int64_t)n0 * p.ld + slice * COLS + 4 * sub;
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.other), 0, p.cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.cap * 4, 0x00020000);
    // ---- prologue by role: the storing waves stage the source block (LDS-DMA: 16 rows x 64 B per wave-instruction), the
    // computing waves build the tables
    if (!computes) {
        const float *src = blk + (int64_t)p.src0 * p.F;
        for (int r0 = (wid - kPpCompute) * 16; r0 < nn; r0 += 16 * kPpStore)
            if (r0 + grp < nn)
                __builtin_amdgcn_global_load_lds(
                    (const void __attribute__((address_space(1))) *)(src + (int64_t)(r0 + grp) * p.ld),
                    (void __attribute__((address_space(3))) *)(smem + L.buf[0] + r0 * RB), 16, 0, 0);
    } else {
        if (threadIdx.x < 8)                                     // the two rows of zeros
            *reinterpret_cast<float4 *>(smem + ((threadIdx.x >> 2) ? L.buf[1] : L.buf[0]) + rp * RB + 16 * (threadIdx.x & 3)) =
                make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = threadIdx.x; r <= rp; r += 64 * kPpCompute) {
            const bool live = r < nn;
            const int b = live ? p.ptr[n0 + r] : 0, d = (live ? p.ptr[n0 + r + 1] : 0) - b;
            const int din = live ? p.deg_ptr[n0 + r + 1] - p.deg_ptr[n0 + r] : 0;
            *reinterpret_cast<float *>(smem + L.dis + 4 * r) = inv_sqrt_count(din);
            if (r == rp) break;
            *reinterpret_cast<int2 *>(smem + L.bounds + 8 * r) = make_int2(b, d);
            const u32x4 i0 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * b, 0, 0);
            const u32x4 i1 = __builtin_amdgcn_raw_buffer_load_b128(ro, 4 * b + 16, 0, 0);
            const unsigned g[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
            unsigned l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned loc = g[j] - (unsigned)n0;
                l[j] = (j < d && loc < (unsigned)nn) ? loc : (unsigned)rp;
            }
            *reinterpret_cast<uint4 *>(smem + L.ids + 16 * r) =
                make_uint4(l[0] | l[1] << 16, l[2] | l[3] << 16, l[4] | l[5] << 16, l[6] | l[7] << 16);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): slice and tables have landed
    lds_barrier();

    const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned dbase = sbase + L.dis;
    const bool want_rm = p.rowmax != nullptr;
    const int steps = (rp + 16 * kPpCompute - 1) / (16 * kPpCompute);          // <= 6 for 1,024 nodes
    float pm[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        pm[s] = 0.f;
        const int row = (s * kPpCompute + wid) * 16 + grp;
        if (computes && s < steps && row < rp && want_rm && (p.rm_mode & 1))
            pm[s] = chain_absmax(*reinterpret_cast<const float4 *>(smem + L.buf[0] + row * RB + 16 * sub));
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.slab + (int64_t)n0 * p.ld, 0, (int)((unsigned)nn * (unsigned)p.ld * 4u), 0x00020000);
    const unsigned ldb = (unsigned)p.ld * 4u;
    for (int h = 0; h < p.K; ++h) {
        const int so = (h & 1) ? L.buf[1] : L.buf[0], dof = (h & 1) ? L.buf[0] : L.buf[1];
        if (computes) {
            const unsigned lbase = sbase + so + 16u * sub;
            const unsigned gbase = lbase - (unsigned)n0 * (unsigned)RB, zsub = lbase + (unsigned)rp * RB;
            const LdsWindow win{lbase, (unsigned)nn * (unsigned)RB};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                int row = (s * kPpCompute + wid) * 16 + grp;
                if ((s * kPpCompute + wid) * 16 >= rp) continue;                // (wave-uniform; rp <= 1,152)
                asm volatile("" : "+v"(row));
                const uint4 iv = *reinterpret_cast<const uint4 *>(smem + L.ids + 16 * row);
                const int2 bd = *reinterpret_cast<const int2 *>(smem + L.bounds + 8 * row);
                const float di = *reinterpret_cast<const float *>(smem + L.dis + 4 * row);
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!__any(bd.y > 6))
                    gcn_slots<6, 0, RB>(a, iv, di, lbase, dbase);
                else if (!__any(bd.y > 7))
                    gcn_slots<7, 0, RB>(a, iv, di, lbase, dbase);
                else {
                    gcn_slots<4, 0, RB>(a, iv, di, lbase, dbase);
                    gcn_slots<4, 4, RB>(a, iv, di, lbase, dbase);
                    int pbeg = bd.x, rem = bd.y;
                    while (__any(rem > 8)) {                     // the tail of long rows: ids and weights from memory
                        pbeg += 8, rem -= 8;
                        Chunk n;
                        load_chunk<true>(n, ro, rw, pbeg);
                        chunk_slots<true, 4, 0, 0, RB>(a, n, rem, gbase, zsub, win);
                        if (__any(rem > 4)) chunk_slots<true, 4, 0, 4, RB>(a, n, rem, gbase, zsub, win);
                    }
                }
                *reinterpret_cast<float4 *>(smem + dof + row * RB + 16 * sub) = a;
                pm[s] = fmaxf(pm[s], chain_absmax(a));
            }
        } else if (h >= 1) {
            // block h (what hop h - 1 produced: the buffer the computing waves are gathering from) -> memory
            const unsigned dcol = (unsigned)((p.src0 + h * p.dir) * p.F + slice * COLS + 4 * sub) * 4u;
            for (int r0 = (wid - kPpCompute) * 16; r0 < nn; r0 += 16 * kPpStore) {
                const float4 v = *reinterpret_cast<const float4 *>(smem + so + (r0 + grp) * RB + 16 * sub);
                store_piece(v, rs, (unsigned)(r0 + grp) * ldb + dcol);
            }
        }
        lds_barrier();
    }
    {   // the last block: all 16 waves
        const int so = (p.K & 1) ? L.buf[1] : L.buf[0];
        const unsigned dcol = (unsigned)((p.src0 + p.K * p.dir) * p.F + slice * COLS + 4 * sub) * 4u;
        for (int r0 = wid * 16; r0 < nn; r0 += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(smem + so + (r0 + grp) * RB + 16 * sub);
            store_piece(v, rs, (unsigned)(r0 + grp) * ldb + dcol);
        }
    }
    if (want_rm && computes) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float m = pm[s];
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            const int row = (s * kPpCompute + wid) * 16 + grp;
            if (s < steps && sub == 0 && row < nn) atomicMax(reinterpret_cast<int *>(p.rowmax + n0 + row), __float_as_int(m));
        }
    }
}

static bool launch_chain_pp(unsigned grid, hipStream_t stream, const ChainParams &p, int big) {
    const int rp = (big + 15) / 16 * 16;
    if ((size_t)pp_layout(rp).total > kChainLdsRequest) return false;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hop_chain_pp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kChainLdsRequest) != hipSuccess)
            return false;
        attr_set = true;
    }
    trace_kernel("k_hop_chain_pp");
    hipLaunchKernelGGL(k_hop_chain_pp, dim3(grid), dim3(1024), kChainLdsRequest, stream, p, rp);
    return true;
}

