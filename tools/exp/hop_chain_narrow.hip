// tools/exp/hop_chain_narrow.hip -- round-5 experiment, NOT part of the library: pack + all K hops (+ row maxima) of a first
// layer's narrow input (F = 21 / 25) as ONE launch, one 1024-thread workgroup per graph with its x rows resident in LDS.
// Bit-identical to dc_spmm_f32_pack + 2 x dc_spmm_f32 (tested on the GPU while it was in the library), and SLOWER:
// 30.1 us (soft, F = 21) / 45.6 us (rigid, F = 25) per slab against 19.2 / 21.6 us for the three launches (B = 32, replayed from
// a hipGraph; profiles/r05/d_narrow_chain.txt); headline 0.659 - 0.668 ms against 0.648 - 0.652.  Why: only 32 workgroups; the
// neighbours' rows are gathered with ds_read_b32 from rows of 84 / 100 bytes at effectively random ids - 4 - 5-way bank
// conflicts on every gather - and the 40-neighbour pole rows of the UV spheres walk their tail serially in every hop.  The
// per-hop kernels read the same rows from L2 as coalesced 84-byte segments, 256+ workgroups at once.  Kept for the record
// (fragments of dc_hopchain.hip: needs its Chunk / load_chunk / kChainGraphs / trace_kernel to compile).
// ---- narrow inputs: pack + K hops of a layer's OWN input in one launch ---------------------------------------------------
// The encoder's first TAGConv layers hop over the raw graph.x (21 / 25 floats per node, models/model.py:44-50,71,77): three
// dependent launches of a few microseconds each per branch (dc_spmm_f32_pack + 2 x dc_spmm_f32), i.e. mostly launch gaps on
// the chain of small kernels a new batch starts with.  Here ONE 1024-thread workgroup owns a whole graph: its x rows go
// global -> LDS once (nn x F floats, <= 100 KB), thread r owns row r, every hop gathers the neighbours' rows from LDS
// (ds_read_b32, F odd: consecutive rows start in different banks), keeps the new row in registers until every thread has
// read, writes it back, and the block leaves for the slab as whole rows (consecutive lanes = consecutive floats of a row).
// Terms, order and rounding are dc_spmm_f32's (multiply and add rounded separately, neighbours in p order, sum started at +0):
// bit-identical to the three launches it replaces.  Optionally rowmax[i] = max |slab[i, 0:width]|.
struct NarrowParams {
    const int32_t *ptr, *other;
    const float *w, *x;
    float *slab, *rowmax;
    int64_t ldx, ld;
    int32_t cap;                      // elements of other / w (range check of the 16-byte id / weight loads)
    int K, width, wpad, nseg;
    int32_t node_ptr[kChainGraphs + 1];
};

template <int F>
__global__ void __launch_bounds__(1024)
k_hop_chain_narrow(NarrowParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);                 // [nn][F]
    const int seg = blockIdx.x, n0 = p.node_ptr[seg], nn = p.node_ptr[seg + 1] - n0;
    const int r = threadIdx.x;
    // x rows of the graph: contiguous when ldx == F, else row by row - either way consecutive lanes take consecutive floats
    for (int i = threadIdx.x; i < nn * F; i += 1024) {
        const int row = i / F, c = i - row * F;
        xs[i] = p.x[(int64_t)(n0 + row) * p.ldx + c];
    }
    if (threadIdx.x < F) xs[1024 * F + threadIdx.x] = 0.f;       // the row of zeros (never overwritten: rows end at 1024)
    __syncthreads();
    float *out0 = p.slab + (int64_t)n0 * p.ld;
    const int pad = p.wpad - p.width;
    // block 0 (the packed input) and the zero padding behind the last block
    for (int i = threadIdx.x; i < nn * F; i += 1024) {
        const int row = i / F, c = i - row * F;
        out0[(int64_t)row * p.ld + c] = xs[i];
    }
    for (int i = threadIdx.x; i < nn * pad; i += 1024) {
        const int row = i / pad, c = i - row * pad;
        out0[(int64_t)row * p.ld + p.width + c] = 0.f;
    }
    const bool live = r < nn;
    const int pb = live ? p.ptr[n0 + r] : 0, pe = live ? p.ptr[n0 + r + 1] : 0;
    // the row's first 8 neighbours (local ids, weights) stay in registers for all K hops: the adjacency is the same for every
    // hop, and per-edge loads inside the hop loop are a chain of dependent L2 round trips (first version of this kernel: 3 %
    // SLOWER in the step than the three launches it replaces).  A slot past the row's end, or a neighbour outside the graph (a
    // wrong layout), reads a row of zeros behind the graph's rows with weight 0: +0 added to a sum that started at +0 leaves it
    // unchanged (a real row would do for finite data, but 0 x inf = NaN).
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.other), 0, p.cap * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w ? p.w : reinterpret_cast<const float *>(p.other)), 0,
                                          p.cap * 4, 0x00020000);
    Chunk ck;
    load_chunk<true>(ck, ro, rw, pb);
    const unsigned gid[8] = {ck.i0.x, ck.i0.y, ck.i0.z, ck.i0.w, ck.i1.x, ck.i1.y, ck.i1.z, ck.i1.w};
    const unsigned gwb[8] = {ck.w0.x, ck.w0.y, ck.w0.z, ck.w0.w, ck.w1.x, ck.w1.y, ck.w1.z, ck.w1.w};
    unsigned nid[8];
    float nwt[8];
    const int deg = pe - pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned loc = gid[j] - (unsigned)n0;
        const bool ok = j < deg && loc < (unsigned)nn;
        nid[j] = ok ? loc * (unsigned)F : 1024u * (unsigned)F;
        nwt[j] = ok ? (p.w ? __uint_as_float(gwb[j]) : 1.0f) : 0.0f;
    }
    float rm = 0.f;
    if (live && p.rowmax) {
#pragma unroll
        for (int c = 0; c < F; ++c) rm = fmaxf(rm, fabsf(xs[r * F + c]));
    }
    for (int h = 0; h < p.K; ++h) {
        float acc[F];
#pragma unroll
        for (int c = 0; c < F; ++c) acc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!__any(j < deg)) break;                          // (wave-uniform: no lane of the wave has a j-th neighbour)
            const float *src = xs + nid[j];
            const float wt = nwt[j];
#pragma unroll
            for (int c = 0; c < F; ++c) {
                const float m = wt * src[c];
                acc[c] = acc[c] + m;
            }
        }
        for (int q = pb + 8; q < pe; ++q) {                      // rows with more than 8 neighbours (mesh poles, hubs)
            const unsigned loc = (unsigned)(p.other[q] - n0);
            const float wt = p.w ? p.w[q] : 1.0f;
            if (loc < (unsigned)nn) {
                const float *src = xs + loc * F;
#pragma unroll
                for (int c = 0; c < F; ++c) {
                    const float m = wt * src[c];
                    acc[c] = acc[c] + m;
                }
            }
        }
        __syncthreads();                                         // every thread has read block h
        if (live) {
#pragma unroll
            for (int c = 0; c < F; ++c) {
                xs[r * F + c] = acc[c];
                rm = fmaxf(rm, fabsf(acc[c]));
            }
        }
        __syncthreads();
        float *outh = out0 + (h + 1) * F;
        for (int i = threadIdx.x; i < nn * F; i += 1024) {
            const int row = i / F, c = i - row * F;
            outh[(int64_t)row * p.ld + c] = xs[i];
        }
    }
    if (live && p.rowmax) p.rowmax[n0 + r] = rm;
}

template <int F>
static bool launch_narrow(hipStream_t stream, const NarrowParams &p, int max_nodes) {
    const size_t lds = (size_t)(max_nodes + 1) * F * 4;          // the graph's rows + a row of zeros
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hop_chain_narrow<F>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return false;
        attr = lds;
    }
    static const std::string name = "k_hop_chain_narrow<" + std::to_string(F) + ">";
    trace_kernel(name.c_str());
    hipLaunchKernelGGL((k_hop_chain_narrow<F>), dim3((unsigned)p.nseg), dim3(1024), lds, stream, p);
    return true;
}


extern "C" int dc_hop_chain_narrow_supported(int64_t F) { return F == 21 || F == 25 || F == 16 || F == 32 || F == 8; }

extern "C" int dc_hop_chain_narrow_f32(const int32_t *ptr, const int32_t *other, const float *w, int64_t cap,
                                       const int64_t *node_ptr_host, int nseg, const float *x, int64_t ldx, float *slab,
                                       int64_t ld, int64_t N, int64_t F, int K, int64_t width, int64_t wpad, float *rowmax,
                                       dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N >= 0 && K >= 0 && nseg >= 0, "dc_hop_chain_narrow_f32: negative size");
    DC_REQUIRE(dc_hop_chain_narrow_supported(F), "dc_hop_chain_narrow_f32: F = %lld has no instantiation (8, 16, 21, 25, 32)",
               (long long)F);
    if (N == 0 || nseg == 0) return DC_OK;
    DC_REQUIRE(ptr && other && x && slab && node_ptr_host, "dc_hop_chain_narrow_f32: null pointer");
    DC_REQUIRE(ldx >= F && width == (int64_t)(K + 1) * F && wpad >= width && ld >= wpad,
               "dc_hop_chain_narrow_f32: needs ldx >= F, width == (K + 1) F, wpad >= width, ld >= wpad");
    DC_REQUIRE((const void *)x != (const void *)slab, "dc_hop_chain_narrow_f32: the slab must not alias x");
    DC_REQUIRE(N < (int64_t)1 << 24 && ld < ((int64_t)1 << 20) && cap >= 0 && cap < (int64_t)1 << 29,
               "dc_hop_chain_narrow_f32: size out of range");
    DC_REQUIRE(node_ptr_host[0] == 0 && node_ptr_host[nseg] == N, "dc_hop_chain_narrow_f32: the graphs' offsets must cover [0, N]");
    for (int i = 0; i < nseg; ++i) {
        const int64_t dn = node_ptr_host[i + 1] - node_ptr_host[i];
        DC_REQUIRE(dn >= 0 && dn <= 1024, "dc_hop_chain_narrow_f32: graph %d has %lld nodes (cap 1024)", i, (long long)dn);
    }
    NarrowParams p{};
    p.ptr = ptr, p.other = other, p.w = w, p.x = x, p.slab = slab, p.rowmax = rowmax, p.ldx = ldx, p.ld = ld;
    p.K = K, p.width = (int)width, p.wpad = (int)wpad, p.cap = (int32_t)cap;
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
        bool ok = false;
        switch ((int)F) {
        case 8: ok = launch_narrow<8>(stream, p, 1024); break;
        case 16: ok = launch_narrow<16>(stream, p, 1024); break;
        case 21: ok = launch_narrow<21>(stream, p, 1024); break;
        case 25: ok = launch_narrow<25>(stream, p, 1024); break;
        default: ok = launch_narrow<32>(stream, p, 1024); break;
        }
        DC_REQUIRE(ok, "dc_hop_chain_narrow_f32: cannot reserve the kernel's LDS");
    }
    return check_launch("dc_hop_chain_narrow_f32");
}
