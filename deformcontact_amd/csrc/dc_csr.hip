// dc_csr.hip -- edge_index -> stably sorted adjacency (+ gcn_norm weights), gfx950.
//
// Replaces the implicit destination ordering of PyG's scatter_add_ and the
// gcn_norm call TAGConv/GCNConv.forward repeats on every invocation
// (/root/reference/models/model.py:71,77).  Integer work, L2/HBM-bound, once per
// batched edge_index.
//
// One pipeline builds ONE side (dc_csr_build: edges grouped by edge_index[key_row])
// or BOTH sides (dc_graph_build: by destination for the forward hop and by source
// for the transposed hop of the backward pass) -- the two sides share every launch
// (gridDim.y = side), so a whole GraphIndex costs 5 launches for the batch sizes of
// the reference (2N <= kScanSmall) and 7 beyond, on one stream, no host sync:
//
//   k_init       counters = 0 (1 with self loops), status = 0, big-group count = 0
//   k_count      histogram of the key endpoint (int atomics; order-free)
//   k_scan_*     exclusive scan -> ptr[N+1] (one block per side when small, else
//                reduce / tile scan / apply); groups longer than kRankLoop are listed
//   k_fill       bucket every edge id into its group in ARBITRARY order
//   k_emit       small groups: rank each id inside its group (count of smaller ids);
//                listed (big) groups: one workgroup sorts the group's ids (bitonic
//                network with ascending comparators only, in LDS up to 4096 ids,
//                in place in global memory beyond) -- O(S log^2 S), no O(deg^2) walk
//                on hubs; then perm/other/w are written at ptr[key]+rank, so the
//                result is the stable sort no matter how the atomics interleaved.
//
// Large edge lists (run_build: bucket_plan) take the bucketed build further down instead - k_bk_count / k_bk_scan /
// k_bk_scatter partition the slots by node bucket, k_bk_build finishes a bucket per workgroup in LDS, k_bk_weights
// writes the gcn_norm weights: same arrays, no atomic per edge, indifferent to the order of the edges.
#include "dc_common.h"

namespace dc {

constexpr int kScanTile = 1024;      // elements per scan block (256 threads x 4)
constexpr int kScanSmall = 1 << 16;  // per-side N up to which ONE block scans a side
constexpr int kRankLoop = 48;        // groups up to this length rank by counting
constexpr int kBigBlocks = 64;       // extra workgroups of k_emit that sort listed groups
constexpr int kSortLds = 4096;       // ids sorted in LDS (16 KiB); longer groups in place

// The edge set: up to kMaxParts edge_index arrays read as ONE concatenated edge list (edge ids run
// through the parts in order) over ONE node space (part p's node ids are shifted by node_off[p]): a
// block-diagonal union - e.g. the soft and the rigid graph of a batch - without materialising the
// merged [2, E] array.  One part with offset 0 is the plain edge_index.
constexpr int kMaxParts = DC_MAX_PARTS;
struct Edges {
    const int64_t *src[kMaxParts], *dst[kMaxParts];   // rows 0 / 1 of each part's edge_index
    int64_t e_beg[kMaxParts + 1];                      // first edge id of each part (e_beg[nparts] = E)
    int64_t node_off[kMaxParts], nodes[kMaxParts];     // id shift and node count (range check) of each part
    int nparts;
};

// endpoints of edge e in the merged node space; false when either lies outside its part's [0, nodes)
// (The part's fields are SELECTED, not indexed: `ed.src[p]` with a per-lane p is a vector load from the kernel-argument
// segment - two dependent memory round trips in front of every edge load; the selects keep every field a scalar load.)
__device__ __forceinline__ bool load_edge(const Edges &ed, int64_t e, int64_t &s, int64_t &d) {
    const int64_t *sp = ed.src[0], *dp = ed.dst[0];
    int64_t beg = ed.e_beg[0], off = ed.node_off[0], nodes = ed.nodes[0];
#pragma unroll
    for (int q = 1; q < kMaxParts; ++q) {
        const bool in = q < ed.nparts && e >= ed.e_beg[q];
        sp = in ? ed.src[q] : sp, dp = in ? ed.dst[q] : dp;
        beg = in ? ed.e_beg[q] : beg, off = in ? ed.node_off[q] : off, nodes = in ? ed.nodes[q] : nodes;
    }
    const int64_t l = e - beg;
    s = sp[l], d = dp[l];
    const bool ok = s >= 0 && s < nodes && d >= 0 && d < nodes;
    s += off, d += off;
    return ok;
}

struct Side {
    int32_t *cnt;          // [N]   counters (workspace)
    int32_t *cur;          // [N]   fill cursors (workspace; a separate array so the scan's loads of
                           //       cnt are independent of its stores)
    int32_t *tmp;          // [E+N] edge ids bucketed per group (workspace)
    int32_t *tiles;        // [ntiles+2] scan tile sums (workspace)
    int32_t *big;          // [1 + N] count + list of groups longer than kRankLoop (workspace)
    int32_t *ptr;          // [N+1] out
    int32_t *other;        // [E'] out
    int32_t *perm;         // [E'] out
    float *w;              // [E'] out or null
    const int32_t *deg_ptr;  // offsets whose differences are the in-degrees gcn_norm uses
    int key_is_dst;
    // bucketed build (workspace): [nb + 1] slots per bucket, after k_bk_scan the first slot of each bucket;
    // [nb] partition cursors; [E + N] partitioned records: edge id (>= E: the appended self loop of node id - E),
    // other endpoint, key - first node of its bucket
    int32_t *bk_start, *bk_cur, *bk_eid, *bk_oth;
    uint16_t *bk_kl;
};

struct Build {
    Side s[2];
    Edges ed;
    int64_t E, N;
    int self_loops;
    int32_t *status;
};

__global__ void __launch_bounds__(256)
k_init(Build b) {
    const Side &sd = b.s[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.N) sd.cnt[i] = b.self_loops ? 1 : 0;
    if (i == 0) {
        sd.big[0] = 0;
        if (blockIdx.y == 0) *b.status = 0;
    }
}

// Block-local histogram (r03).  Consecutive edges of a mesh batch name nearby nodes (triangle-major edge order, locally
// numbered meshes): the 256 edges of a block hit ~100 distinct keys inside a window of a few hundred ids, yet each edge
// paid a global int atomic (k_count / k_fill were 11 - 12 us each at B = 32, atomic-latency bound).  Here a block keeps
// kBins counters in LDS for the id window [base, base + kBins) around its first edge, counts in-window keys there and
// sends ONE global atomic per non-empty bin; keys outside the window (arbitrary graphs) keep the direct global atomic.
constexpr int kBins = 1024;

__global__ void __launch_bounds__(256)
k_count(Build b, int nsides) {
    __shared__ int32_t bins[2][kBins];
    __shared__ int64_t base[2];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 2 * kBins; i += 256) (&bins[0][0])[i] = 0;
    // every thread fetches its own edge BEFORE the barrier; thread 0's edge (always < E: the grid covers E edges) places
    // the windows - garbage ids only move them - so the window costs no memory round trip of its own
    int64_t k = 0, o = 0;
    bool ok = true;
    if (e < b.E) {
        int64_t src, dst;
        ok = load_edge(b.ed, e, src, dst);
        k = b.s[0].key_is_dst ? dst : src, o = b.s[0].key_is_dst ? src : dst;
    }
    if (threadIdx.x == 0) base[0] = k - kBins / 2, base[1] = o - kBins / 2;
    __syncthreads();
    if (e < b.E) {
        if (!ok) {
            atomicOr(b.status, 1);
        } else if (!(b.self_loops && k == o)) {
            const int64_t rk = k - base[0], ro = o - base[1];
            if (rk >= 0 && rk < kBins) atomicAdd(&bins[0][rk], 1);
            else atomicAdd(&b.s[0].cnt[k], 1);
            if (nsides == 2) {                            // side 1 groups by side 0's other row
                if (ro >= 0 && ro < kBins) atomicAdd(&bins[1][ro], 1);
                else atomicAdd(&b.s[1].cnt[o], 1);
            }
        }
    }
    __syncthreads();
    for (int sd = 0; sd < nsides; ++sd)
        for (int i = threadIdx.x; i < kBins; i += 256) {
            const int c = bins[sd][i];
            if (c) atomicAdd(&b.s[sd].cnt[base[sd] + i], c);
        }
}

// ---- exclusive scan of cnt[0..N) into ptr[0..N], ptr[N] = total -----------
__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        int t = __shfl_up(v, d, kWave);
        if ((threadIdx.x & (kWave - 1)) >= d) v += t;
    }
    return v;
}

// inclusive scan across a block of NW waves; returns this thread's inclusive
// value, *total = block sum
template <int NW>
__device__ __forceinline__ int block_incl_scan(int v, int *total) {
    __shared__ int wsum[NW];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wid) off += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    __syncthreads();
    return inc + off;
}

__device__ __forceinline__ void note_big(const Side &sd, int64_t node, int len) {
    if (len > kRankLoop) sd.big[1 + atomicAdd(&sd.big[0], 1)] = (int32_t)node;
}

// one 1024-thread block (16 waves) per side: counts -> ptr, cursors, big-group list.  Wave w owns the
// contiguous segment [w * seg, (w + 1) * seg) and walks it in 256-element chunks, lane l taking the
// 16 bytes at chunk + 4 l: every load and store is one fully coalesced 1 KiB wave access (a run per
// THREAD - 64 cache lines per instruction - made this 18-25 us; r02 build profile).  Pass 1: segment
// totals; the block scans the 16 totals; pass 2: wave-level scan per chunk with a running carry.
__global__ void __launch_bounds__(1024)
k_scan_small(Build b) {
    const Side &sd = b.s[blockIdx.x];
    const int32_t *__restrict__ cnt = sd.cnt;
    int32_t *__restrict__ ptr = sd.ptr;
    int32_t *__restrict__ cur = sd.cur;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t seg = (((b.N + 15) / 16) + 255) & ~int64_t(255);
    const int64_t beg = (int64_t)w * seg, end = beg + seg < b.N ? beg + seg : b.N;
    auto load4 = [&](int64_t i) {
        int4 v = make_int4(0, 0, 0, 0);
        if (i + 4 <= end) v = *reinterpret_cast<const int4 *>(cnt + i);       // cnt is 16-byte aligned
        else {
            if (i < end) v.x = cnt[i];
            if (i + 1 < end) v.y = cnt[i + 1];
            if (i + 2 < end) v.z = cnt[i + 2];
        }
        return v;
    };
    int s = 0;
    for (int64_t c = beg; c < end; c += 256) {
        const int4 v = load4(c + 4 * lane);
        s += v.x + v.y + v.z + v.w;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    __shared__ int wtot[16];
    if (lane == 0) wtot[w] = s;
    __syncthreads();
    int carry = 0, total = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < w) carry += wtot[j];
        total += wtot[j];
    }
    for (int64_t c = beg; c < end; c += 256) {
        const int64_t i = c + 4 * lane;
        const int4 v = load4(i);
        const int mine = v.x + v.y + v.z + v.w;
        const int incl = wave_incl_scan(mine);
        int4 o;
        o.x = carry + incl - mine, o.y = o.x + v.x, o.z = o.y + v.y, o.w = o.z + v.z;
        if (i + 4 <= end) {
            *reinterpret_cast<int4 *>(ptr + i) = o;
            *reinterpret_cast<int4 *>(cur + i) = o;
        } else {
            if (i < end) ptr[i] = o.x, cur[i] = o.x;
            if (i + 1 < end) ptr[i + 1] = o.y, cur[i + 1] = o.y;
            if (i + 2 < end) ptr[i + 2] = o.z, cur[i + 2] = o.z;
        }
        if (v.x > kRankLoop || v.y > kRankLoop || v.z > kRankLoop || v.w > kRankLoop) {
            if (i < end) note_big(sd, i, v.x);
            if (i + 1 < end) note_big(sd, i + 1, v.y);
            if (i + 2 < end) note_big(sd, i + 2, v.z);
            if (i + 3 < end) note_big(sd, i + 3, v.w);
        }
        carry += __shfl(incl, 63);
    }
    if (threadIdx.x == 0) ptr[b.N] = total;
}

__global__ void __launch_bounds__(256)
k_scan_reduce(Build b) {
    const Side &sd = b.s[blockIdx.y];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < b.N) s += sd.cnt[base + j];
    int total;
    block_incl_scan<4>(s, &total);
    if (threadIdx.x == 0) sd.tiles[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
k_scan_tiles(Build b, int64_t ntiles) {   // one block per side, in place -> exclusive
    const Side &sd = b.s[blockIdx.x];
    int carry = 0;
    for (int64_t base = 0; base < ntiles; base += 256) {
        const int64_t i = base + threadIdx.x;
        const int v = i < ntiles ? sd.tiles[i] : 0;
        int total;
        const int inc = block_incl_scan<4>(v, &total);
        if (i < ntiles) sd.tiles[i] = carry + inc - v;
        carry += total;
    }
    if (threadIdx.x == 0) sd.tiles[ntiles] = carry;
}

__global__ void __launch_bounds__(256)
k_scan_apply(Build b, int64_t ntiles) {
    const Side &sd = b.s[blockIdx.y];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = base + j < b.N ? sd.cnt[base + j] : 0;
        s += v[j];
    }
    int total;
    int run = block_incl_scan<4>(s, &total) - s + sd.tiles[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < b.N) {
            sd.ptr[base + j] = run;
            sd.cur[base + j] = run;   // the fill cursor
            note_big(sd, base + j, v[j]);
        }
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sd.ptr[b.N] = sd.tiles[ntiles];
}

// Bucket every edge id into its group (arbitrary order inside a group; k_emit ranks).  Same block-local scheme as
// k_count: an edge takes its slot inside the block's share of a group from an LDS counter, the block reserves the share
// with ONE global atomic per non-empty bin, then every edge writes at reserved base + local slot.
__global__ void __launch_bounds__(256)
k_fill(Build b, int nsides) {
    __shared__ int32_t bins[2][kBins];
    __shared__ int64_t base[2];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 2 * kBins; i += 256) (&bins[0][0])[i] = 0;
    int64_t key[2] = {-1, -1};                             // -1: this thread places nothing on that side
    int64_t k0 = t - b.E, o0 = t - b.E;                    // thread 0 of a block of appended self loops: keys = node ids
    if (t < b.E) {                                         // (own edge fetched before the barrier, as in k_count)
        int64_t src, dst;
        const bool ok = load_edge(b.ed, t, src, dst);
        k0 = b.s[0].key_is_dst ? dst : src, o0 = b.s[0].key_is_dst ? src : dst;
        if (ok && !(b.self_loops && k0 == o0)) key[0] = k0, key[1] = o0;
    } else if (b.self_loops && t < b.E + b.N) {
        key[0] = key[1] = t - b.E;
    }
    if (threadIdx.x == 0) base[0] = k0 - kBins / 2, base[1] = o0 - kBins / 2;
    __syncthreads();
    int slot[2] = {-1, -1};                                // local slot inside the block's share (-1: direct global)
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        if (sd >= nsides || key[sd] < 0) continue;
        const int64_t r = key[sd] - base[sd];
        if (r >= 0 && r < kBins) slot[sd] = atomicAdd(&bins[sd][r], 1);
    }
    __syncthreads();
    for (int sd = 0; sd < nsides; ++sd)                    // bins: count -> reserved global base
        for (int i = threadIdx.x; i < kBins; i += 256) {
            const int c = bins[sd][i];
            if (c) bins[sd][i] = atomicAdd(&b.s[sd].cur[base[sd] + i], c);
        }
    __syncthreads();
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
        if (sd >= nsides || key[sd] < 0) continue;
        const int pos = slot[sd] >= 0 ? bins[sd][key[sd] - base[sd]] + slot[sd]
                                      : atomicAdd(&b.s[sd].cur[key[sd]], 1);
        b.s[sd].tmp[pos] = (int32_t)t;
    }
}

__device__ __forceinline__ float inv_sqrt_deg(const int32_t *deg_ptr, int64_t v) {
    return inv_sqrt_count(deg_ptr[v + 1] - deg_ptr[v]);   // deg.pow(-0.5), inf -> 0
}

__device__ __forceinline__ void emit_one(const Build &b, const Side &sd, int32_t eid, int64_t k,
                                         int32_t out) {
    int64_t o = k;
    if (eid < b.E) {
        int64_t src, dst;
        load_edge(b.ed, eid, src, dst);
        o = sd.key_is_dst ? src : dst;
    }
    sd.perm[out] = eid;
    sd.other[out] = (int32_t)o;
    if (sd.w) {
        // gcn_norm: dis[row] * 1 * dis[col]  (row = source, col = destination)
        const int64_t src = sd.key_is_dst ? o : k, dst = sd.key_is_dst ? k : o;
        sd.w[out] = inv_sqrt_deg(sd.deg_ptr, src) * 1.0f * inv_sqrt_deg(sd.deg_ptr, dst);
    }
}

// Sort a[0..S) ascending with the bitonic network in its all-ascending-comparator form (the
// first step of every merge compares mirrored positions), so positions >= S can be treated as
// +infinity without ever being stored.  Ids are distinct: no stability question.  (The element type is the
// accessor's: int32 ids, or 64-bit (id, payload) records ordered by the id in the high word.)
template <typename Get, typename Put>
__device__ __forceinline__ void bitonic_sort(int S, Get get, Put put) {
    int P = 2;
    while (P < S) P <<= 1;
    for (int size = 2; size <= P; size <<= 1) {
        const int half = size >> 1;
        for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
            const int blk = t / half, off = t - blk * half;
            const int i = blk * size + off, j = blk * size + size - 1 - off;
            if (j < S) {
                const auto a = get(i), c = get(j);
                if (a > c) { put(i, c); put(j, a); }
            }
        }
        __syncthreads();
        for (int stride = half >> 1; stride >= 1; stride >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int i = 2 * stride * (t / stride) + (t % stride), j = i + stride;
                if (j < S) {
                    const auto a = get(i), c = get(j);
                    if (a > c) { put(i, c); put(j, a); }
                }
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256)
k_emit(Build b, unsigned slot_blocks) {
    const Side &sd = b.s[blockIdx.y];
    if (blockIdx.x >= slot_blocks) {
        // ---- listed groups: sort the ids of the whole group, then emit in order ----
        __shared__ int32_t lds[kSortLds];
        const int nbig = sd.big[0];
        for (int g = blockIdx.x - slot_blocks; g < nbig; g += kBigBlocks) {
            const int64_t k = sd.big[1 + g];
            const int32_t beg = sd.ptr[k], S = sd.ptr[k + 1] - beg;
            int32_t *seg = sd.tmp + beg;
            if (S <= kSortLds) {
                for (int q = threadIdx.x; q < S; q += blockDim.x) lds[q] = seg[q];
                __syncthreads();
                bitonic_sort(S, [&](int i) { return lds[i]; }, [&](int i, int32_t v) { lds[i] = v; });
                for (int q = threadIdx.x; q < S; q += blockDim.x) emit_one(b, sd, lds[q], k, beg + q);
                __syncthreads();
            } else {
                // in place in global memory; one workgroup owns the segment: its own stores are
                // visible to its own later loads once the barrier has drained them (volatile
                // accesses keep them out of registers / the non-coherent path)
                volatile int32_t *vs = seg;
                bitonic_sort(S, [&](int i) { return vs[i]; }, [&](int i, int32_t v) { vs[i] = v; });
                for (int q = threadIdx.x; q < S; q += blockDim.x) emit_one(b, sd, vs[q], k, beg + q);
                __syncthreads();
            }
        }
        return;
    }
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= sd.ptr[b.N]) return;
    const int32_t eid = sd.tmp[p];
    int64_t k = eid - b.E;
    if (eid < b.E) {
        int64_t src, dst;
        load_edge(b.ed, eid, src, dst);
        k = sd.key_is_dst ? dst : src;
    }
    const int32_t beg = sd.ptr[k], end = sd.ptr[k + 1];
    if (end - beg > kRankLoop) return;          // a listed group: the sorting workgroups emit it
    int rank = 0;
    for (int32_t q = beg; q < end; ++q) rank += sd.tmp[q] < eid;
    emit_one(b, sd, eid, k, beg + rank);
}

// ---- segmented build (r03): a batch whose graphs are known, contiguous node AND edge ranges ---------------------
// Batch.from_data_list concatenates meshes: graph i owns nodes [node_ptr[i], node_ptr[i+1]) and edges
// [edge_ptr[i], edge_ptr[i+1]), and none of its edges leaves its node range.  Then the stable sort by destination (or
// source) of the whole edge list is the concatenation of the per-graph sorts, every graph fits in LDS (a 1024-node
// mesh has ~6k directed edges) and the five dependent launches of the global pipeline - count, scan, fill, emit, each
// bound by launch + atomic latency, ~41 us at B = 32 - collapse into ONE launch: workgroup (graph, side) counts in
// LDS, scans its <= kSegNodes counters, buckets the edge ids (LDS cursors), ranks each id inside its group and writes
// ptr / other / perm / w at the graph's own offsets (consecutive threads -> consecutive output slots).  No global
// atomics, no workspace, no device-side offsets: the layout is host data and travels in the kernel arguments
// (kSegChunk graphs per launch).  The result is the arrays dc_graph_build writes, bit for bit.
constexpr int kSegNodes = DC_SEG_MAX_NODES;   // per-graph caps (LDS: 3 x 2 B x edges + 3 x 4 B x nodes = 144 KiB)
constexpr int kSegEdges = DC_SEG_MAX_EDGES;

constexpr int kSegChunk = 96;                 // graphs per launch (their offsets travel as kernel arguments)

struct SegBuild {
    const int64_t *src, *dst;            // rows 0 / 1 of edge_index
    int32_t node_ptr[kSegChunk + 1];     // offsets of this launch's graphs (host-validated: ascending, within caps)
    int32_t edge_ptr[kSegChunk + 1];
    int64_t E, N;
    int nseg, last;                      // graphs in this launch; last != 0: this launch holds the batch's last graph
    int32_t *ptr[2], *other[2], *perm[2];
    float *w[2];
    int32_t *status;
};

__global__ void __launch_bounds__(1024)
k_build_segment(SegBuild b) {
    __shared__ uint16_t key16[kSegEdges], oth16[kSegEdges], tmp[kSegEdges];
    __shared__ int32_t excl[kSegNodes + 4], cur[kSegNodes], degin[kSegNodes];
    const int seg = blockIdx.x, side = blockIdx.y, tid = threadIdx.x;   // side 0: by destination, 1: by source
    const int n0 = b.node_ptr[seg], e0 = b.edge_ptr[seg];
    const int nn = b.node_ptr[seg + 1] - n0, ne = b.edge_ptr[seg + 1] - e0;
    for (int i = tid; i < nn; i += 1024) cur[i] = 0, degin[i] = 0;
    __syncthreads();
    bool bad = false;
    const int64_t *__restrict__ gs = b.src + e0, *__restrict__ gd = b.dst + e0;
    for (int base = 0; base < ne; base += 8 * 1024) {
        int64_t sv[8], dv[8];                          // all loads of the chunk in flight before the first use
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = base + j * 1024 + tid;
            sv[j] = dv[j] = 0;
            if (e < ne) sv[j] = gs[e], dv[j] = gd[e];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = base + j * 1024 + tid;
            if (e >= ne) continue;
            int64_t s = sv[j] - n0, d = dv[j] - n0;
            if (s < 0 || s >= nn || d < 0 || d >= nn) bad = true, s = d = 0;   // flagged; placed on node 0 so that
            const int k = (int)(side ? s : d), o = (int)(side ? d : s);        // every output slot is still written
            key16[e] = (uint16_t)k, oth16[e] = (uint16_t)o;
            atomicAdd(&cur[k], 1);
            if (side) atomicAdd(&degin[d], 1);         // gcn_norm degrees are in-degrees (= side 0's own counts)
        }
    }
    if (bad) atomicOr(b.status, 1);
    __syncthreads();
    int v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * tid + j;
        v[j] = i < nn ? cur[i] : 0;
        sum += v[j];
    }
    int total;
    int run = block_incl_scan<16>(sum, &total) - sum;
    int32_t *__restrict__ ptr = b.ptr[side];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * tid + j;
        if (i < nn) {
            excl[i] = run, cur[i] = run;
            if (!side) degin[i] = v[j];
            ptr[n0 + i] = e0 + run;
        }
        run += v[j];
    }
    if (tid == 0) {
        excl[nn] = total;
        if (b.last && seg == b.nseg - 1) ptr[b.N] = (int32_t)b.E;
    }
    __syncthreads();
    for (int e = tid; e < ne; e += 1024) tmp[atomicAdd(&cur[key16[e]], 1)] = (uint16_t)e;
    __syncthreads();
    int32_t *__restrict__ other = b.other[side], *__restrict__ perm = b.perm[side];
    float *__restrict__ w = b.w[side];
    for (int p = tid; p < ne; p += 1024) {
        const int e = tmp[p], k = key16[e], o = oth16[e];
        const int beg = excl[k], end = excl[k + 1];
        int rank = 0;
        for (int q = beg; q < end; ++q) rank += tmp[q] < e;
        const int64_t out = (int64_t)e0 + beg + rank;
        perm[out] = e0 + e;
        other[out] = n0 + o;
        if (w) {
            const int sl = side ? k : o, dl = side ? o : k;      // gcn_norm: dis[source] * 1 * dis[destination]
            w[out] = inv_sqrt_count(degin[sl]) * 1.0f * inv_sqrt_count(degin[dl]);
        }
    }
}

__global__ void __launch_bounds__(256)
k_invert_perm(const int32_t *perm, const int32_t *n_ptr, int32_t *pos_of, int64_t max_edges) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < max_edges && p < *n_ptr) pos_of[perm[p]] = (int32_t)p;
}

// order-independent 64-bit content hash of an int64 array (sum of mixed (value, position) words)
__global__ void __launch_bounds__(256)
k_hash_i64(const int64_t *v, int64_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)v[i] * 0x9E3779B97F4A7C15ull +
                               ((unsigned long long)i + 1) * 0xC2B2AE3D27D4EB4Full;
        z ^= z >> 29;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 32;
        acc += z;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// 30-bit Morton (Z-order) code of each point: 10 bits per axis after mapping [lo, lo + 1/inv) to
// [0, 1024); sorting nodes by it puts spatial neighbours - the rows a radius graph's hop gathers -
// next to each other, i.e. into the same XCD's L2 under the hop's contiguous block -> XCD chunks
__device__ __forceinline__ unsigned spread10(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(256)
k_morton(const float *pos, int64_t ld, int64_t n, float lx, float ly, float lz, float sx, float sy,
         float sz, int64_t *codes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto q = [](float v, float lo, float sc) {
        const float t = (v - lo) * sc * 1024.0f;
        return (unsigned)(t < 0.f ? 0.f : (t > 1023.f ? 1023.f : t));
    };
    const unsigned a = q(pos[i * ld], lx, sx), b = q(pos[i * ld + 1], ly, sy), c = q(pos[i * ld + 2], lz, sz);
    codes[i] = (int64_t)(spread10(a) | (spread10(b) << 1) | (spread10(c) << 2));
}

// ---- bucketed build (r04): edge lists in ARBITRARY order -----------------------------------------------------------
// The pipeline above is fast when consecutive edges name nearby nodes (mesh batches: every key inside the block's
// LDS window).  A radius graph whose nodes were relabelled (Morton order) keeps its edge ORDER: 92 % of the keys of
// the 100k-point graph fall outside any window (tools/exp/csr_ab.sh), and every one of them costs a device-scope
// atomic - k_count 78 us, of which 70 us are those atomics (8 us without them), k_fill 112 us, k_emit (dependent
// gathers through tmp / edge_index / ptr) 62 us for 1.1 M edges: ~10 G scattered atomics per second is what the chip
// gives, whatever the XCD mapping.  This build does not issue one per edge.  Nodes are cut into buckets of 2^shift
// consecutive ids; a partition pass (histogram per workgroup in LDS, ONE global atomic per workgroup and non-empty
// bucket) moves (edge id, other endpoint, key - bucket base) records into per-bucket ranges - which are the bucket's
// final slot range too, since ptr is monotone in the node id - and ONE workgroup per (bucket, side) does the rest in
// LDS: degrees, their scan (= ptr), slots by LDS cursors, rank of every id inside its group, perm / other.  Buckets
// larger than the LDS pass are processed in several passes over runs of consecutive nodes; a single node beyond it is
// sorted in place in its output range.  The weights need both endpoints' degrees: a last pass over the finished ptr.
// Same arrays as the pipeline above, bit for bit (the stable sort by key is unique).
constexpr int kBkEdges = 12288;       // slots of one pass of a bucket held in LDS (8 B record + 2 B local key)
constexpr int kBkMaxShift = 11;       // <= 2048 nodes per bucket (LDS offsets + cursors)
constexpr int kBkNodes = 1 << kBkMaxShift;
constexpr int kBkMaxBuckets = 8192;   // per side (LDS histograms of the partition passes: 2 x 32 KiB)
constexpr int kBkEpt = 4;             // slots per thread of the partition passes (1024 threads: 4096 per workgroup)

struct Buckets {
    int shift, nb;         // 2^shift nodes per bucket, nb buckets per side
};

// slot t of the build - an input edge (t < E) or an appended self loop -> key of side 0 and the other endpoint;
// false when the slot places nothing (invalid ids: *bad; an input self loop that is being replaced; padding)
__device__ __forceinline__ bool slot_keys(const Build &b, int64_t t, int32_t &k, int32_t &o, bool &bad) {
    if (t < b.E) {
        int64_t src, dst;
        if (!load_edge(b.ed, t, src, dst)) {
            bad = true;
            return false;
        }
        k = (int32_t)(b.s[0].key_is_dst ? dst : src), o = (int32_t)(b.s[0].key_is_dst ? src : dst);
        return !(b.self_loops && k == o);
    }
    if (b.self_loops && t < b.E + b.N) {
        k = o = (int32_t)(t - b.E);
        return true;
    }
    return false;
}

__global__ void __launch_bounds__(1024)
k_bk_count(Build b, Buckets bk, int nsides) {
    __shared__ int32_t hist[2][kBkMaxBuckets];
    for (int i = threadIdx.x; i < bk.nb; i += 1024) hist[0][i] = 0, hist[1][i] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * (1024 * kBkEpt) + threadIdx.x;
    int32_t k[kBkEpt], o[kBkEpt];
    bool ok[kBkEpt], bad = false;
#pragma unroll
    for (int j = 0; j < kBkEpt; ++j) ok[j] = slot_keys(b, t0 + j * 1024, k[j], o[j], bad);
#pragma unroll
    for (int j = 0; j < kBkEpt; ++j) {
        if (!ok[j]) continue;
        atomicAdd(&hist[0][k[j] >> bk.shift], 1);
        if (nsides == 2) atomicAdd(&hist[1][o[j] >> bk.shift], 1);
    }
    if (bad) atomicOr(b.status, 1);
    __syncthreads();
    for (int sd = 0; sd < nsides; ++sd)
        for (int i = threadIdx.x; i < bk.nb; i += 1024) {
            const int c = hist[sd][i];
            if (c) atomicAdd(&b.s[sd].bk_start[i], c);
        }
}

__global__ void __launch_bounds__(1024)
k_bk_scan(Build b, Buckets bk) {
    const int sd = blockIdx.x, tid = threadIdx.x;
    constexpr int kPer = kBkMaxBuckets / 1024;
    int v[kPer], sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = kPer * tid + j;
        v[j] = i < bk.nb ? b.s[sd].bk_start[i] : 0;
        sum += v[j];
    }
    int total;
    int run = block_incl_scan<16>(sum, &total) - sum;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = kPer * tid + j;
        if (i < bk.nb) b.s[sd].bk_start[i] = run, b.s[sd].bk_cur[i] = run;
        run += v[j];
    }
    if (tid == 0) b.s[sd].bk_start[bk.nb] = total, b.s[sd].ptr[b.N] = total;
}

__global__ void __launch_bounds__(1024)
k_bk_scatter(Build b, Buckets bk, int nsides) {
    __shared__ int32_t hist[2][kBkMaxBuckets];
    for (int i = threadIdx.x; i < bk.nb; i += 1024) hist[0][i] = 0, hist[1][i] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * (1024 * kBkEpt) + threadIdx.x;
    int32_t k[kBkEpt], o[kBkEpt], s0[kBkEpt], s1[kBkEpt];
    bool ok[kBkEpt], bad = false;                          // (invalid ids were flagged by k_bk_count)
#pragma unroll
    for (int j = 0; j < kBkEpt; ++j) ok[j] = slot_keys(b, t0 + j * 1024, k[j], o[j], bad);
#pragma unroll
    for (int j = 0; j < kBkEpt; ++j) {                     // place inside the workgroup's share of the bucket
        s0[j] = s1[j] = 0;
        if (!ok[j]) continue;
        s0[j] = atomicAdd(&hist[0][k[j] >> bk.shift], 1);
        if (nsides == 2) s1[j] = atomicAdd(&hist[1][o[j] >> bk.shift], 1);
    }
    __syncthreads();
    for (int sd = 0; sd < nsides; ++sd)                    // share sizes -> reserved first record
        for (int i = threadIdx.x; i < bk.nb; i += 1024) {
            const int c = hist[sd][i];
            if (c) hist[sd][i] = atomicAdd(&b.s[sd].bk_cur[i], c);
        }
    __syncthreads();
    const int32_t mask = (1 << bk.shift) - 1;
#pragma unroll
    for (int j = 0; j < kBkEpt; ++j) {
        if (!ok[j]) continue;
        const int32_t t = (int32_t)(t0 + j * 1024);
        int pos = hist[0][k[j] >> bk.shift] + s0[j];
        b.s[0].bk_eid[pos] = t, b.s[0].bk_oth[pos] = o[j], b.s[0].bk_kl[pos] = (uint16_t)(k[j] & mask);
        if (nsides == 2) {
            pos = hist[1][o[j] >> bk.shift] + s1[j];
            b.s[1].bk_eid[pos] = t, b.s[1].bk_oth[pos] = k[j], b.s[1].bk_kl[pos] = (uint16_t)(o[j] & mask);
        }
    }
}

__global__ void __launch_bounds__(1024)
k_bk_build(Build b, Buckets bk) {
    __shared__ unsigned long long rec[kBkEdges];            // (edge id << 32) | other endpoint, by slot of the pass
    __shared__ uint16_t skl[kBkEdges];                      // local key of the slot
    __shared__ int32_t excl[kBkNodes + 4], cur[kBkNodes], biglist[kBkNodes];
    __shared__ int nbig;
    const int sd = blockIdx.y, bkt = blockIdx.x, tid = threadIdx.x;
    const Side &S = b.s[sd];
    const int32_t n0 = bkt << bk.shift;
    const int nn = (int)(b.N - n0 < (1 << bk.shift) ? b.N - n0 : (1 << bk.shift));
    const int32_t r0 = b.s[sd].bk_start[bkt], n = b.s[sd].bk_start[bkt + 1] - r0;
    const int32_t *__restrict__ ge = b.s[sd].bk_eid + r0, *__restrict__ go = b.s[sd].bk_oth + r0;
    const uint16_t *__restrict__ gk = b.s[sd].bk_kl + r0;
    // degrees of the bucket's nodes, their scan = this bucket's slice of ptr
    for (int i = tid; i < nn; i += 1024) cur[i] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 8 * 1024) {             // (a chunk's loads in flight before the first use: one
        uint16_t kv[8];                                    // workgroup per CU - what hides the latency is the
#pragma unroll                                             // number of loads a wave has outstanding)
        for (int j = 0; j < 8; ++j) {
            const int i = c0 + j * 1024 + tid;
            kv[j] = i < n ? gk[i] : (uint16_t)0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j * 1024 + tid < n) atomicAdd(&cur[kv[j]], 1);
    }
    __syncthreads();
    constexpr int kPer = kBkNodes / 1024;
    int v[kPer], sum = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = kPer * tid + j;
        v[j] = i < nn ? cur[i] : 0;
        sum += v[j];
    }
    int total;
    int run = block_incl_scan<16>(sum, &total) - sum;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = kPer * tid + j;
        if (i < nn) excl[i] = run, S.ptr[n0 + i] = r0 + run;
        run += v[j];
    }
    if (tid == 0) excl[nn] = total;
    __syncthreads();
    auto emit = [&](unsigned long long r, int kl, int32_t out) {
        S.perm[out] = (int32_t)(r >> 32);
        S.other[out] = (int32_t)(uint32_t)r;
        if (S.w) S.w[out] = __int_as_float(n0 + kl);       // the slot's key, for k_bk_weights
    };
    for (int lo = 0; lo < nn;) {
        // the longest run of nodes [lo, hi) whose slots fit one LDS pass (uniform binary search in the offsets)
        const int base = excl[lo];
        int hi = lo, z = nn;
        while (hi < z) {
            const int m = (hi + z + 1) >> 1;
            if (excl[m] - base <= kBkEdges) hi = m;
            else z = m - 1;
        }
        if (hi == lo) {
            // one node with more slots than a pass holds: its ids go to its perm range in arrival order, the
            // workgroup sorts them there (in place in global memory, as k_emit does for its longest groups) and
            // looks the other endpoints up again
            const int deg = excl[lo + 1] - base;
            volatile int32_t *vs = S.perm + r0 + base;
            if (tid == 0) nbig = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 1024)
                if (gk[i] == lo) vs[atomicAdd(&nbig, 1)] = ge[i];
            __syncthreads();
            bitonic_sort(deg, [&](int i) { return (int32_t)vs[i]; }, [&](int i, int32_t x) { vs[i] = x; });
            for (int q = tid; q < deg; q += 1024) {
                const int32_t eid = vs[q];
                int64_t o = n0 + lo;
                if (eid < b.E) {
                    int64_t src, dst;
                    load_edge(b.ed, eid, src, dst);
                    o = S.key_is_dst ? src : dst;
                }
                S.other[r0 + base + q] = (int32_t)o;
                if (S.w) S.w[r0 + base + q] = __int_as_float(n0 + lo);
            }
            __syncthreads();
            lo += 1;
            continue;
        }
        const int m = excl[hi] - base;
        if (m > 0) {
            for (int i = lo + tid; i < hi; i += 1024) cur[i] = excl[i] - base;
            if (tid == 0) nbig = 0;
            __syncthreads();
            for (int c0 = 0; c0 < n; c0 += 4 * 1024) {
                int kv[4];
                int32_t ev[4], ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = c0 + j * 1024 + tid;
                    kv[j] = -1, ev[j] = ov[j] = 0;
                    if (i < n) kv[j] = gk[i], ev[j] = ge[i], ov[j] = go[i];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kv[j] < lo || kv[j] >= hi) continue;
                    const int pos = atomicAdd(&cur[kv[j]], 1);
                    rec[pos] = ((unsigned long long)(uint32_t)ev[j] << 32) | (uint32_t)ov[j];
                    skl[pos] = (uint16_t)kv[j];
                }
            }
            __syncthreads();
            for (int p = tid; p < m; p += 1024) {
                const int kl = skl[p];
                const unsigned long long r = rec[p];
                const int gb = excl[kl] - base, gend = excl[kl + 1] - base;
                if (gend - gb > kRankLoop) {                // a long group: sorted by the workgroup below
                    if (p == gb) biglist[atomicAdd(&nbig, 1)] = kl;
                    continue;
                }
                int rank = 0;
                int q = gb;
                for (; q + 4 <= gend; q += 4) {             // four independent LDS reads per trip
                    const unsigned long long a0 = rec[q], a1 = rec[q + 1], a2 = rec[q + 2], a3 = rec[q + 3];
                    rank += (a0 < r) + (a1 < r) + (a2 < r) + (a3 < r);
                }
                for (; q < gend; ++q) rank += rec[q] < r;
                emit(r, kl, r0 + base + gb + rank);
            }
            __syncthreads();
            for (int g = 0; g < nbig; ++g) {
                const int kl = biglist[g], gb = excl[kl] - base, len = excl[kl + 1] - excl[kl];
                bitonic_sort(len, [&](int i) { return rec[gb + i]; }, [&](int i, unsigned long long x) { rec[gb + i] = x; });
                for (int q = tid; q < len; q += 1024) emit(rec[gb + q], kl, r0 + base + gb + q);
                __syncthreads();
            }
        }
        lo = hi;
    }
}

// gcn_norm weights of a finished side: dis[source] * 1 * dis[destination] with the degrees deg_ptr names (the
// expression of emit_one).  One thread per slot; k_bk_build left the slot's key in w[slot].
__global__ void __launch_bounds__(256)
k_bk_weights(Build b) {
    const Side &sd = b.s[blockIdx.y];
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!sd.w || p >= sd.ptr[b.N]) return;
    const float dk = inv_sqrt_deg(sd.deg_ptr, __float_as_int(sd.w[p])), dother = inv_sqrt_deg(sd.deg_ptr, sd.other[p]);
    sd.w[p] = sd.key_is_dst ? dother * 1.0f * dk : dk * 1.0f * dother;
}

static inline int64_t align16(int64_t b) { return (b + 15) & ~int64_t(15); }

// Can a build over E edges and N nodes take the bucketed path (bucket_plan below)?  Its slots are E, or E + N with self
// loops - the workspace entries do not know which, so E + N decides.  Mesh batches (the reference's configurations) stay
// far below the threshold and do not pay for the bucket arrays (~10 B x (E + N) + 64 KiB per side).
static bool may_bucket(int64_t E, int64_t N) {
    const char *mode = getenv("DC_CSR_BUCKETS"), *min_s = getenv("DC_CSR_BUCKETS_MIN");
    if (mode && atoi(mode) == 0) return false;
    if (mode && atoi(mode) == 1) return true;
    return E + N >= (min_s ? atoll(min_s) : (int64_t)1 << 19);
}

static inline int64_t side_bytes(int64_t E, int64_t N) {
    const int64_t ntiles = (N + kScanTile - 1) / kScanTile;
    const int64_t windowed = 2 * align16(4 * (N + 4)) + align16(4 * (E + N + 1)) + align16(4 * (ntiles + 2)) +
                             align16(4 * (N + 2));
    if (!may_bucket(E, N)) return windowed;
    return windowed + 2 * align16(4 * (kBkMaxBuckets + 1)) + 2 * align16(4 * (E + N)) + align16(2 * (E + N));   // bucketed build
}

static char *carve_side(Side &sd, char *ws, int64_t E, int64_t N) {
    const int64_t ntiles = (N + kScanTile - 1) / kScanTile;
    sd.cnt = (int32_t *)ws;
    ws += align16(4 * (N + 4));
    sd.cur = (int32_t *)ws;
    ws += align16(4 * (N + 4));
    sd.tmp = (int32_t *)ws;
    ws += align16(4 * (E + N + 1));
    sd.tiles = (int32_t *)ws;
    ws += align16(4 * (ntiles + 2));
    sd.big = (int32_t *)ws;
    ws += align16(4 * (N + 2));
    if (!may_bucket(E, N)) {                    // (run_build cannot choose the bucketed path then: slots <= E + N)
        sd.bk_start = sd.bk_cur = sd.bk_eid = sd.bk_oth = nullptr;
        sd.bk_kl = nullptr;
        return ws;
    }
    sd.bk_start = (int32_t *)ws;
    ws += align16(4 * (kBkMaxBuckets + 1));
    sd.bk_cur = (int32_t *)ws;
    ws += align16(4 * (kBkMaxBuckets + 1));
    sd.bk_eid = (int32_t *)ws;
    ws += align16(4 * (E + N));
    sd.bk_oth = (int32_t *)ws;
    ws += align16(4 * (E + N));
    sd.bk_kl = (uint16_t *)ws;
    ws += align16(2 * (E + N));
    return ws;
}

// The bucketed build is for edge lists the windowed pipeline cannot serve: large ones (DC_CSR_BUCKETS_MIN slots,
// default 2^19 - above the mesh batches of the reference's configurations, merged branches and self loops included,
// which keep their order and are served by the windowed pipeline in ~30 us; the host cannot see the order).  Nodes per bucket: a
// power of two such that an average bucket fills at most a quarter of one LDS pass (the dense regions of a point
// cloud hold several times the average; 100k-point graph: 256 nodes, 154 us against 169 us with 512).  DC_CSR_BUCKETS = 0 / 1 forces the
// choice, DC_CSR_BUCKET_SHIFT the bucket size (tests: multi-pass buckets and single-node passes at small sizes).
static bool bucket_plan(int64_t slots, int64_t N, Buckets &bk) {
    const char *mode = getenv("DC_CSR_BUCKETS"), *min_s = getenv("DC_CSR_BUCKETS_MIN"), *sh = getenv("DC_CSR_BUCKET_SHIFT");
    if (mode && atoi(mode) == 0) return false;
    const int64_t min_slots = min_s ? atoll(min_s) : (int64_t)1 << 19;
    if (!(mode && atoi(mode) == 1) && slots < min_slots) return false;
    int s = 4;
    while (s < kBkMaxShift && ((int64_t)2 << s) * slots <= (int64_t)(kBkEdges / 4) * N) ++s;
    if (sh) s = atoi(sh) < 1 ? 1 : (atoi(sh) > kBkMaxShift ? kBkMaxShift : atoi(sh));
    while (s < kBkMaxShift && ((N + ((int64_t)1 << s) - 1) >> s) > kBkMaxBuckets) ++s;
    const int64_t nb = (N + ((int64_t)1 << s) - 1) >> s;
    if (nb > kBkMaxBuckets) return false;
    bk.shift = s, bk.nb = (int)nb;
    return true;
}

static int run_build(Build &b, int nsides, hipStream_t stream, const char *what) {
    const int64_t E = b.E, N = b.N;
    if (N == 0) {
        for (int s = 0; s < nsides; ++s) hipMemsetAsync(b.s[s].ptr, 0, sizeof(int32_t), stream);
        hipMemsetAsync(b.status, 0, sizeof(int32_t), stream);
        return check_launch(what);
    }
    const int64_t ntiles = (N + kScanTile - 1) / kScanTile;
    const int64_t slots = E + (b.self_loops ? N : 0);
    Buckets bk{};
    if (slots > 0 && bucket_plan(slots, N, bk)) {
        for (int s = 0; s < nsides; ++s)
            DC_REQUIRE(b.s[s].bk_start, "%s: the workspace was sized without the bucket arrays (DC_CSR_BUCKETS* changed "
                                        "between sizing and building?)", what);
        for (int s = 0; s < nsides; ++s) hipMemsetAsync(b.s[s].bk_start, 0, 4 * (size_t)(bk.nb + 1), stream);
        hipMemsetAsync(b.status, 0, sizeof(int32_t), stream);
        const unsigned pb = (unsigned)((slots + 1024 * kBkEpt - 1) / (1024 * kBkEpt));
        DC_LAUNCH(k_bk_count, dim3(pb), dim3(1024), 0, stream, b, bk, nsides);
        DC_LAUNCH(k_bk_scan, dim3(nsides), dim3(1024), 0, stream, b, bk);
        DC_LAUNCH(k_bk_scatter, dim3(pb), dim3(1024), 0, stream, b, bk, nsides);
        DC_LAUNCH(k_bk_build, dim3((unsigned)bk.nb, nsides), dim3(1024), 0, stream, b, bk);
        bool any_w = false;
        for (int s = 0; s < nsides; ++s) any_w = any_w || b.s[s].w;
        if (any_w) DC_LAUNCH(k_bk_weights, dim3((unsigned)((slots + 255) / 256), nsides), dim3(256), 0, stream, b);
        return check_launch(what);
    }
    DC_LAUNCH(k_init, dim3((unsigned)((N + 255) / 256), nsides), dim3(256), 0, stream, b);
    if (E > 0)
        DC_LAUNCH(k_count, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, b, nsides);
    bool ptr16 = true;                          // k_scan_small stores ptr with 16-byte accesses
    for (int s = 0; s < nsides; ++s) ptr16 = ptr16 && (((uintptr_t)b.s[s].ptr & 15) == 0);
    if (N <= kScanSmall && ptr16) {
        DC_LAUNCH(k_scan_small, dim3(nsides), dim3(1024), 0, stream, b);
    } else {
        DC_LAUNCH(k_scan_reduce, dim3((unsigned)ntiles, nsides), dim3(256), 0, stream, b);
        DC_LAUNCH(k_scan_tiles, dim3(nsides), dim3(256), 0, stream, b, ntiles);
        DC_LAUNCH(k_scan_apply, dim3((unsigned)ntiles, nsides), dim3(256), 0, stream, b, ntiles);
    }
    if (slots > 0) {
        const unsigned sb = (unsigned)((slots + 255) / 256);
        DC_LAUNCH(k_fill, dim3(sb), dim3(256), 0, stream, b, nsides);
        DC_LAUNCH(k_emit, dim3(sb + kBigBlocks, nsides), dim3(256), 0, stream, b, sb);
    }
    return check_launch(what);
}

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return DC_EINVAL;
    return side_bytes(E, N);
}

extern "C" int dc_graph_build_plan(int64_t E, int64_t N, int self_loops, int *bucket_shift, int *buckets) {
    DC_REQUIRE(E >= 0 && N >= 0, "dc_graph_build_plan: negative size E=%lld N=%lld", (long long)E, (long long)N);
    Buckets bk{};
    const int64_t slots = E + (self_loops ? N : 0);
    if (N == 0 || slots == 0 || !bucket_plan(slots, N, bk)) return 0;
    if (bucket_shift) *bucket_shift = bk.shift;
    if (buckets) *buckets = bk.nb;
    return 1;
}

extern "C" int64_t dc_graph_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return DC_EINVAL;
    return 2 * side_bytes(E, N);
}

extern "C" int dc_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int key_row,
                            int self_loops, int32_t *ptr, int32_t *other, int32_t *perm,
                            const int32_t *deg_ptr, float *w, int32_t *status, void *workspace,
                            int64_t workspace_bytes, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(E >= 0 && N >= 0, "dc_csr_build: negative size E=%lld N=%lld", (long long)E,
               (long long)N);
    DC_REQUIRE(E + N < (int64_t)INT32_MAX, "dc_csr_build: E+N=%lld exceeds int32 indexing",
               (long long)(E + N));
    DC_REQUIRE(key_row == 0 || key_row == 1, "dc_csr_build: key_row must be 0 or 1");
    DC_REQUIRE(ptr && status && workspace, "dc_csr_build: null ptr/status/workspace");
    DC_REQUIRE(E == 0 || (edge_index && other && perm), "dc_csr_build: null edge arrays");
    DC_REQUIRE(!(w && !deg_ptr && key_row != 1),
               "dc_csr_build: w with key_row=0 needs deg_ptr (the key_row=1 ptr)");
    DC_REQUIRE(workspace_bytes >= dc_csr_workspace_bytes(E, N),
               "dc_csr_build: workspace too small (%lld < %lld)", (long long)workspace_bytes,
               (long long)dc_csr_workspace_bytes(E, N));
    DC_REQUIRE(((uintptr_t)workspace & 15) == 0, "dc_csr_build: workspace not 16-byte aligned");

    Build b{};
    b.E = E, b.N = N, b.self_loops = self_loops, b.status = status;
    Side &sd = b.s[0];
    b.ed.nparts = 1, b.ed.src[0] = edge_index, b.ed.dst[0] = edge_index + E;
    b.ed.e_beg[0] = 0, b.ed.e_beg[1] = E, b.ed.node_off[0] = 0, b.ed.nodes[0] = N;
    carve_side(sd, (char *)workspace, E, N);
    sd.ptr = ptr, sd.other = other, sd.perm = perm, sd.w = w;
    sd.deg_ptr = deg_ptr ? deg_ptr : ptr;
    sd.key_is_dst = key_row;
    return run_build(b, 1, stream, "dc_csr_build");
}

extern "C" int dc_graph_build(const int64_t *edge_index, int64_t E, int64_t N, int self_loops,
                              int32_t *ptr_f, int32_t *other_f, int32_t *perm_f, float *w_f,
                              int32_t *ptr_b, int32_t *other_b, int32_t *perm_b, float *w_b,
                              int32_t *status, void *workspace, int64_t workspace_bytes,
                              dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(E >= 0 && N >= 0, "dc_graph_build: negative size E=%lld N=%lld", (long long)E,
               (long long)N);
    DC_REQUIRE(E + N < (int64_t)INT32_MAX, "dc_graph_build: E+N=%lld exceeds int32 indexing",
               (long long)(E + N));
    DC_REQUIRE(ptr_f && ptr_b && status && workspace, "dc_graph_build: null ptr/status/workspace");
    DC_REQUIRE(E == 0 || (edge_index && other_f && perm_f && other_b && perm_b),
               "dc_graph_build: null edge arrays");
    DC_REQUIRE((w_f == nullptr) == (w_b == nullptr), "dc_graph_build: w_f and w_b go together");
    DC_REQUIRE(workspace_bytes >= dc_graph_workspace_bytes(E, N),
               "dc_graph_build: workspace too small (%lld < %lld)", (long long)workspace_bytes,
               (long long)dc_graph_workspace_bytes(E, N));
    DC_REQUIRE(((uintptr_t)workspace & 15) == 0, "dc_graph_build: workspace not 16-byte aligned");

    Build b{};
    b.E = E, b.N = N, b.self_loops = self_loops, b.status = status;
    char *ws = (char *)workspace;
    Side &f = b.s[0], &t = b.s[1];                       // f: by destination, t: by source
    b.ed.nparts = 1, b.ed.src[0] = edge_index, b.ed.dst[0] = edge_index + E;
    b.ed.e_beg[0] = 0, b.ed.e_beg[1] = E, b.ed.node_off[0] = 0, b.ed.nodes[0] = N;
    ws = carve_side(f, ws, E, N);
    carve_side(t, ws, E, N);
    f.ptr = ptr_f, f.other = other_f, f.perm = perm_f, f.w = w_f, f.deg_ptr = ptr_f, f.key_is_dst = 1;
    t.ptr = ptr_b, t.other = other_b, t.perm = perm_b, t.w = w_b, t.deg_ptr = ptr_f, t.key_is_dst = 0;
    return run_build(b, 2, stream, "dc_graph_build");
}

extern "C" int dc_graph_build_parts(const int64_t *const *edge_index_parts, const int64_t *E_parts,
                                    const int64_t *node_offset_parts, const int64_t *nodes_parts,
                                    int nparts, int64_t N, int self_loops,
                                    int32_t *ptr_f, int32_t *other_f, int32_t *perm_f, float *w_f,
                                    int32_t *ptr_b, int32_t *other_b, int32_t *perm_b, float *w_b,
                                    int32_t *status, void *workspace, int64_t workspace_bytes,
                                    dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(nparts >= 1 && nparts <= kMaxParts && edge_index_parts && E_parts && node_offset_parts &&
                   nodes_parts && N >= 0,
               "dc_graph_build_parts: 1..%d parts with their sizes and offsets required", kMaxParts);
    Build b{};
    int64_t E = 0;
    for (int p = 0; p < nparts; ++p) {
        DC_REQUIRE(E_parts[p] >= 0 && nodes_parts[p] >= 0 && node_offset_parts[p] >= 0 &&
                       node_offset_parts[p] + nodes_parts[p] <= N && (E_parts[p] == 0 || edge_index_parts[p]),
                   "dc_graph_build_parts: part %d: bad size / offset / null edge_index", p);
        DC_REQUIRE(p == 0 || node_offset_parts[p] >= node_offset_parts[p - 1] + nodes_parts[p - 1],
                   "dc_graph_build_parts: part %d overlaps the previous one (offsets must ascend)", p);
        b.ed.src[p] = edge_index_parts[p], b.ed.dst[p] = edge_index_parts[p] + E_parts[p];
        b.ed.e_beg[p] = E, b.ed.node_off[p] = node_offset_parts[p], b.ed.nodes[p] = nodes_parts[p];
        E += E_parts[p];
    }
    for (int p = nparts; p <= kMaxParts; ++p) b.ed.e_beg[p] = E;
    b.ed.nparts = nparts;
    DC_REQUIRE(E + N < (int64_t)INT32_MAX, "dc_graph_build_parts: E+N=%lld exceeds int32 indexing",
               (long long)(E + N));
    DC_REQUIRE(ptr_f && ptr_b && status && workspace, "dc_graph_build_parts: null ptr/status/workspace");
    DC_REQUIRE(E == 0 || (other_f && perm_f && other_b && perm_b), "dc_graph_build_parts: null edge arrays");
    DC_REQUIRE((w_f == nullptr) == (w_b == nullptr), "dc_graph_build_parts: w_f and w_b go together");
    DC_REQUIRE(workspace_bytes >= dc_graph_workspace_bytes(E, N),
               "dc_graph_build_parts: workspace too small (%lld < %lld)", (long long)workspace_bytes,
               (long long)dc_graph_workspace_bytes(E, N));
    DC_REQUIRE(((uintptr_t)workspace & 15) == 0, "dc_graph_build_parts: workspace not 16-byte aligned");
    b.E = E, b.N = N, b.self_loops = self_loops, b.status = status;
    char *ws = (char *)workspace;
    Side &f = b.s[0], &t = b.s[1];
    ws = carve_side(f, ws, E, N);
    carve_side(t, ws, E, N);
    f.ptr = ptr_f, f.other = other_f, f.perm = perm_f, f.w = w_f, f.deg_ptr = ptr_f, f.key_is_dst = 1;
    t.ptr = ptr_b, t.other = other_b, t.perm = perm_b, t.w = w_b, t.deg_ptr = ptr_f, t.key_is_dst = 0;
    return run_build(b, 2, stream, "dc_graph_build_parts");
}

extern "C" int dc_graph_build_segmented(const int64_t *edge_index, int64_t E, int64_t N,
                                        const int64_t *node_ptr_host, const int64_t *edge_ptr_host, int nseg,
                                        int32_t *ptr_f, int32_t *other_f, int32_t *perm_f, float *w_f,
                                        int32_t *ptr_b, int32_t *other_b, int32_t *perm_b, float *w_b,
                                        int32_t *status, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(E > 0 && N > 0 && nseg >= 1, "dc_graph_build_segmented: needs E > 0, N > 0 and >= 1 graph");
    DC_REQUIRE(E + N < (int64_t)INT32_MAX, "dc_graph_build_segmented: E+N=%lld exceeds int32 indexing",
               (long long)(E + N));
    DC_REQUIRE(edge_index && node_ptr_host && edge_ptr_host && ptr_f && ptr_b && other_f && perm_f && other_b &&
                   perm_b && status, "dc_graph_build_segmented: null pointer");
    DC_REQUIRE((w_f == nullptr) == (w_b == nullptr), "dc_graph_build_segmented: w_f and w_b go together");
    DC_REQUIRE(node_ptr_host[0] == 0 && edge_ptr_host[0] == 0 && node_ptr_host[nseg] == N && edge_ptr_host[nseg] == E,
               "dc_graph_build_segmented: the graphs' offsets must cover [0, N] and [0, E]");
    for (int i = 0; i < nseg; ++i) {
        const int64_t dn = node_ptr_host[i + 1] - node_ptr_host[i], de = edge_ptr_host[i + 1] - edge_ptr_host[i];
        DC_REQUIRE(dn >= 0 && de >= 0, "dc_graph_build_segmented: offsets of graph %d descend", i);
        DC_REQUIRE(dn > 0 || de == 0, "dc_graph_build_segmented: graph %d has %lld edges but no node: use dc_graph_build",
                   i, (long long)de);
        DC_REQUIRE(dn <= kSegNodes && de <= kSegEdges,
                   "dc_graph_build_segmented: graph %d (%lld nodes, %lld edges) exceeds the per-graph caps (%d / %d): "
                   "use dc_graph_build", i, (long long)dn, (long long)de, kSegNodes, kSegEdges);
    }
    // NOTE: the kernel only ORs flags into *status (workgroups of one launch cannot order a clear before their ORs, and a
    // memset node per build would sit on the critical path of every training step): the word is STICKY across rebuilds
    // into the same buffers until the caller clears it (GraphIndex.validate() does after reporting) - see the header
    SegBuild b{};
    b.src = edge_index, b.dst = edge_index + E, b.E = E, b.N = N, b.status = status;
    b.ptr[0] = ptr_f, b.other[0] = other_f, b.perm[0] = perm_f, b.w[0] = w_f;
    b.ptr[1] = ptr_b, b.other[1] = other_b, b.perm[1] = perm_b, b.w[1] = w_b;
    for (int first = 0; first < nseg; first += kSegChunk) {
        const int cnt = nseg - first < kSegChunk ? nseg - first : kSegChunk;
        for (int i = 0; i <= cnt; ++i) {
            b.node_ptr[i] = (int32_t)node_ptr_host[first + i];
            b.edge_ptr[i] = (int32_t)edge_ptr_host[first + i];
        }
        b.nseg = cnt, b.last = first + cnt == nseg;
        DC_LAUNCH(k_build_segment, dim3((unsigned)cnt, 2), dim3(1024), 0, stream, b);
    }
    return check_launch("dc_graph_build_segmented");
}

extern "C" int dc_invert_perm(const int32_t *perm, const int32_t *ptr_last, int32_t *pos_of,
                              int64_t max_edges, dc_stream_t stream) {
    DC_REQUIRE(max_edges >= 0, "dc_invert_perm: negative size");
    if (max_edges == 0) return DC_OK;
    DC_REQUIRE(perm && ptr_last && pos_of, "dc_invert_perm: null pointer");
    DC_LAUNCH(k_invert_perm, dim3((max_edges + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, perm, ptr_last, pos_of, max_edges);
    return check_launch("dc_invert_perm");
}

extern "C" int dc_hash_i64(const int64_t *v, int64_t n, uint64_t *out, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(n >= 0 && out, "dc_hash_i64: negative size or null out");
    hipMemsetAsync(out, 0, sizeof(uint64_t), stream);
    if (n == 0) return check_launch("dc_hash_i64");
    DC_REQUIRE(v, "dc_hash_i64: null input");
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    DC_LAUNCH(k_hash_i64, dim3(grid), dim3(256), 0, stream, v, n,
                       (unsigned long long *)out);
    return check_launch("dc_hash_i64");
}

extern "C" int dc_morton_codes(const float *pos, int64_t ld, int64_t n, const float *lo_host,
                               const float *inv_extent_host, int64_t *codes, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(n >= 0, "dc_morton_codes: negative size");
    if (n == 0) return DC_OK;
    DC_REQUIRE(pos && lo_host && inv_extent_host && codes && ld >= 3, "dc_morton_codes: bad arguments");
    DC_LAUNCH(k_morton, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pos, ld, n,
                       lo_host[0], lo_host[1], lo_host[2], inv_extent_host[0], inv_extent_host[1],
                       inv_extent_host[2], codes);
    return check_launch("dc_morton_codes");
}
