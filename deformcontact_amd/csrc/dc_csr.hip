// dc_csr.hip -- edge_index -> stably sorted adjacency (+ gcn_norm weights), gfx950.
//
// Replaces the implicit destination ordering of PyG's scatter_add_ and the
// gcn_norm call TAGConv/GCNConv.forward repeats on every invocation
// (/root/reference/models/model.py:71,77).  Integer work, HBM/L2-bound, built
// once per batched edge_index and cached by the host.
//
// Pipeline (all on one stream, no host sync):
//   k_count      histogram of the key endpoint (int atomics; order-free)
//   k_scan_*     exclusive scan -> ptr[N+1]               (3 small kernels)
//   k_fill       bucket every edge id into its group in ARBITRARY order
//   k_rank_emit  rank each id inside its group (count of smaller ids) and
//                write perm/other/w at ptr[key]+rank -> the order is the
//                stable sort no matter how the atomics in k_fill interleaved.
#include "dc_common.h"

namespace dc {

constexpr int kScanTile = 1024;   // elements per scan block (256 threads x 4)

struct EdgeView {
    const int64_t *key;
    const int64_t *oth;
    int64_t E, N;
    int self_loops;
};

__global__ void __launch_bounds__(256)
k_fill_i32(int32_t *p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void __launch_bounds__(256)
k_count(EdgeView ev, int32_t *cnt, int32_t *status) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ev.E) return;
    const int64_t k = ev.key[e], o = ev.oth[e];
    if (k < 0 || k >= ev.N || o < 0 || o >= ev.N) {
        atomicOr(status, 1);
        return;
    }
    if (ev.self_loops && k == o) return;
    atomicAdd(&cnt[k], 1);
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

// inclusive scan across a 256-thread block; returns this thread's inclusive
// value, *total = block sum
__device__ __forceinline__ int block_incl_scan256(int v, int *total) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wid) off += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return inc + off;
}

__global__ void __launch_bounds__(256)
k_scan_reduce(const int32_t *cnt, int64_t N, int32_t *tile_sums) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < N) s += cnt[base + j];
    int total;
    block_incl_scan256(s, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
k_scan_tiles(int32_t *tile_sums, int64_t ntiles) {   // single block, in place -> exclusive
    int carry = 0;
    for (int64_t base = 0; base < ntiles; base += 256) {
        const int64_t i = base + threadIdx.x;
        const int v = i < ntiles ? tile_sums[i] : 0;
        int total;
        const int inc = block_incl_scan256(v, &total);
        if (i < ntiles) tile_sums[i] = carry + inc - v;
        carry += total;
    }
    if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

__global__ void __launch_bounds__(256)
k_scan_apply(int32_t *cnt /* in: counts, out: cursor = ptr */, int64_t N,
             const int32_t *tile_sums, int64_t ntiles, int32_t *ptr) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = base + j < N ? cnt[base + j] : 0;
        s += v[j];
    }
    int total;
    int run = block_incl_scan256(s, &total) - s + tile_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < N) {
            ptr[base + j] = run;
            cnt[base + j] = run;   // becomes the fill cursor
        }
        run += v[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ptr[N] = tile_sums[ntiles];
}

__global__ void __launch_bounds__(256)
k_fill(EdgeView ev, int32_t *cursor, int32_t *tmp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ev.E) {
        const int64_t k = ev.key[t], o = ev.oth[t];
        if (k < 0 || k >= ev.N || o < 0 || o >= ev.N) return;
        if (ev.self_loops && k == o) return;
        tmp[atomicAdd(&cursor[k], 1)] = (int32_t)t;
    } else if (ev.self_loops && t < ev.E + ev.N) {
        const int64_t k = t - ev.E;
        tmp[atomicAdd(&cursor[k], 1)] = (int32_t)t;
    }
}

__device__ __forceinline__ float inv_sqrt_deg(const int32_t *deg_ptr, int64_t v) {
    const int d = deg_ptr[v + 1] - deg_ptr[v];
    return d > 0 ? 1.0f / sqrtf((float)d) : 0.0f;   // deg.pow(-0.5), inf -> 0
}

__global__ void __launch_bounds__(256)
k_rank_emit(EdgeView ev, const int32_t *ptr, const int32_t *tmp, int32_t *other,
            int32_t *perm, const int32_t *deg_ptr, float *w, int key_is_dst) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ptr[ev.N]) return;
    const int32_t eid = tmp[p];
    int64_t k, o;
    if (eid < ev.E) {
        k = ev.key[eid];
        o = ev.oth[eid];
    } else {
        k = o = eid - ev.E;
    }
    const int32_t beg = ptr[k], end = ptr[k + 1];
    int rank = 0;
    for (int32_t q = beg; q < end; ++q) rank += tmp[q] < eid;
    const int32_t out = beg + rank;
    perm[out] = eid;
    other[out] = (int32_t)o;
    if (w) {
        // gcn_norm: dis[row] * 1 * dis[col]  (row = source, col = destination)
        const int64_t src = key_is_dst ? o : k, dst = key_is_dst ? k : o;
        w[out] = inv_sqrt_deg(deg_ptr, src) * 1.0f * inv_sqrt_deg(deg_ptr, dst);
    }
}

__global__ void __launch_bounds__(256)
k_invert_perm(const int32_t *perm, const int32_t *n_ptr, int32_t *pos_of, int64_t max_edges) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < max_edges && p < *n_ptr) pos_of[perm[p]] = (int32_t)p;
}

static inline int64_t align16(int64_t b) { return (b + 15) & ~int64_t(15); }

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return DC_EINVAL;
    const int64_t ntiles = (N + kScanTile - 1) / kScanTile;
    return align16(4 * (N + 1)) + align16(4 * (E + N + 1)) + align16(4 * (ntiles + 2));
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

    const int64_t ntiles = (N + kScanTile - 1) / kScanTile;
    char *ws = (char *)workspace;
    int32_t *cnt = (int32_t *)ws;
    ws += align16(4 * (N + 1));
    int32_t *tmp = (int32_t *)ws;
    ws += align16(4 * (E + N + 1));
    int32_t *tile_sums = (int32_t *)ws;

    EdgeView ev{edge_index + (key_row ? E : 0), edge_index + (key_row ? 0 : E), E, N, self_loops};
    const int64_t slots = E + (self_loops ? N : 0);

    if (N == 0) {
        hipMemsetAsync(ptr, 0, sizeof(int32_t), stream);
        return check_launch("dc_csr_build(memset)");
    }
    if (self_loops)
        hipLaunchKernelGGL(k_fill_i32, dim3((N + 255) / 256), dim3(256), 0, stream, cnt, N, 1);
    else
        hipMemsetAsync(cnt, 0, sizeof(int32_t) * N, stream);
    if (E > 0)
        hipLaunchKernelGGL(k_count, dim3((E + 255) / 256), dim3(256), 0, stream, ev, cnt, status);
    hipLaunchKernelGGL(k_scan_reduce, dim3(ntiles), dim3(256), 0, stream, cnt, N, tile_sums);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, stream, tile_sums, ntiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(256), 0, stream, cnt, N, tile_sums, ntiles,
                       ptr);
    if (slots > 0) {
        hipLaunchKernelGGL(k_fill, dim3((slots + 255) / 256), dim3(256), 0, stream, ev, cnt, tmp);
        hipLaunchKernelGGL(k_rank_emit, dim3((slots + 255) / 256), dim3(256), 0, stream, ev, ptr,
                           tmp, other, perm, deg_ptr ? deg_ptr : ptr, w, key_row);
    }
    return check_launch("dc_csr_build");
}

extern "C" int dc_invert_perm(const int32_t *perm, const int32_t *ptr_last, int32_t *pos_of,
                              int64_t max_edges, dc_stream_t stream) {
    DC_REQUIRE(max_edges >= 0, "dc_invert_perm: negative size");
    if (max_edges == 0) return DC_OK;
    DC_REQUIRE(perm && ptr_last && pos_of, "dc_invert_perm: null pointer");
    hipLaunchKernelGGL(k_invert_perm, dim3((max_edges + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, perm, ptr_last, pos_of, max_edges);
    return check_launch("dc_invert_perm");
}
