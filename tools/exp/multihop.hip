// tools/exp/multihop.hip -- EXPERIMENT retired from the product library in r03 (was dc_multihop.hip, off by default since r01):
// K chained hops with the meshes resident in LDS.  Bit-identical to K single hops; slower than them on MI355X at the
// everyday-deform shape (F = 256: 52.8 vs 49.1 us for three hops, F = 21: 40.0 vs 30.7 us; profiles/r02/k_kbench_multihop.txt).
// Kept for the record; not built by deformcontact_amd/build.py.
// dc_multihop.hip -- K chained hops of a block-diagonal batch in ONE launch, features in LDS.
//
// A PyG batch is a block-diagonal union of small meshes (everyday-deform: 1,024 / 762 nodes
// each); message passing never leaves a mesh.  TAGConv needs x_k = A_hat x_{k-1} for k = 1..K
// (/root/reference/models/model.py:71,77 -> PyG tag_conv.py: K = 3 dependent propagate calls),
// and its backward the chain g_{k-1} = G_{k-1} + A_hat^T g_k.  Run hop by hop, every x_k goes
// out to HBM/L2 and comes back through K separate gathers.  Here one workgroup owns one
// (node segment, 16-column slice): the slice of the source block (<= 1024 nodes x 64 B = 64 KiB)
// AND the segment's adjacency (<= 6656 {neighbour, weight} records = 52 KiB) are staged in LDS
// once; each hop gathers neighbour rows from LDS (ds_read_b128) into registers, then overwrites
// the LDS slice and writes the produced block to global memory exactly once.  HBM traffic per layer: read 1 block + write K blocks, instead of K x (gather + write).
//
// Arithmetic and order are those of dc_spmm_f32 (separately rounded multiply and add, stable
// edge order, addend first), so results are bit-identical to K single hops.
#include "../../deformcontact_amd/csrc/dc_common.h"

#pragma clang fp contract(off)

namespace dc {

constexpr int kMhCols = 16;          // columns per slice (one 64-byte LDS row per node)
constexpr int kMhNodes = 1024;       // max nodes per segment
constexpr int kMhEdges = 6656;       // max edges per segment (8 B each in LDS)
constexpr int kMhTasks = 4;          // (node, 4-column quad) tasks per thread: 1024*4 / 1024 threads

struct MultihopParams {
    const int32_t *ptr, *other;
    const float *w;
    const int32_t *seg_ptr;      // [nseg+1] node ranges; no edge leaves a range
    float *slab;                 // [N, ld]: K+1 column blocks of width F
    int64_t ld;
    int F, K, nslices;
    int src0, dir, accumulate;   // step s: src block = src0 + s*dir, dst block = src + dir
    int vec;                     // 1: every (block, slice) start is 16-byte aligned, F % 4 == 0
};

struct EdgeRec {
    int j;       // neighbour, local to the segment
    float w;
};

__device__ __forceinline__ float4 ld_quad(const float *p, int nvalid, bool vec) {
    if (vec && nvalid >= 4) return *reinterpret_cast<const float4 *>(p);
    return make_float4(nvalid > 0 ? p[0] : 0.f, nvalid > 1 ? p[1] : 0.f, nvalid > 2 ? p[2] : 0.f,
                       nvalid > 3 ? p[3] : 0.f);
}

__device__ __forceinline__ void st_quad(float *p, const float4 &v, int nvalid, bool vec) {
    if (vec && nvalid >= 4) {
        *reinterpret_cast<float4 *>(p) = v;
        return;
    }
    if (nvalid > 0) p[0] = v.x;
    if (nvalid > 1) p[1] = v.y;
    if (nvalid > 2) p[2] = v.z;
    if (nvalid > 3) p[3] = v.w;
}

// LDS: feat [kMhNodes][16] fp32 (64 KiB) | edges [kMhEdges] {j, w} (52 KiB) | lptr [kMhNodes+1]
// One feature buffer is enough: a hop is computed into registers (4 float4 per thread), then,
// behind a barrier, written back over the buffer and out to global memory.
__global__ void __launch_bounds__(1024)
k_multihop(MultihopParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *feat = reinterpret_cast<float *>(smem);
    EdgeRec *edges = reinterpret_cast<EdgeRec *>(smem + kMhNodes * kMhCols * 4);
    int *lptr = reinterpret_cast<int *>(smem + kMhNodes * kMhCols * 4 + kMhEdges * 8);

    const int seg = blockIdx.x / p.nslices, slice = blockIdx.x % p.nslices;
    const int n0 = p.seg_ptr[seg], nn = p.seg_ptr[seg + 1] - n0;
    const int c0 = slice * kMhCols, cw = min(kMhCols, p.F - c0);
    const int e0 = p.ptr[n0], ne = p.ptr[n0 + nn] - e0;
    const bool vec = p.vec != 0;

    // staging: fixed trip counts, all global loads of a loop issued before the first is used
    // (a strided "for i < n" loop would pay one memory latency per iteration)
    {
        int lp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = threadIdx.x + u * 1024;
            lp[u] = (i <= nn) ? p.ptr[n0 + i] : 0;
        }
        constexpr int EU = (kMhEdges + 1023) / 1024;
        int oj[EU];
        float ow[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int i = threadIdx.x + u * 1024;
            oj[u] = (i < ne) ? p.other[e0 + i] : 0;
            ow[u] = (i < ne && p.w) ? p.w[e0 + i] : 1.0f;
        }
        const float *src = p.slab + (int64_t)n0 * p.ld + (int64_t)p.src0 * p.F + c0;
        float4 fv[kMhTasks];
#pragma unroll
        for (int u = 0; u < kMhTasks; ++u) {
            const int t = threadIdx.x + u * 1024;
            const int node = t >> 2, q = t & 3;
            fv[u] = (t < nn * 4) ? ld_quad(src + (int64_t)node * p.ld + 4 * q, cw - 4 * q, vec)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = threadIdx.x + u * 1024;
            if (i <= nn) lptr[i] = lp[u] - e0;
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int i = threadIdx.x + u * 1024;
            if (i < ne) edges[i] = EdgeRec{oj[u] - n0, ow[u]};
        }
#pragma unroll
        for (int u = 0; u < kMhTasks; ++u) {
            const int t = threadIdx.x + u * 1024;
            if (t < nn * 4) *reinterpret_cast<float4 *>(feat + (t >> 2) * kMhCols + 4 * (t & 3)) = fv[u];
        }
    }
    const int ntask = nn * 4;
    __syncthreads();
    for (int s = 0; s < p.K; ++s) {
        const int dstb = p.src0 + (s + 1) * p.dir;
        float *dst = p.slab + (int64_t)n0 * p.ld + (int64_t)dstb * p.F + c0;
        float4 acc[kMhTasks];
#pragma unroll
        for (int u = 0; u < kMhTasks; ++u) {
            const int t = threadIdx.x + u * 1024;
            acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < ntask) {
                const int node = t >> 2, q = t & 3;
                if (p.accumulate) acc[u] = ld_quad(dst + (int64_t)node * p.ld + 4 * q, cw - 4 * q, vec);
                const int beg = lptr[node], end = lptr[node + 1];
                for (int e = beg; e < end; ++e) {
                    const EdgeRec r = edges[e];
                    const float4 v = *reinterpret_cast<const float4 *>(feat + r.j * kMhCols + 4 * q);
                    const float mx = r.w * v.x, my = r.w * v.y, mz = r.w * v.z, mw = r.w * v.w;
                    acc[u].x = acc[u].x + mx;
                    acc[u].y = acc[u].y + my;
                    acc[u].z = acc[u].z + mz;
                    acc[u].w = acc[u].w + mw;
                }
            }
        }
        __syncthreads();                      // every gather of this hop is done
#pragma unroll
        for (int u = 0; u < kMhTasks; ++u) {
            const int t = threadIdx.x + u * 1024;
            if (t < ntask) {
                const int node = t >> 2, q = t & 3;
                *reinterpret_cast<float4 *>(feat + node * kMhCols + 4 * q) = acc[u];
                st_quad(dst + (int64_t)node * p.ld + 4 * q, acc[u], cw - 4 * q, vec);
            }
        }
        __syncthreads();
    }
}

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_multihop_max_segment_nodes(void) { return kMhNodes; }
extern "C" int64_t dc_multihop_max_segment_edges(void) { return kMhEdges; }

extern "C" int dc_multihop_f32(const int32_t *ptr, const int32_t *other, const float *w,
                               const int32_t *seg_ptr, int64_t nseg, float *slab, int64_t ld,
                               int64_t F, int K, int src_block, int dir, int accumulate,
                               dc_stream_t stream) {
    DC_REQUIRE(nseg >= 0 && F >= 1 && K >= 0 && ld >= F, "dc_multihop_f32: bad sizes");
    if (nseg == 0 || K == 0) return DC_OK;
    DC_REQUIRE(ptr && other && seg_ptr && slab, "dc_multihop_f32: null pointer");
    DC_REQUIRE(dir == 1 || dir == -1, "dc_multihop_f32: dir must be +1 or -1");
    DC_REQUIRE(src_block >= 0 && src_block + K * dir >= 0 && (int64_t)(src_block + 1) * F <= ld &&
                   (int64_t)(src_block + K * dir + 1) * F <= ld,
               "dc_multihop_f32: column blocks outside the slab");
    MultihopParams p{};
    p.ptr = ptr, p.other = other, p.w = w, p.seg_ptr = seg_ptr, p.slab = slab, p.ld = ld;
    p.F = (int)F, p.K = K, p.src0 = src_block, p.dir = dir, p.accumulate = accumulate;
    p.nslices = (int)((F + kMhCols - 1) / kMhCols);
    p.vec = (((uintptr_t)slab & 15) == 0 && ld % 4 == 0 && F % 4 == 0) ? 1 : 0;
    const size_t bytes = (size_t)kMhNodes * kMhCols * 4 + (size_t)kMhEdges * 8 + (kMhNodes + 1) * 4 + 12;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_multihop),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_multihop, dim3((unsigned)(nseg * p.nslices)), dim3(1024), bytes,
                       (hipStream_t)stream, p);
    return check_launch("dc_multihop_f32");
}
