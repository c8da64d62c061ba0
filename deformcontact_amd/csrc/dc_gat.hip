// dc_gat.hip -- per-edge pieces of GATConv (heads = 1) on the sorted adjacency.
//
// Replaces, for the optional "GATConv" backbone (/root/reference/models/model.py:39),
// PyG 2.5.2 gat_conv.py edge_update (alpha_j + alpha_i -> leaky_relu -> softmax over
// the incoming edges of i, utils/_softmax.py) and the backward of that chain.
// All arrays are in destination-sorted order (dc_csr_build key_row=1,
// self_loops=1); the aggregation itself is dc_spmm_f32 with w = alpha.
#include "dc_common.h"

namespace dc {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

// kSub lanes per destination segment (mean in-degree of the reference's meshes + the self loop is
// 7): lane `sub` of a group walks edges beg + sub, beg + sub + kSub, ...  Segments are contiguous in
// p, so the alpha / galpha / ge accesses of a wave are coalesced; max / sum / dot are reduced
// inside the group with xor shuffles.  Group reductions of floats are order-fixed (deterministic).
constexpr int kSub = 8;

__device__ __forceinline__ float sub_max(float v) {
#pragma unroll
    for (int d = kSub / 2; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
    for (int d = kSub / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

__global__ void __launch_bounds__(256)
k_gat_softmax_fwd(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
                  const float *__restrict__ a_src, const float *__restrict__ a_dst, float slope,
                  float *__restrict__ alpha, int64_t N) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kSub;
    const int sub = threadIdx.x % kSub;
    const bool live = i < N;
    const int beg = live ? ptr[i] : 0, end = live ? ptr[i + 1] : 0;
    const float ad = live ? a_dst[i] : 0.f;
    float m = -INFINITY;
    for (int p = beg + sub; p < end; p += kSub) m = fmaxf(m, lrelu(a_src[other[p]] + ad, slope));
    m = sub_max(m);
    float s = 0.f;
    for (int p = beg + sub; p < end; p += kSub) {
        const float ex = expf(lrelu(a_src[other[p]] + ad, slope) - m);
        alpha[p] = ex;
        s += ex;
    }
    const float denom = sub_sum(s) + 1e-16f;
    for (int p = beg + sub; p < end; p += kSub) alpha[p] = alpha[p] / denom;   // own elements only
}

__global__ void __launch_bounds__(256)
k_gat_softmax_bwd(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
                  const float *__restrict__ a_src, const float *__restrict__ a_dst, float slope,
                  const float *__restrict__ alpha, const float *__restrict__ galpha,
                  float *__restrict__ ge, float *__restrict__ g_a_dst, int64_t N) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kSub;
    const int sub = threadIdx.x % kSub;
    const bool live = i < N;
    const int beg = live ? ptr[i] : 0, end = live ? ptr[i + 1] : 0;
    const float ad = live ? a_dst[i] : 0.f;
    float dot = 0.f;
    for (int p = beg + sub; p < end; p += kSub) dot += alpha[p] * galpha[p];
    dot = sub_sum(dot);
    float acc = 0.f;
    for (int p = beg + sub; p < end; p += kSub) {
        const float s = a_src[other[p]] + ad;
        const float g = alpha[p] * (galpha[p] - dot) * (s > 0.f ? 1.0f : slope);
        ge[p] = g;
        acc += g;
    }
    acc = sub_sum(acc);
    if (live && sub == 0) g_a_dst[i] = acc;
}

// d[p] = <g[i,:], h[other[p],:]> ; one wave per destination row, the row of g held in registers
// (up to kGRegs * 64 * VEC columns; wider rows re-read it), VEC = 4: 16-byte loads
template <int VEC>
__global__ void __launch_bounds__(256)
k_sddmm(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
        const float *__restrict__ g, int64_t ldg, const float *__restrict__ h, int64_t ldh,
        float *__restrict__ d, int64_t N, int F) {
    constexpr int kGRegs = 2;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = __builtin_amdgcn_readfirstlane((int)(lb * 4u + (threadIdx.x >> 6)));
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    const int beg = ptr[row], end = ptr[row + 1];
    float gr[kGRegs][VEC];
#pragma unroll
    for (int j = 0; j < kGRegs; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = (j * kWave + lane) * VEC + v;
            gr[j][v] = c < F ? g[row * ldg + c] : 0.f;
        }
    for (int p = beg; p < end; ++p) {
        const int64_t s = other[p];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < kGRegs; ++j) {
            const int c0 = (j * kWave + lane) * VEC;
            if (VEC == 4) {
                if (c0 < F) {
                    const float4 hv = *reinterpret_cast<const float4 *>(h + s * ldh + c0);
                    acc += gr[j][0] * hv.x + gr[j][1] * hv.y + gr[j][2] * hv.z + gr[j][3] * hv.w;
                }
            } else if (c0 < F) {
                acc += gr[j][0] * h[s * ldh + c0];
            }
        }
        for (int c = kGRegs * kWave * VEC + lane; c < F; c += kWave) acc += g[row * ldg + c] * h[s * ldh + c];
        acc = wave_sum(acc);
        if (lane == 0) d[p] = acc;
    }
}

__global__ void __launch_bounds__(256)
k_segment_sum(const int32_t *__restrict__ ptr, const int32_t *__restrict__ map,
              const float *__restrict__ v, float *__restrict__ out, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float acc = 0.f;
    for (int p = ptr[i], end = ptr[i + 1]; p < end; ++p) acc += v[map ? map[p] : p];
    out[i] = acc;
}

__global__ void __launch_bounds__(256)
k_gather(const float *__restrict__ v, const int32_t *__restrict__ idx, float *__restrict__ out,
         const int32_t *n_ptr, int64_t cap) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < cap && p < *n_ptr) out[p] = v[idx[p]];
}

__global__ void __launch_bounds__(256)
k_compose(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t *__restrict__ out,
          const int32_t *n_ptr, int64_t cap) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < cap && p < *n_ptr) out[p] = a[b[p]];
}

}  // namespace dc

using namespace dc;

extern "C" int dc_gat_edge_softmax_fwd(const int32_t *ptr, const int32_t *other,
                                       const float *a_src, const float *a_dst, float slope,
                                       float *alpha, int64_t N, dc_stream_t stream) {
    DC_REQUIRE(N >= 0, "dc_gat_edge_softmax_fwd: negative N");
    if (N == 0) return DC_OK;
    DC_REQUIRE(ptr && other && a_src && a_dst && alpha, "dc_gat_edge_softmax_fwd: null pointer");
    DC_LAUNCH(k_gat_softmax_fwd, dim3((unsigned)((N * kSub + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ptr, other, a_src, a_dst, slope, alpha, N);
    return check_launch("dc_gat_edge_softmax_fwd");
}

extern "C" int dc_gat_edge_softmax_bwd(const int32_t *ptr, const int32_t *other,
                                       const float *a_src, const float *a_dst, float slope,
                                       const float *alpha, const float *galpha, float *ge,
                                       float *g_a_dst, int64_t N, dc_stream_t stream) {
    DC_REQUIRE(N >= 0, "dc_gat_edge_softmax_bwd: negative N");
    if (N == 0) return DC_OK;
    DC_REQUIRE(ptr && other && a_src && a_dst && alpha && galpha && ge && g_a_dst,
               "dc_gat_edge_softmax_bwd: null pointer");
    DC_LAUNCH(k_gat_softmax_bwd, dim3((unsigned)((N * kSub + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ptr, other, a_src, a_dst, slope, alpha, galpha, ge, g_a_dst, N);
    return check_launch("dc_gat_edge_softmax_bwd");
}

extern "C" int dc_sddmm_f32(const int32_t *ptr, const int32_t *other, const float *g, int64_t ldg,
                            const float *h, int64_t ldh, float *d, int64_t N, int64_t F,
                            dc_stream_t stream) {
    DC_REQUIRE(N >= 0 && F >= 0 && F < (1 << 24), "dc_sddmm_f32: bad size");
    if (N == 0) return DC_OK;
    DC_REQUIRE(ptr && other && g && h && d, "dc_sddmm_f32: null pointer");
    DC_REQUIRE(ldg >= F && ldh >= F, "dc_sddmm_f32: leading dimension smaller than F");
    const bool vec4 = F % 4 == 0 && ldg % 4 == 0 && ldh % 4 == 0 && (((uintptr_t)g | (uintptr_t)h) & 15) == 0;
    if (vec4)
        DC_LAUNCH((k_sddmm<4>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           ptr, other, g, ldg, h, ldh, d, N, (int)F);
    else
        DC_LAUNCH((k_sddmm<1>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           ptr, other, g, ldg, h, ldh, d, N, (int)F);
    return check_launch("dc_sddmm_f32");
}

extern "C" int dc_segment_sum_f32(const int32_t *ptr, const int32_t *map, const float *v,
                                  float *out, int64_t N, dc_stream_t stream) {
    DC_REQUIRE(N >= 0, "dc_segment_sum_f32: negative N");
    if (N == 0) return DC_OK;
    DC_REQUIRE(ptr && v && out, "dc_segment_sum_f32: null pointer");
    DC_LAUNCH(k_segment_sum, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, ptr,
                       map, v, out, N);
    return check_launch("dc_segment_sum_f32");
}

extern "C" int dc_gather_f32(const float *v, const int32_t *idx, float *out,
                             const int32_t *count_ptr, int64_t cap, dc_stream_t stream) {
    DC_REQUIRE(cap >= 0, "dc_gather_f32: negative size");
    if (cap == 0) return DC_OK;
    DC_REQUIRE(v && idx && out && count_ptr, "dc_gather_f32: null pointer");
    DC_LAUNCH(k_gather, dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, idx,
                       out, count_ptr, cap);
    return check_launch("dc_gather_f32");
}

extern "C" int dc_compose_perm(const int32_t *a, const int32_t *b, int32_t *out,
                               const int32_t *count_ptr, int64_t cap, dc_stream_t stream) {
    DC_REQUIRE(cap >= 0, "dc_compose_perm: negative size");
    if (cap == 0) return DC_OK;
    DC_REQUIRE(a && b && out && count_ptr, "dc_compose_perm: null pointer");
    DC_LAUNCH(k_compose, dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b,
                       out, count_ptr, cap);
    return check_launch("dc_compose_perm");
}
