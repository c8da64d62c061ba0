// dc_loss.hip -- the reference's two training losses in one pass over the nodes (gfx950).
//
//   l1  = mean_{i,c} |pred[i,c] - target[i,c]|                          (train.py:51  nn.L1Loss)
//   gcl = (1/E) sum_e || (t[dst_e] - t[src_e]) - (p[dst_e] - p[src_e]) ||_2   (models/losses.py:7-19)
//
// and, in the same launch, their gradients w.r.t. pred (so the backward is two scalings).  The
// reference runs 4 gathers of 12-byte rows, a norm, two reductions and the L1 as ~10 ATen kernels
// forward and as many backward.  Here one thread owns one node: the consistency terms of its
// IN-edges (sorted adjacency by destination) give the loss sum - every edge is the in-edge of
// exactly one node - and the gradient of every edge endpoint comes from the node's in-edges
// (it is the destination) and out-edges (adjacency by source: it is the source).  No float atomics:
// per-block partial sums are combined in block order by a second one-block kernel (deterministic).
#include "dc_common.h"

namespace dc {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float *p, int64_t ld, int64_t i) {
    return {p[i * ld], p[i * ld + 1], p[i * ld + 2]};
}

__global__ void __launch_bounds__(256)
k_loss_nodes(const int32_t *__restrict__ ptr_f, const int32_t *__restrict__ oth_f,
             const int32_t *__restrict__ ptr_b, const int32_t *__restrict__ oth_b,
             const float *__restrict__ pred, int64_t ldp, const float *__restrict__ tgt, int64_t ldt,
             int64_t N, float inv_3n, float inv_e, float *__restrict__ g_l1,
             float *__restrict__ g_gcl, float *__restrict__ partial) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s_l1 = 0.f, s_gc = 0.f;
    if (i < N) {
        const V3 p = ld3(pred, ldp, i), t = ld3(tgt, ldt, i);
        const float dx = p.x - t.x, dy = p.y - t.y, dz = p.z - t.z;
        s_l1 = fabsf(dx) + fabsf(dy) + fabsf(dz);
        auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
        g_l1[i * 3] = sgn(dx) * inv_3n;
        g_l1[i * 3 + 1] = sgn(dy) * inv_3n;
        g_l1[i * 3 + 2] = sgn(dz) * inv_3n;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        // in-edges src -> i:  d = (t_i - t_src) - (p_i - p_src);  d loss / d p_i = -d / |d|
        for (int q = ptr_f[i]; q < ptr_f[i + 1]; ++q) {
            const int64_t s = oth_f[q];
            const V3 ps = ld3(pred, ldp, s), ts = ld3(tgt, ldt, s);
            const float ex = (t.x - ts.x) - (p.x - ps.x), ey = (t.y - ts.y) - (p.y - ps.y),
                        ez = (t.z - ts.z) - (p.z - ps.z);
            const float nrm = sqrtf(ex * ex + ey * ey + ez * ez);
            s_gc += nrm;
            const float r = nrm > 0.f ? 1.f / nrm : 0.f;      // torch: zero gradient at a zero norm
            gx -= ex * r, gy -= ey * r, gz -= ez * r;
        }
        // out-edges i -> dst:  d = (t_dst - t_i) - (p_dst - p_i);  d loss / d p_i = +d / |d|
        for (int q = ptr_b[i]; q < ptr_b[i + 1]; ++q) {
            const int64_t d = oth_b[q];
            const V3 pd = ld3(pred, ldp, d), td = ld3(tgt, ldt, d);
            const float ex = (td.x - t.x) - (pd.x - p.x), ey = (td.y - t.y) - (pd.y - p.y),
                        ez = (td.z - t.z) - (pd.z - p.z);
            const float nrm = sqrtf(ex * ex + ey * ey + ez * ez);
            const float r = nrm > 0.f ? 1.f / nrm : 0.f;
            gx += ex * r, gy += ey * r, gz += ez * r;
        }
        g_gcl[i * 3] = gx * inv_e;
        g_gcl[i * 3 + 1] = gy * inv_e;
        g_gcl[i * 3 + 2] = gz * inv_e;
    }
    __shared__ float red[2][4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s_l1 += __shfl_xor(s_l1, o);
        s_gc += __shfl_xor(s_gc, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s_l1;
        red[1][threadIdx.x >> 6] = s_gc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256)
k_loss_final(const float *partial, int64_t nblk, float inv_3n, float inv_e, float *out) {
    // fixed order: thread t sums partials t, t+256, ...; then a fixed tree
    double a = 0.0, b = 0.0;
    for (int64_t j = threadIdx.x; j < nblk; j += 256) {
        a += partial[2 * j];
        b += partial[2 * j + 1];
    }
    __shared__ double ra[256], rb[256];
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            ra[threadIdx.x] += ra[threadIdx.x + s];
            rb[threadIdx.x] += rb[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(ra[0] * (double)inv_3n);
        out[1] = (float)(rb[0] * (double)inv_e);
    }
}

}  // namespace dc

using namespace dc;

extern "C" int64_t dc_contact_loss_workspace_bytes(int64_t N) {
    if (N < 0) return DC_EINVAL;
    return 8 * ((N + 255) / 256) + 16;
}

extern "C" int dc_contact_loss(const int32_t *ptr_f, const int32_t *other_f, const int32_t *ptr_b,
                               const int32_t *other_b, const float *pred, int64_t ld_pred,
                               const float *target, int64_t ld_target, int64_t N, int64_t E,
                               float *grad_l1, float *grad_gcl, float *losses, void *workspace,
                               int64_t workspace_bytes, dc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DC_REQUIRE(N > 0 && E >= 0, "dc_contact_loss: needs N > 0, E >= 0 (N=%lld E=%lld)", (long long)N,
               (long long)E);
    DC_REQUIRE(ptr_f && ptr_b && pred && target && grad_l1 && grad_gcl && losses && workspace,
               "dc_contact_loss: null pointer");
    DC_REQUIRE(E == 0 || (other_f && other_b), "dc_contact_loss: null edge arrays");
    DC_REQUIRE(ld_pred >= 3 && ld_target >= 3, "dc_contact_loss: positions are [N, 3]");
    DC_REQUIRE(workspace_bytes >= dc_contact_loss_workspace_bytes(N), "dc_contact_loss: workspace too small");
    const int64_t nblk = (N + 255) / 256;
    const float inv_3n = 1.0f / (3.0f * (float)N);
    // mean over edges as the reference divides: sum / E (E = 0: the reference divides by zero -> nan/inf)
    const float inv_e = 1.0f / (float)E;
    DC_LAUNCH(k_loss_nodes, dim3((unsigned)nblk), dim3(256), 0, stream, ptr_f, other_f, ptr_b,
                       other_b, pred, ld_pred, target, ld_target, N, inv_3n, inv_e, grad_l1, grad_gcl,
                       (float *)workspace);
    DC_LAUNCH(k_loss_final, dim3(1), dim3(256), 0, stream, (const float *)workspace, nblk, inv_3n,
                       inv_e, losses);
    return check_launch("dc_contact_loss");
}
