// dc_optim.hip -- Adam over one flat fp32 parameter / gradient bucket (gfx950).
//
// The reference trains with torch.optim.Adam(lr=4e-4) at its defaults
// (/root/reference/train.py:20).  With parameters and gradients living in flat buckets
// (deformcontact_amd/dp.py) the whole update is ONE elementwise pass: 4 reads + 3 writes of
// 4 B per parameter, HBM/L2-bound, instead of torch's multi-tensor launch sequence.
// Same arithmetic as torch's single-tensor Adam (no amsgrad, no weight decay, no maximize):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step count t is a device scalar incremented by a 1-thread kernel ahead of the update, so
// the pair of launches is hipGraph-replayable (no host-side state).
#include "dc_common.h"

namespace dc {

__global__ void __launch_bounds__(256)
k_adam_flat(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
            float *__restrict__ v, int64_t n, float *step, float lr, float b1, float b2, float eps,
            int zero_grad) {
    // step[0] = number of completed updates; this one is update t = step[0] + 1.  The last block
    // to finish publishes t, after every block has read it: two-level self-resetting tickets
    // (step[2 + b % 32], then step[1]) so that no single address sees more than ~grid/32 atomics.
    const float t = step[0] + 1.0f;
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float step_size = lr / bc1, rbc2 = sqrtf(bc2);
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 + 4 <= n) {
        const float4 gg = *reinterpret_cast<const float4 *>(g + i4);
        float4 mm = *reinterpret_cast<const float4 *>(m + i4);
        float4 vv = *reinterpret_cast<const float4 *>(v + i4);
        float4 pp = *reinterpret_cast<const float4 *>(p + i4);
#define DC_ADAM(c)                                                   \
    mm.c = mm.c + (gg.c - mm.c) * (1.0f - b1);                       \
    vv.c = b2 * vv.c + (1.0f - b2) * gg.c * gg.c;                    \
    pp.c = pp.c - step_size * (mm.c / (sqrtf(vv.c) / rbc2 + eps));
        DC_ADAM(x) DC_ADAM(y) DC_ADAM(z) DC_ADAM(w)
#undef DC_ADAM
        *reinterpret_cast<float4 *>(m + i4) = mm;
        *reinterpret_cast<float4 *>(v + i4) = vv;
        *reinterpret_cast<float4 *>(p + i4) = pp;
        if (zero_grad) *reinterpret_cast<float4 *>(g + i4) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float gi = g[i];
            const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
            const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
            m[i] = mi, v[i] = vi;
            p[i] = p[i] - step_size * (mi / (sqrtf(vi) / rbc2 + eps));
            if (zero_grad) g[i] = 0.f;
        }
    }
    __syncthreads();
    // no fence: every wave of this block has consumed t (the barrier above) before the ticket is
    // taken, and the published count is only read by the NEXT launch (a device-scope fence costs
    // an L2 write-back per block on the multi-XCD part: measured 25 us instead of 6)
    if (threadIdx.x == 0) {
        unsigned *tk = reinterpret_cast<unsigned *>(step);
        const unsigned sub = blockIdx.x & 31u, nsub = gridDim.x < 32u ? gridDim.x : 32u;
        const unsigned members = (gridDim.x - sub + 31u) / 32u;       // blocks with b % 32 == sub
        if (atomicAdd(tk + 2 + sub, 1u) == members - 1) {
            tk[2 + sub] = 0u;
            if (atomicAdd(tk + 1, 1u) == nsub - 1) {
                tk[1] = 0u;
                step[0] = t;
            }
        }
    }
}

}  // namespace dc

extern "C" int dc_adam_flat(float *p, float *g, float *m, float *v, int64_t n, float *step,
                            float lr, float beta1, float beta2, float eps, int zero_grad,
                            dc_stream_t stream) {
    DC_REQUIRE(n >= 0, "dc_adam_flat: negative size");
    if (n == 0) return DC_OK;
    DC_REQUIRE(p && g && m && v && step, "dc_adam_flat: null pointer");
    DC_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "dc_adam_flat: buffers must be 16-byte aligned");
    const int64_t threads = (n + 3) / 4;
    DC_LAUNCH(dc::k_adam_flat, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, p, g, m, v, n, step, lr, beta1, beta2, eps, zero_grad);
    return dc::check_launch("dc_adam_flat");
}

// ---- small packing helpers of the narrow-layer (concatenated K segment) path --------------
namespace dc {

// slab[i, 0:F] = x[i, 0:F]; slab[i, width:wpad] = 0   (one launch instead of copy + fill)
__global__ void __launch_bounds__(256)
k_pack_input(const float *__restrict__ x, int64_t ldx, float *__restrict__ slab, int64_t lds,
             int64_t N, int F, int width, int wpad) {
    const int per_row = F + (wpad - width);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * per_row) return;
    const int64_t row = i / per_row;
    const int c = (int)(i % per_row);
    if (c < F)
        slab[row * lds + c] = x[row * ldx + c];
    else
        slab[row * lds + width + (c - F)] = 0.f;
}

struct PackWParams {
    const float *w[2 * DC_MAX_SEG];
    float *wcat;
    int64_t Fo;
    int nw, fi, wpad;
};

// wcat[o, j*fi + f] = w[j][o, f], zero in [nw*fi, wpad)
__global__ void __launch_bounds__(256)
k_pack_weights(PackWParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.Fo * p.wpad) return;
    const int64_t o = i / p.wpad;
    const int c = (int)(i % p.wpad), j = c / p.fi;
    p.wcat[i] = j < p.nw ? p.w[j][o * p.fi + (c - j * p.fi)] : 0.f;
}

}  // namespace dc

extern "C" int dc_tag_pack_input(const float *x, int64_t ldx, float *slab, int64_t ld_slab,
                                 int64_t N, int64_t F, int64_t width, int64_t wpad,
                                 dc_stream_t stream) {
    DC_REQUIRE(N >= 0 && F >= 1 && width >= F && wpad >= width && ld_slab >= wpad && ldx >= F,
               "dc_tag_pack_input: bad sizes");
    if (N == 0) return DC_OK;
    DC_REQUIRE(x && slab, "dc_tag_pack_input: null pointer");
    const int64_t total = N * (F + (wpad - width));
    DC_LAUNCH(dc::k_pack_input, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, slab, ld_slab, N, (int)F, (int)width, (int)wpad);
    return dc::check_launch("dc_tag_pack_input");
}

extern "C" int dc_tag_pack_weights(const float *const *ws, int nw, float *wcat, int64_t Fo,
                                   int64_t fi, int64_t wpad, dc_stream_t stream) {
    DC_REQUIRE(nw >= 1 && nw <= 2 * DC_MAX_SEG && Fo >= 1 && fi >= 1 && wpad >= nw * fi,
               "dc_tag_pack_weights: bad sizes");
    DC_REQUIRE(ws && wcat, "dc_tag_pack_weights: null pointer");
    dc::PackWParams p{};
    for (int j = 0; j < nw; ++j) {
        DC_REQUIRE(ws[j], "dc_tag_pack_weights: null weight %d", j);
        p.w[j] = ws[j];
    }
    p.wcat = wcat, p.Fo = Fo, p.nw = nw, p.fi = (int)fi, p.wpad = (int)wpad;
    const int64_t total = Fo * wpad;
    DC_LAUNCH(dc::k_pack_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return dc::check_launch("dc_tag_pack_weights");
}
