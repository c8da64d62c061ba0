// astream.hip -- how fast can a 128-row tile walk its rows?  (experiment behind k_fwd_h2w's staging shape)
// One 512-thread workgroup per 128 rows of a row-major fp32 matrix [N, K] (leading dimension ld); per step the
// workgroup reads CH bytes of each row (CH = 128, 256, 512), D steps in flight (register ring), sums everything
// into one float per thread so that nothing is optimised away.  hipcc -O3 --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CH, int D>
__global__ void __launch_bounds__(512) k_astream(const float *__restrict__ x, int64_t ld, int K, float *out) {
    constexpr int TPR = CH / 16;              // threads per row
    constexpr int RPP = 512 / TPR;            // rows per pass
    constexpr int NL = 128 / RPP;             // loads per thread and step
    const int t = threadIdx.x, c = t % TPR, r = t / TPR;
    const float *base = x + ((int64_t)blockIdx.x * 128 + r) * ld + 4 * c;
    const int nstep = K * 4 / CH;
    f32x4 ring[D][NL];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int j = 0; j < NL; ++j)
            ring[d][j] = *reinterpret_cast<const f32x4 *>(base + (int64_t)j * RPP * ld + (d < nstep ? d : nstep - 1) * (CH / 4));
    int s = 0;
    for (; s + D < nstep; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                acc += ring[d][j];
                ring[d][j] = *reinterpret_cast<const f32x4 *>(base + (int64_t)j * RPP * ld + (s + D + d < nstep ? s + D + d : nstep - 1) * (CH / 4));
            }
            __syncthreads();                  // the GEMM has a barrier per stage
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int j = 0; j < NL; ++j) acc += ring[d][j];
    out[(int64_t)blockIdx.x * 512 + t] = acc[0] + acc[1] + acc[2] + acc[3];
}

extern "C" int astream_run(int ch, int d, const float *x, int64_t ld, int64_t n, int k, float *out, void *stream) {
    const dim3 g((unsigned)(n / 128)), b(512);
    hipStream_t s = (hipStream_t)stream;
#define GO(CH, D) if (ch == CH && d == D) { hipLaunchKernelGGL((k_astream<CH, D>), g, b, 0, s, x, ld, k, out); return 0; }
    GO(128, 2) GO(128, 3) GO(128, 4) GO(128, 6) GO(128, 8)
    GO(256, 1) GO(256, 2) GO(256, 3) GO(256, 4)
    GO(512, 1) GO(512, 2) GO(512, 3)
    GO(1024, 1) GO(1024, 2)
#undef GO
    return 1;
}
