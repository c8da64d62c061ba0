// tools/mfma_peak.hip -- what fp32 MFMA rate does THIS device sustain? (diagnostic)
// hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float rnd(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

// RANDOM: operands are 8 register pairs of random data (toggling like a real GEMM);
// otherwise near-constant operands (what a naive peak probe measures)
template <int NACC, bool RANDOM>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float a[8], b[8];
    unsigned seed = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
    for (int i = 0; i < 8; ++i) {
        a[i] = RANDOM ? rnd(seed) : a0 + threadIdx.x * 1e-3f;
        b[i] = RANDOM ? rnd(seed) : b0 - threadIdx.x * 1e-3f;
    }
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool RANDOM>
void run(int blocks_per_cu, int iters) {
    float *out;
    int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC, RANDOM><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NACC, RANDOM><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
    printf("%s NACC=%d waves/SIMD=%d: %.3f ms  %.1f TF/s\n", RANDOM ? "random  " : "constant", NACC, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<4, false>(1, 20000);
    run<4, true>(1, 20000);
    run<2, false>(4, 20000);
    run<2, true>(4, 20000);
    run<4, true>(2, 40000);
    run<4, false>(2, 40000);
    return 0;
}
