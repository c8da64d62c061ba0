// tools/mfma_peak_bf16.hip -- sustained v_mfma_f32_32x32x16_bf16 rate on THIS device, random operands
// hipcc -O3 --offload-arch=gfx950 tools/mfma_peak_bf16.hip -o /tmp/p16 && /tmp/p16
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ float rnd(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a[4], b[4];
    unsigned seed = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)rnd(seed);
            b[i][e] = (__bf16)rnd(seed);
        }
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
    float *out;
    int blocks = 256 * blocks_per_cu;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NACC><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("bf16 32x32x16 NACC=%d waves/SIMD=%d: %.3f ms  %.1f TF/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    run<4>(1, 40000);
    run<4>(2, 40000);
    run<4>(4, 20000);
    return 0;
}
