// Developer microbenchmark (GPU box): issue rate of v_mfma_f32_16x16x4_f32 (and v_mfma_f32_32x32x2_f32) with 16 / 4 independent accumulators, one or two
// wavefronts per SIMD:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_rate tools/microbench/mfma16_rate.hip && /tmp/mfma16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float w4 __attribute__((ext_vector_type(4)));
typedef float w16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k16(float* out, int iters, unsigned long long* cyc) {
    w4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = w4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void __launch_bounds__(256) k32(float* out, int iters, unsigned long long* cyc) {
    w16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        for (int kind = 0; kind < 2; ++kind) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k16, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
                else hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n = (double)iters * (kind == 0 ? 16 : 4);
            const double flop = n * (kind == 0 ? 2048.0 : 4096.0) * wgs * 4;
            printf("%s, %4d workgroups of 4 wavefronts: %.1f us, %.1f counter ticks per MFMA in wavefront 0, %.1f ns per MFMA, %.1f TFLOP/s\n",
                   kind == 0 ? "16x16x4 f32 (16 accumulators)" : "32x32x2 f32 ( 4 accumulators)", wgs, ms * 1e3, c / n, ms * 1e6 / n, flop / (ms * 1e-3) * 1e-12);
        }
    }
    return 0;
}
