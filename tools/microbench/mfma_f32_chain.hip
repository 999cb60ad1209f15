// Developer microbenchmark (GPU box): issue rate of v_mfma_f32_32x32x2_f32 with one dependent accumulator chain vs two independent
// ones, one and two wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_chain mfma_f32_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int n, float a, float b) {
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.0f; c1[i] = 1.0f; }
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            if (CHAINS == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    for (int wgs = 256; wgs <= 1024; wgs *= 2)
        for (int chains = 1; chains <= 2; ++chains) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (chains == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, n, 1.0f, 2.0f);
                else hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, n, 1.0f, 2.0f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)n * 8 * chains;                       // MFMAs per wave
            const double waves_per_simd = wgs * 4 / 1024.0;
            printf("wgs %4d (%.0f waves/SIMD) chains %d: %.1f us, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", wgs, waves_per_simd, chains, ms * 1e3,
                   ms * 1e6 / (mf * waves_per_simd), mf * wgs * 4 * 4096.0 / (ms * 1e-3) * 1e-12);
        }
    return 0;
}
