// Developer microbenchmark (GPU box): how fast can ONE wave issue fp32 VALU work on gfx950?
//   chains = number of independent dependency chains interleaved in program order (1 = fully dependent).
// Prints shader cycles per v_fma_f32 for 1 wave per SIMD and for 2 waves per SIMD (same SIMD, block of 512 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int CH>
__global__ void k(float* out, long long* cyc, int iters) {
    float a[CH];
    float m = 1.0000001f, c = 1e-7f;
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = (float)threadIdx.x + i;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int CH> void run(int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&cyc, 64 * sizeof(long long));
    int iters = 2000;
    hipLaunchKernelGGL(k<CH>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<CH>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long hh[64]; hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
    int nw = threads / 64; long long lo = hh[0], hi = hh[1], self = 0;
    for (int w = 0; w < nw; ++w) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; if (hh[2 * w + 1] - hh[2 * w] > self) self = hh[2 * w + 1] - hh[2 * w]; }
    int wps = nw / 4 ? nw / 4 : 1;
    printf("chains %2d  threads %4d (%d wave/SIMD): slowest wave %.2f cycles per v_fma_f32; SIMD throughput %.3f wave-instr per cycle\n", CH, threads, wps,
           (double)self / ((double)iters * 64.0), (double)wps * iters * 64.0 / (double)(hi - lo));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {64, 256, 512, 1024}) { run<1>(th); run<2>(th); run<4>(th); run<8>(th); }
    return 0;
}
