// Developer microbenchmark (GPU box): single-wave issue rate of packed fp32 VALU ops on gfx950 (v_pk_fma_f32, v_pk_mul_f32,
// v_pk_add_f32) next to v_fma_f32, for 1 and 8 independent chains, 1 wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int CH, int OP>
__global__ void k(float* out, long long* cyc, int iters) {
    float2v a[CH];
    float2v m = {1.0000001f, 0.9999999f}, c = {1e-7f, 2e-7f};
#pragma unroll
    for (int i = 0; i < CH; ++i) { a[i].x = (float)threadIdx.x + i; a[i].y = (float)threadIdx.x - i; }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(m), "v"(c));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int CH, int OP> void run(int threads, const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096 * sizeof(float)); (void)hipMalloc(&cyc, 64 * sizeof(long long));
    int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<CH, OP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long hh[64]; (void)hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
    int nw = threads / 64; long long lo = hh[0], hi = hh[1];
    for (int w = 0; w < nw; ++w) { if (hh[2 * w] < lo) lo = hh[2 * w]; if (hh[2 * w + 1] > hi) hi = hh[2 * w + 1]; }
    int wps = nw / 4 ? nw / 4 : 1;
    printf("%-28s chains %d, %d wave/SIMD: %.2f cycles per instruction per wave (all waves: %.3f instr/cycle/SIMD)\n", name, CH, wps,
           (double)(hi - lo) / ((double)iters * 64.0), (double)wps * iters * 64.0 / (double)(hi - lo));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run<1, 3>(th, "v_fma_f32"); run<8, 3>(th, "v_fma_f32");
        run<1, 0>(th, "v_pk_fma_f32"); run<8, 0>(th, "v_pk_fma_f32");
        run<8, 4>(th, "v_pk_fma_f32 op_sel_hi");
        run<1, 1>(th, "v_pk_mul_f32"); run<8, 1>(th, "v_pk_mul_f32");
        run<1, 2>(th, "v_pk_add_f32"); run<8, 2>(th, "v_pk_add_f32");
    }
    return 0;
}
