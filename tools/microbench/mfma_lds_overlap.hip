// Developer microbenchmark (GPU box): what the LDS traffic of a GEMM tile loop costs the MFMA stream on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_overlap mfma_lds_overlap.hip
// A workgroup = NM MFMA wavefronts per SIMD (each: 16 v_mfma_f32_32x32x2_f32 per "tile", ACCS independent accumulators used in turn, NOP wait
// states of s_nop behind every MFMA, RD ds_read_b128 per tile spread between the MFMAs) + NW LDS-writing wavefronts per SIMD (4 ds_write_b128 per
// tile each).  No barriers, no global memory: only the issue / LDS interference.  Reported: time per tile against the 16 x 64 cycles of the MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int ACCS, int NOP, int RD, int NW>
__global__ void __launch_bounds__(64 * 4 * (NM + NW)) k(float* out, int tiles, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float S[8][64 * 36];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.0f;
    if (wave < 4 * NM) {
        f32x16 c[2];
        for (int i = 0; i < 16; ++i) { c[0][i] = 0.0f; c[1][i] = 1.0f; }
        f32x4 a[8];
        for (int i = 0; i < 8; ++i) a[i] = f32x4{a0, b0, a0, b0};
        const float* R = S[wave & 7] + (lane & 31) * 36 + 4 * (lane >> 5);
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (RD && (j % (16 / (RD ? RD : 1))) == 0) { a[(j / (16 / (RD ? RD : 1))) & 7] = *(const volatile f32x4*)(R + 8 * ((j / (16 / (RD ? RD : 1))) & 3)); }
                c[j % ACCS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j >> 2) & 7][j & 3], a[((j >> 2) + 4) & 7][j & 3], c[j % ACCS], 0, 0, 0);
                if (NOP >= 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" :: "n"(NOP >= 0 ? NOP : 0)); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i];
    } else {
        float* W = S[wave & 7] + lane * 36;
        const f32x4 x = f32x4{a0, b0, (float)lane, 1.0f};
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { *(volatile f32x4*)(W + 4 * j) = x; }
            __builtin_amdgcn_s_sleep(NM * 8);                        // about one tile of MFMA time between the bursts (64 cycles per unit)
        }
        s = W[1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int ACCS, int NOP, int RD, int NW>
static void test(float* out, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = 200;
    float ms = 0.0f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, ACCS, NOP, RD, NW>), dim3(256), dim3(64 * 4 * (NM + NW)), 0, 0, out, tiles, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ideal_cycles = (double)tiles * NM * 16 * 64;
    printf("%-58s %7.1f us = %.2f x the MFMA time at 2.4 GHz\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / ideal_cycles);
}
int main() {
    float* out; (void)hipMalloc(&out, 1 << 24);
    test<1, 1, -1, 0, 0>(out, "1 MFMA wave/SIMD");
    test<2, 1, -1, 0, 0>(out, "2 MFMA waves/SIMD");
    test<1, 1, -1, 8, 0>(out, "1 MFMA wave/SIMD, 8 ds_read_b128 per tile");
    test<2, 1, -1, 8, 0>(out, "2 MFMA waves/SIMD, 8 ds_read_b128 per tile");
    test<1, 1, -1, 0, 1>(out, "1 MFMA wave/SIMD + writer");
    test<2, 1, -1, 0, 1>(out, "2 MFMA waves/SIMD + writer");
    test<2, 1, -1, 0, 2>(out, "2 MFMA waves/SIMD + 2 writers");
    test<2, 1, -1, 8, 2>(out, "2 MFMA waves/SIMD, 8 reads + 2 writers   (the GEMM today)");
    test<2, 1, 11, 8, 2>(out, "2 MFMA waves/SIMD, 8 reads + 2 writers, s_nop 11");
    test<1, 2, -1, 8, 1>(out, "1 MFMA wave/SIMD x 2 accs, 8 reads + 1 writer");
    test<1, 2, 11, 8, 1>(out, "1 MFMA wave/SIMD x 2 accs, 8 reads + 1 writer, s_nop 11");
    test<1, 2, 13, 8, 1>(out, "1 MFMA wave/SIMD x 2 accs, 8 reads + 1 writer, s_nop 13");
    test<1, 2, 11, 8, 2>(out, "1 MFMA wave/SIMD x 2 accs, 8 reads + 2 writers, s_nop 11");
    test<1, 1, 11, 8, 1>(out, "1 MFMA wave/SIMD x 1 acc, 8 reads + 1 writer, s_nop 11");
    test<1, 1, 13, 8, 1>(out, "1 MFMA wave/SIMD x 1 acc, 8 reads + 1 writer, s_nop 13");
    return 0;
}
