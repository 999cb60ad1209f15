// Developer microbenchmark (GPU box): does v_mfma_f32_32x32x2_f32 of one wavefront overlap with the VALU / LDS-write work of ANOTHER wavefront
// on the same SIMD?  Workgroups of 8 wavefronts: 0-3 run an MFMA chain (one per SIMD), 4-7 run `work` (one per SIMD).  Timed: MFMA only, work
// only, both.  If the two add up, the staging wavefronts of the trainer's GEMM cannot hide behind its MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// WORK: 0 = dependent v_fma chain, 1 = 8 independent v_fma chains, 2 = v_cndmask / compare mix (the fix-ups), 3 = ds_write_b128, 4 = v_pk_fma chain
// MK: 0 = one dependent accumulator chain, 1 = two independent chains, 2 = one chain with s_nop 11 (12 issue slots of 4 cycles) after every MFMA (the wait for the
// matrix pipe spent in s_nop instead of in the issue stage), 3 = v_mfma_f32_32x32x16_bf16 chain, 4 = one chain, s_nop 14; 5 / 6 = s_nop 13 / 7
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int WORK, int MK, int PRIO>
__global__ void __launch_bounds__(512) k(float* out, int n_mfma, int n_work, int mode, float a, float b) {
    __shared__ __attribute__((aligned(16))) float S[4][64 * 36 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.0f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 c, d;
        for (int i = 0; i < 16; ++i) { c[i] = 0.0f; d[i] = 1.0f; }
        bf16x8 ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
        for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (MK == 3) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c, 0, 0, 0); continue; }
                if (MK == 1 && (j & 1)) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d, 0, 0, 0);
                else c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
                if (MK == 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 11"); __builtin_amdgcn_sched_barrier(0); }
                if (MK == 5) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 13"); __builtin_amdgcn_sched_barrier(0); }
                if (MK == 6) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); }
                if (MK == 4) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 14"); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        for (int i = 0; i < 16; ++i) s += c[i] + d[i];
    } else {
        if (!(mode & 2)) return;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + (float)(lane + i);
        float* W = S[wave - 4];
        for (int i = 0; i < n_work; ++i) {
            if (WORK == 0) {
#pragma unroll
                for (int j = 0; j < 64; ++j) x[0] = __builtin_fmaf(x[0], a, b);
            } else if (WORK == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[q] = __builtin_fmaf(x[q], a, b);
            } else if (WORK == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[q] = (x[q] < (float)(i + j) && lane + q < n_work) ? x[q] * a : b;
            } else if (WORK == 3) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    *(f32x4*)(W + (lane & 63) * 36 + 4 * (j & 7)) = f32x4{x[0], x[1], x[2], x[3]};
                    asm volatile("" ::: "memory");
                }
            } else {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p = {x[0], x[1]}, q = {a, b};
#pragma unroll
                for (int j = 0; j < 64; ++j) p = __builtin_elementwise_fma(p, q, q);
                x[0] = p.x; x[1] = p.y;
            }
        }
        for (int i = 0; i < 8; ++i) s += x[i];
        s += W[lane];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WORK, int MK, int PRIO>
static float run(float* out, int wgs, int n_mfma, int n_work, int mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.0f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WORK, MK, PRIO>), dim3(wgs), dim3(512), 0, 0, out, n_mfma, n_work, mode, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e3f;
}
template <int WORK, int MK = 0, int PRIO = 0>
static void test(float* out, const char* name, int per_iter) {
    for (int wgs = 256; wgs <= 512; wgs *= 2) {
        const int n_mfma = 200;                                     // 3200 MFMAs = 204.8 k cycles of the matrix pipe
        // work sized to about the same time alone: calibrate with one run
        int n_work = 400;
        const float t1 = run<WORK, MK, PRIO>(out, wgs, n_mfma, n_work, 2);
        const float tm = run<WORK, MK, PRIO>(out, wgs, n_mfma, n_work, 1);
        n_work = (int)(n_work * tm / t1 * 0.5f);                    // half the MFMA time
        const float tw = run<WORK, MK, PRIO>(out, wgs, n_mfma, n_work, 2);
        const float tb = run<WORK, MK, PRIO>(out, wgs, n_mfma, n_work, 3);
        printf("%-28s wgs %3d: mfma %7.1f us, work %7.1f us (%d x %d instr), both %7.1f us -> overlap %.0f %% of the work\n", name, wgs, tm, tw, n_work, per_iter, tb,
               100.0 * (tm + tw - tb) / tw);
    }
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    test<0>(out, "dependent v_fma chain", 64);
    test<1>(out, "8 independent v_fma chains", 64);
    test<2>(out, "compare + cndmask + mul", 64);
    test<3>(out, "ds_write_b128", 16);
    test<4>(out, "v_pk_fma chain", 64);
    test<1, 0, 1>(out, "8 v_fma chains, prio 3", 64);
    test<1, 1>(out, "8 v_fma chains | 2 accs", 64);
    test<1, 2>(out, "8 v_fma chains | nop 11", 64);
    test<1, 4>(out, "8 v_fma chains | nop 14", 64);
    test<1, 4, 1>(out, "8 v_fma, prio 3 | nop 14", 64);
    test<1, 3>(out, "8 v_fma chains | bf16 mfma", 64);
    test<3, 2>(out, "ds_write_b128 | nop 11", 16);
    test<3, 3>(out, "ds_write_b128 | bf16 mfma", 16);
    test<2, 4>(out, "cmp+cndmask+mul | nop 14", 64);
    test<1, 5>(out, "8 v_fma chains | nop 13", 64);
    test<1, 6>(out, "8 v_fma chains | nop 7", 64);
    test<2, 5>(out, "cmp+cndmask+mul | nop 13", 64);
    test<3, 5>(out, "ds_write_b128 | nop 13", 16);
    return 0;
}
