// ppo_dw_direct.hip - the weight / bias gradients of a minibatch step, gfx950, fp32 MFMA, operands straight from memory (no LDS staging).
//
//   [dW | db] [N1, N2 + 1] = dZ^T [N1, rows] . [X | 1] [rows, N2 + 1]        (dZ [rows, N1], X [rows, N2] row-major; rows = the minibatch = the K of the product)
//
// Both operands are contiguous along their OUTPUT index (n1 resp. n2) and strided along k.  That is exactly what a fragment of v_mfma_f32_16x16x4_f32
// needs - lane (i = l % 16, kk = l / 16) supplies operand[k0 + kk][tile row / column i] - when a tile is allowed to take every 4th row / column:
//   lane (lr, kk) loads operand[k0 + kk][c0 + 4 lr .. + 3] with ONE dwordx4; register q of it is the fragment of "tile q" = columns c0 + 4 j + q, j = 0..15.
// One dwordx4 per operand therefore feeds 4 x 4 = 16 MFMAs (a 64 x 64 block of the output, interleaved tiles) per 4-k step: 2 loads per 512 cycles of the
// matrix pipe, no LDS, no barrier in the K loop; a lane's four B tiles are 4 consecutive output columns, so the block leaves as dwordx4 rows as well.
// (k_gemm of ppo_kernels.hip stages both operands through LDS in 64 x 32 tiles with eight dword loads per thread and tile: 28 % of the fp32 MFMA peak on
// these shapes.)
//
// Work decomposition: a task = one 64 x 64 output block of one problem; a workgroup = 4 wavefronts = 4 consecutive 256-row pieces of one 1024-row chunk
// of one task; the four partial blocks are summed through LDS (fixed order) into the chunk's slab part[chunk][N1][N2 + 1], and the slabs of a problem are
// summed by tfp_sum_partials_multi as before (fixed order: deterministic results).  Loads run 4 steps ahead of the MFMAs (a register ring).
// What bounds it (MI355X, the eight products of a minibatch step, rows = 8192: 64 us against 92 us for k_gemm_group; 41 us of matrix-pipe time at the
// nominal clock, 47 us at the 2.1 GHz a pure MFMA stream sustains - tools/microbench/mfma16_rate.hip): how the blocks fall on the CUs.  A workgroup
// is a fixed piece of work (1024 MFMAs per wavefront: 15.6 us of a CU's matrix pipes) and the dispatcher hands workgroups to CUs by free slots, not by
// load: with 776 equal workgroups a few CUs carried four (75 us).  The blocks of the two heads (N1 = 9 / 1) now cost a quarter (one plain A tile) and come
// last, which leaves 744 full workgroups = at most three per CU: 64 us.  Below that: 544 of them would still be full units (2.1 per CU: some CUs carry
// three) - a persistent form that pulls blocks from a queue, partial blocks as plain tiles, would come to ~45 us (not built).  Measured and not kept:
// 128-row pieces, 8 steps of loads in flight instead of 4, a 2 x 2 arrangement of the wavefronts on 128 x 128 blocks with the second fetch of every
// operand left to L1, and the same with both operands staged through LDS (one fetch per workgroup): all 75 - 85 us - the operand stream (5 TB/s) is not
// what binds.  Counters of the kept form (profiles/r6_zz_dw_pmc.txt): the matrix pipes are busy 72 % of the launch (98.6 M MFMA-busy cycles over 1024 SIMDs =
// 96 k of 134 k cycles), L2 hit rate 71 %, L1 53 %.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/trifinger_ppo.h"

typedef float w4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float gfloat;

#define DW_MAXP 8
#define DW_MAXTASK 160
#ifndef DW_WROWS
#define DW_WROWS 256          // rows (k) per wavefront
#endif
#define DW_CHUNK (4 * DW_WROWS)   // rows per workgroup = per slab
#define DW_DEPTH 4            // K steps in flight
#define DW_LP 68              // LDS pitch of a 64-wide block row

struct DwProblem { const float* A; const float* B; float* part; int rows, N1, N2; };
struct DwArgs {
    DwProblem p[DW_MAXP];
    unsigned task[DW_MAXTASK];      // problem | a-block << 4 | b-block << 12
    int ntasks, nchunks;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ w4 dw_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(w4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ float dw_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// the K loop of one wavefront: 16 accumulators (A tile r x B tile q), loads DW_DEPTH steps ahead.  ONES: the block holds the bias column - the lane also
// sums its A values (asum[r] = sum over the lane's k of dZ[k][a0 + 4 lr + r]); B is never touched in registers (a select on the fragments made the
// compiler wait for ALL loads in flight at the head of every iteration).  TINY (N1 <= 16: the mu / value heads): ONE plain A tile (row lr, a dword per
// lane) against the four B tiles - 4 MFMAs per step instead of 16, so that these blocks do not cost a full unit of a CU's time each.
template <bool ONES, bool TINY>
__device__ __forceinline__ void dw_loop(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, unsigned va, unsigned vb, unsigned oa, unsigned ob, unsigned sa,
                                        unsigned sb, w4 (&acc)[4][4], w4& asum) {
    w4 fa[DW_DEPTH], fb[DW_DEPTH];
#pragma unroll
    for (int d = 0; d < DW_DEPTH; ++d) {
        if (TINY) fa[d][0] = dw_load1(ra, va, oa); else fa[d] = dw_load4(ra, va, oa);
        fb[d] = dw_load4(rb, vb, ob);
        oa += sa; ob += sb;
    }
    for (int t = 0; t < DW_WROWS / 4; t += DW_DEPTH) {
#pragma unroll
        for (int d = 0; d < DW_DEPTH; ++d) {
            const w4 x = fa[d], y = fb[d];
#pragma unroll
            for (int r = 0; r < (TINY ? 1 : 4); ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[r], y[q], acc[r][q], 0, 0, 0);
            }
            if (ONES) { if (TINY) asum[0] += x[0]; else asum += x; }
            if (TINY) fa[d][0] = dw_load1(ra, va, oa); else fa[d] = dw_load4(ra, va, oa);      // step t + d + DW_DEPTH (past the wavefront's piece: loaded, never used)
            fb[d] = dw_load4(rb, vb, ob);
            oa += sa; ob += sb;
            __builtin_amdgcn_sched_group_barrier(0x008, TINY ? 4 : 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
    }
}

// three workgroups per CU at 8 chunks x 93 full blocks = 744 workgroups: no CU carries a fourth (the launch ends with its most loaded CU)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) k_dw_direct(const DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red_[];   // [2][64 * DW_LP]: 34.8 KB
    float (*red)[64 * DW_LP] = (float (*)[64 * DW_LP])red_;
    // Workgroups go to the 8 XCDs round-robin in launch order and each XCD has its own 4 MB L2: every block of ONE chunk is sent to one XCD, so that the
    // chunk's rows of dZ and X are fetched from memory once and re-read - 4 to 7 times, once per block row / column - out of that L2
    int ti, chunk;
    if ((a.nchunks & 7) == 0) { const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3; chunk = xcd + 8 * (idx / a.ntasks); ti = idx % a.ntasks; }
    else { ti = (int)blockIdx.x % a.ntasks; chunk = (int)blockIdx.x / a.ntasks; }
    const unsigned tk = a.task[ti];
    const DwProblem& P = a.p[tk & 15u];
    const int a0 = (int)((tk >> 4) & 255u) * 64, b0 = (int)(tk >> 12) * 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15, kk = lane >> 4;
    const int N1 = P.N1, N2 = P.N2, rows = P.rows;
    const int krow0 = chunk * DW_CHUNK + wave * DW_WROWS;
    if (chunk * DW_CHUNK >= rows) return;                          // (uniform over the workgroup)
    const bool tiny = N1 <= 16;                                    // uniform
    const __amdgpu_buffer_rsrc_t ra = dw_rsrc(P.A, (unsigned)rows * (unsigned)N1 * 4u), rb = dw_rsrc(P.B, (unsigned)rows * (unsigned)N2 * 4u);
    // lane part of the addresses: row kk of the step, columns c0 + 4 lr .. + 3 (tiny: column lr of dZ); the step's first row rides in the scalar offset
    const unsigned va = 4u * (unsigned)(kk * N1 + (tiny ? min(lr, N1 - 1) : a0 + 4 * lr)), vb = 4u * (unsigned)(kk * N2 + b0 + 4 * lr);
    const unsigned sa = 16u * (unsigned)N1, sb = 16u * (unsigned)N2;               // bytes per step (4 rows)
    const bool has_ones = b0 <= N2 && N2 < b0 + 64;                // uniform: output column N2 (the bias gradient) lies in this block
    w4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = w4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    w4 asum = w4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned oa = (unsigned)krow0 * (unsigned)N1 * 4u, ob = (unsigned)krow0 * (unsigned)N2 * 4u;
    if (krow0 < rows) {
        if (tiny) { if (has_ones) dw_loop<true, true>(ra, rb, va, vb, oa, ob, sa, sb, acc, asum); else dw_loop<false, true>(ra, rb, va, vb, oa, ob, sa, sb, acc, asum); }
        else { if (has_ones) dw_loop<true, false>(ra, rb, va, vb, oa, ob, sa, sb, acc, asum); else dw_loop<false, false>(ra, rb, va, vb, oa, ob, sa, sb, acc, asum); }
    }
    // ---- the four partial blocks -> their sum (w0 + w2) + (w1 + w3), a fixed order -> the chunk's slab, through TWO block buffers in LDS (35 KB).
    // C layout: tile (r, q), lane l, register t holds output row a0 + 4 (4 (l / 16) + t) + r, column b0 + 4 (l % 16) + q: a lane's four q are 4 consecutive
    // columns.  (tiny: the one A tile is a plain one - register t of tile (0, q) holds output row 4 (l / 16) + t; rows 16 .. 63 of the buffers are not used) ----
    if (has_ones) {                                                // the bias column replaces the block's column N2: the lane's sums over its k, then over
        const int cl = N2 - b0, lr_c = cl >> 2, q_c = cl & 3;      // the four k-groups of the wavefront, into the registers that hold that column
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (tiny && r > 0) break;
            float v = asum[r];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            // the sum of block row 4 lr' + r (tiny: row lr') sits in every lane with l % 16 = lr'; the lane that holds that row of column N2 in register t is
            // the one with l % 16 = lr_c whose 16 kk + 4 t (tiny: 4 kk + t) is that row
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float got = __shfl(v, 4 * kk + t, 64);
                if (lr == lr_c) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { if (q == q_c) acc[r][q][t] = got; }
                }
            }
        }
    }
    auto blk = [&](int buf, int r, int t) __attribute__((always_inline)) { return (w4*)(&red[buf][((tiny ? 4 * kk + t : 16 * kk + 4 * t + r)) * DW_LP + 4 * lr]); };
    if (wave >= 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (tiny && r > 0) break;
#pragma unroll
            for (int t = 0; t < 4; ++t) *blk(wave - 2, r, t) = w4{acc[r][0][t], acc[r][1][t], acc[r][2][t], acc[r][3][t]};
        }
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (tiny && r > 0) break;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                w4* p = blk(wave, r, t);
                *p = w4{acc[r][0][t], acc[r][1][t], acc[r][2][t], acc[r][3][t]} + *p;
            }
        }
    }
    __syncthreads();
    const int ld = N2 + 1;
    gfloat* slab = (gfloat*)(P.part + (size_t)chunk * (size_t)N1 * (size_t)ld);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = (int)(threadIdx.x >> 4) + 16 * it, c = 4 * (int)(threadIdx.x & 15);
        if (a0 + row < N1) {
            const w4 s = *(const w4*)(&red[0][row * DW_LP + c]) + *(const w4*)(&red[1][row * DW_LP + c]);
            gfloat* o = slab + (size_t)(a0 + row) * ld + b0 + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) { if (b0 + c + q < ld) o[q] = s[q]; }
        }
    }
}

extern "C" {

// [dW | db] chunk slabs of n <= 8 problems in ONE launch: part[p] receives ceil(rows[p] / 1024) slabs of N1[p] * (N2[p] + 1) floats (sum them with
// tfp_sum_partials_multi, splits = ceil(rows / 1024)).  A[p] = dZ [rows, N1], B[p] = X [rows, N2], row-major.  -4: more than 160 blocks of 64 x 64
// (the caller uses tfp_gemm_tn_partials_group).
int tfp_gemm_tn_partials_direct_chunk(void) { return DW_CHUNK; }
int tfp_gemm_tn_partials_direct(const void* const* A, const void* const* B, void* const* part, const int32_t* rows, const int32_t* N1, const int32_t* N2,
                                int32_t n, void* stream) {
    if (n <= 0 || n > DW_MAXP) return -1;
    DwArgs a; memset(&a, 0, sizeof(a));
    int nt = 0, maxrows = 0;
    const int blk = 64;
    for (int p = 0; p < n; ++p) {
        if (rows[p] <= 0 || N1[p] <= 0 || N2[p] <= 0 || !A[p] || !B[p] || !part[p]) return -1;
        if ((uint64_t)rows[p] * (uint64_t)(N1[p] > N2[p] ? N1[p] : N2[p]) * 4u >= (1ull << 32)) return -4;
        a.p[p].A = (const float*)A[p]; a.p[p].B = (const float*)B[p]; a.p[p].part = (float*)part[p];
        a.p[p].rows = rows[p]; a.p[p].N1 = N1[p]; a.p[p].N2 = N2[p];
        if (rows[p] > maxrows) maxrows = rows[p];
    }
    // the blocks of the full problems first, those of the tiny ones (N1 <= 16: a quarter of the time each) last: they fill what the others leave
    for (int pass = 0; pass < 2; ++pass) {
        for (int p = 0; p < n; ++p) {
            if ((N1[p] <= 16) != (pass == 1)) continue;
            const int ab = (N1[p] + blk - 1) / blk, bb = (N2[p] + 1 + blk - 1) / blk;
            if (ab > 255 || bb > 255) return -4;
            for (int i = 0; i < ab; ++i) for (int j = 0; j < bb; ++j) {
                if (nt >= DW_MAXTASK) return -4;
                a.task[nt++] = (unsigned)p | ((unsigned)i << 4) | ((unsigned)j << 12);
            }
        }
    }
    a.ntasks = nt;
    a.nchunks = (maxrows + DW_CHUNK - 1) / DW_CHUNK;
    const size_t lds = sizeof(float) * 2 * 64 * DW_LP;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_dw_direct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_dw_direct, dim3((unsigned)(nt * a.nchunks)), dim3(256), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
