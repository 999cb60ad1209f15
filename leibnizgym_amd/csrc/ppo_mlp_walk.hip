// ppo_mlp_walk.hip - the trainer's two MLPs as ONE launch per direction ("network walk"), gfx950, fp32 MFMA.
//
// The per-layer products of ppo_kernels.hip (k_gemm / k_gemm_group) pay, for every layer, a launch, a cold first tile, a tail, a round trip of
// the activations through memory, and 64-wide tiles that pad N = 200 to 256 and N = 100 to 128.  Here a workgroup owns 32 ROWS of the minibatch
// and walks ALL layers of one network with them:
//
//   forward   x [32, D0] -> LDS;  y1 = elu(x W1^T + b1) -> LDS (+ memory);  y2 = elu(y1 W2^T + b2) -> LDS (+ memory);  ...;  y4 -> memory
//   backward  g [32, D4] -> LDS;  dZ3 = (g W4) * elu'(y3) -> LDS + memory;  dZ2 = (dZ3 W3) * elu'(y2) -> LDS + memory;  dZ1 = (dZ2 W2) * elu'(y1) -> memory
//
// (the weight gradients dW_l = dZ_l^T y_{l-1} stay a split-K product over the whole minibatch: tfp_gemm_tn_partials_group).  M = 8192 rows and two
// networks are 512 workgroups = two per CU, one round; the activations of a row block never leave the CU (two LDS buffers used alternately: 27 + 52 KB
// for 41/113 -> 400 -> 200 -> 100 -> 9/1, so two workgroups share a CU's 160 KB), the weights stream from L2 (1 MB for both networks).
// Two INDEPENDENT workgroups per CU = two wavefronts per SIMD that are not tied by each other's barriers: the epilogue, the barrier waits and the memory
// latencies of one run in the shadow of the other's MFMAs (a first form with 64 rows and one wavefront per SIMD spent half its time in exactly those:
// docs/HISTORY.md section 12).
//
// Inside a layer the four wavefronts of a workgroup split the [32, N] output by columns: a wavefront owns all 32 rows (two 16-row tiles) and a quarter
// of the 16-column tiles (N = 400: 7 / 6 / 6 / 6 tiles, 200: 4 / 3 / 3 / 3, 100: 2 / 2 / 2 / 1), i.e. up to 14 accumulators of v_mfma_f32_16x16x4_f32.
// A wavefront's weights need no LDS: the B fragment of a 16 x 16 tile is loaded from memory straight in MFMA layout -
//   forward  (W [N, K], k contiguous): lane (n = l % 16, kk = l / 16) reads W[n][k0 + 4 kk .. + 3] with one dwordx4 load and feeds the four values
//            to four MFMAs; the A fragment is the same shape out of LDS (one ds_read_b128), so MFMA s of a 16-k step sums k0 + 4 kk + s, kk = 0..3:
//            the order of the k inside a step is free as long as both operands use the same one;
//   backward (W [K, N], n contiguous): the same lane reads W[k0 + 4 kk + s][n], s = 0..3, with four dword loads (a wave instruction = 4 rows x 64 B).
// A step visits the column tiles in turn: the 8 MFMAs of tile j (4 k-groups x 2 row tiles), then the load of tile j's fragment for the NEXT step into
// the registers those MFMAs have just released.  LDS rows have a pitch of 16 q + 4 floats: conflict-free for the b128 fragment reads (16 rows x 4
// k-groups) and for the epilogue's column writes.
//
// The epilogue of a layer adds the bias, applies ELU (forward) or multiplies by elu'(y) = y > 0 ? 1 : y + 1 of the saved output (backward) and writes
// the tile to LDS as the next layer's A operand (zero in the padding columns).  Hidden outputs reach memory FROM LDS, as dwordx4 rows, one chunk per
// thread and K step of the NEXT layer (the buffer is that layer's A operand and stays intact throughout): 46 MB of stores per forward pass are spread
// over the whole launch instead of arriving in bursts at the layer boundaries.  ELU is x > 0 ? x : exp(x) - 1 with exp on v_exp_f32 (the form torch's
// kernel uses: exp - 1, not expm1; absolute error <= 1.2e-7).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/trifinger_ppo.h"

typedef float w4 __attribute__((ext_vector_type(4)));
typedef float w4u __attribute__((ext_vector_type(4), aligned(4)));      // a dwordx4 load needs dword alignment only (rows of 41 / 113 floats)

#ifndef WALK_DBG
#define WALK_DBG 0     // developer builds (timing only, wrong results): 1 = the B loads do not advance along K (every step re-reads the first one: L1 hits),
#endif                 // 2 = no MFMAs (loads, LDS traffic and epilogues only)
#ifdef WALK_TIMING     // developer build: s_memtime stamps of one workgroup (per wavefront: start, input staged, then per layer: K loop done, epilogue done, barrier passed)
__device__ unsigned long long g_walk_t[4 * 32];
#ifndef WALK_TIMING_WG
#define WALK_TIMING_WG 0
#endif
#define WSTAMP(i_) do { if (blockIdx.x == WALK_TIMING_WG && (threadIdx.x & 63) == 0) g_walk_t[(threadIdx.x >> 6) * 32 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define WSTAMP(i_)
#endif
#define WALK_ROWS 32
#define WALK_MAXL 4

struct WalkNet {
    const float* x;                    // forward: network input [M, dim[0]]; backward: gradient of the network output [M, dim[nl]]
    const float* W[WALK_MAXL];         // W[l]: [dim[l + 1], dim[l]] row-major (torch.nn.Linear.weight)
    const float* b[WALK_MAXL];         // forward: bias [dim[l + 1]]
    const float* yin[WALK_MAXL];       // backward: yin[l] = saved output of layer l, [M, dim[l + 1]]
    float* y[WALK_MAXL];               // forward: output of layer l (NULL: not stored; the last one must be given); backward: y[l] = dZ of layer l (l < nl - 1)
    int dim[WALK_MAXL + 1];
    int act[WALK_MAXL];                // 1: ELU behind layer l
    int nl;
};
struct WalkArgs { WalkNet net[2]; int n_nets, M, blocks_per_net, p_floats; };

__device__ __forceinline__ int pitch_of(int d) { return ((d + 15) & ~15) + 4; }

__device__ __forceinline__ float elu_fast(float v) {
    // exp(v) - 1 for v <= 0 through v_exp_f32 (2^x): |error| <= 1.2e-7 absolute
    const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896340736f) - 1.0f;
    return v > 0.0f ? v : e;
}

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float gfloat;           // a pointer the compiler must treat as global memory (a flat access counts in vmcnt AND lgkmcnt)
// Buffer addressing (raw buffer, stride 0): a load takes the matrix as a 128-bit resource in scalar registers, a 32-bit per-lane byte offset and a scalar
// byte offset - the K position of the step - so the K loop holds no address arithmetic at all; what lies past the end of the matrix reads as zero and
// stores past the end are dropped (the range check of the hardware), which is what the ragged edges need.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ w4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(w4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// What the step before left in the LDS operand of this one (a hidden output / dZ, [32][pitch], width W, W % 4 == 0) on its way to memory [M, W]: every
// thread owns the 16-byte chunks tid, tid + 256, ... of the row block (memory is linear in the chunk index) and moves ONE of them per K step of the
// running layer - read from LDS at the top of the step, stored at its end, no branch: a thread that has run out of chunks stores out of range.
struct StreamOut {
    u4 dst;                            // buffer resource of the row block in memory (base, 0, bytes, format word)
    const float* src;                  // the LDS operand
    int pitch, cpr, total, dr, dc;     // chunks per row, chunks of the block, the (row, chunk-in-row) advance of 256 chunks
    int idx, row, c4;
    unsigned woff;
    __device__ __forceinline__ void init(float* G, const float* Xs, int pitch_, int W, int row0, int M) {
        const int rows = min(WALK_ROWS, M - row0);
        const uintptr_t base = (uintptr_t)(G + (size_t)row0 * W);
        dst = u4{(unsigned)base, (unsigned)(base >> 32) & 0xffffu, (unsigned)(rows * W) * 4u, 0x00020000u};
        src = Xs; pitch = pitch_;
        cpr = W >> 2;
        total = rows * cpr;
        dr = 256 / cpr; dc = 256 - dr * cpr;
        row = (int)threadIdx.x / cpr; c4 = (int)threadIdx.x - row * cpr;
        idx = threadIdx.x;
    }
    __device__ __forceinline__ w4 read() {
        const bool ok = idx < total;
        woff = ok ? (unsigned)idx * 16u : (unsigned)total * 16u;      // = num_records: just past the end (a huge offset could wrap in the range check)
        return *(const w4*)(src + (ok ? row : 0) * pitch + 4 * (ok ? c4 : 0));
    }
    __device__ __forceinline__ void write(const w4& v) {
        // The store is written as inline assembly ON PURPOSE: on gfx9 loads and stores share vmcnt, and with a store pending the compiler's wait-count
        // pass treats the counter as out of order - every wait for a B fragment in the K loop becomes vmcnt(0), i.e. the wavefront sits out the latency of
        // the load it issued last, in every step.  Hidden from that pass the store only makes the hardware counter larger than the compiler's model, so
        // its waits stay safe (loads return in order among themselves), and the data registers are read in issue order (no expcnt on gfx9).
        const u4 rs = u4{(unsigned)__builtin_amdgcn_readfirstlane((int)dst[0]), (unsigned)__builtin_amdgcn_readfirstlane((int)dst[1]),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)dst[2]), (unsigned)__builtin_amdgcn_readfirstlane((int)dst[3])};
        // (s_nop 4: a vector instruction that writes a scalar register - the v_readfirstlane above - needs 5 wait states before a memory instruction reads
        // that register; the compiler inserts them for its own instructions, not in front of an asm: without them the store ran with a stale descriptor)
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(v), "v"(woff), "s"(rs) : "memory");
        idx += 256; c4 += dc; row += dr;
        if (c4 >= cpr) { c4 -= cpr; row += 1; }
    }
    // what is left when the K loop is over (or for a wavefront without tiles in this layer): ordinary stores, the compiler's own
    __device__ __forceinline__ void flush() {
        const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)dst[0]), d1 = (unsigned)__builtin_amdgcn_readfirstlane((int)dst[1]);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)d0) | ((uintptr_t)(d1 & 0xffffu) << 32)), 0,
                                                                             __builtin_amdgcn_readfirstlane((int)dst[2]), 0x00020000);
        while (__builtin_amdgcn_ballot_w64(idx < total) != 0ull) {
            const w4 v = read();
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rd, (int)woff, 0, 0);
            idx += 256; c4 += dc; row += dr;
            if (c4 >= cpr) { c4 -= cpr; row += 1; }
        }
    }
};

// after the last MFMA of a layer, before anything but an MFMA touches the accumulators.  The compiler inserts the wait states an 8-pass MFMA needs before a
// vector instruction reads its result; the explicit ones and the empty asm ties date from the inline-assembly form of the loop and are kept because they
// keep the epilogue's reads behind the whole K loop (64 cycles per layer, under 0.5% of the walk) - the shape the parity tests and profiles were taken on.
template <int NCT>
__device__ __forceinline__ void walk_mfma_drain(w4 (&acc)[2][NCT]) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // every accumulator passes through an (empty) volatile asm behind the wait states: a vector instruction that reads one cannot be scheduled above them
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NCT; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
}

// One layer for this wavefront: acc[i][j] += X[16 i .., :] . op(W)[:, 16 (ct0 + j) ..] over K, NCT column tiles (exact: no guards inside).
//   Xs: LDS [32][px], columns K .. ceil16(K) are zero.  KMAJ = false: W[n * ldw + k] (forward); true: W[k * ldw + n] (backward).  SO: a stream-out
//   rides along (one chunk per step).
// K = 16 nfull + tail.  Every step is the same code: a fragment that reaches past the end of the matrix reads zeros, one that reaches past the end of a
// row (forward) reads the next row's finite values against A values that are zero.  Only a forward K that is no multiple of 4 (the first layer: 41, 113)
// has a dwordx4 fragment straddling the end of the LAST row with values that count: that ragged step loads every element on its own.
template <int NCT, bool KMAJ, bool SO, class EpiLoad>
__device__ __forceinline__ void walk_layer(const float* Xs, int px, const float* __restrict__ W, int ldw, int K, int N, int ct0, int nct, StreamOut& so,
                                           w4 (&acc)[2][NCT], EpiLoad&& epi_load) {
    const int lane = threadIdx.x & 63, lr = lane & 15, kk = lane >> 4;
    // (measured on gfx950: the range check of a raw buffer covers vector offset + scalar offset - with num_records reduced by the scalar offset the last
    // row of W came back as zeros one step early - so one resource serves every step and the K position rides in the scalar offset)
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W, (unsigned)(KMAJ ? K * ldw : N * ldw) * 4u);
    unsigned boff[NCT];                                            // this lane's fragment of tile j at k0 = 0: byte offset into W
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
        const int n = min((ct0 + min(j, nct - 1)) * 16 + lr, N - 1);                   // columns past N / tiles past nct: copies (discarded in the epilogue)
        boff[j] = 4u * (KMAJ ? (unsigned)(n + 4 * kk * ldw) : (unsigned)(n * ldw + 4 * kk));
    }
    const float* xa = Xs + lr * px + 4 * kk;
    const bool ragged = !KMAJ && (K & 3) != 0;
    const int nvec = ragged ? (K >> 4) : ((K + 15) >> 4), tail = ragged ? (K & 15) : 0;     // steps of the uniform form; elements of the ragged step
    // The MFMAs are builtins and the order of a step is pinned with sched_group_barrier (the LDS reads of the next A fragments, then per column tile its
    // 8 MFMAs followed by the load of its next B fragment).  Left to itself the scheduler hoisted all loads to the top of the step, which made every wait
    // a vmcnt(0); sched_barrier(0) between tiles cut the region and brought accumulator copies (v_accvgpr shuffles) with it.  Writing the MFMAs as volatile
    // inline assembly was tried and dropped: the hazard recogniser does not see them, and the compiler placed vector copies of accumulators behind MFMAs
    // that had not finished (docs/HISTORY.md section 12).
    auto loadb = [&](int t, int j, w4& b) __attribute__((always_inline)) {
        const int k0 = WALK_DBG == 1 ? 0 : 16 * t;
        if (KMAJ) {
            const unsigned s0 = (unsigned)(k0 * ldw) * 4u, sl = (unsigned)ldw * 4u;      // rows k0 + s (the lane's 4 kk rows are in boff)
            b = w4{buf_load1(rw, boff[j], s0), buf_load1(rw, boff[j], s0 + sl), buf_load1(rw, boff[j], s0 + 2u * sl), buf_load1(rw, boff[j], s0 + 3u * sl)};
        } else {
            b = buf_load4(rw, boff[j], (unsigned)k0 * 4u);
        }
    };
    auto loadb_ragged = [&](int j, w4& b) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = min(16 * nvec + 4 * kk + s, K - 1);
            b[s] = buf_load1(rw, boff[j] + 4u * (unsigned)(k - 4 * kk), 0u);             // boff holds the 4 kk part
        }
    };
    auto loada = [&](int t, w4 (&a)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const w4*)(xa + i * 16 * px + 16 * t);
    };
    auto mma_tile = [&](const w4 (&a)[2], const w4& b, int j, int smax) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < smax) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (WALK_DBG == 2) { acc[i][j][0] += a[i][s] + b[s]; continue; }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[s], acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    w4 a[2], an[2], b[NCT];
    if (nvec == 0) {                                                // K < 16 and ragged: the ragged step alone
        loada(0, a);
#pragma unroll
        for (int j = 0; j < NCT; ++j) loadb_ragged(j, b[j]);
        if (SO) so.flush();
#pragma unroll
        for (int j = 0; j < NCT; ++j) { mma_tile(a, b[j], j, tail); epi_load(j); }
        return;
    }
    loada(0, a);
#pragma unroll
    for (int j = 0; j < NCT; ++j) loadb(0, j, b[j]);
    for (int t = 0; t + 1 < nvec; ++t) {
        w4 sv;
        if (SO) sv = so.read();
        loada(t + 1, an);
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            mma_tile(a, b[j], j, 4);
            loadb(t + 1, j, b[j]);
        }
        if (SO) so.write(sv);
        __builtin_amdgcn_sched_group_barrier(0x100, SO ? 3 : 2, 0);
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, KMAJ ? 4 : 1, 0);
        }
        a[0] = an[0]; a[1] = an[1];
    }
    if (tail) {
        loada(nvec, an);
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            mma_tile(a, b[j], j, 4);
            loadb_ragged(j, b[j]);
        }
        if (SO) so.flush();
#pragma unroll
        for (int j = 0; j < NCT; ++j) { mma_tile(an, b[j], j, tail); epi_load(j); }     // MFMA s covers k0 + 4 kk + s: needed while s < tail
    } else {
        if (SO) so.flush();
#pragma unroll
        for (int j = 0; j < NCT; ++j) { mma_tile(a, b[j], j, 4); epi_load(j); }         // the last step has no fragment to fetch: the epilogue's loads go there
    }
}

// layer + epilogue for a wavefront with NCT column tiles.  Ys: LDS output [32][py] or nullptr (last product).  G: memory output [M, N], written here only
// for the last product (the others leave through StreamOut during the next step).
//   BWD = false: v = acc + bias, ELU when act;  BWD = true: v = acc * elu'(E[row, col]) when E (the saved output of the layer this is the dZ of)
template <int NCT, bool BWD, bool SO>
__device__ __forceinline__ void walk_run(const float* Xs, int px, const float* __restrict__ W, int K, int N, const float* __restrict__ bias, int act,
                                         const float* __restrict__ E, float* Ys, int py, float* __restrict__ G, int row0, int M, int ct0, int nct,
                                         StreamOut& so, int stamp) {
    const int lane = threadIdx.x & 63, lr = lane & 15, kk = lane >> 4;
    const int colb = ct0 * 16 + lr;                                // + 16 j
    const int rowb = 4 * kk;                                       // + 16 i + r: C layout of the 16 x 16 forms, column = lane % 16, row = 4 (lane / 16) + register
    // Every store issued so far (a stream-out fallback, the previous walk step) is waited for HERE, explicitly: on gfx9 loads and stores share vmcnt, and a
    // store that MAY be pending at the head of the K loop (the compiler merges the paths into it) turns the first wait of every step into vmcnt(0).
    // (lgkmcnt too: a FLAT access that may be pending - the compiler does not always prove a pointer out of the argument block global - forces both to 0)
    __builtin_amdgcn_s_waitcnt(0);
    // what the epilogue needs from memory is requested BEFORE the K loop (a round trip to memory is 2 - 3 thousand cycles; a lone wavefront would sit it out)
    float bv[NCT];
    w4 e[2][BWD ? NCT : 1];
    unsigned eoff[2][4];
    if (BWD) {
        if (E) {                                                    // rows past M / columns past N: clamped copies, never stored
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) eoff[i][r] = (unsigned)min(row0 + rowb + 16 * i + r, M - 1) * (unsigned)N;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NCT; ++j) bv[j] = bias ? ((const gfloat*)bias)[min(colb + 16 * min(j, nct - 1), N - 1)] : 0.0f;
    }
    // backward: the 8 saved outputs a tile's epilogue needs are requested in the LAST K step, where the tile has no next fragment to fetch (before the
    // loop they would be 56 more loads in flight than the 6-bit vmcnt can count)
    auto epi_load = [&](int j) __attribute__((always_inline)) {
        if (BWD) {
            if (E) {
                const unsigned c = (unsigned)min(colb + 16 * min(j, nct - 1), N - 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[i][BWD ? j : 0][r] = ((const gfloat*)E)[eoff[i][r] + c];
                }
            }
        }
    };
    w4 acc[2][NCT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = w4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NCT; ++j) asm volatile("" : "+v"(acc[i][j]));       // the zeroing instructions stay in front of ...
    }
    asm volatile("s_nop 7" ::: "memory");                          // ... the wait states before the first MFMA reads an accumulator as srcC
    walk_layer<NCT, BWD, SO>(Xs, px, W, BWD ? N : K, K, N, ct0, nct, so, acc, epi_load);
    walk_mfma_drain<NCT>(acc);
    WSTAMP(2 + 3 * stamp);
    if (BWD) {
        if (E) {
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float yv = e[i][j][r]; acc[i][j][r] *= (yv > 0.0f ? 1.0f : yv + 1.0f); }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r] + bv[j]; acc[i][j][r] = act ? elu_fast(v) : v; }
            }
        }
    }
    const bool rows_in = row0 + WALK_ROWS <= M;                    // uniform: every row of the block exists
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
        if (j < nct) {
            const int col = colb + 16 * j;
            const bool full = (ct0 + j) * 16 + 16 <= N;            // uniform: the tile lies wholly inside the matrix (all but possibly the last one)
            if (Ys) {                                              // next step's A operand; zero in the padding columns
                float* yp = Ys + rowb * py + col;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) yp[(16 * i + r) * py] = (full || col < N) ? acc[i][j][r] : 0.0f;
                }
            } else if (G) {                                        // last product of the walk: straight to memory
                gfloat* gp = (gfloat*)(G + (size_t)(row0 + rowb) * N + col);
                if (full && rows_in) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) gp[(16 * i + r) * N] = acc[i][j][r];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { if (col < N && row0 + rowb + 16 * i + r < M) gp[(16 * i + r) * N] = acc[i][j][r]; }
                    }
                }
            }
        }
    }
}

// this wavefront's share of a layer: all 32 rows and a quarter of the 16-column tiles (the first `nct_all % 4` wavefronts take one more); the tile count
// selects the instantiation with exactly that many accumulator columns
template <bool BWD>
__device__ __forceinline__ void walk_dispatch(const float* Xs, int px, const float* __restrict__ W, int K, int N, const float* __restrict__ bias, int act,
                                              const float* __restrict__ E, float* Ys, int py, float* __restrict__ G, int row0, int M, StreamOut& so,
                                              bool has_so, int stamp) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nct_all = (N + 15) >> 4, base = nct_all >> 2, rem = nct_all & 3;
    const int nct = base + (wave < rem ? 1 : 0), ct0 = wave * base + min(wave, rem);
    if (nct <= 0) { if (has_so) so.flush(); return; }
#define WALK_CASE(n_) do { if (has_so) walk_run<n_, BWD, true>(Xs, px, W, K, N, bias, act, E, Ys, py, G, row0, M, ct0, nct, so, stamp); \
                           else walk_run<n_, BWD, false>(Xs, px, W, K, N, bias, act, E, Ys, py, G, row0, M, ct0, nct, so, stamp); } while (0)
    switch (nct) {
        case 1: WALK_CASE(1); break;
        case 2: WALK_CASE(2); break;
        case 3: WALK_CASE(3); break;
        case 4: WALK_CASE(4); break;
        case 5: WALK_CASE(5); break;
        case 6: WALK_CASE(6); break;
        default: WALK_CASE(7); break;
    }
#undef WALK_CASE
}

// two workgroups of four wavefronts per CU: two wavefronts per SIMD, 256 registers each
template <bool BWD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_mlp_walk(const WalkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ni = (int)blockIdx.x / a.blocks_per_net;
    if (ni >= a.n_nets) return;
    const WalkNet& net = a.net[ni];
    const int row0 = ((int)blockIdx.x - ni * a.blocks_per_net) * WALK_ROWS;
    if (row0 >= a.M) return;
    float* P = lds;
    float* Q = lds + a.p_floats;
    WSTAMP(0);
    const int nl = net.nl;
    // ---- the row block's input -> P: the block is one contiguous piece of memory (rows past M: zero), read linearly, sixteen loads in flight per thread;
    // the padding columns D .. ceil16(D) are zeroed by the threads that own them ----
    {
        const int D = BWD ? net.dim[nl] : net.dim[0], D16 = (D + 15) & ~15, px = D16 + 4;
        const int rows = min(WALK_ROWS, a.M - row0), total = rows * D;
        const gfloat* src = (const gfloat*)(net.x + (size_t)row0 * D);
        const int dr = 256 / D, dc = 256 - dr * D;
        int row = (int)threadIdx.x / D, col = (int)threadIdx.x - row * D;
        for (int c = threadIdx.x; c < WALK_ROWS * D; c += 16 * 256) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[min(c + 256 * u, total - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (c + 256 * u < WALK_ROWS * D) P[row * px + col] = (c + 256 * u < total) ? v[u] : 0.0f;
                col += dc; row += dr;
                if (col >= D) { col -= D; row += 1; }
            }
        }
        const int pad = D16 - D;
        for (int e = threadIdx.x; e < WALK_ROWS * pad; e += 256) { const int r = e / pad; P[r * px + D + (e - r * pad)] = 0.0f; }
    }
    __syncthreads();
    WSTAMP(1);
    float* cur = P;
    float* nxt = Q;
    float* pend = nullptr;                                         // memory destination of what `cur` holds (the hidden output / dZ of the step before)
#pragma unroll 1
    for (int s = 0; s < nl; ++s) {
        // forward: layer l = s maps dim[l] -> dim[l + 1]; backward: step s undoes layer l = nl - 1 - s: K = dim[l + 1] -> N = dim[l]; the product of the
        // first layer's input gradient is not formed
        const int l = BWD ? nl - 1 - s : s;
        if (BWD && l == 0) break;
        const int K = BWD ? net.dim[l + 1] : net.dim[l], N = BWD ? net.dim[l] : net.dim[l + 1];
        const int px = pitch_of(K), py = pitch_of(N);
        const bool lastp = BWD ? (l == 1) : (l == nl - 1);
        float* Ys = lastp ? nullptr : nxt;
        StreamOut so;
        bool has_so = false;
        if (pend) {
            if ((K & 3) == 0 && (((uintptr_t)pend) & 15) == 0) {
                so.init(pend, cur, px, K, row0, a.M);
                has_so = true;
            } else {                                                // rows that are no multiple of 16 bytes: element by element, here and now
                const int rows = min(WALK_ROWS, a.M - row0), total = rows * K;
                gfloat* g0 = (gfloat*)(pend + (size_t)row0 * K);
                const int dr = 256 / K, dc = 256 - dr * K;
                int row = (int)threadIdx.x / K, col = (int)threadIdx.x - row * K;
                for (int c = threadIdx.x; c < total; c += 256) {
                    g0[c] = cur[row * px + col];
                    col += dc; row += dr;
                    if (col >= K) { col -= K; row += 1; }
                }
            }
        }
        const float* E = BWD ? (net.act[l - 1] ? net.yin[l - 1] : nullptr) : nullptr;
        float* G = BWD ? net.y[l - 1] : net.y[l];
        const int act = BWD ? 0 : net.act[l];
        const float* bias = BWD ? nullptr : net.b[l];
        walk_dispatch<BWD>(cur, px, net.W[l], K, N, bias, act, E, Ys, py, G, row0, a.M, so, has_so, s);
        WSTAMP(3 + 3 * s);
        if (!lastp) __syncthreads();
        WSTAMP(4 + 3 * s);
        pend = lastp ? nullptr : G;
        float* t = cur; cur = nxt; nxt = t;
    }
}

static int walk_prepare(WalkArgs& a, int M, bool bwd, size_t* lds_bytes) {
    // the two LDS buffers are used alternately: the input and every second product in P, the others in Q
    int p = 0, q = 0;
    for (int ni = 0; ni < a.n_nets; ++ni) {
        const WalkNet& n = a.net[ni];
        if (n.nl < 1 || n.nl > WALK_MAXL) return -1;
        for (int l = 0; l <= n.nl; ++l) if (n.dim[l] <= 0) return -1;
        for (int l = 0; l < n.nl; ++l) if (!n.W[l] && !(bwd && l == 0)) return -1;
        if (!n.x) return -1;
        for (int l = 1; l <= n.nl; ++l) if (n.dim[l] > 416) return -4;                 // 26 column tiles: 7 + 7 + 6 + 6 accumulator columns
        if (n.dim[0] > 416) return -4;
        int seq[WALK_MAXL + 1], cnt = 0;
        if (!bwd) { for (int l = 0; l < n.nl; ++l) seq[cnt++] = n.dim[l]; }            // operands that live in LDS: input and all but the last output
        else { for (int l = n.nl; l >= 2; --l) seq[cnt++] = n.dim[l]; if (n.nl == 1) seq[cnt++] = n.dim[1]; }
        for (int i = 0; i < cnt; ++i) {
            const int fl = WALK_ROWS * ((((seq[i] + 15) & ~15)) + 4);
            if (i & 1) { if (fl > q) q = fl; } else { if (fl > p) p = fl; }
        }
    }
    a.p_floats = p;
    a.M = M;
    a.blocks_per_net = (M + WALK_ROWS - 1) / WALK_ROWS;
    *lds_bytes = (size_t)(p + q) * sizeof(float);
    return *lds_bytes <= 80 * 1024 ? 0 : -4;                       // half a CU's LDS: two workgroups per CU
}

extern "C" {

static int walk_launch(const TfpMlp* nets, int32_t n_nets, int32_t M, bool bwd, void* stream) {
    if (!nets || n_nets < 1 || n_nets > 2 || M <= 0) return -1;
    WalkArgs a; memset(&a, 0, sizeof(a));
    a.n_nets = n_nets;
    for (int ni = 0; ni < n_nets; ++ni) {
        WalkNet& d = a.net[ni]; const TfpMlp& s = nets[ni];
        if (s.n_layers < 1 || s.n_layers > WALK_MAXL) return -1;
        d.x = s.x; d.nl = s.n_layers;
        for (int l = 0; l < WALK_MAXL; ++l) { d.W[l] = s.W[l]; d.b[l] = s.b[l]; d.yin[l] = s.yin[l]; d.y[l] = s.y[l]; d.act[l] = s.act[l]; }
        for (int l = 0; l <= WALK_MAXL; ++l) d.dim[l] = s.dim[l];
        if (!bwd && !s.y[s.n_layers - 1]) return -1;
        if (bwd) for (int l = 0; l + 1 < s.n_layers; ++l) { if (!s.y[l]) return -1; if (s.act[l] && !s.yin[l]) return -1; }
    }
    size_t lds = 0;
    const int rc = walk_prepare(a, M, bwd, &lds);
    if (rc) return rc;
    static bool attr_set[2] = {false, false};
    if (!attr_set[bwd ? 1 : 0]) {
        const void* f = bwd ? (const void*)k_mlp_walk<true> : (const void*)k_mlp_walk<false>;
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return -2;
        attr_set[bwd ? 1 : 0] = true;
    }
    const dim3 grid(a.blocks_per_net * n_nets), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (bwd) hipLaunchKernelGGL(k_mlp_walk<true>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(k_mlp_walk<false>, grid, block, lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// forward of n_nets <= 2 Linear / ELU stacks over the same M rows in ONE launch; -4: the shapes do not fit the walk (a layer wider than 416, LDS):
// the caller runs the layers one by one (tfp_linear_fwd[_group])
int tfp_mlp_forward(const TfpMlp* nets, int32_t n_nets, int32_t M, void* stream) { return walk_launch(nets, n_nets, M, false, stream); }
// the input-gradient chain of the same stacks in ONE launch: x = gradient of the network output, yin[l] = saved output of layer l, y[l] = dZ of layer l
// (l < n_layers - 1; the dZ of the last layer is x itself)
int tfp_mlp_backward(const TfpMlp* nets, int32_t n_nets, int32_t M, void* stream) { return walk_launch(nets, n_nets, M, true, stream); }

#ifdef WALK_TIMING
int tfp_walk_debug_read(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_t), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -3; }
#endif
}  // extern "C"
