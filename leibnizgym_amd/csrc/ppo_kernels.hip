// ppo_kernels.hip - hand-written gfx950 kernels of the in-repo PPO trainer (leibnizgym_amd/ppo.py), C ABI, raw device pointers.
//
// The minibatch step of the trainer is launch-bound: two small MLPs and, between them, ~60 elementwise / reduction launches of
// the PPO objective (negative log-likelihood, ratio, clipped surrogate, value loss, bounds loss, KL, and the same again
// backwards through autograd).  tfp_ppo_loss does the whole objective, forward AND backward, in one launch: one sample per
// lane, the gradients with respect to the policy mean, the value and the log-std come out directly, the scalar terms are
// reduced with DPP-free wave shuffles and one atomic per wave.
//
// Objective (RL-Games a2c_continuous with a central value network; reference resources/config/rlg/asymm.yaml):
//   nlp_i   = sum_a [ 0.5 ((x_ia - mu_ia) / sigma_a)^2 + log sigma_a + 0.5 log 2 pi ]
//   ratio_i = exp(old_nlp_i - nlp_i)
//   a_loss  = mean_i max(-adv_i ratio_i, -adv_i clamp(ratio_i, 1 - e, 1 + e))
//   c_loss  = mean_i (v_i - ret_i)^2
//   b_loss  = mean_i sum_a [ relu(mu_ia - 1.1)^2 + relu(-1.1 - mu_ia)^2 ]
//   ent     = sum_a (log sigma_a + 0.5 + 0.5 log 2 pi)
//   loss    = a_loss + v_coef c_loss - ent_coef ent + bounds_coef b_loss
//   kl      = mean_i sum_a 0.5 ((mu_ia - old_mu_ia) / sigma_a)^2          (statistic only)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <string.h>

#define MAX_A 18

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// d_logstd and the loss are sums over the blocks.  They are accumulated in accumulators of the library that are ZERO between launches and handed to the
// caller's buffers by the block that finishes last (a ticket), which also clears them again: the caller's outputs need no launch that zeroes them first.
// One objective at a time per device (the trainer's single stream); results as before up to the order of the atomic sums.
__device__ float g_loss_acc[24];
__device__ unsigned g_loss_ticket;
template <int A>
__global__ void __launch_bounds__(256) k_ppo_loss(const float* __restrict__ mu, const float* __restrict__ log_std, const float* __restrict__ act,
                                                  const float* __restrict__ old_nlp, const float* __restrict__ adv,
                                                  const float* __restrict__ old_mu, const float* __restrict__ v,
                                                  const float* __restrict__ ret, int B, float e_clip, float v_coef, float ent_coef,
                                                  float bounds_coef, float* __restrict__ d_mu, float* __restrict__ d_v,
                                                  float* __restrict__ d_logstd, float* __restrict__ loss_out, float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < B;
    const float invB = 1.0f / (float)B;
    float ls[A], inv_sig[A];
#pragma unroll
    for (int a = 0; a < A; ++a) { ls[a] = log_std[a]; inv_sig[a] = expf(-ls[a]); }
    float a_term = 0.0f, c_term = 0.0f, b_term = 0.0f, kl_term = 0.0f;
    float dls[A];
#pragma unroll
    for (int a = 0; a < A; ++a) dls[a] = 0.0f;
    if (valid) {
        float m[A], z[A], nlp = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            m[a] = mu[(size_t)i * A + a];
            z[a] = (act[(size_t)i * A + a] - m[a]) * inv_sig[a];
            nlp += 0.5f * z[a] * z[a] + ls[a] + 0.9189385332046727f;
            const float dk = (m[a] - old_mu[(size_t)i * A + a]) * inv_sig[a];
            kl_term += 0.5f * dk * dk;
        }
        const float ratio = expf(old_nlp[i] - nlp);
        const float ad = adv[i];
        const float s1 = -ad * ratio;
        const float rc = fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
        const float s2 = -ad * rc;
        a_term = fmaxf(s1, s2);
        // d a_term / d ratio: the unclipped branch when it is the larger one (or inside the clip range, where both agree)
        const bool inside = (ratio >= 1.0f - e_clip) && (ratio <= 1.0f + e_clip);
        const float g_ratio = (inside || s1 > s2) ? -ad : 0.0f;
        const float g_nlp = g_ratio * (-ratio) * invB;            // d loss / d nlp_i
        const float dv_ = v[i] - ret[i];
        c_term = dv_ * dv_;
        d_v[i] = v_coef * 2.0f * dv_ * invB;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const float hi = fmaxf(m[a] - 1.1f, 0.0f), lo = fmaxf(-1.1f - m[a], 0.0f);
            b_term += hi * hi + lo * lo;
            // d nlp / d mu = -(x - mu) / sigma^2 = -z / sigma ;  d nlp / d log sigma = 1 - z^2
            d_mu[(size_t)i * A + a] = g_nlp * (-z[a] * inv_sig[a]) + bounds_coef * invB * 2.0f * (hi - lo);
            dls[a] = g_nlp * (1.0f - z[a] * z[a]);
        }
    }
    // reductions: wave shuffles, then the four waves of the block through LDS, ONE atomic per block and quantity (atomics on
    // the same address serialise at ~100 ns each: 128 waves on 14 addresses cost 13 us, 32 blocks cost 3)
    __shared__ float red[4][A + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ent = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        ent += ls[a] + 1.4189385332046727f;
        const float s = wave_sum(dls[a]);
        if (lane == 0) red[wave][a] = s;
    }
    const float sa = wave_sum(a_term) * invB, sc = wave_sum(c_term) * invB, sb = wave_sum(b_term) * invB, sk = wave_sum(kl_term) * invB;
    if (lane == 0) { red[wave][A] = sa; red[wave][A + 1] = sc; red[wave][A + 2] = sb; red[wave][A + 3] = sk; }
    __syncthreads();
    if (threadIdx.x < A + 4) {
        const int q = threadIdx.x;
        const float t = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        if (q < A) {
            float add = t;
            if (blockIdx.x == 0) add += -ent_coef;               // the batch-independent entropy term, once
            if (add != 0.0f) atomicAdd(&g_loss_acc[q], add);
        } else if (q == A) {
            // thread A also assembles the block's share of the loss (it needs the three other sums of the block)
            const float bsa = t;
            const float bsc = (red[0][A + 1] + red[1][A + 1]) + (red[2][A + 1] + red[3][A + 1]);
            const float bsb = (red[0][A + 2] + red[1][A + 2]) + (red[2][A + 2] + red[3][A + 2]);
            float part = bsa + v_coef * bsc + bounds_coef * bsb;
            if (blockIdx.x == 0) part += -ent_coef * ent;
            atomicAdd(&g_loss_acc[A], part);
            atomicAdd(&stats[0], part);
            atomicAdd(&stats[1], bsa);
        } else if (q == A + 1) {
            atomicAdd(&stats[2], t);
        } else if (q == A + 3) {
            atomicAdd(&stats[3], t);
        }
    }
    // the last block to get here hands the sums over and restores the zero state
    __shared__ unsigned last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(&g_loss_ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (last) {
        __threadfence();
        if (threadIdx.x <= A) {
            const float sum = atomicExch(&g_loss_acc[threadIdx.x], 0.0f);
            if (threadIdx.x < A) d_logstd[threadIdx.x] = sum; else loss_out[0] = sum;
        }
        if (threadIdx.x == 0) g_loss_ticket = 0u;
    }
}

__global__ void k_reset_loss_state() {
    if (threadIdx.x < 24) g_loss_acc[threadIdx.x] = 0.0f;
    if (threadIdx.x == 0) g_loss_ticket = 0u;
}

// ---- gradient-norm truncation + Adam over FLAT buffers, two parameter groups (actor | central value network) ------------------
// torch's fused multi-tensor Adam and clip_grad_norm_ take ~110 us per step for these 32 small tensors (two 38 us launches for
// the two groups plus the norm / scale launches); over one flat buffer of 264 k floats the same arithmetic is two 5 us launches.
// Group 0 = elements [0, n0), group 1 = [n0, n1).  k_grad_sqnorms also advances the step counter (kernel boundary = ordering).
__global__ void __launch_bounds__(256) k_grad_sqnorms(const float* __restrict__ g, int n0, int n1, float* __restrict__ sq, float* __restrict__ step) {
    // step[1] is the count of COMPLETED steps (written by k_clip_adam, stable during this launch); this step is tn = step[1] + 1 and sums into the half
    // of sq that belongs to its parity - the half k_clip_adam of the step before cleared (no launch that zeroes sq)
    // The counter is a float (it feeds powf) and must keep ALTERNATING for ever - its parity selects the half of sq -, so it does not count past 2^23 + 1: from
    // there it swings between 2^23 (even) and 2^23 + 1 (odd), both exact in fp32 (at 2^24 a float + 1 rounds back onto itself and the parity would freeze: the
    // squared norms would then pile up in one half and the truncation coefficient fall towards 0).  The bias corrections 1 - beta^t are 1.0f long before that.
    const float tp = step[1];
    const float tn = tp >= 8388609.0f ? 8388608.0f : tp + 1.0f;
    const int par = (int)tn & 1;
    float a0 = 0.0f, a1 = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += gridDim.x * blockDim.x) {
        const float x = g[i];
        if (i < n0) a0 += x * x; else a1 += x * x;
    }
    __shared__ float red[4][2];
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int q = threadIdx.x;
        const float t = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        if (t != 0.0f) atomicAdd(&sq[2 * par + q], t);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) step[0] = tn;
}
// torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (||g|| + 1e-6)) per group; then torch.optim.Adam (no weight decay, no
// amsgrad): m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void __launch_bounds__(256) k_clip_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int n0, int n1, float* __restrict__ sq,
                                                   float* __restrict__ step, const float* __restrict__ lr, float max0, float max1,
                                                   float b1, float b2, float eps) {
    const float t = step[0];                                    // this step (written by k_grad_sqnorms, stable during this launch)
    const int par = (int)t & 1;
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float c0 = fminf(max0 / (sqrtf(sq[2 * par]) + 1e-6f), 1.0f), c1 = fminf(max1 / (sqrtf(sq[2 * par + 1]) + 1e-6f), 1.0f);
    if (blockIdx.x == 0 && threadIdx.x < 2) sq[2 * (1 - par) + threadIdx.x] = 0.0f;      // the other half: the next step sums into it (nobody touches it now)
    if (blockIdx.x == 0 && threadIdx.x == 0) step[1] = t;
    const float s0 = lr[0] / bc1, s1 = lr[1] / bc1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += gridDim.x * blockDim.x) {
        const bool g0 = i < n0;
        const float gi = g[i] * (g0 ? c0 : c1);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= (g0 ? s0 : s1) * mi / (sqrtf(vi) / bc2s + eps);
    }
}

extern "C" {

int tfp_api_version(void) { return 3; }

// d_logstd [A] and loss_out [1] are overwritten; stats [4] (loss, a_loss, c_loss, kl) ACCUMULATE across calls.
}  // extern "C"
// (No launch of this file zeroes a small buffer any more - the objective and the optimiser keep their accumulators clean themselves.  When one did, it was a
// kernel and not hipMemsetAsync: inside a captured HIP graph the memset / memcpy NODES of this ROCm stack proved unsafe, docs/HISTORY.md section 8.)
extern "C" {
int tfp_ppo_loss(const float* mu, const float* log_std, const float* act, const float* old_nlp, const float* adv, const float* old_mu,
                 const float* v, const float* ret, int32_t B, int32_t A, float e_clip, float v_coef, float ent_coef, float bounds_coef,
                 float* d_mu, float* d_v, float* d_logstd, float* loss_out, float* stats, void* stream) {
    if (B <= 0 || (A != 9 && A != 18)) return -1;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((B + 255) / 256), block(256);
    if (A == 9)
        hipLaunchKernelGGL((k_ppo_loss<9>), grid, block, 0, s, mu, log_std, act, old_nlp, adv, old_mu, v, ret, B, e_clip, v_coef, ent_coef,
                           bounds_coef, d_mu, d_v, d_logstd, loss_out, stats);
    else
        hipLaunchKernelGGL((k_ppo_loss<18>), grid, block, 0, s, mu, log_std, act, old_nlp, adv, old_mu, v, ret, B, e_clip, v_coef, ent_coef,
                           bounds_coef, d_mu, d_v, d_logstd, loss_out, stats);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// the zero state of tfp_ppo_loss's accumulators (after a launch that failed or was aborted; a completed launch leaves them clean itself)
int tfp_reset_state(void* stream) {
    hipLaunchKernelGGL(k_reset_loss_state, dim3(1), dim3(64), 0, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// One optimisation step over flat buffers: sq [4] is scratch (all zero before the first step; the launches keep the half of the next step zero), step [2]
// = (this step, completed steps) and lr [2] live on the device (graph capture).
int tfp_clip_adam(float* p, const float* g, float* m, float* v, int32_t n0, int32_t n1, float* sq, float* step, const float* lr,
                  float max_norm0, float max_norm1, float beta1, float beta2, float eps, void* stream) {
    if (n1 <= 0 || n0 < 0 || n0 > n1) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (n1 + 256 * 4 - 1) / (256 * 4);
    hipLaunchKernelGGL(k_grad_sqnorms, dim3(blocks), dim3(256), 0, s, g, n0, n1, sq, step);
    hipLaunchKernelGGL(k_clip_adam, dim3(blocks), dim3(256), 0, s, p, g, m, v, n0, n1, sq, step, lr, max_norm0, max_norm1, beta1, beta2, eps);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"

// =====================================================================================================================
// fp32 MFMA GEMMs for the two small MLPs (41/113 -> 400 -> 200 -> 100 -> 9/1, batch 8192)
// =====================================================================================================================
// rocBLAS picks 256x128 macro tiles for these shapes: [8192 x 400] x [400 x 200] becomes 64 workgroups on a 256-CU chip (19 TF/s),
// and the activation derivative / bias gradient are launches and passes over memory of their own.  Here every product is
//     C[i, j] = sum_k opA(i, k) * opB(j, k)
// on 64 x 64 block tiles (4 wavefronts, one 32 x 32 v_mfma_f32_32x32x2_f32 accumulator each: exact fp32, an fmaf chain), K tiles of
// 32.  Both operands sit in LDS as [row][k] with a pitch of 36 floats: a lane reads FOUR consecutive k of its row with one
// ds_read_b128 (lanes 0-31: k0..k0+3, lanes 32-63: k0+4..k0+7) and feeds them to four MFMAs - the order in which the k of a tile are
// summed is free as long as both operands use the same one - so a K tile costs 8 LDS reads per 16 MFMAs.  The pitch makes any 16 rows
// that differ mod 16 hit disjoint bank windows (36 r mod 64 = 4 (9 r mod 16)), which covers the lane groups of ds_read_b128 and of
// ds_write_b128.  An operand is staged from memory in one of two ways:
//   k-contiguous (x, dY, W of the forward): lane -> (row 16 w + (l & 15), k (l >> 4) * 8 .. +7): two dwordx4 loads (K % 4 == 0)
//   k-major      (W of dX, both operands of dW): lane -> (column t & 63, k (t >> 6) * 8 .. +7): eight coalesced dword loads
// and written with two ds_write_b128; the next K tile is prefetched into registers while the current one is multiplied.
//   forward   y  = act(x W^T + b):        A = x  [M, K] k-contiguous, B = W [N, K] k-contiguous; bias and ELU fused into the store
//   dX        dx = dZ W:                  A = dY [M, K] k-contiguous (dZ = dY * elu'(Y) formed in the load), B = W [K, N] k-major
//   dW, db    [dW | db] = dZ^T [x | 1]:   A = dY [rows, N1] k-major (same fusion), B = x [rows, N2] k-major with a column of ones
//                                         appended; split over row chunks (blockIdx.z), the partial products summed by k_sum_partials
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GT 64            // block tile (rows and columns)
#define GK 32            // K tile
#define GP 36            // LDS row pitch in floats

__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }   // elu'(z) in terms of y = elu(z)

// this thread's 8 values of the K tile starting at k0: operand P (leading dimension ld, `rows` valid rows / columns, first one r0)
//   KMAJ: P[k * ld + r]; else P[r * ld + k];  VEC (k-contiguous only): ld % 4 == 0, kend % 4 == 0 -> dwordx4
//   ONES (k-major only): column `rows - 1` is a column of ones that P does not hold (ld = rows - 1)
// Every load is unconditional (addresses clamped into the matrix, the value discarded by stage_fix): the loads of GD tiles stay in
// flight behind one another and the compiler's vmcnt bookkeeping stays exact.
template <bool KMAJ, bool VEC, bool ONES>
__device__ __forceinline__ void stage_load(const float* __restrict__ P, int ld, int rows, int r0, int k0, int kend, int tid, float (&v)[8]) {
    if (KMAJ) {
        const int real = ONES ? rows - 1 : rows;
        const int r = min(r0 + (tid & 63), real - 1), kb = k0 + (tid >> 6) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = P[(size_t)min(kb + c, kend - 1) * ld + r];
    } else {
        if (VEC) {                                                  // lane -> (rows t >> 3 and (t >> 3) + 32, k (t & 7) * 4 .. +3): 8 lanes read
#pragma unroll                                                      // one whole 128-byte line of a row - a wave-load touches 8 lines, all of each
            for (int q = 0; q < 2; ++q) {
                const int r = min(r0 + (tid >> 3) + 32 * q, rows - 1);
                const f32x4 x = *(const f32x4*)(P + (size_t)r * ld + min(k0 + (tid & 7) * 4, kend - 4));
#pragma unroll
                for (int c = 0; c < 4; ++c) v[4 * q + c] = x[c];
            }
        } else {                                                    // lane -> (k = t & 31, rows (t >> 5) + 8 c): 128 contiguous bytes per half-wave
            const int k = min(k0 + (tid & 31), kend - 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = P[(size_t)min(r0 + (tid >> 5) + 8 * c, rows - 1) * ld + k];
        }
    }
}

// what the clamped loads fetched outside the matrix becomes zero (or the column of ones); DZ: v *= elu'(y)
template <bool KMAJ, bool VEC, bool DZ, bool ONES>
__device__ __forceinline__ void stage_fix(int rows, int r0, int k0, int kend, int tid, float (&v)[8], const float (&y)[8]) {
    if (!KMAJ && !VEC) {
        const bool ink = k0 + (tid & 31) < kend;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float x = v[c];
            if (DZ) x *= elu_grad_from_out(y[c]);
            v[c] = (ink && r0 + (tid >> 5) + 8 * c < rows) ? x : 0.0f;
        }
        return;
    }
    if (!KMAJ) {                                                    // VEC: value 4 q + c belongs to row (t >> 3) + 32 q, k (t & 7) * 4 + c
        const int kb = k0 + (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool inr = r0 + (tid >> 3) + 32 * q < rows;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = v[4 * q + c];
                if (DZ) x *= elu_grad_from_out(y[4 * q + c]);
                v[4 * q + c] = (inr && kb + c < kend) ? x : 0.0f;
            }
        }
        return;
    }
    const int real = ONES ? rows - 1 : rows;
    const int r = r0 + (tid & 63);
    const int kb = k0 + (tid >> 6) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x = v[c];
        if (DZ) x *= elu_grad_from_out(y[c]);
        const bool ink = kb + c < kend;
        x = (ink && r < real) ? x : ((ONES && ink && r == real) ? 1.0f : 0.0f);
        v[c] = x;
    }
}

template <bool KMAJ, bool VEC>
__device__ __forceinline__ void stage_store(float* S, int tid, const float (&v)[8]) {
    if (!KMAJ && !VEC) {
#pragma unroll
        for (int c = 0; c < 8; ++c) S[((tid >> 5) + 8 * c) * GP + (tid & 31)] = v[c];
        return;
    }
    if (!KMAJ) {                                                    // VEC: 8 consecutive lanes write the 128 contiguous bytes of one LDS row
        *(f32x4*)(S + (tid >> 3) * GP + (tid & 7) * 4) = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(S + ((tid >> 3) + 32) * GP + (tid & 7) * 4) = f32x4{v[4], v[5], v[6], v[7]};
        return;
    }
    f32x4* d = (f32x4*)(S + (tid & 63) * GP + (tid >> 6) * 8);
    d[0] = f32x4{v[0], v[1], v[2], v[3]};
    d[1] = f32x4{v[4], v[5], v[6], v[7]};
}

#ifndef GEMM_DBG
#define GEMM_DBG 0        // developer builds (timing only, wrong results): 1 = no MFMAs and no LDS reads, 2 = no global loads inside the loop, 4 = no LDS reads,
                          // 5 = no global loads and no LDS writes inside the loop: the staging wavefronts only keep the barriers (tools/gemm_decompose.sh)
#endif
#ifndef GEMM_SCHED
#define GEMM_SCHED 1
#endif
#ifndef GEMM_PRIO
#define GEMM_PRIO 3      // issue priority of the producer wavefronts (the consumers' MFMAs fill what is left)
#endif
#ifndef GD
#define GD 3             // K tiles in flight per workgroup (register ring of the producer wavefronts)
#endif
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, F, I + 1>(static_cast<F&&>(f)); }
}

// ACT: 0 none, 1 ELU (with bias; forward only), -1 no bias.  SPLIT: the K range is cut into chunks of `chunk`, blockIdx.z takes one and
// writes its own [M, N] slab of C.
#ifdef GEMM_TIMING
__device__ unsigned g_dbg[8 * 8];
__device__ unsigned long long g_wg[2048 * 2];     // [workgroup][start, end] in s_memrealtime ticks (100 MHz)                  // developer build: [wave][0 total, 1 barrier wait, 2 work] cycles of workgroup 0's waves
#define TNOW() ((unsigned)__builtin_readcyclecounter())
#define TBAR() do { const unsigned t0_ = TNOW(); __syncthreads(); t_bar += TNOW() - t0_; } while (0)
#else
#define TNOW() 0u
#define TBAR() __syncthreads()
#endif
#ifndef GEMM_WPE
#define GEMM_WPE 4       // wavefronts per SIMD the register allocation must leave room for: two workgroups of 8 wavefronts per CU (the k-contiguous scalar-load
                         // instantiations - first layers, K = 41 / 113 - took 146 registers and ran ONE workgroup per CU: 20.7 -> 16.0 us, 29.2 -> 24.0 us;
                         // under the hint exactly these two forward instantiations spill 6 registers (28 B of scratch per lane, `make resource-usage-ppo`) and
                         // are still the faster form; every other instantiation is below 128 registers by itself and is not affected)
#endif
// The body of one 64 x 64 output tile: workgroup (bx0, by0, bz0) of an (nx, ny, nz) grid of ONE product.  k_gemm launches a product on its own grid;
// k_gemm_group packs the grids of up to 8 independent products of the same kind (the two networks' layers; the eight weight gradients of a
// minibatch step) into one launch: these products are 5 - 25 us each, and a launch costs its dispatch, its cold first loads and its tail, all of
// which the next product's workgroups now fill.
template <bool AK, bool BK, bool AVEC, bool BVEC, int ACT, bool DZ, bool ONES, bool SPLIT>
__device__ __forceinline__ void gemm_tile(float (&S)[2][2][GT * GP], const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                          const float* __restrict__ Y, float* __restrict__ C, int M, int N, int K, int lda, int ldb, int chunk,
                                          int bx0, int by0, int bz0, int nx, int ny, int nz, const float* __restrict__ E = nullptr) {
    // Role-specialised wavefronts: waves 0-3 multiply (one 32 x 32 accumulator each: LDS reads and MFMAs, nothing else), waves 4-7 stage
    // (global loads, the fix-ups, LDS writes).  A SIMD hosts one of each per workgroup, so the staging instructions of the producers issue
    // in the shadow of the consumers' MFMAs instead of between them; tile t is multiplied out of LDS buffer t & 1 while tile t + 1 is
    // written into the other one, one workgroup barrier per tile.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;                               // index inside the role
    const int wr = (wave & 3) >> 1, wc = wave & 1;
    // Workgroups are handed to the 8 XCDs round-robin in launch order; each XCD has its own L2.  The workgroups that share operand rows
    // (the column blocks of one row block; for the split form: every tile of one row chunk) are renumbered onto ONE XCD.
    int bx = bx0, by = by0, bz = bz0;
    {
        const int L = bx + nx * (by + ny * bz), xcd = L & 7, idx = L >> 3;
        if (SPLIT) {
            if ((nz & 7) == 0) { const int w = nx * ny, in = idx % w; bz = (idx / w) * 8 + xcd; by = in / nx; bx = in - by * nx; }
        } else if ((ny & 7) == 0) { by = (idx / nx) * 8 + xcd; bx = idx % nx; }
    }
    const int r0 = by * GT, c0 = bx * GT;
    int kbeg = 0, kend = K;
    if (SPLIT) { kbeg = bz * chunk; kend = min(kbeg + chunk, K); C += (size_t)bz * (size_t)M * (size_t)N; }
    const int ntiles = (kend - kbeg + GK - 1) / GK;
    unsigned t_bar = 0u; const unsigned t_start = TNOW(); (void)t_bar; (void)t_start;
#ifdef GEMM_TIMING
    const int wg_lin = bx0 + nx * (by0 + ny * bz0);
    if (threadIdx.x == 0 && wg_lin < 2048) g_wg[2 * wg_lin] = __builtin_amdgcn_s_memrealtime();
#endif
    if (producer) {
        __builtin_amdgcn_s_setprio(GEMM_PRIO);                      // staging instructions go first whenever they are ready: they are few, the
        // wave-uniform: can a tile of this workgroup hold values that stage_fix must replace?  The loads are clamped into the matrix, so an interior
        // tile (all 64 rows / columns inside, the K tile below kend) is staged as loaded: a staging wavefront's vector instructions do not overlap
        // with the MFMAs of the multiplying wavefront on its SIMD (tools/microbench/mfma_valu_overlap.hip)
        const bool a_edge = r0 + GT > M, b_edge = c0 + GT > (ONES ? N - 1 : N);
        float ra[GD][8], rb[GD][8], ry[DZ ? GD : 1][8];
#pragma unroll
        for (int d = 0; d < GD; ++d) {                              // tiles past the end are re-reads of the last one, never used
            const int k0 = kbeg + min(d, ntiles - 1) * GK;
            stage_load<AK, AVEC, false>(A, lda, M, r0, k0, kend, tid, ra[d]);
            if (DZ) stage_load<AK, AVEC, false>(Y, lda, M, r0, k0, kend, tid, ry[DZ ? d : 0]);
            stage_load<BK, BVEC, ONES>(B, ldb, N, c0, k0, kend, tid, rb[d]);
            __builtin_amdgcn_sched_barrier(0);                      // keep the slots' loads in slot order: vmcnt waits count on it
        }
        // stage(slot d, tile t): registers -> LDS buffer t & 1, then the slot takes tile t + GD (in flight through GD multiplications)
        auto stage = [&](auto dc, int t) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
            const int k0 = kbeg + t * GK;
            const bool k_edge = k0 + GK > kend;
            if (DZ || a_edge || k_edge) stage_fix<AK, AVEC, DZ, false>(M, r0, k0, kend, tid, ra[d], ry[DZ ? d : 0]);
            if (b_edge || k_edge) stage_fix<BK, BVEC, false, ONES>(N, c0, k0, kend, tid, rb[d], rb[d]);
            if (GEMM_DBG != 5) {
                stage_store<AK, AVEC>(S[t & 1][0], tid, ra[d]);
                stage_store<BK, BVEC>(S[t & 1][1], tid, rb[d]);
            }
            const int kn = kbeg + min(t + GD, ntiles - 1) * GK;
            if (GEMM_DBG != 2 && GEMM_DBG != 5) {
                stage_load<AK, AVEC, false>(A, lda, M, r0, kn, kend, tid, ra[d]);
                if (DZ) stage_load<AK, AVEC, false>(Y, lda, M, r0, kn, kend, tid, ry[DZ ? d : 0]);
                stage_load<BK, BVEC, ONES>(B, ldb, N, c0, GEMM_DBG == 3 ? kbeg : kn, kend, tid, rb[d]);   // (3: developer build, B re-read from L1)
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        stage(std::integral_constant<int, 0>{}, 0);
        TBAR();                                            // tile 0 staged
        // iteration t (consumers multiply tile t): stage tile t + 1 from slot (t + 1) % GD, then the barrier that ends the iteration
        int t = 0;
        for (; t + GD <= ntiles; t += GD) {                         // branch-free body, GD iterations per trip: a tile past the end stages zeros
            static_for<GD>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                stage(std::integral_constant<int, (i + 1) % GD>{}, t + 1 + i);
                TBAR();
            });
        }
        static_for<GD - 1>([&](auto ic) __attribute__((always_inline)) {   // the remaining ntiles % GD iterations
            constexpr int i = decltype(ic)::value;
            if (t + i < ntiles) {
                stage(std::integral_constant<int, (i + 1) % GD>{}, t + 1 + i);
                TBAR();
            }
        });
#ifdef GEMM_TIMING
        if (blockIdx.x == 1 && blockIdx.y == 8 && lane == 0) { g_dbg[wave * 8] = TNOW() - t_start; g_dbg[wave * 8 + 1] = t_bar; }
#endif
        // the consumers' epilogue has one more workgroup barrier (tile through LDS): the producers arrive at it too before they leave,
        // so that every wavefront of the workgroup executes the same number of barriers (no reliance on how the hardware treats
        // wavefronts that have already ended)
        if (!SPLIT && (N & 3) == 0) __syncthreads();
        return;
    }
    // ---- consumers ----
    const bool live = r0 + wr * 32 < M && c0 + wc * 32 < N;       // wave-uniform: a tile wholly outside the matrix is not multiplied
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int aoff = (wr * 32 + (lane & 31)) * GP + 4 * (lane >> 5), boff = (wc * 32 + (lane & 31)) * GP + 4 * (lane >> 5);
    TBAR();                                                // tile 0 staged
    for (int t = 0; t < ntiles; ++t) {
        if (GEMM_DBG != 1 && live) {
            f32x4 a[GK / 8], b[GK / 8];
#pragma unroll
            for (int ks = 0; ks < GK / 8; ++ks) {
                if (GEMM_DBG == 4) { a[ks] = f32x4{1.0f, 2.0f, 3.0f, (float)t}; b[ks] = f32x4{0.5f, (float)lane, 1.5f, 2.5f}; continue; }   // developer build: no LDS reads
                a[ks] = *(const f32x4*)(S[t & 1][0] + aoff + 8 * ks);
                b[ks] = *(const f32x4*)(S[t & 1][1] + boff + 8 * ks);
            }
#pragma unroll
            for (int ks = 0; ks < GK / 8; ++ks) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][s], b[ks][s], acc, 0, 0, 0);
            }
        }
        TBAR();
    }
#ifdef GEMM_TIMING
    if (blockIdx.x == 1 && blockIdx.y == 8 && lane == 0) { g_dbg[wave * 8] = TNOW() - t_start; g_dbg[wave * 8 + 1] = t_bar; g_dbg[wave * 8 + 2] = (unsigned)ntiles; }
#endif
    // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = c0 + wc * 32 + (lane & 31);
    const float bv = (ACT >= 0 && bias && col < N) ? bias[col] : 0.0f;
    if (!SPLIT && (N & 3) == 0) {
        // The 64 x 64 tile leaves through LDS as dwordx4 stores (a wave instruction = 4 rows x 256 contiguous bytes) instead of 16 dword
        // stores per lane of 128-byte pieces: the tail of these short kernels is bound by store ISSUE (the producers have staged their last tile; the
        // K loop's last barrier lies behind every LDS read, so the staging buffers are free).  N % 4 == 0 keeps the rows 16-byte aligned.
        float* T = &S[0][0][0];                                     // [64][GTP]: 17 KB of the 36 KB
        constexpr int GTP = GT + 4;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            float v = acc[reg] + bv;
            if (ACT == 1) v = v > 0.0f ? v : expm1f(v);              // ELU, alpha = 1
            T[(wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * GTP + wc * 32 + (lane & 31)] = live ? v : 0.0f;
        }
        __syncthreads();                                            // all eight wavefronts (the producers arrive from their own exit path)
        const int t = threadIdx.x;                                  // 0 .. 255
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = (t >> 4) + 16 * it, cc = (t & 15) * 4;
            if (r0 + rr < M && c0 + cc < N) {
                f32x4 v = *(const f32x4*)(T + rr * GTP + cc);
                if (ACT < 0 && E) {                                 // (input-gradient kinds only) the product leaves as dZ of the layer below: times elu'(its output)
                    const f32x4 y = *(const f32x4*)(E + (size_t)(r0 + rr) * N + c0 + cc);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] *= elu_grad_from_out(y[c]);
                }
                *(f32x4*)(C + (size_t)(r0 + rr) * N + c0 + cc) = v;
            }
        }
    } else if (live && col < N) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (row < M) {
                float v = acc[reg] + bv;
                if (ACT == 1) v = v > 0.0f ? v : expm1f(v);          // ELU, alpha = 1
                if (ACT < 0 && E) v *= elu_grad_from_out(E[(size_t)row * N + col]);
                C[(size_t)row * N + col] = v;
            }
        }
    }
#ifdef GEMM_TIMING
    if (threadIdx.x == 0 && wg_lin < 2048) g_wg[2 * wg_lin + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// The vector-load forward kinds need 63 - 67 registers: held to 64, four workgroups of 8 wavefronts are resident per CU (LDS: 4 x 36 KB) and the 1024
// workgroups of a grouped pair of 8192 x 400 -> 200 layers run as ONE round; at 67 registers three are resident and the fourth quarter of the grid runs as
// a second round behind the first: 44 -> 54 us for that launch (tools/experiments/grp_bench.py).
#define GEMM_WPE_OF(AVEC_, BVEC_, DZ_) (((AVEC_) && (BVEC_) && !(DZ_)) ? 8 : GEMM_WPE)     // (the other kinds below 70 registers held to 64 as well: no measurable change)
template <bool AK, bool BK, bool AVEC, bool BVEC, int ACT, bool DZ, bool ONES, bool SPLIT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(GEMM_WPE_OF(AVEC, BVEC, DZ)))) k_gemm(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                              const float* __restrict__ Y, float* __restrict__ C, int M, int N, int K, int lda, int ldb, int chunk,
                                              const float* __restrict__ E) {
    __shared__ __attribute__((aligned(16))) float S[2][2][GT * GP];   // [buffer][operand]
    gemm_tile<AK, BK, AVEC, BVEC, ACT, DZ, ONES, SPLIT>(S, A, B, bias, Y, C, M, N, K, lda, ldb, chunk, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z,
                                                         (int)gridDim.x, (int)gridDim.y, (int)gridDim.z, SPLIT ? nullptr : E);
}

// up to 8 products of one kind in one launch: workgroup L of the launch belongs to product p = the last one with first[p] <= L; every product's share is
// padded to a multiple of 8 workgroups so that its tiles meet the XCDs as they do in a launch of their own (the padding workgroups leave at once)
struct GemmGroup {
    const float* A[8]; const float* B[8]; const float* bias[8]; const float* Y[8]; const float* E[8]; float* C[8];
    int M[8], N[8], K[8], lda[8], ldb[8], chunk[8], nx[8], ny[8], nz[8], first[9], n;
};
template <bool AK, bool BK, bool AVEC, bool BVEC, int ACT, bool DZ, bool ONES, bool SPLIT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(GEMM_WPE_OF(AVEC, BVEC, DZ)))) k_gemm_group(const GemmGroup g) {
    __shared__ __attribute__((aligned(16))) float S[2][2][GT * GP];
    const int L = (int)blockIdx.x;
    int p = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) p = (j < g.n && L >= g.first[j]) ? j : p;
    const int l = L - g.first[p], nx = g.nx[p], ny = g.ny[p], nz = g.nz[p];
    if (l >= nx * ny * nz) return;                                   // padding (uniform over the workgroup)
    const int bx = l % nx, t = l / nx, by = t % ny, bz = t / ny;
    gemm_tile<AK, BK, AVEC, BVEC, ACT, DZ, ONES, SPLIT>(S, g.A[p], g.B[p], g.bias[p], g.Y[p], g.C[p], g.M[p], g.N[p], g.K[p], g.lda[p], g.ldb[p], g.chunk[p],
                                                         bx, by, bz, nx, ny, nz, SPLIT ? nullptr : g.E[p]);
}
// host side: append product (grid nx x ny x nz) to the group
static void group_add(GemmGroup& g, int& total, const float* A, const float* B, const float* bias, const float* Y, float* C, int M, int N, int K, int lda, int ldb,
                      int chunk, int nx, int ny, int nz, const float* E = nullptr) {
    const int p = g.n;
    g.E[p] = E;
    g.A[p] = A; g.B[p] = B; g.bias[p] = bias; g.Y[p] = Y; g.C[p] = C; g.M[p] = M; g.N[p] = N; g.K[p] = K; g.lda[p] = lda; g.ldb[p] = ldb; g.chunk[p] = chunk;
    g.nx[p] = nx; g.ny[p] = ny; g.nz[p] = nz; g.first[p] = total;
    total += (nx * ny * nz + 7) & ~7;
    g.n = p + 1;
}

// gw[N1, N2] and gb[N1] from `splits` slabs of [N1, N2 + 1] (fixed summation order: deterministic)
__global__ void __launch_bounds__(256) k_sum_partials(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb, int splits,
                                                      int N1, int N2) {
    const int i = blockIdx.x * 256 + threadIdx.x, tot = N1 * (N2 + 1);
    if (i >= tot) return;
    float s = 0.0f;
#pragma unroll 8
    for (int z = 0; z < splits; ++z) s += part[(size_t)z * tot + i];
    const int r = i / (N2 + 1), c = i - r * (N2 + 1);
    if (c < N2) gw[(size_t)r * N2 + c] = s; else gb[r] = s;
}

extern "C" {
// C[M,N] = act(A[M,K] . W[N,K]^T + bias); act: 0 none, 1 ELU
int tfp_linear_fwd(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT), block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (K & 3) == 0 && (((uintptr_t)A | (uintptr_t)W) & 15) == 0;
#define FWD(V, ACT_) hipLaunchKernelGGL((k_gemm<false, false, V, V, ACT_, false, false, false>), grid, block, 0, s, A, W, bias, nullptr, C, M, N, K, K, K, 0, nullptr)
    if (vec) { if (act) FWD(true, 1); else FWD(true, 0); }
    else { if (act) FWD(false, 1); else FWD(false, 0); }
#undef FWD
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// C[M,N] = (dZ[M,K] . B[K,N]) * elu'(Yout[M,N]) with dZ = A (Y == NULL) or A * elu'(Y); Yout == NULL: no factor
int tfp_gemm_nn_dz(const float* A, const float* Y, const float* B, const float* Yout, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    if (Yout && (N & 3) == 0 && ((uintptr_t)Yout & 15) != 0) return -1;      // the vector epilogue reads Yout as it writes C
    dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT), block(512);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (K & 3) == 0 && (((uintptr_t)A | (uintptr_t)Y) & 15) == 0;
#define NN(V, DZ_) hipLaunchKernelGGL((k_gemm<false, true, V, false, -1, DZ_, false, false>), grid, block, 0, s, A, B, nullptr, Y, C, M, N, K, K, N, 0, Yout)
    if (vec) { if (Y) NN(true, true); else NN(true, false); }
    else { if (Y) NN(false, true); else NN(false, false); }
#undef NN
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int tfp_gemm_nn(const float* A, const float* Y, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
    return tfp_gemm_nn_dz(A, Y, B, nullptr, C, M, N, K, stream);
}
// gw[N1, N2] = dZ^T B, gb[N1] = column sums of dZ, for dZ[rows, N1] = A or A * elu'(Y), B[rows, N2]; `part` is scratch for
// ceil(rows / chunk) slabs of [N1, N2 + 1] floats
int tfp_gemm_tn_bias(const float* A, const float* Y, const float* B, float* part, float* gw, float* gb, int32_t rows, int32_t N1, int32_t N2,
                     int32_t chunk, void* stream) {
    if (rows <= 0 || N1 <= 0 || N2 <= 0 || chunk <= 0 || (chunk % GK) != 0) return -1;
    const int splits = (rows + chunk - 1) / chunk;
    dim3 grid((N2 + 1 + GT - 1) / GT, (N1 + GT - 1) / GT, splits), block(512);
    hipStream_t s = (hipStream_t)stream;
    if (Y) hipLaunchKernelGGL((k_gemm<true, true, false, false, -1, true, true, true>), grid, block, 0, s, A, B, nullptr, Y, part, N1, N2 + 1, rows, N1, N2, chunk, nullptr);
    else hipLaunchKernelGGL((k_gemm<true, true, false, false, -1, false, true, true>), grid, block, 0, s, A, B, nullptr, nullptr, part, N1, N2 + 1, rows, N1, N2, chunk, nullptr);
    const int tot = N1 * (N2 + 1);
    hipLaunchKernelGGL(k_sum_partials, dim3((tot + 255) / 256), dim3(256), 0, s, part, gw, gb, splits, N1, N2);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// ---- the same three products for n <= 8 independent problems in ONE launch (host arrays of n device pointers / sizes).  All problems of a call must be of
// one kind - the same alignment class (K % 4 == 0 and 16-byte aligned operands, or not), the same activation, Y given for all or for none: -4 otherwise
// (the caller then launches them one by one) ----
int tfp_linear_fwd_group(const void* const* A, const void* const* W, const void* const* bias, void* const* C, const int32_t* M, const int32_t* N, const int32_t* K,
                         int32_t act, int32_t n, void* stream) {
    if (n <= 0 || n > 8) return -1;
    GemmGroup g; memset(&g, 0, sizeof(g));
    int total = 0, nvec = 0;
    for (int p = 0; p < n; ++p) {
        if (M[p] <= 0 || N[p] <= 0 || K[p] <= 0) return -1;
        nvec += ((K[p] & 3) == 0 && (((uintptr_t)A[p] | (uintptr_t)W[p]) & 15) == 0) ? 1 : 0;
        group_add(g, total, (const float*)A[p], (const float*)W[p], (const float*)bias[p], nullptr, (float*)C[p], M[p], N[p], K[p], K[p], K[p], 0,
                  (N[p] + GT - 1) / GT, (M[p] + GT - 1) / GT, 1);
    }
    for (int p = n; p < 9; ++p) g.first[p] = total;
    if (nvec != 0 && nvec != n) return -4;
    hipStream_t s = (hipStream_t)stream;
#define FWDG(V, ACT_) hipLaunchKernelGGL((k_gemm_group<false, false, V, V, ACT_, false, false, false>), dim3(total), dim3(512), 0, s, g)
    if (nvec) { if (act) FWDG(true, 1); else FWDG(true, 0); }
    else { if (act) FWDG(false, 1); else FWDG(false, 0); }
#undef FWDG
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int tfp_gemm_nn_dz_group(const void* const* A, const void* const* Y, const void* const* B, const void* const* Yout, void* const* C, const int32_t* M,
                         const int32_t* N, const int32_t* K, int32_t n, void* stream) {
    if (n <= 0 || n > 8) return -1;
    GemmGroup g; memset(&g, 0, sizeof(g));
    int total = 0, nvec = 0, ny_ = 0;
    for (int p = 0; p < n; ++p) {
        if (M[p] <= 0 || N[p] <= 0 || K[p] <= 0) return -1;
        const float* y = Y ? (const float*)Y[p] : nullptr;
        nvec += ((K[p] & 3) == 0 && (((uintptr_t)A[p] | (uintptr_t)y) & 15) == 0) ? 1 : 0;
        ny_ += y ? 1 : 0;
        const float* e = Yout ? (const float*)Yout[p] : nullptr;
        if (e && (N[p] & 3) == 0 && ((uintptr_t)e & 15) != 0) return -1;
        group_add(g, total, (const float*)A[p], (const float*)B[p], nullptr, y, (float*)C[p], M[p], N[p], K[p], K[p], N[p], 0,
                  (N[p] + GT - 1) / GT, (M[p] + GT - 1) / GT, 1, e);
    }
    for (int p = n; p < 9; ++p) g.first[p] = total;
    if ((nvec != 0 && nvec != n) || (ny_ != 0 && ny_ != n)) return -4;
    hipStream_t s = (hipStream_t)stream;
#define NNG(V, DZ_) hipLaunchKernelGGL((k_gemm_group<false, true, V, false, -1, DZ_, false, false>), dim3(total), dim3(512), 0, s, g)
    if (nvec) { if (ny_) NNG(true, true); else NNG(true, false); }
    else { if (ny_) NNG(false, true); else NNG(false, false); }
#undef NNG
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int tfp_gemm_nn_group(const void* const* A, const void* const* Y, const void* const* B, void* const* C, const int32_t* M, const int32_t* N, const int32_t* K,
                      int32_t n, void* stream) {
    return tfp_gemm_nn_dz_group(A, Y, B, nullptr, C, M, N, K, n, stream);
}
int tfp_gemm_tn_partials_group(const void* const* A, const void* const* Y, const void* const* B, void* const* part, const int32_t* rows, const int32_t* N1,
                               const int32_t* N2, int32_t chunk, int32_t n, void* stream) {
    if (n <= 0 || n > 8 || chunk <= 0 || (chunk % GK) != 0) return -1;
    GemmGroup g; memset(&g, 0, sizeof(g));
    int total = 0, ny_ = 0;
    for (int p = 0; p < n; ++p) {
        if (rows[p] <= 0 || N1[p] <= 0 || N2[p] <= 0) return -1;
        const float* y = Y ? (const float*)Y[p] : nullptr;
        ny_ += y ? 1 : 0;
        group_add(g, total, (const float*)A[p], (const float*)B[p], nullptr, y, (float*)part[p], N1[p], N2[p] + 1, rows[p], N1[p], N2[p], chunk,
                  (N2[p] + 1 + GT - 1) / GT, (N1[p] + GT - 1) / GT, (rows[p] + chunk - 1) / chunk);
    }
    for (int p = n; p < 9; ++p) g.first[p] = total;
    if (ny_ != 0 && ny_ != n) return -4;
    hipStream_t s = (hipStream_t)stream;
    if (ny_) hipLaunchKernelGGL((k_gemm_group<true, true, false, false, -1, true, true, true>), dim3(total), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_group<true, true, false, false, -1, false, true, true>), dim3(total), dim3(512), 0, s, g);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
#ifdef GEMM_TIMING
int tfp_debug_read_wg(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -3; }
int tfp_debug_read(unsigned* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned) * 64) == hipSuccess ? 0 : -3; }
#endif
}  // extern "C"

// ---- one launch for the minibatch gather: dst_k[i, :] = src_k[idx[i], :] for up to 8 row-major float arrays of different widths ----
struct GatherArgs { const float* src[8]; float* dst[8]; int width[8]; int first[8]; int n; };   // first[k]: first column of array k in the concatenation
__global__ void __launch_bounds__(256) k_gather_rows(GatherArgs ga, const long long* __restrict__ idx, int rows, int total_width) {
    // one WAVEFRONT per row: the source row index is read once, every array of the row is copied by the 64 lanes in turn (consecutive lanes, consecutive
    // floats; uniform loop bounds).  The first form - one thread per element of the concatenated row, a 64-bit division and an 8-way search per element - took
    // 9.1 us for the 8192 x 175 floats of a minibatch.
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const long long i = idx[r];
#pragma unroll 1
    for (int k = 0; k < ga.n; ++k) {
        const int w = ga.width[k];
        const float* __restrict__ s = ga.src[k] + (size_t)i * w;
        float* __restrict__ d = ga.dst[k] + (size_t)r * w;
        for (int c = lane; c < w; c += 64) d[c] = s[c];
    }
}

// ---- the rollout's bookkeeping of one environment step in one launch each (they stand where ~25 elementwise / copy launches of 3 - 4 us were:
// exp, mul, add, the negative log-likelihood chain, seven copies into the rollout buffers, clones of obs / states, reward scaling, the done cast) ----
// One wavefront per env: a = mu + sigma * eps (sigma = exp(log_std), formed by the caller once per rollout; a product and a sum, rounded separately like
// the two torch operators), nlp = sum over the actions of 0.5 ((a - mu) / sigma)^2 + log_std + 0.5 log(2 pi), and the row of the step filed into slot t of the buffers (obs, states, act, mu, nlp, val).
__global__ void __launch_bounds__(256) k_rollout_record(const float* __restrict__ obs, int Do, const float* __restrict__ states, int Ds, const float* __restrict__ mu,
                                                        const float* __restrict__ log_std, const float* __restrict__ sigma, const float* __restrict__ eps,
                                                        const float* __restrict__ val, int n, int A,
                                                        float* __restrict__ b_obs, float* __restrict__ b_states, float* __restrict__ b_act, float* __restrict__ b_mu,
                                                        float* __restrict__ b_nlp, float* __restrict__ b_val) {
#pragma clang fp contract(off)                               // no fused multiply-adds below (HIP's __fmul_rn / __fadd_rn are inline operators that carry the
                                                             // translation unit's contraction mode with them: they do not prevent it)
    const int env = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (env >= n) return;
    for (int c = lane; c < Do; c += 64) b_obs[(size_t)env * Do + c] = obs[(size_t)env * Do + c];
    for (int c = lane; c < Ds; c += 64) b_states[(size_t)env * Ds + c] = states[(size_t)env * Ds + c];
    float part = 0.0f;
    for (int c = lane; c < A; c += 64) {
        const float m = mu[(size_t)env * A + c], ls = log_std[c], sd = sigma[c];
        const float se = sd * eps[(size_t)env * A + c];
        const float a = m + se;
        b_act[(size_t)env * A + c] = a;
        b_mu[(size_t)env * A + c] = m;
        const float z = (a - m) / sd, zz = z * z, h = 0.5f * zz;
        part += (h + ls) + 0.91893853320467274178f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) { b_nlp[env] = part; b_val[env] = val[env]; }
}
// rew[t] = r * scale, done[t] = float(d != 0); d: one byte per env (torch.bool / uint8)
__global__ void __launch_bounds__(256) k_rollout_reward(const float* __restrict__ r, const unsigned char* __restrict__ d, float scale, int n, float* __restrict__ b_rew,
                                                        float* __restrict__ b_done) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    b_rew[i] = r[i] * scale;
    b_done[i] = d[i] ? 1.0f : 0.0f;
}
// generalised advantage estimation, one thread per env walking its T steps backwards; every product and sum rounded separately in the order of the torch
// loop it replaces (8 launches per step of the horizon): nd = 1 - done; delta = (rew + (gamma val[t+1]) nd) - val[t]; last = delta + ((gamma tau) nd) last
__global__ void __launch_bounds__(256) k_gae(const float* __restrict__ rew, const float* __restrict__ done, const float* __restrict__ val, float gamma, float gamma_tau,
                                             int T, int n, float* __restrict__ adv, float* __restrict__ ret) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float last = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const float nd = 1.0f - done[(size_t)t * n + i], v0 = val[(size_t)t * n + i];
        const float gv = gamma * val[(size_t)(t + 1) * n + i], gvn = gv * nd, s1 = rew[(size_t)t * n + i] + gvn, delta = s1 - v0;
        const float gn = gamma_tau * nd, gl = gn * last;
        last = delta + gl;
        adv[(size_t)t * n + i] = last;
        ret[(size_t)t * n + i] = last + v0;
    }
}

// ---- one launch for the chunk sums of every layer of a backward pass ----
struct SumArgs { const float* part[8]; float* gw[8]; float* gb[8]; int splits[8]; int n1[8]; int n2[8]; int first[9]; int n; };
__global__ void __launch_bounds__(256) k_sum_partials_multi(SumArgs sa) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= sa.first[sa.n]) return;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) k = (j < sa.n && e >= sa.first[j]) ? j : k;
    const int i = e - sa.first[k], N2 = sa.n2[k], tot = sa.n1[k] * (N2 + 1);
    const float* part = sa.part[k];
    float s = 0.0f;
#pragma unroll 8
    for (int z = 0; z < sa.splits[k]; ++z) s += part[(size_t)z * tot + i];
    const int r = i / (N2 + 1), c = i - r * (N2 + 1);
    if (c < N2) sa.gw[k][(size_t)r * N2 + c] = s; else sa.gb[k][r] = s;
}

extern "C" {
// dst[k][i, :] = src[k][idx[i], :], k < n <= 8 (float rows of widths[k]); idx: int64 [rows]
int tfp_gather_rows(const void* const* src, void* const* dst, const int32_t* widths, int32_t n, const void* idx, int32_t rows, void* stream) {
    if (n <= 0 || n > 8 || rows <= 0) return -1;
    GatherArgs ga;
    int tw = 0;
    for (int k = 0; k < 8; ++k) {
        ga.src[k] = k < n ? (const float*)src[k] : nullptr; ga.dst[k] = k < n ? (float*)dst[k] : nullptr;
        ga.width[k] = k < n ? widths[k] : 0; ga.first[k] = tw;
        if (k < n) { if (widths[k] <= 0) return -1; tw += widths[k]; }
    }
    ga.n = n;
    const long long total = (long long)rows * tw;
    (void)total;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ga, (const long long*)idx, rows, tw);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int tfp_rollout_record(const float* obs, int32_t Do, const float* states, int32_t Ds, const float* mu, const float* log_std, const float* sigma, const float* eps,
                       const float* val, int32_t n, int32_t A, float* b_obs, float* b_states, float* b_act, float* b_mu, float* b_nlp, float* b_val, void* stream) {
    if (n <= 0 || A <= 0 || Do <= 0 || Ds < 0 || (Ds > 0 && (!states || !b_states))) return -1;
    hipLaunchKernelGGL(k_rollout_record, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, obs, Do, states, Ds, mu, log_std, sigma, eps, val, n, A, b_obs, b_states,
                       b_act, b_mu, b_nlp, b_val);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int tfp_rollout_reward(const float* r, const void* done_bytes, float scale, int32_t n, float* b_rew, float* b_done, void* stream) {
    if (n <= 0) return -1;
    hipLaunchKernelGGL(k_rollout_reward, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, r, (const unsigned char*)done_bytes, scale, n, b_rew, b_done);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int tfp_gae(const float* rew, const float* done, const float* val, float gamma, float gamma_tau, int32_t T, int32_t n, float* adv, float* ret, void* stream) {
    if (n <= 0 || T <= 0) return -1;
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rew, done, val, gamma, gamma_tau, T, n, adv, ret);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// the products of tfp_gemm_tn_partials summed for n <= 8 layers in one launch (fixed order over the chunks: deterministic)
int tfp_sum_partials_multi(const void* const* part, void* const* gw, void* const* gb, const int32_t* splits, const int32_t* n1, const int32_t* n2,
                           int32_t n, void* stream) {
    if (n <= 0 || n > 8) return -1;
    SumArgs sa;
    int tot = 0;
    for (int k = 0; k < 8; ++k) {
        sa.part[k] = k < n ? (const float*)part[k] : nullptr; sa.gw[k] = k < n ? (float*)gw[k] : nullptr; sa.gb[k] = k < n ? (float*)gb[k] : nullptr;
        sa.splits[k] = k < n ? splits[k] : 0; sa.n1[k] = k < n ? n1[k] : 0; sa.n2[k] = k < n ? n2[k] : 0; sa.first[k] = tot;
        if (k < n) tot += n1[k] * (n2[k] + 1);
    }
    for (int k = n; k < 9; ++k) sa.first[k] = tot;
    sa.n = n;
    hipLaunchKernelGGL(k_sum_partials_multi, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, sa);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
// the weight / bias gradient products only (chunk slabs in `part`); the caller sums them (tfp_sum_partials_multi)
int tfp_gemm_tn_partials(const float* A, const float* Y, const float* B, float* part, int32_t rows, int32_t N1, int32_t N2, int32_t chunk, void* stream) {
    if (rows <= 0 || N1 <= 0 || N2 <= 0 || chunk <= 0 || (chunk % GK) != 0) return -1;
    const int splits = (rows + chunk - 1) / chunk;
    dim3 grid((N2 + 1 + GT - 1) / GT, (N1 + GT - 1) / GT, splits), block(512);
    hipStream_t s = (hipStream_t)stream;
    if (Y) hipLaunchKernelGGL((k_gemm<true, true, false, false, -1, true, true, true>), grid, block, 0, s, A, B, nullptr, Y, part, N1, N2 + 1, rows, N1, N2, chunk, nullptr);
    else hipLaunchKernelGGL((k_gemm<true, true, false, false, -1, false, true, true>), grid, block, 0, s, A, B, nullptr, nullptr, part, N1, N2 + 1, rows, N1, N2, chunk, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
}  // extern "C"
