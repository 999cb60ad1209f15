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
#include <stdint.h>

#define MAX_A 18

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

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
            if (add != 0.0f) atomicAdd(&d_logstd[q], add);
        } else if (q == A) {
            // thread A also assembles the block's share of the loss (it needs the three other sums of the block)
            const float bsa = t;
            const float bsc = (red[0][A + 1] + red[1][A + 1]) + (red[2][A + 1] + red[3][A + 1]);
            const float bsb = (red[0][A + 2] + red[1][A + 2]) + (red[2][A + 2] + red[3][A + 2]);
            float part = bsa + v_coef * bsc + bounds_coef * bsb;
            if (blockIdx.x == 0) part += -ent_coef * ent;
            atomicAdd(&loss_out[0], part);
            atomicAdd(&stats[0], part);
            atomicAdd(&stats[1], bsa);
        } else if (q == A + 1) {
            atomicAdd(&stats[2], t);
        } else if (q == A + 3) {
            atomicAdd(&stats[3], t);
        }
    }
}

// ---- gradient-norm truncation + Adam over FLAT buffers, two parameter groups (actor | central value network) ------------------
// torch's fused multi-tensor Adam and clip_grad_norm_ take ~110 us per step for these 32 small tensors (two 38 us launches for
// the two groups plus the norm / scale launches); over one flat buffer of 264 k floats the same arithmetic is two 5 us launches.
// Group 0 = elements [0, n0), group 1 = [n0, n1).  k_grad_sqnorms also advances the step counter (kernel boundary = ordering).
__global__ void __launch_bounds__(256) k_grad_sqnorms(const float* __restrict__ g, int n0, int n1, float* __restrict__ sq, float* __restrict__ step) {
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
        if (t != 0.0f) atomicAdd(&sq[q], t);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) step[0] += 1.0f;
}
// torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (||g|| + 1e-6)) per group; then torch.optim.Adam (no weight decay, no
// amsgrad): m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void __launch_bounds__(256) k_clip_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int n0, int n1, const float* __restrict__ sq,
                                                   const float* __restrict__ step, const float* __restrict__ lr, float max0, float max1,
                                                   float b1, float b2, float eps) {
    const float t = step[0];
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float c0 = fminf(max0 / (sqrtf(sq[0]) + 1e-6f), 1.0f), c1 = fminf(max1 / (sqrtf(sq[1]) + 1e-6f), 1.0f);
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

int tfp_api_version(void) { return 1; }

// d_logstd [A] and loss_out [1] are zeroed here (on the stream); stats [4] (loss, a_loss, c_loss, kl) ACCUMULATE across calls.
int tfp_ppo_loss(const float* mu, const float* log_std, const float* act, const float* old_nlp, const float* adv, const float* old_mu,
                 const float* v, const float* ret, int32_t B, int32_t A, float e_clip, float v_coef, float ent_coef, float bounds_coef,
                 float* d_mu, float* d_v, float* d_logstd, float* loss_out, float* stats, void* stream) {
    if (B <= 0 || (A != 9 && A != 18)) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (loss_out == d_logstd + A) {                              // one buffer [A + 1]: one fill
        if (hipMemsetAsync(d_logstd, 0, sizeof(float) * (A + 1), s) != hipSuccess) return -2;
    } else {
        if (hipMemsetAsync(d_logstd, 0, sizeof(float) * A, s) != hipSuccess) return -2;
        if (hipMemsetAsync(loss_out, 0, sizeof(float), s) != hipSuccess) return -2;
    }
    dim3 grid((B + 255) / 256), block(256);
    if (A == 9)
        hipLaunchKernelGGL((k_ppo_loss<9>), grid, block, 0, s, mu, log_std, act, old_nlp, adv, old_mu, v, ret, B, e_clip, v_coef, ent_coef,
                           bounds_coef, d_mu, d_v, d_logstd, loss_out, stats);
    else
        hipLaunchKernelGGL((k_ppo_loss<18>), grid, block, 0, s, mu, log_std, act, old_nlp, adv, old_mu, v, ret, B, e_clip, v_coef, ent_coef,
                           bounds_coef, d_mu, d_v, d_logstd, loss_out, stats);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// One optimisation step over flat buffers: sq [2] is scratch (zeroed here), step [1] and lr [2] live on the device (graph capture).
int tfp_clip_adam(float* p, const float* g, float* m, float* v, int32_t n0, int32_t n1, float* sq, float* step, const float* lr,
                  float max_norm0, float max_norm1, float beta1, float beta2, float eps, void* stream) {
    if (n1 <= 0 || n0 < 0 || n0 > n1) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sq, 0, 2 * sizeof(float), s) != hipSuccess) return -2;
    const int blocks = (n1 + 256 * 4 - 1) / (256 * 4);
    hipLaunchKernelGGL(k_grad_sqnorms, dim3(blocks), dim3(256), 0, s, g, n0, n1, sq, step);
    hipLaunchKernelGGL(k_clip_adam, dim3(blocks), dim3(256), 0, s, p, g, m, v, n0, n1, sq, step, lr, max_norm0, max_norm1, beta1, beta2, eps);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
