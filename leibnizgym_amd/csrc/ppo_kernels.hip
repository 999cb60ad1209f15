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
    // reductions: wave shuffles, one atomic per wave and quantity
    const int lane = threadIdx.x & 63;
    float ent = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        ent += ls[a] + 1.4189385332046727f;
        const float s = wave_sum(dls[a]);
        if (lane == 0 && s != 0.0f) atomicAdd(&d_logstd[a], s);
    }
    const float sa = wave_sum(a_term) * invB, sc = wave_sum(c_term) * invB, sb = wave_sum(b_term) * invB, sk = wave_sum(kl_term) * invB;
    if (lane == 0) {
        const float part = sa + v_coef * sc + bounds_coef * sb;
        atomicAdd(&loss_out[0], part);
        atomicAdd(&stats[0], part);
        atomicAdd(&stats[1], sa);
        atomicAdd(&stats[2], sc);
        atomicAdd(&stats[3], sk);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {                      // the batch-independent entropy term, once
        atomicAdd(&loss_out[0], -ent_coef * ent);
        atomicAdd(&stats[0], -ent_coef * ent);
#pragma unroll
        for (int a = 0; a < A; ++a) { if (ent_coef != 0.0f) atomicAdd(&d_logstd[a], -ent_coef); }
    }
}

// dz = dy * elu'(y) with elu'(y) = 1 for y > 0, y + 1 otherwise (alpha = 1: elu(x) = e^x - 1, derivative e^x = y + 1), written in
// place of dy, and db = column sums of dz.  One block per 64 rows, a thread per column (strided over N): coalesced rows.
__global__ void __launch_bounds__(256) k_elu_bwd_bias(float* __restrict__ dy, const float* __restrict__ y, int M, int N,
                                                      float* __restrict__ db) {
    const int r0 = blockIdx.x * 64;
    const int r1 = min(r0 + 64, M);
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        float s = 0.0f;
        for (int r = r0; r < r1; ++r) {
            const size_t k = (size_t)r * N + j;
            const float yy = y[k];
            const float g = dy[k] * (yy > 0.0f ? 1.0f : yy + 1.0f);
            dy[k] = g;
            s += g;
        }
        atomicAdd(&db[j], s);
    }
}

// column sums of a row-major [M, N] matrix (bias gradient of a layer without activation)
__global__ void __launch_bounds__(256) k_col_sum(const float* __restrict__ x, int M, int N, float* __restrict__ out) {
    const int r0 = blockIdx.x * 64;
    const int r1 = min(r0 + 64, M);
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        float s = 0.0f;
        for (int r = r0; r < r1; ++r) s += x[(size_t)r * N + j];
        atomicAdd(&out[j], s);
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
    if (hipMemsetAsync(d_logstd, 0, sizeof(float) * A, s) != hipSuccess) return -2;
    if (hipMemsetAsync(loss_out, 0, sizeof(float), s) != hipSuccess) return -2;
    dim3 grid((B + 255) / 256), block(256);
    if (A == 9)
        hipLaunchKernelGGL((k_ppo_loss<9>), grid, block, 0, s, mu, log_std, act, old_nlp, adv, old_mu, v, ret, B, e_clip, v_coef, ent_coef,
                           bounds_coef, d_mu, d_v, d_logstd, loss_out, stats);
    else
        hipLaunchKernelGGL((k_ppo_loss<18>), grid, block, 0, s, mu, log_std, act, old_nlp, adv, old_mu, v, ret, B, e_clip, v_coef, ent_coef,
                           bounds_coef, d_mu, d_v, d_logstd, loss_out, stats);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// in place: dy <- dy * elu'(y); db [N] <- column sums (zeroed here)
int tfp_elu_bwd_bias(float* dy, const float* y, int32_t M, int32_t N, float* db, void* stream) {
    if (M <= 0 || N <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(db, 0, sizeof(float) * N, s) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_elu_bwd_bias, dim3((M + 63) / 64), dim3(256), 0, s, dy, y, M, N, db);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int tfp_col_sum(const float* x, int32_t M, int32_t N, float* out, void* stream) {
    if (M <= 0 || N <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * N, s) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_col_sum, dim3((M + 63) / 64), dim3(256), 0, s, x, M, N, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
