/* C ABI of the trainer kernels (leibnizgym_amd/csrc/libtrifinger_ppo.so, built from csrc/ppo_kernels.hip for gfx950).
 *
 * These entry points are what the in-repo PPO (leibnizgym_amd/ppo.py; SURVEY.md section 8f-1, BASELINE.json configs[4]) runs its
 * minibatch step on when RL-Games is not installed.  They replace, for the asymmetric actor-critic of the reference's
 * resources/config/rlg/asymm.yaml:2-90 (MLP [400, 200, 100] ELU on obs 41 / states 113, fixed sigma, central value network), the
 * PyTorch operators RL-Games' a2c_continuous agent launches per minibatch: the Linear / ELU layers forwards and backwards
 * (asymm.yaml:12-33, 70-90), the clipped-surrogate / value / bounds objective (asymm.yaml:38-68: e_clip, critic_coef, bounds_loss_coef,
 * entropy_coef), gradient truncation (grad_norm, truncate_grads) and Adam (learning_rate, central_value_config.lr).
 * Plain pointers and sizes; every pointer is DEVICE memory (float32, contiguous, row-major) unless stated; `stream` is a hipStream_t
 * (NULL = the default stream).  Return value: 0 on success, -1 invalid argument, -2 / -3 a HIP call or launch failed.
 * The Python binding is leibnizgym_amd/ppo_kernels.py (ctypes); tests/test_ppo_kernels.py holds every entry point against a plain
 * PyTorch fp32 reference of the same operation.  Arithmetic: fp32 throughout; the matrix products run on v_mfma_f32_32x32x2_f32
 * (fp32 inputs, fp32 accumulation). */
#ifndef TRIFINGER_PPO_H
#define TRIFINGER_PPO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int tfp_api_version(void);                                     /* 3 (2: sq [4] / step [2] of tfp_clip_adam; 3: tfp_mlp_forward / tfp_mlp_backward, tfp_reset_state) */

/* The whole PPO objective of one minibatch, value and gradients, in one launch (B samples, A = 9 or 18 actions):
 *   loss = a_loss + v_coef c_loss - ent_coef entropy + bounds_coef b_loss   (formulas: csrc/ppo_kernels.hip header)
 * Outputs: d_mu [B, A], d_v [B], d_logstd [A], loss_out [1]; stats [4] += (loss, a_loss, c_loss, kl). */
int tfp_ppo_loss(const float* mu, const float* log_std, const float* act, const float* old_nlp, const float* adv, const float* old_mu,
                 const float* v, const float* ret, int32_t B, int32_t A, float e_clip, float v_coef, float ent_coef, float bounds_coef,
                 float* d_mu, float* d_v, float* d_logstd, float* loss_out, float* stats, void* stream);

/* Gradient-norm truncation + Adam for two parameter groups over one flat buffer: group 0 = [0, n0), group 1 = [n0, n1).
 * sq [4] scratch - all zero before the first step, never touched by the caller afterwards (the two launches of a step keep the half the next step sums
 * into clear: no zeroing launch) -, step [2] = (this step, completed steps) on the device, both zero at the start, lr [2] learning rates (device).
 * The counter is a float and stops counting at 2^23 + 1 (from there it alternates between 2^23 and 2^23 + 1: its parity selects the half of sq and must
 * never freeze; 1 - beta^t is 1.0f by then), so `completed steps` read back from a run longer than 8.4 M steps is a lower bound.
 * ONE call at a time per device and process for tfp_ppo_loss (it sums into accumulators of the library that the last block hands over and clears:
 * two calls in flight on different streams would mix their sums); tfp_reset_state() restores the zero state after a launch that was aborted. */
int tfp_clip_adam(float* p, const float* g, float* m, float* v, int32_t n0, int32_t n1, float* sq, float* step, const float* lr,
                  float max_norm0, float max_norm1, float beta1, float beta2, float eps, void* stream);

/* the accumulators / ticket of tfp_ppo_loss back to zero (one tiny launch on `stream`): only needed after a launch of it failed or was aborted */
int tfp_reset_state(void* stream);

/* C[M, N] = act(A[M, K] . W[N, K]^T + bias[N]); act: 0 none, 1 ELU (torch.nn.Linear followed by torch.nn.ELU) */
int tfp_linear_fwd(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K, int32_t act, void* stream);

/* Input gradient: C[M, N] = dZ[M, K] . B[K, N], dZ = A, or A * elu'(Y) when Y (the layer's ELU output, [M, K]) is not NULL */
int tfp_gemm_nn(const float* A, const float* Y, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream);
/* ... the same product leaving as the dZ of the layer BELOW: C[M, N] = (dZ . B) * elu'(Yout[M, N]) with Yout that layer's ELU output (NULL: no factor).
 * A backward walk that chains these (A = the dZ the call above produced, Y = NULL) multiplies every activation gradient once, where the product is
 * stored, instead of in the operand loads of the two products that consume it; same bits.  N % 4 == 0 needs Yout 16-byte aligned. */
int tfp_gemm_nn_dz(const float* A, const float* Y, const float* B, const float* Yout, float* C, int32_t M, int32_t N, int32_t K, void* stream);

/* Weight and bias gradient: gw[N1, N2] = dZ^T B, gb[N1] = column sums of dZ, for dZ[rows, N1] (as above), B[rows, N2].
 * `part` is scratch for ceil(rows / chunk) slabs of N1 * (N2 + 1) floats; chunk must be a multiple of 32.  The sum over the row
 * chunks runs in a fixed order (deterministic). */
int tfp_gemm_tn_bias(const float* A, const float* Y, const float* B, float* part, float* gw, float* gb, int32_t rows, int32_t N1, int32_t N2,
                     int32_t chunk, void* stream);
/* ... the chunk products only; tfp_sum_partials_multi then sums the slabs of up to 8 layers in one launch
 * (host arrays of n device pointers / sizes) */
int tfp_gemm_tn_partials(const float* A, const float* Y, const float* B, float* part, int32_t rows, int32_t N1, int32_t N2, int32_t chunk,
                         void* stream);
int tfp_sum_partials_multi(const void* const* part, void* const* gw, void* const* gb, const int32_t* splits, const int32_t* n1,
                           const int32_t* n2, int32_t n, void* stream);

/* The three products above for n <= 8 INDEPENDENT problems in ONE launch (host arrays of n device pointers / sizes): the layers of the actor and of the
 * central value network side by side, the eight weight gradients of a minibatch step together.  Each of these products takes 5 - 25 us on its own, of
 * which the dispatch, the cold first loads and the tail are a third; packed into one grid the next problem's workgroups fill them.  All problems of a call
 * must be of one kind - the same alignment class (K % 4 == 0 with 16-byte aligned operands, or not), one activation, Y given for all or for none:
 * otherwise -4, and the caller launches them one by one.  Results are bit-identical to the single-problem entry points (same tiles, same order). */
int tfp_linear_fwd_group(const void* const* A, const void* const* W, const void* const* bias, void* const* C, const int32_t* M, const int32_t* N,
                         const int32_t* K, int32_t act, int32_t n, void* stream);
int tfp_gemm_nn_group(const void* const* A, const void* const* Y /* may be NULL */, const void* const* B, void* const* C, const int32_t* M, const int32_t* N,
                      const int32_t* K, int32_t n, void* stream);
int tfp_gemm_nn_dz_group(const void* const* A, const void* const* Y /* may be NULL */, const void* const* B, const void* const* Yout /* may be NULL */,
                         void* const* C, const int32_t* M, const int32_t* N, const int32_t* K, int32_t n, void* stream);
int tfp_gemm_tn_partials_group(const void* const* A, const void* const* Y /* may be NULL */, const void* const* B, void* const* part, const int32_t* rows,
                               const int32_t* N1, const int32_t* N2, int32_t chunk, int32_t n, void* stream);

/* The weight / bias gradients of n <= 8 problems in ONE launch without LDS staging (csrc/ppo_dw_direct.hip): 64 x 64 output blocks whose operands come
 * straight from memory as interleaved MFMA fragments (one dwordx4 per operand and 4-k step feeds 16 MFMAs), a workgroup = four 256-row pieces of one
 * chunk of tfp_gemm_tn_partials_direct_chunk() = 1024 rows, summed through LDS in a fixed order.  A[p] = dZ [rows, N1] (already times the activation derivative), B[p] = the layer input [rows, N2];
 * part[p] receives ceil(rows / chunk) slabs of N1 * (N2 + 1) floats, [dW | db] per slab: sum them with tfp_sum_partials_multi (splits = ceil(rows / chunk)).
 * -4: more than 160 blocks in the call or an operand beyond 4 GB (the caller uses tfp_gemm_tn_partials_group). */
int tfp_gemm_tn_partials_direct_chunk(void);                    /* rows per slab of the build (1024) */
int tfp_gemm_tn_partials_direct(const void* const* A, const void* const* B, void* const* part, const int32_t* rows, const int32_t* N1, const int32_t* N2,
                                int32_t n, void* stream);

/* The rollout's bookkeeping of one environment step (leibnizgym_amd/ppo.py::PPOTrainer.rollout; RL-Games' a2c_common.play_steps does the same with
 * PyTorch operators): one launch samples the action a = mu + sigma * eps (sigma = exp(log_std) [A] and eps ~ N(0, 1) [n, A] supplied by the caller), its negative log-likelihood, and files
 * the step into slot t of the caller's buffers (b_*: pointers to that slot; states / b_states may be NULL with Ds = 0); one launch scales the reward and
 * converts the done flags (one byte per env); one launch runs generalised advantage estimation over the horizon (adv, ret: [T, n]; val: [T + 1, n]).
 * Every product and sum is rounded separately in the order of the PyTorch expressions they replace: the buffers hold the same bits - except b_nlp, whose
 * A terms are the same bits but are summed by a wave butterfly, not in torch's reduction order: equal to the torch expression to 2e-6, and not the bits
 * tfp_ppo_loss recomputes (it sums a = 0..A-1 in turn), so the probability ratio of the first minibatch is 1 to rounding, not exactly. */
int tfp_rollout_record(const float* obs, int32_t Do, const float* states, int32_t Ds, const float* mu, const float* log_std, const float* sigma, const float* eps,
                       const float* val, int32_t n, int32_t A, float* b_obs, float* b_states, float* b_act, float* b_mu, float* b_nlp, float* b_val, void* stream);
int tfp_rollout_reward(const float* r, const void* done_bytes, float scale, int32_t n, float* b_rew, float* b_done, void* stream);
int tfp_gae(const float* rew, const float* done, const float* val, float gamma, float gamma_tau, int32_t T, int32_t n, float* adv, float* ret, void* stream);

/* The network walk (csrc/ppo_mlp_walk.hip): ALL layers of n_nets <= 2 Linear / ELU stacks over the same M rows in ONE launch per direction - what
 * RL-Games' a2c network runs as 2 x (4 Linear + 3 ELU) operators forwards and as many again backwards (asymm.yaml:12-33, 70-90).  A workgroup owns 64 rows
 * and walks the layers with them, activations in LDS, weights streamed from L2 as MFMA fragments (v_mfma_f32_16x16x4_f32, fp32 in, fp32 accumulate).
 *   forward:  x = input [M, dim[0]]; W[l] [dim[l+1], dim[l]], b[l] [dim[l+1]], act[l] (1: ELU behind layer l); y[l] [M, dim[l+1]] receives the output of
 *             layer l - hidden ones may be NULL (not stored: the rollout), the last one must be given.  yin is not read.
 *   backward: x = gradient of the network OUTPUT [M, dim[n_layers]]; yin[l] = the output of layer l the forward stored; y[l] [M, dim[l+1]] receives dZ_l =
 *             d loss / d (pre-activation of layer l) for l < n_layers - 1 (dZ of the last layer is x itself); b is not read.  The weight gradients are
 *             dZ_l^T [input_l | 1]: tfp_gemm_tn_partials_group on what this call stored.
 * Limits: n_layers <= 4, every dim[l >= 1] <= 416, the LDS budget of 160 KB (two alternating activation buffers of 64 rows).  -4: the shapes do not fit;
 * the caller runs the layers one by one (tfp_linear_fwd_group / tfp_gemm_nn_dz_group).  ELU = x > 0 ? x : exp(x) - 1 with exp on v_exp_f32 (|error| <= 1.2e-7). */
typedef struct {
    const float* x;
    const float* W[4];
    const float* b[4];
    const float* yin[4];
    float* y[4];
    int32_t dim[5];
    int32_t act[4];
    int32_t n_layers;
} TfpMlp;
int tfp_mlp_forward(const TfpMlp* nets, int32_t n_nets, int32_t M, void* stream);
int tfp_mlp_backward(const TfpMlp* nets, int32_t n_nets, int32_t M, void* stream);

/* Minibatch gather: dst[k][i, :] = src[k][idx[i], :] for n <= 8 float arrays of widths[k] columns (host arrays of n device pointers);
 * idx: int64 [rows] on the device */
int tfp_gather_rows(const void* const* src, void* const* dst, const int32_t* widths, int32_t n, const void* idx, int32_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif
