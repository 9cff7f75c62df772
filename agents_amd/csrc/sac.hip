// SAC: tanh-squashed Normal actor head (sample, log-probability, backward) and the three losses.
//
// Follows, op for op (fp32, built with -ffp-contract=off so that the elementwise arithmetic can be
// compared with the torch-CPU oracle in oracle/sac.py):
//   TanhNormalProjectionNetwork.call              tf_agents/agents/sac/tanh_normal_projection_network.py:108-143
//   std_clip_transform                            tf_agents/agents/sac/sac_agent.py:48-57
//   SquashToSpecNormal (Shift(Scale) o Tanh)      tf_agents/distributions/utils.py:40-160
//   tanh_bijector_stable forward_log_det_jacobian 2 (log 2 - x - softplus(-2x))
//   SacAgent._actions_and_log_probs               tf_agents/agents/sac/sac_agent.py:533-558
//   SacAgent.critic_loss / actor_loss / alpha_loss  :559-740
//   common.aggregate_losses                       tf_agents/utils/common.py:1400-1476
// The pre-tanh sample x of a drawn action is kept (TFP's bijector cache does the same when
// log_prob is asked for a tensor the distribution just sampled), so log pi is evaluated at x, not at
// atanh(action).
#include "common.h"
#include "agents_amd.h"
#include "sac_sample.h"
#include "sac_loss.h"
#include <math.h>

// One thread per (sample, action dimension); the A per-dimension log-density terms of a sample are
// then summed in dimension order by its d == 0 thread through LDS (deterministic; a thread per
// sample walking 17 dimensions of exp / log / tanh / Philox took 18 us for 256 samples).
// z = [mean | raw_std] (the projection Dense output, [B, 2A]).  eps_in (nullable): externally
// supplied N(0,1) noise [B, A]; else Box-Muller on Philox(counter = (b*A+d, call), key = seed)
// like aa_normal_sample.  Requires A <= 256 (one sample never straddles a workgroup).
__global__ void __launch_bounds__(256)
aa_sac_sample_kernel(const float* __restrict__ z, int64_t B, int A, int per_block,
                     const float* __restrict__ act_mean, const float* __restrict__ act_mag,
                     int std_kind, const float* __restrict__ eps_in, uint32_t seed_lo,
                     uint32_t seed_hi, int64_t* __restrict__ call_counter,
                     int64_t* __restrict__ arrival, float* __restrict__ action,
                     float* __restrict__ logp, float* __restrict__ save_tanh,
                     float* __restrict__ save_sigma, float* __restrict__ save_eps) {
  __shared__ float terms[256];
  const uint64_t call = call_counter != nullptr ? (uint64_t)call_counter[0] : 0ull;
  const int local = threadIdx.x / A, d = threadIdx.x - local * A;
  const int64_t b = (int64_t)blockIdx.x * per_block + local;
  const bool live = local < per_block && b < B;
  float term = 0.f;
  if (live) {
    const AaSacElem o = aa_sac_sample_elem(z[b * 2 * A + d], z[b * 2 * A + A + d], std_kind, eps_in,
                                           (uint64_t)(b * A + d), call, seed_lo, seed_hi,
                                           act_mean[d], act_mag[d]);
    action[b * A + d] = o.action;
    term = o.term;
    if (save_tanh != nullptr) {
      save_tanh[b * A + d] = o.t;
      save_sigma[b * A + d] = o.sigma;
      save_eps[b * A + d] = o.eps;
    }
  }
  terms[threadIdx.x] = term;
  __syncthreads();
  if (live && d == 0) {
    float lp = 0.f;
    for (int k = 0; k < A; ++k) lp += terms[local * A + k];   // dimension order, as before
    logp[b] = lp;
  }
  // the Philox call counter moves on inside the launch (every workgroup has read it by the time
  // the last one finishes), not in a one-thread launch of its own
  if (arrival != nullptr && eps_in == nullptr && call_counter != nullptr)
    aa_advance_when_all_done(call_counter, arrival, 1, gridDim.x);
}

// dz[B,2A] from dL/daction [B,A] and dL/dlog_pi [B] (reparameterised sample x = mu + sigma eps):
//   g_x = da * mag * (1 - t^2) + dlogp * 2t ;  dmu = g_x ;  dsigma = g_x * eps - dlogp / sigma
//   draw = dsigma * sigma   (exp; zero outside [-20, 2] with the clip transform)
__global__ void __launch_bounds__(256)
aa_sac_head_bwd_kernel(const float* __restrict__ z, int64_t B, int A,
                       const float* __restrict__ act_mag, int std_kind,
                       const float* __restrict__ save_tanh, const float* __restrict__ save_sigma,
                       const float* __restrict__ save_eps, const float* __restrict__ daction,
                       int64_t ld_da, const float* __restrict__ daction2, int64_t ld_da2,
                       const float* __restrict__ dlogp, float* __restrict__ dz) {
  const int64_t total = B * A, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / A;
    const int d = (int)(i - b * A);
    const float dl = dlogp[b];
    // d loss / d action: one tensor, or the sum of two (the twin critics' input gradients)
    float da = daction != nullptr ? daction[b * ld_da + d] : 0.f;
    if (daction2 != nullptr) da = da + daction2[b * ld_da2 + d];
    float gx, draw;
    aa_sac_head_bwd_elem(save_tanh[i], save_sigma[i], save_eps[i], dl, da, act_mag[d],
                         z[b * 2 * A + A + d], std_kind, &gx, &draw);
    dz[b * 2 * A + d] = gx;
    dz[b * 2 * A + A + d] = draw;
  }
}

// critic loss (sac_agent.py:559-644), one workgroup, deterministic.
__global__ void __launch_bounds__(256)
aa_sac_critic_loss_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                          const float* __restrict__ tq1, const float* __restrict__ tq2,
                          const float* __restrict__ next_logp, const float* __restrict__ reward,
                          const float* __restrict__ discount, const float* __restrict__ weights,
                          const float* __restrict__ log_alpha, float gamma, float reward_scale,
                          int loss_kind, float loss_weight, int64_t B, float global_batch,
                          float* __restrict__ loss_out, float* __restrict__ td_target_out,
                          float* __restrict__ dq1, float* __restrict__ dq2) {
  __shared__ float red[16];
  const float alpha = expf(log_alpha[0]);
  float local = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
    const AaSacCriticElem o = aa_sac_critic_elem(q1[b], q2[b], tq1[b], tq2[b], next_logp[b],
                                                 reward[b], discount[b], weights, b, alpha, gamma,
                                                 reward_scale, loss_kind);
    local += o.wl;
    if (td_target_out != nullptr) td_target_out[b] = o.td;
    if (dq1 != nullptr) {
      dq1[b] = (loss_weight * o.g1 * o.w) / global_batch;
      dq2[b] = (loss_weight * o.g2 * o.w) / global_batch;
    }
  }
  const float total = aa_block_sum(local, red);
  if (threadIdx.x == 0) loss_out[0] = loss_weight * (total / global_batch);
}

// actor loss (sac_agent.py:646-694): mean_b w (alpha log_pi - min(q1, q2)).
__global__ void __launch_bounds__(256)
aa_sac_actor_loss_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                         const float* __restrict__ logp, const float* __restrict__ weights,
                         const float* __restrict__ log_alpha, float loss_weight, int64_t B,
                         float global_batch, float* __restrict__ loss_out,
                         float* __restrict__ dq1, float* __restrict__ dq2,
                         float* __restrict__ dlogp) {
  __shared__ float red[16];
  const float alpha = expf(log_alpha[0]);
  float local = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
    const AaSacActorElem o = aa_sac_actor_elem(q1[b], q2[b], logp[b], weights, b, alpha,
                                               loss_weight, global_batch);
    local += o.wl;
    if (dq1 != nullptr) {
      dq1[b] = o.dq1;
      dq2[b] = o.dq2;
      dlogp[b] = o.dlogp;
    }
  }
  const float total = aa_block_sum(local, red);
  if (threadIdx.x == 0) loss_out[0] = loss_weight * (total / global_batch);
}

// alpha loss (sac_agent.py:696-740): mean_b w * c(log_alpha) * stop_gradient(-log_pi - H),
// c = log_alpha or exp(log_alpha).  grad_out[0] = d loss / d log_alpha.
__global__ void __launch_bounds__(256)
aa_sac_alpha_loss_kernel(const float* __restrict__ logp, const float* __restrict__ weights,
                         const float* __restrict__ log_alpha, float target_entropy,
                         int use_log_alpha, float loss_weight, int64_t B, float global_batch,
                         float* __restrict__ loss_out, float* __restrict__ grad_out) {
  __shared__ float red[16];
  const float la = log_alpha[0];
  const float coef = use_log_alpha ? la : expf(la);
  float local = 0.f, gsum = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
    const float diff = -logp[b] - target_entropy;
    const float l = coef * diff;
    float w = 1.f, wl = l, wd = diff;
    if (weights != nullptr) {
      w = weights[b];
      wl = (w == 0.f) ? 0.f : l * w;
      wd = (w == 0.f) ? 0.f : diff * w;
    }
    local += wl;
    gsum += wd;
  }
  const float total = aa_block_sum(local, red);
  __syncthreads();
  const float gtot = aa_block_sum(gsum, red);
  if (threadIdx.x == 0) {
    loss_out[0] = loss_weight * (total / global_batch);
    if (grad_out != nullptr)
      grad_out[0] = loss_weight * ((use_log_alpha ? 1.0f : expf(la)) * gtot / global_batch);
  }
}

// The tail of a SAC train step in ONE launch: alpha loss + gradient (as above), one Adam step on
// log_alpha with that gradient (the arithmetic of aa_adam_kernel, optim.hip: t = steps + 1,
// alpha_t = lr sqrt(1 - b2^t) / (1 - b1^t), m += (g - m)(1 - b1), v += (g g - v)(1 - b2),
// p -= m alpha_t / (sqrt(v) + eps)) and the LossInfo pack of aa_pack_sum3_kernel -- three
// single-workgroup launches of ~5 us each otherwise.
__global__ void __launch_bounds__(256)
aa_sac_alpha_step_kernel(const float* __restrict__ logp, const float* __restrict__ weights,
                         float* __restrict__ log_alpha, float target_entropy, int use_log_alpha,
                         float loss_weight, int64_t B, float global_batch,
                         float* __restrict__ loss_out, float* __restrict__ grad_out,
                         float* __restrict__ adam_m, float* __restrict__ adam_v,
                         int64_t* __restrict__ adam_steps, float lr, float beta1, float beta2,
                         float eps, const float* __restrict__ critic_loss,
                         const float* __restrict__ actor_loss, float* __restrict__ packed4) {
  __shared__ float red[16];
  const float la = log_alpha[0];
  const float coef = use_log_alpha ? la : expf(la);
  float local = 0.f, gsum = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
    const float diff = -logp[b] - target_entropy;
    const float l = coef * diff;
    float w = 1.f, wl = l, wd = diff;
    if (weights != nullptr) {
      w = weights[b];
      wl = (w == 0.f) ? 0.f : l * w;
      wd = (w == 0.f) ? 0.f : diff * w;
    }
    local += wl;
    gsum += wd;
  }
  const float total = aa_block_sum(local, red);
  __syncthreads();
  const float gtot = aa_block_sum(gsum, red);
  if (threadIdx.x == 0) {
    const float loss = loss_weight * (total / global_batch);
    const float grad = loss_weight * ((use_log_alpha ? 1.0f : expf(la)) * gtot / global_batch);
    loss_out[0] = loss;
    grad_out[0] = grad;
    const float t = (float)(adam_steps[0] + 1);
    const float b1p = powf(beta1, t), b2p = powf(beta2, t);
    const float alpha_t = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    float m = adam_m[0], v = adam_v[0];
    m = m + (grad - m) * (1.0f - beta1);
    v = v + (grad * grad - v) * (1.0f - beta2);
    log_alpha[0] = la - (m * alpha_t) / (sqrtf(v) + eps);
    adam_m[0] = m;
    adam_v[0] = v;
    adam_steps[0] += 1;
    const float x = critic_loss[0], y = actor_loss[0];
    packed4[0] = (x + y) + loss;
    packed4[1] = x;
    packed4[2] = y;
    packed4[3] = loss;
  }
}

extern "C" {

int aa_sac_sample(const float* z, int64_t B, int32_t A, const float* act_mean,
                  const float* act_mag, int32_t std_kind, const float* eps_in, uint64_t seed,
                  int64_t* call_counter_dev, int64_t* arrival_dev, float* action, float* logp,
                  float* save_tanh, float* save_sigma, float* save_eps, void* stream) {
  if (!z || !act_mean || !act_mag || !action || !logp || B <= 0 || A <= 0) return AA_ERR_INVALID;
  if (std_kind != AA_SAC_STD_EXP && std_kind != AA_SAC_STD_CLIP_EXP) return AA_ERR_INVALID;
  if ((save_tanh == nullptr) != (save_sigma == nullptr) ||
      (save_tanh == nullptr) != (save_eps == nullptr))
    return AA_ERR_INVALID;
  if (A > 256) return AA_ERR_RANGE;
  const int per_block = 256 / A;
  const int64_t blocks = (B + per_block - 1) / per_block;
  if (blocks > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_sac_sample_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, z, B, A, per_block, act_mean, act_mag, std_kind, eps_in,
                     (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev, arrival_dev, action,
                     logp,
                     save_tanh, save_sigma, save_eps);
  return aa_launch_status();
}

int aa_sac_head_backward(const float* z, int64_t B, int32_t A, const float* act_mag,
                         int32_t std_kind, const float* save_tanh, const float* save_sigma,
                         const float* save_eps, const float* daction, int64_t ld_daction,
                         const float* daction2, int64_t ld_daction2, const float* dlogp,
                         float* dz, void* stream) {
  if (!z || !act_mag || !save_tanh || !save_sigma || !save_eps || !dlogp || !dz || B <= 0 ||
      A <= 0)
    return AA_ERR_INVALID;
  if ((daction != nullptr && ld_daction < A) || (daction2 != nullptr && ld_daction2 < A) ||
      (daction2 != nullptr && daction == nullptr))
    return AA_ERR_INVALID;
  int64_t blocks = (B * A + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aa_sac_head_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, z, B, A, act_mag, std_kind, save_tanh, save_sigma,
                     save_eps, daction, ld_daction, daction2, ld_daction2, dlogp, dz);
  return aa_launch_status();
}

int aa_sac_critic_loss(const float* q1, const float* q2, const float* tq1, const float* tq2,
                       const float* next_logp, const float* reward, const float* discount,
                       const float* weights, const float* log_alpha_dev, float gamma,
                       float reward_scale, int32_t loss_kind, float loss_weight, int64_t B,
                       float global_batch, float* loss_out, float* td_target_out, float* dq1,
                       float* dq2, void* stream) {
  if (!q1 || !q2 || !tq1 || !tq2 || !next_logp || !reward || !discount || !log_alpha_dev ||
      !loss_out || B <= 0 || !(global_batch > 0.f))
    return AA_ERR_INVALID;
  if ((dq1 == nullptr) != (dq2 == nullptr)) return AA_ERR_INVALID;
  if (loss_kind != AA_LOSS_HUBER && loss_kind != AA_LOSS_SQUARED) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_sac_critic_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q1, q2,
                     tq1, tq2, next_logp, reward, discount, weights, log_alpha_dev, gamma,
                     reward_scale, loss_kind, loss_weight, B, global_batch, loss_out,
                     td_target_out, dq1, dq2);
  return aa_launch_status();
}

int aa_sac_actor_loss(const float* q1, const float* q2, const float* logp, const float* weights,
                      const float* log_alpha_dev, float loss_weight, int64_t B,
                      float global_batch, float* loss_out, float* dq1, float* dq2, float* dlogp,
                      void* stream) {
  if (!q1 || !q2 || !logp || !log_alpha_dev || !loss_out || B <= 0 || !(global_batch > 0.f))
    return AA_ERR_INVALID;
  if ((dq1 == nullptr) != (dq2 == nullptr) || (dq1 == nullptr) != (dlogp == nullptr))
    return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_sac_actor_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q1, q2,
                     logp, weights, log_alpha_dev, loss_weight, B, global_batch, loss_out, dq1, dq2,
                     dlogp);
  return aa_launch_status();
}

int aa_sac_alpha_loss(const float* logp, const float* weights, const float* log_alpha_dev,
                      float target_entropy, int32_t use_log_alpha, float loss_weight, int64_t B,
                      float global_batch, float* loss_out, float* grad_out, void* stream) {
  if (!logp || !log_alpha_dev || !loss_out || B <= 0 || !(global_batch > 0.f))
    return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_sac_alpha_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp,
                     weights, log_alpha_dev, target_entropy, use_log_alpha, loss_weight, B,
                     global_batch, loss_out, grad_out);
  return aa_launch_status();
}

int aa_sac_alpha_step(const float* logp, const float* weights, float* log_alpha_dev,
                      float target_entropy, int32_t use_log_alpha, float loss_weight, int64_t B,
                      float global_batch, float* loss_out, float* grad_out, float* adam_m,
                      float* adam_v, int64_t* adam_steps_dev, float lr, float beta1, float beta2,
                      float eps, const float* critic_loss, const float* actor_loss,
                      float* packed4, void* stream) {
  if (!logp || !log_alpha_dev || !loss_out || !grad_out || !adam_m || !adam_v || !adam_steps_dev ||
      !critic_loss || !actor_loss || !packed4 || B <= 0 || !(global_batch > 0.f))
    return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_sac_alpha_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp,
                     weights, log_alpha_dev, target_entropy, use_log_alpha, loss_weight, B,
                     global_batch, loss_out, grad_out, adam_m, adam_v, adam_steps_dev, lr, beta1,
                     beta2, eps, critic_loss, actor_loss, packed4);
  return aa_launch_status();
}

}  // extern "C"
