// PPO clipped-surrogate + value + entropy loss, forward and backward in one pass, for the
// diagonal-Normal policy head built by PPOActorNetwork (loc = mean + mag*tanh(z),
// scale = softplus(state-independent bias)).  -ffp-contract=off.
//   PPOAgent.get_loss                    tf_agents/agents/ppo/ppo_agent.py:481-615
//   policy_gradient_loss                 tf_agents/agents/ppo/ppo_agent.py:1329-1512
//   value_estimation_loss                tf_agents/agents/ppo/ppo_agent.py:1203-1327
//   entropy_regularization_loss          tf_agents/agents/ppo/ppo_agent.py:1159-1201
//   common.log_probability / entropy     tf_agents/utils/common.py:682-756
//   PPOActorNetwork (tanh_and_scale, softplus bias)  tf_agents/agents/ppo/ppo_actor_network.py:30-113
//   TFP MultivariateNormalDiag log_prob / entropy closed forms (third-party, restated)
// Every term is mean over ALL N elements of (term * weight) -- masked entries stay in the
// denominator (ppo_agent_test.py:701-708) -- divided additionally by the replica count.
#include "common.h"
#include "agents_amd.h"
#include "mlp_small_common.h"

#define AA_PPO_MAXD 64
#define AA_PPO_P 256
#define AA_HALF_LOG_2PI 0.91893853320467274178f

__device__ static inline float aa_softplus(float x) {
  // tf.math.softplus: log(exp(x) + 1), evaluated stably
  return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x));
}

__global__ void __launch_bounds__(256)
aa_ppo_loss_kernel(const float* __restrict__ z, const float* __restrict__ std_bias,
                   const float* __restrict__ act_mean, const float* __restrict__ act_mag,
                   const float* __restrict__ actions, const float* __restrict__ old_logp,
                   const float* __restrict__ adv, const float* __restrict__ returns,
                   const float* __restrict__ vpred, const float* __restrict__ old_vpred,
                   const float* __restrict__ weights, int64_t N, int D, float clip_eps,
                   float value_clip, float c_v, float c_e, float denom, float logp_clip,
                   float* __restrict__ dz, float* __restrict__ dbias_elem,
                   float* __restrict__ dv, float* __restrict__ partial) {
  __shared__ float red[16];
  __shared__ float s_scale[AA_PPO_MAXD], s_dsp[AA_PPO_MAXD], s_logs[AA_PPO_MAXD];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float b = std_bias[d];
    const float sc = aa_softplus(b);
    s_scale[d] = sc;
    s_logs[d] = logf(sc);
    s_dsp[d] = 1.0f / (1.0f + expf(-b));  // d softplus / d bias = sigmoid
  }
  __syncthreads();
  float sum_pg = 0.f, sum_v = 0.f, sum_ent = 0.f, sum_clip = 0.f, sum_entw = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const float w = weights != nullptr ? weights[i] : 1.0f;
    // ---- log-prob & entropy of the current policy ------------------------------------------
    float lp = 0.f, ent = 0.f;
    for (int d = 0; d < D; ++d) {
      const float zz = z[i * D + d];
      const float th = act_mag != nullptr ? tanhf(zz) : zz;
      const float loc = act_mag != nullptr ? act_mean[d] + act_mag[d] * th : zz;
      const float sc = s_scale[d];
      const float xs = actions[i * D + d] / sc, ls = loc / sc;
      const float diff = xs - ls;
      lp += -0.5f * (diff * diff) - (AA_HALF_LOG_2PI + s_logs[d]);
      ent += 0.5f + AA_HALF_LOG_2PI + s_logs[d];
    }
    float lp_c = lp;
    bool lp_live = true;
    if (logp_clip > 0.f) {
      lp_c = fminf(fmaxf(lp, -logp_clip), logp_clip);
      lp_live = (lp >= -logp_clip) && (lp <= logp_clip);
    }
    // ---- clipped surrogate ------------------------------------------------------------------
    const float a = adv[i];
    const float ratio = expf(lp_c - old_logp[i]);
    const float ratio_c = fminf(fmaxf(ratio, 1.0f - clip_eps), 1.0f + clip_eps);
    const float obj = ratio * a, obj_c = ratio_c * a;
    float pg;
    bool grad_through_ratio;
    if (clip_eps > 0.f) {
      pg = -fminf(obj, obj_c);
      grad_through_ratio = obj <= obj_c;  // tf.minimum routes the gradient to x when x <= y
      sum_clip += fabsf(ratio - 1.0f) > clip_eps ? 1.0f : 0.0f;
    } else {
      pg = -obj;
      grad_through_ratio = true;
    }
    const float pg_w = (w == 0.f) ? 0.f : pg * w;
    sum_pg += pg_w;
    // d(sum pg*w/denom)/d lp
    float dlp = 0.f;
    if (grad_through_ratio && lp_live) dlp = -(a * ratio) * w / denom;
    // ---- value loss --------------------------------------------------------------------------
    const float R = returns[i], V = vpred[i];
    float verr = (R - V) * (R - V);
    float dverr_dV = -2.0f * (R - V);
    if (value_clip > 0.f && old_vpred != nullptr) {
      const float ov = old_vpred[i];
      const float dlt = V - ov;
      const float dc = fminf(fmaxf(dlt, -value_clip), value_clip);
      const float Vc = ov + dc;
      const float verr_c = (R - Vc) * (R - Vc);
      if (verr_c > verr) {  // tf.maximum: gradient to x when x >= y
        verr = verr_c;
        const bool live = dlt >= -value_clip && dlt <= value_clip;
        dverr_dV = live ? -2.0f * (R - Vc) : 0.f;
      }
    }
    sum_v += (w == 0.f) ? 0.f : verr * w;
    dv[i] = c_v * dverr_dV * w / denom;
    // ---- entropy regularisation ---------------------------------------------------------------
    sum_ent += (w == 0.f) ? 0.f : (-ent) * w;
    sum_entw += ent * w;
    const float dent = (c_e > 0.f) ? (-c_e * w / denom) : 0.f;  // d loss / d entropy
    // ---- back through the Normal head ----------------------------------------------------------
    for (int d = 0; d < D; ++d) {
      const float zz = z[i * D + d];
      const float th = act_mag != nullptr ? tanhf(zz) : zz;
      const float loc = act_mag != nullptr ? act_mean[d] + act_mag[d] * th : zz;
      const float sc = s_scale[d];
      const float diff = actions[i * D + d] - loc;
      const float dlp_dloc = diff / (sc * sc);
      const float dlp_dsc = (diff * diff) / (sc * sc * sc) - 1.0f / sc;
      float dloc_dz = 1.0f;
      if (act_mag != nullptr) dloc_dz = act_mag[d] * (1.0f - th * th);
      dz[i * D + d] = dlp * dlp_dloc * dloc_dz;
      dbias_elem[i * D + d] = (dlp * dlp_dsc + dent * (1.0f / sc)) * s_dsp[d];
    }
  }
  float t;
  t = aa_block_sum(sum_pg, red);   if (threadIdx.x == 0) partial[blockIdx.x * 5 + 0] = t;
  t = aa_block_sum(sum_v, red);    if (threadIdx.x == 0) partial[blockIdx.x * 5 + 1] = t;
  t = aa_block_sum(sum_ent, red);  if (threadIdx.x == 0) partial[blockIdx.x * 5 + 2] = t;
  t = aa_block_sum(sum_clip, red); if (threadIdx.x == 0) partial[blockIdx.x * 5 + 3] = t;
  t = aa_block_sum(sum_entw, red); if (threadIdx.x == 0) partial[blockIdx.x * 5 + 4] = t;
}

// stats: [0] policy_gradient_loss [1] value_estimation_loss [2] entropy_regularization_loss
//        [3] clip_fraction [4] mean(entropy*weights) [5] total (0+1+2)
__global__ void aa_ppo_finish_kernel(const float* __restrict__ partial, int P, float denom,
                                     float n_elems, float c_v, float c_e,
                                     float* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < P; ++p)
    for (int k = 0; k < 5; ++k) s[k] += partial[p * 5 + k];
  const float pg = s[0] / denom;
  const float v = (s[1] / denom) * c_v;
  const float e = c_e > 0.f ? (s[2] / denom) * c_e : 0.f;
  stats[0] = pg;
  stats[1] = v;
  stats[2] = e;
  stats[3] = s[3] / n_elems;
  stats[4] = s[4] / n_elems;
  stats[5] = pg + v + e;
  stats[6] = 0.f;
  stats[7] = 0.f;
}

__global__ void aa_pack_small_kernel(const float* __restrict__ src, int n,
                                     const float* __restrict__ addend, int add_at,
                                     float* __restrict__ out) {
  const int i = threadIdx.x;
  const float ad = addend != nullptr ? addend[0] : 0.f;
  if (i < n) out[i] = i == add_at ? src[i] + ad : src[i];
  if (i == n) out[n] = ad;
}

__global__ void __launch_bounds__(256)
aa_add_strided_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                      int64_t ldb, int64_t rows, int64_t cols, float* __restrict__ out) {
  const int64_t n = rows * cols, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / cols, c = i - r * cols;
    out[i] = a[r * lda + c] + b[r * ldb + c];
  }
}

__global__ void aa_pack_sum3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                    const float* __restrict__ c, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    const float x = a[0], y = b[0], z = c[0];
    out[0] = (x + y) + z;
    out[1] = x; out[2] = y; out[3] = z;
  }
}

__global__ void __launch_bounds__(256)
aa_axpy_kernel(float* __restrict__ g, const float* __restrict__ p, int64_t n, float c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    g[i] = g[i] + c * p[i];
}


// =============================================================================================
// General diagonal-Normal form: the actor network hands over loc[N,D] and scale[N,D] (any head:
// PPOActorNetwork, NormalProjectionNetwork, the reference tests' DummyActorNet) and gets the
// gradients wrt both back.  Adds the KL penalty terms of PPOAgent:
//   kl_penalty_loss / kl_cutoff_loss / adaptive_kl_loss   ppo_agent.py:1514-1640
//   _kl_divergence -> ppo_utils.nested_kl_divergence        ppo_utils.py:194-227
//   TFP kl_normal_normal closed form (third-party, restated):
//     kl(a||b) = 0.5*((mu_a - mu_b)/sigma_b)^2 + 0.5*expm1(2*(log sigma_a - log sigma_b))
//                - (log sigma_a - log sigma_b)
// Three launches: forward sums -> scalars (+ the KL gradient coefficient, which depends on the
// batch mean) -> per-sample gradients.
// =============================================================================================
#define AA_PPO_NSUM 6   // pg, value, entropy, clip count, entropy*w, kl*w

struct PpoDistArgs {
  const float *loc, *scale, *old_loc, *old_scale, *actions, *old_logp, *adv, *returns, *vpred,
      *old_vpred, *weights;
  int64_t N;
  int D;
  float clip_eps, value_clip, c_v, c_e, denom, logp_clip;
};

__device__ static inline void aa_ppo_sample_terms(const PpoDistArgs& a, int64_t i, float* lp_out,
                                                  float* ent_out, float* kl_out) {
  float lp = 0.f, ent = 0.f, kl = 0.f;
  for (int d = 0; d < a.D; ++d) {
    const float mu = a.loc[i * a.D + d], sc = a.scale[i * a.D + d];
    const float ls = logf(sc);
    const float diff = a.actions[i * a.D + d] / sc - mu / sc;
    lp += -0.5f * (diff * diff) - (AA_HALF_LOG_2PI + ls);
    ent += 0.5f + AA_HALF_LOG_2PI + ls;
    if (a.old_loc != nullptr) {
      const float mo = a.old_loc[i * a.D + d], so = a.old_scale[i * a.D + d];
      const float dl = logf(so) - ls;
      const float dm = mo / sc - mu / sc;
      kl += 0.5f * (dm * dm) + 0.5f * expm1f(2.0f * dl) - dl;
    }
  }
  *lp_out = lp;
  *ent_out = ent;
  *kl_out = kl;
}

__global__ void __launch_bounds__(256)
aa_ppo_dist_fwd_kernel(PpoDistArgs a, float* __restrict__ partial) {
  __shared__ float red[16];
  float s[AA_PPO_NSUM] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += stride) {
    const float w = a.weights != nullptr ? a.weights[i] : 1.0f;
    float lp, ent, kl;
    aa_ppo_sample_terms(a, i, &lp, &ent, &kl);
    if (a.logp_clip > 0.f) lp = fminf(fmaxf(lp, -a.logp_clip), a.logp_clip);
    const float adv = a.adv[i];
    const float ratio = expf(lp - a.old_logp[i]);
    const float ratio_c = fminf(fmaxf(ratio, 1.0f - a.clip_eps), 1.0f + a.clip_eps);
    float pg;
    if (a.clip_eps > 0.f) {
      pg = -fminf(ratio * adv, ratio_c * adv);
      s[3] += fabsf(ratio - 1.0f) > a.clip_eps ? 1.0f : 0.0f;
    } else {
      pg = -(ratio * adv);
    }
    s[0] += (w == 0.f) ? 0.f : pg * w;
    const float R = a.returns[i], V = a.vpred[i];
    float verr = (R - V) * (R - V);
    if (a.value_clip > 0.f && a.old_vpred != nullptr) {
      const float ov = a.old_vpred[i];
      const float Vc = ov + fminf(fmaxf(V - ov, -a.value_clip), a.value_clip);
      verr = fmaxf(verr, (R - Vc) * (R - Vc));
    }
    s[1] += (w == 0.f) ? 0.f : verr * w;
    s[2] += (w == 0.f) ? 0.f : (-ent) * w;
    s[4] += ent * w;
    s[5] += kl * w;
  }
  for (int k = 0; k < AA_PPO_NSUM; ++k) {
    const float t = aa_block_sum(s[k], red);
    if (threadIdx.x == 0) partial[blockIdx.x * AA_PPO_NSUM + k] = t;
  }
}

// stats: [0] policy_gradient_loss [1] value_estimation_loss [2] entropy_regularization_loss
// [3] clip_fraction [4] mean(entropy*w) [5] kl_penalty_loss [6] total [7] mean(kl*w)
// [8] d loss / d mean(kl*w) [9] adaptive_kl_loss [10] kl_cutoff_loss
__global__ void aa_ppo_dist_finish_kernel(const float* __restrict__ partial, int P, float denom,
                                          float n_elems, float c_v, float c_e,
                                          const float* __restrict__ kl_beta, float kl_cutoff_coef,
                                          float kl_cutoff, float* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s[AA_PPO_NSUM] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < P; ++p)
    for (int k = 0; k < AA_PPO_NSUM; ++k) s[k] += partial[p * AA_PPO_NSUM + k];
  const float pg = s[0] / denom;
  const float v = (s[1] / denom) * c_v;
  const float e = c_e > 0.f ? (s[2] / denom) * c_e : 0.f;
  const float mean_kl = s[5] / n_elems;
  const float beta = kl_beta != nullptr ? kl_beta[0] : 0.f;
  const float adaptive = beta * mean_kl;
  float cutoff_loss = 0.f, dcut = 0.f;
  if (kl_cutoff_coef > 0.f && kl_cutoff > 0.f) {
    const float over = fmaxf(mean_kl - kl_cutoff, 0.0f);
    cutoff_loss = kl_cutoff_coef * (over * over);
    dcut = 2.0f * kl_cutoff_coef * over;
  }
  stats[0] = pg;
  stats[1] = v;
  stats[2] = e;
  stats[3] = s[3] / n_elems;
  stats[4] = s[4] / n_elems;
  stats[5] = adaptive + cutoff_loss;
  stats[6] = pg + v + e + adaptive + cutoff_loss;
  stats[7] = mean_kl;
  stats[8] = beta + dcut;
  stats[9] = adaptive;
  stats[10] = cutoff_loss;
}

__global__ void __launch_bounds__(256)
aa_ppo_dist_bwd_kernel(PpoDistArgs a, const float* __restrict__ stats, float n_elems,
                       float* __restrict__ dloc, float* __restrict__ dscale,
                       float* __restrict__ dv) {
  const float kl_coef = stats[8] / n_elems;  // d loss / d (kl_i * w_i)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += stride) {
    const float w = a.weights != nullptr ? a.weights[i] : 1.0f;
    float lp, ent, kl;
    aa_ppo_sample_terms(a, i, &lp, &ent, &kl);
    bool lp_live = true;
    float lp_c = lp;
    if (a.logp_clip > 0.f) {
      lp_c = fminf(fmaxf(lp, -a.logp_clip), a.logp_clip);
      lp_live = (lp >= -a.logp_clip) && (lp <= a.logp_clip);
    }
    const float adv = a.adv[i];
    const float ratio = expf(lp_c - a.old_logp[i]);
    const float ratio_c = fminf(fmaxf(ratio, 1.0f - a.clip_eps), 1.0f + a.clip_eps);
    bool through = true;
    if (a.clip_eps > 0.f) through = (ratio * adv) <= (ratio_c * adv);  // tf.minimum grad rule
    float dlp = 0.f;
    if (through && lp_live) dlp = -(adv * ratio) * w / a.denom;
    const float R = a.returns[i], V = a.vpred[i];
    float dverr = -2.0f * (R - V);
    if (a.value_clip > 0.f && a.old_vpred != nullptr) {
      const float ov = a.old_vpred[i];
      const float dlt = V - ov;
      const float Vc = ov + fminf(fmaxf(dlt, -a.value_clip), a.value_clip);
      if ((R - Vc) * (R - Vc) > (R - V) * (R - V)) {  // tf.maximum: gradient to x when x >= y
        const bool live = dlt >= -a.value_clip && dlt <= a.value_clip;
        dverr = live ? -2.0f * (R - Vc) : 0.f;
      }
    }
    dv[i] = a.c_v * dverr * w / a.denom;
    const float dent = (a.c_e > 0.f) ? (-a.c_e * w / a.denom) : 0.f;
    const float dkl = (a.old_loc != nullptr) ? kl_coef * w : 0.f;
    for (int d = 0; d < a.D; ++d) {
      const float mu = a.loc[i * a.D + d], sc = a.scale[i * a.D + d];
      const float diff = a.actions[i * a.D + d] - mu;
      float gl = dlp * (diff / (sc * sc));
      float gs = dlp * ((diff * diff) / (sc * sc * sc) - 1.0f / sc) + dent * (1.0f / sc);
      if (a.old_loc != nullptr) {
        const float mo = a.old_loc[i * a.D + d], so = a.old_scale[i * a.D + d];
        const float dm = mu - mo;
        gl += dkl * (dm / (sc * sc));
        gs += dkl * (-(dm * dm) / (sc * sc * sc) - (so * so) / (sc * sc * sc) + 1.0f / sc);
      }
      dloc[i * a.D + d] = gl;
      dscale[i * a.D + d] = gs;
    }
  }
}

// PPOActorNetwork head (ppo_actor_network.py:42-113): loc = mean + mag*tanh(z) (or z when the
// spec is unbounded / mean,mag null), scale = softplus(bias) broadcast over the batch.
__global__ void __launch_bounds__(256)
aa_ppo_head_fwd_kernel(const float* __restrict__ z, const float* __restrict__ std_bias,
                       const float* __restrict__ act_mean, const float* __restrict__ act_mag,
                       int64_t N, int D, float* __restrict__ loc, float* __restrict__ scale) {
  const int64_t total = N * D, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int d = (int)(e % D);
    const float zz = z[e];
    loc[e] = act_mag != nullptr ? act_mean[d] + act_mag[d] * tanhf(zz) : zz;
    scale[e] = aa_softplus(std_bias[d]);
  }
}
// aa_ppo_head_fwd_kernel + aa_normal_sample_kernel (+ the clip to the action spec and the advance of
// the Philox call counter) in one launch: the collect policy's PPOPolicy._action
// (policies/actor_policy.py through agents/ppo/ppo_policy.py) per environment step.  Element e
// draws from Philox(counter = (e, call), key = seed) exactly as aa_normal_sample does.
__global__ void __launch_bounds__(256)
aa_ppo_head_fwd_sample_kernel(const float* __restrict__ z, const float* __restrict__ std_bias,
                              const float* __restrict__ act_mean,
                              const float* __restrict__ act_mag, int64_t N, int D,
                              float* __restrict__ loc, float* __restrict__ scale,
                              uint32_t seed_lo, uint32_t seed_hi, int64_t* __restrict__ call_counter,
                              int64_t* __restrict__ arrival, const float* __restrict__ clip_lo,
                              const float* __restrict__ clip_hi, float* __restrict__ action) {
  const uint64_t call = (uint64_t)call_counter[0];
  const int64_t total = N * D, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int d = (int)(e % D);
    const float zz = z[e];
    const float l = act_mag != nullptr ? act_mean[d] + act_mag[d] * tanhf(zz) : zz;
    const float sc = aa_softplus(std_bias[d]);
    loc[e] = l;
    scale[e] = sc;
    const Philox4 r = philox4x32_10((uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), seed_lo, seed_hi);
    const float u1 = 1.0f - aa_u01(r.x);  // (0, 1]
    const float u2 = aa_u01(r.y);
    const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    float a = l + sc * eps;
    if (clip_lo != nullptr) a = fmaxf(fminf(a, clip_hi[d]), clip_lo[d]);
    action[e] = a;
  }
  aa_advance_when_all_done(call_counter, arrival, 1, gridDim.x);
}
// ---- a whole collect-policy step in ONE launch ---------------------------------------------------
// PPOPolicy._action (agents/ppo/ppo_policy.py:231-241 on policies/actor_policy.py): normalise the
// observation, run the actor body and the value body on it, apply the actor head, draw the action,
// clip it, advance the Philox counter.  blockIdx.y = 0: actor body + head + draw; 1: value body.
// A workgroup owns TILE samples and walks its network's layers with activations and the layer's
// weights in LDS -- aa_mlp_small_fwd_kernel's arithmetic (explicit fmaf), aa_norm_apply_kernel's
// normalisation, aa_ppo_head_fwd_kernel's head and aa_normal_sample_kernel's draw, each as in the
// launch it replaces (this file is built without FMA contraction, as normalizer.hip is).
struct AaPolicyStep {
  const float* x;
  int64_t ldx, B;
  const float* nrm_mean;      // all three NULL: no normalisation
  const float* nrm_var_num;
  const float* nrm_var_den;   // nullable (EMA normaliser: var_num is the variance)
  float nrm_eps, nrm_clip;
  int has_norm;
  const float* params[2];
  AaMlpDesc d[2];
  float* value_out;           // [B] (network 1's single output)
  const float* std_bias;
  const float* act_mean;
  const float* act_mag;
  int D;
  float* loc;
  float* scale;
  uint32_t seed_lo, seed_hi;
  int64_t* call_counter;
  int64_t* arrival;
  const float* clip_lo;
  const float* clip_hi;
  float* action;
};

template <int TILE>
__global__ void __launch_bounds__(256) aa_ppo_policy_step_kernel(AaPolicyStep P) {
  constexpr int QN = 256 / TILE, OPT = AA_MLP_MAXW / QN;
  __shared__ __attribute__((aligned(16))) float Ws[AA_MLP_MAXW * AA_MLP_MAXW];
  __shared__ __attribute__((aligned(16))) float bs[AA_MLP_MAXW];
  __shared__ __attribute__((aligned(16))) float h[2][TILE][AA_MLP_MAXW + 4];
  const int g = blockIdx.y;
  const float* __restrict__ params = P.params[g];
  const AaMlpDesc& d = P.d[g];
  const int s = threadIdx.x / QN, q = threadIdx.x % QN;
  const int64_t b = (int64_t)blockIdx.x * TILE + s;
  // the Philox call number is read before anything else (every workgroup has it by the time the
  // last one reports in: aa_advance_when_all_done below)
  const uint64_t call = g == 0 ? (uint64_t)P.call_counter[0] : 0ull;
  for (int i = threadIdx.x; i < TILE * AA_MLP_MAXW; i += blockDim.x) {
    const int ss = i >> 6, k = i & 63;
    const int64_t bb = (int64_t)blockIdx.x * TILE + ss;
    float v = 0.f;
    if (bb < P.B && k < d.dims[0]) {
      v = P.x[bb * P.ldx + k];
      if (P.has_norm) {
        float var = P.nrm_var_num[k];
        if (P.nrm_var_den != nullptr) var = var / P.nrm_var_den[k];
        const float inv = 1.0f / sqrtf(var + P.nrm_eps);
        const float m = P.nrm_mean != nullptr ? P.nrm_mean[k] : 0.0f;
        v = v * inv + (-m * inv);
        if (P.nrm_clip > 0.f) v = fminf(fmaxf(v, -P.nrm_clip), P.nrm_clip);
      }
    }
    h[0][ss][k] = v;
  }
  int cur = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    const int n_in = d.dims[l], n_out = d.dims[l + 1];
    __syncthreads();
    aa_mlp_stage_w(params, d.k_off[l], d.b_off[l], n_in, n_out, Ws, bs);
    __syncthreads();
    float acc[OPT];
#pragma unroll
    for (int j = 0; j < OPT; ++j) acc[j] = bs[OPT * q + j];
    for (int k = 0; k < n_in; ++k) {
      const float hk = h[cur][s][k];
      const float4* wr = reinterpret_cast<const float4*>(Ws + k * AA_MLP_MAXW + OPT * q);
#pragma unroll
      for (int v = 0; v < OPT / 4; ++v) {
        const float4 w = wr[v];
        acc[4 * v + 0] = fmaf(hk, w.x, acc[4 * v + 0]);
        acc[4 * v + 1] = fmaf(hk, w.y, acc[4 * v + 1]);
        acc[4 * v + 2] = fmaf(hk, w.z, acc[4 * v + 2]);
        acc[4 * v + 3] = fmaf(hk, w.w, acc[4 * v + 3]);
      }
    }
    const bool last = l == d.n_layers - 1;
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
      const int col = OPT * q + j;
      const float v = aa_mlp_act(acc[j], d.acts[l]);
      h[cur ^ 1][s][col] = col < n_out ? v : 0.f;
      if (g == 1 && last && b < P.B && col == 0) P.value_out[b] = v;
    }
    cur ^= 1;
  }
  if (g != 0) return;
  __syncthreads();      // h[cur][sample][dimension] = the actor body's output z of this tile
  const int D = P.D;
  for (int i = threadIdx.x; i < TILE * D; i += blockDim.x) {
    const int ss = i / D, dd = i - ss * D;
    const int64_t bb = (int64_t)blockIdx.x * TILE + ss;
    if (bb >= P.B) continue;
    const int64_t e = bb * D + dd;
    const float zz = h[cur][ss][dd];
    const float l = P.act_mag != nullptr ? P.act_mean[dd] + P.act_mag[dd] * tanhf(zz) : zz;
    const float sc = aa_softplus(P.std_bias[dd]);
    P.loc[e] = l;
    P.scale[e] = sc;
    const Philox4 r = philox4x32_10((uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), P.seed_lo, P.seed_hi);
    const float u1 = 1.0f - aa_u01(r.x);  // (0, 1]
    const float u2 = aa_u01(r.y);
    const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    float a = l + sc * eps;
    if (P.clip_lo != nullptr) a = fmaxf(fminf(a, P.clip_hi[dd]), P.clip_lo[dd]);
    P.action[e] = a;
  }
  aa_advance_when_all_done(P.call_counter, P.arrival, 1, gridDim.x);
}

__global__ void __launch_bounds__(256)
aa_ppo_head_bwd_kernel(const float* __restrict__ z, const float* __restrict__ std_bias,
                       const float* __restrict__ act_mag, const float* __restrict__ dloc,
                       const float* __restrict__ dscale, int64_t N, int D,
                       float* __restrict__ dz, float* __restrict__ dbias_elem) {
  const int64_t total = N * D, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int d = (int)(e % D);
    float g = dloc[e];
    if (act_mag != nullptr) {
      const float th = tanhf(z[e]);
      g *= act_mag[d] * (1.0f - th * th);
    }
    dz[e] = g;
    dbias_elem[e] = dscale[e] * (1.0f / (1.0f + expf(-std_bias[d])));  // softplus' = sigmoid
  }
}

// log N(x; loc, scale) summed over D (common.log_probability, utils/common.py:682-717)
__global__ void __launch_bounds__(256)
aa_normal_log_prob_kernel(const float* __restrict__ loc, const float* __restrict__ scale,
                          const float* __restrict__ x, int64_t N, int D,
                          float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    float lp = 0.f;
    for (int d = 0; d < D; ++d) {
      const float sc = scale[i * D + d];
      const float diff = x[i * D + d] / sc - loc[i * D + d] / sc;
      lp += -0.5f * (diff * diff) - (AA_HALF_LOG_2PI + logf(sc));
    }
    out[i] = lp;
  }
}

// action = loc + scale * eps, eps ~ N(0,1) by Box-Muller on the package's Philox stream:
// counter = (element index lo, hi, call counter lo, hi), key = seed; u1 from word 0, u2 from 1.
__global__ void __launch_bounds__(256)
aa_normal_sample_kernel(const float* __restrict__ loc, const float* __restrict__ scale,
                        int64_t n, uint32_t seed_lo, uint32_t seed_hi,
                        const int64_t* __restrict__ call_counter, float* __restrict__ out) {
  const uint64_t call = (uint64_t)call_counter[0];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), seed_lo, seed_hi);
    const float u1 = 1.0f - aa_u01(r.x);  // (0, 1]
    const float u2 = aa_u01(r.y);
    const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    out[i] = loc[i] + scale[i] * eps;
  }
}

// RandomTFPolicy on a bounded continuous spec: out[i, d] = lo[d] + (hi[d] - lo[d]) * u, u in [0, 1)
// from Philox(counter = (element index, call counter), key = seed), word 0.
__global__ void __launch_bounds__(256)
aa_uniform_sample_kernel(const float* __restrict__ lo, const float* __restrict__ hi, int64_t N,
                         int D, uint32_t seed_lo, uint32_t seed_hi,
                         const int64_t* __restrict__ call_counter, float* __restrict__ out) {
  const uint64_t call = (uint64_t)call_counter[0];
  const int64_t total = N * D, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int d = (int)(i % D);
    const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), seed_lo, seed_hi);
    const float l = lo[d], h = hi[d];
    float v = l + (h - l) * aa_u01(r.x);
    out[i] = v < h ? v : l;       // rounding must not reach the open end of [lo, hi)
  }
}

// discounts for the return / GAE scans: discount * gamma * (next_step_type != LAST)
// (ppo_agent.py:630-676, utils/common.py:883-895), over the first T of T+1 columns.
__global__ void __launch_bounds__(256)
aa_ppo_discounts_kernel(const float* __restrict__ discount, const int32_t* __restrict__ next_st,
                        float gamma, int64_t B, int64_t T1, float* __restrict__ out) {
  const int64_t T = T1 - 1, total = B * T, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t b = e / T, t = e - b * T;
    const float d = discount[b * T1 + t] * gamma;
    out[e] = d * (next_st[b * T1 + t] != 2 ? 1.0f : 0.0f);
  }
}
// weights * ~is_boundary * ~(return == 0 & advantage == 0)   (ppo_utils.py:35-59)
__global__ void __launch_bounds__(256)
aa_ppo_mask_kernel(const int32_t* __restrict__ step_type, const float* __restrict__ ret,
                   const float* __restrict__ adv, const float* __restrict__ weights, int64_t n,
                   float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool valid = (step_type[i] != 2) && !(ret[i] == 0.f && adv[i] == 0.f);
    const float m = valid ? 1.0f : 0.0f;
    out[i] = weights != nullptr ? weights[i] * m : m;
  }
}
// adaptive KL beta update (ppo_agent.py:1642-1690): x 1/1.5 below target*(1-tol), x 1.5 above
// target*(1+tol), clipped to [1e-15, 1e17].
__global__ void aa_ppo_update_beta_kernel(const float* __restrict__ mean_kl, float target,
                                          float tol, float* __restrict__ beta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float m = mean_kl[0];
  float f = 1.0f;
  if (m < target * (1.0f - tol)) f = 1.0f / 1.5f;
  else if (m > target * (1.0f + tol)) f = 1.5f;
  beta[0] = fminf(fmaxf(beta[0] * f, 10e-16f), 10e16f);
}

static inline unsigned aa_ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" {

// stats must hold 8 + 5*256 floats (8 results + per-block partials).
int aa_ppo_loss(const float* z, const float* std_bias, const float* act_mean,
                const float* act_mag, const float* actions, const float* old_logp,
                const float* adv, const float* returns, const float* vpred,
                const float* old_vpred, const float* weights, int64_t N, int32_t D,
                float clip_eps, float value_clip, float c_v, float c_e, float denom,
                float logp_clip, int32_t flags, float* dz, float* dbias_elem, float* dv,
                float* stats, void* stream) {
  (void)flags;
  if (!z || !std_bias || !actions || !old_logp || !adv || !returns || !vpred || !dz ||
      !dbias_elem || !dv || !stats)
    return AA_ERR_INVALID;
  if (N <= 0 || D <= 0 || D > AA_PPO_MAXD || !(denom > 0.f)) return AA_ERR_INVALID;
  if ((act_mean == nullptr) != (act_mag == nullptr)) return AA_ERR_INVALID;
  int P = (int)((N + 255) / 256);
  if (P > AA_PPO_P) P = AA_PPO_P;
  hipStream_t st = (hipStream_t)stream;
  float* partial = stats + 8;
  hipLaunchKernelGGL(aa_ppo_loss_kernel, dim3(P), dim3(256), 0, st, z, std_bias, act_mean, act_mag,
                     actions, old_logp, adv, returns, vpred, old_vpred, weights, N, (int)D,
                     clip_eps, value_clip, c_v, c_e, denom, logp_clip, dz, dbias_elem, dv,
                     partial);
  hipLaunchKernelGGL(aa_ppo_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, P,
                     denom, (float)N, c_v, c_e, stats);
  return aa_launch_status();
}

// out[0..n) = src[0..n); out[n] = addend ? *addend : 0; out[add_at] += out[n].  One launch that
// gives a LossInfo storage of its own (the loss kernels' stats vector is overwritten by the next
// evaluation) and folds the regularisation term into the total -- instead of clone + add + clone.
int aa_pack_small_f32(const float* src, int32_t n, const float* addend, int32_t add_at, float* out,
                      void* stream) {
  if (!src || !out || n <= 0 || n > 255 || add_at < 0 || add_at >= n) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_pack_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, n,
                     addend, add_at, out);
  return aa_launch_status();
}

// out[r, c] = a[r * lda + c] + b[r * ldb + c], out dense [rows, cols]: the action gradient through
// the two critics of SAC (column slices of their input-gradient buffers) summed into the buffer
// the head's backward reads
// ---- several strided row copies in one launch ------------------------------------------------
// dst_i[r, c] = src_i[r, c] for up to 8 (src, dst) pairs of 4-byte-element matrices with `rows`
// rows each: the AsTransition slices of a [B, 2, ...] SAC batch (observation / action / reward /
// discount of frame 0, observation of frame 1) and the [observation | action] inputs of the twin
// critics are assembled by ONE launch instead of one torch copy kernel per slice and per critic
// (17 per train step: agents/sac/sac_agent.py:533-640, data_converter.py:300-380).
struct AaCopySegs {
  int n;
  const uint32_t* src[8];
  uint32_t* dst[8];
  int64_t src_pitch[8], dst_pitch[8];   // in elements
  int cols[8];
};

__global__ void __launch_bounds__(256) aa_copy_segments_kernel(AaCopySegs S, int64_t rows) {
  const int seg = blockIdx.x;
  const uint32_t* src = S.src[seg];
  uint32_t* dst = S.dst[seg];
  const int cols = S.cols[seg];
  const int64_t sp = S.src_pitch[seg], dp = S.dst_pitch[seg];
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y)
    for (int c = threadIdx.x; c < cols; c += 256) dst[r * dp + c] = src[r * sp + c];
}

extern "C" int aa_copy_segments(const void* const* src_h, void* const* dst_h,
                                const int64_t* src_pitch_h, const int64_t* dst_pitch_h,
                                const int32_t* cols_h, int32_t n_segments, int64_t rows,
                                void* stream) {
  if (n_segments < 1 || n_segments > 8 || rows <= 0) return AA_ERR_INVALID;
  AaCopySegs S;
  S.n = n_segments;
  for (int i = 0; i < n_segments; ++i) {
    if (src_h[i] == nullptr || dst_h[i] == nullptr || cols_h[i] <= 0 || src_pitch_h[i] < cols_h[i] ||
        dst_pitch_h[i] < cols_h[i])
      return AA_ERR_INVALID;
    if ((((uintptr_t)src_h[i] | (uintptr_t)dst_h[i]) & 3) != 0) return AA_ERR_INVALID;
    S.src[i] = (const uint32_t*)src_h[i];
    S.dst[i] = (uint32_t*)dst_h[i];
    S.src_pitch[i] = src_pitch_h[i];
    S.dst_pitch[i] = dst_pitch_h[i];
    S.cols[i] = cols_h[i];
  }
  const unsigned gy = rows < 256 ? (unsigned)rows : 256u;
  hipLaunchKernelGGL(aa_copy_segments_kernel, dim3((unsigned)n_segments, gy), dim3(256), 0,
                     (hipStream_t)stream, S, rows);
  return aa_launch_status();
}

int aa_add_strided_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows,
                       int64_t cols, float* out, void* stream) {
  if (!a || !b || !out || rows <= 0 || cols <= 0 || lda < cols || ldb < cols) return AA_ERR_INVALID;
  int64_t blocks = (rows * cols + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(aa_add_strided_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, a, lda, b, ldb, rows, cols, out);
  return aa_launch_status();
}

// out = [a + b + c, a, b, c]: SacAgent's LossInfo (total, critic, actor, alpha) in one launch
int aa_pack_sum3_f32(const float* a, const float* b, const float* c, float* out4, void* stream) {
  if (!a || !b || !c || !out4) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_pack_sum3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, out4);
  return aa_launch_status();
}

// g += c * p   (L2 regularisation gradients; keras regularizers / tf.nn.l2_loss)
int aa_add_l2_grad(float* g, const float* p, int64_t n, float c, void* stream) {
  if (!g || !p || n <= 0) return AA_ERR_INVALID;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(aa_axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g,
                     p, n, c);
  return aa_launch_status();
}


#define AA_PPO_STATS 16
// stats must hold AA_PPO_STATS + AA_PPO_NSUM*256 floats.
int aa_ppo_loss_dist(const float* loc, const float* scale, const float* old_loc,
                     const float* old_scale, const float* actions, const float* old_logp,
                     const float* adv, const float* returns, const float* vpred,
                     const float* old_vpred, const float* weights, int64_t N, int32_t D,
                     float clip_eps, float value_clip, float c_v, float c_e, float denom,
                     float logp_clip, const float* kl_beta_dev, float kl_cutoff_coef,
                     float kl_cutoff, float* dloc, float* dscale, float* dv, float* stats,
                     void* stream) {
  if (!loc || !scale || !actions || !old_logp || !adv || !returns || !vpred || !stats)
    return AA_ERR_INVALID;
  if (N <= 0 || D <= 0 || !(denom > 0.f)) return AA_ERR_INVALID;
  if ((old_loc == nullptr) != (old_scale == nullptr)) return AA_ERR_INVALID;
  if ((dloc == nullptr) != (dscale == nullptr) || (dloc == nullptr) != (dv == nullptr))
    return AA_ERR_INVALID;
  PpoDistArgs a;
  a.loc = loc; a.scale = scale; a.old_loc = old_loc; a.old_scale = old_scale;
  a.actions = actions; a.old_logp = old_logp; a.adv = adv; a.returns = returns; a.vpred = vpred;
  a.old_vpred = old_vpred; a.weights = weights; a.N = N; a.D = (int)D;
  a.clip_eps = clip_eps; a.value_clip = value_clip; a.c_v = c_v; a.c_e = c_e; a.denom = denom;
  a.logp_clip = logp_clip;
  int P = (int)((N + 255) / 256);
  if (P > AA_PPO_P) P = AA_PPO_P;
  hipStream_t st = (hipStream_t)stream;
  float* partial = stats + AA_PPO_STATS;
  hipLaunchKernelGGL(aa_ppo_dist_fwd_kernel, dim3(P), dim3(256), 0, st, a, partial);
  hipLaunchKernelGGL(aa_ppo_dist_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)partial,
                     P, denom, (float)N, c_v, c_e, kl_beta_dev, kl_cutoff_coef, kl_cutoff, stats);
  if (dloc != nullptr)
    hipLaunchKernelGGL(aa_ppo_dist_bwd_kernel, dim3(P), dim3(256), 0, st, a,
                       (const float*)stats, (float)N, dloc, dscale, dv);
  return aa_launch_status();
}

int aa_ppo_head_forward(const float* z, const float* std_bias, const float* act_mean,
                        const float* act_mag, int64_t N, int32_t D, float* loc, float* scale,
                        void* stream) {
  if (!z || !std_bias || !loc || !scale || N <= 0 || D <= 0) return AA_ERR_INVALID;
  if ((act_mean == nullptr) != (act_mag == nullptr)) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_ppo_head_fwd_kernel, dim3(aa_ew_blocks(N * D)), dim3(256), 0,
                     (hipStream_t)stream, z, std_bias, act_mean, act_mag, N, (int)D, loc, scale);
  return aa_launch_status();
}

int aa_ppo_head_forward_sample(const float* z, const float* std_bias, const float* act_mean,
                               const float* act_mag, int64_t N, int32_t D, float* loc, float* scale,
                               uint64_t seed, int64_t* call_counter_dev, int64_t* arrival_dev,
                               const float* clip_lo, const float* clip_hi, float* action,
                               void* stream) {
  if (!z || !std_bias || !loc || !scale || !call_counter_dev || !arrival_dev || !action ||
      N <= 0 || D <= 0)
    return AA_ERR_INVALID;
  if ((act_mean == nullptr) != (act_mag == nullptr) || (clip_lo == nullptr) != (clip_hi == nullptr))
    return AA_ERR_INVALID;
  // (at most 64 workgroups: they all arrive on one word when they are done)
  unsigned blocks = aa_ew_blocks(N * D);
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(aa_ppo_head_fwd_sample_kernel, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, z, std_bias, act_mean, act_mag, N, (int)D, loc, scale,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), call_counter_dev,
                     arrival_dev, clip_lo, clip_hi, action);
  return aa_launch_status();
}

int aa_ppo_policy_step(const aa_ppo_policy_step_desc* t, void* stream) {
  if (t == nullptr || !t->x || t->B <= 0 || !t->params_a || !t->params_b || !t->value_out ||
      !t->std_bias || !t->loc || !t->scale || !t->call_counter_dev || !t->arrival_dev ||
      !t->action || t->D <= 0)
    return AA_ERR_INVALID;
  if ((t->act_mean == nullptr) != (t->act_mag == nullptr) ||
      (t->clip_lo == nullptr) != (t->clip_hi == nullptr))
    return AA_ERR_INVALID;
  if (t->nrm_var_num == nullptr && (t->nrm_mean != nullptr || t->nrm_var_den != nullptr))
    return AA_ERR_INVALID;
  AaPolicyStep P = {};
  int rc = aa_mlp_fill(P.d[0], t->n_layers_a, t->dims_a, t->acts_a, t->k_off_a, t->b_off_a);
  if (rc != AA_OK) return rc;
  rc = aa_mlp_fill(P.d[1], t->n_layers_b, t->dims_b, t->acts_b, t->k_off_b, t->b_off_b);
  if (rc != AA_OK) return rc;
  if (t->dims_a[0] != t->dims_b[0] || t->ldx < t->dims_a[0] ||
      t->dims_a[t->n_layers_a] != t->D || t->dims_b[t->n_layers_b] != 1)
    return AA_ERR_INVALID;
  P.x = t->x; P.ldx = t->ldx; P.B = t->B;
  P.nrm_mean = t->nrm_mean; P.nrm_var_num = t->nrm_var_num; P.nrm_var_den = t->nrm_var_den;
  P.nrm_eps = t->nrm_eps; P.nrm_clip = t->nrm_clip; P.has_norm = t->nrm_var_num != nullptr;
  P.params[0] = t->params_a; P.params[1] = t->params_b;
  P.value_out = t->value_out;
  P.std_bias = t->std_bias; P.act_mean = t->act_mean; P.act_mag = t->act_mag; P.D = t->D;
  P.loc = t->loc; P.scale = t->scale;
  P.seed_lo = (uint32_t)(t->seed & 0xffffffffu); P.seed_hi = (uint32_t)(t->seed >> 32);
  P.call_counter = t->call_counter_dev; P.arrival = t->arrival_dev;
  P.clip_lo = t->clip_lo; P.clip_hi = t->clip_hi; P.action = t->action;
  const int tile = aa_mlp_tile(t->B);
  const int64_t grid = (t->B + tile - 1) / tile;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)grid, 2);
  if (tile == 16) hipLaunchKernelGGL(aa_ppo_policy_step_kernel<16>, g, dim3(256), 0, st, P);
  else if (tile == 32) hipLaunchKernelGGL(aa_ppo_policy_step_kernel<32>, g, dim3(256), 0, st, P);
  else hipLaunchKernelGGL(aa_ppo_policy_step_kernel<64>, g, dim3(256), 0, st, P);
  return aa_launch_status();
}

int aa_ppo_head_backward(const float* z, const float* std_bias, const float* act_mag,
                         const float* dloc, const float* dscale, int64_t N, int32_t D, float* dz,
                         float* dbias_elem, void* stream) {
  if (!z || !std_bias || !dloc || !dscale || !dz || !dbias_elem || N <= 0 || D <= 0)
    return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_ppo_head_bwd_kernel, dim3(aa_ew_blocks(N * D)), dim3(256), 0,
                     (hipStream_t)stream, z, std_bias, act_mag, dloc, dscale, N, (int)D, dz,
                     dbias_elem);
  return aa_launch_status();
}

int aa_normal_log_prob(const float* loc, const float* scale, const float* x, int64_t N, int32_t D,
                       float* out, void* stream) {
  if (!loc || !scale || !x || !out || N <= 0 || D <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_normal_log_prob_kernel, dim3(aa_ew_blocks(N)), dim3(256), 0,
                     (hipStream_t)stream, loc, scale, x, N, (int)D, out);
  return aa_launch_status();
}

int aa_normal_sample(const float* loc, const float* scale, int64_t n, uint64_t seed,
                     const int64_t* call_counter_dev, float* out, void* stream) {
  if (!loc || !scale || !out || !call_counter_dev || n <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_normal_sample_kernel, dim3(aa_ew_blocks(n)), dim3(256), 0,
                     (hipStream_t)stream, loc, scale, n, (uint32_t)(seed & 0xffffffffu),
                     (uint32_t)(seed >> 32), call_counter_dev, out);
  return aa_launch_status();
}

int aa_uniform_sample(const float* lo, const float* hi, int64_t N, int32_t D, uint64_t seed,
                      const int64_t* call_counter_dev, float* out, void* stream) {
  if (!lo || !hi || !out || !call_counter_dev || N <= 0 || D <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_uniform_sample_kernel, dim3(aa_ew_blocks(N * D)), dim3(256), 0,
                     (hipStream_t)stream, lo, hi, N, (int)D, (uint32_t)(seed & 0xffffffffu),
                     (uint32_t)(seed >> 32), call_counter_dev, out);
  return aa_launch_status();
}

int aa_ppo_discounts(const float* discount, const int32_t* next_step_type, float gamma, int64_t B,
                     int64_t T1, float* out, void* stream) {
  if (!discount || !next_step_type || !out || B <= 0 || T1 < 2) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_ppo_discounts_kernel, dim3(aa_ew_blocks(B * (T1 - 1))), dim3(256), 0,
                     (hipStream_t)stream, discount, next_step_type, gamma, B, T1, out);
  return aa_launch_status();
}

int aa_ppo_trajectory_mask(const int32_t* step_type, const float* returns, const float* advantages,
                           const float* weights, int64_t n, float* out, void* stream) {
  if (!step_type || !returns || !advantages || !out || n <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_ppo_mask_kernel, dim3(aa_ew_blocks(n)), dim3(256), 0, (hipStream_t)stream,
                     step_type, returns, advantages, weights, n, out);
  return aa_launch_status();
}

int aa_ppo_update_kl_beta(const float* mean_kl_dev, float target, float tolerance, float* beta_dev,
                          void* stream) {
  if (!mean_kl_dev || !beta_dev) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_ppo_update_beta_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     mean_kl_dev, target, tolerance, beta_dev);
  return aa_launch_status();
}

}  // extern "C"
