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

__global__ void __launch_bounds__(256)
aa_axpy_kernel(float* __restrict__ g, const float* __restrict__ p, int64_t n, float c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    g[i] = g[i] + c * p[i];
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

// g += c * p   (L2 regularisation gradients; keras regularizers / tf.nn.l2_loss)
int aa_add_l2_grad(float* g, const float* p, int64_t n, float c, void* stream) {
  if (!g || !p || n <= 0) return AA_ERR_INVALID;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(aa_axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g,
                     p, n, c);
  return aa_launch_status();
}

}  // extern "C"
