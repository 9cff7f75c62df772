// Per-sample arithmetic of SAC's critic and actor losses and of the actor head's backward -- shared by
// the loss kernels (csrc/sac.hip) and the dout generators of the wide-MLP gradient chain
// (csrc/mlp_wide.hip: aa_mlp_wide_backward_gen), so that both are the same instructions (both files
// are built with -ffp-contract=off).  See sac.hip's header for the reference lines.
#pragma once
#include "common.h"
#include "agents_amd.h"
#include <math.h>

struct AaSacCriticElem {
  float wl, td, g1, g2, w;     // weighted loss term, TD target, dL/d(e) factors, sample weight
};
// critic loss of one sample (sac_agent.py:559-644); weights nullable.
__device__ static inline AaSacCriticElem aa_sac_critic_elem(
    float q1, float q2, float tq1, float tq2, float next_logp, float reward, float discount,
    const float* __restrict__ weights, int64_t b, float alpha, float gamma, float reward_scale,
    int loss_kind) {
  AaSacCriticElem o;
  const float tq = fminf(tq1, tq2) - alpha * next_logp;
  const float td = reward_scale * reward + (gamma * discount) * tq;
  float l = 0.f;
  if (loss_kind == AA_LOSS_SQUARED) {   // tf.math.squared_difference(td_targets, pred)
    const float e1 = td - q1, e2 = td - q2;
    l = e1 * e1 + e2 * e2;
    o.g1 = -2.0f * e1;
    o.g2 = -2.0f * e2;
  } else {                              // element_wise_huber_loss
    const float e1 = q1 - td, e2 = q2 - td;
    const float a1 = fabsf(e1), a2 = fabsf(e2);
    const float c1 = fminf(a1, 1.f), c2 = fminf(a2, 1.f);
    l = (0.5f * c1 * c1 + (a1 - c1)) + (0.5f * c2 * c2 + (a2 - c2));
    o.g1 = e1 > 1.f ? 1.f : (e1 < -1.f ? -1.f : e1);
    o.g2 = e2 > 1.f ? 1.f : (e2 < -1.f ? -1.f : e2);
  }
  float w = 1.f, wl = l;
  if (weights != nullptr) {
    w = weights[b];
    wl = (w == 0.f) ? 0.f : l * w;
  }
  o.wl = wl; o.td = td; o.w = w;
  return o;
}

struct AaSacActorElem {
  float wl, dq1, dq2, dlogp;
};
// actor loss of one sample (sac_agent.py:646-694): w (alpha log_pi - min(q1, q2)).
__device__ static inline AaSacActorElem aa_sac_actor_elem(float a, float c, float logp,
                                                          const float* __restrict__ weights,
                                                          int64_t b, float alpha, float loss_weight,
                                                          float global_batch) {
  AaSacActorElem o;
  const float l = alpha * logp - fminf(a, c);
  float w = 1.f, wl = l;
  if (weights != nullptr) {
    w = weights[b];
    wl = (w == 0.f) ? 0.f : l * w;
  }
  o.wl = wl;
  const float g = (loss_weight * w) / global_batch;
  // tf.minimum: the gradient goes to x where x <= y, else to y
  o.dq1 = a <= c ? -g : 0.f;
  o.dq2 = a <= c ? 0.f : -g;
  o.dlogp = alpha * g;
  return o;
}

// d loss / d head output of element (b, d) of the actor head (aa_sac_head_backward): returns
// (d / d mean, d / d raw_std).
__device__ static inline void aa_sac_head_bwd_elem(float t, float sigma, float eps, float dl,
                                                   float da, float mag, float raw, int std_kind,
                                                   float* gx_out, float* draw_out) {
  const float gx = da * (mag * (1.0f - t * t)) + dl * (2.0f * t);
  const float dsigma = gx * eps - dl / sigma;
  float draw = dsigma * sigma;
  if (std_kind == AA_SAC_STD_CLIP_EXP && (raw < -20.f || raw > 2.f)) draw = 0.f;
  *gx_out = gx;
  *draw_out = draw;
}
