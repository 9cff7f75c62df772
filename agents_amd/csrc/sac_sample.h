// The tanh-squashed Normal sample of one (sample, action dimension) pair -- shared by
// aa_sac_sample_kernel (csrc/sac.hip) and the sample tail of aa_mlp_wide_fwd_kernel
// (csrc/mlp_wide.hip), so that the two are the same arithmetic instruction for instruction (both
// files are built with -ffp-contract=off).  See sac.hip's header for the reference lines.
#pragma once
#include "common.h"
#include "agents_amd.h"
#include <math.h>

#define AA_HALF_LOG_2PI_SAC 0.91893853320467274178f
#define AA_LOG2_SAC 0.69314718055994530942f

__device__ static inline float aa_softplus_f(float t) {
  return fmaxf(t, 0.f) + log1pf(expf(-fabsf(t)));
}

struct AaSacElem {
  float action, term, t, sigma, eps;
};

// mu / raw: the head's mean and raw scale of this dimension; i = b * A + d (the Philox counter of
// the pair); eps_in nullable (externally supplied noise, read at [i]).
__device__ static inline AaSacElem aa_sac_sample_elem(float mu, float raw, int std_kind,
                                                      const float* __restrict__ eps_in,
                                                      uint64_t i, uint64_t call, uint32_t seed_lo,
                                                      uint32_t seed_hi, float mean_d, float mag) {
  if (std_kind == AA_SAC_STD_CLIP_EXP) raw = fminf(fmaxf(raw, -20.f), 2.f);
  const float sigma = expf(raw);
  float eps;
  if (eps_in != nullptr) {
    eps = eps_in[i];
  } else {
    const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), seed_lo, seed_hi);
    const float u1 = 1.0f - aa_u01(r.x);
    const float u2 = aa_u01(r.y);
    eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
  }
  const float x = mu + sigma * eps;
  const float t = tanhf(x);
  AaSacElem o;
  o.action = mean_d + mag * t;
  const float e = (x - mu) / sigma;   // what MultivariateNormalDiag.log_prob recomputes
  const float fldj = 2.0f * (AA_LOG2_SAC - x - aa_softplus_f(-2.0f * x));
  o.term = -0.5f * (e * e) - logf(sigma) - AA_HALF_LOG_2PI_SAC - logf(fabsf(mag)) - fldj;
  o.t = t; o.sigma = sigma; o.eps = eps;
  return o;
}
