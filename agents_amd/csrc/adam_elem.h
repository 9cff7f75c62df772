// The per-element arithmetic of TensorFlow's ApplyAdam and of soft_variables_update, shared by
// every launch that applies them (optim.hip: aa_adam_kernel; mlp_wide.hip: the weight-gradient
// launch that also steps the optimizer) -- one copy, so they agree bit for bit.  All including
// translation units are compiled with -ffp-contract=off.
#pragma once

__device__ static inline float adam_elem(float& p, float g, float& m, float& v, float alpha,
                                         float omb1, float omb2, float eps) {
  m = m + (g - m) * omb1;
  v = v + (g * g - v) * omb2;
  p = p - (m * alpha) / (sqrtf(v) + eps);
  return p;
}
// alpha_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)
__device__ static inline float adam_alpha(float lr, float beta1, float beta2, float t) {
  const float b1p = powf(beta1, t), b2p = powf(beta2, t);
  return lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}
__device__ static inline float soft_update_elem(float target, float p, float tau) {
  const float omt = 1.0f - tau;
  return omt * target + tau * p;
}
