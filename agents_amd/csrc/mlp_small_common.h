// Pieces of the small-MLP kernels (csrc/mlp_small.hip) shared with the PPO policy-step kernel
// (csrc/ppo.hip): layout descriptor, activation, weight staging, tile choice.  The layer arithmetic
// is explicit fmaf on both sides, so it does not depend on the translation unit's -ffp-contract.
#pragma once
#include "common.h"
#include "agents_amd.h"

#define AA_MLP_MAXW 64

struct AaMlpDesc {
  int n_layers;
  int dims[AA_MLP_MAX_LAYERS + 1];     // dims[0] = input width
  int acts[AA_MLP_MAX_LAYERS];
  int64_t k_off[AA_MLP_MAX_LAYERS];    // float offsets into the flat parameter / gradient buffer
  int64_t b_off[AA_MLP_MAX_LAYERS];
};

__device__ static inline float aa_mlp_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}
__device__ static inline float aa_mlp_actgrad(float y, int act) {
  if (act == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

// W [n_in][n_out] row-major in HBM -> LDS [64][64] (zero padded); bias -> LDS [64].
__device__ static inline void aa_mlp_stage_w(const float* __restrict__ params, int64_t k_off,
                                             int64_t b_off, int n_in, int n_out, float* Ws,
                                             float* bs) {
  for (int i = threadIdx.x; i < AA_MLP_MAXW * AA_MLP_MAXW; i += blockDim.x) {
    const int k = i >> 6, j = i & 63;
    Ws[i] = (k < n_in && j < n_out) ? params[k_off + (int64_t)k * n_out + j] : 0.f;
  }
  if (threadIdx.x < AA_MLP_MAXW) bs[threadIdx.x] = threadIdx.x < n_out ? params[b_off + threadIdx.x]
                                                                       : 0.f;
}

static int aa_mlp_tile(int64_t B) {
  if (B >= 64 * 512) return 64;
  if (B >= 32 * 512) return 32;
  return 16;
}

static int aa_mlp_fill(AaMlpDesc& d, int n_layers, const int32_t* dims, const int32_t* acts,
                       const int64_t* k_off, const int64_t* b_off) {
  if (n_layers < 1 || n_layers > AA_MLP_MAX_LAYERS || !dims || !acts || !k_off || !b_off)
    return AA_ERR_INVALID;
  d.n_layers = n_layers;
  for (int i = 0; i <= n_layers; ++i) {
    if (dims[i] < 1 || dims[i] > AA_MLP_MAXW) return AA_ERR_RANGE;
    d.dims[i] = dims[i];
  }
  for (int i = 0; i < n_layers; ++i) {
    if (acts[i] < AA_ACT_NONE || acts[i] > AA_ACT_TANH) return AA_ERR_INVALID;
    d.acts[i] = acts[i];
    d.k_off[i] = k_off[i];
    d.b_off[i] = b_off[i];
  }
  return AA_OK;
}

