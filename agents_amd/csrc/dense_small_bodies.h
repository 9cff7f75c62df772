// Device bodies of the small-N Dense backward kernels (csrc/dense_small.hip), shared with the fused
// "TD loss + Q-head backward" launch of csrc/dqn.hip.  Explicit fmaf everywhere: the results do not
// depend on the including file's -ffp-contract setting.
#pragma once
#include "common.h"
#include "agents_amd.h"

#define AA_SMALLN_MAX 16

__device__ static inline float aa_sm_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}
__device__ static inline float aa_sm_actgrad(float y, int kind) {
  if (kind == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (kind == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

template <int N>
__device__ static inline void aa_dense_small_dx_body(const float* __restrict__ dz,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ mask_src,
                                                     int mask_kind, int64_t M, int K,
                                                     float* __restrict__ dx, unsigned block,
                                                     unsigned n_blocks) {
  const int64_t total = M * (int64_t)K;
  for (int64_t i = (int64_t)block * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)n_blocks * blockDim.x) {
    const int64_t m = i / K;
    const int k = (int)(i - m * K);
    const float* dzr = dz + m * N;
    const float* wk = w + (size_t)k * N;
    float v = 0.f;
#pragma unroll
    for (int n = 0; n < N; ++n) v = fmaf(dzr[n], wk[n], v);
    if (mask_kind != 0) v *= aa_sm_actgrad(mask_src[i], mask_kind);
    dx[i] = v;
  }
}

template <int N>
__device__ static inline void aa_dense_small_dw_body(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ dz, int64_t M,
                                                     int K, float* __restrict__ dw,
                                                     float* __restrict__ db, unsigned block,
                                                     unsigned n_blocks) {
  __shared__ float red[3][64][N];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k = block * 64 + lane;
  float acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0.f;
  // wave `wid` owns rows [m_lo, m_hi): a fixed quarter of M
  const int64_t per = (M + 3) / 4;
  const int64_t m_lo = wid * per, m_hi = (m_lo + per < M) ? m_lo + per : M;
  if (k < K) {
    int64_t m = m_lo;
    for (; m + 8 <= m_hi; m += 8) {     // eight independent row loads in flight
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(m + u) * ldx + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* dzr = dz + (m + u) * N;   // wave-uniform address: scalar loads
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = fmaf(xv[u], dzr[n], acc[n]);
      }
    }
    for (; m < m_hi; ++m) {
      const float xv = x[m * ldx + k];
      const float* dzr = dz + m * N;
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] = fmaf(xv, dzr[n], acc[n]);
    }
  }
  if (wid > 0) {
#pragma unroll
    for (int n = 0; n < N; ++n) red[wid - 1][lane][n] = acc[n];
  }
  __syncthreads();
  if (wid == 0 && k < K) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] += red[j][lane][n];
#pragma unroll
    for (int n = 0; n < N; ++n) dw[(size_t)k * N + n] = acc[n];
  }
  // bias gradient: db[n] = sum_m dz[m,n], by the last workgroup's spare wave in fixed m order
  if (db != nullptr && block == n_blocks - 1 && wid == 1) {
    float s[N];
#pragma unroll
    for (int n = 0; n < N; ++n) s[n] = 0.f;
    for (int64_t m = lane; m < M; m += 64) {
#pragma unroll
      for (int n = 0; n < N; ++n) s[n] += dz[m * N + n];
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float v = s[n];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) db[n] = v;
    }
  }
}

