// Dense layers with a handful of output units (the Q-value / value heads: keras Dense(num_actions),
// tf_agents/networks/q_network.py:139-150, value_network.py) -- forward, input gradient and weight
// gradient as three small latency-tuned kernels instead of MFMA GEMM launches that would fill
// 6 of 32 tile columns and need a split-K pass:   y[M,N] = act(x[M,K] W[K,N] + b),  N <= 16.
//   forward : one wave per row; lanes stride K with 16-byte loads; W (K*N*4 bytes <= 32 KiB for
//             the heads) is read through the scalar/vector caches; 64-lane shuffle reduction.
//   dX      : dx[m,k] = (sum_n dz[m,n] W[k,n]) * act'(mask[m,k])   -- N MACs per element.
//   dW,db   : dW[k,n] = sum_m x[m,k] dz[m,n]; 64 consecutive k per workgroup, the M rows split
//             over the 4 waves and combined in LDS in wave order (deterministic).
// Summation order differs from the MFMA GEMM's (fp32 rounding noise, covered by the 1e-5 relative
// loss tolerance); every kernel is deterministic run to run.
#include "common.h"
#include "agents_amd.h"
#include "dense_small_bodies.h"
#include <float.h>

template <int N>
__global__ void __launch_bounds__(256)
aa_dense_small_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                          const float* __restrict__ bias, int act, int64_t M, int K,
                          float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 4 + wid;
  if (m >= M) return;
  const float* xr = x + m * ldx;
  float acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0.f;
  const bool vec = ((K & 3) == 0) && ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  if (vec) {
    for (int k = 4 * lane; k < K; k += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + k);
      const float* wk = w + (size_t)k * N;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        acc[n] = fmaf(xv.x, wk[n], acc[n]);
        acc[n] = fmaf(xv.y, wk[N + n], acc[n]);
        acc[n] = fmaf(xv.z, wk[2 * N + n], acc[n]);
        acc[n] = fmaf(xv.w, wk[3 * N + n], acc[n]);
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      const float xv = xr[k];
      const float* wk = w + (size_t)k * N;
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] = fmaf(xv, wk[n], acc[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    float v = acc[n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    acc[n] = v;
  }
  if (lane < N) {
    float v = 0.f;
#pragma unroll
    for (int n = 0; n < N; ++n) v = (lane == n) ? acc[n] : v;
    if (bias != nullptr) v += bias[lane];
    y[m * N + lane] = aa_sm_act(v, act);
  }
}

// The same head, reading its input as the raw split-K slabs of the layer below:
//   h[m,k] = act1( sum_z slab[z][m][k] + bias1[k] )  (written out: the backward pass needs it)
//   y[m,n] = act2( sum_k h[m,k] W[k,n] + b[n] )
// The slab sum is the "launch-boundary reduce" done in the consumer's prologue: the separate
// aa_splitk_reduce launch (and its round trip of h through memory before the head reads it) goes
// away.  z-order sum from 0.f, bias, activation: the arithmetic of aa_splitk_reduce_kernel<*,1>;
// the head's accumulation order is aa_dense_small_fwd_kernel's -- both results are bit-identical
// to the two-launch path.  One wave per row, one row per workgroup (256 rows -> 256 CUs).
template <int N>
__global__ void __launch_bounds__(64)
aa_dense_small_fwd_slabs_kernel(const float* __restrict__ slab, int splits, int64_t M, int K,
                                const float* __restrict__ bias1, int act1,
                                float* __restrict__ h, int64_t ldh, const float* __restrict__ w,
                                const float* __restrict__ bias, int act, float* __restrict__ y) {
  const int lane = threadIdx.x;
  const int64_t m = blockIdx.x;
  const size_t MK = (size_t)M * K;
  float acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0.f;
  // Two 256-column passes per trip (K = 512: the whole row in ONE trip): everything both passes
  // need -- up to eight slabs each, the hidden layer's bias, the four rows of the head's kernel --
  // is requested before anything is used, so a row costs one memory round trip, not one per pass
  // (round 3 had already merged the four dependent round trips INSIDE a pass).  Unconditional
  // loads from clamped addresses; the adds keep the z order of aa_splitk_reduce_kernel<*,1> and
  // the pass order of the one-pass loop: bit-identical results.
  for (int k0 = 4 * lane; k0 < K; k0 += 512) {
    float4 t[2][8], b1[2];
    float wr[2][4 * N];
    int kk[2];
    bool on[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int k = k0 + 256 * h2;
      on[h2] = k < K;
      kk[h2] = on[h2] ? k : k0;
      const float* src = slab + (size_t)m * K + kk[h2];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        t[h2][u] =
            *reinterpret_cast<const float4*>(src + (size_t)(u < splits ? u : splits - 1) * MK);
      b1[h2] = *reinterpret_cast<const float4*>(bias1 != nullptr ? bias1 + kk[h2] : src);
      const float* wk = w + (size_t)kk[h2] * N;
#pragma unroll
      for (int j = 0; j < 4 * N; ++j) wr[h2][j] = wk[j];
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      if (!on[h2]) continue;
      const int k = kk[h2];
      const float* src = slab + (size_t)m * K + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < splits) {
          v.x += t[h2][u].x; v.y += t[h2][u].y; v.z += t[h2][u].z; v.w += t[h2][u].w;
        }
      for (int z = 8; z < splits; ++z) {
        const float4 tz = *reinterpret_cast<const float4*>(src + (size_t)z * MK);
        v.x += tz.x; v.y += tz.y; v.z += tz.z; v.w += tz.w;
      }
      if (bias1 != nullptr) { v.x += b1[h2].x; v.y += b1[h2].y; v.z += b1[h2].z; v.w += b1[h2].w; }
      v.x = aa_sm_act(v.x, act1); v.y = aa_sm_act(v.y, act1);
      v.z = aa_sm_act(v.z, act1); v.w = aa_sm_act(v.w, act1);
      *reinterpret_cast<float4*>(h + m * ldh + k) = v;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        acc[n] = fmaf(v.x, wr[h2][n], acc[n]);
        acc[n] = fmaf(v.y, wr[h2][N + n], acc[n]);
        acc[n] = fmaf(v.z, wr[h2][2 * N + n], acc[n]);
        acc[n] = fmaf(v.w, wr[h2][3 * N + n], acc[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    float v = acc[n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    acc[n] = v;
  }
  if (lane < N) {
    float v = 0.f;
#pragma unroll
    for (int n = 0; n < N; ++n) v = (lane == n) ? acc[n] : v;
    if (bias != nullptr) v += bias[lane];
    y[m * N + lane] = aa_sm_act(v, act);
  }
}

template <int N>
__global__ void __launch_bounds__(256)
aa_dense_small_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                         const float* __restrict__ mask_src, int mask_kind, int64_t M, int K,
                         float* __restrict__ dx) {
  aa_dense_small_dx_body<N>(dz, w, mask_src, mask_kind, M, K, dx, blockIdx.x, gridDim.x);
}

template <int N>
__global__ void __launch_bounds__(256)
aa_dense_small_dw_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dz,
                         int64_t M, int K, float* __restrict__ dw, float* __restrict__ db) {
  aa_dense_small_dw_body<N>(x, ldx, dz, M, K, dw, db, blockIdx.x, gridDim.x);
}

// dX and dW / db of the head in ONE launch: the first n_dw workgroups are the weight-gradient
// ones (few and long: they go first), the rest sweep the input gradient.  Same bodies, same
// results as the two separate launches.
template <int N>
__global__ void __launch_bounds__(256)
aa_dense_small_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dz,
                          const float* __restrict__ w, const float* __restrict__ mask_src,
                          int mask_kind, int64_t M, int K, float* __restrict__ dx,
                          float* __restrict__ dw, float* __restrict__ db, unsigned n_dw) {
  if (blockIdx.x < n_dw)   // uniform per workgroup: the barrier inside the body is safe
    aa_dense_small_dw_body<N>(x, ldx, dz, M, K, dw, db, blockIdx.x, n_dw);
  else
    aa_dense_small_dx_body<N>(dz, w, mask_src, mask_kind, M, K, dx, blockIdx.x - n_dw,
                              gridDim.x - n_dw);
}

template <int N>
static int aa_small_launch(int which, const float* a, int64_t lda, const float* b, const float* c,
                           int i0, int64_t M, int K, float* out, float* out2, hipStream_t st) {
  if (which == 0) {
    hipLaunchKernelGGL(aa_dense_small_fwd_kernel<N>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0,
                       st, a, lda, b, c, i0, M, K, out);
  } else if (which == 1) {
    int64_t blocks = (M * K + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(aa_dense_small_dx_kernel<N>, dim3((unsigned)blocks), dim3(256), 0, st, a, b,
                       c, i0, M, K, out);
  } else {
    hipLaunchKernelGGL(aa_dense_small_dw_kernel<N>, dim3((unsigned)((K + 63) / 64)), dim3(256), 0,
                       st, a, lda, b, M, K, out, out2);
  }
  return aa_launch_status();
}

static int aa_small_dispatch(int N, int which, const float* a, int64_t lda, const float* b,
                             const float* c, int i0, int64_t M, int K, float* out, float* out2,
                             hipStream_t st) {
  switch (N) {
#define AA_CASE(n) \
  case n: return aa_small_launch<n>(which, a, lda, b, c, i0, M, K, out, out2, st);
    AA_CASE(1) AA_CASE(2) AA_CASE(3) AA_CASE(4) AA_CASE(5) AA_CASE(6) AA_CASE(7) AA_CASE(8)
    AA_CASE(9) AA_CASE(10) AA_CASE(11) AA_CASE(12) AA_CASE(13) AA_CASE(14) AA_CASE(15) AA_CASE(16)
#undef AA_CASE
    default: return AA_ERR_RANGE;
  }
}

extern "C" {

int aa_dense_small_forward(const float* x, int64_t ldx, const float* w, const float* bias,
                           int32_t act, int64_t M, int32_t K, int32_t N, float* y, void* stream) {
  if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0 || ldx < K) return AA_ERR_INVALID;
  if (N > AA_SMALLN_MAX) return AA_ERR_RANGE;
  return aa_small_dispatch(N, 0, x, ldx, w, bias, act, M, K, y, nullptr, (hipStream_t)stream);
}

int aa_dense_small_forward_slabs(const float* slabs, int32_t splits, int64_t M, int32_t K,
                                 const float* bias1, int32_t act1, float* h, int64_t ldh,
                                 const float* w, const float* bias, int32_t act, int32_t N,
                                 float* y, void* stream) {
  if (!slabs || !h || !w || !y || splits < 1 || M <= 0 || K <= 0 || N <= 0 || ldh < K)
    return AA_ERR_INVALID;
  if (N > AA_SMALLN_MAX || M > 0x7fffffffLL) return AA_ERR_RANGE;
  // 16-byte rows everywhere: the slab sum is a float4 stream
  if ((K & 3) != 0 || (ldh & 3) != 0 || (((uintptr_t)slabs | (uintptr_t)h) & 15) != 0)
    return AA_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  switch (N) {
#define AA_SM_TAIL(NN)                                                                          \
    case NN:                                                                                    \
      hipLaunchKernelGGL((aa_dense_small_fwd_slabs_kernel<NN>), dim3((unsigned)M), dim3(64), 0, \
                         st, slabs, splits, M, K, bias1, act1, h, ldh, w, bias, act, y);        \
      break;
    AA_SM_TAIL(1) AA_SM_TAIL(2) AA_SM_TAIL(3) AA_SM_TAIL(4) AA_SM_TAIL(5) AA_SM_TAIL(6)
    AA_SM_TAIL(7) AA_SM_TAIL(8) AA_SM_TAIL(9) AA_SM_TAIL(10) AA_SM_TAIL(11) AA_SM_TAIL(12)
    AA_SM_TAIL(13) AA_SM_TAIL(14) AA_SM_TAIL(15) AA_SM_TAIL(16)
#undef AA_SM_TAIL
    default: return AA_ERR_RANGE;
  }
  return aa_launch_status();
}

int aa_dense_small_dx(const float* dz, const float* w, const float* mask_src, int32_t mask_kind,
                      int64_t M, int32_t K, int32_t N, float* dx, void* stream) {
  if (!dz || !w || !dx || M <= 0 || K <= 0 || N <= 0) return AA_ERR_INVALID;
  if (N > AA_SMALLN_MAX) return AA_ERR_RANGE;
  return aa_small_dispatch(N, 1, dz, 0, w, mask_src, mask_src ? mask_kind : 0, M, K, dx, nullptr,
                           (hipStream_t)stream);
}

int aa_dense_small_backward(const float* x, int64_t ldx, const float* dz, const float* w,
                            const float* mask_src, int32_t mask_kind, int64_t M, int32_t K,
                            int32_t N, float* dx, float* dw, float* db, void* stream) {
  if (!x || !dz || !w || !dx || !dw || M <= 0 || K <= 0 || N <= 0 || ldx < K)
    return AA_ERR_INVALID;
  if (N > AA_SMALLN_MAX) return AA_ERR_RANGE;
  const unsigned n_dw = (unsigned)((K + 63) / 64);
  int64_t n_dx = (M * K + 255) / 256;
  if (n_dx > 2048) n_dx = 2048;
  const int mk = mask_src ? mask_kind : 0;
  hipStream_t st = (hipStream_t)stream;
  switch (N) {
#define AA_SM_BWD(NN)                                                                          \
    case NN:                                                                                   \
      hipLaunchKernelGGL((aa_dense_small_bwd_kernel<NN>), dim3(n_dw + (unsigned)n_dx),         \
                         dim3(256), 0, st, x, ldx, dz, w, mask_src, mk, M, K, dx, dw, db, n_dw); \
      break;
    AA_SM_BWD(1) AA_SM_BWD(2) AA_SM_BWD(3) AA_SM_BWD(4) AA_SM_BWD(5) AA_SM_BWD(6) AA_SM_BWD(7)
    AA_SM_BWD(8) AA_SM_BWD(9) AA_SM_BWD(10) AA_SM_BWD(11) AA_SM_BWD(12) AA_SM_BWD(13)
    AA_SM_BWD(14) AA_SM_BWD(15) AA_SM_BWD(16)
#undef AA_SM_BWD
    default: return AA_ERR_RANGE;
  }
  return aa_launch_status();
}

int aa_dense_small_dw(const float* x, int64_t ldx, const float* dz, int64_t M, int32_t K,
                      int32_t N, float* dw, float* db, void* stream) {
  if (!x || !dz || !dw || M <= 0 || K <= 0 || N <= 0 || ldx < K) return AA_ERR_INVALID;
  if (N > AA_SMALLN_MAX) return AA_ERR_RANGE;
  return aa_small_dispatch(N, 2, x, ldx, dz, nullptr, 0, M, K, dw, db, (hipStream_t)stream);
}

}  // extern "C"
