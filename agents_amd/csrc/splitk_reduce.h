// Deterministic split-K slab reduction + epilogue, shared by the GEMM family (gemm.hip) and the
// per-frame convolution weight gradient (conv_dw_frame_x6.hip), and the activation helpers of
// their epilogues.
#pragma once
#include "common.h"
#include "agents_amd.h"

__device__ static inline float aa_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}
__device__ static inline float aa_actgrad(float y, int kind) {
  // derivative of the activation expressed through its OUTPUT y
  if (kind == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (kind == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}


// out[m][n] = epilogue( sum_z slab[z][m][n] ), deterministic.  A workgroup owns IPB consecutive
// float4 (or scalar) items and ZL "z-lanes": lane zl sums slabs zl, zl+ZL, zl+2ZL, ... in that
// order, four loads in flight, and the ZL partials are then added in lane order through LDS -- a
// fixed association for a given (splits, ZL), so the result is reproducible run to run.  With one
// z-lane this is the plain z-order sum.  (A single thread walking 247 slabs of the conv1 weight
// gradient serially took 60 us for 8 MB; 16 z-lanes bring it to the memory system's pace.)
// Elements [M*N, M*N + N) of the index space are the fused bias-gradient rows that follow the
// slabs: colsum_out[n] = sum_z slab_end[z][n].
// The sum itself, shared by every consumer of slabs (the reduce launches below and the optimizer
// that reads gradients straight from slabs, optim.hip): `emit(i, tail, v)` is called by the thread
// that holds the finished sum of item i (tail: a bias-gradient item, i >= MN).  n_tail = number of
// bias-gradient elements that follow the slabs (0: none).
template <int VEC, int ZL, class Emit>
__device__ static inline void
aa_splitk_reduce_walk(const float* __restrict__ slab, int splits, size_t MN, int n_tail,
                      unsigned block, unsigned n_blocks, Emit emit) {
  constexpr int IPB = 256 / ZL;
  __shared__ float part[ZL][IPB][VEC];
  const size_t total = (MN + (size_t)n_tail) / VEC;
  const float* cs_rows = slab + (size_t)splits * MN;
  const int it = threadIdx.x % IPB, zl = threadIdx.x / IPB;
  for (size_t q0 = (size_t)block * IPB; q0 < total; q0 += (size_t)n_blocks * IPB) {
    const size_t q = q0 + it;
    const bool live = q < total;
    const size_t i = q * VEC;
    const bool tail = live && i >= MN;   // bias-gradient element(s)
    const float* src = tail ? cs_rows + (i - MN) : slab + i;
    const size_t zstride = tail ? (size_t)n_tail : MN;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    if (live) {
      int z = zl;
      for (; z + 3 * ZL < splits; z += 4 * ZL) {
        float t[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* ps = src + (size_t)(z + u * ZL) * zstride;
          if constexpr (VEC == 4) {
            const float4 f = *reinterpret_cast<const float4*>(ps);
            t[u][0] = f.x; t[u][1] = f.y; t[u][2] = f.z; t[u][3] = f.w;
          } else {
            t[u][0] = ps[0];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] += t[u][e];
      }
      for (; z < splits; z += ZL) {
        const float* ps = src + (size_t)z * zstride;
        if constexpr (VEC == 4) {
          const float4 f = *reinterpret_cast<const float4*>(ps);
          v[0] += f.x; v[1] += f.y; v[2] += f.z; v[3] += f.w;
        } else {
          v[0] += ps[0];
        }
      }
    }
    if constexpr (ZL > 1) {
      __syncthreads();   // previous round's readers are done with `part`
#pragma unroll
      for (int e = 0; e < VEC; ++e) part[zl][it][e] = v[e];
      __syncthreads();
      if (zl != 0) continue;
#pragma unroll
      for (int j = 1; j < ZL; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += part[j][it][e];
    }
    if (!live) continue;
    emit(i, tail, v);
  }
}

template <int VEC, int ZL>
__device__ static inline void
aa_splitk_reduce_body(const float* __restrict__ slab, int splits, int M, int N,
                      float* __restrict__ C, int ldc, const float* __restrict__ bias, int act,
                      const float* __restrict__ mask_src, int ldm, int mask_kind,
                      float* __restrict__ colsum_out, unsigned block, unsigned n_blocks) {
  const size_t MN = (size_t)M * N;
  aa_splitk_reduce_walk<VEC, ZL>(
      slab, splits, MN, colsum_out != nullptr ? N : 0, block, n_blocks,
      [&](size_t i, bool tail, float (&v)[VEC]) {
        if (tail) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) colsum_out[i - MN + e] = v[e];
          return;
        }
        const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float x = v[e];
          if (bias != nullptr) x += bias[n + e];
          x = aa_act(x, act);
          if (mask_kind != 0) x *= aa_actgrad(mask_src[(size_t)m * ldm + n + e], mask_kind);
          v[e] = x;
        }
        if constexpr (VEC == 4) {
          *reinterpret_cast<float4*>(C + (size_t)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          C[(size_t)m * ldc + n] = v[0];
        }
      });
}


template <int VEC, int ZL>
__global__ void __launch_bounds__(256)
aa_splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                        float* __restrict__ C, int ldc, const float* __restrict__ bias, int act,
                        const float* __restrict__ mask_src, int ldm, int mask_kind,
                        float* __restrict__ colsum_out) {
  aa_splitk_reduce_body<VEC, ZL>(slab, splits, M, N, C, ldc, bias, act, mask_src, ldm, mask_kind,
                                 colsum_out, blockIdx.x, gridDim.x);
}

// Several independent slab sets in ONE launch (the conv weight gradients of consecutive layers:
// their reduces used to be a launch each on the backward pass's side stream).  Segment s owns the
// blocks [first[s], first[s + 1]); inside a segment the arithmetic is aa_splitk_reduce_body's:
// bit-identical to one launch per segment.
#define AA_REDUCE_MAX_SEGS 4
struct AaReduceSegs {
  int n;
  int first[AA_REDUCE_MAX_SEGS + 1];
  const float* slab[AA_REDUCE_MAX_SEGS];
  int splits[AA_REDUCE_MAX_SEGS], M[AA_REDUCE_MAX_SEGS], N[AA_REDUCE_MAX_SEGS];
  float* C[AA_REDUCE_MAX_SEGS];
  float* colsum[AA_REDUCE_MAX_SEGS];
};
template <int VEC, int ZL>
__global__ void __launch_bounds__(256) aa_splitk_reduce_multi_kernel(AaReduceSegs g) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < AA_REDUCE_MAX_SEGS; ++k)
    if (k < g.n && (int)blockIdx.x >= g.first[k]) s = k;
  // (uniform per workgroup: scalar selects of the segment's descriptor)
  const float* slab = g.slab[0]; int splits = g.splits[0], M = g.M[0], N = g.N[0];
  float* C = g.C[0]; float* cs = g.colsum[0]; int b0 = g.first[0], b1 = g.first[1];
#pragma unroll
  for (int k = 1; k < AA_REDUCE_MAX_SEGS; ++k)
    if (s == k) {
      slab = g.slab[k]; splits = g.splits[k]; M = g.M[k]; N = g.N[k];
      C = g.C[k]; cs = g.colsum[k]; b0 = g.first[k]; b1 = g.first[k + 1];
    }
  aa_splitk_reduce_body<VEC, ZL>(slab, splits, M, N, C, N, nullptr, 0, nullptr, 0, 0, cs,
                                 blockIdx.x - (unsigned)b0, (unsigned)(b1 - b0));
}
