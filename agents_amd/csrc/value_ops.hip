// Reverse-time scans (discounted return, GAE) and the advantage normaliser.
// Built with -ffp-contract=off; per-lane op order is exactly the reference's tf.scan body so the
// results agree bit-for-bit with oracle/value_ops.py.
//   discounted_return                   tf_agents/utils/value_ops.py:21-99
//   generalized_advantage_estimation    tf_agents/utils/value_ops.py:102-164
//   _normalize_advantages               tf_agents/agents/ppo/ppo_agent.py:100-110
//
// One lane per trajectory b, sequential over T (first-order linear recurrence).  Batch-major
// [B,T] inputs are walked through a 64x64 LDS tile per wave so global traffic stays coalesced
// (lanes read along t, the scan runs along the tile's other axis).
#include "common.h"
#include "agents_amd.h"

#define AA_SCAN_TILE 64

// MODE 0: discounted return   acc = acc*discount + reward          (init final_value or 0)
// MODE 1: GAE                 delta = r + d*v_next - v ; acc = delta + (d*lambda)*acc  (init 0)
template <int MODE>
__global__ void __launch_bounds__(64)
aa_scan_kernel(const float* __restrict__ values, const float* __restrict__ final_value,
               const float* __restrict__ discounts, const float* __restrict__ rewards,
               float td_lambda, int64_t B, int64_t T, int64_t sb, int64_t st,
               float* __restrict__ out) {
  // tile[t][b] with +1 padding: conflict-free both for row-wise fills and column-wise scans
  __shared__ float t_r[AA_SCAN_TILE][AA_SCAN_TILE + 1];
  __shared__ float t_d[AA_SCAN_TILE][AA_SCAN_TILE + 1];
  __shared__ float t_v[AA_SCAN_TILE][AA_SCAN_TILE + 1];
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * AA_SCAN_TILE;
  const int64_t b = b0 + lane;
  const bool t_contig = (st == 1);  // batch-major: lanes read along t
  float acc = 0.f;
  float v_next = 0.f;
  if (b < B) {
    if (MODE == 0) acc = final_value != nullptr ? final_value[b] : 0.f;
    if (MODE == 1) v_next = final_value[b];
  }
  for (int64_t t_hi = T; t_hi > 0; t_hi -= AA_SCAN_TILE) {
    const int64_t t_lo = t_hi > AA_SCAN_TILE ? t_hi - AA_SCAN_TILE : 0;
    const int nt = (int)(t_hi - t_lo);
    // ---- fill ---------------------------------------------------------------------------
    if (t_contig) {
      for (int bb = 0; bb < AA_SCAN_TILE; ++bb) {
        if (b0 + bb < B && lane < nt) {
          const int64_t o = (b0 + bb) * sb + (t_lo + lane);
          t_r[lane][bb] = rewards[o];
          t_d[lane][bb] = discounts[o];
          if (MODE == 1) t_v[lane][bb] = values[o];
        }
      }
    } else {
      for (int tt = 0; tt < nt; ++tt) {
        if (b < B) {
          const int64_t o = b * sb + (t_lo + tt) * st;
          t_r[tt][lane] = rewards[o];
          t_d[tt][lane] = discounts[o];
          if (MODE == 1) t_v[tt][lane] = values[o];
        }
      }
    }
    __syncthreads();
    // ---- scan (reverse) -------------------------------------------------------------------
    if (b < B) {
      for (int tt = nt - 1; tt >= 0; --tt) {
        const float r = t_r[tt][lane], d = t_d[tt][lane];
        if (MODE == 0) {
          acc = acc * d + r;
        } else {
          const float v = t_v[tt][lane];
          const float delta = r + d * v_next - v;
          acc = delta + (d * td_lambda) * acc;
          v_next = v;
        }
        t_r[tt][lane] = acc;  // reuse the reward tile for the output
      }
    }
    __syncthreads();
    // ---- drain ----------------------------------------------------------------------------
    if (t_contig) {
      for (int bb = 0; bb < AA_SCAN_TILE; ++bb)
        if (b0 + bb < B && lane < nt) out[(b0 + bb) * sb + (t_lo + lane)] = t_r[lane][bb];
    } else {
      for (int tt = 0; tt < nt; ++tt)
        if (b < B) out[b * sb + (t_lo + tt) * st] = t_r[tt][lane];
    }
    __syncthreads();
  }
}

// ---- batch-major inputs (unit time stride): the layout both PPO call sites use ------------------
// Round 5.  The kernel above gives a 64-lane workgroup 64 trajectories: [2048 x 128] is 32 waves on a
// 1,024-SIMD chip, each filling and draining its tiles with a serial 64-iteration loop -- 49 us for
// 4.2 MB (rocprofv3, PPO configs[2]).  The recurrence itself is cheap (two dependent fp32 ops per
// step once delta is taken out of the chain: ~0.6 us for 128 steps); what has to be spread over
// lanes is the memory phase.  Here a 256-lane workgroup owns SC_TB = 16 trajectories: all of its
// lanes fetch a [16 x 128] chunk of every operand with coalesced loads that are ALL requested
// before the first is used (unconditional, from clamped addresses), transpose it through LDS
// (pitch 17: conflict-free both ways), sixteen lanes of wave 0 run the recurrence in the reference's
// op order (bit-exact with the kernel above and oracle/value_ops.py), and all lanes store the
// result coalesced.  For T > 128 the next chunk's loads are in flight during the scan.
// 128 workgroups for B = 2,048.
#define SC_TB 16
#define SC_TT 128
#define SC_PER ((SC_TB * SC_TT) / 256)

template <int MODE>
__global__ void __launch_bounds__(256)
aa_scan_bm_kernel(const float* __restrict__ values, const float* __restrict__ final_value,
                  const float* __restrict__ discounts, const float* __restrict__ rewards,
                  float td_lambda, int64_t B, int64_t T, int64_t sb, float* __restrict__ out) {
  __shared__ float t_r[SC_TT][SC_TB + 1];
  __shared__ float t_d[SC_TT][SC_TB + 1];
  __shared__ float t_v[MODE == 1 ? SC_TT : 1][SC_TB + 1];
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * SC_TB;
  float acc = 0.f, v_next = 0.f;
  if (tid < SC_TB) {
    const int64_t bc = b0 + tid < B ? b0 + tid : B - 1;
    if (MODE == 0) acc = final_value != nullptr ? final_value[bc] : 0.f;
    if (MODE == 1) v_next = final_value[bc];
  }
  float rr[SC_PER], dd[SC_PER], vv[SC_PER];
  // element e = tid + 256 k of a chunk: trajectory e / SC_TT, step e % SC_TT (consecutive lanes
  // read consecutive steps of one trajectory: 512-byte runs)
  auto request = [&](int64_t t_lo, int nt) {
#pragma unroll
    for (int k = 0; k < SC_PER; ++k) {
      const int e = tid + 256 * k;
      const int row = e / SC_TT, tt = e - row * SC_TT;
      const int64_t bc = b0 + row < B ? b0 + row : B - 1;
      const int64_t o = bc * sb + t_lo + (tt < nt ? tt : nt - 1);
      rr[k] = rewards[o];
      dd[k] = discounts[o];
      if (MODE == 1) vv[k] = values[o];
    }
  };
  int64_t t_hi = T;
  int64_t t_lo = t_hi > SC_TT ? t_hi - SC_TT : 0;
  request(t_lo, (int)(t_hi - t_lo));
  while (t_hi > 0) {
    const int nt = (int)(t_hi - t_lo);
#pragma unroll
    for (int k = 0; k < SC_PER; ++k) {
      const int e = tid + 256 * k;
      const int row = e / SC_TT, tt = e - row * SC_TT;
      t_r[tt][row] = rr[k];
      t_d[tt][row] = dd[k];
      if (MODE == 1) t_v[tt][row] = vv[k];
    }
    __syncthreads();
    const int64_t n_hi = t_lo;
    const int64_t n_lo = n_hi > SC_TT ? n_hi - SC_TT : 0;
    if (n_hi > 0) request(n_lo, (int)(n_hi - n_lo));     // in flight during the scan
    if (tid < SC_TB) {
#pragma unroll 8
      for (int tt = nt - 1; tt >= 0; --tt) {
        const float r = t_r[tt][tid], d = t_d[tt][tid];
        if (MODE == 0) {
          acc = acc * d + r;
        } else {
          const float v = t_v[tt][tid];
          const float delta = r + d * v_next - v;
          acc = delta + (d * td_lambda) * acc;
          v_next = v;
        }
        t_r[tt][tid] = acc;  // the reward tile becomes the output tile
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SC_PER; ++k) {
      const int e = tid + 256 * k;
      const int row = e / SC_TT, tt = e - row * SC_TT;
      if (b0 + row < B && tt < nt) out[(b0 + row) * sb + t_lo + tt] = t_r[tt][row];
    }
    __syncthreads();
    t_hi = n_hi;
    t_lo = n_lo;
  }
}

template <int MODE>
static void aa_scan_launch(const float* values, const float* final_value, const float* discounts,
                           const float* rewards, float td_lambda, int64_t B, int64_t T,
                           int64_t sb, int64_t st, float* out, hipStream_t stream) {
  if (st == 1) {
    const dim3 grid((unsigned)((B + SC_TB - 1) / SC_TB));
    hipLaunchKernelGGL(aa_scan_bm_kernel<MODE>, grid, dim3(256), 0, stream, values, final_value,
                       discounts, rewards, td_lambda, B, T, sb, out);
  } else {
    const dim3 grid((unsigned)((B + AA_SCAN_TILE - 1) / AA_SCAN_TILE));
    hipLaunchKernelGGL(aa_scan_kernel<MODE>, grid, dim3(64), 0, stream, values, final_value,
                       discounts, rewards, td_lambda, B, T, sb, st, out);
  }
}

// Two-pass moments over n elements in ONE workgroup launch pair (n is B*T <= a few million):
// pass 1: per-block partial sums -> mean ; pass 2: partial sum of squared deviations -> var
// (tf.nn.moments computes mean, then mean(squared_difference(x, stop_gradient(mean)))).
__global__ void __launch_bounds__(256)
aa_partial_sum_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ mean_p,
                      int squared_dev, float* __restrict__ partial) {
  __shared__ float red[16];
  const float mean = squared_dev ? mean_p[0] : 0.f;
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    if (squared_dev) {
      const float dlt = v - mean;
      s += dlt * dlt;
    } else {
      s += v;
    }
  }
  const float t = aa_block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ void aa_finish_mean_kernel(const float* __restrict__ partial, int P, float n,
                                      float* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < P; ++i) s += partial[i];
    dst[0] = s / n;
  }
}
__global__ void __launch_bounds__(256)
aa_apply_norm_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ stats,
                     float eps, float* __restrict__ out) {
  // tf.nn.batch_normalization with offset=None, scale=None: x*inv + (-mean*inv), inv = rsqrt(var+eps)
  const float mean = stats[0], var = stats[1];
  const float inv = 1.0f / sqrtf(var + eps);
  const float shift = -mean * inv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = x[i] * inv + shift;
}

// n <= AA_NORM_SMALL (a PPO minibatch): mean, variance and the normalisation in ONE single-block
// launch instead of five (same two-pass definition; the block-wide sums are in a fixed order).
#define AA_NORM_SMALL 16384
__global__ void __launch_bounds__(1024)
aa_normalize_small_kernel(const float* __restrict__ x, int n, float eps, float* __restrict__ out,
                          float* __restrict__ stats) {
  __shared__ float red[16];
  __shared__ float bc[2];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  float t = aa_block_sum(s, red);
  if (threadIdx.x == 0) bc[0] = t / (float)n;
  __syncthreads();
  const float mean = bc[0];
  float d = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float dl = x[i] - mean;
    d += dl * dl;
  }
  t = aa_block_sum(d, red);
  if (threadIdx.x == 0) {
    bc[1] = t / (float)n;
    stats[0] = mean;
    stats[1] = bc[1];
  }
  __syncthreads();
  const float inv = 1.0f / sqrtf(bc[1] + eps);
  const float shift = -mean * inv;
  for (int i = threadIdx.x; i < n; i += 1024) out[i] = x[i] * inv + shift;
}

#define AA_NORM_P 256

extern "C" {

int aa_discounted_return(const float* rewards, const float* discounts, const float* final_value,
                         int64_t B, int64_t T, int64_t stride_b, int64_t stride_t, float* out,
                         void* stream) {
  if (!rewards || !discounts || !out || B <= 0 || T <= 0) return AA_ERR_INVALID;
  aa_scan_launch<0>(nullptr, final_value, discounts, rewards, 0.f, B, T, stride_b, stride_t, out,
                    (hipStream_t)stream);
  return aa_launch_status();
}

int aa_gae(const float* values, const float* final_value, const float* discounts,
           const float* rewards, float td_lambda, int64_t B, int64_t T, int64_t stride_b,
           int64_t stride_t, float* out, void* stream) {
  if (!values || !final_value || !rewards || !discounts || !out || B <= 0 || T <= 0)
    return AA_ERR_INVALID;
  aa_scan_launch<1>(values, final_value, discounts, rewards, td_lambda, B, T, stride_b, stride_t,
                    out, (hipStream_t)stream);
  return aa_launch_status();
}

// stats_out must hold 2 + AA_NORM_P floats: [mean, var, partials...]
int aa_normalize_moments(const float* x, int64_t n, float eps, float* out, float* stats_out,
                         void* stream) {
  if (!x || !out || !stats_out || n <= 0) return AA_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n <= AA_NORM_SMALL) {
    hipLaunchKernelGGL(aa_normalize_small_kernel, dim3(1), dim3(1024), 0, st, x, (int)n, eps, out,
                       stats_out);
    return aa_launch_status();
  }
  float* partial = stats_out + 2;
  int P = (int)((n + 255) / 256);
  if (P > AA_NORM_P) P = AA_NORM_P;
  hipLaunchKernelGGL(aa_partial_sum_kernel, dim3(P), dim3(256), 0, st, x, n,
                     (const float*)nullptr, 0, partial);
  hipLaunchKernelGGL(aa_finish_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, P,
                     (float)n, stats_out + 0);
  hipLaunchKernelGGL(aa_partial_sum_kernel, dim3(P), dim3(256), 0, st, x, n,
                     (const float*)stats_out, 1, partial);
  hipLaunchKernelGGL(aa_finish_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, P,
                     (float)n, stats_out + 1);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(aa_apply_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n,
                     (const float*)stats_out, eps, out);
  return aa_launch_status();
}

}  // extern "C"
