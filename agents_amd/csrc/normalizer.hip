// StreamingTensorNormalizer / EMATensorNormalizer (tf_agents/utils/tensor_normalizer.py:134-474)
// as two launches per update and one per normalisation, for x viewed as [n_outer, J] fp32 with the
// J elements of the tensor spec fastest.
//
//   update   pass 1: a workgroup owns a contiguous range of outer rows and produces, per element
//            j, the range's (mean, M2 about that mean) -- a two-pass sum over rows it re-reads from
//            L2 -- or, for the EMA normaliser, (sum x, sum (x - moving_mean)^2).
//            pass 2: one thread per j merges the workgroups' partials IN WORKGROUP ORDER
//            (Chan et al., the same identity the reference uses) and then applies the reference's
//            own state update: parallel_variance_calculation + kahan_summation (:397-474), op for
//            op, with n_a = the batch and n_b = the running state.
//   apply    y = clip(x * inv + (-mean * inv)), inv = 1 / sqrt(var + eps): the expression
//            tf.nn.batch_normalization evaluates (:176-186); var = M2 / count for the streaming
//            normaliser (:367-370).
// HBM-bound: 4 B read (x2, second read from L2) per element for the update, 8 B per element for
// the normalisation.  Everything is deterministic (fixed tree / fixed merge order, no atomics).
// Compiled with -ffp-contract=off: the state update is then the reference's fp32 arithmetic.
#include "common.h"
#include "agents_amd.h"

#define AA_NORM_THREADS 256
#define AA_NORM_MAX_GROUPS 1024

// mode 0: streaming (block mean, block M2); mode 1: EMA (sum x, sum (x - ref_mean)^2)
template <int MODE>
__global__ void __launch_bounds__(AA_NORM_THREADS)
aa_norm_partial_kernel(const float* __restrict__ x, int64_t n_outer, int J, int64_t rows_per_group,
                       const float* __restrict__ ref_mean, float* __restrict__ partial) {
  // threads are laid out (r, j): r = t / Jt rows in flight, Jt = min(J, 256) elements per pass
  __shared__ float red[AA_NORM_THREADS];
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_group;
  int64_t r1 = r0 + rows_per_group;
  if (r1 > n_outer) r1 = n_outer;
  const float n_rows = (float)(r1 - r0);
  float* out_a = partial + ((int64_t)blockIdx.x * 2 + 0) * J;
  float* out_b = partial + ((int64_t)blockIdx.x * 2 + 1) * J;
  for (int j0 = 0; j0 < J; j0 += AA_NORM_THREADS) {
    const int Jt = (J - j0) < AA_NORM_THREADS ? (J - j0) : AA_NORM_THREADS;
    const int R = AA_NORM_THREADS / Jt;          // rows in flight
    const int r = t / Jt, j = t - r * Jt;
    const bool live = r < R;
    // ---- first sum: x ------------------------------------------------------------------------
    float s = 0.f;
    if (live)
      for (int64_t row = r0 + r; row < r1; row += R) s += x[row * J + j0 + j];
    red[t] = s;
    __syncthreads();
    float a = 0.f;
    if (t < Jt) {
      for (int k = 0; k < R; ++k) a += red[k * Jt + t];   // fixed order over the row lanes
      if (MODE == 0) a = a / n_rows;                       // the range's mean
    }
    __syncthreads();
    if (t < Jt) red[t] = (MODE == 0) ? a : ref_mean[j0 + t];
    __syncthreads();
    const float centre = red[j];
    __syncthreads();
    // ---- second sum: squared differences -----------------------------------------------------
    float q = 0.f;
    if (live)
      for (int64_t row = r0 + r; row < r1; row += R) {
        const float d = x[row * J + j0 + j] - centre;
        q += d * d;
      }
    red[t] = q;
    __syncthreads();
    if (t < Jt) {
      float b = 0.f;
      for (int k = 0; k < R; ++k) b += red[k * Jt + t];
      out_a[j0 + t] = a;
      out_b[j0 + t] = b;
    }
    __syncthreads();
  }
}

// state rows: [0] count, [1] avg, [2] m2, [3] m2_carry  (each J floats)
__global__ void aa_streaming_norm_finish_kernel(const float* __restrict__ partial, int n_groups,
                                                int64_t n_outer, int64_t rows_per_group, int J,
                                                float* __restrict__ state) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  // merge the workgroups' (n, mean, M2) in workgroup order
  float n_acc = 0.f, mean = 0.f, m2 = 0.f;
  for (int g = 0; g < n_groups; ++g) {
    int64_t rows = n_outer - (int64_t)g * rows_per_group;
    if (rows > rows_per_group) rows = rows_per_group;
    const float n_g = (float)rows;
    const float mean_g = partial[((int64_t)g * 2 + 0) * J + j];
    const float m2_g = partial[((int64_t)g * 2 + 1) * J + j];
    const float n_new = n_acc + n_g;
    const float delta = mean_g - mean;
    const float w = n_g / n_new;
    mean = mean + delta * w;
    m2 = (m2 + m2_g) + (delta * n_acc) * (delta * w);
    n_acc = n_new;
  }
  // tensor_normalizer.py:330-346 + parallel_variance_calculation (:397-449): a = batch, b = state
  const float n_a = (float)n_outer, avg_a = mean, m2_a = m2;
  const float n_b = state[0 * (int64_t)J + j], avg_b = state[1 * (int64_t)J + j];
  const float m2_b = state[2 * (int64_t)J + j], m2_b_c = state[3 * (int64_t)J + j];
  const float n_ab = n_a + n_b;
  const float delta = avg_b - avg_a;
  const float s_delta = delta * n_b / n_ab;
  const float avg_ab = avg_a + s_delta;
  const float value = m2_a + (delta * n_a * s_delta);
  // kahan_summation(accumulator=m2_b, carry=m2_b_c, value) (:452-474)
  const float kd = value - m2_b_c;
  const float acc = m2_b + kd;
  const float carry = (acc - m2_b) - kd;
  state[0 * (int64_t)J + j] = n_ab;
  state[1 * (int64_t)J + j] = avg_ab;
  state[2 * (int64_t)J + j] = acc;
  state[3 * (int64_t)J + j] = carry;
}

// EMA (:252-281): mean_var += rate * (batch_mean - mean_var); var_var += rate * (batch_var -
// var_var), batch_var taken about the OLD moving mean.  state rows: [0] mean, [1] var.
__global__ void aa_ema_norm_finish_kernel(const float* __restrict__ partial, int n_groups,
                                          int64_t n_outer, int J, float rate,
                                          float* __restrict__ state) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  float sx = 0.f, sq = 0.f;
  for (int g = 0; g < n_groups; ++g) {
    sx += partial[((int64_t)g * 2 + 0) * J + j];
    sq += partial[((int64_t)g * 2 + 1) * J + j];
  }
  const float n = (float)n_outer;
  const float mean = sx / n, var = sq / n;
  const float m_old = state[j], v_old = state[(int64_t)J + j];
  state[j] = m_old + rate * (mean - m_old);
  state[(int64_t)J + j] = v_old + rate * (var - v_old);
}

__global__ void __launch_bounds__(256)
aa_norm_apply_kernel(const float* __restrict__ x, int64_t n_total, int J,
                     const float* __restrict__ mean, const float* __restrict__ var_num,
                     const float* __restrict__ var_den, float eps, float clip,
                     float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += stride) {
    const int j = (int)(i % J);
    float var = var_num[j];
    if (var_den) var = var / var_den[j];
    const float inv = 1.0f / sqrtf(var + eps);
    const float m = mean ? mean[j] : 0.0f;
    float y = x[i] * inv + (-m * inv);
    if (clip > 0.f) y = fminf(fmaxf(y, -clip), clip);
    out[i] = y;
  }
}

static int aa_norm_groups(int64_t n_outer, int J, int64_t* rows_per_group) {
  // at least 64 rows (or 16 K elements) per workgroup, at most AA_NORM_MAX_GROUPS workgroups
  int64_t min_rows = 16384 / (J > 0 ? J : 1);
  if (min_rows < 64) min_rows = 64;
  int64_t groups = (n_outer + min_rows - 1) / min_rows;
  if (groups > AA_NORM_MAX_GROUPS) groups = AA_NORM_MAX_GROUPS;
  if (groups < 1) groups = 1;
  int64_t rows = (n_outer + groups - 1) / groups;
  groups = (n_outer + rows - 1) / rows;
  *rows_per_group = rows;
  return (int)groups;
}

extern "C" {

int64_t aa_norm_scratch_floats(int64_t n_inner) {
  return n_inner > 0 ? (int64_t)AA_NORM_MAX_GROUPS * 2 * n_inner : 0;
}

int aa_streaming_norm_update(const float* x, int64_t n_outer, int64_t n_inner, float* state,
                             float* scratch, void* stream) {
  if (!x || !state || !scratch || n_outer <= 0 || n_inner <= 0) return AA_ERR_INVALID;
  if (n_inner > (1 << 24)) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  int64_t rows = 0;
  const int groups = aa_norm_groups(n_outer, (int)n_inner, &rows);
  hipLaunchKernelGGL(aa_norm_partial_kernel<0>, dim3(groups), dim3(AA_NORM_THREADS), 0, st, x,
                     n_outer, (int)n_inner, rows, (const float*)nullptr, scratch);
  hipLaunchKernelGGL(aa_streaming_norm_finish_kernel, dim3((unsigned)((n_inner + 63) / 64)),
                     dim3(64), 0, st, (const float*)scratch, groups, n_outer, rows, (int)n_inner,
                     state);
  return aa_launch_status();
}

int aa_ema_norm_update(const float* x, int64_t n_outer, int64_t n_inner, float rate, float* state,
                       float* scratch, void* stream) {
  if (!x || !state || !scratch || n_outer <= 0 || n_inner <= 0) return AA_ERR_INVALID;
  if (n_inner > (1 << 24)) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  int64_t rows = 0;
  const int groups = aa_norm_groups(n_outer, (int)n_inner, &rows);
  hipLaunchKernelGGL(aa_norm_partial_kernel<1>, dim3(groups), dim3(AA_NORM_THREADS), 0, st, x,
                     n_outer, (int)n_inner, rows, (const float*)state, scratch);
  hipLaunchKernelGGL(aa_ema_norm_finish_kernel, dim3((unsigned)((n_inner + 63) / 64)), dim3(64),
                     0, st, (const float*)scratch, groups, n_outer, (int)n_inner, rate, state);
  return aa_launch_status();
}

int aa_norm_apply(const float* x, int64_t n_outer, int64_t n_inner, const float* mean,
                  const float* var_num, const float* var_den, float variance_epsilon,
                  float clip_value, float* out, void* stream) {
  if (!x || !var_num || !out || n_outer < 0 || n_inner <= 0) return AA_ERR_INVALID;
  if (n_outer == 0) return AA_OK;
  const int64_t n = n_outer * n_inner;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aa_norm_apply_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, n, (int)n_inner, mean, var_num, var_den,
                     variance_epsilon, clip_value, out);
  return aa_launch_status();
}

}  // extern "C"
