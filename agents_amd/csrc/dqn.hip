// DQN TD-target / element-wise loss / gradient kernel.
//
// Follows, op for op (fp32, no FMA contraction: this file is built with -ffp-contract=off so it
// agrees bit-for-bit with the numpy oracle in oracle/dqn.py):
//   trajectory.to_n_step_transition      tf_agents/trajectories/trajectory.py:716-850
//   value_ops.discounted_return (foldr)  tf_agents/utils/value_ops.py:21-99
//   common.index_with_actions            tf_agents/utils/common.py:367-411
//   compute_td_targets                   tf_agents/agents/dqn/dqn_agent.py:75-78
//   DqnAgent._td_loss / _loss            tf_agents/agents/dqn/dqn_agent.py:451-460, 462-579
//   DdqnAgent._compute_next_q_values     tf_agents/agents/dqn/dqn_agent.py:659-700
//   element_wise_huber_loss / squared    tf_agents/utils/common.py:1199-1208
//   aggregate_losses                     tf_agents/utils/common.py:1400-1476
//   QPolicy mask -> logits dtype.min     tf_agents/policies/q_policy.py:175-180
#include "common.h"
#include "agents_amd.h"
#include "dense_small_bodies.h"
#include <float.h>

template <bool I64>
__device__ static inline int64_t aa_load_action(const void* a, int64_t i) {
  if (I64) return reinterpret_cast<const int64_t*>(a)[i];
  return (int64_t) reinterpret_cast<const int32_t*>(a)[i];
}

// Everything the per-sample TD computation reads and the launch writes.
struct TdArgs {
  const float* q_online;
  const float* q_next_target;
  const float* q_next_select;   // nullable (Double DQN: the online net on the next observation)
  const int32_t* next_mask;     // nullable
  const void* actions;
  int64_t action_stride;
  const float* reward;
  const float* discount;
  const int32_t* step_type;
  const float* weights;         // nullable
  int64_t B;
  int T, A;
  float gamma, gpow, gamma_loss, reward_scale;
  int loss_kind;
  float global_batch;
  float* loss_out;
  float* td_loss_out;
  float* td_error_out;
  float* dq_out;
  float* field_sums_out;        // nullable
};

// One sample: n-step return, greedy next action, TD target, element-wise loss, mask, weight.
// Returns the taken action; gq = d(total loss) / d q_online[b, act].
template <bool I64>
__device__ static inline int64_t aa_td_sample(const TdArgs& P, int64_t b, float& loss,
                                              float& td_error, float& weighted, float& gq) {
  const int T = P.T, A = P.A, n = T - 1;
  // ---- n-step return over the first n frames (foldr: acc = acc*disc + r) ------------------
  float ret = 0.f;
  float dprod = 1.f;
  for (int t = n - 1; t >= 0; --t) {
    const float d = P.discount[b * T + t];
    ret = ret * (P.gamma * d) + P.reward[b * T + t];
  }
  for (int t = 0; t < n; ++t) dprod = dprod * P.discount[b * T + t];
  // gpow = float32(gamma ** (n-1)) evaluated in float64 on the host, like the python-float
  // constant the reference folds into the graph (trajectory.py:826-829).
  const float final_discount = P.gpow * dprod;

  // ---- greedy next action: first arg-max of the selecting net's (masked) Q ----------------
  const float* qsel = (P.q_next_select != nullptr ? P.q_next_select : P.q_next_target) + b * A;
  int best = 0;
  float bestv = -FLT_MAX;
  bool any = false;
  for (int a = 0; a < A; ++a) {
    float v = qsel[a];
    if (P.next_mask != nullptr && P.next_mask[b * A + a] == 0) v = -FLT_MAX;  // logits.dtype.min
    if (!any || v > bestv) {
      best = a;
      bestv = v;
      any = true;
    }
  }
  const float next_q = P.q_next_target[b * A + best];

  const int64_t act = aa_load_action<I64>(P.actions, b * P.action_stride);
  const float q = P.q_online[b * A + act];

  const float rewards = P.reward_scale * ret;
  const float discounts = P.gamma_loss * final_discount;  // DqnAgent._loss(gamma=...)
  const float td_target = rewards + discounts * next_q;
  td_error = td_target - q;
  float dloss_dq;
  if (P.loss_kind == AA_LOSS_TARGETS) {
    // the caller evaluates its own td_errors_loss_fn(td_targets, q_values) on these two vectors
    loss = td_target;
    td_error = q;
    weighted = 0.f;
    gq = 0.f;
    return act;
  }
  if (P.loss_kind == AA_LOSS_HUBER) {
    // tf.compat.v1.losses.huber_loss(labels=td_target, predictions=q, delta=1)
    const float err = q - td_target;
    const float abs_err = fabsf(err);
    const float quad = fminf(abs_err, 1.0f);
    const float lin = abs_err - quad;
    loss = 0.5f * quad * quad + 1.0f * lin;
    dloss_dq = err > 1.0f ? 1.0f : (err < -1.0f ? -1.0f : err);
  } else {
    const float err = td_target - q;  // mean_squared_error(labels, predictions) elementwise
    loss = err * err;
    dloss_dq = -2.0f * err;
  }
  const float valid = P.step_type[b * T + 0] != 2 ? 1.0f : 0.0f;  // ~time_steps.is_last()
  td_error = valid * td_error;
  loss = valid * loss;
  float w = 1.0f;
  weighted = loss;
  if (P.weights != nullptr) {
    w = P.weights[b];
    weighted = (w == 0.0f) ? 0.0f : loss * w;  // tf.math.multiply_no_nan
  }
  gq = (valid * dloss_dq * w) / P.global_batch;
  return act;
}

// All samples by one workgroup of 256 threads: per-sample outputs to global memory (write_out)
// and / or dq rows to `dq_rows` (LDS or global, [B][A]); the three sums are valid in thread 0.
template <bool I64>
__device__ static inline void aa_td_all(const TdArgs& P, bool write_out, float* dq_rows,
                                        float* red /* >= 16 floats */) {
  float local = 0.f, sum_loss = 0.f, sum_err = 0.f;
  for (int64_t b = threadIdx.x; b < P.B; b += blockDim.x) {
    float loss, td_error, weighted, gq;
    const int64_t act = aa_td_sample<I64>(P, b, loss, td_error, weighted, gq);
    if (write_out) {
      P.td_loss_out[b] = loss;
      P.td_error_out[b] = td_error;
      for (int a = 0; a < P.A; ++a) P.dq_out[b * P.A + a] = (a == act) ? gq : 0.f;
    }
    if (dq_rows != nullptr)
      for (int a = 0; a < P.A; ++a) dq_rows[b * P.A + a] = (a == act) ? gq : 0.f;
    local += weighted;
    sum_loss += loss;
    sum_err += td_error;
  }
  if (!write_out) return;
  const float total = aa_block_sum(local, red);
  if (threadIdx.x == 0) P.loss_out[0] = total / P.global_batch;
  if (P.field_sums_out != nullptr && P.loss_kind != AA_LOSS_TARGETS) {
    // the Learner's SUM over all axes of the LossInfo fields
    const float s0 = aa_block_sum(sum_loss, red);
    const float s1 = aa_block_sum(sum_err, red);
    if (threadIdx.x == 0) { P.field_sums_out[0] = s0; P.field_sums_out[1] = s1; }
  }
}

template <bool I64>
__global__ void __launch_bounds__(256) aa_dqn_td_loss_kernel(TdArgs P) {
  __shared__ float red[16];
  aa_td_all<I64>(P, true, nullptr, red);
}

// ---- TD loss + backward of the Q head in ONE launch ------------------------------------------------
// dL/dq only depends on a sample's own values (the mean's 1/B is a constant), so every workgroup of
// the head's backward launch (csrc/dense_small.hip: aa_dense_small_bwd_kernel -- the first n_dw
// workgroups the weight / bias gradient, the rest the input gradient) recomputes the [B, A] rows of
// dL/dq into LDS (B <= 512: one or two samples per thread, ~20 KB of L2 reads) and runs its
// unchanged body on them; workgroup 0 also writes the loss, td_loss / td_error, dq and the field
// sums.  Same bits as aa_dqn_td_loss followed by aa_dense_small_backward, one launch (and one
// graph node on the critical chain) less.
template <int N, bool I64>
__global__ void __launch_bounds__(256)
aa_dqn_loss_head_bwd_kernel(TdArgs P, const float* __restrict__ x, int64_t ldx,
                            const float* __restrict__ w, const float* __restrict__ mask_src,
                            int mask_kind, int K, float* __restrict__ dx, float* __restrict__ dw,
                            float* __restrict__ db, unsigned n_dw) {
  __shared__ float s_dq[512 * N];
  __shared__ float red[16];
  aa_td_all<I64>(P, blockIdx.x == 0, s_dq, red);
  __syncthreads();
  if (blockIdx.x < n_dw)   // uniform per workgroup: the barrier inside the body is safe
    aa_dense_small_dw_body<N>(x, ldx, s_dq, P.B, K, dw, db, blockIdx.x, n_dw);
  else
    aa_dense_small_dx_body<N>(s_dq, w, mask_src, mask_kind, P.B, K, dx, blockIdx.x - n_dw,
                              gridDim.x - n_dw);
}

static int aa_td_args(TdArgs* P, const float* q_online, const float* q_next_target,
                      const float* q_next_select, const int32_t* next_mask, const void* actions,
                      int64_t action_stride, const float* reward, const float* discount,
                      const int32_t* step_type, const float* weights, int64_t B, int32_t T,
                      int32_t A, double gamma, double gamma_loss, double reward_scale,
                      int32_t loss_kind, float global_batch, float* loss_out, float* td_loss_out,
                      float* td_error_out, float* dq_out, float* field_sums_out) {
  if (q_online == nullptr || q_next_target == nullptr || actions == nullptr || reward == nullptr ||
      discount == nullptr || step_type == nullptr || loss_out == nullptr ||
      td_loss_out == nullptr || td_error_out == nullptr || dq_out == nullptr)
    return AA_ERR_INVALID;
  if (B <= 0 || T < 2 || A <= 0 || !(global_batch > 0.f)) return AA_ERR_INVALID;
  if (loss_kind != AA_LOSS_HUBER && loss_kind != AA_LOSS_SQUARED && loss_kind != AA_LOSS_TARGETS)
    return AA_ERR_INVALID;
  // float32(gamma ** (n-1)) with the power taken in float64 (python-float semantics)
  double gp = 1.0;
  for (int t = 0; t < T - 2; ++t) gp *= gamma;
  *P = TdArgs{q_online, q_next_target, q_next_select, next_mask, actions, action_stride, reward,
              discount, step_type, weights, B, T, A, (float)gamma, (float)gp, (float)gamma_loss,
              (float)reward_scale, loss_kind, global_batch, loss_out, td_loss_out, td_error_out,
              dq_out, field_sums_out};
  return AA_OK;
}

extern "C" int aa_dqn_td_loss_sums(const float* q_online, const float* q_next_target,
                              const float* q_next_select, const int32_t* next_mask,
                              const void* actions, int32_t actions_are_i64, int64_t action_stride,
                              const float* reward, const float* discount,
                              const int32_t* step_type, const float* weights, int64_t B,
                              int32_t T, int32_t A, double gamma, double gamma_loss,
                              double reward_scale,
                              int32_t loss_kind, float global_batch, float* loss_out,
                              float* td_loss_out, float* td_error_out, float* dq_out,
                              float* field_sums_out, void* stream) {
  TdArgs P;
  const int rc = aa_td_args(&P, q_online, q_next_target, q_next_select, next_mask, actions,
                            action_stride, reward, discount, step_type, weights, B, T, A, gamma,
                            gamma_loss, reward_scale, loss_kind, global_batch, loss_out,
                            td_loss_out, td_error_out, dq_out, field_sums_out);
  if (rc != AA_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (actions_are_i64)
    hipLaunchKernelGGL(aa_dqn_td_loss_kernel<true>, dim3(1), dim3(256), 0, st, P);
  else
    hipLaunchKernelGGL(aa_dqn_td_loss_kernel<false>, dim3(1), dim3(256), 0, st, P);
  return aa_launch_status();
}

// aa_dqn_td_loss_sums + aa_dense_small_backward of the Q head (x[B,K] = the head's input, w[K,A]
// its kernel, mask_src / mask_kind = the previous layer's activation derivative, dx[B,K], dw[K,A],
// db[A] nullable) in one launch; B <= 512, A <= 16.
extern "C" int aa_dqn_loss_head_backward(
    const float* q_online, const float* q_next_target, const float* q_next_select,
    const int32_t* next_mask, const void* actions, int32_t actions_are_i64, int64_t action_stride,
    const float* reward, const float* discount, const int32_t* step_type, const float* weights,
    int64_t B, int32_t T, int32_t A, double gamma, double gamma_loss, double reward_scale,
    int32_t loss_kind, float global_batch, float* loss_out, float* td_loss_out,
    float* td_error_out, float* dq_out, float* field_sums_out, const float* x, int64_t ldx,
    const float* w, const float* mask_src, int32_t mask_kind, int32_t K, float* dx, float* dw,
    float* db, void* stream) {
  TdArgs P;
  const int rc = aa_td_args(&P, q_online, q_next_target, q_next_select, next_mask, actions,
                            action_stride, reward, discount, step_type, weights, B, T, A, gamma,
                            gamma_loss, reward_scale, loss_kind, global_batch, loss_out,
                            td_loss_out, td_error_out, dq_out, field_sums_out);
  if (rc != AA_OK) return rc;
  if (loss_kind == AA_LOSS_TARGETS) return AA_ERR_INVALID;   // no dL/dq to back-propagate
  if (!x || !w || !dx || !dw || K <= 0 || ldx < K) return AA_ERR_INVALID;
  if (A > AA_SMALLN_MAX || B > 512) return AA_ERR_RANGE;
  const unsigned n_dw = (unsigned)((K + 63) / 64);
  int64_t n_dx = (B * K + 255) / 256;
  if (n_dx > 2048) n_dx = 2048;
  const int mk = mask_src ? mask_kind : 0;
  hipStream_t st = (hipStream_t)stream;
#define AA_LHB(NN)                                                                               \
  case NN:                                                                                       \
    if (actions_are_i64)                                                                         \
      hipLaunchKernelGGL((aa_dqn_loss_head_bwd_kernel<NN, true>), dim3(n_dw + (unsigned)n_dx),   \
                         dim3(256), 0, st, P, x, ldx, w, mask_src, mk, K, dx, dw, db, n_dw);     \
    else                                                                                         \
      hipLaunchKernelGGL((aa_dqn_loss_head_bwd_kernel<NN, false>), dim3(n_dw + (unsigned)n_dx),  \
                         dim3(256), 0, st, P, x, ldx, w, mask_src, mk, K, dx, dw, db, n_dw);     \
    break;
  switch (A) {
    AA_LHB(1) AA_LHB(2) AA_LHB(3) AA_LHB(4) AA_LHB(5) AA_LHB(6) AA_LHB(7) AA_LHB(8) AA_LHB(9)
    AA_LHB(10) AA_LHB(11) AA_LHB(12) AA_LHB(13) AA_LHB(14) AA_LHB(15) AA_LHB(16)
    default: return AA_ERR_RANGE;
  }
#undef AA_LHB
  return aa_launch_status();
}

extern "C" int aa_dqn_td_loss(const float* q_online, const float* q_next_target,
                              const float* q_next_select, const int32_t* next_mask,
                              const void* actions, int32_t actions_are_i64, int64_t action_stride,
                              const float* reward, const float* discount,
                              const int32_t* step_type, const float* weights, int64_t B,
                              int32_t T, int32_t A, double gamma, double gamma_loss,
                              double reward_scale, int32_t loss_kind, float global_batch,
                              float* loss_out, float* td_loss_out, float* td_error_out,
                              float* dq_out, void* stream) {
  return aa_dqn_td_loss_sums(q_online, q_next_target, q_next_select, next_mask, actions,
                             actions_are_i64, action_stride, reward, discount, step_type, weights,
                             B, T, A, gamma, gamma_loss, reward_scale, loss_kind, global_batch,
                             loss_out, td_loss_out, td_error_out, dq_out, nullptr, stream);
}
