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
#include <float.h>

template <bool I64>
__device__ static inline int64_t aa_load_action(const void* a, int64_t i) {
  if (I64) return reinterpret_cast<const int64_t*>(a)[i];
  return (int64_t) reinterpret_cast<const int32_t*>(a)[i];
}

template <bool I64>
__global__ void __launch_bounds__(256)
aa_dqn_td_loss_kernel(const float* __restrict__ q_online, const float* __restrict__ q_next_target,
                      const float* __restrict__ q_next_select,
                      const int32_t* __restrict__ next_mask, const void* __restrict__ actions,
                      int64_t action_stride, const float* __restrict__ reward,
                      const float* __restrict__ discount, const int32_t* __restrict__ step_type,
                      const float* __restrict__ weights, int64_t B, int T, int A, float gamma,
                      float gpow, float gamma_loss,
                      float reward_scale, int loss_kind, float global_batch,
                      float* __restrict__ loss_out, float* __restrict__ td_loss_out,
                      float* __restrict__ td_error_out, float* __restrict__ dq_out,
                      float* __restrict__ field_sums_out) {
  __shared__ float red[16];
  const int n = T - 1;
  float local = 0.f, sum_loss = 0.f, sum_err = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
    // ---- n-step return over the first n frames (foldr: acc = acc*disc + r) ------------------
    float ret = 0.f;
    float dprod = 1.f;
    for (int t = n - 1; t >= 0; --t) {
      const float d = discount[b * T + t];
      ret = ret * (gamma * d) + reward[b * T + t];
    }
    for (int t = 0; t < n; ++t) dprod = dprod * discount[b * T + t];
    // gpow = float32(gamma ** (n-1)) evaluated in float64 on the host, like the python-float
    // constant the reference folds into the graph (trajectory.py:826-829).
    const float final_discount = gpow * dprod;

    // ---- greedy next action: first arg-max of the selecting net's (masked) Q ----------------
    const float* qsel = (q_next_select != nullptr ? q_next_select : q_next_target) + b * A;
    int best = 0;
    float bestv = -FLT_MAX;
    bool any = false;
    for (int a = 0; a < A; ++a) {
      float v = qsel[a];
      if (next_mask != nullptr && next_mask[b * A + a] == 0) v = -FLT_MAX;  // logits.dtype.min
      if (!any || v > bestv) {
        best = a;
        bestv = v;
        any = true;
      }
    }
    const float next_q = q_next_target[b * A + best];

    const int64_t act = aa_load_action<I64>(actions, b * action_stride);
    const float q = q_online[b * A + act];

    const float rewards = reward_scale * ret;
    const float discounts = gamma_loss * final_discount;  // DqnAgent._loss(gamma=...)
    const float td_target = rewards + discounts * next_q;
    float td_error = td_target - q;
    float loss, dloss_dq;
    if (loss_kind == AA_LOSS_HUBER) {
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
    const float valid = step_type[b * T + 0] != 2 ? 1.0f : 0.0f;  // ~time_steps.is_last()
    td_error = valid * td_error;
    loss = valid * loss;
    float w = 1.0f;
    float weighted = loss;
    if (weights != nullptr) {
      w = weights[b];
      weighted = (w == 0.0f) ? 0.0f : loss * w;  // tf.math.multiply_no_nan
    }
    td_loss_out[b] = loss;
    td_error_out[b] = td_error;
    local += weighted;
    sum_loss += loss;
    sum_err += td_error;
    const float gq = (valid * dloss_dq * w) / global_batch;
    for (int a = 0; a < A; ++a) dq_out[b * A + a] = (a == act) ? gq : 0.f;
  }
  const float total = aa_block_sum(local, red);
  if (threadIdx.x == 0) loss_out[0] = total / global_batch;
  if (field_sums_out != nullptr) {   // the Learner's SUM over all axes of the LossInfo fields
    const float s0 = aa_block_sum(sum_loss, red);
    const float s1 = aa_block_sum(sum_err, red);
    if (threadIdx.x == 0) { field_sums_out[0] = s0; field_sums_out[1] = s1; }
  }
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
  if (q_online == nullptr || q_next_target == nullptr || actions == nullptr || reward == nullptr ||
      discount == nullptr || step_type == nullptr || loss_out == nullptr ||
      td_loss_out == nullptr || td_error_out == nullptr || dq_out == nullptr)
    return AA_ERR_INVALID;
  if (B <= 0 || T < 2 || A <= 0 || !(global_batch > 0.f)) return AA_ERR_INVALID;
  if (loss_kind != AA_LOSS_HUBER && loss_kind != AA_LOSS_SQUARED) return AA_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  // float32(gamma ** (n-1)) with the power taken in float64 (python-float semantics)
  double gp = 1.0;
  for (int t = 0; t < T - 2; ++t) gp *= gamma;
  const float gpow = (float)gp;
  if (actions_are_i64)
    hipLaunchKernelGGL(aa_dqn_td_loss_kernel<true>, dim3(1), dim3(256), 0, st, q_online,
                       q_next_target, q_next_select, next_mask, actions, action_stride, reward,
                       discount, step_type, weights, B, T, A, (float)gamma, gpow, (float)gamma_loss,
                       (float)reward_scale, loss_kind,
                       global_batch, loss_out, td_loss_out, td_error_out, dq_out, field_sums_out);
  else
    hipLaunchKernelGGL(aa_dqn_td_loss_kernel<false>, dim3(1), dim3(256), 0, st, q_online,
                       q_next_target, q_next_select, next_mask, actions, action_stride, reward,
                       discount, step_type, weights, B, T, A, (float)gamma, gpow, (float)gamma_loss,
                       (float)reward_scale, loss_kind,
                       global_batch, loss_out, td_loss_out, td_error_out, dq_out, field_sums_out);
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
