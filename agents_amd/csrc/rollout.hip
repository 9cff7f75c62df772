// Collect-side kernels: epsilon-greedy action selection on Q values and a device-resident
// synthetic vector environment (no device->host->Python hop per env step, which is what
// tf_agents/environments/tf_py_environment.py:296-326 pays through tf.numpy_function).
//   EpsilonGreedyPolicy._action   tf_agents/policies/epsilon_greedy_policy.py:120-143
//   QPolicy._distribution         tf_agents/policies/q_policy.py:150-194
//   GreedyPolicy (mode = argmax)  tf_agents/policies/greedy_policy.py:70-89
//   RandomTFEnvironment           tf_agents/environments/random_tf_environment.py
//   auto-reset contract           tf_agents/environments/py_environment.py:233-239
#include "common.h"
#include <chrono>
#include "agents_amd.h"
#include <float.h>

// Stream: Philox(counter = (b_lo, b_hi, call_lo, call_hi), key = seed):
//   x0 -> u = u01(x0) ; explore iff u < epsilon ; random action = min + x1 mod A'
//   (A' = number of allowed actions under the mask; the x1 mod A'-th allowed action is taken)
template <bool I64>
__global__ void aa_eps_greedy_kernel(const float* __restrict__ q, const int32_t* __restrict__ mask,
                                     int64_t B, int A, float epsilon,
                                     const float* __restrict__ epsilon_dev, uint32_t k0,
                                     uint32_t k1, int64_t* call_dev, int64_t* arrival,
                                     int64_t action_min, void* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float eps = epsilon_dev != nullptr ? *epsilon_dev : epsilon;
  const uint64_t call = call_dev != nullptr ? (uint64_t)(*call_dev) : 0ull;
  if (arrival != nullptr) aa_advance_when_all_done(call_dev, arrival, 1, gridDim.x);
  if (b >= B) return;
  int best = 0;
  float bestv = 0.f;
  bool any = false;
  int allowed = 0;
  for (int a = 0; a < A; ++a) {
    float v = q[b * A + a];
    const bool ok = mask == nullptr || mask[b * A + a] != 0;
    if (!ok) v = -FLT_MAX;
    allowed += ok ? 1 : 0;
    if (!any || v > bestv) {
      best = a;
      bestv = v;
      any = true;
    }
  }
  int act = best;
  if (eps > 0.f) {
    const Philox4 r = philox4x32_10((uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), k0, k1);
    if (aa_u01(r.x) < eps && allowed > 0) {
      int pick = (int)(r.y % (uint32_t)allowed);
      for (int a = 0; a < A; ++a) {
        const bool ok = mask == nullptr || mask[b * A + a] != 0;
        if (ok) {
          if (pick == 0) {
            act = a;
            break;
          }
          --pick;
        }
      }
    }
  }
  const int64_t v = action_min + act;
  if (I64)
    reinterpret_cast<int64_t*>(out)[b] = v;
  else
    reinterpret_cast<int32_t*>(out)[b] = (int32_t)v;
}

// BoltzmannPolicy over a Q table (see include/agents_amd.h: aa_boltzmann_action).  The softmax CDF
// is evaluated in float64 so that the numpy restatement (oracle/policy.py) picks the same action
// unless u * sum lands within ~1e-16 relative of a partial sum.
template <bool I64>
__global__ void aa_boltzmann_kernel(const float* __restrict__ q, const int32_t* __restrict__ mask,
                                    int64_t B, int A, float temperature,
                                    const float* __restrict__ temperature_dev, uint32_t k0,
                                    uint32_t k1, int64_t* call_dev, int64_t* arrival,
                                    int64_t action_min, void* __restrict__ out,
                                    float* __restrict__ logits_out, int sample) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float T = temperature_dev != nullptr ? *temperature_dev : temperature;
  const uint64_t call = (sample && call_dev != nullptr) ? (uint64_t)(*call_dev) : 0ull;
  if (sample && arrival != nullptr) aa_advance_when_all_done(call_dev, arrival, 1, gridDim.x);
  if (b >= B) return;
  float lmax = -FLT_MAX;
  int last_ok = -1;
  for (int a = 0; a < A; ++a) {
    const bool ok = mask == nullptr || mask[b * A + a] != 0;
    const float l = ok ? q[b * A + a] / T : -FLT_MAX;
    if (logits_out != nullptr) logits_out[b * A + a] = l;
    if (ok) last_ok = a;
    lmax = l > lmax ? l : lmax;
  }
  if (!sample || out == nullptr) return;
  double total = 0.0;
  for (int a = 0; a < A; ++a) {
    const bool ok = mask == nullptr || mask[b * A + a] != 0;
    if (ok) total += exp((double)(q[b * A + a] / T) - (double)lmax);
  }
  const Philox4 r = philox4x32_10((uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)call,
                                  (uint32_t)(call >> 32), k0, k1);
  const double thr = (double)aa_u01(r.x) * total;
  int act = last_ok < 0 ? 0 : last_ok;
  double cum = 0.0;
  for (int a = 0; a < A; ++a) {
    const bool ok = mask == nullptr || mask[b * A + a] != 0;
    if (!ok) continue;
    cum += exp((double)(q[b * A + a] / T) - (double)lmax);
    if (cum > thr) {
      act = a;
      break;
    }
  }
  const int64_t v = action_min + act;
  if (I64)
    reinterpret_cast<int64_t*>(out)[b] = v;
  else
    reinterpret_cast<int32_t*>(out)[b] = (int32_t)v;
}

// Synthetic env.  Per env b and step s:
//   header draw  Philox(counter = (0xFFFFFFFF, b, s_lo, s_hi), key = seed):
//       x0 -> episode end iff u01(x0) < p_end ; x1 -> reward: u<.05 -> -1, u<.95 -> 0, else +1
//   observation  Philox(counter = (j, b, s_lo, s_hi)), j = 16-byte chunk index within the row
//       u8 : the 16 random bytes ;  f32 : lo + u01(x_i)*(hi-lo) for the 4 lanes
//   If the CURRENT step is LAST the env resets: FIRST, reward 0, discount 1 (action ignored).
//   Else: LAST (discount 0) with prob p_end, otherwise MID (discount 1).
__global__ void __launch_bounds__(256)
aa_vecenv_step_kernel(const int32_t* __restrict__ cur_step_type, int64_t B, int64_t obs_elems,
                      int obs_kind, float obs_lo, float obs_hi, float p_end, uint32_t k0,
                      uint32_t k1, int64_t* step_dev, int64_t* arrival, int force_first,
                      int32_t* __restrict__ step_type_out, float* __restrict__ reward_out,
                      float* __restrict__ discount_out, void* __restrict__ obs_out,
                      int64_t chunks_per_row) {
  const uint64_t s = step_dev != nullptr ? (uint64_t)(*step_dev) : 0ull;
  // every workgroup has the old value by the time the last arriver moves the counter (sharded
  // arrival words, common.h: a grid of ~1,800 groups on one word would serialise 20 us of atomics)
  if (arrival != nullptr) aa_advance_sharded(step_dev, arrival, 1, gridDim.x);
  const int64_t total = B * chunks_per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / chunks_per_row;
    const int64_t j = i - b * chunks_per_row;
    const Philox4 r = philox4x32_10((uint32_t)j, (uint32_t)b, (uint32_t)s, (uint32_t)(s >> 32),
                                    k0, k1);
    if (obs_kind == AA_OBS_U8) {
      uint8_t* row = reinterpret_cast<uint8_t*>(obs_out) + b * obs_elems;
      const int64_t e0 = j * 16;
      if (e0 + 16 <= obs_elems && ((obs_elems & 15) == 0)) {
        *reinterpret_cast<uint4*>(row + e0) = make_uint4(r.x, r.y, r.z, r.w);
      } else {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        for (int t = 0; t < 16 && e0 + t < obs_elems; ++t)
          row[e0 + t] = (uint8_t)(w[t >> 2] >> (8 * (t & 3)));
      }
    } else {
      float* row = reinterpret_cast<float*>(obs_out) + b * obs_elems;
      const int64_t e0 = j * 4;
      const uint32_t w[4] = {r.x, r.y, r.z, r.w};
      for (int t = 0; t < 4 && e0 + t < obs_elems; ++t)
        row[e0 + t] = obs_lo + aa_u01(w[t]) * (obs_hi - obs_lo);
    }
    if (j == 0) {
      const Philox4 h = philox4x32_10(0xFFFFFFFFu, (uint32_t)b, (uint32_t)s, (uint32_t)(s >> 32),
                                      k0, k1);
      const bool reset = force_first || cur_step_type[b] == 2;
      int32_t st;
      float rew, disc;
      if (reset) {
        st = 0; rew = 0.f; disc = 1.f;
      } else {
        const bool end = aa_u01(h.x) < p_end;
        const float u = aa_u01(h.y);
        rew = u < 0.05f ? -1.f : (u < 0.95f ? 0.f : 1.f);
        st = end ? 2 : 1;
        disc = end ? 0.f : 1.f;
      }
      step_type_out[b] = st;
      reward_out[b] = rew;
      discount_out[b] = disc;
    }
  }
}

__global__ void aa_counter_bump_kernel(int64_t* c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}

// DynamicStepDriver loop counter: counter[b] += (step_type[b] != LAST); *total += that sum
// (tf_agents/drivers/dynamic_step_driver.py:113,170).  One workgroup, deterministic.
__global__ void __launch_bounds__(256)
aa_count_steps_kernel(const int32_t* __restrict__ step_type, int64_t B,
                      int32_t* __restrict__ counter, int64_t* __restrict__ total,
                      int64_t* mailbox) {
  __shared__ float red[16];
  float s = 0.f;
  // eight elements per thread and pass, every load of a pass requested before the first use: the
  // one-element loop took a memory round trip per element (13 us for 4,096 environments on the
  // rocprofv3 timeline of the SAC loop; 2.4 us for 256)
  for (int64_t b0 = threadIdx.x; b0 < B; b0 += 8 * (int64_t)blockDim.x) {
    int st[8], cv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + u * (int64_t)blockDim.x;
      const int64_t bc = b < B ? b : B - 1;         // clamped: unconditional loads
      st[u] = step_type[bc];
      cv[u] = counter != nullptr ? counter[bc] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + u * (int64_t)blockDim.x;
      if (b < B) {
        const int inc = st[u] != 2 ? 1 : 0;
        if (counter != nullptr) counter[b] = cv[u] + inc;
        s += (float)inc;
      }
    }
  }
  const float t = aa_block_sum(s, red);
  if (threadIdx.x == 0) {
    const int64_t tot = *total + (int64_t)(t + 0.5f);
    *total = tot;
    if (mailbox != nullptr) {
      // host-visible: value, fence, then the sequence word the host spins on
      __hip_atomic_store(&mailbox[1], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      const int64_t seq = __hip_atomic_load(&mailbox[0], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_SYSTEM) + 1;
      __hip_atomic_store(&mailbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

extern "C" {

int aa_count_steps(const int32_t* step_type, int64_t B, int32_t* counter_dev, int64_t* total_dev,
                   int64_t* mailbox, void* stream) {
  if (!step_type || !total_dev || B <= 0 || B > (1 << 24)) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_count_steps_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_type,
                     B, counter_dev, total_dev, mailbox);
  return aa_launch_status();
}

int aa_mailbox_create(int64_t n_words, int64_t** host_ptr, int64_t** dev_ptr) {
  if (n_words <= 0 || !host_ptr || !dev_ptr) return AA_ERR_INVALID;
  void* h = nullptr;
  if (hipHostMalloc(&h, (size_t)n_words * sizeof(int64_t),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
    return AA_ERR_LAUNCH;
  for (int64_t i = 0; i < n_words; ++i) ((volatile int64_t*)h)[i] = 0;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipHostFree(h);
    return AA_ERR_LAUNCH;
  }
  *host_ptr = (int64_t*)h;
  *dev_ptr = (int64_t*)d;
  return AA_OK;
}

int aa_mailbox_destroy(int64_t* host_ptr) {
  if (!host_ptr) return AA_ERR_INVALID;
  return hipHostFree(host_ptr) == hipSuccess ? AA_OK : AA_ERR_LAUNCH;
}

int aa_mailbox_wait(const int64_t* host_ptr, int64_t seq, int64_t timeout_us, int64_t* value) {
  if (!host_ptr || !value) return AA_ERR_INVALID;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const int64_t cur = __atomic_load_n(&host_ptr[0], __ATOMIC_ACQUIRE);
    if (cur >= seq) {
      *value = __atomic_load_n(&host_ptr[1], __ATOMIC_RELAXED);
      return AA_OK;
    }
    if ((++spins & 0x3ff) == 0) {
      const auto dt = std::chrono::duration_cast<std::chrono::microseconds>(
                          std::chrono::steady_clock::now() - t0).count();
      if (dt > timeout_us) return AA_ERR_TIMEOUT;
    }
  }
}

int aa_eps_greedy_action(const float* q, const int32_t* mask, int64_t B, int32_t A, float epsilon,
                         const float* epsilon_dev, uint64_t seed, int64_t* call_counter_dev,
                         int64_t* arrival_dev, int64_t action_min, void* actions_out,
                         int32_t actions_are_i64, void* stream) {
  if (!q || !actions_out || B <= 0 || A <= 0) return AA_ERR_INVALID;
  const dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int64_t* arrival = (arrival_dev != nullptr && grid.x <= AA_MAX_ARRIVAL_GROUPS) ? arrival_dev
                                                                                 : nullptr;
  if (actions_are_i64)
    hipLaunchKernelGGL(aa_eps_greedy_kernel<true>, grid, block, 0, st, q, mask, B, A, epsilon,
                       epsilon_dev, (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev,
                       arrival, action_min, actions_out);
  else
    hipLaunchKernelGGL(aa_eps_greedy_kernel<false>, grid, block, 0, st, q, mask, B, A, epsilon,
                       epsilon_dev, (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev,
                       arrival, action_min, actions_out);
  if (arrival_dev != nullptr && arrival == nullptr)   // large grid: one-thread bump launch
    hipLaunchKernelGGL(aa_counter_bump_kernel, dim3(1), dim3(64), 0, st, call_counter_dev);
  return aa_launch_status();
}

int aa_boltzmann_action(const float* q, const int32_t* mask, int64_t B, int32_t A,
                        float temperature, const float* temperature_dev, uint64_t seed,
                        int64_t* call_counter_dev, int64_t* arrival_dev, int64_t action_min,
                        void* actions_out, int32_t actions_are_i64, float* logits_out,
                        int32_t sample, void* stream) {
  if (!q || B <= 0 || A <= 0 || (!actions_out && !logits_out)) return AA_ERR_INVALID;
  if (sample && (!actions_out || !call_counter_dev)) return AA_ERR_INVALID;
  if (temperature_dev == nullptr && !(temperature > 0.f)) return AA_ERR_INVALID;
  const dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  int64_t* arrival = (sample && arrival_dev != nullptr && grid.x <= AA_MAX_ARRIVAL_GROUPS)
                         ? arrival_dev : nullptr;
  if (actions_are_i64)
    hipLaunchKernelGGL(aa_boltzmann_kernel<true>, grid, block, 0, st, q, mask, B, A, temperature,
                       temperature_dev, (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev,
                       arrival, action_min, actions_out, logits_out, sample);
  else
    hipLaunchKernelGGL(aa_boltzmann_kernel<false>, grid, block, 0, st, q, mask, B, A, temperature,
                       temperature_dev, (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev,
                       arrival, action_min, actions_out, logits_out, sample);
  if (sample && arrival_dev != nullptr && arrival == nullptr)   // large grid: one-thread bump
    hipLaunchKernelGGL(aa_counter_bump_kernel, dim3(1), dim3(64), 0, st, call_counter_dev);
  return aa_launch_status();
}

int aa_vecenv_random_step(const int32_t* cur_step_type, int64_t B, int64_t obs_elems,
                          int32_t obs_kind, float obs_lo, float obs_hi, float p_end,
                          uint64_t seed, int64_t* step_counter_dev, int64_t* arrival_dev,
                          int32_t force_first, int32_t* step_type_out, float* reward_out,
                          float* discount_out, void* obs_out, void* stream) {
  if (B <= 0 || obs_elems <= 0 || !step_type_out || !reward_out || !discount_out || !obs_out)
    return AA_ERR_INVALID;
  if (!force_first && !cur_step_type) return AA_ERR_INVALID;
  if (obs_kind != AA_OBS_U8 && obs_kind != AA_OBS_F32) return AA_ERR_INVALID;
  if (B >= 0xFFFFFFFFLL) return AA_ERR_RANGE;
  const int64_t per = obs_kind == AA_OBS_U8 ? 16 : 4;
  const int64_t chunks = (obs_elems + per - 1) / per;
  if (chunks >= 0xFFFFFFFFLL) return AA_ERR_RANGE;
  if (obs_kind == AA_OBS_U8 && (obs_elems & 15) == 0 && (((uintptr_t)obs_out) & 15) != 0)
    return AA_ERR_INVALID;
  int64_t blocks = (B * chunks + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aa_vecenv_step_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, cur_step_type, B, obs_elems, obs_kind, obs_lo, obs_hi,
                     p_end, (uint32_t)seed, (uint32_t)(seed >> 32), step_counter_dev, arrival_dev,
                     force_first, step_type_out, reward_out, discount_out, obs_out, chunks);
  return aa_launch_status();
}

}  // extern "C"
