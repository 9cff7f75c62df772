// Fused flat-buffer optimizers, target-network soft update, gradient clipping.  HBM-bound
// elementwise kernels over the flat fp32 parameter / gradient / slot buffers
// (16-byte vectors, grid-stride).  Built with -ffp-contract=off: op order is the oracle's.
//   keras Adam / RMSprop apply (agents/dqn/examples/v2/train_eval.py:180,
//                               examples/dqn/mnih15/dqn_train_eval_atari.py:176-182)
//   common.soft_variables_update   tf_agents/utils/common.py:250-346
//   eager_utils.clip_gradient_norms tf_agents/utils/eager_utils.py:227-246 (per tensor)
//   tf.clip_by_global_norm          tf_agents/agents/ppo/ppo_agent.py:948-949
#include "common.h"
#include "agents_amd.h"
#include "adam_elem.h"
#include "x6_common.h"
#include "splitk_reduce.h"

#define AA_EW_THREADS 256

// ---- split planes kept current by the optimizer ------------------------------------------------
// The bf16x6 convolutions read their filters as three bf16 planes in MFMA-fragment order
// (conv_pair_x6.hip / conv_dx_frame_x6.hip: "split once").  Re-splitting in a pre-pass per forward /
// backward costs five launches per DQN iteration; re-splitting in launches of their own after the
// optimizer step costs three on the one point every stream waits for (measured: -12 %).  Here the
// optimizer kernel, which holds every new parameter value in a register anyway, writes its three
// pieces straight to where each consumer will read them: pos[i - lo] is the bf16 index of the hi
// piece of parameter i in `planes`, the mid / lo pieces follow at +stride / +2*stride (-1: this
// parameter is not part of the target).  Same rounding sequence as cx_split8.
__device__ static inline void aa_planes_put(const aa_plane_scatter& S, int64_t i, float v) {
#pragma unroll
  for (int t = 0; t < AA_MAX_PLANE_TARGETS; ++t) {
    if (t < S.n && i >= S.lo[t] && i < S.hi[t]) {
      const int32_t pos = S.pos[t][i - S.lo[t]];
      if (pos >= 0) {
        uint16_t* base = S.planes[t] + pos;
        float r = v;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const unsigned pk = cx_pk_bf16(r, 0.f);
          base[(int64_t)s * S.stride[t]] = (uint16_t)(pk & 0xffffu);
          r -= __uint_as_float(pk << 16);
        }
      }
    }
  }
}

__device__ static inline bool aa_planes_touch(const aa_plane_scatter& S, int64_t i0, int64_t i1) {
  bool hit = false;
#pragma unroll
  for (int t = 0; t < AA_MAX_PLANE_TARGETS; ++t)
    hit = hit || (t < S.n && i1 > S.lo[t] && i0 < S.hi[t]);
  return hit;
}
static inline unsigned aa_ew_blocks(int64_t n_vec) {
  int64_t b = (n_vec + AA_EW_THREADS - 1) / AA_EW_THREADS;
  if (b > 2048) b = 2048;  // 256 CUs x 8
  if (b < 1) b = 1;
  return (unsigned)b;
}


// `arrival` != nullptr: *step_dev holds the number of steps taken so far; every workgroup reads it
// at its start (t = that + 1) and the LAST one to finish advances it -- instead of a one-thread
// aa_counter_add launch in front of every optimizer step (three per SAC iteration).
template <bool PLANES>
__global__ void __launch_bounds__(AA_EW_THREADS)
aa_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
               float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
               int64_t* __restrict__ step_dev, int64_t* __restrict__ arrival,
               aa_plane_scatter S, float* __restrict__ target, float tau) {
  __shared__ float s_alpha;
  if (threadIdx.x == 0) {
    const float t = (float)(*step_dev + (arrival != nullptr ? 1 : 0));
    s_alpha = adam_alpha(lr, beta1, beta2, t);
  }
  __syncthreads();
  const float alpha = s_alpha, omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  const int64_t nv = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam_elem(pp.x, gg.x, mm.x, vv.x, alpha, omb1, omb2, eps);
    adam_elem(pp.y, gg.y, mm.y, vv.y, alpha, omb1, omb2, eps);
    adam_elem(pp.z, gg.z, mm.z, vv.z, alpha, omb1, omb2, eps);
    adam_elem(pp.w, gg.w, mm.w, vv.w, alpha, omb1, omb2, eps);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (target != nullptr) {
      // soft_variables_update of the target copy from the parameters just written
      // (aa_soft_update_kernel's expression), in the same pass
      float4 a = reinterpret_cast<float4*>(target)[i];
      a.x = soft_update_elem(a.x, pp.x, tau);
      a.y = soft_update_elem(a.y, pp.y, tau);
      a.z = soft_update_elem(a.z, pp.z, tau);
      a.w = soft_update_elem(a.w, pp.w, tau);
      reinterpret_cast<float4*>(target)[i] = a;
    }
    if (PLANES && aa_planes_touch(S, 4 * i, 4 * i + 4)) {
      aa_planes_put(S, 4 * i, pp.x); aa_planes_put(S, 4 * i + 1, pp.y);
      aa_planes_put(S, 4 * i + 2, pp.z); aa_planes_put(S, 4 * i + 3, pp.w);
    }
  }
  for (int64_t i = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pn = adam_elem(p[i], g[i], m[i], v[i], alpha, omb1, omb2, eps);
    if (PLANES) aa_planes_put(S, i, pn);
    if (target != nullptr) target[i] = (1.0f - tau) * target[i] + tau * pn;
  }
  if (arrival != nullptr) aa_advance_when_all_done(step_dev, arrival, 1, gridDim.x);
}

template <bool CENTERED, bool MOMENTUM>
__device__ static inline void rms_elem(float& p, float g, float& ms, float* mg, float* mom,
                                       float lr, float rho, float omr, float momentum,
                                       float eps) {
  ms = rho * ms + omr * (g * g);
  float denom;
  if (CENTERED) {
    *mg = rho * (*mg) + omr * g;
    denom = ms - (*mg) * (*mg) + eps;
  } else {
    denom = ms + eps;
  }
  const float inc = lr * g * (1.0f / sqrtf(denom));
  if (MOMENTUM) {
    *mom = momentum * (*mom) + inc;
    p = p - *mom;
  } else {
    p = p - inc;
  }
}

template <bool CENTERED, bool MOMENTUM, bool PLANES>
__global__ void __launch_bounds__(AA_EW_THREADS)
aa_rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                  float* __restrict__ mg, float* __restrict__ mom, int64_t n, float lr, float rho,
                  float momentum, float eps, aa_plane_scatter S) {
  const float omr = 1.0f - rho;
  const int64_t nv = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 s = reinterpret_cast<float4*>(ms)[i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), mo = make_float4(0.f, 0.f, 0.f, 0.f);
    if (CENTERED) a = reinterpret_cast<float4*>(mg)[i];
    if (MOMENTUM) mo = reinterpret_cast<float4*>(mom)[i];
    rms_elem<CENTERED, MOMENTUM>(pp.x, gg.x, s.x, &a.x, &mo.x, lr, rho, omr, momentum, eps);
    rms_elem<CENTERED, MOMENTUM>(pp.y, gg.y, s.y, &a.y, &mo.y, lr, rho, omr, momentum, eps);
    rms_elem<CENTERED, MOMENTUM>(pp.z, gg.z, s.z, &a.z, &mo.z, lr, rho, omr, momentum, eps);
    rms_elem<CENTERED, MOMENTUM>(pp.w, gg.w, s.w, &a.w, &mo.w, lr, rho, omr, momentum, eps);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(ms)[i] = s;
    if (CENTERED) reinterpret_cast<float4*>(mg)[i] = a;
    if (MOMENTUM) reinterpret_cast<float4*>(mom)[i] = mo;
    if (PLANES && aa_planes_touch(S, 4 * i, 4 * i + 4)) {
      aa_planes_put(S, 4 * i, pp.x); aa_planes_put(S, 4 * i + 1, pp.y);
      aa_planes_put(S, 4 * i + 2, pp.z); aa_planes_put(S, 4 * i + 3, pp.w);
    }
  }
  for (int64_t i = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = CENTERED ? mg[i] : 0.f, mo = MOMENTUM ? mom[i] : 0.f;
    rms_elem<CENTERED, MOMENTUM>(p[i], g[i], ms[i], &a, &mo, lr, rho, omr, momentum, eps);
    if (CENTERED) mg[i] = a;
    if (MOMENTUM) mom[i] = mo;
    if (PLANES) aa_planes_put(S, i, p[i]);
  }
}

// RMSprop reading some gradients straight from split-K slabs.  The conv weight gradients of the
// DQN step leave per-frame-group slabs (conv_dw_frame_x6.hip, conv_u8_bf16.h) that a reduce launch
// used to sum into the flat gradient buffer right before this kernel read it back: when nothing
// sits between backward and the optimizer (no clipping, no all-reduce) the leading workgroups of
// the optimizer launch do that sum themselves -- aa_splitk_reduce_walk, i.e. the association of the
// reduce launch, bit for bit -- and update the segment's parameters with it; the other workgroups
// walk the rest of the flat buffer as aa_rmsprop_kernel does.  The sums are also stored to g (0.3 MB
// for the Atari net), so that the flat gradient buffer is complete once the step has run.
struct AaSlabSrc {
  int n;
  int first[AA_MAX_GRAD_SLABS + 1];          // workgroups [first[s], first[s + 1]) sum segment s
  const float* slab[AA_MAX_GRAD_SLABS];
  int splits[AA_MAX_GRAD_SLABS], mn[AA_MAX_GRAD_SLABS], n_tail[AA_MAX_GRAD_SLABS];
  long long off[AA_MAX_GRAD_SLABS];
};

// Up to eight fp32 scalars copied side by side by the launch's last workgroup: the Learner's
// reduced LossInfo (train/learner.py:322-337) leaves with the optimizer launch instead of with a
// launch of its own behind it (one node less on the stream every lane of the next iteration waits
// for).  The sources were written by the loss launch, long before this one.
#define AA_MAX_PACK 8
struct AaPack {
  const float* src[AA_MAX_PACK];
  float* dst;
  int n;
};

template <bool CENTERED, bool MOMENTUM, bool PLANES>
__global__ void __launch_bounds__(AA_EW_THREADS)
aa_rmsprop_slabs_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ ms,
                        float* __restrict__ mg, float* __restrict__ mom, int64_t n, float lr,
                        float rho, float momentum, float eps, aa_plane_scatter S, AaSlabSrc G,
                        AaPack K) {
  if (K.n > 0 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < K.n) {
    const float* src = K.src[0];
#pragma unroll
    for (int j = 1; j < AA_MAX_PACK; ++j)
      if ((int)threadIdx.x == j) src = K.src[j];
    K.dst[threadIdx.x] = *src;
  }
  const float omr = 1.0f - rho;
  auto update4 = [&](int64_t i, const float4 gg) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 s = reinterpret_cast<float4*>(ms)[i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), mo = make_float4(0.f, 0.f, 0.f, 0.f);
    if (CENTERED) a = reinterpret_cast<float4*>(mg)[i];
    if (MOMENTUM) mo = reinterpret_cast<float4*>(mom)[i];
    rms_elem<CENTERED, MOMENTUM>(pp.x, gg.x, s.x, &a.x, &mo.x, lr, rho, omr, momentum, eps);
    rms_elem<CENTERED, MOMENTUM>(pp.y, gg.y, s.y, &a.y, &mo.y, lr, rho, omr, momentum, eps);
    rms_elem<CENTERED, MOMENTUM>(pp.z, gg.z, s.z, &a.z, &mo.z, lr, rho, omr, momentum, eps);
    rms_elem<CENTERED, MOMENTUM>(pp.w, gg.w, s.w, &a.w, &mo.w, lr, rho, omr, momentum, eps);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(ms)[i] = s;
    if (CENTERED) reinterpret_cast<float4*>(mg)[i] = a;
    if (MOMENTUM) reinterpret_cast<float4*>(mom)[i] = mo;
    if (PLANES && aa_planes_touch(S, 4 * i, 4 * i + 4)) {
      aa_planes_put(S, 4 * i, pp.x); aa_planes_put(S, 4 * i + 1, pp.y);
      aa_planes_put(S, 4 * i + 2, pp.z); aa_planes_put(S, 4 * i + 3, pp.w);
    }
  };
  const int slab_blocks = G.first[G.n];
  if ((int)blockIdx.x < slab_blocks) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < AA_MAX_GRAD_SLABS; ++k)
      if (k < G.n && (int)blockIdx.x >= G.first[k]) s = k;
    const float* slab = G.slab[0]; int splits = G.splits[0], mn = G.mn[0], nt = G.n_tail[0];
    long long off = G.off[0]; int b0 = G.first[0], b1 = G.first[1];
#pragma unroll
    for (int k = 1; k < AA_MAX_GRAD_SLABS; ++k)
      if (s == k) {
        slab = G.slab[k]; splits = G.splits[k]; mn = G.mn[k]; nt = G.n_tail[k];
        off = G.off[k]; b0 = G.first[k]; b1 = G.first[k + 1];
      }
    aa_splitk_reduce_walk<4, 16>(slab, splits, (size_t)mn, nt, blockIdx.x - (unsigned)b0,
                                 (unsigned)(b1 - b0), [&](size_t i, bool, float (&v)[4]) {
                                   const float4 gg = make_float4(v[0], v[1], v[2], v[3]);
                                   reinterpret_cast<float4*>(g)[(off + (long long)i) >> 2] = gg;
                                   update4((off + (long long)i) >> 2, gg);
                                 });
    return;
  }
  const int64_t nv = n / 4;
  const int64_t fb = (int64_t)blockIdx.x - slab_blocks;
  const int64_t stride = ((int64_t)gridDim.x - slab_blocks) * blockDim.x;
  for (int64_t i = fb * blockDim.x + threadIdx.x; i < nv; i += stride) {
    bool covered = false;
#pragma unroll
    for (int k = 0; k < AA_MAX_GRAD_SLABS; ++k)
      covered = covered || (k < G.n && 4 * i >= G.off[k] && 4 * i < G.off[k] + G.mn[k] + G.n_tail[k]);
    if (covered) continue;
    update4(i, reinterpret_cast<const float4*>(g)[i]);
  }
  for (int64_t i = nv * 4 + fb * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = CENTERED ? mg[i] : 0.f, mo = MOMENTUM ? mom[i] : 0.f;
    rms_elem<CENTERED, MOMENTUM>(p[i], g[i], ms[i], &a, &mo, lr, rho, omr, momentum, eps);
    if (CENTERED) mg[i] = a;
    if (MOMENTUM) mom[i] = mo;
    if (PLANES) aa_planes_put(S, i, p[i]);
  }
}

__global__ void __launch_bounds__(AA_EW_THREADS)
aa_sgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = p[i] - lr * g[i];
}

__global__ void __launch_bounds__(AA_EW_THREADS)
aa_soft_update_kernel(float* __restrict__ t, const float* __restrict__ s, int64_t n, float tau) {
  const float omt = 1.0f - tau;
  const int64_t nv = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float4 a = reinterpret_cast<float4*>(t)[i];
    const float4 b = reinterpret_cast<const float4*>(s)[i];
    a.x = omt * a.x + tau * b.x;
    a.y = omt * a.y + tau * b.y;
    a.z = omt * a.z + tau * b.z;
    a.w = omt * a.w + tau * b.w;
    reinterpret_cast<float4*>(t)[i] = a;
  }
  for (int64_t i = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    t[i] = omt * t[i] + tau * s[i];
}

// one workgroup per segment (tensor); deterministic tree
__global__ void __launch_bounds__(1024)
aa_segment_sumsq_kernel(const float* __restrict__ g, const int64_t* __restrict__ off,
                        float* __restrict__ out) {
  __shared__ float red[16];
  const int64_t lo = off[blockIdx.x], hi = off[blockIdx.x + 1];
  float s = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) s += g[i] * g[i];
  const float t = aa_block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}

__global__ void __launch_bounds__(AA_EW_THREADS)
aa_clip_kernel(float* __restrict__ g, const int64_t* __restrict__ off, int n_seg,
               const float* __restrict__ sumsq, float clip, int per_tensor) {
  const int seg = blockIdx.y;
  const int64_t lo = off[seg], hi = off[seg + 1];
  float scale;
  if (per_tensor) {
    // tf.clip_by_norm: t * clip_norm / max(l2norm, clip_norm)
    const float nrm = sqrtf(sumsq[seg]);
    scale = clip / fmaxf(nrm, clip);
  } else {
    // tf.clip_by_global_norm: scale = clip_norm * min(1/global_norm, 1/clip_norm)
    float tot = 0.f;
    for (int i = 0; i < n_seg; ++i) tot += sumsq[i];
    const float gn = sqrtf(tot);
    scale = clip * fminf(1.0f / gn, 1.0f / clip);
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride)
    g[i] = g[i] * scale;
}

extern "C" {

static int aa_planes_check(const aa_plane_scatter* S, int64_t n) {
  if (S == nullptr) return AA_OK;
  if (S->n < 0 || S->n > AA_MAX_PLANE_TARGETS) return AA_ERR_INVALID;
  for (int t = 0; t < S->n; ++t)
    if (S->pos[t] == nullptr || S->planes[t] == nullptr || S->lo[t] < 0 || S->hi[t] > n ||
        S->lo[t] >= S->hi[t] || S->stride[t] <= 0)
      return AA_ERR_INVALID;
  return AA_OK;
}

static int aa_adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                          float beta1, float beta2, float eps, int64_t* step_dev, int64_t* arrival,
                          const aa_plane_scatter* planes, void* stream, float* target = nullptr,
                          float tau = 0.f) {
  if (!p || !g || !m || !v || !step_dev || n <= 0) return AA_ERR_INVALID;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)target) & 15) != 0)
    return AA_ERR_INVALID;
  if (aa_planes_check(planes, n) != AA_OK) return AA_ERR_INVALID;
  const dim3 grid(aa_ew_blocks(n / 4)), block(AA_EW_THREADS);
  if (planes != nullptr && planes->n > 0)
    hipLaunchKernelGGL(aa_adam_kernel<true>, grid, block, 0, (hipStream_t)stream, p, g, m, v, n,
                       lr, beta1, beta2, eps, step_dev, arrival, *planes, target, tau);
  else
    hipLaunchKernelGGL(aa_adam_kernel<false>, grid, block, 0, (hipStream_t)stream, p, g, m, v, n,
                       lr, beta1, beta2, eps, step_dev, arrival, aa_plane_scatter{}, target, tau);
  return aa_launch_status();
}

int aa_adam_step_planes(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                        float beta1, float beta2, float eps, const int64_t* step_dev,
                        const aa_plane_scatter* planes, void* stream) {
  return aa_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, const_cast<int64_t*>(step_dev),
                        nullptr, planes, stream);
}

int aa_adam_step_counted(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                         float beta1, float beta2, float eps, int64_t* steps_taken_dev,
                         int64_t* arrival_dev, const aa_plane_scatter* planes, void* stream) {
  if (arrival_dev == nullptr) return AA_ERR_INVALID;
  return aa_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, steps_taken_dev, arrival_dev, planes,
                        stream);
}

int aa_adam_step_counted_target(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                float beta1, float beta2, float eps, int64_t* steps_taken_dev,
                                int64_t* arrival_dev, float* target, float tau, void* stream) {
  if (arrival_dev == nullptr || target == nullptr) return AA_ERR_INVALID;
  return aa_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, steps_taken_dev, arrival_dev,
                        nullptr, stream, target, tau);
}

int aa_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, const int64_t* step_dev, void* stream) {
  return aa_adam_step_planes(p, g, m, v, n, lr, beta1, beta2, eps, step_dev, nullptr, stream);
}

int aa_rmsprop_step_planes(float* p, const float* g, float* ms, float* mg, float* mom, int64_t n,
                           float lr, float rho, float momentum, float eps,
                           const aa_plane_scatter* planes, void* stream) {
  if (!p || !g || !ms || n <= 0) return AA_ERR_INVALID;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)ms | (uintptr_t)mg | (uintptr_t)mom) & 15) != 0)
    return AA_ERR_INVALID;
  if (aa_planes_check(planes, n) != AA_OK) return AA_ERR_INVALID;
  const dim3 grid(aa_ew_blocks(n / 4)), block(AA_EW_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const bool pl = planes != nullptr && planes->n > 0;
  const aa_plane_scatter S = pl ? *planes : aa_plane_scatter{};
#define AA_RMS(C_, M_)                                                                          \
  do {                                                                                          \
    if (pl)                                                                                     \
      hipLaunchKernelGGL((aa_rmsprop_kernel<C_, M_, true>), grid, block, 0, st, p, g, ms, mg,   \
                         mom, n, lr, rho, momentum, eps, S);                                    \
    else                                                                                        \
      hipLaunchKernelGGL((aa_rmsprop_kernel<C_, M_, false>), grid, block, 0, st, p, g, ms, mg,  \
                         mom, n, lr, rho, momentum, eps, S);                                    \
  } while (0)
  if (mg && mom) AA_RMS(true, true);
  else if (mg) AA_RMS(true, false);
  else if (mom) AA_RMS(false, true);
  else AA_RMS(false, false);
#undef AA_RMS
  return aa_launch_status();
}

int aa_rmsprop_step_slabs(float* p, float* g, float* ms, float* mg, float* mom, int64_t n,
                          float lr, float rho, float momentum, float eps,
                          const aa_plane_scatter* planes, const aa_grad_slabs* slabs,
                          void* stream) {
  return aa_rmsprop_step_slabs_pack(p, g, ms, mg, mom, n, lr, rho, momentum, eps, planes, slabs,
                                    nullptr, 0, nullptr, stream);
}

int aa_rmsprop_step_slabs_pack(float* p, float* g, float* ms, float* mg, float* mom, int64_t n,
                               float lr, float rho, float momentum, float eps,
                               const aa_plane_scatter* planes, const aa_grad_slabs* slabs,
                               const float* const* pack_src_h, int32_t pack_n, float* pack_dst,
                               void* stream) {
  AaPack K;
  K.n = 0;
  K.dst = pack_dst;
  for (int j = 0; j < AA_MAX_PACK; ++j) K.src[j] = nullptr;
  if (pack_n > 0) {
    if (pack_n > AA_MAX_PACK || pack_src_h == nullptr || pack_dst == nullptr)
      return AA_ERR_INVALID;
    for (int j = 0; j < pack_n; ++j) {
      if (pack_src_h[j] == nullptr) return AA_ERR_INVALID;
      K.src[j] = pack_src_h[j];
    }
    K.n = pack_n;
  }
  if (slabs == nullptr || slabs->n <= 0) {
    if (K.n > 0) return AA_ERR_INVALID;     // (the packing rides in the slab-summing launch only)
    return aa_rmsprop_step_planes(p, g, ms, mg, mom, n, lr, rho, momentum, eps, planes, stream);
  }
  if (!p || !g || !ms || n <= 0 || slabs->n > AA_MAX_GRAD_SLABS) return AA_ERR_INVALID;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)ms | (uintptr_t)mg | (uintptr_t)mom) & 15) != 0)
    return AA_ERR_INVALID;
  if (aa_planes_check(planes, n) != AA_OK) return AA_ERR_INVALID;
  AaSlabSrc G;
  G.n = slabs->n;
  int blocks = 0;
  int64_t covered = 0, prev_end = 0;
  for (int s = 0; s < slabs->n; ++s) {
    const int64_t off = slabs->offset[s];
    const int mn = slabs->mn[s], nt = slabs->n_tail[s], z = slabs->splits[s];
    if (slabs->slab[s] == nullptr || (((uintptr_t)slabs->slab[s]) & 15) != 0 || mn <= 0 ||
        nt < 0 || (mn & 3) != 0 || (nt & 3) != 0 || (off & 3) != 0 || off < prev_end ||
        off + mn + nt > n)
      return AA_ERR_INVALID;
    // the 16 z-lane association of the reduce launches (gemm.hip / conv_dw_frame_x6.hip: "deep")
    const int64_t work = ((int64_t)mn + nt) / 4;
    if (z < 32 || work > 65536) return AA_ERR_RANGE;
    prev_end = off + mn + nt;
    covered += mn + nt;
    G.slab[s] = slabs->slab[s]; G.splits[s] = z; G.mn[s] = mn; G.n_tail[s] = nt; G.off[s] = off;
    G.first[s] = blocks;
    int b = (int)((work + 15) / 16);
    if (b > 2048) b = 2048;      // (the reduce launch's block count: same grid-stride walk)
    blocks += b;
    G.first[s + 1] = blocks;
  }
  for (int s = slabs->n; s < AA_MAX_GRAD_SLABS; ++s) {
    G.slab[s] = nullptr; G.splits[s] = 0; G.mn[s] = 0; G.n_tail[s] = 0; G.off[s] = 0;
    G.first[s + 1] = blocks;
  }
  const unsigned flat = aa_ew_blocks((n - covered) / 4);
  const dim3 grid((unsigned)blocks + flat), block(AA_EW_THREADS);
  hipStream_t st = (hipStream_t)stream;
  const bool pl = planes != nullptr && planes->n > 0;
  const aa_plane_scatter S = pl ? *planes : aa_plane_scatter{};
#define AA_RMS(C_, M_)                                                                          \
  do {                                                                                          \
    if (pl)                                                                                     \
      hipLaunchKernelGGL((aa_rmsprop_slabs_kernel<C_, M_, true>), grid, block, 0, st, p, g, ms, \
                         mg, mom, n, lr, rho, momentum, eps, S, G, K);                          \
    else                                                                                        \
      hipLaunchKernelGGL((aa_rmsprop_slabs_kernel<C_, M_, false>), grid, block, 0, st, p, g,    \
                         ms, mg, mom, n, lr, rho, momentum, eps, S, G, K);                      \
  } while (0)
  if (mg && mom) AA_RMS(true, true);
  else if (mg) AA_RMS(true, false);
  else if (mom) AA_RMS(false, true);
  else AA_RMS(false, false);
#undef AA_RMS
  return aa_launch_status();
}

int aa_rmsprop_step(float* p, const float* g, float* ms, float* mg, float* mom, int64_t n,
                    float lr, float rho, float momentum, float eps, void* stream) {
  return aa_rmsprop_step_planes(p, g, ms, mg, mom, n, lr, rho, momentum, eps, nullptr, stream);
}

int aa_sgd_step(float* p, const float* g, int64_t n, float lr, void* stream) {
  if (!p || !g || n <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_sgd_kernel, dim3(aa_ew_blocks(n)), dim3(AA_EW_THREADS), 0,
                     (hipStream_t)stream, p, g, n, lr);
  return aa_launch_status();
}

int aa_soft_update(float* target, const float* source, int64_t n, float tau, void* stream) {
  if (!target || !source || n <= 0) return AA_ERR_INVALID;
  if ((((uintptr_t)target | (uintptr_t)source) & 15) != 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_soft_update_kernel, dim3(aa_ew_blocks(n / 4)), dim3(AA_EW_THREADS), 0,
                     (hipStream_t)stream, target, source, n, tau);
  return aa_launch_status();
}

int aa_segment_sumsq(const float* g, const int64_t* seg_offsets_dev, int32_t n_seg,
                     float* sumsq_out, void* stream) {
  if (!g || !seg_offsets_dev || !sumsq_out || n_seg <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_segment_sumsq_kernel, dim3((unsigned)n_seg), dim3(1024), 0,
                     (hipStream_t)stream, g, seg_offsets_dev, sumsq_out);
  return aa_launch_status();
}

int aa_clip_by_norm(float* g, const int64_t* seg_offsets_dev, int32_t n_seg, const float* sumsq,
                    float clip, int32_t per_tensor, void* stream) {
  if (!g || !seg_offsets_dev || !sumsq || n_seg <= 0 || !(clip > 0.f)) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_clip_kernel, dim3(64, (unsigned)n_seg), dim3(AA_EW_THREADS), 0,
                     (hipStream_t)stream, g, seg_offsets_dev, n_seg, sumsq, clip, per_tensor);
  return aa_launch_status();
}

}  // extern "C"
