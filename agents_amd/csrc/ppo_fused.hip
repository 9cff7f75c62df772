// One PPO minibatch train step in THREE launches (tf_agents/agents/ppo/ppo_agent.py:834-1076 for a
// feed-forward actor / value pair; ppo_agent.py:481-615 get_loss; tf_agents/train/ppo_learner.py:
// 220-248 hands it the minibatch):
//
//   K1  aa_ppo_fused_step_kernel    one workgroup per 16 samples: advantage normalisation over the
//                                   whole minibatch (_normalize_advantages :100-110, recomputed by
//                                   every workgroup from the 16 KB of advantages: no launch, no
//                                   cross-workgroup dependency), trajectory mask (ppo_utils.py:35-59),
//                                   old log-prob (common.log_probability), observation normaliser
//                                   (TensorNormalizer.normalize), actor AND value MLP forward, the
//                                   clipped-surrogate / value / entropy loss with the tanh-Normal head
//                                   (:1159-1512), both backward passes -> this workgroup's gradient
//                                   slab + five loss partial sums
//   K2  aa_ppo_fused_reduce_kernel  slabs -> flat gradient (fixed order), per-workgroup sum of
//                                   squares, LossInfo scalars, Adam step counter
//   K3  aa_ppo_fused_apply_kernel   tf.clip_by_global_norm (:948-949) + Adam (TF ApplyAdam)
//
// K2 and K3 run as ONE launch by default (aa_ppo_fused_reduce_kernel<true>, <= 16,384 parameters):
// every K2 workgroup keeps its 64 summed gradients in registers and applies K3 to them once all
// workgroups have published their sum of squares through tagged 8-byte slots -- a grid barrier
// that moves 8 bytes per workgroup and nothing else; 42.3 -> 40.7 us per step, same bits
// (aa_ppo_fused_merge_apply(0) restores the two launches).
//
// against ~28 launches of the layer-by-layer path (mask, log-prob, moments, normaliser, 2 x MLP
// forward, head forward, loss, head backward, column sums, 2 x MLP backward + slab reduces,
// segment sum of squares, clip, counter, Adam, pack): at 4,096 samples x (64, 64) the step is pure
// launch latency (0.21 ms for 0.27 GFLOP).  The minibatch rows are read THROUGH the shuffle's
// row index (`rows`), so no gather launch is needed either (aa_ppo_fused_epoch runs all the
// minibatches of an epoch from one host call).
//
// The 64-wide layers run on the fp32 matrix cores: v_mfma_f32_16x16x4_f32 is bit-for-bit an fmaf
// chain in k order (MI355X_MICROARCH.md), so nothing changes numerically against the VALU kernels
// of mlp_small.hip except the summation order of the weight gradients.  Every layer is padded to
// 64 x 64 in LDS (row pitch 68 floats: the transposed operand reads of the backward pass are then
// bank-conflict free).
//   forward   H_out[16 x 64] = act(H_in[16 x 64] W[64 x 64] + b)     wave w: columns 16w .. 16w+15
//   dW        H_in^T[64 x 16] G[16 x 64]                              wave w: rows 16w .. 16w+15
//   dX        G[16 x 64] W^T[64 x 64] * act'(H_in)                    wave w: columns 16w .. 16w+15
// Fragment maps of the 16x16x4 form: A lane l = A[l & 15][l >> 4], B lane l = B[l >> 4][l & 15],
// D lane l, register r = D[4 (l >> 4) + r][l & 15].
//
// What a 16-sample workgroup is bound by is LATENCY: 12 dependent layer steps with one wave per
// SIMD and nothing to hide a load, an LDS round trip or a dependent MFMA behind.  Measured with
// in-kernel wall_clock64 stamps (tools/ppo_fused_probe.py), per launch: first version 70 us
// (3-6 us per layer step: rolled k loops, __syncthreads draining the slab stores and the weight
// loads, serial per-sample loops over the action dims, a 440-byte scratch copy of the argument
// struct) -> 41 us (LDS-only barriers, weights prefetched a step ahead in registers, fully
// unrolled k loops with two accumulators, (sample, dim)-parallel loss phases) -> this version:
// the ACTOR and the VALUE network are walked at the same time by two groups of four waves (they
// are independent between the shared input and the loss), which halves the dependent chain.
//
// Built with -ffp-contract=off: the loss arithmetic is op for op the one of ppo.hip.
#include "common.h"
#include "agents_amd.h"

#define PF_TS 16
#define PF_W 64
#define PF_PITCH 68
#define PF_MAXL 3        /* layers per network on this path */
#define PF_THREADS 512   /* 2 groups (actor, value) x 4 waves */
#define PF_HALF_LOG_2PI 0.91893853320467274178f
#define PF_MD AA_PPO_FUSED_MAX_D
#define PF_MAX_EPOCH_STEPS 4096   /* minibatches per host call whose moments are computed ahead */

typedef float pf_f32x4 __attribute__((ext_vector_type(4)));

struct PfArgs {
  aa_ppo_fused_desc d;
  float* slabs;        // [n_wg][total]
  float* partial;      // [n_wg][8]
  const float* moments;  // nullable: {mean, var} of this step's advantages, computed per epoch
  long long* stamps;   // nullable (aa_ppo_fused_debug_stamps): [n_wg][32] wall_clock64 ticks (10 ns)
  unsigned long long* seq;   // launch sequence of this workspace: += 1 per K1 launch (workgroup 0);
                             // the tag of the merged K2's barrier slots
};

static long long* g_pf_stamps = nullptr;
#define PF_STAMP(i)                                                              \
  if (P.stamps != nullptr && threadIdx.x == 0)                                   \
    P.stamps[(size_t)blockIdx.x * 32 + (i)] = wall_clock64();

// Dynamic LDS image of a workgroup (one per CU: ~156 KB of gfx950's 160 KB -- the static_assert
// behind the struct is the guard; a new field or a larger PF_MD must fit under it).
struct PfLds {
  // Round 5: EVERY layer's weights of both groups are resident (104 KB; 160 KB in all): they are
  // requested at kernel start, land during the prologue and are committed once -- a layer step is
  // then LDS reads + MFMAs behind ONE barrier.  (Until round 4 a group held one layer at a time:
  // every step began with barrier / commit 17 KB / issue the next layer's loads / barrier, which the
  // in-kernel timeline put at 1.5-3 us of each 1.7-4.2 us step.)
  float Ws[2][PF_MAXL][PF_W * PF_PITCH];
  float bs[2][PF_MAXL][PF_W];
  float X[PF_TS][PF_PITCH];                  // normalised observations (input of both networks)
  float H[2][PF_MAXL][PF_TS][PF_PITCH];      // layer outputs: [0] actor, [1] value
  float G[2][2][PF_TS][PF_PITCH];            // backward ping-pong per group
  aa_mlp_layout nets[2];
  float red[16];
  float bc[2];
  float w[PF_TS], advn[PF_TS], oldlp[PF_TS], dlp[PF_TS], dent[PF_TS];
  long long rowi[PF_TS];                     // source row of every sample of the tile
  float act[PF_TS][PF_MD], t0[PF_TS][PF_MD], t1[PF_TS][PF_MD], t2[PF_TS][PF_MD];
  float dbias[PF_TS][PF_MD];
  float scale[PF_MD], logs[PF_MD], dsp[PF_MD];
};
static_assert(sizeof(PfLds) <= 160 * 1024, "PfLds must fit the 160 KB of LDS a gfx950 CU has");

__device__ static inline float pf_softplus(float x) {
  return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x));
}
__device__ static inline float pf_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}
__device__ static inline float pf_actgrad(float y, int act) {
  if (act == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

// Workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() carries a release
// fence that also drains the VM counter, i.e. the gradient-slab stores of the backward steps
// (never read by this launch) and the weight prefetch of the next step.
__device__ static inline void pf_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Weights of one layer travel HBM/L2 -> registers -> LDS in two halves, so that the loads of a
// group's NEXT layer step are in flight while the current one computes.  A 64 x 64 matrix is 1024
// float4 = 4 per thread of a 256-thread group; matrices whose rows are not float4-sized (the 6-
// and 1-wide heads: <= 1024 elements) go element-wise.  Only real elements are written to LDS:
// the tile is zero-filled ONCE at kernel start, afterwards it only ever holds finite values, and
// every read beyond a layer's true shape is multiplied by an exact zero or discarded by a guard.
struct PfW {
  float4 v[4];
  float bias;
};

__device__ static inline void pf_prefetch(const float* __restrict__ params,
                                          const aa_mlp_layout& net, int l, int gt, PfW& w) {
  const int n_in = net.dims[l], n_out = net.dims[l + 1];
  const float* W = params + net.k_off[l];
  if ((n_out & 3) == 0 && (net.k_off[l] & 3) == 0) {
    const int total4 = n_in * (n_out >> 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = gt + 256 * u;
      w.v[u] = i < total4 ? reinterpret_cast<const float4*>(W)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    const int total = n_in * n_out;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = gt + 256 * u;
      w.v[u].x = i < total ? W[i] : 0.f;
    }
  }
  w.bias = gt < n_out ? params[net.b_off[l] + gt] : 0.f;
}

__device__ static inline void pf_commit(const aa_mlp_layout& net, int l, int gt, const PfW& w,
                                        float* Ws, float* bs) {
  const int n_in = net.dims[l], n_out = net.dims[l + 1];
  if ((n_out & 3) == 0 && (net.k_off[l] & 3) == 0) {
    const int q4 = n_out >> 2, total4 = n_in * q4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = gt + 256 * u;
      if (i < total4) {
        const int k = i / q4, jq = i - k * q4;
        *reinterpret_cast<float4*>(Ws + k * PF_PITCH + 4 * jq) = w.v[u];
      }
    }
  } else {
    const int total = n_in * n_out;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = gt + 256 * u;
      if (i < total) {
        const int k = i / n_out, j = i - k * n_out;
        Ws[k * PF_PITCH + j] = w.v[u].x;
      }
    }
  }
  if (gt < PF_W) bs[gt] = w.bias;
}

// The layer a group works on in step i of its 2 L steps (forward 0 .. L-1, backward L-1 .. 0), or
// -1 when the group has no layer left in this phase (the other network is deeper).
__device__ static inline int pf_fwd_layer(int L, int i) { return i < L ? i : -1; }
__device__ static inline int pf_bwd_layer(int L, int i) { return i < L ? L - 1 - i : -1; }

// H_out = act(H_in W + b); every one of the 64 columns of H_out is written (zeros beyond n_out).
// The k loop always runs the padded 16 steps with every LDS read issued up front and two
// accumulators; rows / columns beyond the layer's shape multiply exact zeros.
__device__ static inline void pf_forward(const aa_mlp_layout& net, int l, int gt,
                                         const float (*Hin)[PF_PITCH], float (*Hout)[PF_PITCH],
                                         const float* Ws, const float* bs) {
  const int n_out = net.dims[l + 1];
  const int wave = gt >> 6, lane = gt & 63, lr = lane & 15, lg = lane >> 4;
  const int col = 16 * wave + lr;
  pf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (16 * wave < n_out) {          // wave-uniform
    float a[16], b[16];
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      a[kt] = Hin[lr][4 * kt + lg];
      b[kt] = Ws[(4 * kt + lg) * PF_PITCH + col];
    }
#pragma unroll
    for (int kt = 0; kt < 16; kt += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt], b[kt], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt + 1], b[kt + 1], acc1, 0, 0, 0);
    }
  }
  const float bias = bs[col];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    Hout[4 * lg + r][col] =
        col < n_out ? pf_act((acc0[r] + acc1[r]) + bias, net.acts[l]) : 0.f;
}

// One layer of the backward pass.  G = d loss / d (pre-activation of layer l), all 64 columns
// defined.  Writes dW / db of the layer into this workgroup's slab and, for l > 0,
// Gnext = (G W^T) * act'_{l-1}(H_in) = d loss / d (pre-activation of layer l - 1).
__device__ static inline void pf_backward(const aa_mlp_layout& net, int l, int gt,
                                          const float (*Hin)[PF_PITCH], const float (*G)[PF_PITCH],
                                          float (*Gnext)[PF_PITCH], const float* Ws,
                                          float* __restrict__ slab) {
  const int n_in = net.dims[l], n_out = net.dims[l + 1];
  const int wave = gt >> 6, lane = gt & 63, lr = lane & 15, lg = lane >> 4;
  // ---- dW[k][j] = sum_s H_in[s][k] G[s][j]: wave -> rows k = 16 wave ..; column tiles ct ----
  if (16 * wave < n_in) {
    float a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = Hin[4 * t + lg][16 * wave + lr];   // A[i = k local][kk = sample]
    const int nct = (n_out + 15) >> 4;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (ct < nct) {               // wave-uniform
        float b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = G[4 * t + lg][16 * ct + lr];   // B[kk = sample][j local]
        pf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
        const int j = 16 * ct + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * wave + 4 * lg + r;
          if (k < n_in && j < n_out) slab[net.k_off[l] + (int64_t)k * n_out + j] = acc[r];
        }
      }
    }
  }
  // ---- db[j] = sum_s G[s][j] (sample order) ------------------------------------------------------
  if (gt < n_out) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < PF_TS; ++s) sum += G[s][gt];
    slab[net.b_off[l] + gt] = sum;
  }
  // ---- Gnext[s][k] = (sum_j G[s][j] W[k][j]) act'(H_in[s][k]) -------------------------------------
  if (l > 0) {
    const int k = 16 * wave + lr;
    pf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (16 * wave < n_in) {
      float a[16], b[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        a[t] = G[lr][4 * t + lg];                        // A[i = sample][kk = j]
        b[t] = Ws[k * PF_PITCH + 4 * t + lg];            // B[kk = j][n = k] = W[k][j]
      }
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t + 1], b[t + 1], acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = 4 * lg + r;
      Gnext[s][k] =
          k < n_in ? (acc0[r] + acc1[r]) * pf_actgrad(Hin[s][k], net.acts[l - 1]) : 0.f;
    }
  }
}

// mean / variance of the N advantages of one minibatch (tf.nn.moments, two-pass, a fixed order for
// PF_THREADS threads): every workgroup of a step computes them itself, or -- when a whole epoch is
// issued from one host call -- one launch computes them for ALL of the epoch's minibatches ahead
// of time (they depend on the shuffle, not on the weights); the same code, the same bits.
// bc[0] = mean, bc[1] = variance; ends with a barrier.
__device__ static inline void pf_moments(const float* __restrict__ adv,
                                         const int64_t* __restrict__ rows, int64_t N, float* red,
                                         float* bc) {
  const int tid = threadIdx.x;
  const bool keep = N <= 16 * PF_THREADS;      // the 16 values of a thread stay in registers
  float av[16];
  float s = 0.f;
  if (keep) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t i = tid + (int64_t)PF_THREADS * u;
      av[u] = i < N ? adv[rows != nullptr ? rows[i] : i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += av[u];
  } else {
    for (int64_t i = tid; i < N; i += PF_THREADS) s += adv[rows != nullptr ? rows[i] : i];
  }
  float t = aa_block_sum(s, red);
  if (tid == 0) bc[0] = t / (float)N;
  __syncthreads();
  const float mean = bc[0];
  float q = 0.f;
  if (keep) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (tid + (int64_t)PF_THREADS * u < N) {
        const float dl = av[u] - mean;
        q += dl * dl;
      }
    }
  } else {
    for (int64_t i = tid; i < N; i += PF_THREADS) {
      const float dl = adv[rows != nullptr ? rows[i] : i] - mean;
      q += dl * dl;
    }
  }
  t = aa_block_sum(q, red);
  if (tid == 0) bc[1] = t / (float)N;
  __syncthreads();
}

// workgroup s: the moments of minibatch s of an epoch -> out[2 s], out[2 s + 1]
__global__ void __launch_bounds__(PF_THREADS)
aa_ppo_fused_moments_kernel(const float* __restrict__ adv, const int64_t* __restrict__ rows,
                            int64_t N, float* __restrict__ out) {
  __shared__ float red[16];
  __shared__ float bc[2];
  pf_moments(adv, rows + (int64_t)blockIdx.x * N, N, red, bc);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = bc[0];
    out[2 * blockIdx.x + 1] = bc[1];
  }
}

__global__ void __launch_bounds__(PF_THREADS) aa_ppo_fused_step_kernel(PfArgs P) {
  const aa_ppo_fused_desc& d = P.d;
  extern __shared__ __attribute__((aligned(16))) char pf_lds_raw[];
  PfLds& S = *reinterpret_cast<PfLds*>(pf_lds_raw);
  const int tid = threadIdx.x;
  const int grp = tid >> 8, gt = tid & 255;        // group 0: actor, group 1: value
  const int64_t N = d.N, b0 = (int64_t)blockIdx.x * PF_TS;
  const int D = d.D;
  float* slab = P.slabs + (int64_t)blockIdx.x * d.total;
  // source row of minibatch sample b (the shuffle's permutation slice), or b itself
  // The tile's 16 source rows are requested FIRST and handed round through LDS (rowi): every
  // staging load below used to start with its own dependent read of the permutation slice -- two
  // memory round trips in a row (2.8 us of "scalars + obs tile" on the in-kernel timeline).
  if (tid < PF_TS) {
    const int64_t b = b0 + tid;
    S.rowi[tid] = b < N ? (d.rows != nullptr ? d.rows[b] : b) : 0;
  }
  auto row_of = [&](int64_t b) -> int64_t { return S.rowi[b - b0]; };
  PfW wq[PF_MAXL];
  PF_STAMP(0)
  if (tid == 0 && blockIdx.x == 0) *P.seq += 1;
  if (tid == 0) S.nets[0] = d.actor;
  if (tid == 256) S.nets[1] = d.value;
  // every layer of this group's network is in flight during the prologue below
  {
    const aa_mlp_layout& mine = grp == 0 ? d.actor : d.value;
#pragma unroll
    for (int l = 0; l < PF_MAXL; ++l)
      pf_prefetch(d.params, mine, l < mine.n_layers ? l : mine.n_layers - 1, gt, wq[l]);
  }
  {
    float4* z = reinterpret_cast<float4*>(&S.Ws[0][0][0]);
    constexpr int NZ = 2 * PF_MAXL * PF_W * PF_PITCH / 4;
    for (int i = tid; i < NZ; i += PF_THREADS) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < 4 * PF_TS * PF_PITCH; i += PF_THREADS) (&S.G[0][0][0][0])[i] = 0.f;

  // ---- advantage moments over the WHOLE minibatch ----------------------------------------------------
  if (P.moments != nullptr) {
    if (tid == 0) { S.bc[0] = P.moments[0]; S.bc[1] = P.moments[1]; }
    __syncthreads();
  } else {
    pf_moments(d.adv, d.rows, N, S.red, S.bc);
  }
  PF_STAMP(1)
  // ---- the tile's rows: actions, old log-prob terms (thread per (sample, dim)), per-sample
  //      scalars, head constants, the normalised observation tile ----------------------------------
  if (tid < PF_TS * D) {
    const int ss = tid / D, e = tid - ss * D;
    const int64_t b = b0 + ss;
    float act = 0.f, term = 0.f;
    if (b < N) {
      const int64_t r = row_of(b);
      act = d.actions[r * D + e];
      const float sc = d.old_scale[r * D + e];
      const float diff = act / sc - d.old_loc[r * D + e] / sc;
      term = -0.5f * (diff * diff) - (PF_HALF_LOG_2PI + logf(sc));
    }
    S.act[ss][e] = act;
    S.t0[ss][e] = term;
  }
  if (tid >= 256 && tid < 256 + PF_TS) {
    const int ss = tid - 256;
    const int64_t b = b0 + ss;
    float w = 0.f, advn = 0.f;
    if (b < N) {
      const int64_t r = row_of(b);
      const float inv = 1.0f / sqrtf(S.bc[1] + d.adv_eps);
      const float a = d.adv[r];
      advn = a * inv + (-S.bc[0] * inv);          // tf.nn.batch_normalization(adv, mean, var)
      const bool valid = (d.step_type[r] != 2) && !(d.returns[r] == 0.f && a == 0.f);
      const float m = valid ? 1.0f : 0.0f;
      w = d.weights != nullptr ? d.weights[r] * m : m;
    }
    S.w[ss] = w;
    S.advn[ss] = advn;
  }
  if (tid >= 320 && tid < 320 + D) {
    const int e = tid - 320;
    const float bsd = d.params[d.head_off + e];
    const float sc = pf_softplus(bsd);
    S.scale[e] = sc;
    S.logs[e] = logf(sc);
    S.dsp[e] = 1.0f / (1.0f + expf(-bsd));       // d softplus / d bias = sigmoid
  }
#pragma unroll
  for (int u = 0; u < PF_TS * PF_W / PF_THREADS; ++u) {
    const int i = tid + PF_THREADS * u;
    const int ss = i >> 6, k = i & 63;
    const int64_t bb = b0 + ss;
    float v = 0.f;
    if (bb < N && k < d.obs_dim) {
      v = d.obs[row_of(bb) * d.ld_obs + k];
      if (d.nrm_avg != nullptr) {
        const float var = d.nrm_m2[k] / d.nrm_count[k];
        const float inv = 1.0f / sqrtf(var + d.nrm_eps);
        v = v * inv + (-d.nrm_avg[k] * inv);
        if (d.nrm_clip > 0.f) v = fminf(fmaxf(v, -d.nrm_clip), d.nrm_clip);
      }
    }
    S.X[ss][k] = v;
  }
  pf_barrier();
  if (tid < PF_TS) {          // old log-prob: the per-dim terms added in dim order
    float olp = 0.f;
    for (int e = 0; e < D; ++e) olp += S.t0[tid][e];
    S.oldlp[tid] = olp;
  }
  PF_STAMP(2)
  // ---- forward: the actor (group 0) and the value network (group 1) side by side --------------------
  const aa_mlp_layout& net = S.nets[grp];
  const int Lg = net.n_layers;
  const int Lmax = S.nets[0].n_layers > S.nets[1].n_layers ? S.nets[0].n_layers
                                                           : S.nets[1].n_layers;
  // the weights of all layers -> LDS (the zero fill above is complete: the barrier before stamp 2)
#pragma unroll
  for (int l = 0; l < PF_MAXL; ++l)
    if (l < Lg) pf_commit(net, l, gt, wq[l], S.Ws[grp][l], S.bs[grp][l]);
  for (int i = 0; i < Lmax; ++i) {
    const int l = pf_fwd_layer(Lg, i);
    pf_barrier();      // the previous step's outputs (and, for i = 0, the committed weights)
    if (l >= 0)
      pf_forward(net, l, gt, l == 0 ? S.X : S.H[grp][l - 1], S.H[grp][l], S.Ws[grp][l],
                 S.bs[grp][l]);
    PF_STAMP(3 + i)
  }
  pf_barrier();
  PF_STAMP(9)
  // ---- loss (the arithmetic of ppo.hip: aa_ppo_loss_kernel), three short phases --------------------
  const int La = S.nets[0].n_layers, Lv = S.nets[1].n_layers;
  // A: thread per (sample, dim): log-prob / entropy terms of the current policy
  float r_th = 0.f, r_diff = 0.f;      // kept for phase C by the (sample, dim) thread
  if (tid < PF_TS * D) {
    const int ss = tid / D, e = tid - ss * D;
    const float zz = S.H[0][La - 1][ss][e];
    r_th = d.act_mag != nullptr ? tanhf(zz) : zz;
    const float loc = d.act_mag != nullptr ? d.act_mean[e] + d.act_mag[e] * r_th : zz;
    const float sc = S.scale[e];
    const float xs = S.act[ss][e] / sc, ls = loc / sc;
    const float df = xs - ls;
    S.t1[ss][e] = -0.5f * (df * df) - (PF_HALF_LOG_2PI + S.logs[e]);
    S.t2[ss][e] = 0.5f + PF_HALF_LOG_2PI + S.logs[e];
    r_diff = S.act[ss][e] - loc;
  }
  pf_barrier();
  // B: thread per sample: surrogate, value loss, entropy; d loss / d log-prob, d loss / d entropy
  float sum_pg = 0.f, sum_v = 0.f, sum_ent = 0.f, sum_clip = 0.f, sum_entw = 0.f;
  if (tid < PF_TS) {
    const int64_t b = b0 + tid;
    float dv = 0.f, dlp = 0.f, dent = 0.f;
    if (b < N) {
      const int64_t r = row_of(b);
      const float w = S.w[tid];
      float lp = 0.f, ent = 0.f;
      for (int e = 0; e < D; ++e) {
        lp += S.t1[tid][e];
        ent += S.t2[tid][e];
      }
      float lp_c = lp;
      bool lp_live = true;
      if (d.logp_clip > 0.f) {
        lp_c = fminf(fmaxf(lp, -d.logp_clip), d.logp_clip);
        lp_live = (lp >= -d.logp_clip) && (lp <= d.logp_clip);
      }
      const float a = S.advn[tid];
      const float ratio = expf(lp_c - S.oldlp[tid]);
      const float ratio_c = fminf(fmaxf(ratio, 1.0f - d.clip_eps), 1.0f + d.clip_eps);
      const float obj = ratio * a, obj_c = ratio_c * a;
      float pg;
      bool grad_through_ratio;
      if (d.clip_eps > 0.f) {
        pg = -fminf(obj, obj_c);
        grad_through_ratio = obj <= obj_c;   // tf.minimum routes the gradient to x when x <= y
        sum_clip = fabsf(ratio - 1.0f) > d.clip_eps ? 1.0f : 0.0f;
      } else {
        pg = -obj;
        grad_through_ratio = true;
      }
      sum_pg = (w == 0.f) ? 0.f : pg * w;
      if (grad_through_ratio && lp_live) dlp = -(a * ratio) * w / d.denom;
      const float R = d.returns[r], V = S.H[1][Lv - 1][tid][0];
      float verr = (R - V) * (R - V);
      float dverr_dV = -2.0f * (R - V);
      if (d.value_clip > 0.f && d.old_vpred != nullptr) {
        const float ov = d.old_vpred[r];
        const float dlt = V - ov;
        const float dc = fminf(fmaxf(dlt, -d.value_clip), d.value_clip);
        const float Vc = ov + dc;
        const float verr_c = (R - Vc) * (R - Vc);
        if (verr_c > verr) {     // tf.maximum: gradient to x when x >= y
          verr = verr_c;
          const bool live = dlt >= -d.value_clip && dlt <= d.value_clip;
          dverr_dV = live ? -2.0f * (R - Vc) : 0.f;
        }
      }
      sum_v = (w == 0.f) ? 0.f : verr * w;
      dv = d.c_v * dverr_dV * w / d.denom;
      sum_ent = (w == 0.f) ? 0.f : (-ent) * w;
      sum_entw = ent * w;
      dent = (d.c_e > 0.f) ? (-d.c_e * w / d.denom) : 0.f;
    }
    S.G[1][0][tid][0] = dv;        // d loss / d value: column 0 of the value group's gradient tile
    S.dlp[tid] = dlp;
    S.dent[tid] = dent;
  }
  // the tile's five loss sums: 16 lanes of wave 0, xor tree (fixed order)
  if (tid < 64) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      sum_pg += __shfl_xor(sum_pg, off, 64);
      sum_v += __shfl_xor(sum_v, off, 64);
      sum_ent += __shfl_xor(sum_ent, off, 64);
      sum_clip += __shfl_xor(sum_clip, off, 64);
      sum_entw += __shfl_xor(sum_entw, off, 64);
    }
    if (tid == 0) {
      float* p = P.partial + (int64_t)blockIdx.x * 8;
      p[0] = sum_pg; p[1] = sum_v; p[2] = sum_ent; p[3] = sum_clip; p[4] = sum_entw;
    }
  }
  pf_barrier();
  // C: thread per (sample, dim): back through the tanh-Normal head
  if (tid < PF_TS * D) {
    const int ss = tid / D, e = tid - ss * D;
    float gz = 0.f, gb = 0.f;
    if (b0 + ss < N) {
      const float sc = S.scale[e];
      const float dlp = S.dlp[ss], dent = S.dent[ss];
      const float dlp_dloc = r_diff / (sc * sc);
      const float dlp_dsc = (r_diff * r_diff) / (sc * sc * sc) - 1.0f / sc;
      float dloc_dz = 1.0f;
      if (d.act_mag != nullptr) dloc_dz = d.act_mag[e] * (1.0f - r_th * r_th);
      gz = dlp * dlp_dloc * dloc_dz;
      gb = (dlp * dlp_dsc + dent * (1.0f / sc)) * S.dsp[e];
    }
    S.G[0][0][ss][e] = gz;
    S.dbias[ss][e] = gb;
  }
  pf_barrier();
  // std_bias gradient = column sums of the per-sample terms (sample order)
  if (tid < D) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < PF_TS; ++s) sum += S.dbias[s][tid];
    slab[d.head_off + tid] = sum;
  }
  PF_STAMP(10)
  // ---- backward: both groups, top layer first ------------------------------------------------------------
  int cur = 0;
  for (int i = 0; i < Lmax; ++i) {
    const int l = pf_bwd_layer(Lg, i);
    pf_barrier();                  // completes G[grp][cur]
    if (l >= 0) {
      pf_backward(net, l, gt, l == 0 ? S.X : S.H[grp][l - 1], S.G[grp][cur], S.G[grp][cur ^ 1],
                  S.Ws[grp][l], slab);
      cur ^= 1;
    }
    PF_STAMP(11 + i)
  }
  PF_STAMP(17)
}

// K2: grads[i] = sum over slabs (16 z-lanes per parameter, four loads in flight, lanes combined in
// order: the association is fixed for a given slab count), per-workgroup sum of squares for the
// global norm; workgroup 0 also turns the loss partials into the LossInfo vector and advances the
// optimizer's step counter.  Alignment padding of the flat layout is never written by K1: the
// slabs are zero-filled ONCE when the workspace is allocated, so padding gradients read as zero.
//
// MERGED = true is K2 + K3 in one launch (n_red <= PF_MERGE_MAX_WG workgroups, all co-resident):
// each workgroup keeps its 64 summed gradients in registers, publishes its sum of squares, waits
// at a grid barrier until every workgroup has done so, re-adds the partials in K3's order and
// applies K3's arithmetic to its own 64 parameters.  ONLY the sumsq partials and the barrier
// ticket cross workgroups (agent-scope atomics); every other word is read and written by the
// workgroup that owns it, so the results are those of the three-launch path bit for bit.
#define PF_MERGE_MAX_WG 256u
static int32_t g_pf_merge_apply = 1;

struct PfApply {
  float* p; float* m; float* v;
  float lr, beta1, beta2, eps, clip;
  float* sumsq_out;
  // grid barrier: workgroup b publishes {tag = low word of *seq, its sum of squares} as ONE 8-byte
  // word in slot[b]; the tag (bumped by K1, a launch earlier) never repeats in consecutive
  // launches, so a slot that carries it was written by this launch
  unsigned long long* slot;
  const unsigned long long* seq;
};

template <bool MERGED>
__global__ void __launch_bounds__(256)
aa_ppo_fused_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int64_t total,
                           float* __restrict__ grads, float* __restrict__ sumsq_part,
                           const float* __restrict__ partial, aa_ppo_fused_desc d,
                           float* __restrict__ stats, int64_t* __restrict__ step_dev,
                           PfApply ap) {
  // workgroup = 64 consecutive parameters (16 float4) x 16 z-lanes: a 16-lane group reads 256
  // contiguous bytes of one slab (the first version read 4-byte columns 44 KB apart: 30 us for
  // 11 MB); z-lane zl sums slabs zl, zl + 16, ... in that order, the 16 partials are combined
  // in lane order -- a fixed association for a given slab count
  __shared__ float4 part[16][16];
  __shared__ float sq[16];
  __shared__ float tot5[5];
  const int q = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 64 + 4 * q;      // total is a multiple of 4
  const int64_t total4 = total >> 2;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total) {
    const float4* s4 = reinterpret_cast<const float4*>(slabs) + (i >> 2);
    int z = zl;
    for (; z + 48 < n_slabs; z += 64) {
      const float4 t0 = s4[(int64_t)z * total4], t1 = s4[(int64_t)(z + 16) * total4];
      const float4 t2 = s4[(int64_t)(z + 32) * total4], t3 = s4[(int64_t)(z + 48) * total4];
      v.x += t0.x; v.y += t0.y; v.z += t0.z; v.w += t0.w;
      v.x += t1.x; v.y += t1.y; v.z += t1.z; v.w += t1.w;
      v.x += t2.x; v.y += t2.y; v.z += t2.z; v.w += t2.w;
      v.x += t3.x; v.y += t3.y; v.z += t3.z; v.w += t3.w;
    }
    for (; z < n_slabs; z += 16) {
      const float4 t0 = s4[(int64_t)z * total4];
      v.x += t0.x; v.y += t0.y; v.z += t0.z; v.w += t0.w;
    }
  }
  part[zl][q] = v;
  __syncthreads();
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (zl == 0) {
    r = part[0][q];
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 t = part[j][q];
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    if (i < total) {
      if (!MERGED) reinterpret_cast<float4*>(grads)[i >> 2] = r;   // MERGED stores the clipped one
    } else {
      r = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    sq[q] = ((r.x * r.x + r.y * r.y) + r.z * r.z) + r.w * r.w;
  }
  __syncthreads();
  float t_adam = 0.f;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += sq[j];
    if (MERGED) {
      // the optimizer step this launch applies: workgroup 0 stores it only AFTER the barrier, so
      // every workgroup reads the old value here, before it publishes its slot
      t_adam = (float)(*step_dev + 1);
      const unsigned long long word =
          ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(*ap.seq);
      __hip_atomic_store(&ap.slot[blockIdx.x], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      sumsq_part[blockIdx.x] = s;
    }
  }
  if (blockIdx.x == 0) {
    // the five loss sums over the n_slabs workgroups of K1: thread t adds workgroups t, t + 256, ...
    // and the 256 partials are combined by the fixed block tree (a single thread walking 256
    // dependent loads made this workgroup -- and with it the launch -- take 30 us)
    __shared__ float red5[16];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float s = 0.f;
      for (int p = threadIdx.x; p < n_slabs; p += blockDim.x) s += partial[(int64_t)p * 8 + k];
      const float t = aa_block_sum(s, red5);
      if (threadIdx.x == 0) tot5[k] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float n_elems = (float)d.N;
      const float pg = tot5[0] / d.denom;
      const float vl = (tot5[1] / d.denom) * d.c_v;
      const float e = d.c_e > 0.f ? (tot5[2] / d.denom) * d.c_e : 0.f;
      stats[0] = pg;                      // policy_gradient_loss
      stats[1] = vl;                      // value_estimation_loss
      stats[2] = e;                       // entropy_regularization_loss
      stats[3] = tot5[3] / n_elems;       // clip_fraction
      stats[4] = tot5[4] / n_elems;       // mean(entropy * weights)
      stats[5] = 0.f;                     // kl_penalty_loss (no KL terms on this path)
      stats[6] = pg + vl + e;             // total
      stats[7] = 0.f;
      stats[8] = 0.f;                     // l2_regularization_loss
      if (!MERGED && step_dev != nullptr) *step_dev += 1;
    }
  }
  if (!MERGED) return;
  // ---- grid barrier, then K3's clip + Adam on this workgroup's own 64 parameters ----
  __shared__ float s_bc[2];
  __shared__ float red[16];
  // thread k waits for slot k (gridDim.x <= 256 = blockDim.x): value and tag arrive in one word,
  // no other memory is exchanged, so no fence is needed.  K3 has thread k add partials k, k + 256,
  // ... from zero: one partial per thread here, the same sum.
  // this workgroup's own parameters and Adam slots, and the step-dependent factor, are fetched
  // BEFORE the wait: behind the barrier only the clip scale is new
  const int64_t i4 = i >> 2;
  float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = p4, v4 = p4;
  if (zl == 0 && i < total) {
    p4 = reinterpret_cast<const float4*>(ap.p)[i4];
    m4 = reinterpret_cast<const float4*>(ap.m)[i4];
    v4 = reinterpret_cast<const float4*>(ap.v)[i4];
  }
  if (threadIdx.x == 0) {
    const float b1p = powf(ap.beta1, t_adam), b2p = powf(ap.beta2, t_adam);
    s_bc[1] = ap.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
  }
  float s = 0.f;
  if (threadIdx.x < gridDim.x) {
    const unsigned tag = (unsigned)(*ap.seq);
    unsigned spins = 0;
    unsigned long long word;
    while ((unsigned)(word = __hip_atomic_load(&ap.slot[threadIdx.x], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) != tag) {
      // a workgroup that never publishes means the launch was not co-resident (the host refuses
      // such grids): abort loudly instead of hanging the queue
      if (++spins > (1u << 24)) __builtin_trap();
    }
    s += __uint_as_float((unsigned)(word >> 32));
  }
  const float tot = aa_block_sum(s, red);
  if (threadIdx.x == 0) {
    float scale = 1.0f;
    if (ap.clip > 0.f) {
      const float gn = sqrtf(tot);
      scale = ap.clip * fminf(1.0f / gn, 1.0f / ap.clip);
    }
    s_bc[0] = scale;
    if (blockIdx.x == 0) {
      if (ap.sumsq_out != nullptr) ap.sumsq_out[0] = tot;
      *step_dev += 1;
    }
  }
  __syncthreads();
  if (zl == 0 && i < total) {
    const float scale = s_bc[0], alpha = s_bc[1];
    const float omb1 = 1.0f - ap.beta1, omb2 = 1.0f - ap.beta2;
    float gg[4] = {r.x, r.y, r.z, r.w};
    float pp[4] = {p4.x, p4.y, p4.z, p4.w};
    float mm[4] = {m4.x, m4.y, m4.z, m4.w};
    float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gi = gg[c] * scale;
      gg[c] = gi;
      float mi = mm[c], vi = vv[c];
      mi = mi + (gi - mi) * omb1;
      vi = vi + (gi * gi - vi) * omb2;
      pp[c] = pp[c] - (mi * alpha) / (sqrtf(vi) + ap.eps);
      mm[c] = mi;
      vv[c] = vi;
    }
    reinterpret_cast<float4*>(grads)[i4] = make_float4(gg[0], gg[1], gg[2], gg[3]);
    reinterpret_cast<float4*>(ap.p)[i4] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(ap.m)[i4] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(ap.v)[i4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
  }
}

// K3: tf.clip_by_global_norm over the whole flat gradient, then Adam (the arithmetic of optim.hip).
__global__ void __launch_bounds__(256)
aa_ppo_fused_apply_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                          float* __restrict__ v, int64_t n, float lr, float beta1, float beta2,
                          float eps, const int64_t* __restrict__ step_dev,
                          const float* __restrict__ sumsq_part, int n_part, float clip,
                          float* __restrict__ sumsq_out) {
  __shared__ float red[16];
  __shared__ float s_bc[2];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) s += sumsq_part[i];
  const float tot = aa_block_sum(s, red);
  if (threadIdx.x == 0) {
    float scale = 1.0f;
    if (clip > 0.f) {
      const float gn = sqrtf(tot);
      scale = clip * fminf(1.0f / gn, 1.0f / clip);
    }
    const float t = (float)(*step_dev);
    const float b1p = powf(beta1, t), b2p = powf(beta2, t);
    s_bc[0] = scale;
    s_bc[1] = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    if (blockIdx.x == 0 && sumsq_out != nullptr) sumsq_out[0] = tot;
  }
  __syncthreads();
  const float scale = s_bc[0], alpha = s_bc[1], omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * scale;
    g[i] = gi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * omb1;
    vi = vi + (gi * gi - vi) * omb2;
    p[i] = p[i] - (mi * alpha) / (sqrtf(vi) + eps);
    m[i] = mi;
    v[i] = vi;
  }
}

static int pf_check_net(const aa_mlp_layout& net, int in_dim, int out_dim, int64_t total) {
  if (net.n_layers < 1 || net.n_layers > PF_MAXL) return AA_ERR_RANGE;
  if (net.dims[0] != in_dim || net.dims[net.n_layers] != out_dim) return AA_ERR_INVALID;
  for (int l = 0; l <= net.n_layers; ++l)
    if (net.dims[l] < 1 || net.dims[l] > PF_W) return AA_ERR_RANGE;
  for (int l = 0; l < net.n_layers; ++l) {
    if (net.acts[l] < AA_ACT_NONE || net.acts[l] > AA_ACT_TANH) return AA_ERR_INVALID;
    if (net.k_off[l] < 0 || net.k_off[l] + (int64_t)net.dims[l] * net.dims[l + 1] > total ||
        net.b_off[l] < 0 || net.b_off[l] + net.dims[l + 1] > total)
      return AA_ERR_RANGE;
  }
  if (net.acts[net.n_layers - 1] != AA_ACT_NONE) return AA_ERR_INVALID;
  // staging moves 4 float4 (rows of float4-sized width) or 4 elements per thread
  for (int l = 0; l < net.n_layers; ++l) {
    const int n_in = net.dims[l], n_out = net.dims[l + 1];
    const bool vec = (n_out & 3) == 0 && (net.k_off[l] & 3) == 0;
    if (!vec && n_in * n_out > 1024) return AA_ERR_RANGE;
  }
  return AA_OK;
}

extern "C" {

// Measurement aid (tools/ppo_fused_probe.py): every workgroup of the next aa_ppo_fused_step
// launches writes wall_clock64() stamps at its phase boundaries to buf[n_wg][32]; NULL switches
// it off again.
int aa_ppo_fused_debug_stamps(int64_t* buf) {
  g_pf_stamps = reinterpret_cast<long long*>(buf);
  return AA_OK;
}

// Process-wide switch of the merged reduce + apply launch (1 by default); returns the old value.
int32_t aa_ppo_fused_merge_apply(int32_t on) {
  const int32_t old = g_pf_merge_apply;
  if (on >= 0) g_pf_merge_apply = on != 0;
  return old;
}

int64_t aa_ppo_fused_workspace_bytes(int64_t N, int64_t total_params) {
  if (N <= 0 || total_params <= 0) return -1;
  const int64_t n_wg = (N + PF_TS - 1) / PF_TS;
  return (n_wg * total_params + n_wg * 8 + (total_params + 15) / 16 + 16 + 2 * PF_MAX_EPOCH_STEPS +
          2 + 2 * (PF_MERGE_MAX_WG + 1)) * (int64_t)sizeof(float);
}

// n_steps consecutive minibatch steps from ONE host call: step s trains on rows
// rows_dev[s * N .. (s + 1) * N) of the sample arrays the descriptor points at (d.rows is ignored;
// rows_dev NULL with n_steps == 1: the N rows of the arrays themselves, or d.rows if set).
int aa_ppo_fused_epoch(const aa_ppo_fused_desc* dsc, const int64_t* rows_dev, int32_t n_steps,
                       float* grads, float* adam_m, float* adam_v, int64_t* adam_step_dev, float lr,
                       float beta1, float beta2, float adam_eps, float grad_clip, float* stats9,
                       float* sumsq_out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!dsc || !grads || !adam_m || !adam_v || !adam_step_dev || !stats9 || !workspace ||
      n_steps < 1)
    return AA_ERR_INVALID;
  if (rows_dev == nullptr && n_steps != 1) return AA_ERR_INVALID;
  const aa_ppo_fused_desc& d = *dsc;
  if (!d.obs || !d.actions || !d.old_loc || !d.old_scale || !d.returns || !d.adv ||
      !d.step_type || !d.params || d.N <= 0 || d.total <= 0 || !(d.denom > 0.f))
    return AA_ERR_INVALID;
  if (d.D < 1 || d.D > AA_PPO_FUSED_MAX_D || d.obs_dim < 1 || d.obs_dim > PF_W ||
      d.ld_obs < d.obs_dim)
    return AA_ERR_RANGE;
  if ((d.nrm_avg != nullptr) != (d.nrm_m2 != nullptr) ||
      (d.nrm_avg != nullptr) != (d.nrm_count != nullptr))
    return AA_ERR_INVALID;
  if ((d.act_mean != nullptr) != (d.act_mag != nullptr)) return AA_ERR_INVALID;
  int rc = pf_check_net(d.actor, d.obs_dim, d.D, d.total);
  if (rc != AA_OK) return rc;
  rc = pf_check_net(d.value, d.obs_dim, 1, d.total);
  if (rc != AA_OK) return rc;
  if (d.head_off < 0 || d.head_off + d.D > d.total || (d.total & 3) != 0) return AA_ERR_RANGE;
  if (((uintptr_t)d.params & 15) != 0 || ((uintptr_t)grads & 15) != 0 ||
      ((uintptr_t)workspace & 15) != 0)
    return AA_ERR_INVALID;
  if (workspace_bytes < aa_ppo_fused_workspace_bytes(d.N, d.total)) return AA_ERR_RANGE;
  const int64_t n_wg = (d.N + PF_TS - 1) / PF_TS;
  if (n_wg > 0x7fffffffLL) return AA_ERR_RANGE;
  // sizeof(PfLds) of dynamic LDS (see the static_assert): the attribute belongs to the CURRENT device's function object, so it
  // is granted once per device ordinal (a process that steps agents on two GPUs needs it on both)
  static bool lds_granted[64] = {};
  int dev_ord = 0;
  if (hipGetDevice(&dev_ord) != hipSuccess) return AA_ERR_LAUNCH;
  if (dev_ord < 0 || dev_ord >= 64 || !lds_granted[dev_ord]) {
    if (hipFuncSetAttribute((const void*)aa_ppo_fused_step_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(PfLds)) != hipSuccess)
      return AA_ERR_LAUNCH;
    if (dev_ord >= 0 && dev_ord < 64) lds_granted[dev_ord] = true;
  }
  PfArgs P;
  P.d = d;
  P.slabs = reinterpret_cast<float*>(workspace);
  P.partial = P.slabs + n_wg * d.total;
  P.stamps = g_pf_stamps;
  float* sumsq_part = P.partial + n_wg * 8;
  float* moments = sumsq_part + (d.total + 15) / 16 + 16;
  hipStream_t st = (hipStream_t)stream;
  const unsigned n_red = (unsigned)((d.total + 63) / 64);
  int64_t blocks = (d.total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  P.moments = nullptr;
  // K2 + K3 in one launch when its grid barrier is safe: every workgroup of it resident at once
  // (256 threads and < 6 KB of LDS each: one per CU leaves room on any gfx950 part)
  const bool merged = g_pf_merge_apply != 0 && n_red <= PF_MERGE_MAX_WG;
  PfApply ap;
  ap.p = const_cast<float*>(d.params); ap.m = adam_m; ap.v = adam_v;
  ap.lr = lr; ap.beta1 = beta1; ap.beta2 = beta2; ap.eps = adam_eps; ap.clip = grad_clip;
  ap.sumsq_out = sumsq_out;
  // the barrier slots (8 bytes per K2 workgroup) and the launch sequence live behind the moments;
  // the workspace is zero-filled once by its owner (the slab padding already relies on that)
  ap.slot = reinterpret_cast<unsigned long long*>(
      moments + 2 * PF_MAX_EPOCH_STEPS + ((n_wg * d.total + n_wg * 8 + (d.total + 15) / 16) & 1));
  P.seq = ap.slot + PF_MERGE_MAX_WG;
  ap.seq = P.seq;
  const bool ahead = rows_dev != nullptr && n_steps > 1 && n_steps <= PF_MAX_EPOCH_STEPS;
  if (ahead)     // the advantage moments of every minibatch of this call, one launch
    hipLaunchKernelGGL(aa_ppo_fused_moments_kernel, dim3((unsigned)n_steps), dim3(PF_THREADS), 0,
                       st, d.adv, rows_dev, d.N, moments);
  for (int s = 0; s < n_steps; ++s) {
    if (rows_dev != nullptr) P.d.rows = rows_dev + (int64_t)s * d.N;
    if (ahead) P.moments = moments + 2 * s;
    hipLaunchKernelGGL(aa_ppo_fused_step_kernel, dim3((unsigned)n_wg), dim3(PF_THREADS),
                       sizeof(PfLds), st, P);
    if (merged) {
      hipLaunchKernelGGL(aa_ppo_fused_reduce_kernel<true>, dim3(n_red), dim3(256), 0, st,
                         (const float*)P.slabs, (int)n_wg, d.total, grads, sumsq_part,
                         (const float*)P.partial, d, stats9, adam_step_dev, ap);
      continue;
    }
    hipLaunchKernelGGL(aa_ppo_fused_reduce_kernel<false>, dim3(n_red), dim3(256), 0, st,
                       (const float*)P.slabs, (int)n_wg, d.total, grads, sumsq_part,
                       (const float*)P.partial, d, stats9, adam_step_dev, ap);
    hipLaunchKernelGGL(aa_ppo_fused_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       const_cast<float*>(d.params), grads, adam_m, adam_v, d.total, lr, beta1,
                       beta2, adam_eps, (const int64_t*)adam_step_dev, (const float*)sumsq_part,
                       (int)n_red, grad_clip, sumsq_out);
  }
  return aa_launch_status();
}

int aa_ppo_fused_step(const aa_ppo_fused_desc* dsc, float* grads, float* adam_m, float* adam_v,
                      int64_t* adam_step_dev, float lr, float beta1, float beta2, float adam_eps,
                      float grad_clip, float* stats9, float* sumsq_out, void* workspace,
                      int64_t workspace_bytes, void* stream) {
  return aa_ppo_fused_epoch(dsc, nullptr, 1, grads, adam_m, adam_v, adam_step_dev, lr, beta1, beta2,
                            adam_eps, grad_clip, stats9, sumsq_out, workspace, workspace_bytes,
                            stream);
}

}  // extern "C"
