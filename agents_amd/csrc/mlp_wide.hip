// Whole MLPs of "wide" Dense layers (every hidden / output width <= 256, input <= 1024) at small
// batch, forward in ONE launch and backward in TWO, for up to four networks of the same layout per
// launch -- the actor and the twin (target) critics of SAC at batch 256
// (tf_agents/agents/sac/sac_agent.py:286-330 train; networks/critic_network.py:150-170 and
// actor_distribution_network.py: (256, 256) hidden layers in examples/sac/haarnoja18).
//
// Why: one SAC train step is ~40 Dense contractions of 25-50 MFLOP.  As one MFMA GEMM launch each
// (plus a split-K reduce, plus copies that build [observation | action]) the step was ~105
// launches of 4.5-8 us, i.e. bound by launch latency and by the dependent round trips of every
// launch (profiles/r03_a_sac_kernel_stats.csv), not by arithmetic or bytes.  Here:
//   forward  : a workgroup owns FOUR samples and walks all layers.  Activations live in LDS,
//              transposed ([feature][4 samples] -> one broadcast 16-byte read per feature); the
//              weights are streamed once per workgroup straight from L2: wave q takes an eighth
//              of the layer's input features, lane c four output columns, so a wave reads whole
//              rows of W (1 KiB, coalesced) with sixteen rows in flight per lane and does 16 FMAs
//              per 16-byte load.  Partial sums meet in LDS in wave order (deterministic).  The
//              first layer can read its input from two tensors ([observation | action]).
//   backward : (1) the gradient chain g_l -> dz_l = g_l * act'(y_l) -> g_{l-1} = dz_l W_l^T, four
//              samples per workgroup again; W_l^T is never formed: 32-row tiles of W travel
//              L2 -> registers -> LDS with coalesced whole-row loads, two tiles ahead, and are read
//              back one row per 16-lane group (see aa_mlp_wide_chain_kernel); narrow / odd-width
//              layers (heads) take a plain FMA loop.  dz_l is written for (2); the input gradient
//              is optional and can be limited to a column range (the action columns of a
//              critic's input in SAC's actor loss: only those rows of the first kernel are read).
//              (2) every weight gradient dW_l = H_{l-1}^T dz_l (+ bias column sums) of every layer
//              and network in one grid of 32 x 32 tiles on the fp32 matrix cores, full batch per
//              tile (no split-K slabs), every operand of a 64-sample chunk requested before the
//              first MFMA; written, not accumulated.
// Forward: at four samples per workgroup a batch of 256 is 64 workgroups per network, each streaming
// the network's weights (0.66 MB for a SAC critic) at the ~64 B/clk a CU draws from L2: ~4 us,
// against which the 16 FMAs per loaded float4 are balanced (2.6 us of VALU for the widest layer);
// measured 13 us per launch for one network or two.  For batches in the thousands the GEMM path
// (gemm.hip) is the better plan: callers use this one up to AA_MLPW_MAX_BATCH samples.
// Rules these kernels were rebuilt around ("Round 3"): no load under a per-lane or
// per-group condition in a steady state (the compiler then waits for ALL outstanding loads),
// older values pinned before a prefetch is issued, no scratch spills behind a prefetch, a
// run-time loop over the layers (unrolled, a chain kernel was 100 KB of code per launch).
#include "common.h"
#include "agents_amd.h"
#include "sac_sample.h"
#include "sac_loss.h"
#include "adam_elem.h"

#define MW_THREADS 512
#define MW_WAVES 8
#define MW_TS 4
#define MW_MAXW 256
#define MW_MAXIN 1024
#define MW_U 8           // weight rows per batch, two batches in flight per lane (forward)
#define MW_WT_LD (MW_MAXW + 4)

typedef float mw_f4 __attribute__((ext_vector_type(4)));

__device__ static inline float mw_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}
__device__ static inline float mw_actgrad(float y, int act) {
  if (act == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

struct MwNetF {
  const float* params;
  const float* x;
  const float* x2;
  int64_t ldx, ldx2;
  float* y[AA_MLP_MAX_LAYERS];
};
// The SAC actor's sample tail (aa_mlp_wide_forward_sample): net < 0 = none.
struct MwTail {
  int net, A, std_kind;
  const float* act_mean;
  const float* act_mag;
  const float* eps_in;
  uint32_t seed_lo, seed_hi;
  int64_t* call_counter;
  int64_t* arrival;
  float* action;
  float* logp;
  float* save_tanh;
  float* save_sigma;
  float* save_eps;
};
struct MwFwdP {
  aa_mlp_layout lay;
  MwNetF net[AA_MLPW_MAX_NETS];
  int64_t B;
  int x_split;
  long long* stamps;   // nullable (aa_mlp_wide_debug_stamps): [workgroup][16] wall_clock64 ticks
  MwTail tail[2];      // up to two networks of the launch draw their sample (SAC: the actor on the
                       // next observations and on the observations, two policies' counters)
};
static long long* g_mw_stamps = nullptr;
#define MW_STAMP(i)                                  \
  if (p.stamps != nullptr && threadIdx.x == 0)       \
    p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64();

__global__ void __launch_bounds__(MW_THREADS) aa_mlp_wide_fwd_kernel(MwFwdP p) {
  __shared__ mw_f4 hin[MW_MAXIN];               // input features of the 4 samples
  __shared__ mw_f4 hid[2][MW_MAXW];             // hidden activations, ping-pong
  __shared__ __attribute__((aligned(16))) float red[MW_WAVES][MW_TS][MW_MAXW];
  const int g = blockIdx.y;
  const int64_t s0 = (int64_t)blockIdx.x * MW_TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ params = p.net[g].params;
  MW_STAMP(0)
  {
    const float* __restrict__ x = p.net[g].x;
    const float* __restrict__ x2 = p.net[g].x2;
    const int64_t ldx = p.net[g].ldx, ldx2 = p.net[g].ldx2;
    const int n0 = p.lay.dims[0], split = p.x_split;
    float* hf = reinterpret_cast<float*>(hin);
    for (int i = tid; i < n0 * MW_TS; i += MW_THREADS) {
      const int s = i / n0, k = i - s * n0;
      float v = 0.f;
      if (s0 + s < p.B)
        v = k < split ? x[(s0 + s) * ldx + k] : x2[(s0 + s) * ldx2 + (k - split)];
      hf[k * MW_TS + s] = v;
    }
  }
  __syncthreads();
  MW_STAMP(1)
  const mw_f4* hcur = hin;
  const int L = p.lay.n_layers;
  for (int l = 0; l < L; ++l) {
    const int n_in = p.lay.dims[l], n_out = p.lay.dims[l + 1];
    const float* __restrict__ W = params + p.lay.k_off[l];
    const float* __restrict__ bias = params + p.lay.b_off[l];
    // the (<= 2) outputs this thread finishes after the reduction, and their biases: requested
    // now, needed after the weight stream
    const int i_a = tid, i_b = tid + MW_THREADS;
    const int s_a = i_a / n_out, c_a = i_a - s_a * n_out;
    const int s_b = i_b / n_out, c_b = i_b - s_b * n_out;
    const float bias_a = i_a < MW_TS * n_out ? bias[c_a] : 0.f;
    const float bias_b = i_b < MW_TS * n_out ? bias[c_b] : 0.f;
    float acc[MW_TS][4];
#pragma unroll
    for (int s = 0; s < MW_TS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[s][j] = 0.f;
    const bool vec = (n_out & 3) == 0 && (((uintptr_t)W) & 15) == 0 && n_out > 64;
    if (vec) {
      // wave q: rows [kb, ke) (a multiple of MW_U except for the last wave); lane c: four columns
      int kq = (n_in + MW_WAVES - 1) / MW_WAVES;
      kq = ((kq + MW_U - 1) / MW_U) * MW_U;
      const int kb = wave * kq < n_in ? wave * kq : n_in;
      const int ke = kb + kq < n_in ? kb + kq : n_in;
      const int c4 = lane * 4;
      if (c4 < n_out) {
        // two batches of MW_U rows: the next batch's loads are in flight while this one is
        // multiplied.  The steady-state loop is straight-line code (no per-row guards: the
        // compiler then counts the outstanding loads exactly and waits for ONE batch).
        mw_f4 wa[MW_U], wb[MW_U];
        const float* __restrict__ Wc = W + c4;
#define MW_LOAD(w, kk)                                   \
  _Pragma("unroll") for (int u = 0; u < MW_U; ++u)       \
      (w)[u] = *reinterpret_cast<const mw_f4*>(Wc + (int64_t)((kk) + u) * n_out);
#define MW_FMA(w, kk)                                                                    \
  _Pragma("unroll") for (int u = 0; u < MW_U; ++u) {                                     \
    const mw_f4 h = hcur[(kk) + u];                                                      \
    _Pragma("unroll") for (int s = 0; s < MW_TS; ++s)                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[s][j] =                        \
            fmaf(h[s], (w)[u][j], acc[s][j]);                                            \
  }
        const int kfull = kb + ((ke - kb) / MW_U) * MW_U;
        int k = kb;
        if (k < kfull) {
          MW_LOAD(wa, k)
          // steady state: no condition between a load and the multiply that waits for it, so the
          // compiler knows exactly one batch is outstanding behind the one it needs
          while (k + 2 * MW_U < kfull) {
            MW_LOAD(wb, k + MW_U)
            MW_FMA(wa, k)
            MW_LOAD(wa, k + 2 * MW_U)
            MW_FMA(wb, k + MW_U)
            k += 2 * MW_U;
          }
          if (k + MW_U < kfull) {       // two batches left: wa (loaded) and one more
            MW_LOAD(wb, k + MW_U)
            MW_FMA(wa, k)
            MW_FMA(wb, k + MW_U)
            k += 2 * MW_U;
          } else {
            MW_FMA(wa, k)
            k += MW_U;
          }
        }
#undef MW_LOAD
#undef MW_FMA
        for (; k < ke; ++k) {   // the last wave's remainder
          const mw_f4 w = *reinterpret_cast<const mw_f4*>(Wc + (int64_t)k * n_out);
          const mw_f4 h = hcur[k];
#pragma unroll
          for (int s = 0; s < MW_TS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[s][j] = fmaf(h[s], w[j], acc[s][j]);
        }
      }
#pragma unroll
      for (int s = 0; s < MW_TS; ++s)
        *reinterpret_cast<mw_f4*>(&red[wave][s][c4]) =
            mw_f4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
    } else {
      // narrow layer (a head: 1, 6, 34 ... columns) or an odd width: a lane owns ONE column and a
      // share of the rows -- cg column slots (the power of two >= n_out, <= 64... 256), 512 / cg row
      // groups -- so that all lanes load at once even for a single output column
      int cg = 1;
      while (cg < n_out) cg <<= 1;
      const int col = tid & (cg - 1);
      const int kg = tid / cg, nkg = MW_THREADS / cg;
      float a1[MW_TS] = {0.f, 0.f, 0.f, 0.f};
      if (col < n_out) {
        int k = kg;
        for (; k + 3 * nkg < n_in; k += 4 * nkg) {
          float w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = W[(int64_t)(k + u * nkg) * n_out + col];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const mw_f4 h = hcur[k + u * nkg];
#pragma unroll
            for (int s = 0; s < MW_TS; ++s) a1[s] = fmaf(h[s], w[u], a1[s]);
          }
        }
        for (; k < n_in; k += nkg) {
          const float w = W[(int64_t)k * n_out + col];
          const mw_f4 h = hcur[k];
#pragma unroll
          for (int s = 0; s < MW_TS; ++s) a1[s] = fmaf(h[s], w, a1[s]);
        }
      }
      // lanes of a wave that share a column: lane, lane + cg, lane + 2 cg ... (cg < 64)
      for (int off = 32; off >= cg; off >>= 1)
#pragma unroll
        for (int s = 0; s < MW_TS; ++s) a1[s] += __shfl_down(a1[s], off, 64);
      // red[wave][s][col]: for cg >= 64 a wave holds 64 / ... one row group per (wave, col / 64)
      if (cg <= 64) {
        if (lane < cg) {
#pragma unroll
          for (int s = 0; s < MW_TS; ++s) red[wave][s][lane] = col < n_out ? a1[s] : 0.f;
        }
      } else {
        // cg = 128 / 256: nkg = 4 / 2 row groups; this thread's sum goes to slot kg
#pragma unroll
        for (int s = 0; s < MW_TS; ++s) red[kg][s][col] = col < n_out ? a1[s] : 0.f;
        // the other slots of the fixed eight-term sum below hold zeros
        for (int q = nkg + (tid / cg); q < MW_WAVES; q += nkg)
#pragma unroll
          for (int s = 0; s < MW_TS; ++s) red[q][s][col] = 0.f;
      }
    }
    MW_STAMP(2 + 3 * l)
    __syncthreads();
    MW_STAMP(3 + 3 * l)
    float* __restrict__ y = p.net[g].y[l];
    float* hn = reinterpret_cast<float*>(hid[l & 1]);
    const int act = p.lay.acts[l];
    if (i_a < MW_TS * n_out) {
      float v = bias_a;
#pragma unroll
      for (int q = 0; q < MW_WAVES; ++q) v += red[q][s_a][c_a];
      v = mw_act(v, act);
      if (s0 + s_a < p.B) y[(s0 + s_a) * n_out + c_a] = v;
      hn[c_a * MW_TS + s_a] = v;
    }
    if (i_b < MW_TS * n_out) {
      float v = bias_b;
#pragma unroll
      for (int q = 0; q < MW_WAVES; ++q) v += red[q][s_b][c_b];
      v = mw_act(v, act);
      if (s0 + s_b < p.B) y[(s0 + s_b) * n_out + c_b] = v;
      hn[c_b * MW_TS + s_b] = v;
    }
    __syncthreads();
    MW_STAMP(4 + 3 * l)
    hcur = hid[l & 1];
  }
  // ---- sample tail: this network is a SAC actor and hcur holds z = [mean | raw_std] of the
  // workgroup's MW_TS samples ([column][sample]).  One thread per (sample, action dimension), the
  // A log-density terms of a sample summed in dimension order by one thread: aa_sac_sample_kernel's
  // arithmetic (sac_sample.h) and Philox counters, without the launch.
  if (p.tail[0].net == g || p.tail[1].net == g) {
    const MwTail& T = p.tail[0].net == g ? p.tail[0] : p.tail[1];
    const int A = T.A;
    const float* zf = reinterpret_cast<const float*>(hcur);
    float* terms = &red[0][0][0];                 // [MW_TS][A]: the layers are done with `red`
    const uint64_t call = T.call_counter != nullptr ? (uint64_t)T.call_counter[0] : 0ull;
    for (int i = tid; i < MW_TS * A; i += MW_THREADS) {
      const int sl = i / A, d = i - sl * A;
      const int64_t b = s0 + sl;
      float term = 0.f;
      if (b < p.B) {
        const AaSacElem o = aa_sac_sample_elem(zf[d * MW_TS + sl], zf[(A + d) * MW_TS + sl],
                                               T.std_kind, T.eps_in, (uint64_t)(b * A + d), call,
                                               T.seed_lo, T.seed_hi, T.act_mean[d], T.act_mag[d]);
        T.action[b * A + d] = o.action;
        term = o.term;
        if (T.save_tanh != nullptr) {
          T.save_tanh[b * A + d] = o.t;
          T.save_sigma[b * A + d] = o.sigma;
          T.save_eps[b * A + d] = o.eps;
        }
      }
      terms[i] = term;
    }
    __syncthreads();
    if (tid < MW_TS && s0 + tid < p.B) {
      float lp = 0.f;
      for (int k = 0; k < A; ++k) lp += terms[tid * A + k];   // dimension order
      T.logp[s0 + tid] = lp;
    }
    if (T.arrival != nullptr && T.eps_in == nullptr && T.call_counter != nullptr)
      aa_advance_when_all_done(T.call_counter, T.arrival, 1, gridDim.x);
  }
}

// ---- backward (1): the gradient chain ---------------------------------------------------------------
struct MwNetB {
  const float* params;
  const float* y[AA_MLP_MAX_LAYERS];
  const float* dout;
  int64_t ld_dout;
  float* dz[AA_MLP_MAX_LAYERS];
  float* dx;
  int64_t ld_dx;
};
// dout generators (aa_mlp_wide_backward_gen): the launch computes d loss / d output itself instead
// of reading what a loss launch left -- SAC's critic loss, actor loss and actor-head backward are a
// few loads and flops per sample, and each was a launch of its own on the train step's chain.
struct MwGen {
  int kind;                 // 0 none, AA_SAC_GEN_CRITIC / _ACTOR / _HEAD
  const float* q1; const float* q2;
  const float* tq1; const float* tq2; const float* next_logp; const float* reward;
  const float* discount;
  const float* logp;
  const float* weights;
  const float* log_alpha;
  float gamma, reward_scale;
  int loss_kind;
  float loss_weight, global_batch;
  float* loss_out;
  float* td_target_out;
  float* dlogp_out;
  const float* z; int A, std_kind;
  const float* act_mag; const float* save_tanh; const float* save_sigma; const float* save_eps;
  const float* daction; int64_t ld_da; const float* daction2; int64_t ld_da2;
  const float* dlogp;
};
struct MwBwdP {
  aa_mlp_layout lay;
  MwNetB net[AA_MLPW_MAX_NETS];
  int64_t B;
  int dx_lo, dx_hi;
  long long* stamps;   // nullable (aa_mlp_wide_debug_stamps): [workgroup][16] wall_clock64 ticks
  MwGen gen;
};


// Workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() carries a release
// fence that also drains the vector-memory counter, i.e. the dz stores (read by the NEXT launch)
// and the prefetched rows of W (waited for where they are used).
__device__ static inline void mw_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef float mw_acc4 __attribute__((ext_vector_type(4)));

// ---- backward (1): four samples per workgroup, W through LDS in 32-row tiles ----------------------
// g_{l-1}[s][i] = sum_o dz[s][o] W[i][o]: a tile of 32 rows of W travels L2 -> registers -> LDS with
// coalesced whole-row loads (two tiles ahead of the one being multiplied); lane (row, column group
// og) multiplies its 16 columns for the 4 samples (dz in registers) and the 16 partial sums of a
// (row, sample) meet inside a 16-lane DPP row.  64 workgroups per network, each streams the layer's
// weights once.  17.3 us per launch for a critic pair with the action gradient.
// Tried and removed: sixteen samples per workgroup on v_mfma_f32_16x16x4_f32 with the rows of W
// read straight from L2 in MFMA layout (64-byte runs): 21.5 us -- 4.5 us of row loads behind
// sixteen waves and 4.9 us of MFMA per 256 x 256 layer on the in-kernel timeline (3.4 us is ONE CU's
// fp32 MFMA rate for 16 x 256 x 256); partial sums through __shfl_xor (ds_bpermute): 19.4 us;
// through LDS with a two-wave reduce phase: 17.9 us.
#define MW_VT 512

// v of the lane N places to the left inside its 16-lane row (cyclic): a DPP modifier on a VALU
// move, no LDS crossbar.  Four of them (8, 4, 2, 1) leave the row's sum in every lane.
template <int N>
__device__ static inline float mw_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}

__global__ void __launch_bounds__(MW_VT) aa_mlp_wide_chain_kernel(MwBwdP p) {
  __shared__ float dzs[MW_TS][MW_MAXW];
  __shared__ float gs[MW_TS][MW_MAXW];
  __shared__ __attribute__((aligned(16))) float Wt[32][MW_WT_LD];
  const int g = blockIdx.y;
  const int64_t s0 = (int64_t)blockIdx.x * MW_TS;
  const int tid = threadIdx.x;
  const float* __restrict__ params = p.net[g].params;
  const int L = p.lay.n_layers;
  MW_STAMP(0)
  // element (sample, column) = (tid >> 8) + 2 h, tid & 255 for h < 2: every saved activation this
  // thread needs, of every layer, is requested now (unconditionally, from clamped addresses)
  const int ec = tid & 255, es = tid >> 8;
  float yv[AA_MLP_MAX_LAYERS][2];
#pragma unroll
  for (int l = 0; l < AA_MLP_MAX_LAYERS; ++l) {
    const int ll = l < L ? l : L - 1;
    const int n_out = p.lay.dims[ll + 1];
    const float* __restrict__ y = p.net[g].y[ll];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = es + 2 * h;
      const bool live = ec < n_out && s0 + s < p.B;
      yv[l][h] = y[live ? (s0 + s) * n_out + ec : 0];
    }
  }
  if (p.gen.kind == 0) {
    const float* __restrict__ dout = p.net[g].dout;
    const int64_t ld = p.net[g].ld_dout;
    const int n = p.lay.dims[L];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = es + 2 * h;
      const bool live = ec < n && s0 + s < p.B;
      const float v = dout[live ? (s0 + s) * ld + ec : 0];
      if (ec < n) gs[s][ec] = live ? v : 0.f;
    }
  } else {
    // d loss / d output computed here (sac_loss.h: the loss kernels' own per-sample arithmetic)
    const MwGen& G = p.gen;
    const int n = p.lay.dims[L];
    if (G.kind == AA_SAC_GEN_HEAD) {
      const int A = G.A;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int s = es + 2 * h;
        const int64_t b = s0 + s;
        const bool live = ec < n && b < p.B;
        float v = 0.f;
        if (live) {
          const int d = ec < A ? ec : ec - A;
          const int64_t i = b * A + d;
          float da = G.daction != nullptr ? G.daction[b * G.ld_da + d] : 0.f;
          if (G.daction2 != nullptr) da = da + G.daction2[b * G.ld_da2 + d];
          float gx, draw;
          aa_sac_head_bwd_elem(G.save_tanh[i], G.save_sigma[i], G.save_eps[i], G.dlogp[b], da,
                               G.act_mag[d], G.z[b * 2 * A + A + d], G.std_kind, &gx, &draw);
          v = ec < A ? gx : draw;
        }
        if (ec < n) gs[s][ec] = v;
      }
    } else {
      // the twin critics (networks 0 and 1 of the launch): one output column each
      const float alpha = expf(G.log_alpha[0]);
      if (ec == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int s = es + 2 * h;
          const int64_t b = s0 + s;
          float v = 0.f;
          if (b < p.B) {
            if (G.kind == AA_SAC_GEN_CRITIC) {
              const AaSacCriticElem o = aa_sac_critic_elem(
                  G.q1[b], G.q2[b], G.tq1[b], G.tq2[b], G.next_logp[b], G.reward[b], G.discount[b],
                  G.weights, b, alpha, G.gamma, G.reward_scale, G.loss_kind);
              v = (G.loss_weight * (g == 0 ? o.g1 : o.g2) * o.w) / G.global_batch;
              if (g == 0 && G.td_target_out != nullptr) G.td_target_out[b] = o.td;
            } else {
              const AaSacActorElem o = aa_sac_actor_elem(G.q1[b], G.q2[b], G.logp[b], G.weights, b,
                                                         alpha, G.loss_weight, G.global_batch);
              v = g == 0 ? o.dq1 : o.dq2;
              if (g == 0) G.dlogp_out[b] = o.dlogp;
            }
          }
          gs[s][0] = v;
        }
      }
      // the loss value: workgroup (0, 0), with the loss kernel's own reduction (256 lanes walking
      // the batch, fixed tree)
      if (blockIdx.x == 0 && g == 0) {
        __shared__ float gred[16];
        float local = 0.f;
        if (tid < 256) {
          for (int64_t b = tid; b < p.B; b += 256) {
            if (G.kind == AA_SAC_GEN_CRITIC) {
              local += aa_sac_critic_elem(G.q1[b], G.q2[b], G.tq1[b], G.tq2[b], G.next_logp[b],
                                          G.reward[b], G.discount[b], G.weights, b, alpha, G.gamma,
                                          G.reward_scale, G.loss_kind).wl;
            } else {
              local += aa_sac_actor_elem(G.q1[b], G.q2[b], G.logp[b], G.weights, b, alpha,
                                         G.loss_weight, G.global_batch).wl;
            }
          }
        }
        const float total = aa_block_sum(local, gred);
        if (tid == 0) G.loss_out[0] = G.loss_weight * (total / G.global_batch);
      }
    }
  }
  mw_lds_barrier();
  // every act' factor is pinned in its register here, once: a wait for one of them placed behind
  // a layer's (conditional) tile prefetch would have to be a wait for the prefetch as well
#pragma unroll
  for (int l = 0; l < AA_MLP_MAX_LAYERS; ++l)
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(yv[l][h]));
  MW_STAMP(1)
  for (int l = L - 1; l >= 0; --l) {
    const int n_in = p.lay.dims[l], n_out = p.lay.dims[l + 1];
    const float* __restrict__ W = params + p.lay.k_off[l];
    float* __restrict__ dz = p.net[g].dz[l];
    const int act = p.lay.acts[l];
    int lo = 0, hi = n_in;
    float* __restrict__ dx = nullptr;
    if (l == 0) {
      dx = p.net[g].dx;
      lo = dx != nullptr ? p.dx_lo : 0;
      hi = dx != nullptr ? p.dx_hi : 0;
    }
    const int64_t ld_dx = p.net[g].ld_dx;
    // (widths that are a multiple of 64: 128, 192, 256; the others take the plain loop below)
    const bool wide = hi > lo && n_out > 64 && (n_out & 63) == 0 && (((uintptr_t)W) & 15) == 0;
    // the first two tiles of a wide layer are requested BEFORE the dz phase (they do not depend on
    // it); rows are clamped into [lo, hi): loads are unconditional
    const int n4 = n_out >> 2;
    int pr[4], pc[4];
    mw_f4 pre[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * MW_VT;
      pr[q] = wide && e < 32 * n4 ? e / n4 : -1;
      pc[q] = pr[q] >= 0 ? (e - pr[q] * n4) * 4 : 0;
    }
#define MW_ISSUE(slot, row0)                                                         \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
    int r_ = (row0) + (pr[q] >= 0 ? pr[q] : 0);                                      \
    r_ = r_ < hi ? r_ : hi - 1;                                                      \
    pre[slot][q] = *reinterpret_cast<const mw_f4*>(W + (int64_t)r_ * n_out + pc[q]); \
  }
#define MW_COMMIT(slot)                                                              \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) if (pr[q] >= 0)                      \
      *reinterpret_cast<mw_f4*>(&Wt[pr[q]][pc[q]]) = pre[slot][q];
    if (wide) {
      MW_ISSUE(0, lo)
      MW_ISSUE(1, lo + 32)
    }
    // dz = g * act'(y)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = es + 2 * h;
      if (ec < n_out) {
        // the layer's saved activation, selected by OR-ing masked bit patterns: a chain of
        // `if (l == k) yy = yv[k][h]` is turned into an indexed load by the compiler, which sends
        // yv[][] to the stack (48 bytes of scratch per lane, two scratch loads per layer --
        // tools/kernel_resources.py); this form keeps all eight values in registers
        unsigned yb = 0u;
#pragma unroll
        for (int k = 0; k < AA_MLP_MAX_LAYERS; ++k) yb |= l == k ? __float_as_uint(yv[k][h]) : 0u;
        const float yy = __uint_as_float(yb);
        const float d = s0 + s < p.B ? gs[s][ec] * mw_actgrad(yy, act) : 0.f;
        dzs[s][ec] = d;
        if (s0 + s < p.B) dz[(s0 + s) * n_out + ec] = d;
      }
    }
    mw_lds_barrier();
    MW_STAMP(2 + 2 * (L - 1 - l))
    if (wide) {
      // lane = (row r of the wave's four, column group og): group og owns columns
      // 64 j + 4 og .. + 3 (j < 4) -- a 16-lane row of the wave reads 256 consecutive bytes of
      // LDS per instruction (conflict-free) and the 16 partial sums of a (row, sample) meet
      // inside that 16-lane row by DPP row rotations: no partial sums through LDS, no reduce phase that
      // two waves execute while six wait (0.4 us per tile on the in-kernel timeline)
      const int lane = tid & 63, og = lane & 15, il = (tid >> 6) * 4 + (lane >> 4);
      float dzr[MW_TS][16];
#pragma unroll
      for (int s = 0; s < MW_TS; ++s)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int c = (j >> 2) * 64 + og * 4 + (j & 3);
          dzr[s][j] = c < n_out ? dzs[s][c] : 0.f;
        }
      // Tiles in PAIRS, both always executed, every load unconditional (rows clamped into the
      // range; rows beyond it are multiplied and dropped): between a tile's loads and the store
      // to LDS that waits for them lies a fixed number of younger loads, so the compiler waits
      // for exactly that tile (with a conditional issue it waited for everything: 1.1 us per tile
      // on the in-kernel timeline).
#define MW_TILE(slot, row0)                                                                    \
  {                                                                                            \
    MW_COMMIT(slot)                                                                            \
    mw_lds_barrier();                                                                          \
    MW_ISSUE(slot, (row0) + 64)                                                                \
    mw_f4 w_[4];                                                                               \
    _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4)                                           \
        w_[j4] = j4 * 64 < n_out ? *reinterpret_cast<const mw_f4*>(&Wt[il][j4 * 64 + og * 4])  \
                                 : mw_f4{0.f, 0.f, 0.f, 0.f};                                  \
    float acc[MW_TS] = {0.f, 0.f, 0.f, 0.f};                                                   \
    _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4)                                           \
        _Pragma("unroll") for (int s = 0; s < MW_TS; ++s)                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[s] =                             \
                fmaf(dzr[s][j4 * 4 + j], w_[j4][j], acc[s]);                                   \
    _Pragma("unroll") for (int s = 0; s < MW_TS; ++s) {                                        \
      acc[s] += mw_row_ror<8>(acc[s]);                                                         \
      acc[s] += mw_row_ror<4>(acc[s]);                                                         \
      acc[s] += mw_row_ror<2>(acc[s]);                                                         \
      acc[s] += mw_row_ror<1>(acc[s]);                                                         \
    }                                                                                          \
    const int i = (row0) + il;                                                                 \
    if (og == 0 && i < hi) {                                                                   \
      _Pragma("unroll") for (int s = 0; s < MW_TS; ++s) {                                      \
        if (l > 0) gs[s][i] = acc[s];                                                          \
        else if (s0 + s < p.B) dx[(s0 + s) * ld_dx + i] = acc[s];                              \
      }                                                                                        \
    }                                                                                          \
    mw_lds_barrier(); /* every lane is done with this tile before the next one lands in Wt */  \
  }
      for (int i0 = lo; i0 < hi; i0 += 64) {
        MW_TILE(0, i0)
        MW_TILE(1, i0 + 32)
      }
#undef MW_TILE
    } else if (hi > lo) {
      // narrow or odd-width layer: one lane per row of W, plain FMA loop
      for (int i = lo + tid; i < hi; i += MW_VT) {
        const float* __restrict__ wr = W + (int64_t)i * n_out;
        float acc[MW_TS] = {0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < n_out; ++o) {
          const float w = wr[o];
#pragma unroll
          for (int s = 0; s < MW_TS; ++s) acc[s] = fmaf(dzs[s][o], w, acc[s]);
        }
#pragma unroll
        for (int s = 0; s < MW_TS; ++s) {
          if (l > 0) gs[s][i] = acc[s];
          else if (s0 + s < p.B) dx[(s0 + s) * ld_dx + i] = acc[s];
        }
      }
    }
#undef MW_ISSUE
#undef MW_COMMIT
    mw_lds_barrier();
    MW_STAMP(3 + 2 * (L - 1 - l))
  }
}

// ---- backward (2): every dW / db of every layer and network ------------------------------------------
struct MwNetW {
  const float* x;
  const float* x2;
  int64_t ldx, ldx2;
  const float* y[AA_MLP_MAX_LAYERS];
  const float* dz[AA_MLP_MAX_LAYERS];
  float* grads;
};
// The optimizer step of the networks' parameters inside the weight-gradient launch
// (aa_mlp_wide_backward_gen_adam): a tile's workgroup holds the finished gradient of its 32 x 32
// parameters -- Adam (and the soft update of a target copy) is elementwise, so it applies it on
// the spot: aa_adam_kernel's arithmetic (adam_elem.h) and in-launch step counter, one launch less.
struct MwAdam {
  int on;
  float* p[AA_MLPW_MAX_NETS];
  float* m[AA_MLPW_MAX_NETS];
  float* v[AA_MLPW_MAX_NETS];
  float* target[AA_MLPW_MAX_NETS];     // nullable: soft_variables_update of a target copy
  float lr, beta1, beta2, eps, tau;
  int64_t* step_dev;                   // steps taken so far; the last workgroup adds one
  int64_t* arrival;
};
struct MwDwP {
  aa_mlp_layout lay;
  MwNetW net[AA_MLPW_MAX_NETS];
  int64_t B;
  int x_split;
  int tile_start[AA_MLP_MAX_LAYERS + 1];
  MwAdam adam;
};

__global__ void __launch_bounds__(256) aa_mlp_wide_dw_kernel(MwDwP p) {
  __shared__ float part[4][32][33];
  __shared__ float bsum[8][32];
  __shared__ float s_alpha;
  const int g = blockIdx.y;
  const int t = blockIdx.x;
  if (p.adam.on && threadIdx.x == 0)     // every workgroup reads the step count before any adds to it
    s_alpha = adam_alpha(p.adam.lr, p.adam.beta1, p.adam.beta2, (float)(*p.adam.step_dev + 1));
  int l = 0;
  while (l + 1 < p.lay.n_layers && t >= p.tile_start[l + 1]) ++l;
  const int n_in = p.lay.dims[l], n_out = p.lay.dims[l + 1];
  const int tn_count = (n_out + 31) / 32;
  const int local = t - p.tile_start[l];
  const int tm = local / tn_count, tn = local - tm * tn_count;
  const int i0 = tm * 32, o0 = tn * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // layer input H [B][n_in]: the network input (two tensors) or the previous layer's output
  const float* __restrict__ H = l == 0 ? p.net[g].x : p.net[g].y[l - 1];
  const float* __restrict__ H2 = p.net[g].x2;
  const int64_t ldh = l == 0 ? p.net[g].ldx : (int64_t)n_in;
  const int64_t ldh2 = p.net[g].ldx2;
  const int split = l == 0 ? p.x_split : n_in;
  const float* __restrict__ dz = p.net[g].dz[l];
  const int B = (int)p.B;
  int kq = (B + 3) / 4;
  kq = (kq + 3) & ~3;
  const int kb = wave * kq;
  const int ke = kb + kq < B ? kb + kq : B;
  mw_acc4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = mw_acc4{0.f, 0.f, 0.f, 0.f};
  // the optimizer's operands of this thread's four tile elements are requested now and land
  // under the batch loop (clamped addresses, unconditional loads)
  float ap[4], am[4], avv[4], at[4];
  const bool adam_on = p.adam.on != 0;
  const bool has_target = adam_on && p.adam.target[g] != nullptr;
  if (adam_on) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j, r = e >> 5, c = e & 31;
      const int rr = i0 + r < n_in ? i0 + r : n_in - 1, cc = o0 + c < n_out ? o0 + c : n_out - 1;
      const int64_t idx = p.lay.k_off[l] + (int64_t)rr * n_out + cc;
      ap[j] = p.adam.p[g][idx];
      am[j] = p.adam.m[g][idx];
      avv[j] = p.adam.v[g][idx];
      at[j] = has_target ? p.adam.target[g][idx] : 0.f;
    }
  }
  const int lr = lane & 15, lk = lane >> 4;
  // chunks of 64 samples: every operand of the chunk is requested before the first MFMA (one
  // round trip per chunk -- a batch of 256 is one chunk per wave)
  for (int k0 = kb; k0 < ke; k0 += 64) {
    float av[16][2], bv[16][2];
    // loads from clamped (always valid) addresses first, masks afterwards: a load under a
    // condition is waited for at the join with its zero alternative, one round trip each
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int k = k0 + u * 4 + lk;
      k = k < ke ? k : ke - 1;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        int i = i0 + m * 16 + lr;
        i = i < n_in ? i : n_in - 1;
        const float* src = i < split ? H + (int64_t)k * ldh + i : H2 + (int64_t)k * ldh2 + (i - split);
        av[u][m] = *src;
        int o = o0 + m * 16 + lr;
        o = o < n_out ? o : n_out - 1;
        bv[u][m] = dz[(int64_t)k * n_out + o];
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const bool kin = k0 + u * 4 + lk < ke;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (!(kin && i0 + m * 16 + lr < n_in)) av[u][m] = 0.f;
        if (!(kin && o0 + m * 16 + lr < n_out)) bv[u][m] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][a * 16 + 4 * lk + r][b * 16 + lr] = acc[a][b][r];
  // bias gradient: column sums of dz over the batch, by the tiles of the first row band
  const bool do_bias = tm == 0;
  if (do_bias) {
    const int c = tid & 31, kg = tid >> 5;
    float sacc = 0.f;
    if (o0 + c < n_out)
      for (int k = kg; k < B; k += 8) sacc += dz[(int64_t)k * n_out + o0 + c];
    bsum[kg][c] = sacc;
  }
  __syncthreads();
  float* __restrict__ grads = p.net[g].grads;
  const bool adam = p.adam.on != 0;
  const float alpha = adam ? s_alpha : 0.f;       // (written before the barrier above)
  const float omb1 = 1.0f - p.adam.beta1, omb2 = 1.0f - p.adam.beta2;
  auto step = [&](int64_t idx, float gv) {
    float pv = p.adam.p[g][idx], mv = p.adam.m[g][idx], vv = p.adam.v[g][idx];
    adam_elem(pv, gv, mv, vv, alpha, omb1, omb2, p.adam.eps);
    p.adam.p[g][idx] = pv;
    p.adam.m[g][idx] = mv;
    p.adam.v[g][idx] = vv;
    if (p.adam.target[g] != nullptr)
      p.adam.target[g][idx] = soft_update_elem(p.adam.target[g][idx], pv, p.adam.tau);
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + 256 * j;
    const int r = e >> 5, c = e & 31;
    const float v = ((part[0][r][c] + part[1][r][c]) + part[2][r][c]) + part[3][r][c];
    if (i0 + r < n_in && o0 + c < n_out) {
      const int64_t idx = p.lay.k_off[l] + (int64_t)(i0 + r) * n_out + o0 + c;
      grads[idx] = v;
      if (adam) {
        float pv = ap[j], mv = am[j], vv = avv[j];
        adam_elem(pv, v, mv, vv, alpha, omb1, omb2, p.adam.eps);
        p.adam.p[g][idx] = pv;
        p.adam.m[g][idx] = mv;
        p.adam.v[g][idx] = vv;
        if (has_target) p.adam.target[g][idx] = soft_update_elem(at[j], pv, p.adam.tau);
      }
    }
  }
  if (do_bias && tid < 32 && o0 + tid < n_out) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += bsum[q][tid];
    const int64_t idx = p.lay.b_off[l] + o0 + tid;
    grads[idx] = v;
    if (adam) step(idx, v);
  }
  if (adam) aa_advance_when_all_done(p.adam.step_dev, p.adam.arrival, 1, gridDim.x * gridDim.y);
}

static int mw_check_layout(const aa_mlp_layout* lay) {
  if (lay->n_layers < 1 || lay->n_layers > AA_MLP_MAX_LAYERS) return AA_ERR_RANGE;
  if (lay->dims[0] < 1 || lay->dims[0] > MW_MAXIN) return AA_ERR_RANGE;
  for (int l = 0; l < lay->n_layers; ++l) {
    if (lay->dims[l + 1] < 1 || lay->dims[l + 1] > MW_MAXW) return AA_ERR_RANGE;
    if (lay->k_off[l] < 0 || lay->b_off[l] < 0) return AA_ERR_INVALID;
    if (lay->acts[l] != AA_ACT_NONE && lay->acts[l] != AA_ACT_RELU && lay->acts[l] != AA_ACT_TANH)
      return AA_ERR_INVALID;
  }
  return AA_OK;
}

extern "C" {

// Measurement aid (tools/mlp_wide_probe.py): every workgroup of the following
// aa_mlp_wide_backward launches writes wall_clock64() stamps (10 ns ticks) at the phase boundaries
// of its gradient chain to buf[workgroup][16]; NULL switches it off again.
int aa_mlp_wide_debug_stamps(int64_t* buf) {
  g_mw_stamps = reinterpret_cast<long long*>(buf);
  return AA_OK;
}

int aa_mlp_wide_supported(const aa_mlp_layout* layout, int64_t B) {
  if (layout == nullptr || B < 1 || B > AA_MLPW_MAX_BATCH) return 0;
  return mw_check_layout(layout) == AA_OK ? 1 : 0;
}

static int mw_forward(const aa_mlp_wide_fwd* d, const aa_sac_sample_tail* t,
                      const aa_sac_sample_tail* t2, void* stream);

int aa_mlp_wide_forward(const aa_mlp_wide_fwd* d, void* stream) {
  return mw_forward(d, nullptr, nullptr, stream);
}

int aa_mlp_wide_forward_sample(const aa_mlp_wide_fwd* d, const aa_sac_sample_tail* tail,
                               void* stream) {
  if (tail == nullptr) return AA_ERR_INVALID;
  return mw_forward(d, tail, nullptr, stream);
}

int aa_mlp_wide_forward_sample2(const aa_mlp_wide_fwd* d, const aa_sac_sample_tail* tail_a,
                                const aa_sac_sample_tail* tail_b, void* stream) {
  if (tail_a == nullptr || tail_b == nullptr || tail_a->net == tail_b->net) return AA_ERR_INVALID;
  // two draws of one launch must not share a Philox counter: each tail's last workgroup advances
  // its own
  if (tail_a->eps_in == nullptr && tail_b->eps_in == nullptr &&
      (tail_a->call_counter_dev == tail_b->call_counter_dev ||
       tail_a->arrival_dev == tail_b->arrival_dev))
    return AA_ERR_INVALID;
  return mw_forward(d, tail_a, tail_b, stream);
}

static int mw_forward(const aa_mlp_wide_fwd* d, const aa_sac_sample_tail* t_a,
                      const aa_sac_sample_tail* t_b, void* stream) {
  if (d == nullptr || d->B < 1 || d->n_nets < 1 || d->n_nets > AA_MLPW_MAX_NETS) return AA_ERR_INVALID;
  int rc = mw_check_layout(&d->layout);
  if (rc != AA_OK) return rc;
  if (d->x_split < 0 || d->x_split > d->layout.dims[0]) return AA_ERR_INVALID;
  MwFwdP p;
  p.lay = d->layout;
  p.B = d->B;
  p.x_split = d->x_split;
  p.stamps = g_mw_stamps;
  for (int g = 0; g < AA_MLPW_MAX_NETS; ++g) {
    const int s = g < d->n_nets ? g : 0;
    if (d->params[s] == nullptr) return AA_ERR_INVALID;
    if (d->x_split > 0 && (d->x[s] == nullptr || d->ldx[s] < d->x_split)) return AA_ERR_INVALID;
    if (d->x_split < d->layout.dims[0] &&
        (d->x2[s] == nullptr || d->ldx2[s] < d->layout.dims[0] - d->x_split))
      return AA_ERR_INVALID;
    p.net[g].params = d->params[s];
    p.net[g].x = d->x[s];
    p.net[g].x2 = d->x2[s];
    p.net[g].ldx = d->ldx[s];
    p.net[g].ldx2 = d->ldx2[s];
    for (int l = 0; l < AA_MLP_MAX_LAYERS; ++l) {
      if (l < d->layout.n_layers && d->y[s][l] == nullptr) return AA_ERR_INVALID;
      p.net[g].y[l] = d->y[s][l];
    }
  }
  for (int i = 0; i < 2; ++i) {
    const aa_sac_sample_tail* t = i == 0 ? t_a : t_b;
    MwTail& T = p.tail[i];
    T = MwTail{};
    T.net = -1;
    if (t == nullptr) continue;
    const int L = d->layout.n_layers;
    if (t->net < 0 || t->net >= d->n_nets || t->A < 1 || d->layout.dims[L] != 2 * t->A ||
        d->layout.acts[L - 1] != AA_ACT_NONE || MW_TS * t->A > MW_WAVES * MW_TS * MW_MAXW)
      return AA_ERR_INVALID;
    if (!t->act_mean || !t->act_mag || !t->action || !t->logp) return AA_ERR_INVALID;
    if ((t->save_tanh == nullptr) != (t->save_sigma == nullptr) ||
        (t->save_tanh == nullptr) != (t->save_eps == nullptr))
      return AA_ERR_INVALID;
    if (t->eps_in == nullptr && t->call_counter_dev == nullptr) return AA_ERR_INVALID;
    T.net = t->net; T.A = t->A; T.std_kind = t->std_kind;
    T.act_mean = t->act_mean; T.act_mag = t->act_mag; T.eps_in = t->eps_in;
    T.seed_lo = (uint32_t)t->seed; T.seed_hi = (uint32_t)(t->seed >> 32);
    T.call_counter = t->call_counter_dev; T.arrival = t->arrival_dev;
    T.action = t->action; T.logp = t->logp;
    T.save_tanh = t->save_tanh; T.save_sigma = t->save_sigma;
    T.save_eps = t->save_eps;
  }
  const int64_t gx = (d->B + MW_TS - 1) / MW_TS;
  if (gx > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_mlp_wide_fwd_kernel, dim3((unsigned)gx, (unsigned)d->n_nets),
                     dim3(MW_THREADS), 0, (hipStream_t)stream, p);
  return aa_launch_status();
}

static int mw_backward(const aa_mlp_wide_bwd* d, const aa_sac_dout_gen* gen,
                       const aa_mlp_wide_adam* adam, void* stream, bool dw_only = false);

int aa_mlp_wide_backward(const aa_mlp_wide_bwd* d, void* stream) {
  return mw_backward(d, nullptr, nullptr, stream);
}

int aa_mlp_wide_backward_gen(const aa_mlp_wide_bwd* d, const aa_sac_dout_gen* gen, void* stream) {
  if (gen == nullptr) return AA_ERR_INVALID;
  return mw_backward(d, gen, nullptr, stream);
}

int aa_mlp_wide_backward_gen_adam(const aa_mlp_wide_bwd* d, const aa_sac_dout_gen* gen,
                                  const aa_mlp_wide_adam* adam, void* stream) {
  if (adam == nullptr) return AA_ERR_INVALID;
  return mw_backward(d, gen, adam, stream);
}

int aa_mlp_wide_dw_adam(const aa_mlp_wide_bwd* d, const aa_mlp_wide_adam* adam, void* stream) {
  if (d == nullptr || d->grads[0] == nullptr) return AA_ERR_INVALID;
  return mw_backward(d, nullptr, adam, stream, true);
}

static int mw_backward(const aa_mlp_wide_bwd* d, const aa_sac_dout_gen* gen,
                       const aa_mlp_wide_adam* adam, void* stream, bool dw_only) {
  if (d == nullptr || d->B < 1 || d->n_nets < 1 || d->n_nets > AA_MLPW_MAX_NETS) return AA_ERR_INVALID;
  int rc = mw_check_layout(&d->layout);
  if (rc != AA_OK) return rc;
  const aa_mlp_layout& lay = d->layout;
  const int L = lay.n_layers;
  if (d->x_split < 0 || d->x_split > lay.dims[0]) return AA_ERR_INVALID;
  const bool want_dx = d->dx[0] != nullptr;
  const bool want_dw = d->grads[0] != nullptr;
  if (want_dx && (d->dx_lo < 0 || d->dx_hi > lay.dims[0] || d->dx_lo >= d->dx_hi))
    return AA_ERR_INVALID;
  MwBwdP p;
  p.lay = lay;
  p.B = d->B;
  p.dx_lo = d->dx_lo;
  p.dx_hi = d->dx_hi;
  p.stamps = g_mw_stamps;
  MwDwP q;
  q.lay = lay;
  q.B = d->B;
  q.x_split = d->x_split;
  for (int g = 0; g < AA_MLPW_MAX_NETS; ++g) {
    const int s = g < d->n_nets ? g : 0;
    if (d->params[s] == nullptr) return AA_ERR_INVALID;
    if (gen == nullptr && !dw_only && (d->dout[s] == nullptr || d->ld_dout[s] < lay.dims[L]))
      return AA_ERR_INVALID;
    if ((d->dx[s] != nullptr) != want_dx || (d->grads[s] != nullptr) != want_dw)
      return AA_ERR_INVALID;
    if (want_dx && d->ld_dx[s] < d->dx_hi) return AA_ERR_INVALID;
    p.net[g].params = d->params[s];
    p.net[g].dout = d->dout[s];
    p.net[g].ld_dout = d->ld_dout[s];
    p.net[g].dx = d->dx[s];
    p.net[g].ld_dx = d->ld_dx[s];
    q.net[g].x = d->x[s];
    q.net[g].x2 = d->x2[s];
    q.net[g].ldx = d->ldx[s];
    q.net[g].ldx2 = d->ldx2[s];
    q.net[g].grads = d->grads[s];
    for (int l = 0; l < AA_MLP_MAX_LAYERS; ++l) {
      if (l < L && (d->y[s][l] == nullptr || d->dz[s][l] == nullptr)) return AA_ERR_INVALID;
      p.net[g].y[l] = d->y[s][l];
      p.net[g].dz[l] = d->dz[s][l];
      q.net[g].y[l] = d->y[s][l];
      q.net[g].dz[l] = d->dz[s][l];
    }
    if (want_dw) {
      if (d->x_split > 0 && (d->x[s] == nullptr || d->ldx[s] < d->x_split)) return AA_ERR_INVALID;
      if (d->x_split < lay.dims[0] && (d->x2[s] == nullptr || d->ldx2[s] < lay.dims[0] - d->x_split))
        return AA_ERR_INVALID;
    }
  }
  p.gen = MwGen{};
  if (gen != nullptr) {
    const aa_sac_dout_gen& t = *gen;
    MwGen& G = p.gen;
    if (t.kind == AA_SAC_GEN_CRITIC || t.kind == AA_SAC_GEN_ACTOR) {
      if (d->n_nets != 2 || lay.dims[L] != 1 || !t.q1 || !t.q2 || !t.log_alpha || !t.loss_out ||
          !(t.global_batch > 0.f))
        return AA_ERR_INVALID;
      if (t.kind == AA_SAC_GEN_CRITIC &&
          (!t.tq1 || !t.tq2 || !t.next_logp || !t.reward || !t.discount ||
           (t.loss_kind != AA_LOSS_HUBER && t.loss_kind != AA_LOSS_SQUARED)))
        return AA_ERR_INVALID;
      if (t.kind == AA_SAC_GEN_ACTOR && (!t.logp || !t.dlogp_out)) return AA_ERR_INVALID;
    } else if (t.kind == AA_SAC_GEN_HEAD) {
      if (d->n_nets != 1 || t.A < 1 || lay.dims[L] != 2 * t.A || !t.z || !t.act_mag ||
          !t.save_tanh || !t.save_sigma || !t.save_eps || !t.dlogp)
        return AA_ERR_INVALID;
      if ((t.daction != nullptr && t.ld_daction < t.A) ||
          (t.daction2 != nullptr && (t.ld_daction2 < t.A || t.daction == nullptr)))
        return AA_ERR_INVALID;
    } else {
      return AA_ERR_INVALID;
    }
    G.kind = t.kind;
    G.q1 = t.q1; G.q2 = t.q2; G.tq1 = t.tq1; G.tq2 = t.tq2; G.next_logp = t.next_logp;
    G.reward = t.reward; G.discount = t.discount; G.logp = t.logp; G.weights = t.weights;
    G.log_alpha = t.log_alpha; G.gamma = t.gamma; G.reward_scale = t.reward_scale;
    G.loss_kind = t.loss_kind; G.loss_weight = t.loss_weight; G.global_batch = t.global_batch;
    G.loss_out = t.loss_out; G.td_target_out = t.td_target_out; G.dlogp_out = t.dlogp_out;
    G.z = t.z; G.A = t.A; G.std_kind = t.std_kind; G.act_mag = t.act_mag;
    G.save_tanh = t.save_tanh; G.save_sigma = t.save_sigma; G.save_eps = t.save_eps;
    G.daction = t.daction; G.ld_da = t.ld_daction; G.daction2 = t.daction2;
    G.ld_da2 = t.ld_daction2; G.dlogp = t.dlogp;
  }
  q.adam = MwAdam{};
  if (adam != nullptr) {
    if (!want_dw || !adam->step_dev || !adam->arrival_dev) return AA_ERR_INVALID;
    for (int g = 0; g < d->n_nets; ++g) {
      if (!adam->p[g] || !adam->m[g] || !adam->v[g]) return AA_ERR_INVALID;
      // the parameters this launch steps are the ones the gradient chain in front of it read
      if (adam->p[g] != d->params[g]) return AA_ERR_INVALID;
      q.adam.p[g] = adam->p[g]; q.adam.m[g] = adam->m[g]; q.adam.v[g] = adam->v[g];
      q.adam.target[g] = adam->target[g];
    }
    q.adam.on = 1;
    q.adam.lr = adam->lr; q.adam.beta1 = adam->beta1; q.adam.beta2 = adam->beta2;
    q.adam.eps = adam->eps; q.adam.tau = adam->tau;
    q.adam.step_dev = adam->step_dev; q.adam.arrival = adam->arrival_dev;
  }
  if (d->B > 0x7fffffffLL) return AA_ERR_RANGE;
  const int64_t gx = (d->B + MW_TS - 1) / MW_TS;
  if (gx > 0x7fffffffLL) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  if (!dw_only)      // (dw_only: the chain ran in an earlier call, its dz buffers still hold it)
    hipLaunchKernelGGL(aa_mlp_wide_chain_kernel, dim3((unsigned)gx, (unsigned)d->n_nets),
                       dim3(MW_VT), 0, st, p);
  if (want_dw) {
    int tiles = 0;
    for (int l = 0; l < L; ++l) {
      q.tile_start[l] = tiles;
      tiles += ((lay.dims[l] + 31) / 32) * ((lay.dims[l + 1] + 31) / 32);
    }
    for (int l = L; l <= AA_MLP_MAX_LAYERS; ++l) q.tile_start[l] = tiles;
    hipLaunchKernelGGL(aa_mlp_wide_dw_kernel, dim3((unsigned)tiles, (unsigned)d->n_nets), dim3(256),
                       0, st, q);
  }
  return aa_launch_status();
}

}  // extern "C"
