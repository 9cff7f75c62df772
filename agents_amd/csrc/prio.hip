// Prioritized (proportional) replay sampling -- the "segment-tree sampling" of the north star.
//
// The reference has no prioritized TFUniformReplayBuffer (prioritisation exists only through
// Reverb, tf_agents/examples/dqn/gymnasium/d3qn_train_eval.py:162); what it does provide is the
// plumbing this fills: DqnLossInfo.td_error (agents/dqn/dqn_agent.py:50-72), BufferInfo.ids and
// the Learner's after_train_strategy_step_fn((experience, sample_info), loss_info) hook
// (train/learner.py:362-376).  Semantics: proportional prioritisation (Schaul et al. 2016):
// P(i) = p_i / sum_j p_j over the rows whose id is a valid window start
// (tf_uniform_replay_buffer.py:610-635, the same validity rule as uniform sampling).
//
// Instead of a pointer-chasing sum tree this is the flat two-level scan that suits a GPU:
//   priorities are uint32 fixed point (2^-16 units), so every sum is an exact uint64 and neither
//   the summation order nor the hardware changes which row a random number selects -- sampled
//   indices stay bit-exact against the numpy oracle (oracle/prioritized.py);
//   level 1: one workgroup per block of 1,024 rows sums the (validity-masked) priorities with
//            64-lane shuffle reductions (aa_prio_block_sums);
//   level 2: one wave per sample: exclusive scan of the <= 4,096 block sums to find the block
//            (lanes stride the blocks, wave prefix by shuffles, ballot picks the crossing lane),
//            then the same inside the block's 1,024 rows (aa_prio_sample).
// A get_next over the 1 M-row Atari table reads 12 MB (priorities + ids) -- microseconds of HBM
// time -- and is deterministic.
#include "common.h"
#include "agents_amd.h"

#define AA_PRIO_BLOCK 1024

__device__ static inline unsigned long long aa_wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// inclusive prefix sum across the 64 lanes
__device__ static inline unsigned long long aa_wave_scan_u64(unsigned long long v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

__device__ static inline void aa_valid_range(int64_t last_id, int64_t max_len, int64_t T,
                                             int64_t* min_id, int64_t* max_id) {
  if (last_id < max_len) {
    *min_id = 0;
    int64_t m = last_id + 1 - T + 1;
    *max_id = m < 0 ? 0 : m;
  } else {
    *min_id = last_id + 1 - max_len;
    *max_id = last_id + 1 - T + 1;
  }
}

__device__ static inline unsigned aa_masked_prio(const unsigned* __restrict__ pq,
                                                 const int64_t* __restrict__ ids, int64_t row,
                                                 int64_t capacity, int64_t min_id,
                                                 int64_t max_id) {
  if (row >= capacity) return 0u;
  const int64_t id = ids[row];
  return (id >= min_id && id < max_id) ? pq[row] : 0u;
}

__global__ void __launch_bounds__(256)
aa_prio_block_sums_kernel(const unsigned* __restrict__ pq, const int64_t* __restrict__ ids,
                          const int64_t* __restrict__ last_id_p, int64_t capacity,
                          int64_t max_len, int64_t T, unsigned long long* __restrict__ bsum) {
  __shared__ unsigned long long red[4];
  int64_t min_id, max_id;
  aa_valid_range(*last_id_p, max_len, T, &min_id, &max_id);
  const int64_t base = (int64_t)blockIdx.x * AA_PRIO_BLOCK;
  unsigned long long s = 0;
#pragma unroll
  for (int j = 0; j < AA_PRIO_BLOCK / 256; ++j)
    s += aa_masked_prio(pq, ids, base + j * 256 + threadIdx.x, capacity, min_id, max_id);
  s = aa_wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// One wave per sample.  r = Philox 64-bit word mod total selects the row whose cumulative
// priority interval [c_i, c_i + p_i) contains r (rows in index order).
__global__ void __launch_bounds__(256)
aa_prio_sample_kernel(const unsigned* __restrict__ pq, const int64_t* __restrict__ ids,
                      const int64_t* __restrict__ last_id_p, int64_t capacity, int64_t max_len,
                      int64_t T, const unsigned long long* __restrict__ bsum, int n_blocks,
                      int64_t S, uint32_t k0, uint32_t k1, int64_t* call_dev,
                      int64_t* __restrict__ rows, float* __restrict__ probs,
                      int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t call = (uint64_t)*call_dev;
  if (s >= S) return;
  int64_t min_id, max_id;
  aa_valid_range(*last_id_p, max_len, T, &min_id, &max_id);
  // total priority mass
  unsigned long long tot = 0;
  for (int b = lane; b < n_blocks; b += 64) tot += bsum[b];
  tot = aa_wave_sum_u64(tot);
  if (tot == 0ull) {
    if (lane == 0) {
      if (s == 0 && err != nullptr) *err = 1;
      for (int64_t t = 0; t < T; ++t) rows[s * T + t] = 0;
      if (probs) probs[s] = 0.f;
    }
    return;
  }
  const Philox4 rn = philox4x32_10((uint32_t)s, (uint32_t)((uint64_t)s >> 32), (uint32_t)call,
                                   (uint32_t)(call >> 32), k0, k1);
  unsigned long long r = ((((unsigned long long)rn.y) << 32) | rn.x) % tot;
  // ---- level 1: block ------------------------------------------------------------------------
  int blk = -1;
  for (int b0 = 0; b0 < n_blocks && blk < 0; b0 += 64) {
    const int b = b0 + lane;
    const unsigned long long v = b < n_blocks ? bsum[b] : 0ull;
    const unsigned long long inc = aa_wave_scan_u64(v, lane);
    const unsigned long long chunk = __shfl(inc, 63, 64);
    if (r < chunk) {
      const unsigned long long hit = __ballot(r < inc);      // first lane whose prefix exceeds r
      const int l = __ffsll((long long)hit) - 1;
      blk = b0 + l;
      r -= __shfl(inc - v, l, 64);                           // remainder inside that block
    } else {
      r -= chunk;
    }
  }
  // ---- level 2: row inside the block -----------------------------------------------------------
  const int64_t base = (int64_t)blk * AA_PRIO_BLOCK;
  int64_t row = base;
  unsigned prow = 0;
  bool found = false;
  for (int j0 = 0; j0 < AA_PRIO_BLOCK && !found; j0 += 64) {
    const int64_t rr = base + j0 + lane;
    const unsigned p = aa_masked_prio(pq, ids, rr, capacity, min_id, max_id);
    const unsigned long long inc = aa_wave_scan_u64((unsigned long long)p, lane);
    const unsigned long long chunk = __shfl(inc, 63, 64);
    if (r < chunk) {
      const unsigned long long hit = __ballot(r < inc);
      const int l = __ffsll((long long)hit) - 1;
      row = base + j0 + l;
      prow = (unsigned)__shfl((int)p, l, 64);
      found = true;
    } else {
      r -= chunk;
    }
  }
  if (lane == 0) {
    const int64_t seg = row / max_len;
    const int64_t id = ids[row];
    for (int64_t t = 0; t < T; ++t) rows[s * T + t] = (id + t) % max_len + seg * max_len;
    if (probs) probs[s] = (float)((double)prow / (double)tot);
  }
}

// ---- the whole draw in ONE launch ------------------------------------------------------------
// Workgroup b sums its block of 1,024 rows and publishes {launch tag, sum} as one 8-byte
// agent-scope word (a block sum is < 2^42: 22 bits are left for the tag); the first ceil(S / 4)
// workgroups then wait for every slot, build the exclusive prefix of the block sums in LDS ONCE
// (the two-launch form has each sample's wave scan all of them), find a sample's block by binary
// search and its row with ONE round of loads (a lane takes 16 consecutive rows: lane-local prefix,
// wave scan of the lane totals, ballot) where the old kernel walks sixteen dependent 64-row chunks;
// workgroup 0 advances the Philox call counter and the launch sequence once it has seen every
// slot (every workgroup reads both before it publishes).  Only the tagged
// words cross workgroups: no fence.  Integer sums: the selected rows are those of the two-launch
// form and of oracle/prioritized.py whatever the order.
#define AA_PRIO_TAG_BITS 22
#define AA_PRIO_TAG_MASK ((1ull << AA_PRIO_TAG_BITS) - 1ull)
#define AA_PRIO_DRAW_MAX_BLOCKS 8000      /* (n + 1) x 8 bytes of LDS <= 64 KB: 8.19 M rows */

__global__ void __launch_bounds__(256)
aa_prio_draw_kernel(const unsigned* __restrict__ pq, const int64_t* __restrict__ ids,
                    const int64_t* __restrict__ last_id_p, int64_t capacity, int64_t max_len,
                    int64_t T, unsigned long long* slots, int n_blocks, int64_t S, uint32_t k0,
                    uint32_t k1, int64_t* call_dev, unsigned long long* ctl /* launch sequence */,
                    int64_t* __restrict__ rows, int64_t* __restrict__ start_rows,
                    float* __restrict__ probs, int* __restrict__ err) {
  extern __shared__ unsigned long long pre[];     // [n_blocks + 1] exclusive prefix of block sums
  __shared__ unsigned long long red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // both read BEFORE this workgroup publishes: the last sampling workgroup changes them only
  // after every slot has been seen
  // (+ 1: the zero-filled slots of a new workspace must not look like the first launch's)
  const unsigned long long tag = (ctl[0] + 1ull) & AA_PRIO_TAG_MASK;
  const uint64_t call = (uint64_t)*call_dev;
  int64_t min_id, max_id;
  aa_valid_range(*last_id_p, max_len, T, &min_id, &max_id);
  {
    const int64_t base = (int64_t)blockIdx.x * AA_PRIO_BLOCK;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < AA_PRIO_BLOCK / 256; ++j)
      s += aa_masked_prio(pq, ids, base + j * 256 + tid, capacity, min_id, max_id);
    s = aa_wave_sum_u64(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(&slots[blockIdx.x],
                         ((red[0] + red[1] + red[2] + red[3]) << AA_PRIO_TAG_BITS) | tag,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int64_t want = (S + 3) / 4;
  const unsigned n_samp = want < (int64_t)gridDim.x ? (unsigned)want : gridDim.x;
  if (blockIdx.x >= n_samp) return;
  // ---- every block sum, as soon as its workgroup has published it -----------------------------
  for (int kb = tid; kb < n_blocks; kb += 1024) {
    unsigned long long w[4];
    unsigned spins = 0;
    bool all;
    // every slot of this thread that is not in yet is re-read in ONE round of loads (a round
    // costs a trip to the memory side; re-reading them one after the other cost four)
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = ~tag;
    do {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = kb + 256 * u;
        if (k < n_blocks && (w[u] & AA_PRIO_TAG_MASK) != tag)
          w[u] = __hip_atomic_load(&slots[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      all = true;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (kb + 256 * u < n_blocks && (w[u] & AA_PRIO_TAG_MASK) != tag) all = false;
      // a slot that never carries this launch's tag: a workgroup that cannot run (no such grid
      // is launched); abort loudly instead of hanging the queue
      if (!all && ++spins > (1u << 24)) __builtin_trap();
    } while (!all);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (kb + 256 * u < n_blocks) pre[kb + 256 * u] = w[u] >> AA_PRIO_TAG_BITS;
  }
  __syncthreads();
  // This workgroup has seen every slot: every workgroup has published, hence read the call counter
  // and the launch sequence before -- workgroup 0 may advance them now (no arrival count needed).
  if (blockIdx.x == 0 && tid == 0) {
    *call_dev += 1;
    ctl[0] += 1;
  }
  // ---- exclusive prefix in place: a contiguous run per thread, runs combined across the block --
  {
    const int per = (n_blocks + 255) / 256;
    const int lo = tid * per < n_blocks ? tid * per : n_blocks;
    const int hi = lo + per < n_blocks ? lo + per : n_blocks;
    unsigned long long mine = 0;
    for (int k = lo; k < hi; ++k) mine += pre[k];
    const unsigned long long inc = aa_wave_scan_u64(mine, lane);
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    unsigned long long run = inc - mine;
    for (int w2 = 0; w2 < wave; ++w2) run += red[w2];
    for (int k = lo; k < hi; ++k) {
      const unsigned long long v = pre[k];
      pre[k] = run;
      run += v;
    }
    if (tid == 255) pre[n_blocks] = run;     // thread 255's run ends at the grand total
  }
  __syncthreads();
  const unsigned long long tot = pre[n_blocks];
  for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < S; s += (int64_t)n_samp * 4) {
    if (tot == 0ull) {
      if (lane == 0) {
        if (s == 0 && err != nullptr) *err = 1;
        for (int64_t t = 0; t < T; ++t) rows[s * T + t] = 0;
        if (start_rows) start_rows[s] = 0;
        if (probs) probs[s] = 0.f;
      }
      continue;
    }
    const Philox4 rn = philox4x32_10((uint32_t)s, (uint32_t)((uint64_t)s >> 32), (uint32_t)call,
                                     (uint32_t)(call >> 32), k0, k1);
    unsigned long long r = ((((unsigned long long)rn.y) << 32) | rn.x) % tot;
    // the block b with pre[b] <= r < pre[b + 1] (an empty block has pre[b] == pre[b + 1])
    int lo = 0, hi = n_blocks;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= r) lo = mid; else hi = mid;
    }
    r -= pre[lo];
    const int64_t base = (int64_t)lo * AA_PRIO_BLOCK + 16 * lane;
    unsigned p[16];
    int64_t idv[16];
    if ((int64_t)(lo + 1) * AA_PRIO_BLOCK <= capacity) {
      const uint4* p4 = reinterpret_cast<const uint4*>(pq + base);
      const longlong2* i2 = reinterpret_cast<const longlong2*>(ids + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 t = p4[j];
        p[4 * j] = t.x; p[4 * j + 1] = t.y; p[4 * j + 2] = t.z; p[4 * j + 3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const longlong2 t = i2[j];
        idv[2 * j] = t.x; idv[2 * j + 1] = t.y;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (!(idv[j] >= min_id && idv[j] < max_id)) p[j] = 0u;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int64_t rr = base + j;
        idv[j] = rr < capacity ? ids[rr] : -1;
        p[j] = (rr < capacity && idv[j] >= min_id && idv[j] < max_id) ? pq[rr] : 0u;
      }
    }
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) mine += p[j];
    const unsigned long long inc = aa_wave_scan_u64(mine, lane);
    const unsigned long long hit = __ballot(r < inc);        // first lane whose prefix exceeds r
    const int l = __ffsll((long long)hit) - 1;
    if (lane == l) {
      unsigned long long rl = r - (inc - mine);
      int sj = 15;
      unsigned sp = p[15];
      int64_t sid = idv[15];
      bool done = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (!done) {
          if (rl < (unsigned long long)p[j]) { sj = j; sp = p[j]; sid = idv[j]; done = true; }
          else rl -= p[j];
        }
      }
      const int64_t row = base + sj;
      const int64_t seg = row / max_len;
      for (int64_t t = 0; t < T; ++t) rows[s * T + t] = (sid + t) % max_len + seg * max_len;
      if (start_rows) start_rows[s] = sid % max_len + seg * max_len;
      if (probs) probs[s] = (float)((double)sp / (double)tot);
    }
  }
}

__global__ void aa_prio_bump_kernel(int64_t* c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}

// pq[rows[i]] = quantised priority; *max_pq = max(*max_pq, that)  (integer atomics: exact)
__global__ void __launch_bounds__(256)
aa_prio_set_kernel(const int64_t* __restrict__ rows, const float* __restrict__ prio, int64_t n,
                   float alpha, float eps, int64_t capacity, unsigned* __restrict__ pq,
                   unsigned* __restrict__ max_pq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = rows[i];
  if (row < 0 || row >= capacity) return;
  float p = fabsf(prio[i]) + eps;
  if (alpha != 1.0f) p = powf(p, alpha);
  double q = floor((double)p * 65536.0 + 0.5);
  if (!(q >= 1.0)) q = 1.0;                 // a stored row never has zero mass (also NaN -> 1)
  if (q > 4294967295.0) q = 4294967295.0;
  const unsigned v = (unsigned)q;
  pq[row] = v;
  atomicMax(max_pq, v);
}

// rows just written by add_batch (frame id = *last_id) get the running maximum priority
__global__ void __launch_bounds__(256)
aa_prio_on_add_kernel(const int64_t* __restrict__ last_id_p, int64_t batch, int64_t max_len,
                      const unsigned* __restrict__ max_pq, unsigned* __restrict__ pq) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int64_t id = *last_id_p;
  if (id < 0) return;
  pq[b * max_len + id % max_len] = *max_pq;
}

extern "C" {

int64_t aa_prio_workspace_bytes(int64_t capacity) {
  if (capacity <= 0) return -1;
  return ((capacity + AA_PRIO_BLOCK - 1) / AA_PRIO_BLOCK) * (int64_t)sizeof(unsigned long long);
}

int aa_prio_sample_rows(const uint32_t* prio_q, const int64_t* id_table,
                        const int64_t* last_id_dev, int64_t batch, int64_t max_len, int64_t S,
                        int64_t T, uint64_t seed, int64_t* call_counter_dev, void* workspace,
                        int64_t workspace_bytes, int64_t* rows_out, float* prob_out,
                        int* err_flag_dev, void* stream) {
  if (!prio_q || !id_table || !last_id_dev || !call_counter_dev || !rows_out || !workspace ||
      batch <= 0 || max_len <= 0 || S <= 0 || T <= 0)
    return AA_ERR_INVALID;
  const int64_t capacity = batch * max_len;
  const int64_t nb = (capacity + AA_PRIO_BLOCK - 1) / AA_PRIO_BLOCK;
  if (nb > 0x7fffffffLL || workspace_bytes < nb * (int64_t)sizeof(unsigned long long))
    return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* bsum = (unsigned long long*)workspace;
  hipLaunchKernelGGL(aa_prio_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, prio_q,
                     id_table, last_id_dev, capacity, max_len, T, bsum);
  const int64_t grid = (S + 3) / 4;
  hipLaunchKernelGGL(aa_prio_sample_kernel, dim3((unsigned)grid), dim3(256), 0, st, prio_q,
                     id_table, last_id_dev, capacity, max_len, T, bsum, (int)nb, S,
                     (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev, rows_out, prob_out,
                     err_flag_dev);
  hipLaunchKernelGGL(aa_prio_bump_kernel, dim3(1), dim3(64), 0, st, call_counter_dev);
  return aa_launch_status();
}

int64_t aa_prio_draw_workspace_bytes(int64_t capacity) {
  if (capacity <= 0) return -1;
  const int64_t nb = (capacity + AA_PRIO_BLOCK - 1) / AA_PRIO_BLOCK;
  if (nb > AA_PRIO_DRAW_MAX_BLOCKS) return -1;      // the two-launch form serves larger tables
  return (nb + 2) * (int64_t)sizeof(unsigned long long);
}

int aa_prio_draw_rows(const uint32_t* prio_q, const int64_t* id_table, const int64_t* last_id_dev,
                      int64_t batch, int64_t max_len, int64_t S, int64_t T, uint64_t seed,
                      int64_t* call_counter_dev, void* workspace, int64_t workspace_bytes,
                      int64_t* rows_out, int64_t* start_rows_out, float* prob_out,
                      int* err_flag_dev, void* stream) {
  if (!prio_q || !id_table || !last_id_dev || !call_counter_dev || !rows_out || !workspace ||
      batch <= 0 || max_len <= 0 || S <= 0 || T <= 0)
    return AA_ERR_INVALID;
  const int64_t capacity = batch * max_len;
  const int64_t need = aa_prio_draw_workspace_bytes(capacity);
  if (need < 0 || workspace_bytes < need || ((uintptr_t)workspace & 7) != 0) return AA_ERR_RANGE;
  if (((uintptr_t)prio_q & 15) != 0 || ((uintptr_t)id_table & 15) != 0) return AA_ERR_INVALID;
  const int64_t nb = (capacity + AA_PRIO_BLOCK - 1) / AA_PRIO_BLOCK;
  unsigned long long* slots = (unsigned long long*)workspace;
  hipLaunchKernelGGL(aa_prio_draw_kernel, dim3((unsigned)nb), dim3(256),
                     (size_t)(nb + 1) * sizeof(unsigned long long), (hipStream_t)stream, prio_q,
                     id_table, last_id_dev, capacity, max_len, T, slots, (int)nb, S,
                     (uint32_t)seed, (uint32_t)(seed >> 32), call_counter_dev, slots + nb,
                     rows_out, start_rows_out, prob_out, err_flag_dev);
  return aa_launch_status();
}

int aa_prio_set(const int64_t* rows, const float* priorities, int64_t n, float alpha, float eps,
                int64_t capacity, uint32_t* prio_q, uint32_t* max_prio_q_dev, void* stream) {
  if (!rows || !priorities || !prio_q || !max_prio_q_dev || n < 0 || capacity <= 0)
    return AA_ERR_INVALID;
  if (n == 0) return AA_OK;
  hipLaunchKernelGGL(aa_prio_set_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, rows, priorities, n, alpha, eps, capacity, prio_q,
                     max_prio_q_dev);
  return aa_launch_status();
}

int aa_prio_on_add(const int64_t* last_id_dev, int64_t batch, int64_t max_len,
                   const uint32_t* max_prio_q_dev, uint32_t* prio_q, void* stream) {
  if (!last_id_dev || !max_prio_q_dev || !prio_q || batch <= 0 || max_len <= 0)
    return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_prio_on_add_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, last_id_dev, batch, max_len, max_prio_q_dev, prio_q);
  return aa_launch_status();
}

}  // extern "C"
