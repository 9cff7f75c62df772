// Replay ring-buffer kernels: batched row scatter (add_batch), uniform index sampling
// (Philox), row gather (get_next / gather_all).  HBM-bound byte movers.
//
// Replaces, on the MI355X, the TensorFlow primitives the reference leans on:
//   tf_agents/replay_buffers/table.py:112-137   (Table.write  -> scatter_update per leaf)
//   tf_agents/replay_buffers/table.py:86-110    (Table.read   -> sparse_read per leaf)
//   tf_agents/replay_buffers/tf_uniform_replay_buffer.py:182-209 (_add_batch)
//   tf_agents/replay_buffers/tf_uniform_replay_buffer.py:211-310 (_get_next)
//   tf_agents/replay_buffers/tf_uniform_replay_buffer.py:610-635 (_valid_range_ids)
//
// Layout in HBM: one contiguous [capacity, row_bytes] byte table per flattened leaf of the
// data spec, capacity = batch_size * max_length; env b owns rows [b*L, (b+1)*L).  A parallel
// int64 id table [capacity] and one int64 `last_id` word live next to them.
//
// Design: one launch moves ALL leaves -- and, for get_next, also DRAWS the rows (every workgroup
// recomputes its sample's Philox draw: 10 rounds of integer arithmetic against a 28 KB row copy)
// and advances the device-resident call counter / last_id itself, so add_batch and get_next are
// ONE launch each (they were 2 and 2-3: a 256-thread sampling launch, the copy, a 1-thread counter
// bump).  grid = n_rows * n_chunks, a workgroup owns one (row, 32 KiB chunk): an Atari row
// (28,248 B) is one workgroup whose lanes each have up to EIGHT 16-byte loads in flight before the
// first store (2 before: the copies ran at 1.7-1.9 TB/s, bound by bytes in flight per CU, not by
// HBM).  A wave instruction covers 1 KiB of a row.  Table-side accesses are non-temporal (a 28 GB
// table is streamed, never re-read soon); the batch side stays cacheable (conv1 reads it next).
// Narrow leaves (scalars) ride in the chunk-0 workgroup.
#include "common.h"

#define AA_MAX_LEAVES 24
#define AA_RB_CHUNK 32768  // bytes of one row handled per workgroup per leaf
#define AA_RB_THREADS 256
#define AA_RB_INFLIGHT 8   // vectors per lane loaded before the first store
// AA_RB_ARRIVAL_STRIDE / AA_RB_ARRIVAL_WORDS and aa_advance_sharded: common.h

struct AaLeafSet {
  int n;
  char* table[AA_MAX_LEAVES];   // [capacity, row_bytes]
  char* io[AA_MAX_LEAVES];      // items (scatter source) or out (gather destination)
  int64_t row_bytes[AA_MAX_LEAVES];
};

// NT_SRC / NT_DST: the table side of the copy is streamed with non-temporal accesses
template <typename V, bool NT_SRC, bool NT_DST>
__device__ static inline void aa_copy_span(const char* __restrict__ src, char* __restrict__ dst,
                                           int64_t len) {
  // len bytes (<= AA_RB_CHUNK), multiple of sizeof(V); src/dst aligned to sizeof(V).  All loads of
  // a pass are issued before its stores.
  const int n = (int)(len / (int64_t)sizeof(V));
  const V* s = reinterpret_cast<const V*>(src);
  V* d = reinterpret_cast<V*>(dst);
  for (int base = threadIdx.x; base < n; base += AA_RB_INFLIGHT * AA_RB_THREADS) {
    V v[AA_RB_INFLIGHT];
#pragma unroll
    for (int u = 0; u < AA_RB_INFLIGHT; ++u) {
      const int i = base + u * AA_RB_THREADS;
      if (i < n) v[u] = NT_SRC ? __builtin_nontemporal_load(s + i) : s[i];
    }
#pragma unroll
    for (int u = 0; u < AA_RB_INFLIGHT; ++u) {
      const int i = base + u * AA_RB_THREADS;
      if (i < n) {
        if (NT_DST) __builtin_nontemporal_store(v[u], d + i);
        else d[i] = v[u];
      }
    }
  }
}

typedef unsigned int aa_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int aa_u32x2 __attribute__((ext_vector_type(2)));

template <bool NT_SRC, bool NT_DST>
__device__ static inline void aa_copy_row_chunk(const char* src, char* dst, int64_t row_bytes,
                                                int chunk) {
  const int64_t off = (int64_t)chunk * AA_RB_CHUNK;
  if (off >= row_bytes) return;
  int64_t len = row_bytes - off;
  if (len > AA_RB_CHUNK) len = AA_RB_CHUNK;
  src += off;
  dst += off;
  const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)len;
  if ((al & 15) == 0) {
    aa_copy_span<aa_u32x4, NT_SRC, NT_DST>(src, dst, len);
  } else if ((al & 7) == 0) {
    aa_copy_span<aa_u32x2, NT_SRC, NT_DST>(src, dst, len);
  } else if ((al & 3) == 0) {
    aa_copy_span<uint32_t, NT_SRC, NT_DST>(src, dst, len);
  } else {
    aa_copy_span<uint8_t, NT_SRC, NT_DST>(src, dst, len);
  }
}

// ---- add_batch: rows[b] = b*L + (last_id+1) mod L ------------------------------------------
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_scatter_kernel(AaLeafSet leaves, int64_t* __restrict__ id_table, int64_t* last_id,
                     int64_t* arrival, int64_t max_len, int n_chunks) {
  const int64_t b = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x % n_chunks;
  const int64_t id = *last_id + 1;  // last_id itself moves after every group has read it
  // tf.math.mod semantics (floor mod); id >= 0 always here.
  const int64_t row = b * max_len + (id % max_len);
  for (int l = 0; l < leaves.n; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<false, true>(leaves.io[l] + b * rb, leaves.table[l] + row * rb, rb, chunk);
  }
  if (chunk == 0 && threadIdx.x == 0) id_table[row] = id;
  if (arrival != nullptr) aa_advance_sharded(last_id, arrival, 1, gridDim.x);
}

__global__ void aa_rb_bump_kernel(int64_t* last_id, int64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *last_id += inc;
}

// ---- get_next: gather rows -----------------------------------------------------------------
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_gather_kernel(AaLeafSet leaves, const int64_t* __restrict__ id_table,
                    int64_t* __restrict__ ids_out, const int64_t* __restrict__ rows,
                    int n_chunks) {
  const int64_t r = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x % n_chunks;
  const int64_t row = rows[r];
  for (int l = 0; l < leaves.n; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<true, false>(leaves.table[l] + row * rb, leaves.io[l] + r * rb, rb, chunk);
  }
  if (chunk == 0 && threadIdx.x == 0 && ids_out != nullptr) ids_out[r] = id_table[row];
}

// ---- Table.write with explicit rows: table[rows[r]] = values[r] ---------------------------------
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_write_kernel(AaLeafSet leaves, const int64_t* __restrict__ rows, int n_chunks) {
  const int64_t r = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x % n_chunks;
  const int64_t row = rows[r];
  for (int l = 0; l < leaves.n; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<false, true>(leaves.io[l] + r * rb, leaves.table[l] + row * rb, rb, chunk);
  }
}

// ---- tf.data `.unbatch().filter(pred).batch(n)` on device (stream compaction) ----------------------
// The SAC script drops sampled transitions whose first step is an episode boundary and re-batches
// the survivors (agents/sac/examples/v2/train_eval.py:285-296).  keep[s] is the predicate of
// sample s; survivor number k of this batch (k = #keep[0..s)) lands in row (tail + k) mod capacity
// of a ring of pending rows, in source order; the take kernel copies rows head.. out.  The host
// keeps head / tail (it has to learn the survivor count anyway to know when a batch is complete).
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_compact_append_kernel(AaLeafSet leaves, const uint8_t* __restrict__ keep, int64_t n_src,
                            int64_t tail, int64_t capacity, int64_t* __restrict__ kept_out,
                            int n_chunks) {
  __shared__ int part[AA_RB_THREADS / 64];
  const int64_t s = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x % n_chunks;
  int before = 0;
  for (int64_t i = threadIdx.x; i < s; i += AA_RB_THREADS) before += keep[i] != 0;
  for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = before;
  __syncthreads();
  int64_t pos = 0;
  for (int w = 0; w < AA_RB_THREADS / 64; ++w) pos += part[w];
  const bool mine = keep[s] != 0;
  if (chunk == 0 && threadIdx.x == 0 && s == n_src - 1 && kept_out != nullptr)
    *kept_out = pos + (mine ? 1 : 0);
  if (!mine) return;
  const int64_t row = (tail + pos) % capacity;
  for (int l = 0; l < leaves.n; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<false, false>(leaves.io[l] + s * rb, leaves.table[l] + row * rb, rb, chunk);
  }
}

__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_compact_take_kernel(AaLeafSet leaves, int64_t head, int64_t capacity, int n_chunks) {
  const int64_t r = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x % n_chunks;
  const int64_t row = (head + r) % capacity;
  for (int l = 0; l < leaves.n; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<false, false>(leaves.table[l] + row * rb, leaves.io[l] + r * rb, rb, chunk);
  }
}

// ---- uniform sampling of (start id, env block) pairs -----------------------------------------
// Stream definition (canonical for this package; see oracle/replay.py):
//   (x0,x1,x2,x3) = Philox4x32-10(counter = (s_lo, s_hi, call_lo, call_hi), key = (seed_lo, seed_hi))
//   id   = min_id + ((x1<<32 | x0) mod (max_id - min_id))     -- TF-style modulo map, no rejection
//   seg  =           (x3<<32 | x2) mod batch
//   rows[s,t] = (id + t) mod L + seg*L ;  prob = 1 / float32((max_id-min_id)*batch)
// One sample's draw: start id, env block and probability; false when the buffer has no valid id.
__device__ static inline bool aa_rb_draw(int64_t last_id, int64_t batch, int64_t max_len, int64_t T,
                                         int64_t s, uint64_t call, uint32_t k0, uint32_t k1,
                                         int64_t* id, int64_t* seg, float* prob) {
  int64_t min_id, max_id;
  if (last_id < max_len) {
    min_id = 0;
    max_id = last_id + 1 - T + 1;
    if (max_id < 0) max_id = 0;
  } else {
    min_id = last_id + 1 - max_len;
    max_id = last_id + 1 - T + 1;
  }
  const int64_t num_ids = max_id - min_id;
  if (num_ids <= 0) return false;
  const Philox4 r = philox4x32_10((uint32_t)s, (uint32_t)((uint64_t)s >> 32), (uint32_t)call,
                                  (uint32_t)(call >> 32), k0, k1);
  const uint64_t a = ((uint64_t)r.y << 32) | r.x;
  const uint64_t c = ((uint64_t)r.w << 32) | r.z;
  *id = min_id + (int64_t)(a % (uint64_t)num_ids);
  *seg = (int64_t)(c % (uint64_t)batch);
  *prob = 1.0f / (float)(num_ids * batch);
  return true;
}

__global__ void aa_rb_sample_kernel(const int64_t* __restrict__ last_id_p, int64_t batch,
                                    int64_t max_len, int64_t S, int64_t T, uint32_t k0,
                                    uint32_t k1, uint64_t call,
                                    int64_t* call_dev, int bump_in_kernel,
                                    int64_t* __restrict__ rows, float* __restrict__ probs,
                                    int* __restrict__ err) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // graph replays keep the call counter in device memory (a by-value argument would be frozen)
  if (call_dev != nullptr) call += (uint64_t)*call_dev;
  if (bump_in_kernel) {  // single-workgroup launch: every lane has read the counter by now
    __syncthreads();
    if (threadIdx.x == 0) *call_dev += 1;
  }
  if (s >= S) return;
  int64_t id, seg;
  float prob;
  if (!aa_rb_draw(*last_id_p, batch, max_len, T, s, call, k0, k1, &id, &seg, &prob)) {
    if (s == 0 && err != nullptr) *err = 1;
    for (int64_t t = 0; t < T; ++t) rows[s * T + t] = 0;
    if (probs) probs[s] = 0.f;
    return;
  }
  for (int64_t t = 0; t < T; ++t) rows[s * T + t] = (id + t) % max_len + seg * max_len;
  if (probs) probs[s] = prob;
}

// ---- get_next in ONE launch: workgroup (r = s*T + t, chunk) draws sample s itself, copies row
// (id + t) mod L + seg*L of every leaf, and the last workgroup to finish advances the call
// counter (every workgroup has read it by then) ---------------------------------------------------
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_sample_gather_kernel(AaLeafSet leaves, const int64_t* __restrict__ id_table,
                           int64_t* __restrict__ ids_out, float* __restrict__ probs,
                           const int64_t* __restrict__ last_id_p, int64_t batch, int64_t max_len,
                           int64_t T, uint32_t k0, uint32_t k1, uint64_t call,
                           int64_t* call_dev, int64_t* arrival, int* __restrict__ err,
                           int n_chunks) {
  const int64_t r = blockIdx.x / n_chunks;
  const int chunk = blockIdx.x % n_chunks;
  const int64_t s = r / T, t = r - s * T;
  if (call_dev != nullptr) call += (uint64_t)*call_dev;
  int64_t id = 0, seg = 0;
  float prob = 0.f;
  const bool ok = aa_rb_draw(*last_id_p, batch, max_len, T, s, call, k0, k1, &id, &seg, &prob);
  const int64_t row = ok ? (id + t) % max_len + seg * max_len : 0;
  for (int l = 0; l < leaves.n; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<true, false>(leaves.table[l] + row * rb, leaves.io[l] + r * rb, rb, chunk);
  }
  if (chunk == 0 && threadIdx.x == 0) {
    if (ids_out != nullptr) ids_out[r] = id_table[row];
    if (t == 0 && probs != nullptr) probs[s] = prob;
    if (!ok && r == 0 && err != nullptr) *err = 1;
  }
  if (call_dev != nullptr) aa_advance_sharded(call_dev, arrival, 1, gridDim.x);
}

// ---- pseudo-random permutation of [0, n) without a sort -------------------------------------------
// perm[i] = the image of i under a 4-round balanced Feistel network on 2h bits (2^(2h) >= n, h >= 1)
// whose round function is Philox4x32-10(counter = (half, round, call_lo, call_hi), key = seed),
// cycle-walked back into [0, n): a Feistel network is a bijection of [0, 2^(2h)), and following
// the orbit of i until it re-enters [0, n) restricts it to a bijection of [0, n).  Every index is
// computed independently (expected < 4 applications: 2^(2h) < 4 n): no radix sort, no host
// round trip, and the CPU oracle (oracle/perm.py) reproduces it bit for bit.  Replaces the
// tf.data shuffle of train/ppo_learner.py:228-247, whose order the reference does not pin.
__global__ void __launch_bounds__(256)
aa_feistel_perm_kernel(int64_t n, int h, uint32_t k0, uint32_t k1, uint64_t call,
                       int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t mask = (1ull << h) - 1ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t x = (uint64_t)i;
    do {
      uint64_t l = x >> h, r = x & mask;
#pragma unroll
      for (uint32_t round = 0; round < 4; ++round) {
        const Philox4 f = philox4x32_10((uint32_t)r, round, (uint32_t)call, (uint32_t)(call >> 32),
                                        k0, k1);
        const uint64_t t = l ^ ((((uint64_t)f.y << 32) | f.x) & mask);
        l = r;
        r = t;
      }
      x = (l << h) | r;
    } while (x >= (uint64_t)n);
    out[i] = (int64_t)x;
  }
}

// rows[b, i] = (start_id + i) mod L + b*L   (gather_all and deterministic passes)
__global__ void aa_rb_range_rows_kernel(int64_t start_id, int64_t n_ids, int64_t batch,
                                        int64_t max_len, int64_t* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ids * batch) return;
  const int64_t b = i / n_ids, k = i % n_ids;
  rows[i] = (start_id + k) % max_len + b * max_len;
}

static int aa_fill_leaves(AaLeafSet& ls, void* const* tables, void* const* ios,
                          const int64_t* row_bytes, int n, int64_t* max_rb) {
  if (n < 0 || n > AA_MAX_LEAVES) return AA_ERR_RANGE;
  ls.n = n;
  int64_t m = 0;
  for (int i = 0; i < n; ++i) {
    if (row_bytes[i] < 0 || (row_bytes[i] > 0 && (tables[i] == nullptr || ios[i] == nullptr)))
      return AA_ERR_INVALID;
    ls.table[i] = (char*)tables[i];
    ls.io[i] = (char*)ios[i];
    ls.row_bytes[i] = row_bytes[i];
    if (row_bytes[i] > m) m = row_bytes[i];
  }
  *max_rb = m;
  return AA_OK;
}

// Row gather for rows of a few 4-byte words: thread = (row, word); the word's leaf is found by
// walking the (<= AA_MAX_LEAVES) prefix sums held in LDS.
__global__ void __launch_bounds__(256)
aa_rb_gather_small_kernel(AaLeafSet ls, const int64_t* __restrict__ rows, int64_t n_rows,
                          int words_per_row) {
  __shared__ int s_start[AA_MAX_LEAVES + 1];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int l = 0; l < ls.n; ++l) {
      s_start[l] = acc;
      acc += (int)(ls.row_bytes[l] >> 2);
    }
    s_start[ls.n] = acc;
  }
  __syncthreads();
  const int64_t total = n_rows * words_per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / words_per_row;
    const int wd = (int)(i - r * words_per_row);
    int l = 0;
    while (wd >= s_start[l + 1]) ++l;
    const int off = wd - s_start[l];
    const uint32_t* src =
        reinterpret_cast<const uint32_t*>(ls.table[l] + rows[r] * ls.row_bytes[l]) + off;
    reinterpret_cast<uint32_t*>(ls.io[l] + r * ls.row_bytes[l])[off] = *src;
  }
}

extern "C" {

int aa_rb_scatter_rows(void* const* leaf_tables_h, const void* const* leaf_items_h,
                       const int64_t* leaf_row_bytes_h, int n_leaves, int64_t* id_table,
                       int64_t* last_id_dev, int64_t* arrival_dev, int64_t batch, int64_t max_len,
                       void* stream) {
  if (batch <= 0 || max_len <= 0 || id_table == nullptr || last_id_dev == nullptr)
    return AA_ERR_INVALID;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, leaf_tables_h, (void* const*)leaf_items_h, leaf_row_bytes_h,
                          n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  int n_chunks = (int)((max_rb + AA_RB_CHUNK - 1) / AA_RB_CHUNK);
  if (n_chunks < 1) n_chunks = 1;
  const int64_t grid = batch * n_chunks;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  // last_id advances inside the launch (sharded arrival counters: AA_RB_ARRIVAL_WORDS = 144 zero words)
  int64_t* arrival = arrival_dev;
  hipLaunchKernelGGL(aa_rb_scatter_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0, st, ls,
                     id_table, last_id_dev, arrival, max_len, n_chunks);
  if (arrival == nullptr)
    hipLaunchKernelGGL(aa_rb_bump_kernel, dim3(1), dim3(64), 0, st, last_id_dev, (int64_t)1);
  return aa_launch_status();
}

int aa_rb_sample_rows(const int64_t* last_id_dev, int64_t batch, int64_t max_len, int64_t S,
                      int64_t T, uint64_t seed, uint64_t call_counter,
                      int64_t* call_counter_dev, int64_t* rows_out, float* prob_out,
                      int* err_flag_dev, void* stream) {
  if (S <= 0 || T <= 0 || batch <= 0 || max_len <= 0 || rows_out == nullptr ||
      last_id_dev == nullptr)
    return AA_ERR_INVALID;
  const int threads = 256;
  const int64_t grid = (S + threads - 1) / threads;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_sample_kernel, dim3((unsigned)grid), dim3(threads), 0,
                     (hipStream_t)stream, last_id_dev, batch, max_len, S, T, (uint32_t)seed,
                     (uint32_t)(seed >> 32), call_counter, call_counter_dev,
                     (call_counter_dev != nullptr && grid == 1) ? 1 : 0, rows_out, prob_out,
                     err_flag_dev);
  if (call_counter_dev != nullptr && grid != 1)
    hipLaunchKernelGGL(aa_rb_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       call_counter_dev, (int64_t)1);
  return aa_launch_status();
}

int aa_rb_sample_gather(const void* const* leaf_tables_h, void* const* leaf_out_h,
                        const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* id_table,
                        int64_t* ids_out, float* prob_out, const int64_t* last_id_dev,
                        int64_t batch, int64_t max_len, int64_t S, int64_t T, uint64_t seed,
                        uint64_t call_counter, int64_t* call_counter_dev, int64_t* arrival_dev,
                        int* err_flag_dev, void* stream) {
  if (S <= 0 || T <= 0 || batch <= 0 || max_len <= 0 || last_id_dev == nullptr)
    return AA_ERR_INVALID;
  if (call_counter_dev != nullptr && arrival_dev == nullptr) return AA_ERR_INVALID;
  if (ids_out != nullptr && id_table == nullptr) return AA_ERR_INVALID;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, (void* const*)leaf_tables_h, leaf_out_h, leaf_row_bytes_h,
                          n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  int n_chunks = (int)((max_rb + AA_RB_CHUNK - 1) / AA_RB_CHUNK);
  if (n_chunks < 1) n_chunks = 1;
  const int64_t grid = S * T * n_chunks;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_sample_gather_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, id_table, ids_out, prob_out, last_id_dev, batch,
                     max_len, T, (uint32_t)seed, (uint32_t)(seed >> 32), call_counter,
                     call_counter_dev, arrival_dev, err_flag_dev, n_chunks);
  return aa_launch_status();
}

int aa_rb_gather_rows(const void* const* leaf_tables_h, void* const* leaf_out_h,
                      const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* id_table,
                      int64_t* ids_out, const int64_t* rows, int64_t n_rows, void* stream) {
  if (n_rows < 0 || rows == nullptr) return AA_ERR_INVALID;
  if (n_rows == 0) return AA_OK;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, (void* const*)leaf_tables_h, leaf_out_h, leaf_row_bytes_h,
                          n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  if (ids_out != nullptr && id_table == nullptr) return AA_ERR_INVALID;
  // rows of a few words (a PPO minibatch: 4,096 rows x 11 leaves of 4 .. 68 bytes): one thread
  // per 4-byte word instead of one workgroup per (row, 32 KiB chunk) -- 24 us -> a few us
  if (ids_out == nullptr && max_rb <= 512) {
    int64_t words = 0;
    bool ok = true;
    for (int l = 0; l < ls.n; ++l) {
      ok = ok && (ls.row_bytes[l] & 3) == 0 && ((uintptr_t)ls.table[l] & 3) == 0 &&
           ((uintptr_t)ls.io[l] & 3) == 0;
      words += ls.row_bytes[l] >> 2;
    }
    if (ok && words > 0 && words <= 0x7fffffffLL) {
      int64_t blocks = (n_rows * words + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(aa_rb_gather_small_kernel, dim3((unsigned)blocks), dim3(256), 0,
                         (hipStream_t)stream, ls, rows, n_rows, (int)words);
      return aa_launch_status();
    }
  }
  int n_chunks = (int)((max_rb + AA_RB_CHUNK - 1) / AA_RB_CHUNK);
  if (n_chunks < 1) n_chunks = 1;
  const int64_t grid = n_rows * n_chunks;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_gather_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, id_table, ids_out, rows, n_chunks);
  return aa_launch_status();
}

int aa_rb_compact_append(void* const* pending_h, const void* const* src_h,
                         const int64_t* leaf_row_bytes_h, int n_leaves, const uint8_t* keep,
                         int64_t n_src, int64_t tail, int64_t count, int64_t capacity,
                         int64_t* kept_out_dev, void* stream) {
  if (n_src < 0 || capacity <= 0 || tail < 0 || tail >= capacity || count < 0) return AA_ERR_INVALID;
  if (count + n_src > capacity) return AA_ERR_RANGE;  // every source row may survive
  if (n_src == 0) return AA_OK;
  if (keep == nullptr) return AA_ERR_INVALID;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, pending_h, (void* const*)src_h, leaf_row_bytes_h, n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  int n_chunks = (int)((max_rb + AA_RB_CHUNK - 1) / AA_RB_CHUNK);
  if (n_chunks < 1) n_chunks = 1;
  const int64_t grid = n_src * n_chunks;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_compact_append_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, keep, n_src, tail, capacity, kept_out_dev, n_chunks);
  return aa_launch_status();
}

int aa_rb_compact_take(const void* const* pending_h, void* const* out_h,
                       const int64_t* leaf_row_bytes_h, int n_leaves, int64_t head, int64_t n_rows,
                       int64_t count, int64_t capacity, void* stream) {
  if (n_rows < 0 || capacity <= 0 || head < 0 || head >= capacity) return AA_ERR_INVALID;
  if (n_rows > count || count > capacity) return AA_ERR_RANGE;
  if (n_rows == 0) return AA_OK;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, (void* const*)pending_h, out_h, leaf_row_bytes_h, n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  int n_chunks = (int)((max_rb + AA_RB_CHUNK - 1) / AA_RB_CHUNK);
  if (n_chunks < 1) n_chunks = 1;
  const int64_t grid = n_rows * n_chunks;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_compact_take_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, head, capacity, n_chunks);
  return aa_launch_status();
}

int aa_rb_write_rows(void* const* leaf_tables_h, const void* const* leaf_values_h,
                     const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* rows,
                     int64_t n_rows, void* stream) {
  if (n_rows < 0 || rows == nullptr) return AA_ERR_INVALID;
  if (n_rows == 0) return AA_OK;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, leaf_tables_h, (void* const*)leaf_values_h, leaf_row_bytes_h,
                          n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  int n_chunks = (int)((max_rb + AA_RB_CHUNK - 1) / AA_RB_CHUNK);
  if (n_chunks < 1) n_chunks = 1;
  const int64_t grid = n_rows * n_chunks;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_write_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, rows, n_chunks);
  return aa_launch_status();
}

int aa_random_permutation(int64_t n, uint64_t seed, uint64_t call, int64_t* out, void* stream) {
  if (n <= 0 || out == nullptr || n > (1ll << 60)) return AA_ERR_INVALID;
  int h = 1;
  while ((1ull << (2 * h)) < (uint64_t)n) ++h;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aa_feistel_perm_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, n, h, (uint32_t)seed, (uint32_t)(seed >> 32), call, out);
  return aa_launch_status();
}

int aa_rb_range_rows(int64_t start_id, int64_t n_ids, int64_t batch, int64_t max_len,
                     int64_t* rows_out, void* stream) {
  if (n_ids < 0 || batch <= 0 || max_len <= 0 || rows_out == nullptr) return AA_ERR_INVALID;
  const int64_t n = n_ids * batch;
  if (n == 0) return AA_OK;
  const int threads = 256;
  const int64_t grid = (n + threads - 1) / threads;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipLaunchKernelGGL(aa_rb_range_rows_kernel, dim3((unsigned)grid), dim3(threads), 0,
                     (hipStream_t)stream, start_id, n_ids, batch, max_len, rows_out);
  return aa_launch_status();
}

}  // extern "C"
