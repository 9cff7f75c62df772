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
// recomputes its sample's Philox draw: 10 rounds of integer arithmetic against a row copy) and
// advances the device-resident call counter / last_id itself, so add_batch and get_next are ONE
// launch each.  The launch is one round trip to HBM deep, whatever the number of leaves:
//   * WIDE leaves (rows > AA_RB_SMALL bytes: the Atari observation, 28,224 B) are cut into chunks
//     of `chunk_bytes`; workgroup (row, chunk) has every 16-byte load of its chunk in flight
//     before the first store;
//   * NARROW leaves (step_type, action, reward, discount ...: 4-8 bytes a row) and the per-row
//     bookkeeping (ids, probabilities) are moved one (row, leaf) per THREAD by a handful of
//     workgroups placed first in the grid.  (They used to ride in the chunk-0 workgroup, one leaf
//     after the other: six dependent load->store round trips, ~1 us each, in front of a copy
//     that itself takes ~5 us.)
//   * the counter protocol costs no tail: a workgroup announces itself on the sharded arrival
//     words as soon as it has READ the counter (add_batch: a ticket it looks at after its copy,
//     aa_counter_read_arrive; get_next: a fire-and-forget add the grid's last workgroup sums up,
//     aa_arrivals_finish);
//   * get_next's workgroups are FAT (1,024 lanes, the T <= 4 rows of a sample) and ONE wave
//     computes the draw: see aa_rb_sample_gather_kernel for the measurements behind this.
// Table-side accesses are non-temporal (a 28 GB table is streamed, never re-read soon); the batch
// side stays cacheable (conv1 reads it next).
#include "common.h"
#include "agents_amd.h"

#define AA_MAX_LEAVES 24
#define AA_RB_CHUNK 32768  // compaction kernels: bytes of one row handled per workgroup per leaf
#define AA_RB_SMALL 256    // rows up to this many bytes: one (row, leaf) per thread
#define AA_RB_THREADS 256
#define AA_RB_INFLIGHT 8   // vectors per lane loaded before the first store
// AA_RB_ARRIVAL_STRIDE / AA_RB_ARRIVAL_WORDS: common.h

struct AaLeafSet {
  int n;
  int n_big;                    // leaves [0, n_big) are wide, [n_big, n) narrow (aa_fill_leaves)
  char* table[AA_MAX_LEAVES];   // [capacity, row_bytes]
  char* io[AA_MAX_LEAVES];      // items (scatter source) or out (gather destination)
  int64_t row_bytes[AA_MAX_LEAVES];
};

// blockIdx decoding of the row movers: workgroups [0, small_blocks) hold one thread per
// (row, narrow leaf or bookkeeping slot); workgroup small_blocks + row * n_chunks + chunk moves
// bytes [chunk * chunk_bytes, +chunk_bytes) of every wide leaf of `row`.
struct AaRowGrid {
  int small_blocks;
  int n_chunks;
  int chunk_bytes;
  int per_row;      // narrow leaves + bookkeeping slot (0 or 1)
  int rows_per_wg;  // > 1: medium rows (16-byte granular, <= 4 KiB): a workgroup moves that many
                    // rows as ONE flat list of 16-byte vectors (4,096 SAC rows of 1.5 KB as one
                    // workgroup each took 37 us: workgroups cost ~8 ns apiece)
};

// NT_SRC / NT_DST: the table side of the copy is streamed with non-temporal accesses
template <typename V, bool NT_SRC, bool NT_DST>
__device__ static inline void aa_copy_span(const char* __restrict__ src, char* __restrict__ dst,
                                           int64_t len) {
  // len bytes, multiple of sizeof(V); src/dst aligned to sizeof(V).  All loads of a pass are
  // issued before its stores.
  const int n = (int)(len / (int64_t)sizeof(V));
  const V* s = reinterpret_cast<const V*>(src);
  V* d = reinterpret_cast<V*>(dst);
  const int nt = (int)blockDim.x;
  for (int base = threadIdx.x; base < n; base += AA_RB_INFLIGHT * nt) {
    V v[AA_RB_INFLIGHT];
#pragma unroll
    for (int u = 0; u < AA_RB_INFLIGHT; ++u) {
      const int i = base + u * nt;
      if (i < n) v[u] = NT_SRC ? __builtin_nontemporal_load(s + i) : s[i];
    }
#pragma unroll
    for (int u = 0; u < AA_RB_INFLIGHT; ++u) {
      const int i = base + u * nt;
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
                                                int chunk, int chunk_bytes = AA_RB_CHUNK) {
  const int64_t off = (int64_t)chunk * chunk_bytes;
  if (off >= row_bytes) return;
  int64_t len = row_bytes - off;
  if (len > chunk_bytes) len = chunk_bytes;
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

// `nr` medium rows of `rb` bytes (a multiple of 16, rows and bases 16-byte aligned) moved by the
// whole workgroup as one flat list of vectors: every load of a pass before its stores.
template <bool NT_SRC, bool NT_DST, typename SrcRow, typename DstRow>
__device__ static inline void aa_copy_rows_flat(int nr, int64_t rb, SrcRow src_row, DstRow dst_row) {
  const int nvec = (int)(rb >> 4);
  const int total = nr * nvec;
  const int nt = (int)blockDim.x;
  for (int base = threadIdx.x; base < total; base += AA_RB_INFLIGHT * nt) {
    aa_u32x4 v[AA_RB_INFLIGHT];
    int jj[AA_RB_INFLIGHT], cc[AA_RB_INFLIGHT];
#pragma unroll
    for (int u = 0; u < AA_RB_INFLIGHT; ++u) {
      const int i = base + u * nt;
      jj[u] = i / nvec;
      cc[u] = i - jj[u] * nvec;
      if (i < total) {
        const aa_u32x4* s = reinterpret_cast<const aa_u32x4*>(src_row(jj[u])) + cc[u];
        v[u] = NT_SRC ? __builtin_nontemporal_load(s) : *s;
      }
    }
#pragma unroll
    for (int u = 0; u < AA_RB_INFLIGHT; ++u) {
      const int i = base + u * nt;
      if (i < total) {
        aa_u32x4* d = reinterpret_cast<aa_u32x4*>(dst_row(jj[u])) + cc[u];
        if (NT_DST) __builtin_nontemporal_store(v[u], d);
        else *d = v[u];
      }
    }
  }
}

// Narrow rows (<= AA_RB_SMALL bytes), one per LANE, moved by the wave as a whole: EVERY lane of
// the wave calls this (rb = 0: nothing to move) and the loop runs a wave-uniform number of
// passes, so the loads of all lanes and leaves are one batch of exec-masked instructions in front
// of one wait -- not one load -> wait -> store sequence per distinct (alignment, size) branch,
// which is what per-lane control flow compiles to (four serialised round trips for a Trajectory's
// int32 / int64 / float32 leaves + the id).
template <typename W>
__device__ static inline void aa_copy_items_as(const char* src, char* dst, int n) {
  const W* a = reinterpret_cast<const W*>(src);
  W* b = reinterpret_cast<W*>(dst);
  for (int i = 0; __any(i < n); i += 8) {
    W v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i + u < n) v[u] = a[i + u];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i + u < n) b[i + u] = v[u];
  }
}

__device__ static inline void aa_copy_items(const char* src, char* dst, int rb) {
  const bool words = ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)rb) & 3) == 0);
  aa_copy_items_as<uint32_t>(src, dst, words ? rb >> 2 : 0);
  if (__any(!words)) aa_copy_items_as<uint8_t>(src, dst, words ? 0 : rb);
}

// The narrow leaves' pointers, staged in LDS so that a thread can index them by ITS leaf.
struct AaSmallLeaves {
  char* table[AA_MAX_LEAVES];
  char* io[AA_MAX_LEAVES];
  int rb[AA_MAX_LEAVES];
};

#define AA_STAGE_SMALL(leaves, sm)                                      \
  do {                                                                  \
    if (threadIdx.x == 0) {                                             \
      for (int l_ = (leaves).n_big; l_ < (leaves).n; ++l_) {            \
        (sm).table[l_ - (leaves).n_big] = (leaves).table[l_];           \
        (sm).io[l_ - (leaves).n_big] = (leaves).io[l_];                 \
        (sm).rb[l_ - (leaves).n_big] = (int)(leaves).row_bytes[l_];     \
      }                                                                 \
    }                                                                   \
    __syncthreads();                                                    \
  } while (0)

// ---- a device-resident counter every workgroup reads and the launch itself advances -----------
// Thread 0 reads the counter (acquire: the value is HERE before anything below is issued),
// takes a ticket on its shard of the arrival words, and publishes the value to its workgroup
// through LDS -- no other lane touches the counter.  "Arrived" therefore means "has read", and it
// happens at the START of the workgroup; the ticket is looked at by aa_counter_finish after the
// copy, when it has long returned.  The workgroup holding the last ticket of the last shard
// advances the counter: every workgroup has read it by then.  (Shards: see aa_advance_sharded.)
__device__ static inline int64_t aa_counter_read_arrive(const int64_t* counter, int64_t* arrival,
                                                        unsigned long long* ticket) {
  __shared__ int64_t s_value;
  if (threadIdx.x == 0) {
    s_value = __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (arrival != nullptr) {
      unsigned long long* mine = reinterpret_cast<unsigned long long*>(arrival) +
                                 (blockIdx.x & 7u) * AA_RB_ARRIVAL_STRIDE;
      *ticket = __hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  return s_value;
}

__device__ static inline void aa_counter_finish(int64_t* counter, int64_t* arrival,
                                                unsigned long long ticket, int64_t inc,
                                                unsigned n_groups) {
  if (threadIdx.x != 0) return;
  const unsigned k = blockIdx.x & 7u;
  const unsigned long long in_shard = (n_groups + 7u - k) >> 3;
  if (ticket != in_shard - 1ull) return;
  unsigned long long* a = reinterpret_cast<unsigned long long*>(arrival);
  __hip_atomic_store(a + k * AA_RB_ARRIVAL_STRIDE, 0ull, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long* top = a + 8 * AA_RB_ARRIVAL_STRIDE;
  const unsigned long long n_shards = n_groups < 8u ? n_groups : 8u;
  const unsigned long long p2 =
      __hip_atomic_fetch_add(top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p2 == n_shards - 1ull) {
    __hip_atomic_store(top, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *counter += inc;
  }
}

// The ticket-less variant (get_next): every workgroup ADDS one to its shard right after reading
// the counter and never looks at the result (a returning atomic in wave 0 sits in front of the
// draw; nothing waits for this one).  The LAST workgroup of the grid -- dispatched last, and done
// with its copy microseconds after every other workgroup has started -- sums the eight shards
// when it is finished.  The shards are MONOTONIC (never zeroed); word 8 holds the arrivals already
// accounted for by earlier launches, so "every workgroup of THIS launch has read the counter" is
// sum(shards) - consumed == n_groups; then consumed moves up and the counter advances.  If some
// workgroup has not arrived yet the last one polls: that workgroup needs nothing from this one and
// has been dispatched, or will be as slots free up.
// The poll is bounded (two launches sharing the words on different streams must not hang the
// device; ~4 M polls is seconds, a healthy launch needs one or two).  On a timeout the launch is
// reported through *err and the counter is NOT advanced; `consumed` still moves by n_groups, so
// arrivals that trickle in late are absorbed and the following launches see a consistent count
// again (zeroing the shards, as the first version did, left a residue that made every later
// launch time out as well).
__device__ static inline void aa_arrivals_finish(int64_t* counter, int64_t* arrival, int64_t inc,
                                                 unsigned n_groups, int* err) {
  if (blockIdx.x != n_groups - 1u || threadIdx.x >= 64) return;
  unsigned long long* a = reinterpret_cast<unsigned long long*>(arrival);
  const unsigned lane = threadIdx.x;
  unsigned long long consumed = 0;
  if (lane == 0) consumed = a[8 * AA_RB_ARRIVAL_STRIDE];
  consumed = __shfl(consumed, 0, 64);
  unsigned long long total;
  int polls = 0;
  do {
    unsigned long long mine = 0;
    if (lane < 8)
      mine = __hip_atomic_load(a + lane * AA_RB_ARRIVAL_STRIDE, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    total = mine;
    for (int o = 4; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
    total = __shfl(total, 0, 64);
  } while (total - consumed != (unsigned long long)n_groups && ++polls < (1 << 22));
  if (lane == 0) {
    a[8 * AA_RB_ARRIVAL_STRIDE] = consumed + (unsigned long long)n_groups;
    if (total - consumed == (unsigned long long)n_groups)
      *counter += inc;
    else if (err != nullptr)
      *err = 2;   // arrival timeout: the draw of this call is not trustworthy, counter untouched
  }
}

// ---- add_batch: rows[b] = b*L + (last_id+1) mod L ------------------------------------------
// Round 5: DynamicStepDriver's loop counter (counter[b] += step_type[b] != LAST; *total += the sum;
// drivers/dynamic_step_driver.py:113,170 -- csrc/rollout.hip: aa_count_steps_kernel) rides in the
// add_batch launch of the same loop body as ONE extra workgroup (the grid's last), outside the
// arrival protocol of the copy: one launch and one graph node less per collect step.
struct AaStepCount {
  const int32_t* step_type;   // nullptr: no counting (the plain add_batch)
  int64_t B;
  int32_t* counter;           // nullable
  int64_t* total;
  int64_t* mailbox;           // nullable: host-visible {sequence, total}
};

__device__ static inline void aa_rb_count_steps(const AaStepCount& C) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t b0 = threadIdx.x; b0 < C.B; b0 += 8 * (int64_t)blockDim.x) {
    int st[8], cv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + u * (int64_t)blockDim.x;
      const int64_t bc = b < C.B ? b : C.B - 1;         // clamped: unconditional loads
      st[u] = C.step_type[bc];
      cv[u] = C.counter != nullptr ? C.counter[bc] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + u * (int64_t)blockDim.x;
      if (b < C.B) {
        const int inc = st[u] != 2 ? 1 : 0;
        if (C.counter != nullptr) C.counter[b] = cv[u] + inc;
        s += (float)inc;
      }
    }
  }
  const float t = aa_block_sum(s, red);
  if (threadIdx.x == 0) {
    const int64_t tot = *C.total + (int64_t)(t + 0.5f);
    *C.total = tot;
    if (C.mailbox != nullptr) {
      __hip_atomic_store(&C.mailbox[1], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      const int64_t seq = __hip_atomic_load(&C.mailbox[0], __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_SYSTEM) + 1;
      __hip_atomic_store(&C.mailbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_scatter_kernel(AaLeafSet leaves, AaRowGrid g, int64_t* __restrict__ id_table,
                     int64_t* last_id, int64_t* arrival, int64_t max_len, int64_t batch,
                     AaStepCount cnt) {
  __shared__ AaSmallLeaves sm;
  const unsigned n_groups = gridDim.x - (cnt.step_type != nullptr ? 1u : 0u);
  if (blockIdx.x >= n_groups) {     // the counting workgroup: not part of the copy's protocol
    aa_rb_count_steps(cnt);
    return;
  }
  unsigned long long ticket = 0;
  // last_id itself moves after every group has read it
  const int64_t id = aa_counter_read_arrive(last_id, arrival, &ticket) + 1;
  // tf.math.mod semantics (floor mod); id >= 0 always here.
  const int64_t slot = id % max_len;
  if ((int)blockIdx.x < g.small_blocks) {
    AA_STAGE_SMALL(leaves, sm);
    const int64_t i = (int64_t)blockIdx.x * AA_RB_THREADS + threadIdx.x;
    const int64_t b = i / g.per_row;
    const int j = (int)(i - b * g.per_row);
    const int64_t row = b * max_len + slot;
    const bool live = b < batch;
    const bool book = j == leaves.n - leaves.n_big;  // the bookkeeping slot
    if (live && book) id_table[row] = id;
    const int64_t rb = live && !book ? sm.rb[j] : 0;
    aa_copy_items(rb ? sm.io[j] + b * rb : nullptr, rb ? sm.table[j] + row * rb : nullptr,
                  (int)rb);
  } else {
    const int bid = (int)blockIdx.x - g.small_blocks;
    if (g.rows_per_wg > 1) {
      const int64_t b0 = (int64_t)bid * g.rows_per_wg;
      const int nr = (int)(batch - b0 < g.rows_per_wg ? batch - b0 : g.rows_per_wg);
      for (int l = 0; l < leaves.n_big; ++l) {
        const int64_t rb = leaves.row_bytes[l];
        const char* io = leaves.io[l];
        char* tab = leaves.table[l];
        aa_copy_rows_flat<false, true>(
            nr, rb, [&](int j) { return io + (b0 + j) * rb; },
            [&](int j) { return tab + ((b0 + j) * max_len + slot) * rb; });
      }
    } else {
      const int64_t b = bid / g.n_chunks;
      const int chunk = bid % g.n_chunks;
      const int64_t row = b * max_len + slot;
      for (int l = 0; l < leaves.n_big; ++l) {
        const int64_t rb = leaves.row_bytes[l];
        aa_copy_row_chunk<false, true>(leaves.io[l] + b * rb, leaves.table[l] + row * rb, rb,
                                       chunk, g.chunk_bytes);
      }
    }
  }
  if (arrival != nullptr) aa_counter_finish(last_id, arrival, ticket, 1, n_groups);
}

__global__ void aa_rb_bump_kernel(int64_t* last_id, int64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *last_id += inc;
}

// ---- get_next: gather rows -----------------------------------------------------------------
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_gather_kernel(AaLeafSet leaves, AaRowGrid g, const int64_t* __restrict__ id_table,
                    int64_t* __restrict__ ids_out, const int64_t* __restrict__ rows,
                    int64_t n_rows) {
  __shared__ AaSmallLeaves sm;
  if ((int)blockIdx.x < g.small_blocks) {
    AA_STAGE_SMALL(leaves, sm);
    const int64_t i = (int64_t)blockIdx.x * AA_RB_THREADS + threadIdx.x;
    const int64_t r = i / g.per_row;
    const int j = (int)(i - r * g.per_row);
    const bool live = r < n_rows;
    const int64_t row = live ? rows[r] : 0;
    const char* src = nullptr;
    char* dst = nullptr;
    int64_t rb = 0;
    if (live && j == leaves.n - leaves.n_big) {  // the bookkeeping slot: the row's id
      if (ids_out != nullptr) {
        rb = 8;
        src = reinterpret_cast<const char*>(id_table + row);
        dst = reinterpret_cast<char*>(ids_out + r);
      }
    } else if (live) {
      rb = sm.rb[j];
      src = sm.table[j] + row * rb;
      dst = sm.io[j] + r * rb;
    }
    aa_copy_items(src, dst, (int)rb);
    return;
  }
  const int bid = (int)blockIdx.x - g.small_blocks;
  if (g.rows_per_wg > 1) {
    const int64_t r0 = (int64_t)bid * g.rows_per_wg;
    const int nr = (int)(n_rows - r0 < g.rows_per_wg ? n_rows - r0 : g.rows_per_wg);
    for (int l = 0; l < leaves.n_big; ++l) {
      const int64_t rb = leaves.row_bytes[l];
      const char* tab = leaves.table[l];
      char* io = leaves.io[l];
      aa_copy_rows_flat<true, false>(
          nr, rb, [&](int j) { return tab + rows[r0 + j] * rb; },
          [&](int j) { return io + (r0 + j) * rb; });
    }
    return;
  }
  const int64_t r = bid / g.n_chunks;
  const int chunk = bid % g.n_chunks;
  const int64_t row = rows[r];
  for (int l = 0; l < leaves.n_big; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<true, false>(leaves.table[l] + row * rb, leaves.io[l] + r * rb, rb, chunk,
                                   g.chunk_bytes);
  }
}

// ---- Table.write with explicit rows: table[rows[r]] = values[r] ---------------------------------
__global__ void __launch_bounds__(AA_RB_THREADS)
aa_rb_write_kernel(AaLeafSet leaves, AaRowGrid g, const int64_t* __restrict__ rows,
                   int64_t n_rows) {
  __shared__ AaSmallLeaves sm;
  if ((int)blockIdx.x < g.small_blocks) {
    AA_STAGE_SMALL(leaves, sm);
    const int64_t i = (int64_t)blockIdx.x * AA_RB_THREADS + threadIdx.x;
    const int64_t r = i / g.per_row;
    const int j = (int)(i - r * g.per_row);
    const bool live = r < n_rows;
    const int64_t rb = live ? sm.rb[j] : 0;
    aa_copy_items(live ? sm.io[j] + r * rb : nullptr,
                  live ? sm.table[j] + rows[r] * rb : nullptr, (int)rb);
    return;
  }
  const int bid = (int)blockIdx.x - g.small_blocks;
  const int64_t r = bid / g.n_chunks;
  const int chunk = bid % g.n_chunks;
  const int64_t row = rows[r];
  for (int l = 0; l < leaves.n_big; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    aa_copy_row_chunk<false, true>(leaves.io[l] + r * rb, leaves.table[l] + row * rb, rb, chunk,
                                   g.chunk_bytes);
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
// a mod n, bit-exact, without the ~100-instruction 64-bit division when n < 2^32 (every workgroup
// of a get_next launch sits in this arithmetic before it can issue its first row load):
// a = hi 2^32 + lo  ->  x = (hi mod n) 2^32 + lo < n 2^32 <= 2^64 has the same residue, and its
// quotient q < 2^32 is within one of the float64 quotient (relative error 2^-52 of a number below
// 2^32), so one correction step finishes it.
__device__ static inline uint64_t aa_umod64(uint64_t a, uint64_t n) {
  if ((n >> 32) != 0) return a % n;
  const uint32_t n32 = (uint32_t)n;
  const uint32_t hi = (uint32_t)(a >> 32) % n32;
  const uint64_t x = ((uint64_t)hi << 32) | (uint32_t)a;
  const uint64_t q = (uint64_t)((double)x / (double)n32);
  int64_t r = (int64_t)(x - q * (uint64_t)n32);
  if (r < 0) r += n32;
  else if (r >= (int64_t)n32) r -= n32;
  return (uint64_t)r;
}

// the value of a wave-uniform 64-bit quantity, as the compiler can see it (two SGPRs)
__device__ static inline int64_t aa_uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// a mod n for non-negative a, n > 0: 32-bit arithmetic when both fit
__device__ static inline int64_t aa_mod_nonneg(int64_t a, int64_t n) {
  if ((((uint64_t)a | (uint64_t)n) >> 32) == 0) return (int64_t)((uint32_t)a % (uint32_t)n);
  return a % n;
}

// The same draw on the HOST (stamped launches with S <= AA_RB_DRAWN_MAX: the host mirrors last_id
// and the call number, so it can compute the S start rows itself -- a few microseconds of Philox
// -- and hand them to the kernel BY VALUE in the kernel arguments; the device then starts its
// row loads with no draw and no dependent read in front of them).  Same Philox stream, same
// reductions (a % n in 64-bit integers is what aa_umod64 computes), same ring arithmetic:
// bit-identical rows, ids and probabilities (tests/test_gpu_replay.py).
#define AA_RB_DRAWN_MAX 256
struct AaDrawnRows {
  uint32_t idm[AA_RB_DRAWN_MAX];   // first row of the sample inside its env block: id mod max_len
  uint32_t seg[AA_RB_DRAWN_MAX];   // env block
};
static bool aa_rb_draw_host(int64_t last_id, int64_t batch, int64_t max_len, int64_t S, int64_t T,
                            uint64_t call, uint32_t k0, uint32_t k1, AaDrawnRows* out,
                            float* prob) {
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
  for (int64_t s = 0; s < S; ++s) {
    const Philox4 r = philox4x32_10((uint32_t)s, (uint32_t)((uint64_t)s >> 32), (uint32_t)call,
                                    (uint32_t)(call >> 32), k0, k1);
    const uint64_t a = ((uint64_t)r.y << 32) | r.x;
    const uint64_t c = ((uint64_t)r.w << 32) | r.z;
    const int64_t id = min_id + (int64_t)(a % (uint64_t)num_ids);
    int64_t m = id % max_len;
    if (m < 0) m += max_len;
    out->idm[s] = (uint32_t)m;
    out->seg[s] = (uint32_t)(c % (uint64_t)batch);
  }
  *prob = 1.0f / (float)(num_ids * batch);
  return true;
}

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
  *id = min_id + (int64_t)aa_umod64(a, (uint64_t)num_ids);
  *seg = (int64_t)aa_umod64(c, (uint64_t)batch);
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
  for (int64_t t = 0; t < T; ++t) rows[s * T + t] = aa_mod_nonneg(id + t, max_len) + seg * max_len;
  if (probs) probs[s] = prob;
}

// ---- get_next in ONE launch ---------------------------------------------------------------------
// What the launch costs is dependent round trips, not bytes (tools/gather_probe.hip, 512 Atari
// rows = 14.45 MB in + out, per launch inside a HIP graph): the bare copy 5.0 us = 0.71 of the HBM
// peak; one dependent device read in front of it +3.5 us; a returning atomic +0.8 us; 2,048 thin
// workgroups instead of 512: +13 us; the draw computed by all sixteen waves of a fat workgroup:
// +3 us.  Hence:
//   * workgroups are FAT: 1,024 lanes move K consecutive rows of one sample (the two frames of an
//     Atari transition = 56 KB) -- kernel-argument fetch, counter read and draw once per sample;
//     grid = small_blocks + S * (T / K) * n_chunks;
//   * wave 0 alone reads the two device words (last_id, call counter: ONE round trip, both in
//     flight), draws on the scalar unit and hands the rows to the other fifteen waves through LDS;
//   * the arrival add returns nothing and nobody waits for it (aa_arrivals_finish);
//   * the first wide leaf's descriptor is fetched with the kernel arguments, not after the
//     barrier.
// 13.4 -> 10.5 us per launch in isolation (0.27 -> 0.34 of the HBM peak); what is left is the
// counter read (~2 us), the draw (~1.3 us of dependent integer / float64 arithmetic), launch and
// drain.
// STAMPED launches (call_dev == nullptr, last_id_p == nullptr) take the call number and last_id BY
// VALUE from a host that mirrors them (an eager launch per draw instead of a graph replay): no
// counter read, no arrival protocol; workgroup 0 leaves call + 1 in *counter_out so that the
// device-resident counter stays usable by graph-captured draws.  In the DQN loop: 16.7 us per
// launch against 18.7 replayed (the loop's other kernels share the chip), +0.8 % steps/s.
#define AA_RB_SG_THREADS 1024
#define AA_RB_SG_SMALL 256  // lanes of a narrow-leaf workgroup that carry an item (one wave / SIMD)

// What wave 0 of a workgroup hands to the other fifteen through LDS.  The draw (Philox, two 64-bit
// reductions, the ring arithmetic: ~350 instructions) is computed by ONE wave: computed by all
// sixteen it occupies every SIMD of the CU for ~3 us before the first row load is issued
// (rocprofv3: 12.2 us per launch against 5 us for the bare copy).
struct AaDrawn {
  uint64_t call;
  int64_t last_id;
  int64_t row[4];
};

template <int K>
__global__ void __launch_bounds__(AA_RB_SG_THREADS)
aa_rb_sample_gather_kernel(AaLeafSet leaves, AaRowGrid g, const int64_t* __restrict__ id_table,
                           int64_t* __restrict__ ids_out, float* __restrict__ probs,
                           const int64_t* __restrict__ last_id_p, int64_t last_id_value,
                           int64_t batch, int64_t max_len, int64_t S, int64_t T, uint32_t k0,
                           uint32_t k1, uint64_t call, int64_t* call_dev, int64_t* arrival,
                           int64_t* counter_out, int* __restrict__ err) {
  __shared__ AaSmallLeaves sm;
  __shared__ AaDrawn dr;
  const bool small = (int)blockIdx.x < g.small_blocks;
  // (sample, group of K rows, chunk) of a wide-leaf workgroup; the grid is below 2^31 workgroups
  const uint32_t bid = small ? 0u : blockIdx.x - (uint32_t)g.small_blocks;
  const uint32_t groups = (uint32_t)T / K;
  const uint32_t rg = small ? 0u : bid / (uint32_t)g.n_chunks;
  const int chunk = small ? 0 : (int)(bid - rg * (uint32_t)g.n_chunks);
  const int64_t s_big = rg / groups;
  const int64_t t0 = (int64_t)(rg - (uint32_t)s_big * groups) * K;
  // the first wide leaf's descriptor is fetched with the other kernel arguments, not after the
  // barrier (a dependent scalar fetch in front of the row loads): the empty asm pins it here
  const char* const tab0 = leaves.table[0];
  char* const io0 = leaves.io[0];
  const int64_t rb0 = leaves.row_bytes[0];
  asm volatile("" ::"s"(tab0), "s"(io0), "s"(rb0));
  if (threadIdx.x < 64) {
    // both device-resident words are requested together (ONE round trip); the empty asm makes
    // the values land here before anything below is issued
    int64_t last_id = last_id_value;
    if (last_id_p != nullptr) last_id = *last_id_p;
    if (call_dev != nullptr) {
      int64_t seen = *call_dev;
      asm volatile("" : "+v"(seen), "+v"(last_id)::"memory");
      // wave-uniform from here on: the draw below runs on the scalar unit where it can
      seen = aa_uniform64(seen);
      last_id = aa_uniform64(last_id);
      call += (uint64_t)seen;
      // "this workgroup has read the counter": a fire-and-forget add on its shard of the arrival
      // words (nobody waits for it: see aa_arrivals_finish)
      if (threadIdx.x == 0)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(arrival) +
                                   (blockIdx.x & 7u) * AA_RB_ARRIVAL_STRIDE,
                               1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (small) {
      if (threadIdx.x == 0) {
        dr.call = call;
        dr.last_id = last_id;
        for (int l = leaves.n_big; l < leaves.n; ++l) {
          sm.table[l - leaves.n_big] = leaves.table[l];
          sm.io[l - leaves.n_big] = leaves.io[l];
          sm.rb[l - leaves.n_big] = (int)leaves.row_bytes[l];
        }
      }
    } else {
      int64_t id = 0, seg = 0;
      float prob = 0.f;
      const bool ok = aa_rb_draw(last_id, batch, max_len, T, s_big, call, k0, k1, &id, &seg, &prob);
      if (threadIdx.x < K)
        dr.row[threadIdx.x] =
            ok ? aa_mod_nonneg(id + t0 + threadIdx.x, max_len) + seg * max_len : 0;
    }
  }
  // only this wave's LDS traffic is waited for (__syncthreads() would also drain the arrival add)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (small) {
    const int64_t i = (int64_t)blockIdx.x * AA_RB_SG_SMALL + threadIdx.x;
    const int64_t n_rows = S * T;
    if (threadIdx.x < AA_RB_SG_SMALL) {  // whole waves: the other twelve have nothing to do
      const bool live = i < n_rows * g.per_row;
      const bool narrow = ((n_rows * g.per_row) >> 31) == 0;
      const int64_t r = !live ? 0
                        : narrow ? (int64_t)((uint32_t)i / (uint32_t)g.per_row) : i / g.per_row;
      const int j = (int)(i - r * g.per_row);
      const int64_t s = narrow ? (int64_t)((uint32_t)r / (uint32_t)T) : r / T;
      const int64_t t = r - s * T;
      int64_t id = 0, seg = 0;
      float prob = 0.f;
      const bool ok =
          aa_rb_draw(dr.last_id, batch, max_len, T, s, dr.call, k0, k1, &id, &seg, &prob);
      const int64_t row = ok ? aa_mod_nonneg(id + t, max_len) + seg * max_len : 0;
      const bool book = live && j == leaves.n - leaves.n_big;  // the bookkeeping slot
      const char* src = nullptr;
      char* dst = nullptr;
      int64_t rb = 0;
      if (book) {
        if (ids_out != nullptr) {  // the row's id travels like a leaf
          rb = 8;
          src = reinterpret_cast<const char*>(id_table + row);
          dst = reinterpret_cast<char*>(ids_out + r);
        }
      } else if (live) {
        rb = sm.rb[j];
        src = sm.table[j] + row * rb;
        dst = sm.io[j] + r * rb;
      }
      aa_copy_items(src, dst, (int)rb);
      if (book && t == 0 && probs != nullptr) probs[s] = prob;
      if (book && !ok && r == 0 && err != nullptr) *err = 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && call_dev == nullptr && counter_out != nullptr)
      *counter_out = (int64_t)(dr.call + 1);
  } else {
    int64_t row[K];
#pragma unroll
    for (int t = 0; t < K; ++t) row[t] = dr.row[t];
    const int64_t r0 = s_big * T + t0;
    const int64_t off = (int64_t)chunk * g.chunk_bytes;
    for (int l = 0; l < leaves.n_big; ++l) {
      const int64_t rb = l == 0 ? rb0 : leaves.row_bytes[l];
      if (off >= rb) continue;
      int64_t len = rb - off;
      if (len > g.chunk_bytes) len = g.chunk_bytes;
      const char* tab = (l == 0 ? tab0 : leaves.table[l]) + off;
      char* io = (l == 0 ? io0 : leaves.io[l]) + off;
      if ((((uintptr_t)tab | (uintptr_t)io | (uintptr_t)rb | (uintptr_t)len) & 15) == 0) {
        // every 16-byte load of the K rows' chunks is issued before the first store
        const int n = (int)(len >> 4);
        for (int base = threadIdx.x; base < n; base += 2 * AA_RB_SG_THREADS) {
          aa_u32x4 v[K][2];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            const aa_u32x4* src = reinterpret_cast<const aa_u32x4*>(tab + row[t] * rb);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int i = base + u * AA_RB_SG_THREADS;
              if (i < n) v[t][u] = __builtin_nontemporal_load(src + i);
            }
          }
#pragma unroll
          for (int t = 0; t < K; ++t) {
            aa_u32x4* dst = reinterpret_cast<aa_u32x4*>(io + (r0 + t) * rb);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int i = base + u * AA_RB_SG_THREADS;
              if (i < n) dst[i] = v[t][u];
            }
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < K; ++t)
          aa_copy_row_chunk<true, false>(leaves.table[l] + row[t] * rb,
                                         leaves.io[l] + (r0 + t) * rb, rb, chunk, g.chunk_bytes);
      }
    }
    if (g.small_blocks == 0 && blockIdx.x == 0 && threadIdx.x == 0 && call_dev == nullptr &&
        counter_out != nullptr)
      *counter_out = (int64_t)(call + 1);
  }
  if (call_dev != nullptr) aa_arrivals_finish(call_dev, arrival, 1, gridDim.x, err);
}

// The gather of a draw the HOST made (aa_rb_draw_host): rows arrive in the kernel arguments, so a
// workgroup's first instructions are its row loads -- no Philox, no device word to read, no LDS
// hand-over, no arrival protocol.  Same grid and the same movers as aa_rb_sample_gather_kernel.
template <int K>
__global__ void __launch_bounds__(AA_RB_SG_THREADS)
aa_rb_gather_drawn_kernel(AaLeafSet leaves, AaRowGrid g, const int64_t* __restrict__ id_table,
                          int64_t* __restrict__ ids_out, float* __restrict__ probs,
                          AaDrawnRows dv, float prob, int64_t max_len, int64_t S, int64_t T,
                          int64_t* counter_out, int64_t next_call) {
  const bool small = (int)blockIdx.x < g.small_blocks;
  if (small) {
    const int64_t i = (int64_t)blockIdx.x * AA_RB_SG_SMALL + threadIdx.x;
    const int64_t n_rows = S * T;
    if (threadIdx.x < AA_RB_SG_SMALL) {
      const bool live = i < n_rows * g.per_row;
      const int64_t r = !live ? 0 : (int64_t)((uint32_t)i / (uint32_t)g.per_row);
      const int j = (int)(i - r * g.per_row);
      const int64_t s = (int64_t)((uint32_t)r / (uint32_t)T);
      const int64_t t = r - s * T;
      int64_t m = (int64_t)dv.idm[s] + t;
      if (m >= max_len) m -= max_len;
      const int64_t row = m + (int64_t)dv.seg[s] * max_len;
      const bool book = live && j == leaves.n - leaves.n_big;  // the bookkeeping slot
      const char* src = nullptr;
      char* dst = nullptr;
      int64_t rb = 0;
      if (book) {
        if (ids_out != nullptr) {
          rb = 8;
          src = reinterpret_cast<const char*>(id_table + row);
          dst = reinterpret_cast<char*>(ids_out + r);
        }
      } else if (live) {
        const int l = leaves.n_big + j;
        rb = leaves.row_bytes[l];
        src = leaves.table[l] + row * rb;
        dst = leaves.io[l] + r * rb;
      }
      aa_copy_items(src, dst, (int)rb);
      if (book && t == 0 && probs != nullptr) probs[s] = prob;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && counter_out != nullptr) *counter_out = next_call;
    return;
  }
  const uint32_t bid = blockIdx.x - (uint32_t)g.small_blocks;
  const uint32_t groups = (uint32_t)T / K;
  const uint32_t rg = bid / (uint32_t)g.n_chunks;
  const int chunk = (int)(bid - rg * (uint32_t)g.n_chunks);
  const uint32_t s_big = rg / groups;
  const int64_t t0 = (int64_t)(rg - s_big * groups) * K;
  int64_t row[K];
  {
    const int64_t m0 = (int64_t)dv.idm[s_big] + t0, base = (int64_t)dv.seg[s_big] * max_len;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      int64_t m = m0 + t;
      if (m >= max_len) m -= max_len;
      row[t] = m + base;
    }
  }
  const int64_t r0 = (int64_t)s_big * T + t0;
  const int64_t off = (int64_t)chunk * g.chunk_bytes;
  for (int l = 0; l < leaves.n_big; ++l) {
    const int64_t rb = leaves.row_bytes[l];
    if (off >= rb) continue;
    int64_t len = rb - off;
    if (len > g.chunk_bytes) len = g.chunk_bytes;
    const char* tab = leaves.table[l] + off;
    char* io = leaves.io[l] + off;
    if ((((uintptr_t)tab | (uintptr_t)io | (uintptr_t)rb | (uintptr_t)len) & 15) == 0) {
      // every 16-byte load of the K rows' chunks is issued before the first store
      const int n = (int)(len >> 4);
      for (int base = threadIdx.x; base < n; base += 2 * AA_RB_SG_THREADS) {
        aa_u32x4 v[K][2];
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const aa_u32x4* src = reinterpret_cast<const aa_u32x4*>(tab + row[t] * rb);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = base + u * AA_RB_SG_THREADS;
            if (i < n) v[t][u] = __builtin_nontemporal_load(src + i);
          }
        }
#pragma unroll
        for (int t = 0; t < K; ++t) {
          aa_u32x4* dst = reinterpret_cast<aa_u32x4*>(io + (r0 + t) * rb);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int i = base + u * AA_RB_SG_THREADS;
            if (i < n) dst[i] = v[t][u];
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < K; ++t)
        aa_copy_row_chunk<true, false>(leaves.table[l] + row[t] * rb, leaves.io[l] + (r0 + t) * rb,
                                       rb, chunk, g.chunk_bytes);
    }
  }
  if (g.small_blocks == 0 && blockIdx.x == 0 && threadIdx.x == 0 && counter_out != nullptr)
    *counter_out = next_call;
}

// ---- pseudo-random permutation of [0, n) without a sort -------------------------------------------
// perm[i] = the image of i under a 4-round balanced Feistel network on 2h bits (2^(2h) >= n, h >= 1)
// whose round function is Philox4x32-10(counter = (half, round, call_lo, call_hi), key = seed),
// cycle-walked back into [0, n): a Feistel network is a bijection of [0, 2^(2h)), and following
// the orbit of i until it re-enters [0, n) restricts it to a bijection of [0, n).  Every index is
// computed independently (expected < 4 applications: 2^(2h) < 4 n): no radix sort, no host
// round trip, and the CPU oracle (oracle/perm.py) reproduces it bit for bit.  Replaces the
// tf.data shuffle of train/ppo_learner.py:228-247, whose order the reference does not pin.
// Round 5: the orbit of an index takes a geometric number of applications (2^(2h) is up to ~4 n:
// four on average for n = 2,048 x 129), so with one index per lane a wave ran as long as its
// unluckiest lane -- ~15-20 applications, 60 us for 264 K indices (rocprofv3, PPO configs[2]).
// Now a wave owns a contiguous range of indices and a lane whose orbit has re-entered [0, n) stores
// its result and takes the range's next index (ballot + prefix count): every lane applies the
// network in every round until the range is exhausted.  Same permutation, bit for bit.
__global__ void __launch_bounds__(256)
aa_feistel_perm_kernel(int64_t n, int h, uint32_t k0, uint32_t k1, uint64_t call,
                       int64_t* __restrict__ out) {
  const uint64_t mask = (1ull << h) - 1ull;
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t per = (n + n_waves - 1) / n_waves;
  const int64_t lo = wid * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  int64_t next = lo + 64;          // wave-uniform: the first index nobody has taken yet
  int64_t mine = lo + lane;
  bool active = mine < hi;
  uint64_t x = (uint64_t)mine;
  while (__ballot(active) != 0ull) {
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
    const bool done = active && x < (uint64_t)n;
    if (done) out[mine] = (int64_t)x;
    const unsigned long long dm = __ballot(done);
    if (done) {
      mine = next + __popcll(dm & ((1ull << lane) - 1ull));
      x = (uint64_t)mine;
      active = mine < hi;
    }
    next += __popcll(dm);
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

// Leaves are handed to the kernels wide ones first (the order of the leaves does not matter to a
// copy); *max_rb = the widest row.
static int aa_fill_leaves(AaLeafSet& ls, void* const* tables, void* const* ios,
                          const int64_t* row_bytes, int n, int64_t* max_rb) {
  if (n < 0 || n > AA_MAX_LEAVES) return AA_ERR_RANGE;
  ls.n = 0;
  ls.n_big = 0;
  int64_t m = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < n; ++i) {
      if (pass == 0 && (row_bytes[i] < 0 ||
                        (row_bytes[i] > 0 && (tables[i] == nullptr || ios[i] == nullptr))))
        return AA_ERR_INVALID;
      if (row_bytes[i] == 0) continue;  // nothing to move
      const bool big = row_bytes[i] > AA_RB_SMALL;
      if (big != (pass == 0)) continue;
      ls.table[ls.n] = (char*)tables[i];
      ls.io[ls.n] = (char*)ios[i];
      ls.row_bytes[ls.n] = row_bytes[i];
      ++ls.n;
      if (big) ++ls.n_big;
      if (row_bytes[i] > m) m = row_bytes[i];
    }
  }
  *max_rb = m;
  return AA_OK;
}

// bytes of a wide row per workgroup (a former knob; 4 KiB .. 32 KiB were swept:
// measured on 512 Atari rows: 32 KiB 13.2 us, 16 KiB 17.4, 8 KiB 26.1, 4 KiB 36.9 -- workgroups
// are what costs)
static int aa_rb_chunk_bytes() {
  return 32768;
}

// The grid of a row mover over n_rows rows (see AaRowGrid).  `bookkeeping`: one extra thread per
// row for the ids / probabilities; `small_items`: (row, narrow leaf) items of one narrow-leaf
// workgroup; `rows_per_group`: rows
// a wide-leaf workgroup moves (n_rows is a multiple of it).
static int aa_plan_rows(const AaLeafSet& ls, int64_t n_rows, bool bookkeeping, AaRowGrid* g,
                        int64_t* grid, int small_items = AA_RB_THREADS, int rows_per_group = 1) {
  int64_t max_big = 0;
  for (int l = 0; l < ls.n_big; ++l)
    if (ls.row_bytes[l] > max_big) max_big = ls.row_bytes[l];
  g->chunk_bytes = aa_rb_chunk_bytes();
  g->n_chunks = (int)((max_big + g->chunk_bytes - 1) / g->chunk_bytes);
  g->per_row = (ls.n - ls.n_big) + (bookkeeping ? 1 : 0);
  const int64_t small = (n_rows * g->per_row + small_items - 1) / small_items;
  const int64_t total = small + (n_rows / rows_per_group) * g->n_chunks;
  if (total > 0x7fffffffLL) return AA_ERR_RANGE;
  g->small_blocks = (int)small;
  g->rows_per_wg = 1;
  *grid = total;
  return AA_OK;
}

// Medium rows (every wide leaf 16-byte granular and <= 4 KiB): sixteen rows per workgroup.  Only
// for the movers that implement it (scatter, explicit-rows gather).
static int aa_plan_medium_rows(const AaLeafSet& ls, int64_t n_rows, AaRowGrid* g, int64_t* grid) {
  if (ls.n_big == 0 || g->n_chunks != 1) return AA_OK;
  for (int l = 0; l < ls.n_big; ++l) {
    if (ls.row_bytes[l] > 4096 || (ls.row_bytes[l] & 15) != 0 ||
        (((uintptr_t)ls.table[l] | (uintptr_t)ls.io[l]) & 15) != 0)
      return AA_OK;
  }
  g->rows_per_wg = 16;
  *grid = g->small_blocks + (n_rows + 15) / 16;
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
  auto move = [&](int64_t r, int wd) {
    int l = 0;
    while (wd >= s_start[l + 1]) ++l;
    const int off = wd - s_start[l];
    const uint32_t* src =
        reinterpret_cast<const uint32_t*>(ls.table[l] + rows[r] * ls.row_bytes[l]) + off;
    reinterpret_cast<uint32_t*>(ls.io[l] + r * ls.row_bytes[l])[off] = *src;
  };
  if (total < (1ll << 31)) {
    // 32-bit index arithmetic (round 5: the 64-bit division per word was most of the kernel's
    // instructions -- 51 us for the 10.5 M words of a PPO iteration's gather, rocprofv3)
    const unsigned wpr = (unsigned)words_per_row, tot = (unsigned)total, str = (unsigned)stride;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += str) {
      const unsigned r = i / wpr;
      move((int64_t)r, (int)(i - r * wpr));
      if (i + str < i) break;      // (the next index would wrap)
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / words_per_row;
    move(r, (int)(i - r * words_per_row));
  }
}

extern "C" {

int aa_rb_scatter_rows(void* const* leaf_tables_h, const void* const* leaf_items_h,
                       const int64_t* leaf_row_bytes_h, int n_leaves, int64_t* id_table,
                       int64_t* last_id_dev, int64_t* arrival_dev, int64_t batch, int64_t max_len,
                       void* stream) {
  return aa_rb_scatter_rows_count(leaf_tables_h, leaf_items_h, leaf_row_bytes_h, n_leaves,
                                  id_table, last_id_dev, arrival_dev, batch, max_len, nullptr, 0,
                                  nullptr, nullptr, nullptr, stream);
}

int aa_rb_scatter_rows_count(void* const* leaf_tables_h, const void* const* leaf_items_h,
                             const int64_t* leaf_row_bytes_h, int n_leaves, int64_t* id_table,
                             int64_t* last_id_dev, int64_t* arrival_dev, int64_t batch,
                             int64_t max_len, const int32_t* step_type, int64_t n_envs,
                             int32_t* counter_dev, int64_t* total_dev, int64_t* mailbox,
                             void* stream) {
  if (batch <= 0 || max_len <= 0 || id_table == nullptr || last_id_dev == nullptr)
    return AA_ERR_INVALID;
  AaStepCount cnt = {};
  if (step_type != nullptr) {
    if (total_dev == nullptr || n_envs <= 0 || n_envs > (1 << 24)) return AA_ERR_INVALID;
    cnt.step_type = step_type; cnt.B = n_envs; cnt.counter = counter_dev; cnt.total = total_dev;
    cnt.mailbox = mailbox;
  }
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, leaf_tables_h, (void* const*)leaf_items_h, leaf_row_bytes_h,
                          n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  AaRowGrid g;
  int64_t grid = 0;
  rc = aa_plan_rows(ls, batch, true, &g, &grid);
  if (rc != AA_OK) return rc;
  aa_plan_medium_rows(ls, batch, &g, &grid);
  hipStream_t st = (hipStream_t)stream;
  // last_id advances inside the launch (sharded arrival counters: AA_RB_ARRIVAL_WORDS = 144 zero words)
  int64_t* arrival = arrival_dev;
  hipLaunchKernelGGL(aa_rb_scatter_kernel, dim3((unsigned)grid + (step_type != nullptr ? 1u : 0u)),
                     dim3(AA_RB_THREADS), 0, st, ls, g, id_table, last_id_dev, arrival, max_len,
                     batch, cnt);
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

static inline uint64_t aa_tf_uniform_u64(uint64_t seed, uint64_t seed2, uint64_t block, int half) {
  // PhiloxRandom(seed, seed2) skipped to `block`: counter = (block lo, block hi, seed2 lo, seed2 hi)
  const Philox4 r = philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), (uint32_t)seed2,
                                  (uint32_t)(seed2 >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
  return half == 0 ? (((uint64_t)r.y << 32) | r.x) : (((uint64_t)r.w << 32) | r.z);
}

int aa_rb_draw_tf_host(int64_t last_id, int64_t batch, int64_t max_len, int64_t S, int64_t T,
                       uint64_t seed, uint64_t seed2_ids, uint64_t seed2_seg, uint64_t base_blocks,
                       int64_t* rows_out_h, float* prob_out_h) {
  if (S <= 0 || T <= 0 || batch <= 0 || max_len <= 0 || rows_out_h == nullptr)
    return AA_ERR_INVALID;
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
  if (num_ids <= 0) return AA_ERR_RANGE;
  for (int64_t s = 0; s < S; ++s) {
    const uint64_t blk = base_blocks + (uint64_t)(s >> 1);
    const uint64_t a = aa_tf_uniform_u64(seed, seed2_ids, blk, (int)(s & 1));
    const uint64_t c = aa_tf_uniform_u64(seed, seed2_seg, blk, (int)(s & 1));
    const int64_t id = min_id + (int64_t)(a % (uint64_t)num_ids);
    const int64_t seg = (int64_t)(c % (uint64_t)batch);
    for (int64_t t = 0; t < T; ++t) {
      int64_t m = (id + t) % max_len;
      if (m < 0) m += max_len;
      rows_out_h[s * T + t] = m + seg * max_len;
    }
  }
  if (prob_out_h != nullptr) *prob_out_h = 1.0f / (float)(num_ids * batch);
  return AA_OK;
}

static int aa_rb_sample_gather_launch(const void* const* leaf_tables_h, void* const* leaf_out_h,
                                      const int64_t* leaf_row_bytes_h, int n_leaves,
                                      const int64_t* id_table, int64_t* ids_out, float* prob_out,
                                      const int64_t* last_id_dev, int64_t last_id_value,
                                      int64_t batch, int64_t max_len, int64_t S, int64_t T,
                                      uint64_t seed, uint64_t call_counter,
                                      int64_t* call_counter_dev, int64_t* arrival_dev,
                                      int64_t* counter_out, int* err_flag_dev, void* stream) {
  if (S <= 0 || T <= 0 || batch <= 0 || max_len <= 0) return AA_ERR_INVALID;
  if (call_counter_dev != nullptr && arrival_dev == nullptr) return AA_ERR_INVALID;
  if (ids_out != nullptr && id_table == nullptr) return AA_ERR_INVALID;
  AaLeafSet ls;
  int64_t max_rb = 0;
  int rc = aa_fill_leaves(ls, (void* const*)leaf_tables_h, leaf_out_h, leaf_row_bytes_h,
                          n_leaves, &max_rb);
  if (rc != AA_OK) return rc;
  if (S > 0x7fffffffLL / T) return AA_ERR_RANGE;
  // K rows of a sample per workgroup: as many as keep its bytes in flight <= 64 KiB
  const int chunk_bytes = aa_rb_chunk_bytes();
  const int64_t per_row = max_rb < chunk_bytes ? max_rb : chunk_bytes;
  int K = 1;
  if (T % 4 == 0 && 4 * per_row <= 65536) K = 4;
  else if (T % 2 == 0 && 2 * per_row <= 65536) K = 2;
  AaRowGrid g;
  int64_t grid = 0;
  rc = aa_plan_rows(ls, S * T, true, &g, &grid, AA_RB_SG_SMALL, K);
  if (rc != AA_OK) return rc;
  // stamped launch of a small batch: draw on the host, rows by value (see aa_rb_draw_host)
  if (last_id_dev == nullptr && call_counter_dev == nullptr &&
      S <= AA_RB_DRAWN_MAX && T <= max_len && (S * T * (int64_t)g.per_row >> 31) == 0 &&
      batch * max_len < (1LL << 40)) {
    AaDrawnRows dv;
    float prob = 0.f;
    if (aa_rb_draw_host(last_id_value, batch, max_len, S, T, call_counter, (uint32_t)seed,
                        (uint32_t)(seed >> 32), &dv, &prob)) {
#define AA_GD_LAUNCH(KK)                                                                         \
  hipLaunchKernelGGL(aa_rb_gather_drawn_kernel<KK>, dim3((unsigned)grid),                       \
                     dim3(AA_RB_SG_THREADS), 0, (hipStream_t)stream, ls, g, id_table, ids_out,  \
                     prob_out, dv, prob, max_len, S, T, counter_out, (int64_t)(call_counter + 1))
      if (K == 4) AA_GD_LAUNCH(4);
      else if (K == 2) AA_GD_LAUNCH(2);
      else AA_GD_LAUNCH(1);
#undef AA_GD_LAUNCH
      return aa_launch_status();
    }
    // (no valid id: the device kernel below reports it through *err_flag_dev as before)
  }
#define AA_SG_LAUNCH(KK)                                                                          \
  hipLaunchKernelGGL(aa_rb_sample_gather_kernel<KK>, dim3((unsigned)grid),                       \
                     dim3(AA_RB_SG_THREADS), 0, (hipStream_t)stream, ls, g, id_table, ids_out,   \
                     prob_out, last_id_dev, last_id_value, batch, max_len, S, T, (uint32_t)seed, \
                     (uint32_t)(seed >> 32), call_counter, call_counter_dev, arrival_dev,        \
                     counter_out, err_flag_dev)
  if (K == 4) AA_SG_LAUNCH(4);
  else if (K == 2) AA_SG_LAUNCH(2);
  else AA_SG_LAUNCH(1);
#undef AA_SG_LAUNCH
  return aa_launch_status();
}

int aa_rb_sample_gather(const void* const* leaf_tables_h, void* const* leaf_out_h,
                        const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* id_table,
                        int64_t* ids_out, float* prob_out, const int64_t* last_id_dev,
                        int64_t batch, int64_t max_len, int64_t S, int64_t T, uint64_t seed,
                        uint64_t call_counter, int64_t* call_counter_dev, int64_t* arrival_dev,
                        int* err_flag_dev, void* stream) {
  if (last_id_dev == nullptr) return AA_ERR_INVALID;
  return aa_rb_sample_gather_launch(leaf_tables_h, leaf_out_h, leaf_row_bytes_h, n_leaves,
                                    id_table, ids_out, prob_out, last_id_dev, 0, batch, max_len, S,
                                    T, seed, call_counter, call_counter_dev, arrival_dev, nullptr,
                                    err_flag_dev, stream);
}

int aa_rb_sample_gather_stamped(const void* const* leaf_tables_h, void* const* leaf_out_h,
                                const int64_t* leaf_row_bytes_h, int n_leaves,
                                const int64_t* id_table, int64_t* ids_out, float* prob_out,
                                int64_t last_id, int64_t batch, int64_t max_len, int64_t S,
                                int64_t T, uint64_t seed, uint64_t call_counter,
                                int64_t* call_counter_out_dev, int* err_flag_dev, void* stream) {
  return aa_rb_sample_gather_launch(leaf_tables_h, leaf_out_h, leaf_row_bytes_h, n_leaves,
                                    id_table, ids_out, prob_out, nullptr, last_id, batch, max_len,
                                    S, T, seed, call_counter, nullptr, nullptr,
                                    call_counter_out_dev, err_flag_dev, stream);
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
  AaRowGrid g;
  int64_t grid = 0;
  rc = aa_plan_rows(ls, n_rows, ids_out != nullptr, &g, &grid);
  if (rc != AA_OK) return rc;
  if (grid == 0) return AA_OK;
  aa_plan_medium_rows(ls, n_rows, &g, &grid);
  hipLaunchKernelGGL(aa_rb_gather_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, g, id_table, ids_out, rows, n_rows);
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
  AaRowGrid g;
  int64_t grid = 0;
  rc = aa_plan_rows(ls, n_rows, false, &g, &grid);
  if (rc != AA_OK) return rc;
  if (grid == 0) return AA_OK;
  hipLaunchKernelGGL(aa_rb_write_kernel, dim3((unsigned)grid), dim3(AA_RB_THREADS), 0,
                     (hipStream_t)stream, ls, g, rows, n_rows);
  return aa_launch_status();
}

int aa_random_permutation(int64_t n, uint64_t seed, uint64_t call, int64_t* out, void* stream) {
  if (n <= 0 || out == nullptr || n > (1ll << 60)) return AA_ERR_INVALID;
  int h = 1;
  while ((1ull << (2 * h)) < (uint64_t)n) ++h;
  // one wave per SIMD at most (1,024 of them): several indices per lane, so that lanes whose orbit
  // is short refill from their wave's range instead of idling behind the longest one
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256) blocks = 256;
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
