/* agents_amd.h -- C ABI of libagents_amd.so: the MI355X (gfx950) kernels behind the TF-Agents
 * trainer hot path (TFUniformReplayBuffer add/sample, DynamicStepDriver rollout, DqnAgent /
 * PPOAgent loss + update, Learner gradient step).
 *
 * The reference (tensorflow/agents) has NO native code and no FFI of its own: every entry below
 * replaces a TensorFlow / tf-keras / TFP primitive the reference calls from Python.  Each
 * declaration cites that call site (paths relative to the reference root).  The Python side
 * (agents_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the binding a TF-Agents
 * maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative errno-style code
 * (AA_ERR_INVALID = -22 bad argument, AA_ERR_RANGE = -34 size/workspace out of range,
 * AA_ERR_LAUNCH = -5 HIP launch failure).  All pointers are DEVICE pointers unless the name
 * ends in `_h` (host array).  `stream` is a hipStream_t passed as void*.  Nothing synchronises
 * the device.  No function allocates device memory; scratch is caller-provided.
 */
#ifndef AGENTS_AMD_H_
#define AGENTS_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AA_ABI_VERSION 21
int aa_abi_version(void);

/* ---- activations (epilogues / derivative masks) ---------------------------------------- */
#define AA_ACT_NONE 0
#define AA_ACT_RELU 1
#define AA_ACT_TANH 2

/* =========================================================================================
 * Replay ring buffer   (tf_agents/replay_buffers/tf_uniform_replay_buffer.py, table.py)
 * ========================================================================================= */

/* add_batch: id = *last_id + 1; rows[b] = b*max_len + id mod max_len; for every leaf copy
 * items[b] -> table[rows[b]]; id_table[rows[b]] = id; then *last_id = id.
 * Replaces Table.write / tf.compat.v1.scatter_update per leaf (table.py:112-137) driven by
 * TFUniformReplayBuffer._add_batch (tf_uniform_replay_buffer.py:182-209, 582-607). */
int aa_rb_scatter_rows(void* const* leaf_tables_h, const void* const* leaf_items_h,
                       const int64_t* leaf_row_bytes_h, int n_leaves, int64_t* id_table,
                       int64_t* last_id_dev, int64_t* arrival_dev /* nullable, see below */,
                       int64_t batch, int64_t max_len, void* stream);
/* aa_rb_scatter_rows whose launch also runs DynamicStepDriver's loop counter on the step types of
 * the time step the body just produced (aa_count_steps: counter[b] += step_type[b] != LAST, *total
 * += the sum, posted to `mailbox`; drivers/dynamic_step_driver.py:113,170) as one extra workgroup.
 * step_type == NULL: exactly aa_rb_scatter_rows. */
int aa_rb_scatter_rows_count(void* const* leaf_tables_h, const void* const* leaf_items_h,
                             const int64_t* leaf_row_bytes_h, int n_leaves, int64_t* id_table,
                             int64_t* last_id_dev, int64_t* arrival_dev, int64_t batch,
                             int64_t max_len, const int32_t* step_type, int64_t n_envs,
                             int32_t* counter_dev, int64_t* total_dev, int64_t* mailbox,
                             void* stream);
/* `arrival_dev`: zero before the call and left zero; the kernel counts finished workgroups in it
 * so that the LAST one advances the counter every group has read (last_id, Philox call counter,
 * env step counter) -- no second one-thread launch.
 *   aa_eps_greedy_action: ONE int64 word, used for launches of at most 16 workgroups (NULL or
 *     larger: a one-thread bump launch follows);
 *   aa_rb_scatter_rows / aa_rb_sample_gather / aa_vecenv_random_step (NULL = the step counter is
 *     left alone): 144 int64 words, 128-byte aligned = nine counters
 *     on nine cache lines (arrivals sharded over eight by workgroup index, their last arrivers
 *     meet on the ninth: thousands of device-scope atomics on one word -- or one line --
 *     serialise); NULL for the scatter = a one-thread bump launch from the entry point. */

/* get_next index sampling: S independent (start id, env block) pairs from the Philox4x32-10
 * stream (counter = (s, call_counter), key = seed), mapped with _valid_range_ids
 * (tf_uniform_replay_buffer.py:610-635) and rows[s,t] = (id+t) mod L + block*L (:265-292);
 * prob_out[s] = 1/((max-min)*batch) (:255-264).  Sets *err_flag_dev = 1 if the buffer is empty
 * (the reference's assert_greater at :246-253).  Replaces tf.random.uniform(int64) x2. */
int aa_rb_sample_rows(const int64_t* last_id_dev, int64_t batch, int64_t max_len, int64_t S,
                      int64_t T, uint64_t seed, uint64_t call_counter,
                      int64_t* call_counter_dev /* nullable: *call_counter_dev is added to
                      call_counter and then incremented by one on the device, so a captured HIP
                      graph advances the stream without new kernel arguments */,
                      int64_t* rows_out, float* prob_out, int* err_flag_dev, void* stream);

/* The same draw in TensorFlow's stream LAYOUT, made on the HOST (rng="tf" of TFUniformReplayBuffer;
 * SURVEY.md Appendix B; NOT verified against a TensorFlow run -- no TF here -- and labelled so
 * wherever it is used).  What tf.random.uniform(shape=[S], minval, maxval, dtype=int64) is known
 * to do (tensorflow/core: GuardedPhiloxRandom::Init(seed, seed2), ReserveRandomOutputs(n, 256),
 * FillPhiloxRandom with UniformDistribution<PhiloxRandom, int64>):
 *   key = (seed lo32, seed hi32); counter words 2, 3 = (seed2 lo32, seed2 hi32); counter words 0, 1
 *   = a 64-bit block index that starts at 0 and advances by n x 256 per op execution;
 *   output j of an execution takes Philox block (base + j / 2), words 2 (j % 2) and 2 (j % 2) + 1
 *   as the low and high half of a uint64 x, and returns minval + x mod (maxval - minval).
 * The replay buffer's two draws are two ops (tf_uniform_replay_buffer.py:265-272): ids from the
 * op seeded (seed, seed2_ids), env blocks from (seed, seed2_seg); base_blocks = blocks both ops
 * have consumed so far.  rows_out_h[S*T] (HOST memory) as aa_rb_sample_rows; *prob_out_h as there.
 * Returns AA_ERR_RANGE when the valid id range is empty. */
int aa_rb_draw_tf_host(int64_t last_id, int64_t batch, int64_t max_len, int64_t S, int64_t T,
                       uint64_t seed, uint64_t seed2_ids, uint64_t seed2_seg, uint64_t base_blocks,
                       int64_t* rows_out_h, float* prob_out_h);

/* get_next in ONE launch: the draw of aa_rb_sample_rows (same Philox stream, same mapping, bit for
 * bit) is recomputed by every workgroup of sample s, which then copies row (id+t) mod L + block*L
 * of every leaf into out[s*T + t]; ids_out[s*T+t] = id_table[row] (nullable), prob_out[s]
 * (nullable); the last workgroup advances *call_counter_dev (nullable: then `call_counter` alone
 * numbers the call).  `arrival_dev`: 144 int64 words, zero before the FIRST call, owned by the
 * kernel afterwards (eight monotonic arrival shards on their own cache lines + the count already
 * accounted for; they are not zero between calls).  *err_flag_dev: 1 = a draw found no valid range,
 * 2 = the arrival poll timed out (the call counter is then left untouched).  Replaces aa_rb_sample_rows + aa_rb_gather_rows (+ the counter bump) on the
 * get_next path (tf_uniform_replay_buffer.py:211-310, table.py:86-110). */
int aa_rb_sample_gather(const void* const* leaf_tables_h, void* const* leaf_out_h,
                        const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* id_table,
                        int64_t* ids_out, float* prob_out, const int64_t* last_id_dev,
                        int64_t batch, int64_t max_len, int64_t S, int64_t T, uint64_t seed,
                        uint64_t call_counter, int64_t* call_counter_dev, int64_t* arrival_dev,
                        int* err_flag_dev, void* stream);

/* The same draw + gather launched EAGERLY once per draw by a host that mirrors both counters (the
 * replay buffer's last_id and the Philox call counter): they arrive by value, so no workgroup
 * waits for a device read before it can compute its rows and no arrival protocol runs -- the
 * dependent round trip of aa_rb_sample_gather's counter read is ~3 us of a launch whose copy takes
 * 5.  *call_counter_out_dev (nullable) receives call_counter + 1, which keeps the device-resident
 * counter in step for graph-captured draws.  Same stream of samples, bit for bit
 * (tf_uniform_replay_buffer.py:211-310). */
int aa_rb_sample_gather_stamped(const void* const* leaf_tables_h, void* const* leaf_out_h,
                                const int64_t* leaf_row_bytes_h, int n_leaves,
                                const int64_t* id_table, int64_t* ids_out, float* prob_out,
                                int64_t last_id, int64_t batch, int64_t max_len, int64_t S,
                                int64_t T, uint64_t seed, uint64_t call_counter,
                                int64_t* call_counter_out_dev, int* err_flag_dev, void* stream);

/* Row gather of every leaf + the id table: out[r] = table[rows[r]].
 * Replaces Table.read / ResourceVariable.sparse_read per leaf (table.py:86-110). */
int aa_rb_gather_rows(const void* const* leaf_tables_h, void* const* leaf_out_h,
                      const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* id_table,
                      int64_t* ids_out, const int64_t* rows, int64_t n_rows, void* stream);

/* out[i], i in [0, n): a pseudo-random permutation of [0, n) computed index by index (4-round
 * Feistel network keyed by Philox4x32-10(seed, call), cycle-walked into range; oracle/perm.py) --
 * the per-epoch shuffle of PPOLearner's minibatches (train/ppo_learner.py:228-247: tf.data
 * shuffle, order unpinned by the reference) without a device sort. */
int aa_random_permutation(int64_t n, uint64_t seed, uint64_t call, int64_t* out, void* stream);

/* Device-side `.unbatch().filter(pred).batch(n)` of a sampled batch (the SAC script's dataset
 * pipeline, agents/sac/examples/v2/train_eval.py:285-296; tf.data unbatch / filter / batch).
 * append: for the n_src rows of a batch, row s with keep[s] != 0 is copied, in source order, to row
 *   (tail + #keep[0..s)) mod capacity of the pending ring of every leaf; *kept_out_dev = number of
 *   survivors.  `count` = rows pending before the call; AA_ERR_RANGE when count + n_src > capacity.
 * take: out[r] = pending[(head + r) mod capacity], r < n_rows <= count.
 * The caller owns head / tail / count (replay_buffers/dataset.py). */
int aa_rb_compact_append(void* const* pending_h, const void* const* src_h,
                         const int64_t* leaf_row_bytes_h, int n_leaves, const uint8_t* keep,
                         int64_t n_src, int64_t tail, int64_t count, int64_t capacity,
                         int64_t* kept_out_dev, void* stream);
int aa_rb_compact_take(const void* const* pending_h, void* const* out_h,
                       const int64_t* leaf_row_bytes_h, int n_leaves, int64_t head, int64_t n_rows,
                       int64_t count, int64_t capacity, void* stream);

/* Table.write with explicit rows: table[rows[r]] = values[r] for every leaf (table.py:112-137).
 * Rows must be distinct. */
int aa_rb_write_rows(void* const* leaf_tables_h, const void* const* leaf_values_h,
                     const int64_t* leaf_row_bytes_h, int n_leaves, const int64_t* rows,
                     int64_t n_rows, void* stream);

/* rows[b, i] = (start_id + i) mod L + b*L for gather_all (tf_uniform_replay_buffer.py:533-557). */
int aa_rb_range_rows(int64_t start_id, int64_t n_ids, int64_t batch, int64_t max_len,
                     int64_t* rows_out, void* stream);

/* *counter += inc (device-side step / id counters; tf.Variable.assign_add). */
int aa_counter_add(int64_t* counter_dev, int64_t inc, void* stream);

/* ---- HIP-runtime guard (no reference counterpart; csrc/runtime_guard.hip) --------------------
 * hipGraphLaunch of this image's runtime (version 70051831) faults when two of a graph exec's
 * internal parallel streams share the launch stream's hardware queue.  The graph wrappers
 * (agents_amd/utils/graph.py) instantiate every recorded graph until its parallel streams are
 * spread over different queues, and launch the exec that results themselves:
 *   aa_hip_graph_exec_spread    *n_streams_out = the exec's stream count (1 = linear graph),
 *                               *max_on_one_queue_out = the largest number of its parallel
 *                               streams on one queue (<= 1 is safe for every launch stream);
 *                               AA_ERR_UNSUPPORTED on any other runtime version / object layout
 *   aa_hip_graph_instantiate    hipGraphInstantiate of `graph` (a hipGraph_t), repeated -- the
 *                               exec destroyed, one ballast stream added -- while guard != 0 and
 *                               two parallel streams share a queue; on another runtime version
 *                               one plain instantiation (*n_streams_out = -1).  AA_ERR_RANGE
 *                               after 256 attempts
 *   aa_hip_graph_launch         hipGraphLaunch(exec, stream)
 *   aa_hip_graph_exec_destroy   hipGraphExecDestroy (the caller makes sure no launch is in
 *                               flight) */
#define AA_ERR_UNSUPPORTED (-95)
int aa_hip_graph_exec_spread(void* graph_exec, int32_t* n_streams_out,
                             int32_t* max_on_one_queue_out);
int aa_hip_graph_instantiate(void* graph, int32_t guard, void** exec_out, int32_t* n_streams_out,
                             int32_t* max_on_one_queue_out, int32_t* attempts_out);
int aa_hip_graph_launch(void* graph_exec, void* stream);
int aa_hip_graph_exec_destroy(void* graph_exec);

/* A one-thread no-op dispatch named aa_marker_kernel: measurement aid, not part of any reference
 * path.  bench.py brackets its timed region (and every isolated kernel case) with markers and
 * reads rocprofv3's kernel trace by position between them -- the in-loop kernel durations of the
 * benchmark protocol tf_agents/benchmark/utils.py:89-180 has no access to. */
int aa_marker(int32_t id, void* stream);

/* =========================================================================================
 * fp32 MFMA GEMM with dense / conv-patch operand loaders
 *   (keras Dense / Conv2D forward + tf.GradientTape backward:
 *    agents/dqn/dqn_agent.py:412-449, examples/dqn/mnih15/dqn_train_eval_atari.py:80-112)
 * ========================================================================================= */
#define AA_A_ROW 0        /* A(m,k) = A[m*lda + k]                                  */
#define AA_A_COL 1        /* A(m,k) = A[k*lda + m]        (X^T for weight grads)    */
#define AA_A_PATCH 2      /* A(pixel,k): f32 NHWC conv patches (forward)            */
#define AA_A_PATCH_U8 3   /* same, uint8 input, value = (float)u8 / a_div           */
#define AA_A_PATCH_T 4    /* A(k,pixel): transposed patches (conv weight gradient)  */
#define AA_A_PATCH_T_U8 5
#define AA_B_ROW 0        /* B(k,n) = B[k*ldb + n]                                  */
#define AA_B_COL 1        /* B(k,n) = B[n*ldb + k]        (W^T for input grads)     */

typedef struct aa_gemm_desc {
  const void* A;
  const float* B;
  float* C;               /* C[m*ldc + n] */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t a_mode, b_mode;
  /* conv geometry for AA_A_PATCH*: NHWC input [n_img,H,W,Cin], VALID padding */
  int32_t n_img, H, W, Cin, KH, KW, stride;
  int32_t img_pitch;      /* elements between consecutive images; 0 = H*W*Cin (dense) */
  float a_div;            /* uint8 inputs: divisor (255 for the Atari Lambda(x/255) layer) */
  /* epilogue: C = act(acc + bias[n]) * actgrad_{mask_kind}(mask_src[m*ldm + n]) */
  const float* bias;      /* nullable */
  int32_t act;
  const float* mask_src;  /* nullable: forward OUTPUT of the layer whose activation is undone */
  int32_t ldm;
  int32_t mask_kind;
  int32_t force_cfg;      /* 0 = auto; 1..8 = fp32 MFMA tiles 128x64, 128x32, 64x64, 128x128, 64x32,
                           * 32x64, 32x32, 256x32 (LDS-DMA operands only); 9 = the bf16 matrix-core
                           * plans with exact 3-piece splits: uint8 conv forward / weight gradient
                           * with 32 filters (the automatic choice there, csrc/conv_u8_bf16.h);
                           * AA_ERR_INVALID when the shape is not eligible */
  int32_t force_splits;   /* 0 = auto split-K */
  /* nullable, AA_B_ROW only: colsum_out[n] = sum_k B(k,n).  With B = dZ this is the bias
   * gradient (tf.GradientTape of keras BiasAdd), produced by the weight-gradient GEMM that
   * streams dZ anyway instead of by a second pass over it. */
  float* colsum_out;
  int32_t no_dma;         /* 1 = force the register-staged main loop (default 0: operands that are
                           * 16-byte regular go HBM -> LDS by buffer_load ... lds DMA) */
} aa_gemm_desc;

int64_t aa_gemm_f32_workspace_bytes(const aa_gemm_desc* d);
int aa_gemm_f32(const aa_gemm_desc* d, void* workspace, int64_t workspace_bytes, void* stream);
/* aa_gemm_f32 without the split-K reduce launch, for a consumer that sums the partial products in
 * its own prologue (aa_dense_small_forward_slabs): *splits_out = s > 1 -> `workspace` starts with
 * the raw fp32 slabs [s][M][N] (no bias, no activation) and C is untouched; *splits_out = 1 -> the
 * plan was not split and C holds the finished result.  mask_src must be NULL.  colsum_out != NULL
 * (a weight gradient with its bias gradient fused): the [s][N] column-sum rows follow the slabs
 * (aa_rmsprop_step_slabs sums both); with *splits_out == 1 colsum_out holds the final sums. */
int aa_gemm_f32_slabs(const aa_gemm_desc* d, void* workspace, int64_t workspace_bytes,
                      int32_t* splits_out, void* stream);

/* Dense layers with N <= 16 output units (Q-value / value heads: keras Dense(num_actions),
 * networks/q_network.py:139-150): y = act(x W + b), dx = (dz W^T) * act'(mask_src),
 * dW = x^T dz with db = column sums of dz (nullable).  W is [K,N] row-major; x rows have pitch ldx.
 * Latency-tuned kernels used instead of aa_gemm_f32 for these shapes; deterministic. */
int aa_dense_small_forward(const float* x, int64_t ldx, const float* w, const float* bias,
                           int32_t act, int64_t M, int32_t K, int32_t N, float* y, void* stream);
/* The head reading its input as the split-K slabs of the layer below (keras Dense(hidden) ->
 * Dense(num_actions), networks/q_network.py:139-150, in two launches instead of four):
 *   h[m,k] = act1(sum_z slabs[z][m][k] + bias1[k])   (stored, pitch ldh: the backward pass reads it)
 *   y[m,n] = act(sum_k h[m,k] w[k,n] + bias[n])
 * Same arithmetic, in the same order, as aa_gemm_f32's reduce followed by aa_dense_small_forward.
 * K % 4 == 0, ldh % 4 == 0, slabs / h 16-byte aligned. */
int aa_dense_small_forward_slabs(const float* slabs, int32_t splits, int64_t M, int32_t K,
                                 const float* bias1 /* nullable */, int32_t act1, float* h,
                                 int64_t ldh, const float* w, const float* bias /* nullable */,
                                 int32_t act, int32_t N, float* y, void* stream);
int aa_dense_small_dx(const float* dz, const float* w, const float* mask_src /* [M,K] nullable */,
                      int32_t mask_kind, int64_t M, int32_t K, int32_t N, float* dx, void* stream);
/* aa_dense_small_dx and aa_dense_small_dw in one launch (same results): the backward pass of the
 * head, dx [M,K] = (dz w^T) * act'(mask_src), dw [K,N] = x^T dz, db [N] (nullable). */
int aa_dense_small_backward(const float* x, int64_t ldx, const float* dz, const float* w,
                            const float* mask_src /* [M,K] nullable */, int32_t mask_kind,
                            int64_t M, int32_t K, int32_t N, float* dx, float* dw, float* db,
                            void* stream);
int aa_dense_small_dw(const float* x, int64_t ldx, const float* dz, int64_t M, int32_t K,
                      int32_t N, float* dw, float* db /* nullable */, void* stream);

/* Whole small MLPs (<= 4 Dense layers, every width <= 64: the PPO actor / value networks,
 * agents/ppo/ppo_actor_network.py:42-113) forward and backward in ONE launch each, activations and
 * the layer's weights staged in LDS, instead of one GEMM launch (+ reduce, + bias column sum) per
 * layer.  params / grads: flat fp32 buffers, layer l's kernel [dims[l]][dims[l+1]] at k_off[l], bias
 * at b_off[l].  y_*_h: host array of n_layers device pointers, layer l's output [B, dims[l+1]].
 * backward: dy = d loss / d (last layer output); grads is overwritten (deterministic slab sum);
 * dx_out nullable. */
#define AA_MLP_MAX_LAYERS 4
int aa_mlp_small_forward(const float* x, int64_t ldx, const float* params, int32_t n_layers,
                         const int32_t* dims, const int32_t* acts, const int64_t* k_off,
                         const int64_t* b_off, int64_t B, float* const* y_out_h, void* stream);
int64_t aa_mlp_small_workspace_bytes(int64_t B, int64_t total_params);
int aa_mlp_small_backward(const float* x, int64_t ldx, const float* params, int32_t n_layers,
                          const int32_t* dims, const int32_t* acts, const int64_t* k_off,
                          const int64_t* b_off, int64_t B, float* const* y_h, const float* dy,
                          float* grads, int64_t total_params, float* dx_out, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Two consecutive VALID Conv2D layers forward in one launch, one workgroup per frame (the
 * conv2 -> conv3 pair of the Mnih-15 Q-network, examples/dqn/mnih15/dqn_train_eval_atari.py:80-112;
 * keras Conv2D arithmetic, fp32 MFMA).  x: fp32 NHWC [n_img,H,W,Cin] (img_pitch floats between
 * frames, 0 = dense); each layer: w HWIO [KH,KW,Cin_l,Cout], bias nullable, y [n_img,OH,OW,Cout]
 * = act(conv + bias) -- BOTH outputs are written (the backward pass needs the middle one).
 * Limits: Cin % 32 == 0 (both layers' inputs), Cout % 16 == 0, OH*OW <= 128 per layer, both LDS frames <= 150 KiB
 * (aa_conv_pair_supported returns 1 when a shape qualifies; otherwise AA_ERR_RANGE). */
typedef struct aa_conv_layer_desc {
  const float* w;
  const float* bias;
  float* y;
  int32_t KH, KW, stride, Cout, act;
} aa_conv_layer_desc;
int aa_conv_pair_supported(int32_t n_img, int32_t H, int32_t W, int32_t Cin,
                           const aa_conv_layer_desc* first, const aa_conv_layer_desc* second);
int aa_conv_pair_forward(const float* x, int64_t img_pitch, int32_t n_img, int32_t H, int32_t W,
                         int32_t Cin, const aa_conv_layer_desc* first,
                         const aa_conv_layer_desc* second, void* stream);

/* The same pair on the bf16 matrix cores at fp32 accuracy (csrc/conv_pair_x6.hip): every fp32
 * operand is split exactly into three bf16 pieces ONCE (the frame while it is staged into LDS, the
 * middle activation in the first layer's epilogue, the filters by a pre-pass of this call into
 * `workspace`), six of the nine piece products are accumulated in fp32.  Same arguments and
 * outputs as aa_conv_pair_forward plus the scratch; limits: Cin % 32 == 0 (both layers' inputs),
 * Cout % 16 == 0, OH*OW <= 128 per layer, 3 bf16 planes of both padded LDS frames <= 160 KiB
 * (aa_conv_pair_x6_workspace_bytes returns 0 when a shape does not qualify; the forward call then
 * returns AA_ERR_RANGE). */
int64_t aa_conv_pair_x6_workspace_bytes(int32_t n_img, int32_t H, int32_t W, int32_t Cin,
                                        const aa_conv_layer_desc* first,
                                        const aa_conv_layer_desc* second);
int aa_conv_pair_x6_forward(const float* x, int64_t img_pitch, int32_t n_img, int32_t H, int32_t W,
                            int32_t Cin, const aa_conv_layer_desc* first,
                            const aa_conv_layer_desc* second, void* workspace,
                            int64_t workspace_bytes, void* stream);
/* The same call in two halves, so that the half that depends on the weights only can be issued
 * early on another stream (networks/sequential.py does, next to the layer in front of the pair):
 * phases = 1 splits the filter banks into `workspace` (x, y may be NULL), 2 runs the per-frame
 * kernel over a workspace prepared for the same weights and shapes, 3 = aa_conv_pair_x6_forward. */
int aa_conv_pair_x6_phase(const float* x, int64_t img_pitch, int32_t n_img, int32_t H, int32_t W,
                          int32_t Cin, const aa_conv_layer_desc* first,
                          const aa_conv_layer_desc* second, void* workspace,
                          int64_t workspace_bytes, int32_t phases, void* stream);


/* Input gradient of a VALID Conv2D in gather form, one workgroup per frame (no column-gradient
 * slab, no col2im): dx[b,iy,ix,ci] = act'(mask_src[b,iy,ix,ci]) * sum over the patches containing
 * (iy,ix) of dz[b,oy,ox,:] . w[ky,kx,ci,:]   (tf.GradientTape through keras Conv2D,
 * agents/dqn/dqn_agent.py:412-426).  dz [n_img,OH,OW,Cout] dense, w HWIO, dx / mask_src
 * [n_img,H,W,Cin] dense; mask_src nullable (the layer's forward input when it is an activation
 * output).  Limits: Cin % 16 == 0, Cout % 32 == 0, ceil(H/stride)*ceil(W/stride) <= 128, padded dZ
 * frame <= 150 KiB of LDS (aa_conv_dx_frame_supported returns 1 when a shape qualifies, else the
 * call returns AA_ERR_RANGE and aa_gemm_f32 + aa_col2im_f32 remain the general path). */
typedef struct aa_conv_dx_desc {
  const float* dz;
  const float* w;
  const float* mask_src;
  float* dx;
  int32_t n_img, H, W, Cin, KH, KW, stride, Cout, mask_kind;
} aa_conv_dx_desc;
int aa_conv_dx_frame_supported(const aa_conv_dx_desc* d);
int aa_conv_dx_frame(const aa_conv_dx_desc* d, void* stream);

/* The same gradient on the bf16 matrix cores at fp32 accuracy (csrc/conv_dx_frame_x6.hip): dZ is
 * split exactly into three bf16 pieces while it is staged into LDS, the filters by a pre-pass of
 * this call into `workspace` (which also receives the per-class k-step tables); six of the nine
 * piece products are accumulated in fp32.  Limits: Cin % 16 == 0, Cout a power of two >= 32,
 * stride <= 4, largest sub-pixel class <= 128 pixels, padded dZ planes <= 160 KiB of LDS
 * (aa_conv_dx_frame_x6_workspace_bytes returns 0 when a shape does not qualify). */
int64_t aa_conv_dx_frame_x6_workspace_bytes(const aa_conv_dx_desc* d);
int aa_conv_dx_frame_x6(const aa_conv_dx_desc* d, void* workspace, int64_t workspace_bytes,
                        void* stream);
/* phases = 1: filter fragments + k-step tables into `workspace` (weights only; dz, dx may be
 * NULL), 2: the per-frame kernel over a prepared workspace, 3 = aa_conv_dx_frame_x6. */
int aa_conv_dx_frame_x6_phase(const aa_conv_dx_desc* d, void* workspace, int64_t workspace_bytes,
                              int32_t phases, void* stream);

/* Weight (and bias) gradient of a VALID Conv2D over fp32 NHWC frames on the bf16 matrix cores at
 * fp32 accuracy (csrc/conv_dw_frame_x6.hip; tf.GradientTape through keras Conv2D,
 * agents/dqn/dqn_agent.py:412-426):  dw[ky,kx,ci,co] = sum x[b,oy*s+ky,ox*s+kx,ci] dz[b,oy,ox,co],
 * db[co] = sum dz (nullable).  `d` describes the layer as for the input gradient (d->dz, shapes;
 * w / mask_src / dx unused); x is the layer's forward input [n_img,H,W,Cin] dense.  Operands are
 * split once into three bf16 pieces as their frame is staged in LDS (natural [pixel][channel]
 * layout) and read as MFMA fragments with ds_read_b64_tr_b16; workgroup = (group of frames, ky);
 * per-group slabs in `workspace` are summed in fixed order.  Limits: Cout == 64, Cin a power of two
 * >= 16, KW*Cin/16 in {4, 8, 12, 16}, OH*OW <= 128 (aa_conv_dw_frame_x6_workspace_bytes returns 0
 * when a shape does not qualify and the call AA_ERR_RANGE: aa_gemm_f32 is the general path). */
int64_t aa_conv_dw_frame_x6_workspace_bytes(const aa_conv_dx_desc* d);
int aa_conv_dw_frame_x6(const aa_conv_dx_desc* d, const float* x, float* dw, float* db,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* The same in two calls, so that the slab sums of consecutive layers share ONE launch: `_slabs`
 * runs the per-frame kernel and leaves the per-group slabs (+ bias-gradient rows when want_db) in
 * `workspace`; `_reduce` sums the slabs of up to four layers (descs[l], workspaces[l]) into
 * dws[l] / dbs[l] (dbs[l] nullable) in fixed order.  Bit-identical to aa_conv_dw_frame_x6 per
 * layer.  (tf.GradientTape through keras Conv2D, agents/dqn/dqn_agent.py:412-426.) */
int aa_conv_dw_frame_x6_slabs(const aa_conv_dx_desc* d, const float* x, int32_t want_db,
                              void* workspace, int64_t workspace_bytes, void* stream);
int aa_conv_dw_frame_x6_reduce(int32_t n_layers, const aa_conv_dx_desc* const* descs,
                               const void* const* workspaces, float* const* dws,
                               float* const* dbs, void* stream);

/* out[n] = sum_m x[m*ld + n]  (bias gradients).  workspace >= aa_colsum_workspace_bytes. */
int64_t aa_colsum_workspace_bytes(int64_t M, int64_t N);
int aa_colsum_f32(const float* x, int64_t ld, int64_t M, int64_t N, float* out, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* dz = dy * act'(y): gradient through a trailing activation given its output y. */
int aa_act_backward(const float* dy, const float* y, int32_t act, int64_t n, float* dz,
                    void* stream);
/* out[0] = sum x^2 (keras l2 regulariser, tf.nn.l2_loss). */
int aa_sumsq_f32(const float* x, int64_t n, float* out, void* stream);

/* dX[b,iy,ix,c] = sum over patches containing (iy,ix) of dcol[pixel][(ky*KW+kx)*Cin + c], times
 * actgrad(mask_src) -- the input gradient of a VALID Conv2D from its column gradient. */
int aa_col2im_f32(const float* dcol, int32_t n_img, int32_t H, int32_t W, int32_t Cin, int32_t KH,
                  int32_t KW, int32_t stride, float* dx, const float* mask_src, int32_t mask_kind,
                  void* stream);

/* =========================================================================================
 * DQN loss  (agents/dqn/dqn_agent.py:75-78, 451-460, 462-579; utils/common.py:367-411,
 *            1199-1208, 1400-1476; trajectories/trajectory.py:716-850; utils/value_ops.py:21-99)
 * ========================================================================================= */
#define AA_LOSS_HUBER 0
#define AA_LOSS_SQUARED 1
/* No element-wise loss: td_loss_out[b] := td_targets[b] and td_error_out[b] := q_values[b] -- the two
 * arguments of the caller's own td_errors_loss_fn(td_targets, q_values) (agents/dqn/dqn_agent.py:114,
 * 250-251, 458) -- unmasked; loss_out := 0, dq_out := 0, field sums untouched.  aa_dqn_td_loss[_sums]
 * only (the fused head backward needs dL/dq). */
#define AA_LOSS_TARGETS 2

/* Inputs are the [B,T] trajectory fields (T = n_step+1) and the three Q tables [B,A].
 * Computes the n-step return/discount, td_targets, td_error, element-wise loss, the
 * ~is_last mask, sample weights, loss = sum/global_batch, and dq = dLoss/dq_online. */
int aa_dqn_td_loss(const float* q_online, const float* q_next_target,
                   const float* q_next_select /* nullable: online net on next obs (DDQN) */,
                   const int32_t* next_mask /* nullable [B,A]: 1 = action allowed */,
                   const void* actions, int32_t actions_are_i64, int64_t action_stride,
                   const float* reward, const float* discount, const int32_t* step_type,
                   const float* weights /* nullable [B] */, int64_t B, int32_t T, int32_t A,
                   double gamma /* n-step accumulation (agent gamma; python-float semantics) */,
                   double gamma_loss /* DqnAgent._loss(gamma=...) applied to the final discount */,
                   double reward_scale, int32_t loss_kind, float global_batch,
                   float* loss_out, float* td_loss_out, float* td_error_out, float* dq_out,
                   void* stream);
/* Same, plus field_sums_out[2] = {sum_b td_loss[b], sum_b td_error[b]} (nullable): the SUM over all
 * axes that Learner.run applies to every LossInfo field (train/learner.py:322-337), produced by the
 * loss launch instead of by two reduction launches after the optimizer step. */
int aa_dqn_td_loss_sums(const float* q_online, const float* q_next_target,
                        const float* q_next_select, const int32_t* next_mask, const void* actions,
                        int32_t actions_are_i64, int64_t action_stride, const float* reward,
                        const float* discount, const int32_t* step_type, const float* weights,
                        int64_t B, int32_t T, int32_t A, double gamma, double gamma_loss,
                        double reward_scale, int32_t loss_kind, float global_batch,
                        float* loss_out, float* td_loss_out, float* td_error_out, float* dq_out,
                        float* field_sums_out, void* stream);
/* aa_dqn_td_loss_sums and the backward pass of the Q head (the last Dense layer of the Q-network:
 * tf.GradientTape through keras Dense(num_actions), dqn_agent.py:412-426) in ONE launch: dL/dq of a
 * sample only depends on that sample, so every workgroup of the head's backward recomputes the
 * [B, A] rows it needs in LDS.  x[B, K] (row pitch ldx) = the head's input, w[K, A] its kernel,
 * mask_src / mask_kind = the derivative of the previous layer's activation (nullable),
 * dx[B, K], dw[K, A], db[A] (nullable).  Same bits as the two separate calls.  B <= 512, A <= 16. */
int aa_dqn_loss_head_backward(const float* q_online, const float* q_next_target,
                              const float* q_next_select, const int32_t* next_mask,
                              const void* actions, int32_t actions_are_i64, int64_t action_stride,
                              const float* reward, const float* discount,
                              const int32_t* step_type, const float* weights, int64_t B, int32_t T,
                              int32_t A, double gamma, double gamma_loss, double reward_scale,
                              int32_t loss_kind, float global_batch, float* loss_out,
                              float* td_loss_out, float* td_error_out, float* dq_out,
                              float* field_sums_out, const float* x, int64_t ldx, const float* w,
                              const float* mask_src, int32_t mask_kind, int32_t K, float* dx,
                              float* dw, float* db, void* stream);

/* =========================================================================================
 * Optimizers / target update / clipping  (keras optimizers; utils/common.py:250-346;
 *                                          utils/eager_utils.py:227-246; ppo_agent.py:948-949)
 * ========================================================================================= */
/* Adam, TF ApplyAdam form: alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1);
 * v += (g*g-v)(1-b2); p -= m*alpha/(sqrt(v)+eps).  t = *step_dev (already incremented). */
int aa_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, const int64_t* step_dev, void* stream);
/* keras RMSprop: ms = rho*ms + (1-rho) g^2; [mg = rho*mg + (1-rho) g];
 * denom = ms - mg^2 + eps; inc = lr*g*rsqrt(denom); [mom = momentum*mom + inc; p -= mom]. */
int aa_rmsprop_step(float* p, const float* g, float* ms, float* mg /* nullable: centered */,
                    float* mom /* nullable: momentum */, int64_t n, float lr, float rho,
                    float momentum, float eps, void* stream);
/* The same two steps, additionally keeping up to four sets of "split planes" current: the bf16x6
 * convolutions read each fp32 filter as three bf16 pieces (hi, mid, lo) stored in MFMA-fragment
 * order (aa_conv_pair_x6_phase / aa_conv_dx_frame_x6_phase, phase 1); the optimizer holds every
 * new value in a register anyway and writes its pieces to where each consumer reads them, so no
 * pre-pass launch is needed after -- or before -- the variables change (the keras apply op has no
 * such side output: the reference re-reads fp32 variables in every op).  For parameter index i in
 * [lo[t], hi[t]): pos[t][i - lo[t]] = bf16 index of the hi piece inside planes[t], or -1; the mid
 * and lo pieces follow at + stride[t] and + 2 * stride[t].  pos / planes are device pointers. */
#define AA_MAX_PLANE_TARGETS 4
typedef struct {
  int32_t n;
  int32_t stride[AA_MAX_PLANE_TARGETS];
  int64_t lo[AA_MAX_PLANE_TARGETS], hi[AA_MAX_PLANE_TARGETS];
  const int32_t* pos[AA_MAX_PLANE_TARGETS];
  uint16_t* planes[AA_MAX_PLANE_TARGETS];
} aa_plane_scatter;
int aa_adam_step_planes(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                        float beta1, float beta2, float eps, const int64_t* step_dev,
                        const aa_plane_scatter* planes /* nullable */, void* stream);
/* Adam with the step count kept by the launch itself: *steps_taken_dev = steps applied so far (the
 * launch uses t = that + 1 and its last workgroup to finish stores t back; arrival_dev = one int64
 * of scratch, zero before the first call) -- no counter launch in front of the optimizer step. */
int aa_adam_step_counted(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                         float beta1, float beta2, float eps, int64_t* steps_taken_dev,
                         int64_t* arrival_dev, const aa_plane_scatter* planes /* nullable */,
                         void* stream);
/* aa_adam_step_counted followed, in the same pass, by target = (1 - tau) * target + tau * p_new
 * (soft_variables_update, utils/common.py:250-346): SAC's critic update and the soft update of
 * its target critics when target_update_period == 1 (agents/sac/sac_agent.py:286-330, 385-410);
 * same arithmetic as aa_adam_step_counted + aa_soft_update. */
int aa_adam_step_counted_target(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                float beta1, float beta2, float eps, int64_t* steps_taken_dev,
                                int64_t* arrival_dev, float* target, float tau, void* stream);
int aa_rmsprop_step_planes(float* p, const float* g, float* ms, float* mg, float* mom, int64_t n,
                           float lr, float rho, float momentum, float eps,
                           const aa_plane_scatter* planes /* nullable */, void* stream);
/* RMSprop whose gradient comes partly from split-K slabs that have NOT been summed: segment s
 * covers parameters [offset[s], offset[s] + mn[s] + n_tail[s]) (a conv kernel followed by its
 * bias); its gradient is sum_z slab[s][z * mn + i] for i < mn and sum_z slab[s][splits * mn +
 * z * n_tail + (i - mn)] for the bias -- the layout aa_conv_dw_frame_x6_slabs and aa_gemm_f32_slabs
 * leave -- summed with the association of the reduce launch those entry points pair with (16
 * z-lanes; needs splits >= 32 and (mn + n_tail) / 4 <= 65536, AA_ERR_RANGE otherwise), so the
 * parameters come out bit-identical to reduce + aa_rmsprop_step_planes; the sums are also stored
 * to g, which is complete when the launch has run.  Everything else is read from g.  Applies when nothing sits between backward and the optimizer (no clipping, no
 * all-reduce): keras RMSprop.apply_gradients of agents/dqn/dqn_agent.py:412-449 with
 * gradient_clipping=None.  slabs == NULL or n == 0: aa_rmsprop_step_planes. */
#define AA_MAX_GRAD_SLABS 4
typedef struct aa_grad_slabs {
  int32_t n;
  int32_t splits[AA_MAX_GRAD_SLABS], mn[AA_MAX_GRAD_SLABS], n_tail[AA_MAX_GRAD_SLABS];
  int64_t offset[AA_MAX_GRAD_SLABS];
  const float* slab[AA_MAX_GRAD_SLABS];
} aa_grad_slabs;
int aa_rmsprop_step_slabs(float* p, float* g, float* ms, float* mg, float* mom, int64_t n,
                          float lr, float rho, float momentum, float eps,
                          const aa_plane_scatter* planes /* nullable */,
                          const aa_grad_slabs* slabs /* nullable */, void* stream);
/* aa_rmsprop_step_slabs whose launch also copies pack_n <= 8 fp32 device scalars side by side to
 * pack_dst (pack_src_h: HOST array of device pointers): the replica-summed LossInfo that
 * Learner.run returns (train/learner.py:322-337) -- loss, sum(td_loss), sum(td_error), written by
 * the loss launch of the same step -- leaves in storage of its own without a copy launch behind the
 * optimizer step.  slabs must be non-empty when pack_n > 0. */
int aa_rmsprop_step_slabs_pack(float* p, float* g, float* ms, float* mg, float* mom, int64_t n,
                               float lr, float rho, float momentum, float eps,
                               const aa_plane_scatter* planes, const aa_grad_slabs* slabs,
                               const float* const* pack_src_h, int32_t pack_n, float* pack_dst,
                               void* stream);
int aa_sgd_step(float* p, const float* g, int64_t n, float lr, void* stream);
/* t = (1-tau)*t + tau*s  (soft_variables_update, utils/common.py:314-346) */
int aa_soft_update(float* target, const float* source, int64_t n, float tau, void* stream);
/* sumsq_out[s] = sum g[off[s]:off[s+1]]^2 ; offsets are DEVICE int64[n_seg+1] */
int aa_segment_sumsq(const float* g, const int64_t* seg_offsets_dev, int32_t n_seg,
                     float* sumsq_out, void* stream);
/* per_tensor=1: g *= clip/max(norm_s, clip) (tf.clip_by_norm per tensor);
 * per_tensor=0: g *= clip*min(1/gnorm, 1/clip) (tf.clip_by_global_norm). */
int aa_clip_by_norm(float* g, const int64_t* seg_offsets_dev, int32_t n_seg,
                    const float* sumsq, float clip, int32_t per_tensor, void* stream);

/* =========================================================================================
 * Rollout: epsilon-greedy action selection + device-resident synthetic vector env
 *   (policies/epsilon_greedy_policy.py:120-143, q_policy.py:150-194, greedy_policy.py:70-89;
 *    environments/random_tf_environment.py, py_environment.py:233-239 auto-reset contract)
 * ========================================================================================= */
int aa_eps_greedy_action(const float* q, const int32_t* mask /* nullable [B,A] */, int64_t B,
                         int32_t A, float epsilon, const float* epsilon_dev /* nullable */,
                         uint64_t seed, int64_t* call_counter_dev,
                         int64_t* arrival_dev /* nullable: advance *call_counter_dev in-kernel */,
                         int64_t action_min, void* actions_out, int32_t actions_are_i64,
                         void* stream);

/* BoltzmannPolicy._action over QPolicy (policies/boltzmann_policy.py:83-101: the wrapped policy's
 * Categorical with logits / temperature, sampled; policies/q_policy.py:175-180: masked actions get
 * logits = float32 min; DqnAgent's collect policy when boltzmann_temperature is given,
 * agents/dqn/dqn_agent.py:357-360).  Per row b: l_a = q[b,a] / T (IEEE float32 division), masked
 * l_a = -FLT_MAX; p_a = exp(l_a - max l) evaluated and accumulated in FLOAT64 in action order;
 * u = u01(word 0 of Philox(counter = (b, call), key = seed)); the action is the first allowed a
 * whose running sum exceeds u * sum_a p_a (the last allowed one if rounding leaves none).
 * temperature_dev (nullable) overrides `temperature` (a schedule refreshed by the host between
 * graph replays).  logits_out (nullable [B,A]) receives l -- `distribution().action.logits`.
 * sample == 0: no draw, no counter advance; actions_out may be NULL (logits only). */
int aa_boltzmann_action(const float* q, const int32_t* mask /* nullable [B,A] */, int64_t B,
                        int32_t A, float temperature, const float* temperature_dev /* nullable */,
                        uint64_t seed, int64_t* call_counter_dev,
                        int64_t* arrival_dev /* nullable: advance *call_counter_dev in-kernel */,
                        int64_t action_min, void* actions_out, int32_t actions_are_i64,
                        float* logits_out /* nullable */, int32_t sample, void* stream);

/* DynamicStepDriver loop counter: counter[b] += (step_type[b] != LAST); *total_dev += the sum
 * (drivers/dynamic_step_driver.py:113,170).  counter_dev nullable.  mailbox (nullable) is a
 * host-visible int64[2] from aa_mailbox_create: the kernel stores {sequence number, total} there
 * (total first, then a system-scope fence, then ++sequence), which is how the host evaluates the
 * reference's in-graph loop condition `sum(counter) < num_steps` (:113) while HIP graphs of the
 * loop body replay, without a stream synchronisation. */
int aa_count_steps(const int32_t* step_type, int64_t B, int32_t* counter_dev, int64_t* total_dev,
                   int64_t* mailbox, void* stream);

/* Host-visible, device-writable mailbox of n int64 words (coherent pinned host memory), zeroed.
 * *host_ptr is what the host reads, *dev_ptr what kernels are given. */
int aa_mailbox_create(int64_t n_words, int64_t** host_ptr, int64_t** dev_ptr);
int aa_mailbox_destroy(int64_t* host_ptr);
/* Spins until mailbox[0] >= seq or timeout_us elapses; returns 0 and stores mailbox[1] in *value,
 * or AA_ERR_TIMEOUT. */
#define AA_ERR_TIMEOUT (-62)
int aa_mailbox_wait(const int64_t* host_ptr, int64_t seq, int64_t timeout_us, int64_t* value);

#define AA_OBS_U8 0
#define AA_OBS_F32 1
/* One batched step of the synthetic env.  In: cur_step_type[B] (state), actions ignored.
 * Out: next step_type/reward/discount/observation; cur_step_type is NOT modified. */
int aa_vecenv_random_step(const int32_t* cur_step_type, int64_t B, int64_t obs_elems,
                          int32_t obs_kind, float obs_lo, float obs_hi, float p_end,
                          uint64_t seed, int64_t* step_counter_dev,
                          int64_t* arrival_dev /* nullable: advance *step_counter_dev in-kernel */,
                          int32_t force_first, int32_t* step_type_out, float* reward_out,
                          float* discount_out, void* obs_out, void* stream);

/* =========================================================================================
 * Value ops  (utils/value_ops.py:21-99 discounted_return, :102-164 GAE; ppo_agent.py:100-110)
 * ========================================================================================= */
/* Strided [B,T] access: element (b,t) at b*stride_b + t*stride_t. final_value nullable (zeros). */
int aa_discounted_return(const float* rewards, const float* discounts, const float* final_value,
                         int64_t B, int64_t T, int64_t stride_b, int64_t stride_t, float* out,
                         void* stream);
int aa_gae(const float* values, const float* final_value, const float* discounts,
           const float* rewards, float td_lambda, int64_t B, int64_t T, int64_t stride_b,
           int64_t stride_t, float* out, void* stream);
/* out = (x - mean) / sqrt(var + eps) over all n elements (tf.nn.moments + batch_normalization);
 * stats_out holds 2+256 floats: [0]=mean, [1]=var, rest scratch. */
int aa_normalize_moments(const float* x, int64_t n, float eps, float* out, float* stats_out,
                         void* stream);

/* =========================================================================================
 * PPO loss forward+backward for the diagonal-Normal head of PPOActorNetwork
 *   (agents/ppo/ppo_agent.py:481-615, 1159-1201, 1203-1327, 1329-1512;
 *    agents/ppo/ppo_actor_network.py:30-113; utils/common.py:682-756)
 * ========================================================================================= */
/* z[N,D] = means-layer output (pre tanh), std_bias[D] (pre softplus).  act_mean/act_mag[D]
 * nullable together (no tanh squashing).  denom = N * num_replicas.  value_clip/logp_clip/c_e
 * <= 0 disable the respective term.  Outputs: dz[N,D], dbias_elem[N,D] (column-sum it for the
 * bias gradient), dv[N] (d loss / d value prediction), stats[8 + 5*256]:
 * [0] policy_gradient_loss [1] value_estimation_loss [2] entropy_regularization_loss
 * [3] clip_fraction [4] mean(entropy*weights) [5] sum of [0..2]; the rest is scratch. */
int aa_ppo_loss(const float* z, const float* std_bias, const float* act_mean,
                const float* act_mag, const float* actions, const float* old_logp,
                const float* adv, const float* returns, const float* vpred,
                const float* old_vpred, const float* weights, int64_t N, int32_t D,
                float clip_eps, float value_clip, float c_v, float c_e, float denom,
                float logp_clip, int32_t flags, float* dz, float* dbias_elem, float* dv,
                float* stats, void* stream);
/* General diagonal-Normal PPO loss: the actor hands over loc/scale [N,D] of the CURRENT policy
 * (any head) plus the collect-time dist_params (nullable: no KL terms), and gets d loss/d loc,
 * d loss/d scale, d loss/d value back (all three nullable together: forward only).
 * Adds kl_penalty_loss = adaptive (beta * mean(kl*w), beta read from *kl_beta_dev, nullable) +
 * cutoff (kl_cutoff_coef * max(0, mean(kl*w) - kl_cutoff)^2)      (ppo_agent.py:1514-1640).
 * stats[16 + 6*256]: [0] policy_gradient_loss [1] value_estimation_loss
 * [2] entropy_regularization_loss [3] clip_fraction [4] mean(entropy*w) [5] kl_penalty_loss
 * [6] sum of 0,1,2,5 [7] mean(kl*w) [8] d loss/d mean(kl*w) [9] adaptive_kl_loss
 * [10] kl_cutoff_loss; rest scratch. */
int aa_ppo_loss_dist(const float* loc, const float* scale, const float* old_loc,
                     const float* old_scale, const float* actions, const float* old_logp,
                     const float* adv, const float* returns, const float* vpred,
                     const float* old_vpred, const float* weights, int64_t N, int32_t D,
                     float clip_eps, float value_clip, float c_v, float c_e, float denom,
                     float logp_clip, const float* kl_beta_dev, float kl_cutoff_coef,
                     float kl_cutoff, float* dloc, float* dscale, float* dv, float* stats,
                     void* stream);
/* PPOActorNetwork head (ppo_actor_network.py:42-113): loc = mean + mag*tanh(z) (identity when
 * mean/mag are null), scale = softplus(std_bias) broadcast; and its backward. */
int aa_ppo_head_forward(const float* z, const float* std_bias, const float* act_mean,
                        const float* act_mag, int64_t N, int32_t D, float* loc, float* scale,
                        void* stream);
/* aa_ppo_head_forward + aa_normal_sample (+ the clip of the action to [clip_lo, clip_hi], both [D]
 * or both NULL, and the advance of *call_counter_dev by one once every workgroup has read it;
 * arrival_dev: one int64 of scratch, zero before the first call) in one launch: what
 * PPOPolicy._action does per environment step (policies/actor_policy.py:150-230 under
 * agents/ppo/ppo_policy.py).  loc / scale are written where the caller wants them (the policy
 * info of the trajectory), action = loc + scale * eps with aa_normal_sample's Philox draw. */
int aa_ppo_head_forward_sample(const float* z, const float* std_bias, const float* act_mean,
                               const float* act_mag, int64_t N, int32_t D, float* loc, float* scale,
                               uint64_t seed, int64_t* call_counter_dev, int64_t* arrival_dev,
                               const float* clip_lo, const float* clip_hi, float* action,
                               void* stream);
/* The WHOLE collect-policy step in one launch: observation normalisation (aa_norm_apply; nrm_var_num
 * NULL = none, nrm_var_den NULL = var_num is the variance), the actor body (a) and the value body
 * (b) -- two <= 64-wide Dense stacks (aa_mlp_small_forward's layout arguments, host arrays) on the
 * same normalised observation x [B, dims_a[0] == dims_b[0]] --, the actor head, the Normal draw, the
 * clip and the Philox counter advance of aa_ppo_head_forward_sample.  Each piece with the
 * arithmetic of the launch it replaces (bit-identical results).  value_out: [B]. */
typedef struct {
  const float* x; int64_t ldx; int64_t B;
  const float* nrm_mean; const float* nrm_var_num; const float* nrm_var_den;
  float nrm_eps, nrm_clip;
  const float* params_a; int32_t n_layers_a; const int32_t* dims_a; const int32_t* acts_a;
  const int64_t* k_off_a; const int64_t* b_off_a;
  const float* params_b; int32_t n_layers_b; const int32_t* dims_b; const int32_t* acts_b;
  const int64_t* k_off_b; const int64_t* b_off_b;
  float* value_out;
  const float* std_bias; const float* act_mean; const float* act_mag; int32_t D;
  float* loc; float* scale;
  uint64_t seed; int64_t* call_counter_dev; int64_t* arrival_dev;
  const float* clip_lo; const float* clip_hi; float* action;
} aa_ppo_policy_step_desc;
int aa_ppo_policy_step(const aa_ppo_policy_step_desc* d, void* stream);
int aa_ppo_head_backward(const float* z, const float* std_bias, const float* act_mag,
                         const float* dloc, const float* dscale, int64_t N, int32_t D, float* dz,
                         float* dbias_elem, void* stream);
/* out[i] = sum_d log N(x[i,d]; loc[i,d], scale[i,d])   (utils/common.py:682-717) */
int aa_normal_log_prob(const float* loc, const float* scale, const float* x, int64_t N, int32_t D,
                       float* out, void* stream);
/* out = loc + scale * eps, eps ~ N(0,1) (Box-Muller on Philox(counter = (i, *call_counter), key =
 * seed)); replaces tfd.Normal.sample in PPOPolicy._action (policies/actor_policy.py). */
int aa_normal_sample(const float* loc, const float* scale, int64_t n, uint64_t seed,
                     const int64_t* call_counter_dev, float* out, void* stream);
/* out[i,d] = lo[d] + (hi[d] - lo[d]) * u, u ~ U[0,1) (Philox(counter = (element, *call_counter),
 * key = seed)): RandomTFPolicy on a bounded continuous action spec -- replaces tf.random.uniform in
 * tensor_spec.sample_bounded_spec (specs/tensor_spec.py:327-420) under
 * policies/random_tf_policy.py:60-150 (the SAC script's initial collect policy). */
int aa_uniform_sample(const float* lo, const float* hi, int64_t N, int32_t D, uint64_t seed,
                      const int64_t* call_counter_dev, float* out, void* stream);
/* out[b,t] = discount[b,t] * gamma * (next_step_type[b,t] != LAST), t < T1-1
 * (ppo_agent.py:630-676, utils/common.py:883-895); inputs are [B,T1]. */
int aa_ppo_discounts(const float* discount, const int32_t* next_step_type, float gamma, int64_t B,
                     int64_t T1, float* out, void* stream);
/* out = weights (or 1) * (step_type != LAST) * !(return == 0 && advantage == 0)
 * (ppo_utils.make_trajectory_mask, agents/ppo/ppo_utils.py:35-59). */
int aa_ppo_trajectory_mask(const int32_t* step_type, const float* returns, const float* advantages,
                           const float* weights, int64_t n, float* out, void* stream);
/* update_adaptive_kl_beta (ppo_agent.py:1642-1690). */
int aa_ppo_update_kl_beta(const float* mean_kl_dev, float target, float tolerance, float* beta_dev,
                          void* stream);
/* g += c * p  (L2 regularisation gradient on a flat parameter range). */
int aa_add_l2_grad(float* g, const float* p, int64_t n, float c, void* stream);

/* One PPO minibatch train step -- PPOAgent._train's epoch body (ppo_agent.py:895-960) for a
 * feed-forward tanh-Normal actor (PPOActorNetwork, ppo_actor_network.py:30-113) and a value MLP,
 * every layer <= 64 wide, without KL / L2 terms -- in THREE launches (csrc/ppo_fused.hip):
 * [advantage normalisation over the minibatch, trajectory mask, old log-prob, observation
 * normaliser, both forwards, loss, both backwards -> per-workgroup gradient slabs] ->
 * [slab sum, sum of squares, LossInfo scalars, step counter] -> [global-norm clip + Adam].
 * All per-sample inputs are the N rows of the minibatch (N = B * T flattened).
 * params / grads / adam_m / adam_v: the agent's flat buffers of `total` floats laid out
 * [actor body | head: std_bias[D] (+ padding) | value body]; the layouts hold ABSOLUTE float
 * offsets into them.  nrm_*: rows of a StreamingTensorNormalizer state (all NULL: observations
 * are used as they are); act_mean / act_mag NULL: unbounded action spec (loc = z).
 * stats9 = {policy_gradient_loss, value_estimation_loss, entropy_regularization_loss,
 * clip_fraction, mean(entropy * w), kl_penalty_loss = 0, total, 0, l2 = 0}; flat grads hold the
 * CLIPPED gradient afterwards; sumsq_out (nullable) = squared global norm before clipping.
 * The workspace must be zero-filled once after allocation (alignment padding of the flat layout
 * is never written). */
#define AA_PPO_FUSED_MAX_D 16
typedef struct {
  int32_t n_layers;
  int32_t dims[AA_MLP_MAX_LAYERS + 1];
  int32_t acts[AA_MLP_MAX_LAYERS];
  int64_t k_off[AA_MLP_MAX_LAYERS], b_off[AA_MLP_MAX_LAYERS];
} aa_mlp_layout;
/* Whole MLPs of wide Dense layers (every hidden / output width <= 256, input width <= 1024,
 * <= 4 layers) at small batch -- SAC's actor and twin (target) critics, (256, 256) hidden layers at
 * batch 256 (agents/sac/sac_agent.py:286-330, networks/critic_network.py:150-170,
 * networks/actor_distribution_network.py) -- forward in ONE launch, backward in TWO, for up to
 * AA_MLPW_MAX_NETS networks of the same layout per launch (csrc/mlp_wide.hip).  Replaces one GEMM
 * launch (+ split-K reduce) per Dense layer and direction, and the copies that built the
 * [observation | action] input: the first layer reads columns [0, x_split) of its input from x
 * and the rest from x2 (x_split = layout.dims[0]: x only, x2 unused).
 *   params[g] / grads[g]: flat fp32 buffers of network g (layer l's kernel [dims[l]][dims[l+1]] at
 *     k_off[l], bias at b_off[l]);  y[g][l]: layer l's output [B, dims[l+1]] (all written by
 *     forward, read by backward);  dout[g]: d loss / d y[g][last], row stride ld_dout.
 *   backward: dz[g][l] [B, dims[l+1]] receives d loss / d (pre-activation of layer l) (workspace
 *     the weight-gradient launch reads);  dx[g] (all NULL or none): d loss / d input columns
 *     [dx_lo, dx_hi) written to dx[g][b * ld_dx + column] (a critic's action gradient: only those
 *     rows of the first kernel are read);  grads[g] (all NULL or none): every kernel and bias
 *     gradient, written (not accumulated), full batch per 32 x 32 tile, fixed summation order. */
#define AA_MLPW_MAX_NETS 4
#define AA_MLPW_MAX_BATCH 1024
typedef struct {
  aa_mlp_layout layout;
  int32_t n_nets, x_split;
  int64_t B;
  const float* params[AA_MLPW_MAX_NETS];
  const float* x[AA_MLPW_MAX_NETS];
  int64_t ldx[AA_MLPW_MAX_NETS];
  const float* x2[AA_MLPW_MAX_NETS];
  int64_t ldx2[AA_MLPW_MAX_NETS];
  float* y[AA_MLPW_MAX_NETS][AA_MLP_MAX_LAYERS];
} aa_mlp_wide_fwd;
typedef struct {
  aa_mlp_layout layout;
  int32_t n_nets, x_split;
  int64_t B;
  const float* params[AA_MLPW_MAX_NETS];
  const float* x[AA_MLPW_MAX_NETS];
  int64_t ldx[AA_MLPW_MAX_NETS];
  const float* x2[AA_MLPW_MAX_NETS];
  int64_t ldx2[AA_MLPW_MAX_NETS];
  const float* y[AA_MLPW_MAX_NETS][AA_MLP_MAX_LAYERS];
  const float* dout[AA_MLPW_MAX_NETS];
  int64_t ld_dout[AA_MLPW_MAX_NETS];
  float* dz[AA_MLPW_MAX_NETS][AA_MLP_MAX_LAYERS];
  float* dx[AA_MLPW_MAX_NETS];
  int64_t ld_dx[AA_MLPW_MAX_NETS];
  int32_t dx_lo, dx_hi;
  float* grads[AA_MLPW_MAX_NETS];
} aa_mlp_wide_bwd;
/* 1 when (layout, B) is within the limits above, else 0 (callers then take the per-layer path) */
int aa_mlp_wide_supported(const aa_mlp_layout* layout, int64_t B);
int aa_mlp_wide_forward(const aa_mlp_wide_fwd* d, void* stream);
/* aa_mlp_wide_forward whose network `net` is a SAC actor (last layer = [mean | raw_std], 2A wide):
 * the workgroup that has just produced a sample's head output also draws its tanh-squashed action
 * and log-probability -- aa_sac_sample's arithmetic and Philox stream, bit for bit -- instead of a
 * second launch reading the head output back (SacAgent._actions_and_log_probs,
 * agents/sac/sac_agent.py:533-558, right behind the actor network's call).  Fields as the
 * arguments of aa_sac_sample (below); the head output is still written to y[net][last] (the
 * backward pass reads it). */
typedef struct {
  int32_t net, A, std_kind;
  const float* act_mean; const float* act_mag;
  const float* eps_in;                       /* nullable */
  uint64_t seed;
  int64_t* call_counter_dev; int64_t* arrival_dev;   /* arrival nullable, as in aa_sac_sample */
  float* action; float* logp;
  float* save_tanh; float* save_sigma; float* save_eps;   /* all or none */
} aa_sac_sample_tail;
int aa_mlp_wide_forward_sample(const aa_mlp_wide_fwd* d, const aa_sac_sample_tail* tail,
                               void* stream);
/* Two networks of the launch draw their samples (tail_a->net != tail_b->net; drawn noise: two
 * different call counters and arrival words) -- SAC's critic update evaluates the actor on the next
 * observations and its actor update on the observations: the same weights, one launch
 * (sac_agent.py:533-560 and :620-650 read the actor before any of its variables changes). */
int aa_mlp_wide_forward_sample2(const aa_mlp_wide_fwd* d, const aa_sac_sample_tail* tail_a,
                                const aa_sac_sample_tail* tail_b, void* stream);
int aa_mlp_wide_backward(const aa_mlp_wide_bwd* d, void* stream);
/* aa_mlp_wide_backward whose d loss / d output is COMPUTED by the gradient-chain launch instead of
 * read from dout (which is ignored): SAC's critic loss, actor loss and actor-head backward are a
 * few loads and flops per sample, and each was a launch of its own on the train step's chain
 * (agents/sac/sac_agent.py:559-694).  Same arithmetic as aa_sac_critic_loss / aa_sac_actor_loss /
 * aa_sac_head_backward (csrc/sac_loss.h), bit for bit.
 *   AA_SAC_GEN_CRITIC  networks 0, 1 = the twin critics: dout_g = d critic_loss / d q_g; also
 *                      writes *loss_out and td_target_out[B] (nullable)
 *   AA_SAC_GEN_ACTOR   networks 0, 1 = the twin critics: dout_g = d actor_loss / d q_g; also writes
 *                      *loss_out and dlogp_out[B] (d actor_loss / d log_pi)
 *   AA_SAC_GEN_HEAD    network 0 = the actor: dout = aa_sac_head_backward's dz from daction
 *                      (+ daction2), dlogp and the tensors aa_sac_sample saved */
#define AA_SAC_GEN_CRITIC 1
#define AA_SAC_GEN_ACTOR 2
#define AA_SAC_GEN_HEAD 3
typedef struct {
  int32_t kind;
  const float* q1; const float* q2;
  const float* tq1; const float* tq2; const float* next_logp; const float* reward;
  const float* discount;
  const float* logp;
  const float* weights;            /* nullable */
  const float* log_alpha;
  float gamma, reward_scale;
  int32_t loss_kind;
  float loss_weight, global_batch;
  float* loss_out; float* td_target_out; float* dlogp_out;
  const float* z; int32_t A, std_kind;
  const float* act_mag; const float* save_tanh; const float* save_sigma; const float* save_eps;
  const float* daction; int64_t ld_daction; const float* daction2; int64_t ld_daction2;
  const float* dlogp;
} aa_sac_dout_gen;
int aa_mlp_wide_backward_gen(const aa_mlp_wide_bwd* d, const aa_sac_dout_gen* gen, void* stream);
/* aa_mlp_wide_backward[_gen] whose weight-gradient launch also STEPS the optimizer of the networks'
 * parameters: Adam is elementwise, and the workgroup that finishes a 32 x 32 tile of a weight
 * gradient holds it -- TensorFlow's ApplyAdam (aa_adam_step_counted's arithmetic and in-launch
 * step counter: *step_dev = steps taken so far, advanced by the last workgroup; arrival_dev: a
 * zeroed int64 word) and, with target[g] set, soft_variables_update(tau) of a target copy from the
 * parameters just written (aa_adam_step_counted_target).  p[g] must be d->params[g]; m / v /
 * target [g] the same offsets into their flat buffers.  gen may be NULL.  For callers without
 * gradient clipping or a cross-replica gradient reduction in front of the optimizer
 * (sac_agent.py:286-330: critic_loss -> apply_gradients, no clipping by default). */
typedef struct {
  float* p[AA_MLPW_MAX_NETS]; float* m[AA_MLPW_MAX_NETS]; float* v[AA_MLPW_MAX_NETS];
  float* target[AA_MLPW_MAX_NETS];          /* each nullable */
  float lr, beta1, beta2, eps, tau;
  int64_t* step_dev; int64_t* arrival_dev;
} aa_mlp_wide_adam;
int aa_mlp_wide_backward_gen_adam(const aa_mlp_wide_bwd* d, const aa_sac_dout_gen* gen,
                                  const aa_mlp_wide_adam* adam, void* stream);
/* The weight-gradient launch of aa_mlp_wide_backward alone (the gradient chain of an earlier
 * aa_mlp_wide_backward[_gen] call with grads = NULL left dz in place), with the optimizer step if
 * adam != NULL: SAC's actor gradient chain runs beside the collect step, the launch that writes
 * the actor's weights behind it. */
int aa_mlp_wide_dw_adam(const aa_mlp_wide_bwd* d, const aa_mlp_wide_adam* adam, void* stream);
/* Measurement aid: the workgroups of the following aa_mlp_wide_backward launches write
 * wall_clock64() stamps (10 ns ticks) at the phase boundaries of the gradient chain to
 * buf[workgroup][16] (0 start, 1 operands staged, then per layer from the top: dz ready, product
 * done); NULL = off. */
int aa_mlp_wide_debug_stamps(int64_t* buf);

typedef struct {
  const float* obs; int64_t ld_obs; int32_t obs_dim; int32_t D;
  const float* actions; const float* old_loc; const float* old_scale;   /* [N, D] */
  const float* returns; const float* adv; const float* old_vpred;      /* [N]; old_vpred nullable */
  const int32_t* step_type; const float* weights;                      /* [N]; weights nullable */
  int64_t N;
  const int64_t* rows;   /* nullable [N]: minibatch sample b is row rows[b] of the arrays above
                          * (the shuffle's permutation slice, train/ppo_learner.py:228-247) */
  const float* nrm_count; const float* nrm_avg; const float* nrm_m2;   /* [obs_dim] or all NULL */
  float nrm_eps, nrm_clip;
  const float* params; int64_t total; int64_t head_off;
  aa_mlp_layout actor, value;
  const float* act_mean; const float* act_mag;                          /* [D] or both NULL */
  float clip_eps, value_clip, c_v, c_e, denom, logp_clip, adv_eps;
} aa_ppo_fused_desc;
int64_t aa_ppo_fused_workspace_bytes(int64_t N, int64_t total_params);
/* Measurement aid: the workgroups of the following aa_ppo_fused_step calls write wall_clock64()
 * stamps (10 ns ticks) at their phase boundaries to buf[ceil(N / 16)][32]; NULL = off. */
int aa_ppo_fused_debug_stamps(int64_t* buf);
/* The slab reduction and clip + Adam of a fused step run as ONE launch when the flat parameter
 * vector has <= 16384 floats (<= 256 workgroups, all co-resident): each workgroup publishes its
 * sum of squares in a tagged 8-byte slot of the workspace and waits for the others' -- nothing
 * else crosses workgroups, and the results are those of the two-launch form bit for bit.
 * on = 0 / 1 sets the process-wide switch (default 1), on < 0 only reads it; returns the previous
 * value.  The workspace must have been zero-filled once by its owner (slots, launch sequence and
 * the slab padding live in it). */
int32_t aa_ppo_fused_merge_apply(int32_t on);
int aa_ppo_fused_step(const aa_ppo_fused_desc* d, float* grads, float* adam_m, float* adam_v,
                      int64_t* adam_step_dev, float lr, float beta1, float beta2, float adam_eps,
                      float grad_clip /* <= 0: none */, float* stats9, float* sumsq_out,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* n_steps consecutive minibatch steps from one host call -- the minibatch loop of
 * train/ppo_learner.py:220-248 over one epoch's shuffle: step s trains on rows
 * rows_dev[s * N .. (s + 1) * N) of the sample arrays (all frames of the collected batch). */
int aa_ppo_fused_epoch(const aa_ppo_fused_desc* d, const int64_t* rows_dev, int32_t n_steps,
                       float* grads, float* adam_m, float* adam_v, int64_t* adam_step_dev, float lr,
                       float beta1, float beta2, float adam_eps, float grad_clip, float* stats9,
                       float* sumsq_out, void* workspace, int64_t workspace_bytes, void* stream);

/* =========================================================================================
 * Prioritized (proportional) sampling -- the north star's "segment-tree sampling".  No reference
 * class to mirror (prioritisation exists there only through Reverb); plugs into the reference's
 * hooks: DqnLossInfo.td_error (agents/dqn/dqn_agent.py:50-72), BufferInfo.ids and
 * Learner.after_train_strategy_step_fn (train/learner.py:362-376).  Priorities are uint32 fixed
 * point (2^-16 units): sums are exact uint64, sampled indices are bit-exact against the oracle.
 * ========================================================================================= */
int64_t aa_prio_workspace_bytes(int64_t capacity);
/* S rows with P(row) = prio_q[row] / sum over rows whose stored id is a valid window start
 * (tf_uniform_replay_buffer.py:610-635); rows_out[s,t] as aa_rb_sample_rows; prob_out[s] = that
 * probability.  Advances *call_counter_dev by one. */
int aa_prio_sample_rows(const uint32_t* prio_q, const int64_t* id_table,
                        const int64_t* last_id_dev, int64_t batch, int64_t max_len, int64_t S,
                        int64_t T, uint64_t seed, int64_t* call_counter_dev, void* workspace,
                        int64_t workspace_bytes, int64_t* rows_out, float* prob_out,
                        int* err_flag_dev, void* stream);
/* The same draw (same rows, probabilities, counter advance, bit for bit) as ONE launch: the
 * block sums cross workgroups as tagged 8-byte words, the first ceil(S / 4) workgroups build
 * their prefix in LDS once and find a sample's row with one round of loads.  workspace:
 * aa_prio_draw_workspace_bytes(capacity) bytes (-1: more than 8,000 blocks of 1,024 rows -- use
 * aa_prio_sample_rows), 8-byte aligned, zero-filled ONCE by its owner (launch sequence and
 * arrival count live in it).  start_rows_out (nullable): [S] = rows_out[:, 0], contiguous, the
 * argument aa_prio_set wants for the priorities of this batch. */
int64_t aa_prio_draw_workspace_bytes(int64_t capacity);
int aa_prio_draw_rows(const uint32_t* prio_q, const int64_t* id_table, const int64_t* last_id_dev,
                      int64_t batch, int64_t max_len, int64_t S, int64_t T, uint64_t seed,
                      int64_t* call_counter_dev, void* workspace, int64_t workspace_bytes,
                      int64_t* rows_out, int64_t* start_rows_out, float* prob_out,
                      int* err_flag_dev, void* stream);
/* prio_q[rows[i]] = clamp(round((|priorities[i]| + eps)^alpha * 65536), 1, 2^32-1);
 * *max_prio_q_dev = max(itself, those). */
int aa_prio_set(const int64_t* rows, const float* priorities, int64_t n, float alpha, float eps,
                int64_t capacity, uint32_t* prio_q, uint32_t* max_prio_q_dev, void* stream);
/* the rows add_batch just wrote (frame id *last_id_dev) take the running maximum priority */
int aa_prio_on_add(const int64_t* last_id_dev, int64_t batch, int64_t max_len,
                   const uint32_t* max_prio_q_dev, uint32_t* prio_q, void* stream);

/* =========================================================================================
 * SAC  (agents/sac/sac_agent.py:314-410, 533-740; agents/sac/tanh_normal_projection_network.py;
 *       distributions/utils.py:40-160 SquashToSpecNormal)
 * ========================================================================================= */
#define AA_SAC_STD_EXP 0        /* TanhNormalProjectionNetwork default std_transform = tf.exp */
#define AA_SAC_STD_CLIP_EXP 1   /* sac_agent.std_clip_transform: exp(clip(raw, -20, 2))       */
/* Actor head: z = [mean | raw_std] ([B,2A], the projection Dense output) -> reparameterised
 * tanh-squashed sample action = act_mean + act_mag * tanh(mean + sigma*eps) and its log-probability
 * (Normal log-density at the pre-tanh sample minus log|mag| and the stable tanh log-det-Jacobian).
 * eps_in nullable: N(0,1) noise supplied by the caller; else drawn from Philox(seed, *counter).
 * arrival_dev nullable: one int64 of scratch (zero before the first call); when given (and the
 * noise is drawn here) the launch advances *call_counter_dev by one itself once every workgroup
 * has used it -- no counter launch after the sample.
 * save_* (all or none, [B,A]) keep tanh(x), sigma, eps for aa_sac_head_backward. */
int aa_sac_sample(const float* z, int64_t B, int32_t A, const float* act_mean,
                  const float* act_mag, int32_t std_kind, const float* eps_in, uint64_t seed,
                  int64_t* call_counter_dev, int64_t* arrival_dev, float* action, float* logp,
                  float* save_tanh, float* save_sigma, float* save_eps, void* stream);
/* dz[B,2A] = d loss / d head output from d loss / d action (nullable; [B,A] with row stride
 * ld_daction; daction2 nullable: the gradient is daction + daction2 -- the action columns of the
 * twin critics' input gradients, sac_agent.py:646-694, without a launch that adds them first) and
 * d loss / d log_pi. */
int aa_sac_head_backward(const float* z, int64_t B, int32_t A, const float* act_mag,
                         int32_t std_kind, const float* save_tanh, const float* save_sigma,
                         const float* save_eps, const float* daction, int64_t ld_daction,
                         const float* daction2, int64_t ld_daction2, const float* dlogp,
                         float* dz, void* stream);
/* critic_loss: td = scale*r + gamma*d*(min(tq1,tq2) - exp(log_alpha)*next_logp);
 * loss = weight * sum_b w_b (f(td,q1)+f(td,q2)) / global_batch; dq1/dq2 nullable (both or none). */
int aa_sac_critic_loss(const float* q1, const float* q2, const float* tq1, const float* tq2,
                       const float* next_logp, const float* reward, const float* discount,
                       const float* weights, const float* log_alpha_dev, float gamma,
                       float reward_scale, int32_t loss_kind, float loss_weight, int64_t B,
                       float global_batch, float* loss_out, float* td_target_out, float* dq1,
                       float* dq2, void* stream);
/* actor_loss: weight * sum_b w_b (exp(log_alpha) logp - min(q1,q2)) / global_batch. */
int aa_sac_actor_loss(const float* q1, const float* q2, const float* logp, const float* weights,
                      const float* log_alpha_dev, float loss_weight, int64_t B,
                      float global_batch, float* loss_out, float* dq1, float* dq2, float* dlogp,
                      void* stream);
/* alpha_loss: weight * sum_b w_b c(log_alpha) (-logp - target_entropy) / global_batch,
 * c = log_alpha (use_log_alpha) or exp(log_alpha); grad_out[0] = d loss / d log_alpha. */
int aa_sac_alpha_loss(const float* logp, const float* weights, const float* log_alpha_dev,
                      float target_entropy, int32_t use_log_alpha, float loss_weight, int64_t B,
                      float global_batch, float* loss_out, float* grad_out, void* stream);
/* The tail of SacAgent.train in one launch (sac_agent.py:296-330, 696-740): aa_sac_alpha_loss, one
 * Adam step on log_alpha with its gradient (aa_adam_step_counted's arithmetic; *adam_steps_dev =
 * steps taken so far, read and advanced here) and the LossInfo pack of aa_pack_sum3_f32:
 * packed4 = [critic + actor + alpha loss, critic, actor, alpha].  Bit-identical to the three
 * launches it replaces. */
int aa_sac_alpha_step(const float* logp, const float* weights, float* log_alpha_dev,
                      float target_entropy, int32_t use_log_alpha, float loss_weight, int64_t B,
                      float global_batch, float* loss_out, float* grad_out, float* adam_m,
                      float* adam_v, int64_t* adam_steps_dev, float lr, float beta1, float beta2,
                      float eps, const float* critic_loss, const float* actor_loss,
                      float* packed4, void* stream);

/* LossInfo packing (one launch instead of clone + add + clone on the agents' train paths):
 * aa_pack_small_f32: out[0..n) = src[0..n), out[n] = addend ? *addend : 0, out[add_at] += out[n]
 *   (PPOAgent: the loss kernel's stats vector + the l2 regularisation term, ppo_agent.py:566-615);
 * aa_pack_sum3_f32: out4 = [a + b + c, a, b, c] (SacAgent._train total, sac_agent.py:296-330). */
int aa_pack_small_f32(const float* src, int32_t n, const float* addend, int32_t add_at, float* out,
                      void* stream);
int aa_pack_sum3_f32(const float* a, const float* b, const float* c, float* out4, void* stream);
/* Up to 8 strided row copies of 4-byte-element matrices in one launch: dst_i[r*dst_pitch_i + c] =
 * src_i[r*src_pitch_i + c], r < rows, c < cols_i (pitches in elements).  Assembles the AsTransition
 * slices of a [B, 2, ...] batch and the [observation | action] inputs of SAC's twin critics
 * (data_converter.py:300-380, agents/sac/sac_agent.py:533-640) instead of one copy per slice. */
int aa_copy_segments(const void* const* src_h, void* const* dst_h, const int64_t* src_pitch_h,
                     const int64_t* dst_pitch_h, const int32_t* cols_h, int32_t n_segments,
                     int64_t rows, void* stream);
/* out[r, c] = a[r*lda + c] + b[r*ldb + c] (out dense [rows, cols]): d loss / d action through the
 * twin critics of SAC, summed (sac_agent.py:599-640: tape.gradient through both Q networks). */
int aa_add_strided_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows,
                       int64_t cols, float* out, void* stream);

/* =========================================================================================
 * Tensor normalisers   (tf_agents/utils/tensor_normalizer.py)
 *   x is [n_outer, n_inner] fp32, the n_inner elements of one tensor-spec leaf fastest.
 * ========================================================================================= */
/* Floats of caller-provided scratch the two update entries need for a leaf of n_inner elements. */
int64_t aa_norm_scratch_floats(int64_t n_inner);
/* StreamingTensorNormalizer._update_ops (:288-348): batch (n, mean, M2) merged into the running
 * state by parallel_variance_calculation + kahan_summation (:397-474).
 * state = 4 rows of n_inner floats: count (initialised to 1e-8), avg, m2, m2_carry (:288-312). */
int aa_streaming_norm_update(const float* x, int64_t n_outer, int64_t n_inner, float* state,
                             float* scratch, void* stream);
/* EMATensorNormalizer._update_ops (:236-281): state = 2 rows of n_inner floats: mean (init 0),
 * var (init 1); var's batch statistic is taken about the OLD moving mean. */
int aa_ema_norm_update(const float* x, int64_t n_outer, int64_t n_inner, float rate, float* state,
                       float* scratch, void* stream);
/* TensorNormalizer.normalize (:134-206) = tf.nn.batch_normalization without scale / offset:
 * out = clip(x * inv + (-mean * inv)), inv = 1/sqrt(var + variance_epsilon),
 * var = var_num / var_den (var_den nullable: var = var_num); mean nullable (center_mean=False);
 * clip_value <= 0 disables clipping. */
int aa_norm_apply(const float* x, int64_t n_outer, int64_t n_inner, const float* mean,
                  const float* var_num, const float* var_den, float variance_epsilon,
                  float clip_value, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGENTS_AMD_H_ */
