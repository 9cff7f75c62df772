// Dense fp32 contractions (keras Dense forward / input gradient / weight gradient: the fc1 layer of
// the Mnih-15 Q-network, 3136 x 512, and the (256, 256) SAC / PPO MLPs) on the bf16 matrix cores at
// fp32 accuracy (x6_common.h: exact three-piece split, six of nine piece products, fp32
// accumulation), default plan for A in {row-major, column-major} x B in {row-major, column-major}
// with K a multiple of 32.
//
// Why it beats the fp32-MFMA plans here although fc1 is not MFMA-bound: at M = 256 these GEMMs are
// a few hundred small workgroups whose time is ramp + prologue + k-loop + epilogue; the k-loop
// shrinks (24 bf16 MFMAs of 16 cycles per 32-k step and wave against 16 fp32 MFMAs of 64), one
// barrier per k-step, and the operand bytes cross the LDS once as ready bf16 fragments.
//
// Every operand element is split ONCE per workgroup that uses it, on its way into LDS:
//   * a workgroup owns a 64 x 64 output tile over a K range; per 32-k step each thread fetches 8
//     values of A and 8 of B -- two 16-byte loads when K is the contiguous index of the operand
//     (x of the forward pass, dZ of the input gradient, W of the input gradient), eight coalesced
//     4-byte loads when it is not (W of the forward pass, x^T and dZ of the weight gradient: the
//     "transpose" is just which 8 addresses a thread reads) -- splits them (cx_split8) and writes
//     three 16-byte fragments pieces into the [row][k] planes of the next LDS stage;
//   * the loads of step s+1 are issued before the MFMAs of step s, the split + LDS writes come
//     after them: global latency hides under the matrix pipe, one barrier per step;
//   * LDS rows are 96 B apart (64 B of data): the ds_read_b128 fragment reads of a 16-row tile are
//     bank-conflict free under the lane-group model of MI355X_MICROARCH.md.
// 4 waves, each a 32 x 32 quadrant = 2 x 2 MFMA tiles (v_mfma_f32_16x16x32_bf16).
// Split-K slabs and the fused bias gradient (column sums of B) use the conventions of gemm.hip, so
// aa_splitk_reduce_kernel finishes both; blocks are dealt to the XCDs by aa_block_of.
#pragma once
#include "x6_common.h"

#include <type_traits>

#define AA_X6D_BM 64
#define AA_X6D_BN 64
#define AA_X6D_PITCH 96                                   /* bytes per LDS row of a plane */
#define AA_X6D_PLANE (64 * AA_X6D_PITCH)                  /* 64 rows */
#define AA_X6D_STAGE (6 * AA_X6D_PLANE)                   /* A: 3 planes, B: 3 planes */
#define AA_X6D_DEPTH 4                                    /* k-steps of operand values in flight */

// 8 consecutive-k values of operand row `r` for k-step base k0 (thread's octet kq).
// KC: element (r, k) at base[r * ld + k]; otherwise at base[k * ld + r].
template <bool KC>
__device__ static inline void x6d_fetch(const float* __restrict__ base, int ld, int r, int k,
                                        float (&v)[8]) {
  if (KC) {
    const float4* s = reinterpret_cast<const float4*>(base + (size_t)r * ld + k);
    const float4 a = s[0], b = s[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const float* s = base + (size_t)k * ld + r;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s[(size_t)i * ld];
  }
}

#ifdef AA_X6D_STAMPS      /* tools/x6d_probe.hip: per-workgroup wall_clock64 stamps (10 ns ticks) */
__device__ long long* d_x6d_stamps = nullptr;
#define X6D_STAMP(i)                                                             \
  if (d_x6d_stamps != nullptr && threadIdx.x == 0)                               \
    d_x6d_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64();
#else
#define X6D_STAMP(i)
#endif

template <bool A_KC, bool B_KC, bool COLSUM>
__global__ void __launch_bounds__(256) aa_gemm_x6d_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char x6d_lds[];   // [2 stages][A 3 planes | B 3 planes]
  AaBlk blk;
  if (!aa_block_of(p, &blk)) return;
  X6D_STAMP(0)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blk.x * AA_X6D_BM, n0 = blk.y * AA_X6D_BN;
  const int k_begin = blk.z * p.k_per_split;
  int k_end = k_begin + p.k_per_split;
  if (k_end > p.K) k_end = p.K;
  const int nk = (k_end - k_begin) >> 5;
  const float* A = reinterpret_cast<const float*>(p.A);

  // staging role of this thread: operand row + k octet (see x6d_fetch)
  const int a_row = A_KC ? (tid >> 2) : (tid & 63), a_kq = A_KC ? (tid & 3) : (tid >> 6);
  const int b_row = B_KC ? (tid >> 2) : (tid & 63), b_kq = B_KC ? (tid & 3) : (tid >> 6);
  int am = m0 + a_row, bn = n0 + b_row;
  if (am >= p.M) am = p.M - 1;            // clamped rows are computed and never stored
  if (bn >= p.N) bn = p.N - 1;
  const int a_dst = a_row * AA_X6D_PITCH + a_kq * 16;
  const int b_dst = 3 * AA_X6D_PLANE + b_row * AA_X6D_PITCH + b_kq * 16;
  // fused bias gradient: column sums of B over this block's k range (B row-major only)
  const bool do_colsum = COLSUM && blk.x == 0;
  float csum = 0.f;

  cx_f32x4 big[2][2], small[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      big[i][j] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
      small[i][j] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // Operand values run AA_X6D_DEPTH k-steps ahead in a register ring (ring slot = step mod
  // depth): an HBM / L2 round trip is several k-steps of MFMA work at this tile size, and with a
  // single step of lookahead the loop ran at one memory latency per step.  Refills are
  // unconditional (index clamped to the last step): no branches around the loads.
  float va[AA_X6D_DEPTH][8], vb[AA_X6D_DEPTH][8];
  const int last = nk - 1;
  auto fetch = [&](auto dc, int s) {
    constexpr int D_ = decltype(dc)::value;
    const int k = k_begin + 32 * (s < last ? s : last);
    x6d_fetch<A_KC>(A, p.lda, am, k + 8 * a_kq, va[D_]);
    x6d_fetch<B_KC>(p.B, p.ldb, bn, k + 8 * b_kq, vb[D_]);
  };
  auto stash = [&](auto dc, int stage, float count) {      // count: 1 = a real step, 0 = the
    constexpr int D_ = decltype(dc)::value;                //        unread move after the last one
    char* base = x6d_lds + stage * AA_X6D_STAGE;
    uint4 f[3];
    cx_split8(va[D_], f);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<uint4*>(base + pl * AA_X6D_PLANE + a_dst) = f[pl];
    if (COLSUM) {      // (every M tile adds; only blk.x == 0 publishes: no branch in the k loop)
#pragma unroll
      for (int i = 0; i < 8; ++i) csum += count * vb[D_][i];
    }
    cx_split8(vb[D_], f);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<uint4*>(base + pl * AA_X6D_PLANE + b_dst) = f[pl];
  };
  const int a_frag = (wm * 32 + lr) * AA_X6D_PITCH + lg * 16;
  const int b_frag = 3 * AA_X6D_PLANE + (wn * 32 + lr) * AA_X6D_PITCH + lg * 16;
  auto compute = [&](int s) {
    const char* base = x6d_lds + (s & 1) * AA_X6D_STAGE;
    CxFrag a[2][3], b[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        a[i][pl].q = *reinterpret_cast<const uint4*>(base + pl * AA_X6D_PLANE + a_frag +
                                                     i * 16 * AA_X6D_PITCH);
        b[i][pl].q = *reinterpret_cast<const uint4*>(base + pl * AA_X6D_PLANE + b_frag +
                                                     i * 16 * AA_X6D_PITCH);
      }
    // six products per tile, smallest first; consecutive MFMAs on different accumulators
#define X6D_MMA(PA, PB, ACC)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)   \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA].v, b[j][PB].v, ACC[i][j], 0, 0, 0);
    X6D_MMA(2, 0, small)
    X6D_MMA(0, 2, small)
    X6D_MMA(1, 1, small)
    X6D_MMA(1, 0, small)
    X6D_MMA(0, 1, small)
    X6D_MMA(0, 0, big)
#undef X6D_MMA
  };
  // one k-step: ring slot D_ holds step s (already in LDS stage s & 1); refill it with step
  // s + depth, multiply, move step s + 1 (slot D_ + 1) into the other stage, barrier.  The move is
  // unconditional (after the last step it writes a stage nobody reads) so that the MFMAs and the
  // split arithmetic share ONE basic block: with one wave per SIMD (a few hundred workgroups on
  // 256 CUs) nothing else fills the matrix pipe's 16 busy cycles per MFMA, and the scheduler is
  // asked for 1 MFMA : 4 VALU -- the ~100 VALU of the two splits then issue under the 24 MFMAs
  // instead of after them.
  auto step = [&](auto dc, int s) {
    constexpr int D_ = decltype(dc)::value;
    constexpr int NX = (D_ + 1) % AA_X6D_DEPTH;
    fetch(dc, s + AA_X6D_DEPTH);
    compute(s);
    stash(std::integral_constant<int, NX>{}, (s + 1) & 1, s + 1 < nk ? 1.f : 0.f);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
    }
    __syncthreads();
  };
  if (nk > 0) {
    fetch(std::integral_constant<int, 0>{}, 0);
    fetch(std::integral_constant<int, 1>{}, 1);
    fetch(std::integral_constant<int, 2>{}, 2);
    fetch(std::integral_constant<int, 3>{}, 3);
    stash(std::integral_constant<int, 0>{}, 0, 1.f);
  }
  __syncthreads();
  X6D_STAMP(1)
  int s = 0;
  for (; s + 3 < nk; s += 4) {
    step(std::integral_constant<int, 0>{}, s);
    step(std::integral_constant<int, 1>{}, s + 1);
    step(std::integral_constant<int, 2>{}, s + 2);
    step(std::integral_constant<int, 3>{}, s + 3);
  }
  if (s < nk) step(std::integral_constant<int, 0>{}, s);
  if (s + 1 < nk) step(std::integral_constant<int, 1>{}, s + 1);
  if (s + 2 < nk) step(std::integral_constant<int, 2>{}, s + 2);

  X6D_STAMP(2)
  const bool raw = p.splits > 1;
  // ---- fused bias gradient: 4 k-octet partials per column, added in octet order ---------------
  if (do_colsum) {
    float* red = reinterpret_cast<float*>(x6d_lds);      // (all fragment reads are done)
    red[b_kq * 64 + b_row] = csum;
    __syncthreads();
    if (tid < 64 && n0 + tid < p.N) {
      const float sum = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
      if (raw) p.C[(size_t)p.splits * p.M * p.N + (size_t)blk.z * p.N + n0 + tid] = sum;
      else p.colsum_out[n0 + tid] = sum;
    }
  }

  // ---- epilogue: through LDS so that every lane stores 16 contiguous bytes ----------------------
  // (lane-per-column stores of the MFMA layout are 4-byte writes in 64-byte runs: 16 store
  // instructions per lane, store-issue bound at 2-4 us per workgroup.)  Each wave transposes its
  // own 32 x 32 quadrant: no workgroup barrier needed beyond the one that ends the k loop.
  {
    if (do_colsum) __syncthreads();          // `red` above shares the LDS
    float* tile = reinterpret_cast<float*>(x6d_lds) + wave * (32 * 36);   // [32][36] floats
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          tile[(16 * i + 4 * lg + e) * 36 + 16 * j + lr] = big[i][j][e] + small[i][j][e];
    // (same wave wrote and reads: program order + the LDS's in-order return suffice)
    float* C = raw ? p.C + (size_t)blk.z * (size_t)p.M * (size_t)p.N : p.C;
    const int ldc = raw ? p.N : p.ldc;
    const int c4 = lane & 7, r0 = lane >> 3;
    const int n = n0 + wn * 32 + 4 * c4;
    const bool vec_ok = (ldc & 3) == 0 && n + 3 < p.N && ((uintptr_t)C & 15) == 0 &&
                        (p.mask_kind == 0 || ((p.ldm & 3) == 0 && ((uintptr_t)p.mask_src & 15) == 0));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = r0 + 8 * it;
      const int m = m0 + wm * 32 + r;
      if (m >= p.M || n >= p.N) continue;
      const float4 t = *reinterpret_cast<const float4*>(tile + r * 36 + 4 * c4);
      float v[4] = {t.x, t.y, t.z, t.w};
      if (!raw) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nq = n + q < p.N ? n + q : p.N - 1;
          v[q] = aa_act(v[q] + (p.bias != nullptr ? p.bias[nq] : 0.f), p.act);
        }
        if (p.mask_kind != 0) {
          if (vec_ok) {
            const float4 y = *reinterpret_cast<const float4*>(p.mask_src + (size_t)m * p.ldm + n);
            v[0] *= aa_actgrad(y.x, p.mask_kind); v[1] *= aa_actgrad(y.y, p.mask_kind);
            v[2] *= aa_actgrad(y.z, p.mask_kind); v[3] *= aa_actgrad(y.w, p.mask_kind);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (n + q < p.N)
                v[q] *= aa_actgrad(p.mask_src[(size_t)m * p.ldm + n + q], p.mask_kind);
          }
        }
      }
      if (vec_ok) {
        *reinterpret_cast<float4*>(C + (size_t)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n + q < p.N) C[(size_t)m * ldc + n + q] = v[q];
      }
    }
  }
  X6D_STAMP(3)
}

// Shapes this plan takes: dense A and B, K % 32 == 0, 16-byte aligned K-contiguous operands.
static bool aa_x6d_ok(const aa_gemm_desc* d) {
  if (d->a_mode != AA_A_ROW && d->a_mode != AA_A_COL) return false;
  if (d->b_mode != AA_B_ROW && d->b_mode != AA_B_COL) return false;
  if (d->K % 32 != 0 || d->K < 64 || d->M < 16 || d->N < 16) return false;
  if (d->colsum_out != nullptr && d->b_mode != AA_B_ROW) return false;
  if (d->a_mode == AA_A_ROW && (d->lda % 4 != 0 || ((uintptr_t)d->A & 15) != 0)) return false;
  if (d->b_mode == AA_B_COL && (d->ldb % 4 != 0 || ((uintptr_t)d->B & 15) != 0)) return false;
  return true;
}

// tiles x splits: aim at >= 192 workgroups; split K (in whole 32-k steps, equal ranges when the
// step count allows) only when the tiles alone are fewer, never below four steps per workgroup
static void aa_x6d_plan(const aa_gemm_desc* d, int* splits, int* k_per_split) {
  const int64_t tiles = ((d->M + AA_X6D_BM - 1) / AA_X6D_BM) * ((d->N + AA_X6D_BN - 1) / AA_X6D_BN);
  const int steps = d->K / 32;
  int s = 1;
  if (d->force_splits > 0) {
    s = d->force_splits;
  } else if (tiles < 160) {
    s = (int)((224 + tiles - 1) / tiles);
    const int max_by_k = steps / 4 > 0 ? steps / 4 : 1;
    if (s > max_by_k) s = max_by_k;
    for (int t = s; t > s - 3 && t >= 1; --t)      // prefer a divisor of the step count nearby
      if (steps % t == 0) { s = t; break; }
  }
  if (s > steps) s = steps;
  const int per = (steps + s - 1) / s;
  *k_per_split = per * 32;
  *splits = (steps + per - 1) / per;
}

static int aa_x6d_launch(const GemmP& p, bool a_kc, bool b_kc, hipStream_t st) {
  const int n = p.gx * p.gy * p.gz;
  const dim3 grid = p.xcd_mode != 0 ? dim3(((n + 7) / 8) * 8, 1, 1) : dim3(p.gx, p.gy, p.gz);
  const size_t smem = 2 * (size_t)AA_X6D_STAGE;      // 72 KiB
  const bool cs = p.colsum_out != nullptr;
  static bool granted[AA_MAX_DEVICES][8] = {{false}};   // per kernel and device
  const int dv = aa_device_ordinal();
  if (dv < 0) return AA_ERR_LAUNCH;
#define AA_X6D_CASE(AKC_, BKC_, CS_)                                                              \
  if (a_kc == AKC_ && b_kc == BKC_ && cs == CS_) {                                                \
    bool& g = granted[dv][(AKC_ ? 4 : 0) + (BKC_ ? 2 : 0) + (CS_ ? 1 : 0)];                       \
    if (!g) {                                                                                     \
      if (hipFuncSetAttribute((const void*)aa_gemm_x6d_kernel<AKC_, BKC_, CS_>,                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
        return AA_ERR_LAUNCH;                                                                     \
      g = true;                                                                                   \
    }                                                                                             \
    hipLaunchKernelGGL((aa_gemm_x6d_kernel<AKC_, BKC_, CS_>), grid, dim3(256), smem, st, p);      \
  }
  AA_X6D_CASE(true, true, false) AA_X6D_CASE(true, false, false) AA_X6D_CASE(true, false, true)
  AA_X6D_CASE(false, false, false) AA_X6D_CASE(false, false, true) AA_X6D_CASE(false, true, false)
#undef AA_X6D_CASE
  return aa_launch_status();
}
