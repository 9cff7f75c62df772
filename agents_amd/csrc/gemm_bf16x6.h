// fp32 forward contractions (keras Conv2D / Dense: y = act(x W + b)) on the bf16 matrix cores at
// fp32 accuracy.
//
// Every fp32 value is EXACTLY the sum of three round-to-nearest bf16 pieces (x = x1 + x2 + x3,
// |x2| <= 2^-9 |x|, |x3| <= 2^-18 |x|).  x * w is then the sum of nine piece products; the three
// smallest (x2 w3, x3 w2, x3 w3 <= 2^-27 |x w|) are below fp32 rounding and dropped.  Each kept
// product of two bf16 numbers is exact in fp32, and v_mfma_f32_32x32x16_bf16 accumulates in fp32,
// so the result carries fp32-class error (tests: within 2x of the fp32 MFMA kernel against
// float64).  Six bf16 MFMAs cost 6/16 of one fp32 MFMA of the same shape.
//
// Structure (same as the uint8 conv1 kernel, conv_u8_bf16.h): a workgroup owns a 32-column slice of
// W over a K range; it splits that slice once into LDS, in B-fragment order (one conflict-free
// ds_read_b128 per fragment).  A wave owns TPW 32-row tiles; its A fragments come straight from
// global memory in MFMA layout through a register ring several groups deep: lane (row r, half h)
// loads the 64 contiguous bytes [32 G + 16 h, +16) of k-group G (conv patches are K-contiguous per
// patch row), so the two halves of a wave consume each 128-byte line whole in one instruction --
// with 32-byte loads every line was fetched from L2 four times (measured: 8 of 23 us).  K is
// permuted accordingly (chunk (G, q) pairs k = 32 G + 16 h + 8 q + e); W is stored in that order.
// The A values are split in registers
// (5.5 VALU per element, hidden under the MFMAs), and meet the LDS fragments in the matrix core.
// No LDS traffic for A and no barrier in the main loop.  K ranges longer than the LDS slice are
// split over blockIdx.z into slabs for aa_splitk_reduce_kernel (deterministic).
//
// STATUS (MI355X, round 1): numerically validated (tests/test_gpu_gemm.py) but NOT the default plan.
// On the DQN shapes it only ties the fp32 MFMA kernels: conv2 23.0 us (fp32 23.0), conv3 20.3
// (18.9), fc1 20.3 (19.5).  In-kernel timestamps on conv2 (162 workgroups x 8 waves): workgroup
// start spread 4.2 us, W split + barrier 2.5 us, main loop 13.4 us = ~1000 cycles per 16-k chunk
// against 192 cycles of MFMA issue: the loop is bound by the ~65 VALU instructions per chunk of the
// in-register A split (v_cvt_pk_bf16_f32 + shifts + subtractions per element and piece) competing
// with the MFMAs for issue slots, not by memory (deeper rings, 64-byte runs per lane, two column
// slices per wave and pinning the refill loads ahead of the MFMAs all left it within +-1.5 us).
// The uint8 conv1 kernels win because a byte converts to bf16 exactly in 1.5 VALU and needs one
// piece, not three.  What would make this pay: producers that emit their activations already split
// (3 bf16 planes), so that consumers load fragments instead of computing them.  Selected only with
// force_cfg = 9.
#pragma once

#define AA_X6_MAX_KS 768      /* k per workgroup: 192 B of LDS each */
#ifndef AA_X6_DEPTH
#define AA_X6_DEPTH 4        /* A chunks (2 x 16-byte loads each) in flight per tile */
#endif

struct AaX6Plan {
  int nw, tn, splits, k_per_split, k_phase;
};

// 8 floats -> three packed bf16 fragments (hi, mid, lo)
__device__ static inline void aa_split8(const float4& lo4, const float4& hi4, AaFrag (&f)[3]) {
  const float a[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  unsigned pc[3][4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    float r0 = a[e], r1 = a[e + 1];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const unsigned pk = aa_pk_bf16(r0, r1);
      pc[s][e >> 1] = pk;
      if (s < 2) {
        r0 -= __uint_as_float(pk << 16);
        r1 -= __uint_as_float(pk & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) f[s].q = make_uint4(pc[s][0], pc[s][1], pc[s][2], pc[s][3]);
}

template <int NW, int TN, bool PATCH>
__global__ void __launch_bounds__(NW * 64)
aa_fwd_bf16x6_kernel(GemmP p, int n_mblk, int n_sgrp, int k_phase) {
  constexpr int NT = NW * 64;
  constexpr int D = AA_X6_DEPTH;
  extern __shared__ __attribute__((aligned(16))) uint4 wfrag[];   // [TN][3][Jp][2][32]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;

  // XCD-aware: consecutive workgroups go to different XCDs; give each XCD a contiguous range of
  // (row block, column group) pairs so the column groups of a row block share one L2.
  const int n_blk = n_mblk * n_sgrp;
  const int per_xcd = (n_blk + 7) >> 3;
  const int idx = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (idx >= n_blk) return;
  const int mblk = idx / n_sgrp, sgrp = idx - mblk * n_sgrp;
  const int n0 = sgrp * (32 * TN);
  const int kbeg = (int)blockIdx.z * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  if (kend > p.K) kend = p.K;

  const int m0 = (mblk * NW + wave) * 32;
  const bool active = m0 < p.M;              // (inactive waves still take part in the barriers)
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* row;
  {
    int m = m0 + r;
    if (m >= p.M) m = p.M - 1;
    row = (PATCH ? A + aa_pix_base(p, m) : A + (size_t)m * p.lda) + 16 * h;
  }
  f32x16 big[TN], small[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int e = 0; e < 16; ++e) { big[tn][e] = 0.f; small[tn][e] = 0.f; }

  for (int kph = kbeg; kph < kend; kph += k_phase) {
    int kpe = kph + k_phase;
    if (kpe > kend) kpe = kend;
    const int J = (kpe - kph) >> 4;          // 16-k chunks of this phase (multiples of 32 k)
    if (kph != kbeg) __syncthreads();        // the previous phase's readers are done
    // ---- W columns: fp32 [k][32 TN] -> three bf16 pieces in fragment order ------------------
    {
      const int c = tid & 31;
      const int n_item = J * 2 * 32;         // (k-octet, column) items per column slice
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const bool c_ok = n0 + 32 * tn + c < p.N;
        const float* bcol = p.B + (c_ok ? n0 + 32 * tn + c : 0);
        uint4* wf = wfrag + (size_t)tn * 3 * J * 64;
        for (int item0 = tid; item0 < n_item; item0 += NT * 4) {
          float v[4][8];   // unconditional loads (clamped): 32 in flight
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            int o = (item0 + u * NT) >> 5;
            if (o > 2 * J - 1) o = 2 * J - 1;
            const float* src = bcol + (size_t)(kph + o * 8) * p.ldb;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = src[(size_t)e * p.ldb];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int item = item0 + u * NT;
            if (item >= n_item) continue;
            const int o = item >> 5;
            AaFrag f[3];
            aa_split8(make_float4(v[u][0], v[u][1], v[u][2], v[u][3]),
                      make_float4(v[u][4], v[u][5], v[u][6], v[u][7]), f);
#pragma unroll
            for (int s = 0; s < 3; ++s)
              wf[((s * J + (((o >> 2) << 1) | (o & 1))) * 2 + ((o >> 1) & 1)) * 32 + c] =
                  c_ok ? f[s].q : make_uint4(0, 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    if (!active) continue;

    // offset (floats) of a 32-k group inside a row / patch; the load stream walks it
    int ld_ky = 0, ld_kin = kph;
    if (PATCH) { ld_ky = kph / p.seg; ld_kin = kph - ld_ky * p.seg; }
    auto next_off = [&]() {
      const int off = PATCH ? ld_ky * p.rowpitch + ld_kin : ld_kin;
      ld_kin += 32;
      if (PATCH && ld_kin >= p.seg) { ld_kin = 0; ++ld_ky; }
      return off;
    };
    float4 buf[D][4];
    auto issue = [&](int d) {
      const int off = next_off();
#pragma unroll
      for (int u = 0; u < 4; ++u) buf[d][u] = *reinterpret_cast<const float4*>(row + off + 4 * u);
    };
    auto mma = [&](int j, AaFrag (&a)[3]) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const uint4* wf = wfrag + (size_t)tn * 3 * J * 64;
        AaFrag b[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) b[s].q = wf[((s * J + j) * 2 + h) * 32 + r];
        // smallest products first
        small[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, small[tn], 0, 0, 0);
        small[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, small[tn], 0, 0, 0);
        small[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, small[tn], 0, 0, 0);
        small[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, small[tn], 0, 0, 0);
        small[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, small[tn], 0, 0, 0);
        big[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, big[tn], 0, 0, 0);
      }
    };
    auto group = [&](int d, int g, bool refill) {   // ring slot d = k-group g (two chunks)
      AaFrag a[2][3];
      aa_split8(buf[d][0], buf[d][1], a[0]);
      aa_split8(buf[d][2], buf[d][3], a[1]);
      if (refill) issue(d);
      // keep the refill HERE: left alone, the scheduler sinks the loads below the MFMAs to the
      // end of the loop body, where the next iteration immediately waits for them
      __builtin_amdgcn_sched_barrier(0);
      mma(2 * g, a[0]);
      mma(2 * g + 1, a[1]);
    };
    const int G = J >> 1;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < G) issue(d);
    int g0 = 0;
    // steady state: every consumed group is replaced by an unconditional load (so the compiler
    // can count outstanding loads instead of draining them)
    for (; g0 + 2 * D <= G; g0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) group(d, g0 + d, true);
    }
    // tail: up to 2 D - 1 groups; the ring is refilled only while groups remain
    for (; g0 < G; g0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int g = g0 + d;
        if (g < G) group(d, g, g + D < G);
      }
    }
  }
  if (!active) return;

  const bool raw = p.splits > 1;
  float* C = raw ? p.C + (size_t)blockIdx.z * (size_t)p.M * (size_t)p.N : p.C;
  const int ldc = raw ? p.N : p.ldc;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + 32 * tn + r;
    if (n >= p.N) continue;
    const float bv = (!raw && p.bias != nullptr) ? p.bias[n] : 0.f;
    auto emit = [&](auto actc) {
      constexpr int ACT = decltype(actc)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = big[tn][e] + small[tn][e];
        if (ACT >= 0) v += bv;
        if (ACT == AA_ACT_RELU) v = v > 0.f ? v : 0.f;
        if (ACT == AA_ACT_TANH) v = tanhf(v);
        C[(size_t)m * ldc + n] = v;
      }
    };
    if (raw) emit(std::integral_constant<int, -1>{});
    else if (p.act == AA_ACT_RELU) emit(std::integral_constant<int, AA_ACT_RELU>{});
    else if (p.act == AA_ACT_TANH) emit(std::integral_constant<int, AA_ACT_TANH>{});
    else emit(std::integral_constant<int, AA_ACT_NONE>{});
  }
}

// Forward contractions this plan takes: fp32 rows (dense or conv patches) x row-major W, K a
// multiple of 32 with 16-byte-aligned 16-float runs, no mask / column-sum epilogue.
static bool aa_fwd_x6_ok(const aa_gemm_desc* d) {
  if (d->b_mode != AA_B_ROW || d->mask_src != nullptr || d->colsum_out != nullptr) return false;
  if (d->K % 32 != 0 || d->K < 32 * AA_X6_DEPTH || d->N < 32 || d->M < 32) return false;
  if (((uintptr_t)d->A & 15) != 0) return false;
  if (d->a_mode == AA_A_ROW) return d->lda % 4 == 0;
  if (d->a_mode == AA_A_PATCH) {
    const int seg = d->KW * d->Cin;
    return seg % 32 == 0 && (d->Cin * d->stride) % 4 == 0 && (d->W * d->Cin) % 4 == 0 &&
           (d->img_pitch == 0 || d->img_pitch % 4 == 0);
  }
  return false;
}

static void aa_fwd_x6_plan(const aa_gemm_desc* d, AaX6Plan* x) {
  const int64_t tiles = (d->M + 31) / 32;
  x->tn = 1;   // (two slices per wave measured slower: 24.6 vs 23.0 us on conv2 -- see the header)
  const int64_t sgrps = (d->N + 32 * x->tn - 1) / (32 * x->tn);
  // waves per workgroup: 8 when that still gives most CUs a workgroup, else 4
  x->nw = (tiles / 8) * sgrps >= 160 ? 8 : (tiles <= 8 ? 8 : 4);
  const int64_t mblk = (tiles + x->nw - 1) / x->nw;
  const int64_t max_ks = AA_X6_MAX_KS;
  int64_t splits = (d->K + max_ks - 1) / max_ks;
  const int64_t blocks = mblk * sgrps;
  if (blocks * splits < 192) {            // few row blocks (Dense layers): split K to fill the chip
    int64_t want = 256 / blocks;
    const int64_t max_by_k = d->K / (32 * AA_X6_DEPTH);   // at least one full ring per workgroup
    if (want > max_by_k) want = max_by_k;
    if (want > splits) splits = want;
  }
  const int64_t groups = d->K / 32;
  for (int64_t s = splits; s * 2 > splits && s >= 1; --s)   // prefer equal K ranges
    if (groups % s == 0 && groups / s * 32 <= max_ks) { splits = s; break; }
  if (d->force_splits > 0) splits = d->force_splits;
  int64_t kps = (d->K + splits - 1) / splits;
  kps = ((kps + 31) / 32) * 32;
  if (kps < 32 * AA_X6_DEPTH) kps = 32 * AA_X6_DEPTH;
  if (kps > max_ks) kps = max_ks;
  x->k_per_split = (int)kps;
  x->splits = (int)((d->K + kps - 1) / kps);
  // LDS holds tn slices of one K phase: 192 B per k per slice, <= 96 KiB
  int64_t phase = kps;
  const int64_t cap = (x->tn == 2 ? 256 : 512);
  if (phase > cap) {
    const int64_t n_ph = (phase + cap - 1) / cap;
    phase = (((phase + n_ph - 1) / n_ph + 31) / 32) * 32;
  }
  x->k_phase = (int)phase;
}

template <int NW, int TN, bool PATCH>
static int aa_fwd_x6_launch_t(const GemmP& p, const AaX6Plan& x, hipStream_t st) {
  const int tiles = (p.M + 31) / 32, n_sgrp = (p.N + 32 * TN - 1) / (32 * TN);
  const int n_mblk = (tiles + NW - 1) / NW;
  const size_t smem = (size_t)TN * 3 * (x.k_phase >> 4) * 2 * 32 * sizeof(uint4);
  static size_t lds_limit = 0;   // dynamic LDS above 64 KiB has to be granted once per process
  if (smem > 65536 && smem > lds_limit) {
    if (hipFuncSetAttribute((const void*)aa_fwd_bf16x6_kernel<NW, TN, PATCH>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return AA_ERR_LAUNCH;
    lds_limit = smem;
  }
  const int per_xcd = (n_mblk * n_sgrp + 7) >> 3;
  hipLaunchKernelGGL((aa_fwd_bf16x6_kernel<NW, TN, PATCH>), dim3(per_xcd * 8, 1, p.splits),
                     dim3(NW * 64), smem, st, p, n_mblk, n_sgrp, x.k_phase);
  return aa_launch_status();
}

static int aa_fwd_x6_launch(const GemmP& p, const AaX6Plan& x, bool patch, hipStream_t st) {
#define AA_X6_CASE(NW_, TN_)                                                              \
  if (x.nw == NW_ && x.tn == TN_)                                                         \
    return patch ? aa_fwd_x6_launch_t<NW_, TN_, true>(p, x, st)                           \
                 : aa_fwd_x6_launch_t<NW_, TN_, false>(p, x, st);
  AA_X6_CASE(8, 1)
  AA_X6_CASE(4, 1)
#undef AA_X6_CASE
  return AA_ERR_INVALID;
}
