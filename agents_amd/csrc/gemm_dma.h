// LDS-DMA main loop of the fp32 MFMA GEMM (included by gemm.hip after GemmP and its helpers).
//
// Why: at the sizes of the trainer hot path (1-2 GFLOP per contraction, tiles of 32x64 so that
// all 256 CUs get work) a K-step is only 256-1024 MFMA cycles per wave, far less than the
// 1-2 us a global load takes, and staging through registers (gemm.hip, aa_gemm_kernel) can keep
// just two K-tiles in flight before the VGPR budget halves the occupancy.  Here the operand tiles
// go HBM -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction, no VGPRs, no
// ds_write pass) into a ring of NS stages, NS-1 K-tiles in flight per workgroup, one raw
// s_barrier per K-step and counted `s_waitcnt vmcnt(N)` (never 0 inside the loop).
//
// LDS images (one stage holds BK = 32 k of both operands; every 16-byte granule is written by one
// lane of one DMA instruction, so any granule permutation is free on the SOURCE side):
//   T (K-contiguous operand: A_ROW, conv patches, B_COL)   [BX rows][8 granules of 4 k]
//        granule kg of row x sits at slot kg ^ ((x >> 1) & 7): a ds_read_b128 of "row = lane"
//        then touches 16 distinct 16-byte slots per 16-lane group -> conflict-free
//   D (X-contiguous operand: A_COL, B_ROW, transposed patches)   [32 k][BX floats], read with
//        ds_read_b32 at consecutive x -> conflict-free, no swizzle needed
//   T-u8 (uint8 conv patches, forward)   [2 granules of 16 k][BX rows]  (granule-major so that the
//        ds_read_b128 of 32 different rows hits 16 distinct slots per group)
//   D-u8 (uint8 transposed patches)      [32 pixels][BX bytes], ds_read_u8 (4 lanes share a dword)
// MFMA k-order inside each group of 8 k: instruction j (0..3) multiplies k = j (lanes 0-31) and
// k = 4 + j (lanes 32-63) -- a permutation of the summation order that lets a lane take its four
// values from ONE 16-byte read of a T image.  A and B use the same assignment, so it is only a
// different (still fixed, deterministic) fp32 summation order.
#pragma once

#define AA_KIND_T_DENSE 0    // f32, elem(x,k) = base[x*ld + k]
#define AA_KIND_T_PATCH 1    // f32 conv patches, x = output pixel, k = patch element
#define AA_KIND_T_PATCH_U8 2
#define AA_KIND_D_DENSE 3    // f32, elem(x,k) = base[k*ld + x]
#define AA_KIND_D_PATCHT 4   // f32 conv patches transposed: x = patch element, k = output pixel
#define AA_KIND_D_PATCHT_U8 5

typedef int aa_i32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor words (base, stride 0, num_records = bytes, gfx9 dword-format flags) for
// the inline-asm DMA below; same descriptor __builtin_amdgcn_make_buffer_rsrc builds.
__device__ static inline aa_i32x4 aa_make_desc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  aa_i32x4 d;
  d.x = (int)(unsigned)(a & 0xffffffffull);
  d.y = (int)(unsigned)((a >> 32) & 0xffffull);
  d.z = (int)bytes;
  d.w = 0x00020000;
  return d;
}

// LDS byte address (offset inside the workgroup's allocation) of a generic pointer into smem.
__device__ static inline unsigned aa_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

// One LDS-DMA piece: 64 lanes x 16 bytes from desc[voff] to LDS[lds_byte + 16*lane].  Issued from
// inline asm ON PURPOSE: hipcc does not see an LDS write, so it neither drains vmcnt before the
// ds_reads of the ring nor before the barrier (it inserts `s_waitcnt vmcnt(0)` ahead of the first
// ds_read of every K-step when the DMA is the builtin) -- completion is tracked by the counted
// aa_wait_vmcnt<N>() below.  M0 is saved/restored because the compiler reserves it.
__device__ static inline void aa_dma16(unsigned voff, aa_i32x4 desc, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(desc), "s"(lds_byte)
      : "memory");
}

template <int KIND, int BX>
struct DmaOp {
  static constexpr bool kIsT = KIND <= AA_KIND_T_PATCH_U8;
  static constexpr bool kIsU8 = KIND == AA_KIND_T_PATCH_U8 || KIND == AA_KIND_D_PATCHT_U8;
  static constexpr int kBytes = BX * AA_BK * (kIsU8 ? 1 : 4);          // real bytes per stage
  static constexpr int kPadded = (kBytes + 4095) / 4096 * 4096;        // 4 waves x 1 KiB pieces
  static constexpr int kNIW = kPadded / 4096;                          // DMA instructions per wave
  static constexpr int kGranules = kBytes / 16;

  unsigned base[kNIW];   // per-lane source byte offset at k-step 0 (>= AA_OOB: never valid)
  int kidx[kNIW];        // T: first k of the lane's granule inside the tile; D: the lane's k row

  // x0: first row/column of this workgroup's tile, X: operand extent in that direction.
  __device__ inline void init(const GemmP& p, int ld, int x0, int X, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < kNIW; ++q) {
      const int G = (q * 4 + wave) * 64 + lane;  // LDS granule index written by this lane
      unsigned b = AA_OOB;
      int kk = 0;
      if (G < kGranules) {
        if constexpr (KIND == AA_KIND_T_DENSE) {
          const int x = G >> 3, kg = (G & 7) ^ ((x >> 1) & 7);
          kk = 4 * kg;
          if (x0 + x < X) b = 4u * ((unsigned)(x0 + x) * (unsigned)ld + (unsigned)kk);
        } else if constexpr (KIND == AA_KIND_T_PATCH) {
          const int x = G >> 3, kg = (G & 7) ^ ((x >> 1) & 7);
          kk = 4 * kg;
          if (x0 + x < X) b = 4u * (unsigned)aa_pix_base(p, x0 + x);
        } else if constexpr (KIND == AA_KIND_T_PATCH_U8) {
          const int g = G / BX, x = G - g * BX;
          kk = 16 * g;
          if (x0 + x < X) b = (unsigned)aa_pix_base(p, x0 + x);
        } else if constexpr (KIND == AA_KIND_D_DENSE) {
          constexpr int V = BX / 4;
          const int k = G / V, xg = G - k * V;
          kk = k;
          if (x0 + 4 * xg < X) b = 4u * ((unsigned)k * (unsigned)ld + (unsigned)(x0 + 4 * xg));
        } else if constexpr (KIND == AA_KIND_D_PATCHT) {
          constexpr int V = BX / 4;
          const int k = G / V, xg = G - k * V;
          kk = k;
          if (x0 + 4 * xg < X) b = 4u * (unsigned)aa_patch_off(p, x0 + 4 * xg);
        } else {
          constexpr int V = BX / 16;
          const int k = G / V, xg = G - k * V;
          kk = k;
          if (x0 + 16 * xg < X) b = (unsigned)aa_patch_off(p, x0 + 16 * xg);
        }
      }
      base[q] = b;
      kidx[q] = kk;
    }
  }

  // Enqueue the DMA of K-tile [k0, k0+32) into `stage` (LDS byte address of this operand's image).
  __device__ inline void issue(const GemmP& p, aa_i32x4 r, int ld, unsigned stage, int k0,
                               int k_end, int wave) const {
#pragma unroll
    for (int q = 0; q < kNIW; ++q) {
      const int k = k0 + kidx[q];
      unsigned off;
      if constexpr (KIND == AA_KIND_T_DENSE) {
        off = base[q] + 4u * (unsigned)k0;
      } else if constexpr (KIND == AA_KIND_T_PATCH) {
        off = base[q] + 4u * (unsigned)aa_patch_off(p, k < k_end ? k : 0);
      } else if constexpr (KIND == AA_KIND_T_PATCH_U8) {
        off = base[q] + (unsigned)aa_patch_off(p, k < k_end ? k : 0);
      } else if constexpr (KIND == AA_KIND_D_DENSE) {
        off = base[q] + 4u * (unsigned)k0 * (unsigned)ld;
      } else if constexpr (KIND == AA_KIND_D_PATCHT) {
        off = base[q] + 4u * (unsigned)aa_pix_base(p, k < k_end ? k : 0);
      } else {
        off = base[q] + (unsigned)aa_pix_base(p, k < k_end ? k : 0);
      }
      // lanes whose base is already >= AA_OOB stay out of range (operand spans are < 2^31)
      off = k < k_end ? off : AA_OOB;
      aa_dma16(off, r, stage + (unsigned)(q * 4 + wave) * 1024u);
    }
  }

  // The four operand values of MFMA j = 0..3 for k-group c (k = 8c + 4*lh + j) of tile row/column
  // `x` (= tile origin + lane & 31).
  __device__ static inline void fetch(const GemmP& p, const char* stage, int x, int c, int lh,
                                      float (&v)[4]) {
    if constexpr (KIND == AA_KIND_T_DENSE || KIND == AA_KIND_T_PATCH) {
      const int kg = 2 * c + lh;
      const float4 t = *reinterpret_cast<const float4*>(
          stage + x * 128 + ((kg ^ ((x >> 1) & 7)) << 4));
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (KIND == AA_KIND_T_PATCH_U8) {
      const uint4 t = *reinterpret_cast<const uint4*>(stage + (((c >> 1) * BX + x) << 4));
      const uint32_t lo = (c & 1) ? t.z : t.x, hi = (c & 1) ? t.w : t.y;
      const uint32_t w = lh ? hi : lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = aa_u8_scale(w, j, p);
    } else if constexpr (KIND == AA_KIND_D_DENSE || KIND == AA_KIND_D_PATCHT) {
      const float* s = reinterpret_cast<const float*>(stage) + (8 * c + 4 * lh) * BX + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = s[j * BX];
    } else {
      const uint8_t* s = reinterpret_cast<const uint8_t*>(stage) + (8 * c + 4 * lh) * BX + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = aa_u8_scale((uint32_t)s[j * BX], 0, p);
    }
  }
};

template <int N>
__device__ static inline void aa_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef AA_GD_STAMPS       /* tools/fc1_probe.hip: per-workgroup wall_clock64 stamps (10 ns ticks) */
__device__ long long* d_gd_stamps = nullptr;
#define GD_STAMP(i)                                                              \
  if (d_gd_stamps != nullptr && threadIdx.x == 0)                                \
    d_gd_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64();
#else
#define GD_STAMP(i)
#endif

// One output tile (blk) of a contraction: the body of aa_gemm_dma_kernel.
template <int AK, int BKIND, int BM, int BN, int WGM, int WGN, int WGK, int NS>
__device__ static inline void aa_gemm_dma_tile(const GemmP& p, const AaBlk& blk) {
  static_assert(WGM * WGN * WGK == 4, "4 waves per workgroup");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  using OA = DmaOp<AK, BM>;
  using OB = DmaOp<BKIND, BN>;
  constexpr int STAGE = OA::kPadded + OB::kPadded;
  constexpr int NIW = OA::kNIW + OB::kNIW;           // DMA instructions per wave per K-tile
  static_assert((NS - 1) * NIW <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);

  GD_STAMP(0)
  const int m0 = blk.x * BM;
  const int n0 = blk.y * BN;
  const int k_begin = blk.z * p.k_per_split;
  int k_end = k_begin + p.k_per_split;
  if (k_end > p.K) k_end = p.K;
  const int nk = (k_end - k_begin + AA_BK - 1) / AA_BK;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wk = wave / (WGM * WGN);
  const int wmn = wave - wk * (WGM * WGN);
  const int wm = wmn / WGN, wn = wmn % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  const aa_i32x4 rA = aa_make_desc(p.A, p.a_bytes);
  const aa_i32x4 rB = aa_make_desc(p.B, p.b_bytes);
  const unsigned lds0 = aa_lds_addr(lds);
  OA la;
  OB lb;
  la.init(p, p.lda, m0, p.M, wave, lane);
  lb.init(p, p.ldb, n0, p.N, wave, lane);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool do_colsum = (BKIND == AA_KIND_D_DENSE) && p.colsum_out != nullptr && blk.x == 0;
  float csum = 0.f;

  // Activation-derivative mask of the epilogue (dX GEMMs): its 16 values per lane are requested
  // HERE, before the first DMA, so the round trip (it was the first dependent access of the
  // epilogue: 4.1 us of epilogue on fc1.dX in the in-kernel timeline) rides under the k loop.
  // Branch-free on purpose -- a load under a condition is waited for at the join: clamped
  // in-range indices, or element 0 of the B operand when the launch has no mask.  One-tile waves
  // only (16 registers held across the loop).
  constexpr bool kPreMask = TM == 1 && TN == 1;
  float mk[16];
  if constexpr (kPreMask) {
    const bool use = p.splits <= 1 && p.mask_kind != 0;
    const float* msrc = use ? p.mask_src : p.B;
    const int nq = n0 + wn * 32 + l31;
    const int nc = nq < p.N ? nq : p.N - 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int mq = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int mc = mq < p.M ? mq : p.M - 1;
      mk[e] = msrc[use ? (size_t)mc * p.ldm + nc : (size_t)0];
    }
  }

  auto issue_tile = [&](int t) {  // K-tile t -> ring slot t % NS (tiles past nk are all-zero)
    const unsigned st = lds0 + (unsigned)(t % NS) * STAGE;
    const int k0 = k_begin + t * AA_BK;
    la.issue(p, rA, p.lda, st, k0, k_end, wave);
    lb.issue(p, rB, p.ldb, st + OA::kPadded, k0, k_end, wave);
  };

  // prologue: NS-1 tiles in flight
#pragma unroll
  for (int t = 0; t < NS - 1; ++t) issue_tile(t);

  constexpr int NCH = 4 / WGK;   // 8-k chunks of a K-tile this wave multiplies
  auto colsum_tile = [&](const char* sb) {
    if constexpr (BKIND == AA_KIND_D_DENSE) {
      if (do_colsum && threadIdx.x < BN) {  // fixed k order -> deterministic bias gradient
        const float* col = reinterpret_cast<const float*>(sb) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < AA_BK; ++k) csum += col[k * BN];
      }
    }
  };
  {
    // Software-pipelined k loop: the operands of tile t+1 travel LDS -> registers while the MFMAs
    // of tile t run.  The in-order form it replaced (wait, barrier, ds_read, MFMA) exposed barrier
    // skew + LDS latency once per tile with one wave per SIMD: in-kernel timeline 0.83 us per tile
    // against 0.43 us of MFMA issue on fc1.fwd (tools/fc1_probe.hip: k loop 9.6 -> 7.6 us, launch
    // 15.2 -> 13.9 us; fc1.dX 18.1 -> 17.2, fc1.dW 17.1 -> 16.5); the same DMA ring with the MFMAs
    // fed from registers alone runs at the MFMA rate (tools/fc1_mem_probe.hip: 9.8 us, of which
    // 6.1 us are the 196 fp32 MFMAs per SIMD at the ~2.05 GHz the matrix pipe sustains).  In the
    // training loop the change is neutral (0.3535 vs 0.3546 ms, three alternating pairs).
    float ra[2][NCH][TM][4], rb[2][NCH][TN][4];
    auto stage = [&](int t, int buf) {   // tile t has landed -> registers of `buf`
      aa_wait_vmcnt<(NS - 2) * NIW>();
      // every wave is done READING the slot the DMA below overwrites (tile t-1's, fetched one
      // step ago) -- its ds_reads have retired -- and tile t has landed for every wave
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_tile(t + NS - 1);
      const char* sa = lds + (t % NS) * STAGE;
      const char* sb = sa + OA::kPadded;
      if (t < nk) colsum_tile(sb);
#pragma unroll
      for (int cq = 0; cq < NCH; ++cq) {
        const int c = cq * WGK + wk;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          OA::fetch(p, sa, wm * (TM * 32) + 32 * i + l31, c, lh, ra[buf][cq][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          OB::fetch(p, sb, wn * (TN * 32) + 32 * j + l31, c, lh, rb[buf][cq][j]);
      }
    };
    auto multiply = [&](int buf) {
#pragma unroll
      for (int cq = 0; cq < NCH; ++cq)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[buf][cq][i][jj], rb[buf][cq][j][jj],
                                                               acc[i][j], 0, 0, 0);
    };
    stage(0, 0);
    GD_STAMP(1)
    int t = 0;
    for (; t + 2 <= nk; t += 2) {
      stage(t + 1, 1);
      multiply(0);
      stage(t + 2, 0);      // (tile nk is the all-zero tail tile: staged, never multiplied)
      multiply(1);
    }
    if (t < nk) multiply(0);
  }
  GD_STAMP(2)
  aa_wait_vmcnt<0>();  // drain the (all-zero) tail DMAs before the LDS is reused
  __syncthreads();

  const bool raw = p.splits > 1;
  if constexpr (BKIND == AA_KIND_D_DENSE) {
    if (do_colsum && threadIdx.x < BN) {
      const int n = n0 + threadIdx.x;
      if (n < p.N) {
        if (raw)
          p.C[(size_t)p.splits * p.M * p.N + (size_t)blk.z * p.N + n] = csum;
        else
          p.colsum_out[n] = csum;
      }
    }
  }

  if constexpr (WGK > 1) {
    float* red = smem;  // [(WGK-1)][WGM*WGN][TM*TN][16][64]
    constexpr int PER = TM * TN * 1024;
    if (wk > 0) {
      float* dst = red + ((wk - 1) * (WGM * WGN) + wmn) * PER + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((i * TN + j) * 16 + e) * 64] = acc[i][j][e];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
    for (int w = 1; w < WGK; ++w) {
      const float* src = red + ((w - 1) * (WGM * WGN) + wmn) * PER + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += src[((i * TN + j) * 16 + e) * 64];
    }
    }
  }

  float* C = raw ? p.C + (size_t)blk.z * (size_t)p.M * (size_t)p.N : p.C;
  const int ldc = raw ? p.N : p.ldc;
  if (WGK == 1 || wk == 0) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (TN * 32) + 32 * j + l31;
      if (n >= p.N) continue;
      const float bv = (!raw && p.bias != nullptr) ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = m0 + wm * (TM * 32) + 32 * i + row;
        if (m >= p.M) continue;
        float v = acc[i][j][e];
        if (!raw) {
          v = aa_act(v + bv, p.act);
          if (p.mask_kind != 0) {
            if constexpr (kPreMask)
              v *= aa_actgrad(mk[e], p.mask_kind);
            else
              v *= aa_actgrad(p.mask_src[(size_t)m * p.ldm + n], p.mask_kind);
          }
        }
        C[(size_t)m * ldc + n] = v;
      }
    }
  }
  }
  GD_STAMP(3)
}

template <int AK, int BKIND, int BM, int BN, int WGM, int WGN, int WGK, int NS>
__global__ void __launch_bounds__(AA_GEMM_THREADS) aa_gemm_dma_kernel(GemmP p) {
  AaBlk blk;
  if (!aa_block_of(p, &blk)) return;
  aa_gemm_dma_tile<AK, BKIND, BM, BN, WGM, WGN, WGK, NS>(p, blk);
}

template <int AK, int BKIND, int BM, int BN, int WGM, int WGN, int WGK, int NS>
static void aa_gemm_dma_launch(const GemmP& p, int splits, hipStream_t st) {
  using OA = DmaOp<AK, BM>;
  using OB = DmaOp<BKIND, BN>;
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, splits);
  if (p.xcd_mode != 0) grid = dim3(((p.gx * p.gy * p.gz + 7) / 8) * 8, 1, 1);
  size_t smem = (size_t)NS * (OA::kPadded + OB::kPadded);
  const size_t red = (size_t)(WGK - 1) * WGM * WGN * TM * TN * 1024 * sizeof(float);
  if (red > smem) smem = red;
  hipLaunchKernelGGL((aa_gemm_dma_kernel<AK, BKIND, BM, BN, WGM, WGN, WGK, NS>), grid,
                     dim3(AA_GEMM_THREADS), smem, st, p);
}
