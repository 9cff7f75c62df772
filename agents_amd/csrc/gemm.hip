// fp32 MFMA GEMM for gfx950 with pluggable operand loaders (dense row/col-major, NHWC conv
// patches as implicit-GEMM A operand in both orientations, uint8 frames with fused /255).
//
//   C[M][N] = epilogue( sum_k A(m,k) * B(k,n) )
//
// This one kernel family carries every contraction of the trainer hot path:
//   Dense forward            A_ROW      x B_ROW   (keras Dense: y = x W + b;  networks/q_network.py:46-158)
//   Dense dX                 A_ROW      x B_COL   (dZ W^T)
//   Dense dW                 A_COL      x B_ROW   (X^T dZ)
//   Conv2D forward (VALID)   A_PATCH*   x B_ROW   (implicit im2col of NHWC input; kernel HWIO = B[K][N])
//   Conv2D dW                A_PATCH_T* x B_ROW   (patch^T dZ, reduction over output pixels, split-K)
//   Conv2D dCol              A_ROW      x B_COL   (dZ W^T, then col2im in nn.hip)
// (reference arithmetic: tf.keras Dense/Conv2D under tf.GradientTape,
//  tf_agents/agents/dqn/dqn_agent.py:412-449, examples/dqn/mnih15/dqn_train_eval_atari.py:80-112)
//
// Machine mapping (MI355X_MICROARCH.md): v_mfma_f32_32x32x2_f32 (exact fp32, 64 cyc/SIMD,
// 157 TFLOP/s chip peak).  256 threads = 4 waves, each wave owns TM x TN 32x32 accumulator
// tiles.  Both operands are staged K-major in LDS (As[k][m], Bs[k][n]) so every MFMA operand
// fetch is a conflict-free stride-1 ds_read_b32 (lane l reads k = l>>5, m|n = l&31).  Global
// loads are 16-byte vectors; operands whose global layout is K-contiguous are transposed on the
// LDS write (row pitch BM+1 -> conflict-free ds_write_b32), M/N-contiguous operands are copied
// with ds_write_b128 (row pitch BM+4).  Double-buffered LDS + register prefetch of tile t+1
// during the MFMAs of tile t, one barrier per K-step (BK = 32).  Split-K over blockIdx.z writes
// fp32 partial slabs that aa_splitk_reduce_kernel sums deterministically with the epilogue.
#include "common.h"
#include <stdlib.h>
#include "agents_amd.h"
#include "splitk_reduce.h"

#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, typename F, int... I>
__device__ static inline void aa_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ static inline void aa_static_for(F&& f) {
  aa_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

#define AA_GEMM_THREADS 256
#define AA_BK 32

struct GemmP {
  const void* A;
  const float* B;
  float* C;          // output, or slab base when splits > 1
  int M, N, K;
  int lda, ldb, ldc;
  // conv patch geometry (NHWC input [Bimg, H, W, Cin])
  int W, Cin, OW, OHW, stride, seg, rowpitch, imgpitch;
  float a_div;
  float a_rcp;       // 1/a_div
  int a_fast;        // (float)u8 / a_div == fma-refined product for all 256 bytes (host-verified)
  unsigned magic_ohw, magic_ow, magic_seg;  // ceil(2^32/d): n / d == mulhi(n, magic) (host-verified range)
  int fastdiv;
  int use_dma;       // operands allow the LDS-DMA main loop (16-byte granules everywhere)
  unsigned a_bytes, b_bytes;  // operand extents for the buffer resources (< 2^31, host-checked)
  float* colsum_out; // nullable: sum_k B(k,n) (bias gradient fused into the dW GEMM)
  int k_per_split;
  int splits;
  const float* bias;
  int act;
  const float* mask_src;
  int ldm;
  int mask_kind;
  int a_vec, b_vec;  // 16-byte vector path usable for dense operands
  // XCD-aware block order (aa_block_of): 0 = the launch's own (x, y, z); 1 / 2 / 3 = a 1-D launch
  // whose blocks are dealt to the 8 XCDs in contiguous runs of the K-split / N-tile / M-tile index
  int xcd_mode, gx, gy, gz;
};

// Which (M tile, N tile, K split) a workgroup computes.  Hardware deals consecutive workgroups to
// consecutive XCDs, each with its own L2: in launch order (x fastest) every XCD ends up touching
// ALL of the operand that x does not index -- fc1 forward (8 M tiles x 8 N tiles x 8 K splits)
// fetched W once per XCD, 59 MB for 9.6 MB of operands (rocprofv3 FETCH_SIZE/WRITE_SIZE).  With
// xcd_mode != 0 the launch is 1-D, XCD c = block & 7 takes the contiguous index range
// [c * per, (c + 1) * per), and the index is decomposed with the dimension that partitions the
// most operand bytes slowest, so an XCD's L2 sees one slice of it.  Pure relabelling: every
// (x, y, z) is computed exactly once by the same code, results are bit-identical.
struct AaBlk { int x, y, z; };
// L < 0: the launch's own block index (3-D for xcd_mode 0).  L >= 0: this workgroup's index inside
// a 1-D range; xcd_mode 0 is then the launch order written out (x fastest).
__device__ static inline bool aa_block_of(const GemmP& p, AaBlk* b, int L = -1) {
  if (p.xcd_mode == 0) {
    if (L < 0) {
      b->x = blockIdx.x; b->y = blockIdx.y; b->z = blockIdx.z;
      return true;
    }
    if (L >= p.gx * p.gy * p.gz) return false;
    b->x = L % p.gx;
    const int r = L / p.gx;
    b->y = r % p.gy;
    b->z = r / p.gy;
    return true;
  }
  const int n = p.gx * p.gy * p.gz;
  const int per = (n + 7) >> 3;
  if (L < 0) L = blockIdx.x;
  const int idx = (L & 7) * per + (L >> 3);
  if (idx >= n) return false;
  if (p.xcd_mode == 1) {
    const int t = p.gx * p.gy;
    b->z = idx / t;
    const int r = idx - b->z * t;
    b->y = r / p.gx;
    b->x = r - b->y * p.gx;
  } else if (p.xcd_mode == 2) {
    const int t = p.gx * p.gz;
    b->y = idx / t;
    const int r = idx - b->y * t;
    b->z = r / p.gx;
    b->x = r - b->z * p.gx;
  } else {
    const int t = p.gy * p.gz;
    b->x = idx / t;
    const int r = idx - b->x * t;
    b->z = r / p.gy;
    b->y = r - b->z * p.gy;
  }
  return true;
}

// Pixel index (b, oy, ox) -> element offset of the patch origin in the NHWC input.
__device__ static inline int aa_pix_base(const GemmP& p, int pix) {
  int b, oy;
  if (p.fastdiv) {
    b = (int)__umulhi((unsigned)pix, p.magic_ohw);
    oy = (int)__umulhi((unsigned)(pix - b * p.OHW), p.magic_ow);
  } else {
    b = pix / p.OHW;
    oy = (pix - b * p.OHW) / p.OW;
  }
  const int rem = pix - b * p.OHW;
  const int ox = rem - oy * p.OW;
  return b * p.imgpitch + (oy * p.stride) * p.rowpitch + (ox * p.stride) * p.Cin;
}
// Patch element index k -> element offset relative to the patch origin.
__device__ static inline int aa_patch_off(const GemmP& p, int k) {
  const int ky = p.fastdiv ? (int)__umulhi((unsigned)k, p.magic_seg) : k / p.seg;
  return ky * p.rowpitch + (k - ky * p.seg);
}

// ------------------------------------------------------------------------------------------
// Staging: "T" = global is K-contiguous -> transpose on LDS write; "D" = global is X-contiguous
// (X = M for A, N for B) -> direct vector copy.  Each returns/consumes a small register array.
// ------------------------------------------------------------------------------------------
template <int BX>
struct StageT {  // float source, K-contiguous.  8 float4 per row (BK=32), 32 rows per pass.
  static constexpr int PASSES = BX / 32;
  float4 r[PASSES];
};
template <int BX>
struct StageD {  // float source, X-contiguous. BX/4 vectors per k-row.
  static constexpr int V = BX / 4;
  static constexpr int RP = AA_GEMM_THREADS / V;
  static constexpr int PASSES = (AA_BK + RP - 1) / RP;
  float4 r[PASSES];
};
struct StageU8 {  // one 16-byte vector of uint8 per thread
  uint4 r;
};

// All loaders are branch-free buffer loads: each operand is addressed through a buffer resource
// (base, byte size) with a 32-bit byte offset per lane; an element outside the operand gets the
// offset AA_OOB, which the hardware range check turns into a zero result.  No divergent branch
// surrounds a load, so hipcc keeps every load of a K-tile in flight together and counts them with
// partial vmcnt waits (branches around loads make it drain vmcnt to 0 at each join).  `vec`
// (wave-uniform) says the operand allows 16-byte loads: base 16-byte aligned, leading dimension
// and the contiguous extent multiples of 4 -- then a vector is entirely inside or outside.
#define AA_OOB 0x80000000u
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t aa_rsrc;

__device__ static inline aa_rsrc aa_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ static inline float4 aa_ld4(aa_rsrc r, unsigned byte_off, bool ok) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? byte_off : AA_OOB, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                     __uint_as_float(v.w));
}
__device__ static inline float aa_ld1(aa_rsrc r, unsigned byte_off, bool ok) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ok ? byte_off : AA_OOB, 0, 0));
}
__device__ static inline uint4 aa_ld16b(aa_rsrc r, unsigned byte_off, bool ok) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? byte_off : AA_OOB, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- dense K-contiguous (A_ROW for A, B_COL for B): elem(x,k) = base[x*ld + k] ------------
template <int BX, int VEC>
__device__ static inline void load_T_dense(StageT<BX>& s, aa_rsrc base, int ld, int x0,
                                           int X, int k0, int k_end) {
  const int kq = threadIdx.x & 7, r = threadIdx.x >> 3;
  const int k = k0 + 4 * kq;
  if constexpr (VEC) {
#pragma unroll
    for (int p = 0; p < StageT<BX>::PASSES; ++p) {
      const int x = x0 + r + 32 * p;
      s.r[p] = aa_ld4(base, 4u * ((unsigned)x * ld + k), x < X && k < k_end);
    }
  } else {
#pragma unroll
    for (int p = 0; p < StageT<BX>::PASSES; ++p) {
      const int x = x0 + r + 32 * p;
      const unsigned o = 4u * ((unsigned)x * ld + k);
      const bool in = x < X;
      s.r[p].x = aa_ld1(base, o + 0, in && k + 0 < k_end);
      s.r[p].y = aa_ld1(base, o + 4, in && k + 1 < k_end);
      s.r[p].z = aa_ld1(base, o + 8, in && k + 2 < k_end);
      s.r[p].w = aa_ld1(base, o + 12, in && k + 3 < k_end);
    }
  }
}
template <int BX, int LDS_LD>
__device__ static inline void store_T(const StageT<BX>& s, float* tile) {
  const int kq = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
  for (int p = 0; p < StageT<BX>::PASSES; ++p) {
    float* d = tile + (4 * kq) * LDS_LD + r + 32 * p;
    d[0 * LDS_LD] = s.r[p].x;
    d[1 * LDS_LD] = s.r[p].y;
    d[2 * LDS_LD] = s.r[p].z;
    d[3 * LDS_LD] = s.r[p].w;
  }
}

// ---- dense X-contiguous (A_COL for A, B_ROW for B): elem(x,k) = base[k*ld + x] ------------
template <int BX, int VEC>
__device__ static inline void load_D_dense(StageD<BX>& s, aa_rsrc base, int ld, int x0,
                                           int X, int k0, int k_end) {
  constexpr int V = StageD<BX>::V, RP = StageD<BX>::RP;
  const int v4 = threadIdx.x % V, r = threadIdx.x / V;
  const int x = x0 + 4 * v4;
  if constexpr (VEC) {
#pragma unroll
    for (int p = 0; p < StageD<BX>::PASSES; ++p) {
      const int kk = r + RP * p;
      const int k = k0 + kk;
      s.r[p] = aa_ld4(base, 4u * ((unsigned)k * ld + x), kk < AA_BK && k < k_end && x < X);
    }
  } else {
#pragma unroll
    for (int p = 0; p < StageD<BX>::PASSES; ++p) {
      const int kk = r + RP * p;
      const int k = k0 + kk;
      const unsigned o = 4u * ((unsigned)k * ld + x);
      const bool in = kk < AA_BK && k < k_end;
      s.r[p].x = aa_ld1(base, o + 0, in && x + 0 < X);
      s.r[p].y = aa_ld1(base, o + 4, in && x + 1 < X);
      s.r[p].z = aa_ld1(base, o + 8, in && x + 2 < X);
      s.r[p].w = aa_ld1(base, o + 12, in && x + 3 < X);
    }
  }
}
template <int BX, int LDS_LD>
__device__ static inline void store_D(const StageD<BX>& s, float* tile) {
  constexpr int V = StageD<BX>::V, RP = StageD<BX>::RP;
  const int v4 = threadIdx.x % V, r = threadIdx.x / V;
#pragma unroll
  for (int p = 0; p < StageD<BX>::PASSES; ++p) {
    const int kk = r + RP * p;
    if (kk < AA_BK) *reinterpret_cast<float4*>(tile + kk * LDS_LD + 4 * v4) = s.r[p];
  }
}

// (float)byte / div, bit-identical to the IEEE quotient: q0 = x*r; q = fma(fma(-d, q0, x), r, q0)
// when the host has verified that for every byte (a_fast), else a true division.
__device__ static inline float aa_u8_scale(uint32_t word, int j, const GemmP& p) {
  const float x = (float)((word >> (8 * j)) & 0xffu);
  if (p.a_fast) {
    const float q0 = x * p.a_rcp;
    return fmaf(fmaf(-p.a_div, q0, x), p.a_rcp, q0);
  }
  return x / p.a_div;
}

// ---- conv patches, forward orientation: A(m = pixel, k = patch element), K-contiguous -------
template <int BX>
__device__ static inline void load_T_patch(StageT<BX>& s, const GemmP& p, aa_rsrc A,
                                           const int* rowbase, int k0, int k_end) {
  const int kq = threadIdx.x & 7;
  const int k = k0 + 4 * kq;
  const bool kin = k < k_end;  // K % 4 == 0 validated on host
  const int koff = aa_patch_off(p, kin ? k : 0);
#pragma unroll
  for (int q = 0; q < StageT<BX>::PASSES; ++q) {
    const bool ok = kin && rowbase[q] >= 0;
    s.r[q] = aa_ld4(A, 4u * (unsigned)(rowbase[q] + koff), ok);
  }
}
// uint8 frames: 2 x 16-byte vectors per pixel row per K-step; thread -> (row = t>>1, half = t&1)
template <int BX>
__device__ static inline void load_T_patch_u8(StageU8& s, const GemmP& p, aa_rsrc A, int rowbase,
                                              int k0, int k_end) {
  const int kq = threadIdx.x & 1;
  const int k = k0 + 16 * kq;
  const bool ok = k < k_end && rowbase >= 0;
  s.r = aa_ld16b(A, (unsigned)(rowbase + aa_patch_off(p, ok ? k : 0)), ok);
}
template <int BX, int LDS_LD>
__device__ static inline void store_T_patch_u8(const StageU8& s, float* tile, const GemmP& p) {
  const int kq = threadIdx.x & 1, r = threadIdx.x >> 1;
  if (r >= BX) return;
  float* d = tile + (16 * kq) * LDS_LD + r;
  const uint32_t w[4] = {s.r.x, s.r.y, s.r.z, s.r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d[(4 * i + j) * LDS_LD] = aa_u8_scale(w[i], j, p);
    }
  }
}

// ---- conv patches, weight-grad orientation: A'(i = patch element, kk = pixel), i-contiguous --
template <int BX>
__device__ static inline void load_D_patchT(StageD<BX>& s, const GemmP& p, aa_rsrc A, int ioff,
                                            bool iin, int k0, int k_end) {
  constexpr int V = StageD<BX>::V, RP = StageD<BX>::RP;
  const int r = threadIdx.x / V;
#pragma unroll
  for (int q = 0; q < StageD<BX>::PASSES; ++q) {
    const int kk = r + RP * q;
    const int pix = k0 + kk;
    const bool ok = iin && kk < AA_BK && pix < k_end;
    s.r[q] = aa_ld4(A, 4u * (unsigned)(aa_pix_base(p, ok ? pix : 0) + ioff), ok);
  }
}
// uint8: 16 patch elements per vector; BX/16 vectors per pixel row.
template <int BX>
__device__ static inline void load_D_patchT_u8(StageU8& s, const GemmP& p, aa_rsrc A, int ioff,
                                               bool iin, int k0, int k_end) {
  constexpr int V = BX / 16;
  const int r = threadIdx.x / V;
  const int pix = k0 + r;
  const bool ok = iin && r < AA_BK && pix < k_end;
  s.r = aa_ld16b(A, (unsigned)(aa_pix_base(p, ok ? pix : 0) + ioff), ok);
}
template <int BX, int LDS_LD>
__device__ static inline void store_D_patchT_u8(const StageU8& s, float* tile, const GemmP& p) {
  constexpr int V = BX / 16;
  const int v16 = threadIdx.x % V, r = threadIdx.x / V;
  if (r >= AA_BK) return;
  float* d = tile + r * LDS_LD + 16 * v16;
  const uint32_t w[4] = {s.r.x, s.r.y, s.r.z, s.r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 f;
    f.x = aa_u8_scale(w[i], 0, p);
    f.y = aa_u8_scale(w[i], 1, p);
    f.z = aa_u8_scale(w[i], 2, p);
    f.w = aa_u8_scale(w[i], 3, p);
    *reinterpret_cast<float4*>(d + 4 * i) = f;
  }
}

#include "gemm_dma.h"

// ------------------------------------------------------------------------------------------
template <int AM, int BMODE, int BM, int BN, int WGM, int WGN, int WGK, int PD, int VEC>
__global__ void __launch_bounds__(AA_GEMM_THREADS) aa_gemm_kernel(GemmP p) {
  static_assert(WGM * WGN * WGK == 4, "4 waves per workgroup");
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  constexpr bool A_IS_T = (AM == AA_A_ROW || AM == AA_A_PATCH || AM == AA_A_PATCH_U8);
  constexpr bool B_IS_T = (BMODE == AA_B_COL);
  constexpr int LDA_S = BM + (A_IS_T ? 1 : 4);
  constexpr int LDB_S = BN + (B_IS_T ? 1 : 4);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BK][LDA_S]
  float* Bs = smem + 2 * AA_BK * LDA_S;      // [2][BK][LDB_S]

  AaBlk blk;
  if (!aa_block_of(p, &blk)) return;
  const int m0 = blk.x * BM;
  const int n0 = blk.y * BN;
  const int k_begin = blk.z * p.k_per_split;
  int k_end = k_begin + p.k_per_split;
  if (k_end > p.K) k_end = p.K;
  const int nk = (k_end - k_begin + AA_BK - 1) / AA_BK;

  // ---- per-thread loader state --------------------------------------------------------
  // PD register stages: the loads of K-tile t+PD are issued while tile t is multiplied, so a
  // global load has PD K-steps of MFMA work to land before its ds_write needs it.
  StageT<BM> aT[PD];
  StageD<BM> aD[PD];
  StageU8 aU[PD];
  StageT<BN> bT[PD];
  StageD<BN> bD[PD];
  int rowbase[StageT<BM>::PASSES];  // A_PATCH: patch origin per staged row (-1 = out of range)
  int rowbase_u8 = -1;
  int ioff = 0;
  bool iin = false;
  if constexpr (AM == AA_A_PATCH) {
    const int r = threadIdx.x >> 3;
#pragma unroll
    for (int q = 0; q < StageT<BM>::PASSES; ++q) {
      const int m = m0 + r + 32 * q;
      rowbase[q] = m < p.M ? aa_pix_base(p, m) : -1;
    }
  }
  if constexpr (AM == AA_A_PATCH_U8) {
    const int r = threadIdx.x >> 1;
    const int m = m0 + r;
    rowbase_u8 = (r < BM && m < p.M) ? aa_pix_base(p, m) : -1;
  }
  if constexpr (AM == AA_A_PATCH_T) {
    const int i = m0 + 4 * (threadIdx.x % StageD<BM>::V);
    iin = i < p.M;
    ioff = iin ? aa_patch_off(p, i) : 0;
  }
  if constexpr (AM == AA_A_PATCH_T_U8) {
    const int i = m0 + 16 * (threadIdx.x % (BM / 16));
    iin = i < p.M;
    ioff = iin ? aa_patch_off(p, i) : 0;
  }
  // fused bias gradient: column sums of the B operand, taken by the first M-tile's workgroups
  const bool do_colsum = (BMODE == AA_B_ROW) && p.colsum_out != nullptr && blk.x == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

  const aa_rsrc rA = aa_make_rsrc(p.A, p.a_bytes);
  const aa_rsrc rB = aa_make_rsrc(p.B, p.b_bytes);
  auto load_tiles = [&](auto slot, int k0) {
    constexpr int u = decltype(slot)::value;
    if constexpr (AM == AA_A_ROW)
      load_T_dense<BM, VEC>(aT[u], rA, p.lda, m0, p.M, k0, k_end);
    else if constexpr (AM == AA_A_COL)
      load_D_dense<BM, VEC>(aD[u], rA, p.lda, m0, p.M, k0, k_end);
    else if constexpr (AM == AA_A_PATCH)
      load_T_patch<BM>(aT[u], p, rA, rowbase, k0, k_end);
    else if constexpr (AM == AA_A_PATCH_U8)
      load_T_patch_u8<BM>(aU[u], p, rA, rowbase_u8, k0, k_end);
    else if constexpr (AM == AA_A_PATCH_T)
      load_D_patchT<BM>(aD[u], p, rA, ioff, iin, k0, k_end);
    else
      load_D_patchT_u8<BM>(aU[u], p, rA, ioff, iin, k0, k_end);
    if constexpr (BMODE == AA_B_ROW)
      load_D_dense<BN, VEC>(bD[u], rB, p.ldb, n0, p.N, k0, k_end);
    else
      load_T_dense<BN, VEC>(bT[u], rB, p.ldb, n0, p.N, k0, k_end);
  };
  auto store_tiles = [&](auto slot, int buf) {
    constexpr int u = decltype(slot)::value;
    float* at = As + buf * AA_BK * LDA_S;
    float* bt = Bs + buf * AA_BK * LDB_S;
    if constexpr (AM == AA_A_ROW || AM == AA_A_PATCH)
      store_T<BM, LDA_S>(aT[u], at);
    else if constexpr (AM == AA_A_COL || AM == AA_A_PATCH_T)
      store_D<BM, LDA_S>(aD[u], at);
    else if constexpr (AM == AA_A_PATCH_U8)
      store_T_patch_u8<BM, LDA_S>(aU[u], at, p);
    else
      store_D_patchT_u8<BM, LDA_S>(aU[u], at, p);
    if constexpr (BMODE == AA_B_ROW) {
      store_D<BN, LDB_S>(bD[u], bt);
      // the registers hold exactly one K-tile of B: add it once, here (unconditionally -- a
      // branch in the K loop would cost more than the adds; only do_colsum groups publish it)
#pragma unroll
      for (int q = 0; q < StageD<BN>::PASSES; ++q) {
        csum.x += bD[u].r[q].x; csum.y += bD[u].r[q].y;
        csum.z += bD[u].r[q].z; csum.w += bD[u].r[q].w;
      }
    } else {
      store_T<BN, LDB_S>(bT[u], bt);
    }
  };

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wk = wave / (WGM * WGN);           // k-slice owner (intra-workgroup split-K)
  const int wmn = wave - wk * (WGM * WGN);
  const int wm = wmn / WGN, wn = wmn % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto mma_tile = [&](int buf) {
    const float* at = As + buf * AA_BK * LDA_S + wm * (TM * 32) + l31;
    const float* bt = Bs + buf * AA_BK * LDB_S + wn * (TN * 32) + l31;
#pragma unroll
    for (int kq = 0; kq < AA_BK / 2 / WGK; ++kq) {
      const int k = 2 * (kq * WGK + wk) + lh;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = at[k * LDA_S + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bt[k * LDB_S + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  // prologue: tiles 0 .. PD-1 in flight, tile 0 staged in LDS buffer 0.  Loads past the last
  // K-tile are issued anyway: every lane is out of range, the hardware returns zeros without
  // touching memory, and the K loop stays free of conditionals (so hipcc counts vmcnt exactly).
  aa_static_for<PD>([&](auto u) { load_tiles(u, k_begin + u.value * AA_BK); });
  store_tiles(std::integral_constant<int, 0>{}, 0);
  __syncthreads();

  // step t (register slot u = t % PD): slot u held tile t (already in LDS) and is refilled with
  // tile t + PD; tile t + 1 lives in slot (u + 1) % PD and is staged after the MFMAs of tile t.
  auto step = [&](auto u, int t) {
    const int buf = t & 1;
    load_tiles(u, k_begin + (t + PD) * AA_BK);
    mma_tile(buf);
    store_tiles(std::integral_constant<int, (u.value + 1) % PD>{}, buf ^ 1);
    __syncthreads();
  };
  if (nk > 0) {
    int t = 0;
    while (true) {
      step(std::integral_constant<int, 0>{}, t);
      if (++t >= nk) break;
      if constexpr (PD > 1) {
        step(std::integral_constant<int, 1 % PD>{}, t);
        if (++t >= nk) break;
      }
      if constexpr (PD > 2) {
        step(std::integral_constant<int, 2 % PD>{}, t);
        if (++t >= nk) break;
      }
    }
  }

  const bool raw = p.splits > 1;

  // ---- fused bias gradient: reduce the per-thread column sums over the k-rows -------------
  if constexpr (BMODE == AA_B_ROW) {
    if (do_colsum) {  // workgroup-uniform
      constexpr int V = StageD<BN>::V, RP = StageD<BN>::RP;
      float* red = smem;  // [RP][BN], staging tiles are dead after the loop's last barrier
      const int v4 = threadIdx.x % V, r = threadIdx.x / V;
      *reinterpret_cast<float4*>(red + r * BN + 4 * v4) = csum;
      __syncthreads();
      if (threadIdx.x < BN) {
        float s = 0.f;
        for (int j = 0; j < RP; ++j) s += red[j * BN + threadIdx.x];
        const int n = n0 + threadIdx.x;
        if (n < p.N) {
          if (raw)  // per-split partial rows after the slabs: [splits][N]
            p.C[(size_t)p.splits * p.M * p.N + (size_t)blk.z * p.N + n] = s;
          else
            p.colsum_out[n] = s;
        }
      }
      __syncthreads();
    }
  }

  // ---- intra-workgroup split-K: waves wk > 0 hand their accumulators to wave wk == 0 -------
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

  // ---- epilogue ------------------------------------------------------------------------
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
          if (p.mask_kind != 0) v *= aa_actgrad(p.mask_src[(size_t)m * p.ldm + n], p.mask_kind);
        }
        C[(size_t)m * ldc + n] = v;
      }
    }
  }
  }
}

// ------------------------------------------------------------------------------------------
// host side: shape-driven tile / split-K selection and dispatch
// ------------------------------------------------------------------------------------------
struct AaTileCfg {
  int bm, bn, wgm, wgn, wgk;
};
// force_cfg - 1 indexes this table
static const AaTileCfg kCfgs[] = {
    {128, 64, 2, 2, 1},   // 1
    {128, 32, 4, 1, 1},   // 2
    {64, 64, 2, 2, 1},    // 3
    {128, 128, 2, 2, 1},  // 4
    {64, 32, 2, 1, 2},    // 5
    {32, 64, 1, 2, 2},    // 6
    {32, 32, 1, 1, 4},    // 7
    {256, 32, 4, 1, 1},   // 8  (LDS-DMA loop only: whole-M tile of the conv1 weight gradient)
};
#define AA_NCFG 8
#define AA_CFG_U8_BF16 9   // force_cfg value of the uint8 x bf16x3 conv forward (conv_u8_bf16.h)
// Tried and dropped (tools/gemm_sweep.py on MI355X): one 32x64 / 64x32 / 64x64 tile per wave with
// a 4-way intra-workgroup K split (two or four independent accumulator chains per wave) -- never
// faster than the plans above on the DQN shapes.  At these sizes a launch costs ~10 us of fixed
// time (dispatch gap, first loads, tail) and runs at a marginal ~115 TFLOP/s beyond that.
// Also tried and dropped: a persistent kernel for conv1 with the 32 KiB filter bank resident in LDS
// and barrier-free waves streaming 32-row tiles through private DMA rings: 42-47 us against 39 us
// for the tiled plan.  Its ablation (no stores / no DMA / no MFMA) put 27 of the 42 us outside the
// MFMAs -- launch 5, stores 7, A DMA 3, operand fetch from LDS + the exact uint8 -> float /255
// (4 VALU per element) ~11 -- which is the floor to attack next, not the tile shape.

#include "conv_u8_bf16.h"

struct AaGemmPlan {
  int cfg;
  int bm, bn;
  int splits, k_per_split;
  size_t ws_bytes;
};

// Operands regular enough for the LDS-DMA main loop (16-byte granules everywhere).
static bool aa_desc_dma_ok(const aa_gemm_desc* d) {
  const bool patch = d->a_mode >= AA_A_PATCH;
  const int a_contig = d->a_mode == AA_A_ROW ? d->K : d->M;
  const int b_contig = d->b_mode == AA_B_ROW ? d->N : d->K;
  const bool a_vec = !patch && d->lda % 4 == 0 && a_contig % 4 == 0 && (((uintptr_t)d->A & 15) == 0);
  const bool b_vec = d->ldb % 4 == 0 && b_contig % 4 == 0 && (((uintptr_t)d->B & 15) == 0);
  return b_vec && (a_vec || patch) && !d->no_dma && d->K > 128;
}

static int aa_gemm_plan(const aa_gemm_desc* d, AaGemmPlan* pl) {
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return AA_ERR_INVALID;
  const int64_t M = d->M, N = d->N, K = d->K;
  auto ntiles = [&](int c) {
    return ((M + kCfgs[c].bm - 1) / kCfgs[c].bm) * ((N + kCfgs[c].bn - 1) / kCfgs[c].bn);
  };
  // Shape-driven choice, fitted to tools/gemm_sweep.py runs on MI355X (256 CUs).  A launch of this
  // kernel has ~5 us of fixed latency (kernarg fetch -> first global loads -> LDS -> MFMA ->
  // stores retire), so what matters for these 1-2 GFLOP problems is that every CU gets two
  // workgroups of balanced length: take the largest tile that still yields >= 512 workgroups;
  // if even the smallest tile gives < 256, split K (partial slabs + one deterministic reduce).
  const bool narrow = N <= 32;
  static const int wide_c[] = {3, 2, 5};   // 128x128, 64x64, 32x64
  static const int thin_c[] = {1, 4, 6};   // 128x32, 64x32, 32x32
  const int* cand = narrow ? thin_c : wide_c;
  int cfg;
  // Weight gradient of a conv with few filters (conv1: M = 256 patch elements, N = 32, K = 102,400
  // pixels): with 32x32 tiles the dZ operand is re-read once per M-tile (PMC: 123 MB fetched
  // against 20 MB algorithmic), so one workgroup takes ALL of M and the pixels are split instead.
  const bool tall_reduce = d->a_mode == AA_A_PATCH_T_U8 &&   // (fp32 images: 72 KB of LDS ring)
                           N <= 32 && M > 128 && M <= 256 && K >= 64 * M && aa_desc_dma_ok(d);
  if (d->a_mode == AA_A_PATCH_U8 &&
      (d->force_cfg == AA_CFG_U8_BF16 || (d->force_cfg == 0 && !d->no_dma && aa_conv_u8_bf16_ok(d)))) {
    if (!aa_conv_u8_bf16_ok(d)) return AA_ERR_INVALID;
    pl->cfg = AA_CFG_U8_BF16 - 1;
    pl->bm = 256; pl->bn = 32;
    pl->splits = 1; pl->k_per_split = (int)K; pl->ws_bytes = 0;
    return AA_OK;
  }
  if (d->a_mode == AA_A_PATCH_T_U8 &&
      (d->force_cfg == AA_CFG_U8_BF16 ||
       (d->force_cfg == 0 && d->force_splits == 0 && !d->no_dma && aa_conv_u8_dw_bf16_ok(d)))) {
    if (!aa_conv_u8_dw_bf16_ok(d)) return AA_ERR_INVALID;
    pl->cfg = AA_CFG_U8_BF16 - 1;
    pl->bm = 256; pl->bn = 32;
    pl->splits = aa_conv_u8_dw_groups(d->n_img);
    pl->k_per_split = (int)((K + pl->splits - 1) / pl->splits);
    pl->ws_bytes = (size_t)pl->splits * (size_t)(M * N + (d->colsum_out ? N : 0)) * sizeof(float);
    return AA_OK;
  }
  // (Round 5: the dense bf16x6 plan -- gemm_x6d.h, force_cfg = 10 / AA_GEMM_X6D=1 -- is gone.  It won
  // in isolation (fc1 forward 18.8 vs 19.3 us, dX 17.0 vs 18.5, dW 15.1 vs 20.5) and lost inside the
  // DQN iteration in two rounds (0.4009 vs 0.3970 ms, 0.434 vs 0.423 ms): its k loop is
  // instruction-issue bound (~220 instructions per 24 MFMAs with the operand split inside it) and
  // its 72 KiB of LDS keep the other streams' workgroups off the CU.  git history has it.)
  if (d->force_cfg > 0) {
    cfg = d->force_cfg - 1;
    if (cfg >= AA_NCFG) return AA_ERR_INVALID;
    if (cfg == 7 && (!aa_desc_dma_ok(d) || d->a_mode != AA_A_PATCH_T_U8)) return AA_ERR_INVALID;
  } else if (tall_reduce) {
    cfg = 1;   // 128x32; measured on MI355X: 41 us with 256 splits (256x32 tile: 42-52 us)
  } else {
    cfg = cand[2];
    for (int i = 0; i < 3; ++i) {
      // the biggest tiles run one workgroup per CU; the middle one is also taken from 1.5
      // workgroups per CU up (fc1.dW, 3136 x 512: 392 tiles of 64x64 run 16.5 us, 784 of 32x64
      // in three dispatch rounds 19.5 us -- tools/fc1_probe.hip)
      const int64_t need = i == 0 ? 1024 : (i == 1 ? 384 : 512);
      if (ntiles(cand[i]) >= need) { cfg = cand[i]; break; }
    }
  }
  pl->cfg = cfg;
  pl->bm = kCfgs[cfg].bm;
  pl->bn = kCfgs[cfg].bn;
  const int64_t tiles = ntiles(cfg);
  int splits = 1;
  if (d->force_splits > 0) {
    splits = d->force_splits;
  } else if (tall_reduce || cfg == 7) {
    splits = (int)(512 / tiles);   // two workgroups per CU, >= 12 K-steps each at conv1's size
    const int max_by_k = (int)(K / (8 * AA_BK));
    if (splits > max_by_k) splits = max_by_k;
    if (splits < 1) splits = 1;
  } else if (tiles < 256) {
    splits = (int)((512 + tiles - 1) / tiles);
    const int max_by_k = (int)(K / (4 * AA_BK));  // at least four K-steps per split
    if (splits > max_by_k) splits = max_by_k;
    if (splits < 1) splits = 1;
    // (tried: contractions below ~128 MFLOP unsplit on however many tiles they have, which saves
    // the slab-reduce launch -- SAC iteration 0.584 vs 0.554 ms: 32 workgroups walking K = 393
    // take longer than 256 walking K = 49 plus the reduce; removed)
  }
  int kps = (int)((K + splits - 1) / splits);
  kps = ((kps + AA_BK - 1) / AA_BK) * AA_BK;
  splits = (int)((K + kps - 1) / kps);
  pl->splits = splits;
  pl->k_per_split = kps;
  pl->ws_bytes = splits > 1 ? (size_t)splits * (size_t)(M * N + (d->colsum_out ? N : 0)) *
                                      sizeof(float)
                            : 0;
  return AA_OK;
}

template <int AM, int BMODE, int BM, int BN, int WGM, int WGN, int WGK, int PD, int VEC>
static void aa_gemm_launch_pd(const GemmP& p, const AaGemmPlan& pl, hipStream_t st) {
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, pl.splits);
  if (p.xcd_mode != 0) grid = dim3(((p.gx * p.gy * p.gz + 7) / 8) * 8, 1, 1);
  constexpr bool A_IS_T = (AM == AA_A_ROW || AM == AA_A_PATCH || AM == AA_A_PATCH_U8);
  constexpr bool B_IS_T = (BMODE == AA_B_COL);
  constexpr int lda_s = BM + (A_IS_T ? 1 : 4), ldb_s = BN + (B_IS_T ? 1 : 4);
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  size_t smem = (size_t)2 * AA_BK * (lda_s + ldb_s) * sizeof(float);
  const size_t red = (size_t)(WGK - 1) * WGM * WGN * TM * TN * 1024 * sizeof(float);
  if (red > smem) smem = red;
  if (smem < 4096) smem = 4096;  // fused column-sum scratch [1024/BN][BN]
  hipLaunchKernelGGL((aa_gemm_kernel<AM, BMODE, BM, BN, WGM, WGN, WGK, PD, VEC>), grid,
                     dim3(AA_GEMM_THREADS), smem, st, p);
}

template <int AM, int BMODE, int BM, int BN, int WGM, int WGN, int WGK>
static void aa_gemm_launch_one(const GemmP& p, const AaGemmPlan& pl, hipStream_t st) {
  constexpr bool dense = AM == AA_A_ROW || AM == AA_A_COL;
  if (p.use_dma) {
    constexpr int AK = AM == AA_A_ROW ? AA_KIND_T_DENSE
                     : AM == AA_A_COL ? AA_KIND_D_DENSE
                     : AM == AA_A_PATCH ? AA_KIND_T_PATCH
                     : AM == AA_A_PATCH_U8 ? AA_KIND_T_PATCH_U8
                     : AM == AA_A_PATCH_T ? AA_KIND_D_PATCHT : AA_KIND_D_PATCHT_U8;
    constexpr int BKIND = BMODE == AA_B_ROW ? AA_KIND_D_DENSE : AA_KIND_T_DENSE;
    constexpr int STAGE = DmaOp<AK, BM>::kPadded + DmaOp<BKIND, BN>::kPadded;
    constexpr int NS = STAGE * 4 <= 65536 ? 4 : (STAGE * 3 <= 65536 ? 3 : 2);
    aa_gemm_dma_launch<AK, BKIND, BM, BN, WGM, WGN, WGK, NS>(p, pl.splits, st);
    return;
  }
  if constexpr (BM > 128) {
    return;  // 256-row tiles exist for the LDS-DMA loop only (the plan never picks them otherwise)
  } else {
  // dense operands that are ragged / unaligned use 4-byte loads (conv patches are always vectors)
  if (!(p.b_vec && (p.a_vec || !dense))) {
    aa_gemm_launch_pd<AM, BMODE, BM, BN, WGM, WGN, WGK, 2, 0>(p, pl, st);
    return;
  }
  // two K-tiles of loads in flight measured best on MI355X (1: 3-8 % slower, 3: VGPR-bound)
  aa_gemm_launch_pd<AM, BMODE, BM, BN, WGM, WGN, WGK, 2, 1>(p, pl, st);
  }
}

template <int AM, int BMODE>
static int aa_gemm_launch_cfg(const GemmP& p, const AaGemmPlan& pl, hipStream_t st) {
  switch (pl.cfg) {
    case 0: aa_gemm_launch_one<AM, BMODE, 128, 64, 2, 2, 1>(p, pl, st); break;
    case 1: aa_gemm_launch_one<AM, BMODE, 128, 32, 4, 1, 1>(p, pl, st); break;
    case 2: aa_gemm_launch_one<AM, BMODE, 64, 64, 2, 2, 1>(p, pl, st); break;
    case 3: aa_gemm_launch_one<AM, BMODE, 128, 128, 2, 2, 1>(p, pl, st); break;
    case 4: aa_gemm_launch_one<AM, BMODE, 64, 32, 2, 1, 2>(p, pl, st); break;
    case 5: aa_gemm_launch_one<AM, BMODE, 32, 64, 1, 2, 2>(p, pl, st); break;
    case 7:
      if (!p.use_dma) return AA_ERR_INVALID;
      aa_gemm_launch_one<AM, BMODE, 256, 32, 4, 1, 1>(p, pl, st);
      break;
    default: aa_gemm_launch_one<AM, BMODE, 32, 32, 1, 1, 4>(p, pl, st); break;
  }
  return aa_launch_status();
}

static unsigned aa_magic(int d) {  // ceil(2^32 / d); d >= 2 (d == 1 handled by the caller)
  return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d);
}

extern "C" {

int64_t aa_gemm_f32_workspace_bytes(const aa_gemm_desc* d) {
  AaGemmPlan pl;
  if (d == nullptr || aa_gemm_plan(d, &pl) != AA_OK) return -1;
  return (int64_t)pl.ws_bytes;
}

// defer_splits != nullptr: when the plan is split-K, stop after the main launch -- the raw slabs
// [splits][M][N] stay at the start of `workspace` for a consumer that sums them in its own prologue
// (*defer_splits = splits); with one split the finished result is in C (*defer_splits = 1).
static int aa_gemm_f32_impl(const aa_gemm_desc* d, void* workspace, int64_t workspace_bytes,
                            void* stream, int* defer_splits) {
  if (d == nullptr || d->A == nullptr || d->B == nullptr || d->C == nullptr) return AA_ERR_INVALID;
  AaGemmPlan pl;
  int rc = aa_gemm_plan(d, &pl);
  if (rc != AA_OK) return rc;
  if (pl.ws_bytes > 0 && (workspace == nullptr || (size_t)workspace_bytes < pl.ws_bytes))
    return AA_ERR_RANGE;
  const bool patch = d->a_mode >= AA_A_PATCH;
  const bool u8 = d->a_mode == AA_A_PATCH_U8 || d->a_mode == AA_A_PATCH_T_U8;
  if (d->colsum_out != nullptr && d->b_mode != AA_B_ROW) return AA_ERR_INVALID;
  GemmP p;
  p.A = d->A;
  p.B = d->B;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.W = p.Cin = p.OW = p.OHW = p.stride = p.seg = p.rowpitch = p.imgpitch = 0;
  p.a_div = d->a_div != 0.f ? d->a_div : 1.f;
  p.a_rcp = 1.0f / p.a_div;
  p.a_fast = 0;
  p.magic_ohw = p.magic_ow = p.magic_seg = 0;
  p.fastdiv = 0;
  p.colsum_out = d->colsum_out;
  if (u8) {
    // the refined product must reproduce the IEEE quotient for every byte value
    int ok = 1;
    for (int b = 0; b < 256 && ok; ++b) {
      const float x = (float)b;
      volatile float q0 = x * p.a_rcp;
      const float q = fmaf(fmaf(-p.a_div, q0, x), p.a_rcp, q0);
      volatile float want = x / p.a_div;
      if (q != want) ok = 0;
    }
    p.a_fast = ok;
  }
  if (patch) {
    if (d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0)
      return AA_ERR_INVALID;
    const int OH = (d->H - d->KH) / d->stride + 1, OW = (d->W - d->KW) / d->stride + 1;
    if (OH <= 0 || OW <= 0) return AA_ERR_INVALID;
    const int seg = d->KW * d->Cin;
    const int vecw = u8 ? 16 : 4;
    // every vector load must stay inside one patch row segment and be 16-byte aligned
    if (seg % vecw != 0 || (d->Cin * d->stride) % vecw != 0 || (d->W * d->Cin) % vecw != 0)
      return AA_ERR_INVALID;
    if (((uintptr_t)d->A & 15) != 0) return AA_ERR_INVALID;
    const int Kp = d->KH * seg;
    const int64_t npix = (int64_t)d->n_img * OH * OW;
    const bool fwd = d->a_mode == AA_A_PATCH || d->a_mode == AA_A_PATCH_U8;
    if (fwd ? (d->K != Kp || d->M != npix) : (d->M != Kp || d->K != npix)) return AA_ERR_INVALID;
    const int64_t dense_pitch = (int64_t)d->H * d->W * d->Cin;
    const int64_t pitch = d->img_pitch > 0 ? (int64_t)d->img_pitch : dense_pitch;
    if (pitch < dense_pitch || pitch % vecw != 0) return AA_ERR_INVALID;
    if ((int64_t)d->n_img * pitch >= 0x7fffffffLL) return AA_ERR_RANGE;
    p.W = d->W; p.Cin = d->Cin; p.OW = OW; p.OHW = OH * OW; p.stride = d->stride;
    p.seg = seg; p.rowpitch = d->W * d->Cin; p.imgpitch = (int)pitch;
    // n / d == mulhi(n, ceil(2^32/d)) whenever n * d < 2^32; n is a pixel or patch-element index
    const int64_t nmax = (npix > Kp ? npix : Kp) + AA_BK + 256;
    const int64_t dmax = p.OHW > seg ? p.OHW : seg;
    if (OW >= 2 && p.OHW >= 2 && seg >= 2 && nmax * dmax < (1ll << 32)) {
      p.fastdiv = 1;
      p.magic_ohw = aa_magic(p.OHW);
      p.magic_ow = aa_magic(OW);
      p.magic_seg = aa_magic(seg);
    }
  }
  // 16-byte loads need an aligned base, a leading dimension and a contiguous extent that are
  // multiples of 4 (so no vector straddles the operand's edge)
  const int a_contig = d->a_mode == AA_A_ROW ? d->K : d->M;
  const int b_contig = d->b_mode == AA_B_ROW ? d->N : d->K;
  p.a_vec = (!patch && d->lda % 4 == 0 && a_contig % 4 == 0 && (((uintptr_t)d->A & 15) == 0));
  p.b_vec = (d->ldb % 4 == 0 && b_contig % 4 == 0 && (((uintptr_t)d->B & 15) == 0));
  {
    // operand spans in bytes: the buffer resources carry them and lane offsets are 32-bit
    int64_t a_span, b_span;
    if (patch) {
      const int64_t pitch = d->img_pitch > 0 ? (int64_t)d->img_pitch : (int64_t)d->H * d->W * d->Cin;
      a_span = ((int64_t)(d->n_img - 1) * pitch + (int64_t)d->H * d->W * d->Cin) * (u8 ? 1 : 4);
    } else if (d->a_mode == AA_A_ROW) {
      a_span = ((int64_t)(d->M - 1) * d->lda + d->K) * 4;
    } else {
      a_span = ((int64_t)(d->K - 1) * d->lda + d->M) * 4;
    }
    if (d->b_mode == AA_B_ROW)
      b_span = ((int64_t)(d->K - 1) * d->ldb + d->N) * 4;
    else
      b_span = ((int64_t)(d->N - 1) * d->ldb + d->K) * 4;
    if (a_span <= 0 || b_span <= 0 || a_span >= (1ll << 31) || b_span >= (1ll << 31))
      return AA_ERR_RANGE;
    p.a_bytes = (unsigned)a_span;
    p.b_bytes = (unsigned)b_span;
  }
  // LDS-DMA main loop for 16-byte-regular operands; contractions with only a few K-steps are
  // epilogue-bound and run better on the register-staged loop (smaller LDS, more groups per CU)
  p.use_dma = aa_desc_dma_ok(d) ? 1 : 0;
  p.k_per_split = pl.k_per_split;
  p.splits = pl.splits;
  // XCD-aware block order (aa_block_of) for the dense contractions: partition across the 8 L2s
  // along the tile dimension that minimises the bytes every XCD has to fetch.  An operand NOT
  // indexed by the partitioned dimension is fetched by all 8 XCDs, a partitioned one by
  // 8 / min(8, tiles) of them.  AA_GEMM_XCD=0 restores the launch order (A/B measurements).
  p.xcd_mode = 0;
  p.gx = (d->M + pl.bm - 1) / pl.bm;
  p.gy = (d->N + pl.bn - 1) / pl.bn;
  p.gz = pl.splits;
  {
    const bool enabled = true;     // XCD-aware block order (round 1: 59 -> 14 MB of fc1 traffic)
    const bool dense = d->a_mode == AA_A_ROW || d->a_mode == AA_A_COL;
    const int64_t n_blocks = (int64_t)p.gx * p.gy * p.gz;
    if (enabled && dense && pl.cfg != AA_CFG_U8_BF16 - 1 && n_blocks >= 64 &&
        n_blocks < (1 << 24)) {
      const double a_bytes = 4.0 * d->M * d->K, b_bytes = 4.0 * d->K * d->N;
      auto share = [](int g) { return 8.0 / (g < 8 ? g : 8); };   // XCDs per slice
      const double cost_m = a_bytes * share(p.gx) + b_bytes * 8.0;   // = launch order
      const double cost_n = a_bytes * 8.0 + b_bytes * share(p.gy);
      const double cost_k = (a_bytes + b_bytes) * share(p.gz);
      if (cost_k <= cost_n && cost_k < cost_m) p.xcd_mode = 1;
      else if (cost_n < cost_m) p.xcd_mode = 2;
      else if (p.gx % 8 != 0) p.xcd_mode = 3;      // launch order already partitions M when 8 | gx
    }
  }
  p.bias = d->bias;
  p.act = d->act;
  p.mask_src = d->mask_src;
  p.ldm = d->ldm;
  p.mask_kind = d->mask_src != nullptr ? d->mask_kind : 0;
  float* slabs = (float*)workspace;
  p.C = pl.splits > 1 ? slabs : d->C;
  hipStream_t st = (hipStream_t)stream;

  if (d->b_mode != AA_B_ROW && d->b_mode != AA_B_COL) return AA_ERR_INVALID;
  if (d->b_mode == AA_B_COL && d->a_mode != AA_A_ROW) return AA_ERR_INVALID;
  switch (d->a_mode) {
    case AA_A_ROW:
      rc = d->b_mode == AA_B_ROW ? aa_gemm_launch_cfg<AA_A_ROW, AA_B_ROW>(p, pl, st)
                                 : aa_gemm_launch_cfg<AA_A_ROW, AA_B_COL>(p, pl, st);
      break;
    case AA_A_COL:
      rc = aa_gemm_launch_cfg<AA_A_COL, AA_B_ROW>(p, pl, st);
      break;
    case AA_A_PATCH:
      rc = aa_gemm_launch_cfg<AA_A_PATCH, AA_B_ROW>(p, pl, st);
      break;
    case AA_A_PATCH_U8:
      if (pl.cfg == AA_CFG_U8_BF16 - 1) {
        aa_conv_u8_bf16_launch(p, st);
        rc = aa_launch_status();
      } else {
        rc = aa_gemm_launch_cfg<AA_A_PATCH_U8, AA_B_ROW>(p, pl, st);
      }
      break;
    case AA_A_PATCH_T: rc = aa_gemm_launch_cfg<AA_A_PATCH_T, AA_B_ROW>(p, pl, st); break;
    case AA_A_PATCH_T_U8:
      if (pl.cfg == AA_CFG_U8_BF16 - 1) {
        rc = aa_conv_u8_dw_bf16_launch(p, d->n_img, d->H * d->W * d->Cin, pl.splits, st);
      } else {
        rc = aa_gemm_launch_cfg<AA_A_PATCH_T_U8, AA_B_ROW>(p, pl, st);
      }
      break;
    default: return AA_ERR_INVALID;
  }
  if (rc != AA_OK) return rc;
  if (defer_splits != nullptr) {
    *defer_splits = pl.splits;
    return rc;
  }
  if (pl.splits > 1) {
    const size_t MN = (size_t)d->M * d->N;
    const bool vec = d->N % 4 == 0 && d->ldc % 4 == 0 && (((uintptr_t)d->C & 15) == 0) &&
                     (d->mask_src == nullptr || d->ldm % 4 == 0);
    const size_t work = (MN + (d->colsum_out ? d->N : 0)) / (vec ? 4 : 1);
    // few items and many slabs: spread the slabs of an item over 16 z-lanes of the workgroup
    const bool deep = pl.splits >= 32 && work <= 65536;
    const int ipb = deep ? 16 : 256;
    int blocks = (int)((work + ipb - 1) / ipb);
    if (blocks > 2048) blocks = 2048;
#define AA_LAUNCH_REDUCE(VEC, ZL)                                                               \
    hipLaunchKernelGGL((aa_splitk_reduce_kernel<VEC, ZL>), dim3(blocks), dim3(256), 0, st,          \
                       (const float*)slabs, pl.splits, d->M, d->N, d->C, d->ldc, d->bias, d->act,  \
                       d->mask_src, d->ldm, p.mask_kind, d->colsum_out)
    if (vec && deep) AA_LAUNCH_REDUCE(4, 16);
    else if (vec) AA_LAUNCH_REDUCE(4, 1);
    else if (deep) AA_LAUNCH_REDUCE(1, 16);
    else AA_LAUNCH_REDUCE(1, 1);
#undef AA_LAUNCH_REDUCE
    rc = aa_launch_status();
  }
  return rc;
}

int aa_gemm_f32(const aa_gemm_desc* d, void* workspace, int64_t workspace_bytes, void* stream) {
  return aa_gemm_f32_impl(d, workspace, workspace_bytes, stream, nullptr);
}

int aa_gemm_f32_slabs(const aa_gemm_desc* d, void* workspace, int64_t workspace_bytes,
                      int32_t* splits_out, void* stream) {
  if (d == nullptr || splits_out == nullptr) return AA_ERR_INVALID;
  // the consumer applies bias / activation itself; masks belong to the input-gradient
  // contractions, which keep the reduce launch.  colsum_out != nullptr (a weight gradient with
  // its fused bias gradient): the column-sum rows follow the slabs, for a consumer that sums both
  // (aa_rmsprop_step_slabs); with *splits_out == 1 C and colsum_out hold the final values.
  if (d->mask_src != nullptr) return AA_ERR_INVALID;
  int splits = 0;
  const int rc = aa_gemm_f32_impl(d, workspace, workspace_bytes, stream, &splits);
  *splits_out = splits;
  return rc;
}

}  // extern "C"
