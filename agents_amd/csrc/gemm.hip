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
#include "agents_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
  int k_per_split;
  int splits;
  const float* bias;
  int act;
  const float* mask_src;
  int ldm;
  int mask_kind;
  int a_vec, b_vec;  // 16-byte vector path usable for dense operands
};

__device__ static inline float aa_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}
__device__ static inline float aa_actgrad(float y, int kind) {
  // derivative of the activation expressed through its OUTPUT y
  if (kind == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (kind == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

// Pixel index (b, oy, ox) -> element offset of the patch origin in the NHWC input.
__device__ static inline int aa_pix_base(const GemmP& p, int pix) {
  const int b = pix / p.OHW;
  const int rem = pix - b * p.OHW;
  const int oy = rem / p.OW;
  const int ox = rem - oy * p.OW;
  return b * p.imgpitch + (oy * p.stride) * p.rowpitch + (ox * p.stride) * p.Cin;
}
// Patch element index k -> element offset relative to the patch origin.
__device__ static inline int aa_patch_off(const GemmP& p, int k) {
  const int ky = k / p.seg;
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

// ---- dense K-contiguous (A_ROW for A, B_COL for B): elem(x,k) = base[x*ld + k] ------------
template <int BX>
__device__ static inline void load_T_dense(StageT<BX>& s, const float* base, int ld, int x0,
                                           int X, int k0, int k_end, int vec) {
  const int kq = threadIdx.x & 7, r = threadIdx.x >> 3;
  const int k = k0 + 4 * kq;
#pragma unroll
  for (int p = 0; p < StageT<BX>::PASSES; ++p) {
    const int x = x0 + r + 32 * p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < X) {
      const float* src = base + (size_t)x * ld + k;
      if (vec && k + 3 < k_end) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        if (k + 0 < k_end) v.x = src[0];
        if (k + 1 < k_end) v.y = src[1];
        if (k + 2 < k_end) v.z = src[2];
        if (k + 3 < k_end) v.w = src[3];
      }
    }
    s.r[p] = v;
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
template <int BX>
__device__ static inline void load_D_dense(StageD<BX>& s, const float* base, int ld, int x0,
                                           int X, int k0, int k_end, int vec) {
  constexpr int V = StageD<BX>::V, RP = StageD<BX>::RP;
  const int v4 = threadIdx.x % V, r = threadIdx.x / V;
  const int x = x0 + 4 * v4;
#pragma unroll
  for (int p = 0; p < StageD<BX>::PASSES; ++p) {
    const int kk = r + RP * p;
    const int k = k0 + kk;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kk < AA_BK && k < k_end) {
      const float* src = base + (size_t)k * ld + x;
      if (vec && x + 3 < X) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        if (x + 0 < X) v.x = src[0];
        if (x + 1 < X) v.y = src[1];
        if (x + 2 < X) v.z = src[2];
        if (x + 3 < X) v.w = src[3];
      }
    }
    s.r[p] = v;
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

// ---- conv patches, forward orientation: A(m = pixel, k = patch element), K-contiguous -------
template <int BX>
__device__ static inline void load_T_patch(StageT<BX>& s, const GemmP& p, const int* rowbase,
                                           int k0, int k_end) {
  const int kq = threadIdx.x & 7;
  const int k = k0 + 4 * kq;
  const bool kin = k < k_end;  // K % 4 == 0 validated on host
  const int koff = kin ? aa_patch_off(p, k) : 0;
  const float* A = reinterpret_cast<const float*>(p.A);
#pragma unroll
  for (int q = 0; q < StageT<BX>::PASSES; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kin && rowbase[q] >= 0) v = *reinterpret_cast<const float4*>(A + rowbase[q] + koff);
    s.r[q] = v;
  }
}
// uint8 frames: 2 x 16-byte vectors per pixel row per K-step; thread -> (row = t>>1, half = t&1)
template <int BX>
__device__ static inline void load_T_patch_u8(StageU8& s, const GemmP& p, int rowbase, int k0,
                                              int k_end) {
  const int kq = threadIdx.x & 1;
  const int k = k0 + 16 * kq;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (k < k_end && rowbase >= 0) {
    const uint8_t* A = reinterpret_cast<const uint8_t*>(p.A);
    v = *reinterpret_cast<const uint4*>(A + rowbase + aa_patch_off(p, k));
  }
  s.r = v;
}
template <int BX, int LDS_LD>
__device__ static inline void store_T_patch_u8(const StageU8& s, float* tile, float div) {
  const int kq = threadIdx.x & 1, r = threadIdx.x >> 1;
  if (r >= BX) return;
  float* d = tile + (16 * kq) * LDS_LD + r;
  const uint32_t w[4] = {s.r.x, s.r.y, s.r.z, s.r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f = (float)((w[i] >> (8 * j)) & 0xffu) / div;
      d[(4 * i + j) * LDS_LD] = f;
    }
  }
}

// ---- conv patches, weight-grad orientation: A'(i = patch element, kk = pixel), i-contiguous --
template <int BX>
__device__ static inline void load_D_patchT(StageD<BX>& s, const GemmP& p, int ioff, bool iin,
                                            int k0, int k_end) {
  constexpr int V = StageD<BX>::V, RP = StageD<BX>::RP;
  const int r = threadIdx.x / V;
  const float* A = reinterpret_cast<const float*>(p.A);
#pragma unroll
  for (int q = 0; q < StageD<BX>::PASSES; ++q) {
    const int kk = r + RP * q;
    const int pix = k0 + kk;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iin && kk < AA_BK && pix < k_end)
      v = *reinterpret_cast<const float4*>(A + aa_pix_base(p, pix) + ioff);
    s.r[q] = v;
  }
}
// uint8: 16 patch elements per vector; BX/16 vectors per pixel row.
template <int BX>
__device__ static inline void load_D_patchT_u8(StageU8& s, const GemmP& p, int ioff, bool iin,
                                               int k0, int k_end) {
  constexpr int V = BX / 16;
  const int r = threadIdx.x / V;
  const int pix = k0 + r;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (iin && r < AA_BK && pix < k_end) {
    const uint8_t* A = reinterpret_cast<const uint8_t*>(p.A);
    v = *reinterpret_cast<const uint4*>(A + aa_pix_base(p, pix) + ioff);
  }
  s.r = v;
}
template <int BX, int LDS_LD>
__device__ static inline void store_D_patchT_u8(const StageU8& s, float* tile, float div) {
  constexpr int V = BX / 16;
  const int v16 = threadIdx.x % V, r = threadIdx.x / V;
  if (r >= AA_BK) return;
  float* d = tile + r * LDS_LD + 16 * v16;
  const uint32_t w[4] = {s.r.x, s.r.y, s.r.z, s.r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 f;
    f.x = (float)((w[i] >> 0) & 0xffu) / div;
    f.y = (float)((w[i] >> 8) & 0xffu) / div;
    f.z = (float)((w[i] >> 16) & 0xffu) / div;
    f.w = (float)((w[i] >> 24) & 0xffu) / div;
    *reinterpret_cast<float4*>(d + 4 * i) = f;
  }
}

// ------------------------------------------------------------------------------------------
template <int AM, int BMODE, int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(AA_GEMM_THREADS) aa_gemm_kernel(GemmP p) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  constexpr bool A_IS_T = (AM == AA_A_ROW || AM == AA_A_PATCH || AM == AA_A_PATCH_U8);
  constexpr bool B_IS_T = (BMODE == AA_B_COL);
  constexpr int LDA_S = BM + (A_IS_T ? 1 : 4);
  constexpr int LDB_S = BN + (B_IS_T ? 1 : 4);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BK][LDA_S]
  float* Bs = smem + 2 * AA_BK * LDA_S;      // [2][BK][LDB_S]

  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int k_begin = blockIdx.z * p.k_per_split;
  int k_end = k_begin + p.k_per_split;
  if (k_end > p.K) k_end = p.K;
  const int nk = (k_end - k_begin + AA_BK - 1) / AA_BK;

  // ---- per-thread loader state --------------------------------------------------------
  StageT<BM> aT;
  StageD<BM> aD;
  StageU8 aU;
  StageT<BN> bT;
  StageD<BN> bD;
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

  auto load_tiles = [&](int k0) {
    if constexpr (AM == AA_A_ROW)
      load_T_dense<BM>(aT, reinterpret_cast<const float*>(p.A), p.lda, m0, p.M, k0, k_end, p.a_vec);
    else if constexpr (AM == AA_A_COL)
      load_D_dense<BM>(aD, reinterpret_cast<const float*>(p.A), p.lda, m0, p.M, k0, k_end, p.a_vec);
    else if constexpr (AM == AA_A_PATCH)
      load_T_patch<BM>(aT, p, rowbase, k0, k_end);
    else if constexpr (AM == AA_A_PATCH_U8)
      load_T_patch_u8<BM>(aU, p, rowbase_u8, k0, k_end);
    else if constexpr (AM == AA_A_PATCH_T)
      load_D_patchT<BM>(aD, p, ioff, iin, k0, k_end);
    else
      load_D_patchT_u8<BM>(aU, p, ioff, iin, k0, k_end);
    if constexpr (BMODE == AA_B_ROW)
      load_D_dense<BN>(bD, p.B, p.ldb, n0, p.N, k0, k_end, p.b_vec);
    else
      load_T_dense<BN>(bT, p.B, p.ldb, n0, p.N, k0, k_end, p.b_vec);
  };
  auto store_tiles = [&](int buf) {
    float* at = As + buf * AA_BK * LDA_S;
    float* bt = Bs + buf * AA_BK * LDB_S;
    if constexpr (AM == AA_A_ROW || AM == AA_A_PATCH)
      store_T<BM, LDA_S>(aT, at);
    else if constexpr (AM == AA_A_COL || AM == AA_A_PATCH_T)
      store_D<BM, LDA_S>(aD, at);
    else if constexpr (AM == AA_A_PATCH_U8)
      store_T_patch_u8<BM, LDA_S>(aU, at, p.a_div);
    else
      store_D_patchT_u8<BM, LDA_S>(aU, at, p.a_div);
    if constexpr (BMODE == AA_B_ROW)
      store_D<BN, LDB_S>(bD, bt);
    else
      store_T<BN, LDB_S>(bT, bt);
  };

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nk > 0) {
    load_tiles(k_begin);
    store_tiles(0);
  }
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < nk;
    if (more) load_tiles(k_begin + (t + 1) * AA_BK);
    const float* at = As + buf * AA_BK * LDA_S + wm * (TM * 32) + l31;
    const float* bt = Bs + buf * AA_BK * LDB_S + wn * (TN * 32) + l31;
#pragma unroll
    for (int kk = 0; kk < AA_BK / 2; ++kk) {
      const int k = 2 * kk + lh;
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
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------
  const bool raw = p.splits > 1;
  float* C = raw ? p.C + (size_t)blockIdx.z * (size_t)p.M * (size_t)p.N : p.C;
  const int ldc = raw ? p.N : p.ldc;
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

// out[m][n] = epilogue( sum_z slab[z][m][n] ), fixed z order => deterministic.
__global__ void __launch_bounds__(256)
aa_splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                        float* __restrict__ C, int ldc, const float* __restrict__ bias, int act,
                        const float* __restrict__ mask_src, int ldm, int mask_kind) {
  const size_t MN = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < MN;
       i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += slab[(size_t)z * MN + i];
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    if (bias != nullptr) v += bias[n];
    v = aa_act(v, act);
    if (mask_kind != 0) v *= aa_actgrad(mask_src[(size_t)m * ldm + n], mask_kind);
    C[(size_t)m * ldc + n] = v;
  }
}

// ------------------------------------------------------------------------------------------
// host side: shape-driven tile / split-K selection and dispatch
// ------------------------------------------------------------------------------------------
struct AaGemmPlan {
  int cfg;       // 0: 128x64, 1: 128x32, 2: 64x64
  int bm, bn;
  int splits, k_per_split;
  size_t ws_bytes;
};

static int aa_gemm_plan(const aa_gemm_desc* d, AaGemmPlan* pl) {
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return AA_ERR_INVALID;
  int cfg;
  if (d->force_cfg > 0) {
    cfg = d->force_cfg - 1;
    if (cfg > 2) return AA_ERR_INVALID;
  } else if (d->N <= 32) {
    cfg = 1;
  } else {
    const int64_t t128 = (int64_t)((d->M + 127) / 128) * ((d->N + 63) / 64);
    cfg = t128 >= 512 ? 0 : 2;
  }
  pl->cfg = cfg;
  pl->bm = cfg == 2 ? 64 : 128;
  pl->bn = cfg == 1 ? 32 : 64;
  const int64_t tiles = (int64_t)((d->M + pl->bm - 1) / pl->bm) * ((d->N + pl->bn - 1) / pl->bn);
  int splits = 1;
  if (d->force_splits > 0) {
    splits = d->force_splits;
  } else if (tiles < 384) {
    splits = (int)((768 + tiles - 1) / tiles);
    const int max_by_k = d->K / (2 * AA_BK);  // at least two K-steps per split
    if (splits > max_by_k) splits = max_by_k;
    if (splits < 1) splits = 1;
  }
  int kps = (d->K + splits - 1) / splits;
  kps = ((kps + AA_BK - 1) / AA_BK) * AA_BK;
  splits = (d->K + kps - 1) / kps;
  pl->splits = splits;
  pl->k_per_split = kps;
  pl->ws_bytes = splits > 1 ? (size_t)splits * (size_t)d->M * (size_t)d->N * sizeof(float) : 0;
  return AA_OK;
}

template <int AM, int BMODE>
static int aa_gemm_launch_cfg(const GemmP& p, const AaGemmPlan& pl, hipStream_t st) {
  dim3 grid((p.M + pl.bm - 1) / pl.bm, (p.N + pl.bn - 1) / pl.bn, pl.splits);
  dim3 block(AA_GEMM_THREADS);
  constexpr bool A_IS_T = (AM == AA_A_ROW || AM == AA_A_PATCH || AM == AA_A_PATCH_U8);
  constexpr bool B_IS_T = (BMODE == AA_B_COL);
  const int lda_s = pl.bm + (A_IS_T ? 1 : 4), ldb_s = pl.bn + (B_IS_T ? 1 : 4);
  const size_t smem = (size_t)2 * AA_BK * (lda_s + ldb_s) * sizeof(float);
  switch (pl.cfg) {
    case 0:
      hipLaunchKernelGGL((aa_gemm_kernel<AM, BMODE, 128, 64, 2, 2>), grid, block, smem, st, p);
      break;
    case 1:
      hipLaunchKernelGGL((aa_gemm_kernel<AM, BMODE, 128, 32, 4, 1>), grid, block, smem, st, p);
      break;
    default:
      hipLaunchKernelGGL((aa_gemm_kernel<AM, BMODE, 64, 64, 2, 2>), grid, block, smem, st, p);
      break;
  }
  return aa_launch_status();
}

extern "C" {

int64_t aa_gemm_f32_workspace_bytes(const aa_gemm_desc* d) {
  AaGemmPlan pl;
  if (d == nullptr || aa_gemm_plan(d, &pl) != AA_OK) return -1;
  return (int64_t)pl.ws_bytes;
}

int aa_gemm_f32(const aa_gemm_desc* d, void* workspace, int64_t workspace_bytes, void* stream) {
  if (d == nullptr || d->A == nullptr || d->B == nullptr || d->C == nullptr) return AA_ERR_INVALID;
  AaGemmPlan pl;
  int rc = aa_gemm_plan(d, &pl);
  if (rc != AA_OK) return rc;
  if (pl.ws_bytes > 0 && (workspace == nullptr || (size_t)workspace_bytes < pl.ws_bytes))
    return AA_ERR_RANGE;
  const bool patch = d->a_mode >= AA_A_PATCH;
  const bool u8 = d->a_mode == AA_A_PATCH_U8 || d->a_mode == AA_A_PATCH_T_U8;
  GemmP p;
  p.A = d->A;
  p.B = d->B;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.W = p.Cin = p.OW = p.OHW = p.stride = p.seg = p.rowpitch = p.imgpitch = 0;
  p.a_div = d->a_div != 0.f ? d->a_div : 1.f;
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
  }
  p.a_vec = (!patch && (d->lda % 4 == 0) && (((uintptr_t)d->A & 15) == 0)) ? 1 : 0;
  p.b_vec = ((d->ldb % 4 == 0) && (((uintptr_t)d->B & 15) == 0)) ? 1 : 0;
  p.k_per_split = pl.k_per_split;
  p.splits = pl.splits;
  p.bias = d->bias;
  p.act = d->act;
  p.mask_src = d->mask_src;
  p.ldm = d->ldm;
  p.mask_kind = d->mask_src != nullptr ? d->mask_kind : 0;
  p.C = pl.splits > 1 ? (float*)workspace : d->C;
  hipStream_t st = (hipStream_t)stream;

  if (d->b_mode != AA_B_ROW && d->b_mode != AA_B_COL) return AA_ERR_INVALID;
  if (d->b_mode == AA_B_COL && d->a_mode != AA_A_ROW) return AA_ERR_INVALID;
  switch (d->a_mode) {
    case AA_A_ROW:
      rc = d->b_mode == AA_B_ROW ? aa_gemm_launch_cfg<AA_A_ROW, AA_B_ROW>(p, pl, st)
                                 : aa_gemm_launch_cfg<AA_A_ROW, AA_B_COL>(p, pl, st);
      break;
    case AA_A_COL: rc = aa_gemm_launch_cfg<AA_A_COL, AA_B_ROW>(p, pl, st); break;
    case AA_A_PATCH: rc = aa_gemm_launch_cfg<AA_A_PATCH, AA_B_ROW>(p, pl, st); break;
    case AA_A_PATCH_U8: rc = aa_gemm_launch_cfg<AA_A_PATCH_U8, AA_B_ROW>(p, pl, st); break;
    case AA_A_PATCH_T: rc = aa_gemm_launch_cfg<AA_A_PATCH_T, AA_B_ROW>(p, pl, st); break;
    case AA_A_PATCH_T_U8: rc = aa_gemm_launch_cfg<AA_A_PATCH_T_U8, AA_B_ROW>(p, pl, st); break;
    default: return AA_ERR_INVALID;
  }
  if (rc != AA_OK) return rc;
  if (pl.splits > 1) {
    const size_t MN = (size_t)d->M * d->N;
    int blocks = (int)((MN + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(aa_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st,
                       (const float*)workspace, pl.splits, d->M, d->N, d->C, d->ldc, d->bias,
                       d->act, d->mask_src, d->ldm, p.mask_kind);
    rc = aa_launch_status();
  }
  return rc;
}

}  // extern "C"
