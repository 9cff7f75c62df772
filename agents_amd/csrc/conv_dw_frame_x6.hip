// Weight (and bias) gradient of a VALID Conv2D over fp32 NHWC frames, per frame, on the bf16 matrix
// cores at fp32 accuracy (x6_common.h: exact three-piece split, six of nine piece products, fp32
// accumulation):
//
//   dW[ky,kx,ci,co] = sum_{b,oy,ox} x[b, oy*s+ky, ox*s+kx, ci] * dZ[b,oy,ox,co]      db[co] = sum dZ
//
// (tf.GradientTape through keras Conv2D, agents/dqn/dqn_agent.py:412-426.)  As a contraction the
// reduction index is the PIXEL, but both operands are stored channel-fastest: the fp32-MFMA GEMM
// (gemm.hip, a_mode PATCH_T) gathers single elements and runs at ~25 % of the fp32 peak on the
// DQN shapes.  Here every element is split ONCE while its frame is staged into three bf16 LDS
// planes that keep the natural [pixel][channel] layout, and the fragments are read with gfx950's
// LDS transpose read: for an operand image whose rows are k (pixels) and whose 16 columns are the
// tile's channels, lane (g = l >> 4, j = l & 15) issues two ds_read_b64_tr_b16 at
//     row k = 8 g + 4 h + (j >> 2), column quad (j & 3)          h = 0, 1
// and receives exactly the MFMA operand of v_mfma_f32_16x16x32_bf16 (8 consecutive k of channel
// l & 15; tools/tr16_probe.hip pins the mapping).  The row address is per lane, so the implicit
// im2col (pixel -> input position of the tap, any stride) costs one add per tap.
//
// Decomposition: workgroup = (group of G consecutive frames, kernel row ky); its 8 waves hold the
// [KW*Cin, Cout] block of dW for that ky in accumulators across the G frames (wave = 2 column tiles
// x RTW row tiles), then write one fp32 slab; aa_splitk_reduce_kernel sums the n_img / G slabs
// (and the column sums of dZ = the bias gradient, accumulated during staging by the ky = 0 groups)
// in fixed order.  The next frame's global loads are in flight, in registers, while the current
// one is multiplied.  Workgroups of one frame group sit on one XCD (they read the same frames).
#include "common.h"
#include "agents_amd.h"
#include "x6_common.h"
#include "splitk_reduce.h"

#include <type_traits>

#define AA_DW6_THREADS 512
#define AA_DW6_NXI 2     /* staging rounds of 512 x (8 channels) for the x rows of one ky */
#define AA_DW6_NZI 2     /* ... for the dZ frame */
#define AA_DW6_MAX_KS 4  /* 32-pixel k-steps per frame */
#define AA_DW6_TP 68     /* floats per row of the epilogue's slab tile in LDS */

struct Dw6P {
  const float* x;     // [n_img][H*W][Cin]
  const float* dz;    // [n_img][OH*OW][Cout]
  float* slab;        // [groups][KH*KW*Cin][Cout], then [groups][Cout] (column sums of dZ)
  int n_img, H, W, Cin, KH, KW, stride, OH, OW, Cout;
  int G, groups, per_xcd, want_db;
  int xpitch;           // bytes per pixel of the bf16 x images (channels * 2 + pad)
  int zpair;            // bytes per PAIR of dZ pixels (2 x 128 + pad, see dw6_plan)
  int xrow;             // bytes per image row of the x planes (W * xpitch + pad, see dw6_plan)
  int xplane, zplane;   // bytes per plane
  int xsh, zsh;         // log2(Cin / 8), log2(Cout / 8)
  unsigned m_ow, m_w;   // ceil(2^16 / OW), ceil(2^16 / W)
#ifdef AA_DW6_DEBUG
  int dbg;              // tools/dw6_probe.hip ablations: 1 = no staging, 2 = no multiply, 4 = no slab
#endif
};
#ifdef AA_DW6_DEBUG
#define AA_DW6_DBG(P, bit) (((P).dbg & (bit)) != 0)
static int g_dw6_dbg = 0, g_dw6_xpad = -1, g_dw6_zpad = -1, g_dw6_xrowpad = -1;
#else
#define AA_DW6_DBG(P, bit) false
#endif

// The transpose read is issued WITHOUT a wait; dw6_wait6 is the s_waitcnt, and names the
// destination registers as read-write operands so that the compiler places every use after it.
__device__ static inline uint2 dw6_tr(uint32_t lds_addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
  return v;
}
typedef float dw6_f32x4 __attribute__((ext_vector_type(4)));

__device__ static inline void dw6_store16(char* d, const uint4& v) {
#ifdef AA_DW6_DEBUG   // pitch sweeps of tools/dw6_probe.hip: rows may be 8 mod 16
  *reinterpret_cast<uint2*>(d) = make_uint2(v.x, v.y);
  *reinterpret_cast<uint2*>(d + 8) = make_uint2(v.z, v.w);
#else
  *reinterpret_cast<uint4*>(d) = v;
#endif
}

template <int N>
__device__ static inline void dw6_wait6(uint2& a, uint2& b, uint2& c, uint2& d, uint2& e,
                                        uint2& f) {
  constexpr int n = N > 15 ? 15 : N;      // the counter field has four bits
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(n));
}

template <int RTW, int KS>
__global__ void __launch_bounds__(AA_DW6_THREADS) aa_conv_dw_frame_x6_kernel(Dw6P P) {
  extern __shared__ __attribute__((aligned(16))) char dw6_lds[];
  // block -> (frame group, ky): the KH workgroups of a group are consecutive on ONE XCD
  const int v = (blockIdx.x & 7) * P.per_xcd + (blockIdx.x >> 3);
  if (v >= P.groups * P.KH) return;
  const int fg = v / P.KH, ky = v - fg * P.KH;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, g = lane >> 4, j = lane & 15;
  const int cp = wave & 1, rg = wave >> 1;
  char* xpl = dw6_lds;
  char* zpl = dw6_lds + 3 * (size_t)P.xplane;
  const int OHW = P.OH * P.OW, HW = P.H * P.W;
  const int zoct = P.Cout >> 3;
  // only the OH input rows oy * s + ky enter this workgroup's products: the x image is [OH][W]
  const int nx_item = (P.OH * P.W) << P.xsh, nz_item = OHW << P.zsh;
  // the dZ planes' pad rows (pixels OHW .. 32 KS) stay zero for the whole launch
  {
    uint4* z = reinterpret_cast<uint4*>(zpl);
    const int n16 = (3 * P.zplane) >> 4;
    for (int i = tid; i < n16; i += AA_DW6_THREADS) z[i] = make_uint4(0, 0, 0, 0);
  }
  // per-lane byte offsets of the k rows this lane addresses (see the header): x rows are the
  // input pixels under tap (ky, 0) of output pixel p, dZ rows are p itself
  int xoff[KS][2], zoff[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // Which pixel is "k" is free as long as both operands agree.  LDS serves a transpose read
      // in two halves of 32 lanes = two 16-lane groups = 8 rows of 32 bytes: with k = 8 g + 4 h + r
      // mapped to pixel 16 h + 4 g + r those are 8 CONSECUTIVE pixels, and a pixel stride of
      // 32 (mod 64) bytes spreads them over all 64 banks (k -> 8 g + 4 h + r itself would put
      // pixels p and p + 8 in one half: same banks whatever the pitch).  That is the x planes
      // (with image rows padded so that the stride holds across the end of an output row); the
      // dZ planes reach the same with pixel pairs (dw6_plan).
      const int p = 32 * ks + 16 * h + 4 * g + (j >> 2);
      const int pc = p < OHW ? p : OHW - 1;       // pad pixels: any valid x row (their dZ is 0)
      const int oy = cx_div(pc, P.m_ow), ox = pc - oy * P.OW;
      xoff[ks][h] = oy * P.xrow + ox * P.stride * P.xpitch + (j & 3) * 8;
      zoff[ks][h] = (p >> 1) * P.zpair + (p & 1) * 128 + (j & 3) * 8;
    }
  // this wave's row tiles: rows (kx, ci) of the ky block, 16 per tile
  int toff[RTW];
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt) {
    const int r0 = (rg * RTW + rt) * 16;
    const int kx = r0 / P.Cin, ci0 = r0 - kx * P.Cin;
    toff[rt] = kx * P.xpitch + ci0 * 2;
  }
  const uint32_t xbase = (uint32_t)(uintptr_t)xpl, zbase = (uint32_t)(uintptr_t)zpl;
  cx_f32x4 big[2][RTW], small[2][RTW];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
      big[c][rt] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
      small[c][rt] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
    }
  float dbsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbsum[e] = 0.f;
  const bool do_db = P.want_db != 0 && ky == 0;

  const int f0 = fg * P.G, f1 = (f0 + P.G < P.n_img) ? f0 + P.G : P.n_img;
  float4 px[AA_DW6_NXI][2], pz[AA_DW6_NZI][2];
  auto fetch = [&](int img) {
    const float4* xs = reinterpret_cast<const float4*>(P.x + (size_t)img * HW * P.Cin);
    const float4* zs = reinterpret_cast<const float4*>(P.dz + (size_t)img * OHW * P.Cout);
#pragma unroll
    for (int u = 0; u < AA_DW6_NXI; ++u) {
      int it = tid + u * AA_DW6_THREADS;
      if (it >= nx_item) it = nx_item - 1;
      const int q = it >> P.xsh, o = it & ((1 << P.xsh) - 1);
      const int oy = cx_div(q, P.m_w), xw = q - oy * P.W;
      const int src = ((((oy * P.stride + ky) * P.W + xw) << P.xsh) + o) * 2;
      px[u][0] = xs[src];
      px[u][1] = xs[src + 1];
    }
#pragma unroll
    for (int u = 0; u < AA_DW6_NZI; ++u) {
      int it = tid + u * AA_DW6_THREADS;
      if (it >= nz_item) it = nz_item - 1;
      pz[u][0] = zs[2 * it];
      pz[u][1] = zs[2 * it + 1];
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < AA_DW6_NXI; ++u) {
      const int it = tid + u * AA_DW6_THREADS;
      if (it >= nx_item) continue;
      const int q = it >> P.xsh, o = it & ((1 << P.xsh) - 1);
      const float a[8] = {px[u][0].x, px[u][0].y, px[u][0].z, px[u][0].w,
                          px[u][1].x, px[u][1].y, px[u][1].z, px[u][1].w};
      uint4 f[3];
      cx_split8(a, f);
      const int qr = cx_div(q, P.m_w);
      char* d = xpl + qr * P.xrow + (q - qr * P.W) * P.xpitch + o * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dw6_store16(d + pl * P.xplane, f[pl]);
    }
#pragma unroll
    for (int u = 0; u < AA_DW6_NZI; ++u) {
      const int it = tid + u * AA_DW6_THREADS;
      if (it >= nz_item) continue;
      const int q = it >> P.zsh, o = it & (zoct - 1);
      const float a[8] = {pz[u][0].x, pz[u][0].y, pz[u][0].z, pz[u][0].w,
                          pz[u][1].x, pz[u][1].y, pz[u][1].z, pz[u][1].w};
      if (do_db) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dbsum[e] += a[e];
      }
      uint4 f[3];
      cx_split8(a, f);
      char* d = zpl + (q >> 1) * P.zpair + (q & 1) * 128 + o * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dw6_store16(d + pl * P.zplane, f[pl]);
    }
  };

  fetch(f0);
  for (int img = f0; img < f1; ++img) {
    __syncthreads();   // zero fill done / the previous frame's readers are done
    if (!AA_DW6_DBG(P, 1)) stage();
    if (img + 1 < f1) fetch(img + 1);   // in flight during the multiply
    __syncthreads();
    if (AA_DW6_DBG(P, 2)) continue;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint2 rb[2][3][2], ra[RTW][3][2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint32_t base = zbase + pl * P.zplane + (cp * 2 + c) * 32;
          rb[c][pl][0] = dw6_tr(base + zoff[ks][0]);
          rb[c][pl][1] = dw6_tr(base + zoff[ks][1]);
        }
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint32_t base = xbase + pl * P.xplane + toff[rt];
          ra[rt][pl][0] = dw6_tr(base + xoff[ks][0]);
          ra[rt][pl][1] = dw6_tr(base + xoff[ks][1]);
        }
      // LDS reads return in order: the dZ fragments and row tile rt are there once at most
      // 6 (RTW - 1 - rt) reads are outstanding -- the MFMAs of a tile overlap the arrival of the next
      dw6_wait6<6 * (RTW - 1)>(rb[0][0][0], rb[0][0][1], rb[0][1][0], rb[0][1][1], rb[0][2][0],
                               rb[0][2][1]);
      dw6_wait6<6 * (RTW - 1)>(rb[1][0][0], rb[1][0][1], rb[1][1][0], rb[1][1][1], rb[1][2][0],
                               rb[1][2][1]);
      CxFrag b[2][3];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[c][pl].q = make_uint4(rb[c][pl][0].x, rb[c][pl][0].y, rb[c][pl][1].x, rb[c][pl][1].y);
      auto tile = [&](auto rtc) {
        constexpr int rt = decltype(rtc)::value;
        dw6_wait6<6 * (RTW - 1 - rt)>(ra[rt][0][0], ra[rt][0][1], ra[rt][1][0], ra[rt][1][1],
                                      ra[rt][2][0], ra[rt][2][1]);
        CxFrag a[1][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[0][pl].q = make_uint4(ra[rt][pl][0].x, ra[rt][pl][0].y, ra[rt][pl][1].x,
                                  ra[rt][pl][1].y);
        cx_f32x4 bg[1], sm[1];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bg[0] = big[c][rt]; sm[0] = small[c][rt];
          cx_mma6<1>(a, b[c], bg, sm);
          big[c][rt] = bg[0]; small[c][rt] = sm[0];
        }
      };
      tile(std::integral_constant<int, 0>{});
      if constexpr (RTW > 1) tile(std::integral_constant<int, 1>{});
      if constexpr (RTW > 2) tile(std::integral_constant<int, 2>{});
      if constexpr (RTW > 3) tile(std::integral_constant<int, 3>{});
    }
  }

  // ---- slab: rows (kx, ci) of this ky, columns co.  The accumulators (D layout: lane (n = j,
  // rows 4 g .. 4 g + 3)) go through LDS so that the slab is written in whole 256-byte rows,
  // 16 bytes per lane (4-byte stores in 64-byte runs took 7 of the kernel's 19 us).
  const int M = P.KH * P.KW * P.Cin;
  const int rows_blk = P.KW * P.Cin;
  float* out = P.slab + ((size_t)fg * M + (size_t)ky * rows_blk) * P.Cout;
  if (AA_DW6_DBG(P, 4) && big[0][0][0] != 12345.f) return;
  __syncthreads();     // every wave is done with the planes
  {
    // [rows_blk][AA_DW6_TP]: a store's four lane groups are rows 4 apart -- at 68 floats per row
    // 16 banks apart (at 80: the same 16 banks, a 4-way conflict on every store)
    float* tile = reinterpret_cast<float*>(dw6_lds);
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = (rg * RTW + rt) * 16 + 4 * g + e;
          tile[row * AA_DW6_TP + (cp * 2 + c) * 16 + j] = big[c][rt][e] + small[c][rt][e];
        }
    __syncthreads();
    const int n4 = rows_blk * 16;          // float4 items: 16 per row (Cout = 64)
    for (int q = tid; q < n4; q += AA_DW6_THREADS) {
      const int row = q >> 4, c4 = q & 15;
      // streamed (non-temporal, 16 bytes per lane): the next reader is another kernel -- the slab
      // reduce -- on every XCD
      const dw6_f32x4 v4 = *reinterpret_cast<const dw6_f32x4*>(tile + row * AA_DW6_TP + c4 * 4);
      __builtin_nontemporal_store(v4, reinterpret_cast<dw6_f32x4*>(out + (size_t)row * P.Cout +
                                                                   c4 * 4));
    }
  }
  // ---- column sums of dZ over this group's frames (ky = 0 workgroups) -----------------------------
  if (do_db) {
    __syncthreads();    // every wave is done with the planes: reuse them as scratch
    float* red = reinterpret_cast<float*>(dw6_lds);       // [512][8]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = dbsum[e];
    __syncthreads();
    if (tid < P.Cout) {
      // thread t staged octet (t' mod zoct) of its items: co = 8 o + e lives in threads with
      // t' mod zoct == o, summed here in thread order
      const int o = tid >> 3, e = tid & 7;
      float s = 0.f;
      for (int t = o; t < AA_DW6_THREADS; t += zoct) s += red[t * 8 + e];
      P.slab[(size_t)P.groups * M * P.Cout + (size_t)fg * P.Cout + tid] = s;
    }
  }
}

// ---- host ------------------------------------------------------------------------------------------
struct Dw6Plan {
  Dw6P P;
  size_t lds, ws;
  int rtw, ks;
};

static int dw6_plan(const aa_conv_dx_desc* d, Dw6Plan* pl) {
  // (the descriptor of the input gradient describes the same layer: dz, w-shape, x-shape)
  if (d == nullptr || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->KH <= 0 ||
      d->KW <= 0 || d->stride <= 0 || d->Cout <= 0)
    return AA_ERR_INVALID;
  if (d->H < d->KH || d->W < d->KW) return AA_ERR_INVALID;
  Dw6P& P = pl->P;
  const int s = d->stride;
  P.n_img = d->n_img; P.H = d->H; P.W = d->W; P.Cin = d->Cin; P.KH = d->KH; P.KW = d->KW;
  P.stride = s; P.Cout = d->Cout;
  P.OH = (d->H - d->KH) / s + 1; P.OW = (d->W - d->KW) / s + 1;
  const int OHW = P.OH * P.OW;
  if (d->Cout != 64 || d->Cin % 16 != 0) return AA_ERR_RANGE;          // 2 x 2 column tiles
  const int row_tiles = d->KW * d->Cin / 16;
  if (row_tiles % 4 != 0 || row_tiles / 4 > 4) return AA_ERR_RANGE;    // 4 row groups of RTW tiles
  pl->rtw = row_tiles / 4;
  pl->ks = (OHW + 31) / 32;
  if (pl->ks > AA_DW6_MAX_KS) return AA_ERR_RANGE;
  if (P.OH * d->W * (d->Cin / 8) > AA_DW6_NXI * AA_DW6_THREADS) return AA_ERR_RANGE;
  if (OHW * (d->Cout / 8) > AA_DW6_NZI * AA_DW6_THREADS) return AA_ERR_RANGE;
  if ((int64_t)P.OW * OHW >= 65536) return AA_ERR_RANGE;               // cx_div range
  if ((d->Cin & (d->Cin - 1)) != 0) return AA_ERR_RANGE;               // shifts in the staging loops
  P.xsh = 0;
  while ((8 << P.xsh) < d->Cin) ++P.xsh;
  P.zsh = 3;                                                           // Cout = 64
  // x pixel pitch: a multiple of 16 bytes (16-byte staging stores) whose stride between
  // consecutive output pixels (s pitches) is 32 mod 64 bytes -- see the kernel
  auto pick = [](int bytes, int stride) {
    for (int pad = 0; pad <= 112; pad += 16)
      if (((bytes + pad) * stride) % 64 == 32) return bytes + pad;
    return bytes + 16;
  };
  P.xpitch = pick(d->Cin * 2, s);
  // dZ (128 bytes per pixel, read at consecutive pixels from a multiple of 8): pixel PAIRS of 256
  // bytes, 32 bytes of pad per pair.  A staging store's 16 lanes write one pair = every bank once
  // (at one pitch per pixel their 2 x 128 bytes + pad wrapped around 256 onto themselves), and a
  // read's 8 pixels sit at 32-byte slots k + {0, 4} (mod 8), k = 0 .. 3: every bank once as well.
  P.zpair = 288;
#ifdef AA_DW6_DEBUG
  if (g_dw6_xpad >= 0) P.xpitch = d->Cin * 2 + g_dw6_xpad;
  if (g_dw6_zpad >= 0) P.zpair = 256 + g_dw6_zpad;
#endif
  // image rows of the x planes: the 8 consecutive output pixels of a read's half wave usually
  // cross an image row (OW = 7 or 9 here); their 32-byte rows keep landing on 8 different bank
  // octets if stepping over the row end moves the address like one more pixel step does,
  // mod 256 bytes: xrow = OW * s * xpitch (mod 256).  (At xrow = W * xpitch nearly every x read
  // had a 2-way conflict: half of the kernel's LDS conflict cycles.)
  P.xrow = d->W * P.xpitch;
  while ((P.xrow - P.OW * s * P.xpitch) % 256 != 0) P.xrow += 16;
#ifdef AA_DW6_DEBUG
  if (g_dw6_xrowpad >= 0) P.xrow = d->W * P.xpitch + g_dw6_xrowpad;
#endif
  P.xplane = P.OH * P.xrow;
  P.zplane = pl->ks * 16 * P.zpair;
  pl->lds = 3 * (size_t)P.xplane + 3 * (size_t)P.zplane;
  const size_t tile_bytes = (size_t)d->KW * d->Cin * AA_DW6_TP * 4;    // the slab tile (epilogue)
  if (pl->lds < tile_bytes) pl->lds = tile_bytes;
  if (pl->lds < (size_t)AA_DW6_THREADS * 8 * 4) pl->lds = (size_t)AA_DW6_THREADS * 8 * 4;
  if (pl->lds > 160 * 1024) return AA_ERR_RANGE;
  if ((int64_t)P.OH * d->W * d->W >= 65536) return AA_ERR_RANGE;       // cx_div range
  P.m_ow = (65536u + P.OW - 1) / P.OW;
  P.m_w = (65536u + d->W - 1) / d->W;
  // frames per workgroup: the fewest that keep the launch within one workgroup per CU (256 CUs:
  // a 257th workgroup would run alone after the others)
  int G = 1;
  while (((d->n_img + G - 1) / G) * d->KH > 256) ++G;
  P.G = G;
  P.groups = (d->n_img + G - 1) / G;
  P.per_xcd = (P.groups * d->KH + 7) / 8;
  const size_t M = (size_t)d->KH * d->KW * d->Cin;
  pl->ws = ((size_t)P.groups * M * d->Cout + (size_t)P.groups * d->Cout) * sizeof(float);
  return AA_OK;
}

extern "C" {

int64_t aa_conv_dw_frame_x6_workspace_bytes(const aa_conv_dx_desc* d) {
  Dw6Plan pl;
  return dw6_plan(d, &pl) == AA_OK ? (int64_t)pl.ws : 0;
}

// Main launch only: the per-group slabs (+ the bias-gradient rows when want_db) are left in
// `workspace` for aa_conv_dw_frame_x6_reduce.
static int dw6_main(const aa_conv_dx_desc* d, const float* x, int want_db, void* workspace,
                    int64_t workspace_bytes, void* stream, Dw6Plan* pl_out) {
  Dw6Plan pl;
  const int rc = dw6_plan(d, &pl);
  if (rc != AA_OK) return rc;
  if (x == nullptr || d->dz == nullptr || workspace == nullptr) return AA_ERR_INVALID;
  if ((((uintptr_t)x | (uintptr_t)d->dz | (uintptr_t)workspace) & 15) != 0) return AA_ERR_INVALID;
  if ((int64_t)pl.ws > workspace_bytes) return AA_ERR_RANGE;
  Dw6P& P = pl.P;
  P.x = x; P.dz = d->dz; P.slab = (float*)workspace; P.want_db = want_db;
#ifdef AA_DW6_DEBUG
  P.dbg = g_dw6_dbg;
#endif
  hipStream_t st = (hipStream_t)stream;
  const int grid = P.per_xcd * 8;
  static size_t lds_limit[AA_MAX_DEVICES][5][5] = {{{0}}};   // > 64 KiB of dynamic LDS: granted
  const int dv = aa_device_ordinal();                          // once per kernel and device
  if (dv < 0) return AA_ERR_LAUNCH;
  int done = 0;
#define AA_DW6_CASE(R_, K_)                                                                     \
  if (pl.rtw == R_ && pl.ks == K_) {                                                            \
    if (pl.lds > 65536 && pl.lds > lds_limit[dv][R_][K_]) {                                         \
      if (hipFuncSetAttribute((const void*)aa_conv_dw_frame_x6_kernel<R_, K_>,                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) !=       \
          hipSuccess)                                                                           \
        return AA_ERR_LAUNCH;                                                                   \
      lds_limit[dv][R_][K_] = pl.lds;                                                             \
    }                                                                                           \
    hipLaunchKernelGGL((aa_conv_dw_frame_x6_kernel<R_, K_>), dim3(grid), dim3(AA_DW6_THREADS),  \
                       pl.lds, st, P);                                                          \
    done = 1;                                                                                   \
  }
#define AA_DW6_ROW(R_) AA_DW6_CASE(R_, 1) AA_DW6_CASE(R_, 2) AA_DW6_CASE(R_, 3) AA_DW6_CASE(R_, 4)
  AA_DW6_ROW(1) AA_DW6_ROW(2) AA_DW6_ROW(3) AA_DW6_ROW(4)
#undef AA_DW6_ROW
#undef AA_DW6_CASE
  if (!done) return AA_ERR_RANGE;
  if (aa_launch_status() != AA_OK) return AA_ERR_LAUNCH;
  if (pl_out != nullptr) *pl_out = pl;
  return AA_OK;
}

int aa_conv_dw_frame_x6_slabs(const aa_conv_dx_desc* d, const float* x, int32_t want_db,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  return dw6_main(d, x, want_db, workspace, workspace_bytes, stream, nullptr);
}

// Deterministic sum of the per-group slabs (+ the bias-gradient rows that follow them) of up to
// four layers in ONE launch; layer l: slabs in workspaces[l] (as left by aa_conv_dw_frame_x6_slabs
// for descs[l]), result in dws[l] / dbs[l] (nullable).  Bit-identical to one reduce launch each.
int aa_conv_dw_frame_x6_reduce(int32_t n_layers, const aa_conv_dx_desc* const* descs,
                               const void* const* workspaces, float* const* dws,
                               float* const* dbs, void* stream) {
  if (n_layers < 1 || n_layers > AA_REDUCE_MAX_SEGS || !descs || !workspaces || !dws || !dbs)
    return AA_ERR_INVALID;
  AaReduceSegs g;
  g.n = n_layers;
  int blocks_total = 0;
  bool deep_all = true;
  for (int l = 0; l < n_layers; ++l) {
    Dw6Plan pl;
    const int rc = dw6_plan(descs[l], &pl);
    if (rc != AA_OK) return rc;
    if (workspaces[l] == nullptr || dws[l] == nullptr) return AA_ERR_INVALID;
    if ((((uintptr_t)workspaces[l] | (uintptr_t)dws[l]) & 15) != 0) return AA_ERR_INVALID;
    const int M = descs[l]->KH * descs[l]->KW * descs[l]->Cin, N = descs[l]->Cout;
    const size_t work = ((size_t)M * N + (dbs[l] != nullptr ? N : 0)) / 4;
    const bool deep = pl.P.groups >= 32 && work <= 65536;
    deep_all = deep_all && deep;
    g.slab[l] = (const float*)workspaces[l];
    g.splits[l] = pl.P.groups; g.M[l] = M; g.N[l] = N;
    g.C[l] = dws[l]; g.colsum[l] = dbs[l];
    g.first[l] = blocks_total;
    // (block counts of aa_conv_dw_frame_x6's own reduce launch: same grid-stride walk per segment)
    int blocks = (int)((work + (deep ? 16 : 256) - 1) / (deep ? 16 : 256));
    if (blocks > 2048) blocks = 2048;
    blocks_total += blocks;
    g.first[l + 1] = blocks_total;
  }
  // the z-lane count is a template parameter: mixed "deep" / flat layers take one launch each
  if (n_layers > 1 && !deep_all) {
    for (int l = 0; l < n_layers; ++l) {
      const int rc = aa_conv_dw_frame_x6_reduce(1, descs + l, workspaces + l, dws + l, dbs + l,
                                                stream);
      if (rc != AA_OK) return rc;
    }
    return AA_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  if (deep_all)
    hipLaunchKernelGGL((aa_splitk_reduce_multi_kernel<4, 16>), dim3(blocks_total), dim3(256), 0,
                       st, g);
  else
    hipLaunchKernelGGL((aa_splitk_reduce_multi_kernel<4, 1>), dim3(blocks_total), dim3(256), 0,
                       st, g);
  return aa_launch_status();
}

int aa_conv_dw_frame_x6(const aa_conv_dx_desc* d, const float* x, float* dw, float* db,
                        void* workspace, int64_t workspace_bytes, void* stream) {
  if (dw == nullptr || (((uintptr_t)dw) & 15) != 0) return AA_ERR_INVALID;
  int rc = dw6_main(d, x, db != nullptr, workspace, workspace_bytes, stream, nullptr);
  if (rc != AA_OK) return rc;
  const void* ws = workspace;
  return aa_conv_dw_frame_x6_reduce(1, &d, &ws, &dw, &db, stream);
}

}  // extern "C"
