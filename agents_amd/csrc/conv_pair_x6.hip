// The conv -> conv pair of csrc/conv_pair.hip (keras Conv2D + activation twice: conv2 -> conv3 of
// the Mnih-15 Q-network, examples/dqn/mnih15/dqn_train_eval_atari.py:80-112) on the bf16 matrix
// cores at fp32 accuracy: one workgroup per frame, both activations written.
//
// Arithmetic (the identity proven for the conv1 kernels, conv_u8_bf16.h, and
// restated on the CPU in oracle/numerics.py): an fp32 value is EXACTLY hi + mid + lo, three
// round-to-nearest bf16 pieces; a product of two bf16 numbers is exact in fp32; of the nine piece
// products of x * w the three smallest (< 2^-27 |x w|) are below fp32 rounding and dropped; the
// other six are accumulated in fp32 by v_mfma_f32_16x16x32_bf16, the five small ones in one
// accumulator and hi * hi in another, added at the end.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA
// of the same shape.
//
// What makes it pay here and not in a generic GEMM (a round-1 forward plan that split both fp32
// operands in registers tied the fp32 MFMA kernels -- "the in-register split of the activations
// costs what the MFMAs save" -- and was removed in round 3): in this kernel every operand element is split ONCE.
//   * the input frame is split while it is staged into LDS (three bf16 planes, one ds_write_b128
//     per 8 channels and plane) -- each element is then READ as a ready fragment by every patch
//     that covers it (conv2: x4, conv3: x9) and every filter tile;
//   * the first layer's output is split in its epilogue, straight into the second layer's LDS
//     planes;
//   * the filters are split by a small pre-pass (aa_conv_pair_x6_split_kernel, 68 K weights for
//     the Atari pair) into fragment order in a caller-provided scratch, so a B fragment is one
//     coalesced 16-byte load per lane and plane from L2.
// The pre-pass runs at EVERY call: nothing has to track when the parameters change.
//
// Machine mapping: K is walked tap by tap in steps of 32 input channels = one MFMA.  Wave w owns
// the 16-filter column tiles {w & 3, (w & 3) + 4, ...} and one half (w >> 2) of the frame's
// 16-pixel row tiles, as in conv_pair.hip.  A fragment (lane: pixel l & 15, channel octet l >> 4)
// = one ds_read_b128 per plane at (patch origin + tap offset); the LDS pixel pitch and row pitch
// are padded (searched on the host against the ds_read_b128 lane-group / 64-bank model of
// MI355X_MICROARCH.md) so that the reads of a row tile are bank-conflict free.
#include "common.h"
#include "agents_amd.h"
#include "x6_common.h"

#include <stdlib.h>
#include <type_traits>

#define AA_CX_THREADS 512
#define AA_CX_MAX_RT 8
#define AA_CX_MAX_KS 64       /* k-steps (32 input channels of one tap) per layer */

struct CxLayer {
  const float* w;      // [KH][KW][Cin][Cout] fp32
  const float* bias;   // nullable
  float* y;            // [n_img][OH*OW][Cout]
  uint4* wf;           // split filters: [kstep][col tile][plane][lane] x 8 bf16
  int H, W, Cin, KH, KW, stride, OH, OW, Cout, act;
  int pitch, rowp, plane;   // bytes per input pixel / input row / plane of this layer's LDS frame
  int ksteps, cgs;          // KH*KW*Cin/32 MFMA steps, Cin/32 of them per tap
  unsigned m_ow, m_w;       // ceil(2^16 / OW), ceil(2^16 / W): n / d == (n * m) >> 16 for the
                            // pixel indices of a frame (n * d < 2^16: both < 2^8 here, checked)
  int tap[AA_CX_MAX_KS];    // byte offset of every k-step inside a patch (host-built: the main
                            // loop is then free of tap / channel-group bookkeeping and branches)
};


struct CxParams {
  const float* x;      // [n_img][H*W*Cin], image pitch img_pitch floats
  int64_t img_pitch;
  int n_img;
  CxLayer l[2];
#ifdef AA_CX_STAMPS
  long long* stamps;   // tools/cx_probe.hip: [workgroup][wave][8] wall_clock64 ticks (10 ns)
#endif
};

#ifdef AA_CX_STAMPS
static long long* g_cx_stamps = nullptr;
__device__ long long* d_cx_stamps = nullptr;    // same buffer, reachable from cx_layer
#define CX_STAMP_L(i)                                                                      \
  if (d_cx_stamps != nullptr && (threadIdx.x & 63) == 0)                                   \
    d_cx_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();
#define CX_STAMP(i)                                                                        \
  if (P.stamps != nullptr && (threadIdx.x & 63) == 0)                                      \
    P.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();
#else
#define CX_STAMP(i)
#define CX_STAMP_L(i)
#endif

__device__ static inline float cx_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}

// ---- filter pre-pass ------------------------------------------------------------------------
// item = (kstep, col tile, lane): lane (c = l & 15, g = l >> 4) holds k = 32 kstep + 8 g + e,
// e = 0..7, of filter column 16 ct + c; k indexes the HWI axes of the HWIO bank flattened.
__global__ void __launch_bounds__(256) aa_conv_pair_x6_split_kernel(CxLayer L0, CxLayer L1) {
  const int n0 = L0.ksteps * (L0.Cout >> 4) * 64;
  const int n1 = L1.ksteps * (L1.Cout >> 4) * 64;
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < n0 + n1;
       it += gridDim.x * blockDim.x) {
    const bool second = it >= n0;
    const CxLayer& L = second ? L1 : L0;
    const int i = second ? it - n0 : it;
    const int lane = i & 63, c = lane & 15, g = lane >> 4;
    const int nct = L.Cout >> 4;
    const int t = i >> 6, ct = t % nct, ks = t / nct;
    const float* src = L.w + (size_t)(ks * 32 + g * 8) * L.Cout + ct * 16 + c;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = src[(size_t)e * L.Cout];
    uint4 f[3];
    cx_split8(a, f);
#pragma unroll
    for (int s = 0; s < 3; ++s) L.wf[((size_t)t * 3 + s) * 64 + lane] = f[s];
  }
}

// ---- one layer of one frame -------------------------------------------------------------------
// src: this layer's three LDS planes; results to global y, and (dst != nullptr) split into the
// next layer's planes.  RT row tiles per wave (compile time).
template <int RT>
__device__ static inline void cx_layer(const CxLayer& L, const char* __restrict__ src, int img,
                                       char* __restrict__ dst, const CxLayer& Ln) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int OHW = L.OH * L.OW;
  const int nct = L.Cout >> 4;
  const int rt0 = (wave >> 2) * RT;
  int pb[RT];          // byte offset of this lane's patch origin (+ its channel octet)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int p = (rt0 + rt) * 16 + lr;
    if (p >= OHW) p = OHW - 1;
    const int oy = cx_div(p, L.m_ow), ox = p - oy * L.OW;
    pb[rt] = oy * L.stride * L.rowp + ox * L.stride * L.pitch + lg * 16;
  }
  // where this lane's four output pixels of every row tile go (epilogue)
  int dpix[RT][4];
  if (dst != nullptr) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int p = (rt0 + rt) * 16 + 4 * lg + e;
        if (p >= OHW) p = OHW - 1;
        const int py = cx_div(p, L.m_ow), px = p - py * L.OW;
        dpix[rt][e] = py * Ln.rowp + px * Ln.pitch;
      }
  }
  const size_t wstep = (size_t)nct * 3 * 64;    // uint4 between k-steps of the split bank
  for (int ct = wave & 3; ct < nct; ct += 4) {
    cx_f32x4 big[RT], small[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      big[rt] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
      small[rt] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint4* wp = L.wf + (size_t)ct * 3 * 64 + lane;
    // the bias of this lane's output channel is requested in FRONT of the k loop (it was the first
    // dependent access of the epilogue: an exposed L2 round trip per layer); unconditional load
    // from a valid address, selected afterwards
    const int co = ct * 16 + lr;
    const float bias_raw = (L.bias != nullptr ? L.bias : L.w)[co];
    auto load_b = [&](CxFrag (&b)[3], int ks) {
#pragma unroll
      for (int s = 0; s < 3; ++s) b[s].q = wp[(size_t)ks * wstep + s * 64];
    };
    auto load_a = [&](CxFrag (&a)[RT][3], int off) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          a[rt][s].q = *reinterpret_cast<const uint4*>(src + s * L.plane + pb[rt] + off);
    };
    auto mma = [&](CxFrag (&a)[RT][3], CxFrag (&b)[3]) {
      // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        small[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][2].v, b[0].v, small[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        small[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][0].v, b[2].v, small[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        small[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][1].v, b[1].v, small[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        small[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][1].v, b[0].v, small[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        small[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][0].v, b[1].v, small[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        big[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][0].v, b[0].v, big[rt], 0, 0, 0);
    };
    if (dst != nullptr) { CX_STAMP_L(6) }
    // Filter fragments run FOUR k-steps ahead in a register ring (an L2 hit under load is
    // ~500-800 cycles, two k-steps of MFMA issue), patch fragments one step ahead (LDS).  All
    // refills are unconditional, with the index clamped to the last k-step: no branches in the
    // body.  Ring slot = k-step & 3; the 0-3 leftover steps reuse the slots in order.
    const int S = L.ksteps, last = S - 1;
    CxFrag a0[RT][3], a1[RT][3], b[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b(b[j], j < last ? j : last);
    load_a(a0, L.tap[0]);
    auto step = [&](auto jc, int ks) {
      constexpr int J = decltype(jc)::value;
      const int kn = ks + J + 1 < last ? ks + J + 1 : last;
      const int kb = ks + J + 4 < last ? ks + J + 4 : last;
      if (J & 1) {
        load_a(a0, L.tap[kn]);
        mma(a1, b[J]);
      } else {
        load_a(a1, L.tap[kn]);
        mma(a0, b[J]);
      }
      load_b(b[J], kb);
    };
    int ks = 0;
    for (; ks + 3 < S; ks += 4) {
      step(std::integral_constant<int, 0>{}, ks);
      step(std::integral_constant<int, 1>{}, ks);
      step(std::integral_constant<int, 2>{}, ks);
      step(std::integral_constant<int, 3>{}, ks);
    }
    if (ks < S) step(std::integral_constant<int, 0>{}, ks);
    if (ks + 1 < S) step(std::integral_constant<int, 1>{}, ks);
    if (ks + 2 < S) step(std::integral_constant<int, 2>{}, ks);
    if (dst != nullptr) { CX_STAMP_L(7) }
    const float bv = L.bias != nullptr ? bias_raw : 0.f;
    // (L.y == nullptr: an output only a backward pass would read, and no backward pass follows)
    float* yimg = L.y != nullptr ? L.y + (size_t)img * OHW * L.Cout + co : nullptr;
    // the activation kind and "feeds a next layer" are resolved ONCE, outside the element loop
    auto emit = [&](auto actc, auto splitc) {
      constexpr int ACT = decltype(actc)::value;
      constexpr bool SPLIT = decltype(splitc)::value;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = (rt0 + rt) * 16 + 4 * lg + e;
          if (p >= OHW) continue;
          float v = (big[rt][e] + small[rt][e]) + bv;
          if (ACT == AA_ACT_RELU) v = v > 0.f ? v : 0.f;
          if (ACT == AA_ACT_TANH) v = tanhf(v);
          if (yimg != nullptr) yimg[p * L.Cout] = v;
          if (SPLIT) {
            // v = hi + mid + lo exactly; one 2-byte store per plane at [pixel][channel]
            const unsigned h = cx_pk_bf16(v, 0.f) & 0xffffu;
            const float r1 = v - __uint_as_float(h << 16);
            const unsigned m = cx_pk_bf16(r1, 0.f) & 0xffffu;
            const float r2 = r1 - __uint_as_float(m << 16);
            const unsigned l = cx_pk_bf16(r2, 0.f) & 0xffffu;
            char* d = dst + dpix[rt][e] + co * 2;
            *reinterpret_cast<unsigned short*>(d) = (unsigned short)h;
            *reinterpret_cast<unsigned short*>(d + Ln.plane) = (unsigned short)m;
            *reinterpret_cast<unsigned short*>(d + 2 * Ln.plane) = (unsigned short)l;
          }
        }
      }
    };
    auto emit_act = [&](auto splitc) {
      if (L.act == AA_ACT_RELU) emit(std::integral_constant<int, AA_ACT_RELU>{}, splitc);
      else if (L.act == AA_ACT_TANH) emit(std::integral_constant<int, AA_ACT_TANH>{}, splitc);
      else emit(std::integral_constant<int, AA_ACT_NONE>{}, splitc);
    };
    if (dst != nullptr) emit_act(std::true_type{});
    else emit_act(std::false_type{});
  }
}

template <int RT0, int RT1, int NW>
__global__ void __launch_bounds__(NW * 64) aa_conv_pair_x6_kernel(CxParams P) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char cx_lds[];
  const CxLayer& L0 = P.l[0];
  const CxLayer& L1 = P.l[1];
  char* s_in = cx_lds;
  char* s_mid = cx_lds + 3 * (size_t)L0.plane;
  const int tid = threadIdx.x;
  const int octs = L0.Cin >> 3;                 // channel octets per pixel (a power of two)
  const int oct_sh = 31 - __builtin_clz(octs);
  const int n_item = L0.H * L0.W * octs;
  for (int img = blockIdx.x; img < P.n_img; img += gridDim.x) {
    __syncthreads();   // the previous frame's readers are done
    CX_STAMP(0)
    const float4* xs = reinterpret_cast<const float4*>(P.x + (size_t)img * P.img_pitch);
    // 8 x 16-byte loads in flight: a 20 x 20 x 32 frame (1,600 items) is ONE trip of the loop,
    // i.e. one memory round trip per frame instead of two
    for (int it0 = tid; it0 < n_item; it0 += 4 * NT) {
      float4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int it = it0 + u * NT;
        if (it >= n_item) it = n_item - 1;
        v[u][0] = xs[2 * it];
        v[u][1] = xs[2 * it + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * NT;
        if (it >= n_item) continue;
        const int q = it >> oct_sh, j = it & (octs - 1);
        const int qy = cx_div(q, L0.m_w), qx = q - qy * L0.W;
        const float a[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w,
                            v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
        uint4 f[3];
        cx_split8(a, f);
        char* d = s_in + qy * L0.rowp + qx * L0.pitch + j * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<uint4*>(d + s * L0.plane) = f[s];
      }
    }
    CX_STAMP(1)
    __syncthreads();
    CX_STAMP(2)
    cx_layer<RT0 * 4 / NW>(L0, s_in, img, s_mid, L1);
    CX_STAMP(3)
    __syncthreads();
    CX_STAMP(4)
    cx_layer<RT1 * 4 / NW>(L1, s_mid, img, nullptr, L1);
    CX_STAMP(5)
  }
}

// ---- host ---------------------------------------------------------------------------------------
static int cx_check(int n_img, int H, int W, int Cin, const aa_conv_layer_desc* a,
                    const aa_conv_layer_desc* b, CxParams* P, size_t* lds_bytes,
                    size_t* ws_bytes) {
  if (n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || a == nullptr || b == nullptr)
    return AA_ERR_INVALID;
  const aa_conv_layer_desc* d[2] = {a, b};
  int h = H, w = W, c = Cin;
  size_t lds = 0, ws = 0;
  for (int i = 0; i < 2; ++i) {
    if (d[i]->KH <= 0 || d[i]->KW <= 0 || d[i]->stride <= 0 || d[i]->Cout <= 0) return AA_ERR_INVALID;
    if (c % 32 != 0 || d[i]->Cout % 16 != 0) return AA_ERR_RANGE;
    const int OH = (h - d[i]->KH) / d[i]->stride + 1, OW = (w - d[i]->KW) / d[i]->stride + 1;
    if (h < d[i]->KH || w < d[i]->KW || OH <= 0 || OW <= 0) return AA_ERR_INVALID;
    if (OH * OW > 16 * AA_CX_MAX_RT) return AA_ERR_RANGE;
    CxLayer& L = P->l[i];
    L.w = d[i]->w; L.bias = d[i]->bias; L.y = d[i]->y;
    L.H = h; L.W = w; L.Cin = c; L.KH = d[i]->KH; L.KW = d[i]->KW; L.stride = d[i]->stride;
    L.OH = OH; L.OW = OW; L.Cout = d[i]->Cout; L.act = d[i]->act;
    L.cgs = c / 32;
    L.ksteps = L.KH * L.KW * L.cgs;
    if (L.ksteps > AA_CX_MAX_KS) return AA_ERR_RANGE;
    L.m_ow = (65536u + OW - 1) / OW;
    L.m_w = (65536u + w - 1) / w;
    // the frame index arithmetic divides by multiply-shift (cx_div): n * d < 2^16
    if (h * w > 65535 / w || (c & (c - 1)) != 0) return AA_ERR_RANGE;
    // the pitch search is a pure function of the shape: cache the last few results
    struct Key { int c, h, w, ow, s, ohw, pitch, rowp; };
    static Key cache[8];
    static int n_cache = 0;
    bool hit = false;
    for (int k = 0; k < n_cache; ++k)
      if (cache[k].c == c && cache[k].h == h && cache[k].w == w && cache[k].ow == OW &&
          cache[k].s == L.stride && cache[k].ohw == OH * OW) {
        L.pitch = cache[k].pitch; L.rowp = cache[k].rowp; hit = true;
      }
    if (!hit) {
      cx_pick_pitch(c, h, w, OW, L.stride, OH * OW, &L.pitch, &L.rowp);
      if (n_cache < 8) cache[n_cache++] = Key{c, h, w, OW, L.stride, OH * OW, L.pitch, L.rowp};
    }
    L.plane = h * L.rowp;
    for (int ks = 0; ks < L.ksteps; ++ks) {
      const int tp = ks / L.cgs, cg = ks - tp * L.cgs, ky = tp / L.KW, kx = tp - ky * L.KW;
      L.tap[ks] = ky * L.rowp + kx * L.pitch + cg * 64;
    }
    lds += 3 * (size_t)L.plane;
    ws += (size_t)L.ksteps * (L.Cout / 16) * 3 * 64 * sizeof(uint4);
    h = OH; w = OW; c = d[i]->Cout;
  }
  if (lds > 160 * 1024) return AA_ERR_RANGE;
  *lds_bytes = lds;
  *ws_bytes = ws;
  return AA_OK;
}

extern "C" {

int64_t aa_conv_pair_x6_workspace_bytes(int32_t n_img, int32_t H, int32_t W, int32_t Cin,
                                        const aa_conv_layer_desc* first,
                                        const aa_conv_layer_desc* second) {
  CxParams P;
  size_t lds, ws;
  return cx_check(n_img, H, W, Cin, first, second, &P, &lds, &ws) == AA_OK ? (int64_t)ws : 0;
}

// phases: 1 = split the filter banks into `workspace` (depends on the weights only), 2 = the
// per-frame kernel over prepared planes, 3 = both.
int aa_conv_pair_x6_phase(const float* x, int64_t img_pitch, int32_t n_img, int32_t H, int32_t W,
                          int32_t Cin, const aa_conv_layer_desc* first,
                          const aa_conv_layer_desc* second, void* workspace,
                          int64_t workspace_bytes, int32_t phases, void* stream) {
  if (workspace == nullptr || phases < 1 || phases > 3) return AA_ERR_INVALID;
  if ((phases & 2) && x == nullptr) return AA_ERR_INVALID;
  CxParams P;
  size_t lds = 0, ws = 0;
  const int rc = cx_check(n_img, H, W, Cin, first, second, &P, &lds, &ws);
  if (rc != AA_OK) return rc;
  if (first->w == nullptr || second->w == nullptr) return AA_ERR_INVALID;
  // first->y may be NULL: the middle activation is only read by a backward pass (the policy's and
  // the target network's forwards have none), the second layer takes it from LDS
  if ((phases & 2) && second->y == nullptr) return AA_ERR_INVALID;
  if ((int64_t)ws > workspace_bytes || ((uintptr_t)workspace & 15) != 0) return AA_ERR_RANGE;
  const int64_t dense = (int64_t)H * W * Cin;
  P.x = x;
  P.img_pitch = img_pitch > 0 ? img_pitch : dense;
  if ((phases & 2) &&
      (P.img_pitch < dense || P.img_pitch % 4 != 0 || ((uintptr_t)x & 15) != 0))
    return AA_ERR_INVALID;
  P.n_img = n_img;
#ifdef AA_CX_STAMPS
  P.stamps = g_cx_stamps;
#endif
  P.l[0].wf = reinterpret_cast<uint4*>(workspace);
  P.l[1].wf = P.l[0].wf + (size_t)P.l[0].ksteps * (P.l[0].Cout / 16) * 3 * 64;
  hipStream_t st = (hipStream_t)stream;
  if (phases & 1) {
    const int items = (P.l[0].ksteps * (P.l[0].Cout / 16) + P.l[1].ksteps * (P.l[1].Cout / 16)) * 64;
    int blocks = (items + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(aa_conv_pair_x6_split_kernel, dim3(blocks), dim3(256), 0, st, P.l[0],
                       P.l[1]);
  }
  if (!(phases & 2)) return aa_launch_status();
  int grid = n_img;
  if (grid > 512) grid = 512;
  auto up = [](int ohw) { const int t = (ohw + 15) / 16; return t <= 2 ? 2 : t <= 4 ? 4 : t <= 6 ? 6 : 8; };
  const int r0 = up(P.l[0].OH * P.l[0].OW), r1 = up(P.l[1].OH * P.l[1].OW);
  // eight waves per workgroup (two per SIMD).  The one-wave-per-SIMD variant (AA_CX_WAVES=4: half
  // the L2 filter traffic) measured 33.3 vs 25.5 us in round 2 and was pruned in round 5.
  const int nw = 8;
  static size_t lds_limit[AA_MAX_DEVICES][32] = {{0}};   // dynamic LDS above 64 KiB is granted
  const int dv = aa_device_ordinal();                    // once per kernel and device
  if (dv < 0) return AA_ERR_LAUNCH;
  int rc2 = AA_ERR_INVALID;
#define AA_CX_CASE(A_, B_, W_)                                                                  \
  if (r0 == A_ && r1 == B_ && nw == W_) {                                                       \
    size_t& lim = lds_limit[dv][((A_ / 2 - 1) * 4 + (B_ / 2 - 1)) * 2 + (W_ == 8)];             \
    if (lds > 65536 && lds > lim) {                                                             \
      if (hipFuncSetAttribute((const void*)aa_conv_pair_x6_kernel<A_, B_, W_>,                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return AA_ERR_LAUNCH;                                                                   \
      lim = lds;                                                                                \
    }                                                                                           \
    hipLaunchKernelGGL((aa_conv_pair_x6_kernel<A_, B_, W_>), dim3(grid), dim3(W_ * 64), lds,    \
                       st, P);                                                                  \
    rc2 = AA_OK;                                                                                \
  }
#define AA_CX_ROW(A_)                                                               \
  AA_CX_CASE(A_, 2, 8) AA_CX_CASE(A_, 4, 8) AA_CX_CASE(A_, 6, 8) AA_CX_CASE(A_, 8, 8)
  AA_CX_ROW(2) AA_CX_ROW(4) AA_CX_ROW(6) AA_CX_ROW(8)
#undef AA_CX_ROW
#undef AA_CX_CASE
  if (rc2 != AA_OK) return rc2;
  return aa_launch_status();
}

int aa_conv_pair_x6_forward(const float* x, int64_t img_pitch, int32_t n_img, int32_t H, int32_t W,
                            int32_t Cin, const aa_conv_layer_desc* first,
                            const aa_conv_layer_desc* second, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  return aa_conv_pair_x6_phase(x, img_pitch, n_img, H, W, Cin, first, second, workspace,
                               workspace_bytes, 3, stream);
}

}  // extern "C"

