// Two consecutive VALID Conv2D layers (keras Conv2D + activation twice: the conv2 -> conv3 pair of
// the Mnih-15 Q-network, examples/dqn/mnih15/dqn_train_eval_atari.py:80-112) forward in ONE
// launch, one workgroup per frame, both layers' activations written for the backward pass.
//
// Why: as separate implicit-GEMM launches the pair costs 23 + 19 us at batch 256 -- each launch pays
// its ramp (workgroup start spread ~4 us), re-reads every input pixel once per overlapping patch
// through L1 (conv2: x4) and round-trips the middle activation through HBM.  Here a frame (20x20x32
// fp32 = 51 KB) is loaded ONCE into LDS, the first layer's output stays in LDS for the second, and
// nothing is staged twice.
//
// Machine mapping: v_mfma_f32_16x16x4_f32 (exact fp32, same arithmetic class as gemm.hip).  Wave w
// owns the 16-filter column tiles {w & 3, (w & 3) + 4, ...} and one half (w >> 2) of the frame's
// 16-pixel row tiles (<= 8 in all):
//   A fragment (lane: pixel l&15, k = l>>4): ds_read_b32 of the LDS frame at the pixel's patch
//     origin + the k-step's patch offset.  The LDS pixel pitch is padded so that
//     stride * pitch == 2 (mod 32): the 16 pixels x 2 k of each half-wave hit 32 distinct banks.
//   B fragment (lane: filter l&15, k = l>>4): straight from global memory (the filter banks are
//     128-147 KB, L2 resident, shared by all workgroups) through a register ring.
// No weight staging and no barrier inside a layer; one barrier between the layers.
#include "common.h"
#include "agents_amd.h"

#include <type_traits>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define AA_CP_MAX_RT 8        /* 16-pixel row tiles per frame and layer */
#define AA_CP_WRING 8         /* k-steps per group = filter loads in flight per wave (32 channels) */
#define AA_CP_THREADS 512      /* 8 waves: two per SIMD, so one wave's LDS latency hides under the
                                * other's MFMAs */

struct CpLayer {
  const float* w;      // [KH][KW][Cin][Cout]
  const float* bias;   // nullable
  float* y;            // [n_img][OH*OW][Cout]
  int H, W, Cin, KH, KW, stride, OH, OW, Cout, act;
  int pitch;           // LDS floats per INPUT pixel of this layer
};

struct CpParams {
  const float* x;      // [n_img][H*W*Cin], image pitch img_pitch floats
  int64_t img_pitch;
  int n_img;
  CpLayer l[2];
};

__device__ static inline float cp_act(float v, int act) {
  if (act == AA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == AA_ACT_TANH) return tanhf(v);
  return v;
}

// One layer for one frame: src = LDS frame [H*W][pitch]; results to global y (+ LDS dst for the
// next layer when dst != nullptr, pixel pitch dpitch).
// RT = row tiles computed per wave (half of the frame's; compile time: no branches between the fragment loads and the MFMAs);
// tiles past the frame's last pixel recompute the last pixel and are not stored.
template <int RT>
__device__ static inline void cp_layer(const CpLayer& L, const float* __restrict__ src, int img,
                                       float* __restrict__ dst, int dpitch) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, lp = lane & 15, lk = lane >> 4;
  const int OHW = L.OH * L.OW;
  const int nct = L.Cout >> 4;
  // waves 0-3 take the first RT row tiles of the frame, waves 4-7 the second RT; wave & 3 picks
  // the column tiles
  const int rt0 = (wave >> 2) * RT;
  // patch origins of this lane's pixel in every row tile (+ its k lane)
  int pb[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int p = (rt0 + rt) * 16 + lp;
    if (p >= OHW) p = OHW - 1;
    const int oy = p / L.OW, ox = p - oy * L.OW;
    pb[rt] = ((oy * L.stride) * L.W + ox * L.stride) * L.pitch + lk;
  }
  const int n_tap = L.KH * L.KW;
  const int64_t wstep = (int64_t)4 * L.Cout;          // floats between k-steps of the filter bank
  for (int ct = wave & 3; ct < nct; ct += 4) {
    f32x4_t acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // K is walked tap by tap and, inside a tap, in groups of AA_CP_WRING k-steps (= 32 channels):
    // inside a group every LDS address is (group base + compile-time offset), so a k-step is RT
    // ds_read_b32, one filter load and RT MFMAs -- no address arithmetic.
    const float* wp = L.w + (size_t)lk * L.Cout + ct * 16 + lp;   // this lane's filter element
    float wr[AA_CP_WRING];
#pragma unroll
    for (int i = 0; i < AA_CP_WRING; ++i) wr[i] = wp[i * wstep];
    const int groups_per_tap = L.Cin / (4 * AA_CP_WRING);
    const int n_group = n_tap * groups_per_tap;
    int ky = 0, kx = 0, cg = 0;
    auto group = [&](auto more_c) {
      constexpr bool MORE = decltype(more_c)::value;   // refill the filter ring for the next group
      const int goff = (ky * L.W + kx) * L.pitch + cg * (4 * AA_CP_WRING);
      if (++cg == groups_per_tap) { cg = 0; if (++kx == L.KW) { kx = 0; ++ky; } }
      const float* sg[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) sg[rt] = src + pb[rt] + goff;
      wp += AA_CP_WRING * wstep;
      float a_cur[RT], a_nxt[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) a_cur[rt] = sg[rt][0];
#pragma unroll
      for (int i = 0; i < AA_CP_WRING; ++i) {
        const float b = wr[i];
        if (MORE) wr[i] = wp[i * wstep];
        if (i + 1 < AA_CP_WRING) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) a_nxt[rt] = sg[rt][4 * (i + 1)];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[rt], b, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a_cur[rt] = a_nxt[rt];
      }
    };
    for (int g = 0; g + 1 < n_group; ++g) group(std::true_type{});
    group(std::false_type{});
    const int co = ct * 16 + lp;
    const float bv = L.bias != nullptr ? L.bias[co] : 0.f;
    float* yimg = L.y + (size_t)img * OHW * L.Cout;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = (rt0 + rt) * 16 + 4 * lk + e;
        if (p >= OHW) continue;
        const float v = cp_act(acc[rt][e] + bv, L.act);
        yimg[(size_t)p * L.Cout + co] = v;
        if (dst != nullptr) dst[p * dpitch + co] = v;
      }
    }
  }
}

template <int RT0, int RT1>
__global__ void __launch_bounds__(AA_CP_THREADS) aa_conv_pair_kernel(CpParams P) {
  extern __shared__ __attribute__((aligned(16))) float cp_lds[];
  const CpLayer& L0 = P.l[0];
  const CpLayer& L1 = P.l[1];
  float* s_in = cp_lds;                                   // [H0*W0][pitch0]
  float* s_mid = cp_lds + (size_t)L0.H * L0.W * L0.pitch; // [OH0*OW0][pitch1]
  const int tid = threadIdx.x;
  const int n4 = (L0.H * L0.W * L0.Cin) >> 2;
  for (int img = blockIdx.x; img < P.n_img; img += gridDim.x) {
    __syncthreads();   // the previous frame's readers are done
    const float4* xs = reinterpret_cast<const float4*>(P.x + (size_t)img * P.img_pitch);
    int i0 = tid;
    for (; i0 + 3 * AA_CP_THREADS < n4; i0 += 4 * AA_CP_THREADS) {   // 4 loads in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = xs[i0 + u * AA_CP_THREADS];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = (i0 + u * AA_CP_THREADS) * 4;
        const int pix = e / L0.Cin, c = e - pix * L0.Cin;
        float* d = s_in + pix * L0.pitch + c;
        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
      }
    }
    for (; i0 < n4; i0 += AA_CP_THREADS) {
      const float4 v = xs[i0];
      const int e = i0 * 4;
      const int pix = e / L0.Cin, c = e - pix * L0.Cin;
      float* d = s_in + pix * L0.pitch + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    cp_layer<RT0 / 2>(L0, s_in, img, s_mid, L1.pitch);
    __syncthreads();
    cp_layer<RT1 / 2>(L1, s_mid, img, nullptr, 0);
  }
}

// LDS pixel pitch (floats) for an input with C channels read at `stride`: the smallest P >= C with
// stride * P == 2 (mod 32).  A wave's ds_read_b32 is served as two halves of 32 lanes = 16 pixels
// x 2 k; their banks (stride P p + k) mod 32 are then all distinct.  Else C + 1.
static int cp_pitch(int C, int stride) {
  for (int P = C; P < C + 66; ++P)
    if ((stride * P) % 32 == 2) return P;
  return C + 1;
}

static int cp_check(int n_img, int H, int W, int Cin, const aa_conv_layer_desc* a,
                    const aa_conv_layer_desc* b, CpParams* P, size_t* lds_bytes) {
  if (n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || a == nullptr || b == nullptr)
    return AA_ERR_INVALID;
  const aa_conv_layer_desc* d[2] = {a, b};
  int h = H, w = W, c = Cin;
  size_t lds = 0;
  for (int i = 0; i < 2; ++i) {
    if (d[i]->KH <= 0 || d[i]->KW <= 0 || d[i]->stride <= 0 || d[i]->Cout <= 0) return AA_ERR_INVALID;
    if (c % (4 * AA_CP_WRING) != 0 || d[i]->Cout % 16 != 0) return AA_ERR_RANGE;
    const int OH = (h - d[i]->KH) / d[i]->stride + 1, OW = (w - d[i]->KW) / d[i]->stride + 1;
    if (h < d[i]->KH || w < d[i]->KW || OH <= 0 || OW <= 0) return AA_ERR_INVALID;
    if (OH * OW > 16 * AA_CP_MAX_RT) return AA_ERR_RANGE;
    CpLayer& L = P->l[i];
    L.w = d[i]->w; L.bias = d[i]->bias; L.y = d[i]->y;
    L.H = h; L.W = w; L.Cin = c; L.KH = d[i]->KH; L.KW = d[i]->KW; L.stride = d[i]->stride;
    L.OH = OH; L.OW = OW; L.Cout = d[i]->Cout; L.act = d[i]->act;
    L.pitch = cp_pitch(c, d[i]->stride);
    lds += (size_t)h * w * L.pitch * sizeof(float);
    h = OH; w = OW; c = d[i]->Cout;
  }
  if (lds > 150 * 1024) return AA_ERR_RANGE;
  *lds_bytes = lds;
  return AA_OK;
}

extern "C" {

int aa_conv_pair_supported(int32_t n_img, int32_t H, int32_t W, int32_t Cin,
                           const aa_conv_layer_desc* first, const aa_conv_layer_desc* second) {
  CpParams P;
  size_t lds;
  return cp_check(n_img, H, W, Cin, first, second, &P, &lds) == AA_OK ? 1 : 0;
}

int aa_conv_pair_forward(const float* x, int64_t img_pitch, int32_t n_img, int32_t H, int32_t W,
                         int32_t Cin, const aa_conv_layer_desc* first,
                         const aa_conv_layer_desc* second, void* stream) {
  if (x == nullptr) return AA_ERR_INVALID;
  CpParams P;
  size_t lds = 0;
  const int rc = cp_check(n_img, H, W, Cin, first, second, &P, &lds);
  if (rc != AA_OK) return rc;
  if (first->w == nullptr || second->w == nullptr || first->y == nullptr || second->y == nullptr)
    return AA_ERR_INVALID;
  const int64_t dense = (int64_t)H * W * Cin;
  P.x = x;
  P.img_pitch = img_pitch > 0 ? img_pitch : dense;
  if (P.img_pitch < dense || P.img_pitch % 4 != 0 || ((uintptr_t)x & 15) != 0) return AA_ERR_INVALID;
  P.n_img = n_img;
  int grid = n_img;
  if (grid > 512) grid = 512;
  // row tiles per layer, rounded up to the instantiated counts {2, 4, 6, 8}
  auto up = [](int ohw) { const int t = (ohw + 15) / 16; return t <= 2 ? 2 : t <= 4 ? 4 : t <= 6 ? 6 : 8; };
  const int r0 = up(P.l[0].OH * P.l[0].OW), r1 = up(P.l[1].OH * P.l[1].OW);
  static size_t lds_limit[AA_MAX_DEVICES][16] = {{0}};   // dynamic LDS above 64 KiB is granted
  const int dv = aa_device_ordinal();                    // once per kernel and device
  if (dv < 0) return AA_ERR_LAUNCH;
  int rc2 = AA_ERR_INVALID;
#define AA_CP_CASE(A_, B_)                                                                      \
  if (r0 == A_ && r1 == B_) {                                                                   \
    size_t& lim = lds_limit[dv][(A_ / 2 - 1) * 4 + (B_ / 2 - 1)];                               \
    if (lds > 65536 && lds > lim) {                                                             \
      if (hipFuncSetAttribute((const void*)aa_conv_pair_kernel<A_, B_>,                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return AA_ERR_LAUNCH;                                                                   \
      lim = lds;                                                                                \
    }                                                                                           \
    hipLaunchKernelGGL((aa_conv_pair_kernel<A_, B_>), dim3(grid), dim3(AA_CP_THREADS), lds,      \
                       (hipStream_t)stream, P);                                                 \
    rc2 = AA_OK;                                                                                \
  }
#define AA_CP_ROW(A_) AA_CP_CASE(A_, 2) AA_CP_CASE(A_, 4) AA_CP_CASE(A_, 6) AA_CP_CASE(A_, 8)
  AA_CP_ROW(2) AA_CP_ROW(4) AA_CP_ROW(6) AA_CP_ROW(8)
#undef AA_CP_ROW
#undef AA_CP_CASE
  if (rc2 != AA_OK) return rc2;
  return aa_launch_status();
}

}  // extern "C"
