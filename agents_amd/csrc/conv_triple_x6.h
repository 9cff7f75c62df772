// conv1 -> conv2 -> conv3 of the Mnih-15 Q-network (examples/dqn/mnih15/dqn_train_eval_atari.py:
// 80-112: Lambda(x / 255), Conv2D(32, 8, 4), Conv2D(64, 4, 2), Conv2D(64, 3, 1), relu each) in ONE
// launch, one workgroup per uint8 frame (included by conv_pair_x6.hip: it reuses cx_layer and the
// filter pre-pass).
//
// Round 5.  The forward pass of a 256-frame batch was two launches -- aa_conv_u8_bf16x3_kernel
// (conv1: 11.7 us, bound by its 13 MB of output stores and 7 MB of patch loads, matrix pipe 12-22 %
// busy) and aa_conv_pair_x6_kernel (conv2 -> conv3, 21.4 us, of which ~4 us are launch ramp, frame
// load and split) -- three times per DQN iteration (collect policy, online network, target
// network).  conv1's output of a frame is 51 KB: exactly the input the pair kernel stages into its
// LDS planes.  Here the workgroup loads the 28 KB uint8 frame instead, computes conv1 on the bf16
// cores from LDS, and its epilogue writes the activations straight into conv2's three bf16 planes
// (the split the pair kernel did while staging) -- conv1's output never travels through memory
// unless the backward pass needs it (y1 != nullptr: the online network's forward), neither does a
// second launch's ramp.
//
// conv1 arithmetic = conv_u8_bf16.h's: a byte is exact in bf16, the fp32 filter is EXACTLY three
// bf16 pieces (the pre-pass split of conv_pair_x6.hip, same fragment order), byte x piece products
// are exact in fp32, one accumulator per piece, summed small to large, and Lambda(x / 255) is applied
// once to the sum by a correctly rounded quotient (Markstein).  The MFMA shape differs
// (16x16x32 here, 32x32x16 there), i.e. only the order of the fp32 accumulations.
//
// Mapping: k-step = 32 consecutive bytes of one patch row (KW * Cin = 32 for the Atari layer: one
// kernel row); the frame is converted to bf16 (exact) while it is staged, so lane (pixel l & 15,
// octet l >> 4) reads its fragment with one ds_read_b128.  Wave w owns column tile w % nct and every
// (NW / nct)-th 16-pixel row tile; the wave's filter fragments (3 pieces x 8 k-steps) are loaded
// once into registers and reused by all of its row tiles.  The frame is staged from the start of
// conv3's input planes on (56 KB as bf16: past their end, inside the 160 KB), which are not written
// before conv2's epilogue.
#pragma once

#define AA_CT_MAX_KS 8        /* k-steps of the first layer (their filter fragments live in registers) */
#define AA_CT_MAX_RT 7        /* 16-pixel row tiles per wave of the first layer */

struct CtFirst {
  const unsigned char* x;   // [n_img] uint8 frames, H * W * Cin bytes each, image pitch img_pitch bytes
  int64_t img_pitch;
  const float* w;           // [KH][KW][Cin][Cout] fp32 (the pre-pass reads it)
  const float* bias;        // nullable
  float* y;                 // nullable: [n_img][OH*OW][Cout] (written only when the caller needs it)
  uint4* wf;                // split filters, fragment order of aa_conv_pair_x6_split_kernel
  int H, W, Cin, KH, KW, stride, OH, OW, Cout, act;
  int rowb;                 // bytes per frame row = W * Cin
  int frame_bytes;          // H * rowb
  int ksteps;               // KH * KW * Cin / 32
  unsigned m_ow;            // ceil(2^16 / OW)
  float a_div, a_rcp;
  int tap[AA_CT_MAX_KS];    // byte offset of every k-step inside a patch
};

struct CtParams {
  CtFirst f;
  int n_img;
  CxLayer l[2];
};

// four bytes -> four bf16 (two packed dwords); exact (conv_u8_bf16.h: aa_u8x4_to_bf16)
__device__ static inline void ct_u8x4_to_bf16(unsigned d, unsigned& lo, unsigned& hi) {
  const unsigned f0 = __float_as_uint((float)(d & 255u));
  const unsigned f1 = __float_as_uint((float)((d >> 8) & 255u));
  const unsigned f2 = __float_as_uint((float)((d >> 16) & 255u));
  const unsigned f3 = __float_as_uint((float)(d >> 24));
  lo = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
  hi = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
}

// The first layer of one frame: `frame` = the uint8 frame in LDS, results split into the next
// layer's planes `dst` (and to global F.y when it is not null).
// Every filter fragment of this wave's column tile of the first layer (3 pieces x 8 k-steps, 96
// registers): requested in front of the frame load (their L2 round trip rides under it), used by
// every row tile of the frame.
template <int NW>
__device__ static inline void ct_load_b(const CtFirst& F, CxFrag (&b)[AA_CT_MAX_KS][3],
                                        float& bias_raw) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nct = F.Cout >> 4;
  const int ct = wave % nct;
  const uint4* wp = F.wf + (size_t)ct * 3 * 64 + lane;
  const size_t wstep = (size_t)nct * 3 * 64;
  const int last = F.ksteps - 1;
#pragma unroll
  for (int ks = 0; ks < AA_CT_MAX_KS; ++ks)
#pragma unroll
    for (int s = 0; s < 3; ++s) b[ks][s].q = wp[(size_t)(ks < last ? ks : last) * wstep + s * 64];
  bias_raw = (F.bias != nullptr ? F.bias : F.w)[ct * 16 + (lane & 15)];
}

template <int NW>
__device__ static inline void ct_first(const CtFirst& F, const char* __restrict__ frame, int img,
                                       char* __restrict__ dst, const CxLayer& Ln,
                                       CxFrag (&b)[AA_CT_MAX_KS][3], float bias_raw) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int OHW = F.OH * F.OW;
  const int nct = F.Cout >> 4;
  const int ngrp = NW / nct;                    // row-tile groups (host: NW % nct == 0)
  const int ct = wave % nct, grp = wave / nct;
  const int n_rt = (OHW + 15) >> 4;
  const int co = ct * 16 + lr;
  const float bv = F.bias != nullptr ? bias_raw : 0.f;
  float* yimg = F.y != nullptr ? F.y + (size_t)img * OHW * F.Cout + co : nullptr;
  // Two row tiles per trip of a ROLLED loop (six independent accumulator chains; the fully
  // unrolled form -- seven tiles x eight k-steps -- was 40 KB of straight-line code that every
  // workgroup fetched once): all sixteen fragment reads of a trip are requested up front.
#pragma unroll 1
  for (int i0 = 0; i0 < AA_CT_MAX_RT; i0 += 2) {
    const int rt0 = grp + i0 * ngrp;
    if (rt0 >= n_rt) break;                    // uniform per wave
    int pb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int p = (rt0 + u * ngrp) * 16 + lr;
      if (p >= OHW) p = OHW - 1;               // (also a second tile past the frame: discarded)
      const int oy = cx_div(p, F.m_ow), ox = p - oy * F.OW;
      pb[u] = oy * F.stride * F.rowb + ox * F.stride * F.Cin + lg * 8;
    }
    CxFrag raw[AA_CT_MAX_KS][2];
#pragma unroll
    for (int ks = 0; ks < AA_CT_MAX_KS; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        raw[ks][u].q = *reinterpret_cast<const uint4*>(frame + 2 * (pb[u] + F.tap[ks]));
    cx_f32x4 acc[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[u][s] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < AA_CT_MAX_KS; ++ks) {
      if (ks < F.ksteps) {                     // uniform
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            acc[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(raw[ks][u].v, b[ks][s].v,
                                                                acc[u][s], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rt = rt0 + u * ngrp;
      if (rt >= n_rt) continue;                // uniform per wave
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = rt * 16 + 4 * lg + e;
        if (p >= OHW) continue;
        // small pieces first, then Lambda(x / a_div) on the sum: the correctly rounded quotient
        // (q0 = s * RN(1/d); q = q0 + (s - d q0) * RN(1/d)), as in conv_u8_bf16.h
        const float sum = (acc[u][2][e] + acc[u][1][e]) + acc[u][0][e];
        const float q0 = sum * F.a_rcp;
        float v = __builtin_fmaf(__builtin_fmaf(-F.a_div, q0, sum), F.a_rcp, q0) + bv;
        if (F.act == AA_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (F.act == AA_ACT_TANH) v = tanhf(v);
        if (yimg != nullptr) yimg[p * F.Cout] = v;
        const unsigned h = cx_pk_bf16(v, 0.f) & 0xffffu;
        const float r1 = v - __uint_as_float(h << 16);
        const unsigned m = cx_pk_bf16(r1, 0.f) & 0xffffu;
        const float r2 = r1 - __uint_as_float(m << 16);
        const unsigned l = cx_pk_bf16(r2, 0.f) & 0xffffu;
        const int py = cx_div(p, F.m_ow), px = p - py * F.OW;   // (the pixel's place in `dst`)
        char* d = dst + py * Ln.rowp + px * Ln.pitch + co * 2;
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(d + Ln.plane) = (unsigned short)m;
        *reinterpret_cast<unsigned short*>(d + 2 * Ln.plane) = (unsigned short)l;
      }
    }
  }
}

template <int RT0, int RT1, int NW>
__global__ void __launch_bounds__(NW * 64) aa_conv_triple_x6_kernel(CtParams P) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char cx_lds[];
  const CtFirst& F = P.f;
  const CxLayer& L0 = P.l[0];
  const CxLayer& L1 = P.l[1];
  char* s_in = cx_lds;                             // conv2's input planes (conv1's split output)
  char* s_mid = cx_lds + 3 * (size_t)L0.plane;     // the frame as bf16, then conv3's input planes
  const int tid = threadIdx.x;
  const int n16 = F.frame_bytes >> 4;
  for (int img = blockIdx.x; img < P.n_img; img += gridDim.x) {
    __syncthreads();   // the previous frame's readers are done
    // (per frame, not per workgroup: held across the pair's layers the 96 registers spill; a
    // workgroup takes one frame at the benchmark's sizes anyway)
    CxFrag b1[AA_CT_MAX_KS][3];
    float bias1;
    ct_load_b<NW>(F, b1, bias1);
    const uint4* xs = reinterpret_cast<const uint4*>(F.x + (size_t)img * F.img_pitch);
    // the whole frame in flight at once: four 16-byte loads per lane and trip (28,224 B = one
    // trip).  Every byte is converted to bf16 (exact) HERE, once: the first version converted in
    // the k loop, once per use and column tile -- 12 VALU per fragment, as much issue time as the
    // MFMAs they fed (conv1 phase 8 us of 30; tools/gemm_one.py conv123.fwd).
    for (int it0 = tid; it0 < n16; it0 += 4 * NT) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * NT;
        v[u] = xs[it < n16 ? it : n16 - 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * NT;
        if (it < n16) {
          uint4 lo, hi;
          ct_u8x4_to_bf16(v[u].x, lo.x, lo.y);
          ct_u8x4_to_bf16(v[u].y, lo.z, lo.w);
          ct_u8x4_to_bf16(v[u].z, hi.x, hi.y);
          ct_u8x4_to_bf16(v[u].w, hi.z, hi.w);
          reinterpret_cast<uint4*>(s_mid)[2 * it] = lo;
          reinterpret_cast<uint4*>(s_mid)[2 * it + 1] = hi;
        }
      }
    }
    __syncthreads();
    ct_first<NW>(F, s_mid, img, s_in, L0, b1, bias1);
    __syncthreads();
    cx_layer<RT0 * 4 / NW>(L0, s_in, img, s_mid, L1);
    __syncthreads();
    cx_layer<RT1 * 4 / NW>(L1, s_mid, img, nullptr, L1);
  }
}

// ---- host ---------------------------------------------------------------------------------------
static int ct_check(int n_img, int H, int W, int Cin, float a_div, const aa_conv_layer_desc* a,
                    const aa_conv_layer_desc* b, const aa_conv_layer_desc* c, CtParams* P,
                    size_t* lds_bytes, size_t* ws1_bytes, size_t* ws_pair_bytes) {
  if (n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || a == nullptr || !(a_div > 0.f))
    return AA_ERR_INVALID;
  if (a->KH <= 0 || a->KW <= 0 || a->stride <= 0 || a->Cout <= 0) return AA_ERR_INVALID;
  if (H < a->KH || W < a->KW) return AA_ERR_INVALID;
  CtFirst& F = P->f;
  const int seg = a->KW * Cin;
  const int OH = (H - a->KH) / a->stride + 1, OW = (W - a->KW) / a->stride + 1;
  // one k-step = 32 consecutive bytes of a patch row; 8-byte fragment reads
  if (seg % 32 != 0 || (a->stride * Cin) % 8 != 0 || (W * Cin) % 16 != 0) return AA_ERR_RANGE;
  if (a->Cout % 16 != 0 || a->Cout > 128 || 8 % (a->Cout / 16) != 0) return AA_ERR_RANGE;
  const int ksteps = a->KH * (seg / 32);
  if (ksteps > AA_CT_MAX_KS) return AA_ERR_RANGE;
  const int OHW = OH * OW, n_rt = (OHW + 15) / 16, ngrp = 8 / (a->Cout / 16);
  if ((n_rt + ngrp - 1) / ngrp > AA_CT_MAX_RT) return AA_ERR_RANGE;
  if (OH * OW > 65535 / OW) return AA_ERR_RANGE;            // cx_div's range
  F.w = a->w; F.bias = a->bias; F.y = a->y;
  F.H = H; F.W = W; F.Cin = Cin; F.KH = a->KH; F.KW = a->KW; F.stride = a->stride;
  F.OH = OH; F.OW = OW; F.Cout = a->Cout; F.act = a->act;
  F.rowb = W * Cin;
  F.frame_bytes = H * F.rowb;
  F.ksteps = ksteps;
  F.m_ow = (65536u + OW - 1) / OW;
  F.a_div = a_div;
  F.a_rcp = 1.0f / a_div;
  for (int ks = 0; ks < AA_CT_MAX_KS; ++ks) {
    const int k = ks < ksteps ? ks : ksteps - 1;
    const int ky = k / (seg / 32), part = k - ky * (seg / 32);
    F.tap[ks] = ky * F.rowb + part * 32;
  }
  CxParams pair;
  size_t lds = 0, wsp = 0;
  const int rc = cx_check(n_img, OH, OW, a->Cout, b, c, &pair, &lds, &wsp);
  if (rc != AA_OK) return rc;
  P->l[0] = pair.l[0];
  P->l[1] = pair.l[1];
  P->n_img = n_img;
  // the frame (as bf16: 2 bytes per byte) is staged from where conv3's input planes will be
  // written on: dead by the time conv2's epilogue writes them
  const size_t with_frame = 3 * (size_t)P->l[0].plane + 2 * (size_t)F.frame_bytes;
  if (with_frame > lds) lds = with_frame;
  if (lds > 160 * 1024) return AA_ERR_RANGE;
  *lds_bytes = lds;
  *ws1_bytes = (size_t)ksteps * (a->Cout / 16) * 3 * 64 * sizeof(uint4);
  *ws_pair_bytes = wsp;
  return AA_OK;
}

extern "C" {

int64_t aa_conv_triple_x6_workspace_bytes(int32_t n_img, int32_t H, int32_t W, int32_t Cin,
                                          const aa_conv_layer_desc* first,
                                          const aa_conv_layer_desc* second,
                                          const aa_conv_layer_desc* third) {
  CtParams P;
  size_t lds, ws1, wsp;
  return ct_check(n_img, H, W, Cin, 255.0f, first, second, third, &P, &lds, &ws1, &wsp) == AA_OK
             ? (int64_t)(ws1 + wsp) : 0;
}

// phases: 1 = split the three filter banks into `workspace` (depends on the weights only), 2 = the
// per-frame kernel over prepared planes, 3 = both.  first->y / second->y may be NULL (outputs that
// only a backward pass would read); third->y is always written.
int aa_conv_triple_x6_phase(const uint8_t* x, int64_t img_pitch, int32_t n_img, int32_t H,
                            int32_t W, int32_t Cin, float a_div,
                            const aa_conv_layer_desc* first, const aa_conv_layer_desc* second,
                            const aa_conv_layer_desc* third, void* workspace,
                            int64_t workspace_bytes, int32_t phases, void* stream) {
  if (workspace == nullptr || phases < 1 || phases > 3) return AA_ERR_INVALID;
  if ((phases & 2) && x == nullptr) return AA_ERR_INVALID;
  CtParams P;
  size_t lds = 0, ws1 = 0, wsp = 0;
  const int rc = ct_check(n_img, H, W, Cin, a_div, first, second, third, &P, &lds, &ws1, &wsp);
  if (rc != AA_OK) return rc;
  if (first->w == nullptr || second->w == nullptr || third->w == nullptr) return AA_ERR_INVALID;
  if ((phases & 2) && third->y == nullptr) return AA_ERR_INVALID;
  if ((int64_t)(ws1 + wsp) > workspace_bytes || ((uintptr_t)workspace & 15) != 0)
    return AA_ERR_RANGE;
  const int64_t dense = (int64_t)H * W * Cin;
  P.f.x = x;
  P.f.img_pitch = img_pitch > 0 ? img_pitch : dense;
  if ((phases & 2) &&
      (P.f.img_pitch < dense || P.f.img_pitch % 16 != 0 || ((uintptr_t)x & 15) != 0))
    return AA_ERR_INVALID;
  P.f.wf = reinterpret_cast<uint4*>(workspace);
  P.l[0].wf = P.f.wf + ws1 / sizeof(uint4);
  P.l[1].wf = P.l[0].wf + (size_t)P.l[0].ksteps * (P.l[0].Cout / 16) * 3 * 64;
  hipStream_t st = (hipStream_t)stream;
  if (phases & 1) {
    // the first layer's bank goes through the pair's pre-pass kernel as a layer of its own (the
    // HWIO bank flattened over HWI is the k order of its fragments as well)
    CxLayer LF = P.l[0];
    LF.w = first->w; LF.wf = P.f.wf; LF.ksteps = P.f.ksteps; LF.Cout = P.f.Cout;
    CxLayer none = P.l[1];
    none.ksteps = 0;
    int items = LF.ksteps * (LF.Cout / 16) * 64;
    int blocks = (items + 255) / 256;
    hipLaunchKernelGGL(aa_conv_pair_x6_split_kernel, dim3(blocks), dim3(256), 0, st, LF, none);
    items = (P.l[0].ksteps * (P.l[0].Cout / 16) + P.l[1].ksteps * (P.l[1].Cout / 16)) * 64;
    blocks = (items + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(aa_conv_pair_x6_split_kernel, dim3(blocks), dim3(256), 0, st, P.l[0],
                       P.l[1]);
  }
  if (!(phases & 2)) return aa_launch_status();
  int grid = n_img;
  if (grid > 1024) grid = 1024;
  auto up = [](int ohw) { const int t = (ohw + 15) / 16; return t <= 2 ? 2 : t <= 4 ? 4 : t <= 6 ? 6 : 8; };
  const int r0 = up(P.l[0].OH * P.l[0].OW), r1 = up(P.l[1].OH * P.l[1].OW);
  static size_t lds_limit[AA_MAX_DEVICES][16] = {{0}};
  const int dv = aa_device_ordinal();
  if (dv < 0) return AA_ERR_LAUNCH;
  int rc2 = AA_ERR_INVALID;
#define AA_CT_CASE(A_, B_)                                                                      \
  if (r0 == A_ && r1 == B_) {                                                                   \
    size_t& lim = lds_limit[dv][(A_ / 2 - 1) * 4 + (B_ / 2 - 1)];                               \
    if (lds > 65536 && lds > lim) {                                                             \
      if (hipFuncSetAttribute((const void*)aa_conv_triple_x6_kernel<A_, B_, 8>,                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return AA_ERR_LAUNCH;                                                                   \
      lim = lds;                                                                                \
    }                                                                                           \
    hipLaunchKernelGGL((aa_conv_triple_x6_kernel<A_, B_, 8>), dim3(grid), dim3(512), lds, st,   \
                       P);                                                                      \
    rc2 = AA_OK;                                                                                \
  }
  // (the shapes of the Atari stack and its near relatives; others keep the two-launch path)
  AA_CT_CASE(6, 4) AA_CT_CASE(6, 2) AA_CT_CASE(4, 2) AA_CT_CASE(4, 4) AA_CT_CASE(8, 4) AA_CT_CASE(2, 2)
  AA_CT_CASE(8, 6) AA_CT_CASE(6, 6) AA_CT_CASE(8, 8)
#undef AA_CT_CASE
  if (rc2 != AA_OK) return AA_ERR_RANGE;
  return aa_launch_status();
}

}  // extern "C"
