// Input gradient of a VALID Conv2D (tf.GradientTape through keras Conv2D,
// agents/dqn/dqn_agent.py:412-426) in gather form, one workgroup per frame:
//
//   dX[b,iy,ix,ci] = act'(x[b,iy,ix,ci]) * sum_{ky,kx,co} dZ[b,(iy-ky)/s,(ix-kx)/s,co] * W[ky,kx,ci,co]
//
// (terms whose (iy-ky, ix-kx) is not a multiple of the stride s or falls outside dZ vanish).
// The GEMM + col2im formulation it replaces writes and re-reads a [pixels x KH KW Cin] column
// gradient (conv2 at batch 256: 42 MB out, 42 MB back in, 51 us); here the dZ frame (7x7x64 or
// 9x9x64) sits zero-padded in LDS and every input pixel GATHERS its taps.
//
// Sub-pixel classes: input pixels with the same (iy mod s, ix mod s) = (py, px) use the same taps
// ky = py + s ty, kx = px + s tx, and for them the sum is a dense stride-1 correlation over dZ with
// a ceil((KH-py)/s) x ceil((KW-px)/s) kernel.  Each class is an implicit GEMM: rows = the class's
// pixels (16 per tile), columns = input channels (16 per tile), K = taps x Cout.
//
// Machine mapping (as conv_pair.hip): v_mfma_f32_16x16x4_f32.  The reduction index inside a tap is
// the output channel co, contiguous in BOTH operands (dZ[pixel][co] in LDS, W[ky][kx][ci][co] in
// global memory), so K is permuted to let every lane fetch 4 consecutive co at once: MFMA k-slot
// kk of k-step i' of a 16-channel subgroup j is co = 16 j + 4 kk + i'.  Per 4 k-steps a lane does
// ONE ds_read_b128 per row tile (A) and ONE 16-byte global load (B) -- with 4-byte fragment loads
// the filter fetch (16 cache lines per instruction, lanes = input channels 256 B apart) bound the
// kernel at 3x the MFMA time.  A wave task = (class, 16-channel tile, half of the class's row
// tiles); 8 waves; filters double buffered.
#include "common.h"
#include "agents_amd.h"

#include <type_traits>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define AA_DXF_THREADS 512
#define AA_DXF_GROUP 8        /* k-steps per group = 32 output channels */
#define AA_DXF_MAX_RT 4       /* row tiles per wave task (two tasks cover a class) */

struct DxfP {
  const float* dz;     // [n_img][OH*OW][Cout]
  const float* w;      // [KH][KW][Cin][Cout]
  const float* mask;   // nullable: the layer's forward input [n_img][H][W][Cin]
  float* dx;           // [n_img][H][W][Cin]
  int n_img, H, W, Cin, KH, KW, stride, OH, OW, Cout, mask_kind;
  int TY, TX;          // max taps per class and axis = zero padding of the LDS frame per side
  int Hp, Wp, pitch;   // padded frame [Hp][Wp][pitch]
};

__device__ static inline float dxf_actgrad(float y, int kind) {
  if (kind == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (kind == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

template <int RT>
__global__ void __launch_bounds__(AA_DXF_THREADS) aa_conv_dx_frame_kernel(DxfP P) {
  extern __shared__ __attribute__((aligned(16))) float dxf_lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, lp = lane & 15, lk = lane >> 4;
  const int s = P.stride;
  const int nct = P.Cin >> 4;
  const int n_task = s * s * nct * 2;
  const int gpt = P.Cout / (4 * AA_DXF_GROUP);      // 32-channel groups per tap
  const int frame_floats = P.Hp * P.Wp * P.pitch;
  // the border stays zero for the whole launch: only the interior is rewritten per frame
  for (int i = tid; i < frame_floats; i += AA_DXF_THREADS) dxf_lds[i] = 0.f;

  for (int img = blockIdx.x; img < P.n_img; img += gridDim.x) {
    __syncthreads();   // zero fill done / the previous frame's readers are done
    {
      const float4* src = reinterpret_cast<const float4*>(P.dz + (size_t)img * P.OH * P.OW * P.Cout);
      const int n4 = (P.OH * P.OW * P.Cout) >> 2;
      for (int i = tid; i < n4; i += AA_DXF_THREADS) {
        const float4 v = src[i];
        const int e = i * 4;
        const int pix = e / P.Cout, c = e - pix * P.Cout;
        const int oy = pix / P.OW, ox = pix - oy * P.OW;
        float* d = dxf_lds + ((oy + P.TY) * P.Wp + ox + P.TX) * P.pitch + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __syncthreads();

    for (int task = wave; task < n_task; task += AA_DXF_THREADS / 64) {
      const int half = task & 1;
      const int ct = (task >> 1) % nct;
      const int cls = (task >> 1) / nct;
      const int py = cls / s, px = cls - py * s;
      if (py >= P.H || px >= P.W) continue;
      const int ny = (P.H - py + s - 1) / s, nx = (P.W - px + s - 1) / s;
      const int nq = ny * nx;                         // pixels of this class
      const int tyc = (P.KH - py + s - 1) / s, txc = (P.KW - px + s - 1) / s;   // its taps
      const int rt0 = half * RT;
      if (rt0 * 16 >= nq) continue;
      // LDS origins of this lane's pixel in every row tile (+ its k lane)
      int pb[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        int q = (rt0 + rt) * 16 + lp;
        if (q >= nq) q = nq - 1;
        const int yq = q / nx, xq = q - yq * nx;
        pb[rt] = ((yq + P.TY) * P.Wp + xq + P.TX) * P.pitch + 4 * lk;
      }
      f32x4_t acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const float* wlane = P.w + (size_t)(ct * 16 + lp) * P.Cout + 4 * lk;   // + tap, group offsets
      const int n_group = tyc * txc * gpt;
      auto wgroup = [&](int g, float (&dst)[AA_DXF_GROUP]) {   // the 8 filter values of group g
        const int tap = g / gpt, cg = g - tap * gpt;
        const int ty = tap / txc, tx = tap - ty * txc;
        const float* wp = wlane + (size_t)((py + s * ty) * P.KW + px + s * tx) * P.Cin * P.Cout +
                          cg * (4 * AA_DXF_GROUP);
#pragma unroll
        for (int j = 0; j < AA_DXF_GROUP / 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(wp + 16 * j);
          dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
        }
      };
      float wc[AA_DXF_GROUP], wn[AA_DXF_GROUP];
      wgroup(0, wc);
      for (int g = 0; g < n_group; ++g) {
        const int tap = g / gpt, cg = g - tap * gpt;
        const int ty = tap / txc, tx = tap - ty * txc;
        const int goff = -(ty * P.Wp + tx) * P.pitch + cg * (4 * AA_DXF_GROUP);
        wgroup(g + 1 < n_group ? g + 1 : g, wn);    // (the last reload repeats group g: unused)
        const float* sg[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) sg[rt] = dxf_lds + pb[rt] + goff;
        float4 a4[AA_DXF_GROUP / 4][RT];
#pragma unroll
        for (int j = 0; j < AA_DXF_GROUP / 4; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            a4[j][rt] = *reinterpret_cast<const float4*>(sg[rt] + 16 * j);
        // k-step outermost: consecutive MFMAs go to different accumulators (no back-to-back
        // dependent issue)
#pragma unroll
        for (int j = 0; j < AA_DXF_GROUP / 4; ++j) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j][rt].x, wc[4 * j], acc[rt], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j][rt].y, wc[4 * j + 1], acc[rt], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j][rt].z, wc[4 * j + 2], acc[rt], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j][rt].w, wc[4 * j + 3], acc[rt], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < AA_DXF_GROUP; ++i) wc[i] = wn[i];
      }
      const int ci = ct * 16 + lp;
      const size_t img_off = (size_t)img * P.H * P.W * P.Cin;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int q = (rt0 + rt) * 16 + 4 * lk + e;
          if (q >= nq) continue;
          const int yq = q / nx, xq = q - yq * nx;
          const size_t o = img_off + ((size_t)(s * yq + py) * P.W + s * xq + px) * P.Cin + ci;
          float v = acc[rt][e];
          if (P.mask_kind != 0) v *= dxf_actgrad(P.mask[o], P.mask_kind);
          P.dx[o] = v;
        }
      }
    }
  }
}

static int dxf_pitch(int C) {   // 16-byte aligned pixels whose b128 fragments spread over the banks
  for (int P = C; P < C + 36; ++P)
    if (P % 32 == 4) return P;
  return C + 4;
}

static int dxf_check(const aa_conv_dx_desc* d, DxfP* P, size_t* lds, int* rt) {
  if (d == nullptr || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->KH <= 0 ||
      d->KW <= 0 || d->stride <= 0 || d->Cout <= 0)
    return AA_ERR_INVALID;
  if (d->H < d->KH || d->W < d->KW) return AA_ERR_INVALID;
  if (d->Cin % 16 != 0 || d->Cout % (4 * AA_DXF_GROUP) != 0) return AA_ERR_RANGE;
  const int s = d->stride;
  P->n_img = d->n_img; P->H = d->H; P->W = d->W; P->Cin = d->Cin; P->KH = d->KH; P->KW = d->KW;
  P->stride = s; P->Cout = d->Cout;
  P->OH = (d->H - d->KH) / s + 1; P->OW = (d->W - d->KW) / s + 1;
  P->TY = (d->KH + s - 1) / s; P->TX = (d->KW + s - 1) / s;
  // a class pixel row reaches dZ rows [y' - (TY-1), y'] with y' <= (H-1)/s <= OH - 1 + TY
  P->Hp = P->OH + 2 * P->TY; P->Wp = P->OW + 2 * P->TX;
  P->pitch = dxf_pitch(d->Cout);
  const int nq = ((d->H + s - 1) / s) * ((d->W + s - 1) / s);     // largest class
  const int tiles = (nq + 15) / 16;
  const int r = (tiles + 1) / 2;
  if (r > AA_DXF_MAX_RT) return AA_ERR_RANGE;
  *rt = r;
  *lds = (size_t)P->Hp * P->Wp * P->pitch * sizeof(float);
  if (*lds > 150 * 1024) return AA_ERR_RANGE;
  return AA_OK;
}

extern "C" {

int aa_conv_dx_frame_supported(const aa_conv_dx_desc* d) {
  DxfP P;
  size_t lds;
  int rt;
  return dxf_check(d, &P, &lds, &rt) == AA_OK ? 1 : 0;
}

int aa_conv_dx_frame(const aa_conv_dx_desc* d, void* stream) {
  DxfP P;
  size_t lds = 0;
  int rt = 0;
  const int rc = dxf_check(d, &P, &lds, &rt);
  if (rc != AA_OK) return rc;
  if (d->dz == nullptr || d->w == nullptr || d->dx == nullptr) return AA_ERR_INVALID;
  if (((uintptr_t)d->dz & 15) != 0 || ((uintptr_t)d->w & 15) != 0) return AA_ERR_INVALID;
  P.dz = d->dz; P.w = d->w; P.dx = d->dx;
  P.mask = d->mask_src;
  P.mask_kind = d->mask_src != nullptr ? d->mask_kind : 0;
  int grid = d->n_img > 512 ? 512 : d->n_img;
  static size_t lds_limit[AA_MAX_DEVICES][AA_DXF_MAX_RT + 1] = {{0}};   // > 64 KiB of dynamic
  const int dv = aa_device_ordinal();                   // LDS: granted per kernel and device
  if (dv < 0) return AA_ERR_LAUNCH;
  int done = 0;
#define AA_DXF_CASE(R_)                                                                         \
  if (rt == R_) {                                                                               \
    if (lds > 65536 && lds > lds_limit[dv][R_]) {                                                   \
      if (hipFuncSetAttribute((const void*)aa_conv_dx_frame_kernel<R_>,                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return AA_ERR_LAUNCH;                                                                   \
      lds_limit[dv][R_] = lds;                                                                    \
    }                                                                                           \
    hipLaunchKernelGGL((aa_conv_dx_frame_kernel<R_>), dim3(grid), dim3(AA_DXF_THREADS), lds,     \
                       (hipStream_t)stream, P);                                                 \
    done = 1;                                                                                   \
  }
  AA_DXF_CASE(1) AA_DXF_CASE(2) AA_DXF_CASE(3) AA_DXF_CASE(4)
#undef AA_DXF_CASE
  if (!done) return AA_ERR_RANGE;
  return aa_launch_status();
}

}  // extern "C"
