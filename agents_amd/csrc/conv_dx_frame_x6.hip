// Input gradient of a VALID Conv2D in gather form (csrc/conv_dx_frame.hip: one workgroup per frame,
// sub-pixel classes, implicit GEMM per class) on the bf16 matrix cores at fp32 accuracy
// (x6_common.h: exact three-piece split, six of nine piece products, fp32 accumulation).
//
//   dX[b,iy,ix,ci] = act'(x[b,iy,ix,ci]) * sum_{ky,kx,co} dZ[b,(iy-ky)/s,(ix-kx)/s,co] * W[ky,kx,ci,co]
//
// Every operand element is split ONCE: the dZ frame while it is staged, zero padded, into three
// bf16 LDS planes; the filters by a pre-pass of the same call into MFMA fragment order in a
// caller-provided scratch (the reduction index inside a tap is the output channel co, contiguous
// in W[ky][kx][ci][co], so a B fragment -- lane (ci, octet) = 8 consecutive co -- is 32 contiguous
// bytes of the fp32 bank).  The pre-pass also writes the k-step table of every sub-pixel class
// ({LDS offset of the tap, fragment index}), which the workgroups copy to LDS: the main loop has no
// tap bookkeeping and no branches.
//
// Machine mapping: a k-step = 32 output channels of one tap = one v_mfma_f32_16x16x32_bf16 per
// piece product.  A wave task = (class, 16-input-channel tile, half of the class's 16-pixel row
// tiles); 8 waves.  A fragment = one ds_read_b128 per plane and row tile (pixel pitch / row pitch
// padded on the host against the ds_read_b128 bank model, x6_common.h); B fragments run four
// k-steps ahead in a register ring (L2 resident, 16-byte loads, 1 KiB contiguous per wave).
#include "common.h"
#include "agents_amd.h"
#include "x6_common.h"

#include <type_traits>

#define AA_DX6_THREADS 512
#define AA_DX6_MAX_RT 4        /* row tiles per wave task (two tasks cover a class) */
#define AA_DX6_MAX_KS 64       /* k-steps per class */
#define AA_DX6_MAX_CLS 16      /* stride^2 */

struct Dx6P {
  const float* dz;     // [n_img][OH*OW][Cout]
  const float* w;      // [KH][KW][Cin][Cout]
  const float* mask;   // nullable: the layer's forward input [n_img][H][W][Cin]
  float* dx;           // [n_img][H][W][Cin]
  uint4* wf;           // split filters: [tap][32-co group][ci tile][plane][lane] x 8 bf16
  int2* tab;           // [class][AA_DX6_MAX_KS]: {A byte offset of the k-step, B fragment row}
  int n_img, H, W, Cin, KH, KW, stride, OH, OW, Cout, mask_kind;
  int TY, TX;          // max taps per class and axis = zero padding of the LDS frame per side
  int Hp, Wp;          // padded frame [Hp][Wp]
  int pitch, rowp, plane;   // bytes per pixel / row / plane
  int gpt;             // 32-channel groups per tap = Cout / 32
  unsigned m_ow;       // ceil(2^16 / OW)
};

__device__ static inline float dx6_actgrad(float y, int kind) {
  if (kind == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (kind == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

// ---- pre-pass: filter fragments + class tables ----------------------------------------------------
__global__ void __launch_bounds__(256) aa_conv_dx6_prep_kernel(Dx6P P) {
  const int nct = P.Cin >> 4;
  const int n_frag = P.KH * P.KW * P.gpt * nct * 64;
  const int s = P.stride;
  const int n_tab = s * s * AA_DX6_MAX_KS;
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < n_frag + n_tab;
       it += gridDim.x * blockDim.x) {
    if (it < n_frag) {
      // item = (tap, group, ci tile, lane): lane (c = l & 15, g = l >> 4) holds
      // co = 32 group + 8 g + e, e = 0..7, of input channel 16 ct + c
      const int lane = it & 63, c = lane & 15, g = lane >> 4;
      const int t = it >> 6, ct = t % nct, tg = t / nct;        // tg = tap * gpt + group
      const int tap = tg / P.gpt, cg = tg - tap * P.gpt;
      const float4* src = reinterpret_cast<const float4*>(
          P.w + ((size_t)tap * P.Cin + ct * 16 + c) * P.Cout + cg * 32 + g * 8);
      const float4 v0 = src[0], v1 = src[1];
      const float a[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      uint4 f[3];
      cx_split8(a, f);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) P.wf[((size_t)t * 3 + pl) * 64 + lane] = f[pl];
    } else {
      const int e = it - n_frag, cls = e / AA_DX6_MAX_KS, ks = e - cls * AA_DX6_MAX_KS;
      const int py = cls / s, px = cls - py * s;
      int2 v = make_int2(0, 0);
      if (py < P.KH && px < P.KW) {
        const int tyc = (P.KH - py + s - 1) / s, txc = (P.KW - px + s - 1) / s;
        if (ks < tyc * txc * P.gpt) {
          const int tap = ks / P.gpt, cg = ks - tap * P.gpt;
          const int ty = tap / txc, tx = tap - ty * txc;
          v.x = -(ty * P.rowp + tx * P.pitch) + cg * 64;
          v.y = (((py + s * ty) * P.KW + px + s * tx) * P.gpt + cg) * nct;
        }
      }
      P.tab[e] = v;
    }
  }
}

template <int RT>
__global__ void __launch_bounds__(AA_DX6_THREADS) aa_conv_dx_frame_x6_kernel(Dx6P P) {
  extern __shared__ __attribute__((aligned(16))) char dx6_lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int s = P.stride;
  const int nct = P.Cin >> 4;
  const int n_task = s * s * nct * 2;
  char* planes = dx6_lds;
  int2* s_tab = reinterpret_cast<int2*>(dx6_lds + 3 * (size_t)P.plane);
  // zero the planes once (the border stays zero for the whole launch) and fetch the class tables
  {
    uint4* z = reinterpret_cast<uint4*>(planes);
    const int n16 = (3 * P.plane) >> 4;
    for (int i = tid; i < n16; i += AA_DX6_THREADS) z[i] = make_uint4(0, 0, 0, 0);
    const int n_tab = s * s * AA_DX6_MAX_KS;
    for (int i = tid; i < n_tab; i += AA_DX6_THREADS) s_tab[i] = P.tab[i];
  }
  const int octs = P.Cout >> 3;
  const int oct_sh = 31 - __builtin_clz(octs);
  const int n_item = P.OH * P.OW * octs;

  for (int img = blockIdx.x; img < P.n_img; img += gridDim.x) {
    __syncthreads();   // zero fill done / the previous frame's readers are done
    {
      const float4* xs = reinterpret_cast<const float4*>(P.dz + (size_t)img * P.OH * P.OW * P.Cout);
      for (int it0 = tid; it0 < n_item; it0 += 2 * AA_DX6_THREADS) {
        float4 v[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int it = it0 + u * AA_DX6_THREADS;
          if (it >= n_item) it = n_item - 1;
          v[u][0] = xs[2 * it];
          v[u][1] = xs[2 * it + 1];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = it0 + u * AA_DX6_THREADS;
          if (it >= n_item) continue;
          const int q = it >> oct_sh, j = it & (octs - 1);
          const int oy = cx_div(q, P.m_ow), ox = q - oy * P.OW;
          const float a[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w,
                              v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
          uint4 f[3];
          cx_split8(a, f);
          char* d = planes + (oy + P.TY) * P.rowp + (ox + P.TX) * P.pitch + j * 16;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(d + pl * P.plane) = f[pl];
        }
      }
    }
    __syncthreads();

    for (int task = wave; task < n_task; task += AA_DX6_THREADS / 64) {
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
      const unsigned m_nx = (65536u + nx - 1) / nx;
      int pb[RT];      // byte offset of this lane's class pixel (+ its channel octet)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        int q = (rt0 + rt) * 16 + lr;
        if (q >= nq) q = nq - 1;
        const int yq = cx_div(q, m_nx), xq = q - yq * nx;
        pb[rt] = (yq + P.TY) * P.rowp + (xq + P.TX) * P.pitch + lg * 16;
      }
      cx_f32x4 big[RT], small[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        big[rt] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
        small[rt] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // The activation-derivative mask of this task's outputs (the layer's forward input at the
      // same positions) is requested HERE, in front of the k loop: it used to be the first
      // dependent access of the task's epilogue -- one exposed memory round trip per task, two
      // tasks per wave and frame on conv2.  Unconditional loads from clamped in-range positions
      // (of the dZ tensor when the launch has no mask): no branch between the loads and the wait.
      const int ci = ct * 16 + lr;
      const int img_off = P.H * P.W * P.Cin;
      const float* mimg = P.mask != nullptr ? P.mask + (size_t)img * img_off + ci : P.dz;
      const int m_lim = P.mask != nullptr ? img_off - ci - 1 : 0;
      float mv[RT][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int q = (rt0 + rt) * 16 + 4 * lg + e;
          if (q >= nq) q = nq - 1;
          const int yq = cx_div(q, m_nx), xq = q - yq * nx;
          int o = ((s * yq + py) * P.W + s * xq + px) * P.Cin;
          o = o < m_lim ? o : m_lim;
          mv[rt][e] = mimg[o];
        }
      const int2* tab = s_tab + cls * AA_DX6_MAX_KS;
      const uint4* wp = P.wf + (size_t)ct * 3 * 64 + lane;
      auto entry = [&](int ks) {
        const int2 v = tab[ks];
        return make_int2(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y));
      };
      auto load_b = [&](CxFrag (&b)[3], int row) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[pl].q = wp[((size_t)row * 3 + pl) * 64];
      };
      auto load_a = [&](CxFrag (&a)[RT][3], int off) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            a[rt][pl].q = *reinterpret_cast<const uint4*>(planes + pl * P.plane + pb[rt] + off);
      };
      // (a class without taps -- stride > kernel extent -- has S = 0 and gets zeros)
      const int S = (tyc > 0 && txc > 0) ? tyc * txc * P.gpt : 0, last = S - 1;
      if (S > 0) {
      CxFrag a0[RT][3], a1[RT][3], b[4][3];
#pragma unroll
      for (int j = 0; j < 4; ++j) load_b(b[j], entry(j < last ? j : last).y);
      load_a(a0, entry(0).x);
      auto step = [&](auto jc, int ks) {
        constexpr int J = decltype(jc)::value;
        const int kn = ks + J + 1 < last ? ks + J + 1 : last;
        const int kb = ks + J + 4 < last ? ks + J + 4 : last;
        if (J & 1) {
          load_a(a0, entry(kn).x);
          cx_mma6<RT>(a1, b[J], big, small);
        } else {
          load_a(a1, entry(kn).x);
          cx_mma6<RT>(a0, b[J], big, small);
        }
        load_b(b[J], entry(kb).y);
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
      }

      float* ximg = P.dx + (size_t)img * img_off + ci;
      auto emit = [&](auto maskc) {
        constexpr int MK = decltype(maskc)::value;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = (rt0 + rt) * 16 + 4 * lg + e;
            if (q >= nq) continue;
            const int yq = cx_div(q, m_nx), xq = q - yq * nx;
            const int o = ((s * yq + py) * P.W + s * xq + px) * P.Cin;
            float v = big[rt][e] + small[rt][e];
            if (MK != 0) v *= dx6_actgrad(mv[rt][e], MK);
            ximg[o] = v;
          }
        }
      };
      if (P.mask_kind == AA_ACT_RELU) emit(std::integral_constant<int, AA_ACT_RELU>{});
      else if (P.mask_kind == AA_ACT_TANH) emit(std::integral_constant<int, AA_ACT_TANH>{});
      else emit(std::integral_constant<int, 0>{});
    }
  }
}

static int dx6_check(const aa_conv_dx_desc* d, Dx6P* P, size_t* lds, size_t* ws, int* rt) {
  if (d == nullptr || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->KH <= 0 ||
      d->KW <= 0 || d->stride <= 0 || d->Cout <= 0)
    return AA_ERR_INVALID;
  if (d->H < d->KH || d->W < d->KW) return AA_ERR_INVALID;
  const int s = d->stride;
  if (d->Cin % 16 != 0 || d->Cout % 32 != 0 || (d->Cout & (d->Cout - 1)) != 0) return AA_ERR_RANGE;
  if (s * s > AA_DX6_MAX_CLS) return AA_ERR_RANGE;
  P->n_img = d->n_img; P->H = d->H; P->W = d->W; P->Cin = d->Cin; P->KH = d->KH; P->KW = d->KW;
  P->stride = s; P->Cout = d->Cout;
  P->OH = (d->H - d->KH) / s + 1; P->OW = (d->W - d->KW) / s + 1;
  P->TY = (d->KH + s - 1) / s; P->TX = (d->KW + s - 1) / s;
  P->Hp = P->OH + 2 * P->TY; P->Wp = P->OW + 2 * P->TX;
  P->gpt = d->Cout / 32;
  if (P->TY * P->TX * P->gpt > AA_DX6_MAX_KS) return AA_ERR_RANGE;
  const int nx0 = (d->W + s - 1) / s, nq = ((d->H + s - 1) / s) * nx0;     // largest class
  if (nq > 255 || P->OH * P->OW > 65535 / P->OW) return AA_ERR_RANGE;      // cx_div ranges
  const int tiles = (nq + 15) / 16;
  const int r = (tiles + 1) / 2;
  if (r > AA_DX6_MAX_RT) return AA_ERR_RANGE;
  *rt = r;
  // LDS pixel / row pitch of a plane: searched against the b128 bank model for the class geometry
  {
    struct Key { int c, wp, nx, nq, pitch, rowp; };
    static Key cache[8];
    static int n_cache = 0;
    bool hit = false;
    for (int k = 0; k < n_cache; ++k)
      if (cache[k].c == d->Cout && cache[k].wp == P->Wp && cache[k].nx == nx0 && cache[k].nq == nq) {
        P->pitch = cache[k].pitch; P->rowp = cache[k].rowp; hit = true;
      }
    if (!hit) {
      cx_pick_pitch(d->Cout, P->Hp, P->Wp, nx0, 1, nq, &P->pitch, &P->rowp);
      if (n_cache < 8) cache[n_cache++] = Key{d->Cout, P->Wp, nx0, nq, P->pitch, P->rowp};
    }
  }
  P->plane = P->Hp * P->rowp;
  P->m_ow = (65536u + P->OW - 1) / P->OW;
  *lds = 3 * (size_t)P->plane + (size_t)s * s * AA_DX6_MAX_KS * sizeof(int2);
  if (*lds > 160 * 1024) return AA_ERR_RANGE;
  const size_t frag = (size_t)d->KH * d->KW * P->gpt * (d->Cin / 16) * 3 * 64 * sizeof(uint4);
  *ws = frag + (size_t)s * s * AA_DX6_MAX_KS * sizeof(int2);
  return AA_OK;
}

extern "C" {

int64_t aa_conv_dx_frame_x6_workspace_bytes(const aa_conv_dx_desc* d) {
  Dx6P P;
  size_t lds, ws;
  int rt;
  return dx6_check(d, &P, &lds, &ws, &rt) == AA_OK ? (int64_t)ws : 0;
}

// phases: 1 = filter fragments + per-class k-step tables into `workspace` (depends on the
// weights only), 2 = the per-frame kernel over a prepared workspace, 3 = both.
int aa_conv_dx_frame_x6_phase(const aa_conv_dx_desc* d, void* workspace, int64_t workspace_bytes,
                              int32_t phases, void* stream) {
  if (phases < 1 || phases > 3) return AA_ERR_INVALID;
  Dx6P P;
  size_t lds = 0, ws = 0;
  int rt = 0;
  const int rc = dx6_check(d, &P, &lds, &ws, &rt);
  if (rc != AA_OK) return rc;
  if (d->w == nullptr || workspace == nullptr) return AA_ERR_INVALID;
  if ((phases & 2) && (d->dz == nullptr || d->dx == nullptr)) return AA_ERR_INVALID;
  if (((uintptr_t)d->dz & 15) != 0 || ((uintptr_t)d->w & 15) != 0 ||
      ((uintptr_t)workspace & 15) != 0)
    return AA_ERR_INVALID;
  if ((int64_t)ws > workspace_bytes) return AA_ERR_RANGE;
  P.dz = d->dz; P.w = d->w; P.dx = d->dx;
  P.mask = d->mask_src;
  P.mask_kind = d->mask_src != nullptr ? d->mask_kind : 0;
  P.wf = reinterpret_cast<uint4*>(workspace);
  const size_t frag = (size_t)d->KH * d->KW * P.gpt * (d->Cin / 16) * 3 * 64;
  P.tab = reinterpret_cast<int2*>(P.wf + frag);
  hipStream_t st = (hipStream_t)stream;
  if (phases & 1) {
    const int items = d->KH * d->KW * P.gpt * (d->Cin / 16) * 64 +
                      d->stride * d->stride * AA_DX6_MAX_KS;
    int blocks = (items + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(aa_conv_dx6_prep_kernel, dim3(blocks), dim3(256), 0, st, P);
  }
  if (!(phases & 2)) return aa_launch_status();
  int grid = d->n_img > 512 ? 512 : d->n_img;
  static size_t lds_limit[AA_MAX_DEVICES][AA_DX6_MAX_RT + 1] = {{0}};   // > 64 KiB of dynamic
  const int dv = aa_device_ordinal();                   // LDS: granted per kernel and device
  if (dv < 0) return AA_ERR_LAUNCH;
  int done = 0;
#define AA_DX6_CASE(R_)                                                                         \
  if (rt == R_) {                                                                               \
    if (lds > 65536 && lds > lds_limit[dv][R_]) {                                                   \
      if (hipFuncSetAttribute((const void*)aa_conv_dx_frame_x6_kernel<R_>,                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return AA_ERR_LAUNCH;                                                                   \
      lds_limit[dv][R_] = lds;                                                                    \
    }                                                                                           \
    hipLaunchKernelGGL((aa_conv_dx_frame_x6_kernel<R_>), dim3(grid), dim3(AA_DX6_THREADS), lds,  \
                       st, P);                                                                  \
    done = 1;                                                                                   \
  }
  AA_DX6_CASE(1) AA_DX6_CASE(2) AA_DX6_CASE(3) AA_DX6_CASE(4)
#undef AA_DX6_CASE
  if (!done) return AA_ERR_RANGE;
  return aa_launch_status();
}

int aa_conv_dx_frame_x6(const aa_conv_dx_desc* d, void* workspace, int64_t workspace_bytes,
                        void* stream) {
  return aa_conv_dx_frame_x6_phase(d, workspace, workspace_bytes, 3, stream);
}

}  // extern "C"
