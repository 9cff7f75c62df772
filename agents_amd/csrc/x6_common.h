// Shared pieces of the "bf16x6" kernels (conv_pair_x6.hip, conv_dx_frame_x6.hip): fp32 contractions
// on the bf16 matrix cores at fp32 accuracy.  An fp32 value is EXACTLY hi + mid + lo, three
// round-to-nearest bf16 pieces; a product of two bf16 numbers is exact in fp32; of the nine piece
// products of x * w the three smallest (< 2^-27 |x w|) are dropped, the other six are accumulated
// in fp32 by v_mfma_f32_16x16x32_bf16 (oracle/numerics.py restates the identities on the CPU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float cx_f32x4 __attribute__((ext_vector_type(4)));
typedef short cx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cx_f32x2 __attribute__((ext_vector_type(2)));

union CxFrag {
  uint4 q;
  cx_bf16x8 v;
};


__device__ static inline unsigned cx_pk_bf16(float lo, float hi) {   // {bf16_rn(hi), bf16_rn(lo)}
  cx_f32x2 f = {lo, hi};
  cx_bf16x2 b = __builtin_convertvector(f, cx_bf16x2);
  return __builtin_bit_cast(unsigned, b);
}

// 8 floats -> three packed bf16 fragments (hi, mid, lo); residuals are exact
__device__ static inline void cx_split8(const float (&a)[8], uint4 (&f)[3]) {
  unsigned pc[3][4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    float r0 = a[e], r1 = a[e + 1];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const unsigned pk = cx_pk_bf16(r0, r1);
      pc[s][e >> 1] = pk;
      if (s < 2) {
        r0 -= __uint_as_float(pk << 16);
        r1 -= __uint_as_float(pk & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) f[s] = make_uint4(pc[s][0], pc[s][1], pc[s][2], pc[s][3]);
}


// n / d for n * d < 2^16 with m = ceil(2^16 / d): exact because n * (d m - 2^16) < n d < 2^16
__device__ static inline int cx_div(int n, unsigned m) { return (int)(((unsigned)n * m) >> 16); }

// The six products of one k-step for RT row tiles: smallest first, consecutive MFMAs on different
// accumulators.
template <int RT>
__device__ static inline void cx_mma6(CxFrag (&a)[RT][3], CxFrag (&b)[3], cx_f32x4 (&big)[RT],
                                      cx_f32x4 (&small)[RT]) {
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
}

// ---- host: LDS layout search ----------------------------------------------------------------------
// LDS cycles of one row tile's A-fragment read under the ds_read_b128 model (4 groups of 16 lanes,
// bank = dword address mod 64, 4 banks per lane), averaged over the frame's row tiles; 1.0 =
// conflict free.
static double cx_read_cost(int pitch, int rowp, int OW, int stride, int OHW) {
  static const int grp[4][16] = {
      {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  const int tiles = (OHW + 15) / 16;
  double tot = 0;
  for (int rt = 0; rt < tiles; ++rt)
    for (int g = 0; g < 4; ++g) {
      int cnt[64];
      int first[64][16];
      for (int b = 0; b < 64; ++b) cnt[b] = 0;
      int worst = 1;
      for (int i = 0; i < 16; ++i) {
        const int lane = grp[g][i], r = lane & 15, k = lane >> 4;
        int p = rt * 16 + r;
        if (p >= OHW) p = OHW - 1;
        const int oy = p / OW, ox = p - oy * OW;
        const int dw = (oy * stride * rowp + ox * stride * pitch + k * 16) / 4;
        for (int d = 0; d < 4; ++d) {
          const int b = (dw + d) & 63;
          bool dup = false;
          for (int j = 0; j < cnt[b]; ++j) dup = dup || first[b][j] == dw + d;
          if (!dup) {
            first[b][cnt[b]++] = dw + d;
            if (cnt[b] > worst) worst = cnt[b];
          }
        }
      }
      tot += worst;
    }
  return tot / (4.0 * tiles);
}

static void cx_pick_pitch(int Cin, int H, int W, int OW, int stride, int OHW, int* pitch,
                          int* rowp) {
  double best = 1e9;
  int64_t best_bytes = 0;
  for (int pad = 0; pad <= 128; pad += 16)
    for (int rpad = 0; rpad <= 240; rpad += 16) {
      const int pt = Cin * 2 + pad, rp = W * pt + rpad;
      const double c = cx_read_cost(pt, rp, OW, stride, OHW);
      const int64_t bytes = (int64_t)H * rp;
      if (c < best - 1e-9 || (c < best + 1e-9 && bytes < best_bytes)) {
        best = c; best_bytes = bytes; *pitch = pt; *rowp = rp;
      }
    }
}

