// Conv2D forward on uint8 frames with 32 filters (the Atari conv1 of the Mnih-15 Q-network,
// examples/dqn/mnih15/dqn_train_eval_atari.py:80-112) on the bf16 matrix cores, at fp32 accuracy.
//
//   y[pix][f] = act( (sum_k u8[pix][k] * w[k][f]) / a_div + bias[f] )
//
// Why this is as accurate as the fp32 MFMA path: a byte is exactly representable in bf16 (8
// significant bits); an fp32 weight is EXACTLY the sum of three round-to-nearest bf16 pieces
// (hi + mid + lo carry 3 x 8 significant bits plus the sign trick of RN residuals); every
// byte x piece product is exact in fp32 (<= 16 significant bits); so the three MFMA chains
// accumulate exact products in fp32 -- the only roundings are the accumulations, as in the fp32
// kernel -- and the Lambda(x / 255) of the reference is applied once to the sum by an IEEE
// division instead of per element at operand fetch.  v_mfma_f32_32x32x16_bf16 runs at 16x the
// fp32 MFMA rate, so three pieces cost 3/16 of the fp32 MFMA time, and the byte -> bf16
// conversion is 1.5 VALU per element (v_cvt_f32_ubyteN + v_perm) against 4 VALU + an LDS
// round trip for the exact (float)u8 / 255 of the fp32 loader.
//
// Layout: a wave owns TPW 32-pixel tiles x all 32 filters.  The A fragments come straight from
// global memory in MFMA layout: lane (r = lane & 31, h = lane >> 5) loads the 16 bytes
// [16h, 16h+16) of a 32-byte chunk of patch row ky of pixel r -- two MFMAs' worth (bytes 0-7 and
// 8-15).  K is permuted accordingly; the filter bank is split and stored in LDS in exactly that
// permuted fragment order ([piece][chunk*2 + q][h][filter] x 8 bf16 = one conflict-free
// ds_read_b128 per fragment), once per workgroup, and each fragment is reused by the TPW tiles.
// No LDS traffic for A, no barrier in the main loop.
#pragma once

typedef short bf16x8_t __attribute__((ext_vector_type(8)));

#define AA_CU8_THREADS 256
#define AA_CU8_MAX_K 320   /* 3 * K * 64 B of LDS <= 60 KiB */

__device__ static inline unsigned aa_bf16_rn_bits(float x) {   // finite x
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ static inline float aa_bf16_to_f32(unsigned b) { return __uint_as_float(b << 16); }
typedef __bf16 aa_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float aa_f32x2_t __attribute__((ext_vector_type(2)));
__device__ static inline unsigned aa_pk_bf16(float lo, float hi) {   // {bf16_rn(hi), bf16_rn(lo)}
  aa_f32x2_t f = {lo, hi};
  aa_bf16x2_t b = __builtin_convertvector(f, aa_bf16x2_t);
  return __builtin_bit_cast(unsigned, b);
}

// four bytes of `d` -> four bf16 (two packed dwords); exact
__device__ static inline void aa_u8x4_to_bf16(unsigned d, unsigned& lo, unsigned& hi) {
  // (float)((d >> 8n) & 255) selects v_cvt_f32_ubyte<n>
  const unsigned f0 = __float_as_uint((float)(d & 255u));
  const unsigned f1 = __float_as_uint((float)((d >> 8) & 255u));
  const unsigned f2 = __float_as_uint((float)((d >> 16) & 255u));
  const unsigned f3 = __float_as_uint((float)(d >> 24));
  lo = __builtin_amdgcn_perm(f1, f0, 0x07060302u);   // {f1[31:16], f0[31:16]}
  hi = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
}

union AaFrag {
  uint4 q;
  bf16x8_t v;
};

template <int TPW, int NCH>
__global__ void __launch_bounds__(AA_CU8_THREADS, 2)   // two workgroups per CU (400 on 256 CUs)
aa_conv_u8_bf16x3_kernel(GemmP p, int n_super, int nch_rt) {
  extern __shared__ __attribute__((aligned(16))) uint4 wfrag[];   // [3][J][2][32]
  const int nch = NCH > 0 ? NCH : nch_rt;     // 32-byte chunks per patch = KH * seg / 32
  const int J = nch * 2;
  const int R = p.seg >> 5;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const unsigned char* A = reinterpret_cast<const unsigned char*>(p.A);
  const int per_xcd = (n_super + 7) >> 3;
  auto chunk_off = [&](int ch) {   // R == 1 (32-byte patch rows) needs no division
    if (R == 1) return ch * p.rowpitch;
    const int ky = ch / R;
    return ky * p.rowpitch + ((ch - ky * R) << 5);
  };
  // pixel origin of wave-tile t of super-tile slot sb (-1: nothing to do there)
  auto tile_pix0 = [&](int sb) {
    const int st = (sb & 7) * per_xcd + (sb >> 3);
    if ((sb >> 3) >= per_xcd || st >= n_super) return -1;
    const int pix0 = (st * 4 + wave) * (32 * TPW);
    return pix0 < p.M ? pix0 : -1;
  };
  // The whole patch of the FIRST super-tile is requested before the filter split below: its
  // round trip to memory runs under the split (each workgroup used to start its loads behind it).
  uint4 a16[NCH > 0 ? NCH : 1][TPW];   // the whole patch of a tile in flight at once
  auto request = [&](int pix0) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      int pix = pix0 + t * 32 + r;
      if (pix >= p.M) pix = p.M - 1;
      const unsigned char* sp = A + aa_pix_base(p, pix) + h * 16;
#pragma unroll
      for (int ch = 0; ch < (NCH > 0 ? NCH : 1); ++ch)
        a16[ch][t] = *reinterpret_cast<const uint4*>(sp + chunk_off(ch));
    }
  };
  if constexpr (NCH > 0) {
    const int pix0 = tile_pix0(blockIdx.x);
    if (pix0 >= 0) request(pix0);
  }

  // ---- filter bank: fp32 [K][32] -> three bf16 pieces in fragment order --------------------
  for (int item = tid; item < (p.K >> 3) * 32; item += AA_CU8_THREADS) {
    const int c = item & 31, k0 = (item >> 5) << 3;
    const int ky = k0 / p.seg, rem = k0 - ky * p.seg;
    const int j = ((ky * R + (rem >> 5)) << 1) | ((rem >> 3) & 1);
    const int h = (rem >> 4) & 1;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = p.B[(size_t)(k0 + e) * p.ldb + c];
    unsigned pc[3][4];   // packed pairs; v_cvt_pk_bf16_f32 rounds to nearest even
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float r0 = wv[e], r1 = wv[e + 1];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned pk = aa_pk_bf16(r0, r1);
        pc[s][e >> 1] = pk;
        if (s < 2) {
          r0 -= __uint_as_float(pk << 16);           // exact residuals
          r1 -= __uint_as_float(pk & 0xffff0000u);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
      wfrag[((s * J + j) * 2 + h) * 32 + c] = make_uint4(pc[s][0], pc[s][1], pc[s][2], pc[s][3]);
  }
  __syncthreads();

  const float bv = p.bias != nullptr ? p.bias[r] : 0.f;
  // XCD-aware: the 8 XCDs take workgroups round-robin; give each a contiguous range of pixels so
  // the 4x patch overlap of a frame is served by ONE L2.
  for (int sb = blockIdx.x; sb < per_xcd * 8; sb += gridDim.x) {
    const int pix0 = tile_pix0(sb);
    if (pix0 < 0) continue;
    const unsigned char* src[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      int pix = pix0 + t * 32 + r;
      if (pix >= p.M) pix = p.M - 1;
      src[t] = A + aa_pix_base(p, pix) + h * 16;
    }
    f32x16 acc[3][TPW];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][t][e] = 0.f;

    auto mma = [&](int ch, const uint4 (&a16)[TPW]) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        AaFrag af[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const unsigned d0 = q == 0 ? a16[t].x : a16[t].z;
          const unsigned d1 = q == 0 ? a16[t].y : a16[t].w;
          aa_u8x4_to_bf16(d0, af[t].q.x, af[t].q.y);
          aa_u8x4_to_bf16(d1, af[t].q.z, af[t].q.w);
        }
        const int j = ch * 2 + q;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          AaFrag bf;
          bf.q = wfrag[((s * J + j) * 2 + h) * 32 + r];
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t].v, bf.v, acc[s][t], 0, 0, 0);
        }
      }
    };

    if constexpr (NCH > 0) {
      if (sb != (int)blockIdx.x) request(pix0);   // (the first tile's: in front of the split)
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) mma(ch, a16[ch]);
    } else {
      uint4 cur[TPW], nxt[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) cur[t] = *reinterpret_cast<const uint4*>(src[t] + chunk_off(0));
      for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) {
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            nxt[t] = *reinterpret_cast<const uint4*>(src[t] + chunk_off(ch + 1));
        }
        mma(ch, cur);
#pragma unroll
        for (int t = 0; t < TPW; ++t) cur[t] = nxt[t];
      }
    }

    // epilogue, specialised on the (uniform) activation and on whole-tile stores
    auto emit = [&](auto actc, auto fullc) {
      constexpr int ACT = decltype(actc)::value;
      constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const int m = pix0 + t * 32 + row;
          if (!FULL && m >= p.M) continue;
          // small pieces first; then the reference's Lambda(x / 255) applied to the sum: the
          // quotient by Markstein's sequence (q0 = s * RN(1/d); q = q0 + (s - d q0) * RN(1/d)),
          // the correctly rounded s / d given a correctly rounded reciprocal
          const float sum = (acc[2][t][e] + acc[1][t][e]) + acc[0][t][e];
          const float q0 = sum * p.a_rcp;
          float v = __builtin_fmaf(__builtin_fmaf(-p.a_div, q0, sum), p.a_rcp, q0) + bv;
          if (ACT == AA_ACT_RELU) v = v > 0.f ? v : 0.f;
          if (ACT == AA_ACT_TANH) v = tanhf(v);
          p.C[(size_t)m * p.ldc + r] = v;
        }
      }
    };
    auto emit_act = [&](auto fullc) {
      if (p.act == AA_ACT_RELU) emit(std::integral_constant<int, AA_ACT_RELU>{}, fullc);
      else if (p.act == AA_ACT_TANH) emit(std::integral_constant<int, AA_ACT_TANH>{}, fullc);
      else emit(std::integral_constant<int, -1>{}, fullc);
    };
    if (pix0 + 32 * TPW <= p.M) emit_act(std::true_type{});
    else emit_act(std::false_type{});
  }
}

// Shapes this kernel takes (checked by the caller): uint8 forward patches, N == 32 filters,
// patch rows that are whole 32-byte chunks, K <= AA_CU8_MAX_K, no mask / column-sum epilogue.
static bool aa_conv_u8_bf16_ok(const aa_gemm_desc* d) {
  const int seg = d->KW * d->Cin;
  return d->a_mode == AA_A_PATCH_U8 && d->b_mode == AA_B_ROW && d->N == 32 && seg % 32 == 0 &&
         d->K == d->KH * seg && d->K <= AA_CU8_MAX_K && d->mask_src == nullptr &&
         d->colsum_out == nullptr;
}

static void aa_conv_u8_bf16_launch(const GemmP& p, hipStream_t st) {
  constexpr int TPW = 2;
  const int n_super = (p.M + 128 * TPW - 1) / (128 * TPW);
  const int nch = p.K >> 5;
  const size_t smem = (size_t)3 * (p.K >> 4) * 2 * 32 * sizeof(uint4);
  const int per_xcd = (n_super + 7) >> 3;
  int grid = per_xcd * 8;
  if (grid > 1024) grid = 1024;
  if (nch == 8)
    hipLaunchKernelGGL((aa_conv_u8_bf16x3_kernel<TPW, 8>), dim3(grid), dim3(AA_CU8_THREADS), smem,
                       st, p, n_super, nch);
  else
    hipLaunchKernelGGL((aa_conv_u8_bf16x3_kernel<TPW, 0>), dim3(grid), dim3(AA_CU8_THREADS), smem,
                       st, p, n_super, nch);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same layer: dW[k][f] = (sum_pix u8[pix][k] * dZ[pix][f]) / a_div, with the
// bias gradient sum_pix dZ[pix][f] fused (tf.GradientTape of keras Conv2D + BiasAdd).
//
// The reduction runs over PIXELS, so the MFMA's K is the pixel index: a 16-pixel chunk is one
// v_mfma_f32_32x32x16_bf16 per (32 patch elements) x (32 filters) x piece.  A workgroup takes whole
// frames.  Per frame it stages, with every load in flight at once, (a) the frame's bytes into LDS
// (28 KB at the Atari size) and (b) the frame's dZ rows split into three exact bf16 pieces, stored
// in B-fragment order (lane = filter, 8 consecutive pixels = one ds_read_b128).  The A fragment
// of a patch element is its byte at 8 consecutive pixels: 8 ds_read_u8 at the pixels' patch origins
// (32 lanes = the 32 contiguous bytes of one patch-row chunk).  Wave w owns the patch-row chunks
// {w, w+4}; the four waves walk the same pixels, so nothing is reduced inside the workgroup.  The
// workgroup's partial gradient is slab g of the deterministic split reduce
// (aa_splitk_reduce_kernel); the 1/a_div quotient is applied to each slab.
// ------------------------------------------------------------------------------------------------
#define AA_CU8_DW_MAX_CH 8      /* patch-row chunks: 256 patch elements max */
#define AA_CU8_DW_WAVES 8
#define AA_CU8_DW_MAX_FRAME 32768   /* frame bytes staged in LDS */
#define AA_CU8_DW_MAX_OHW 512       /* output pixels per frame (3 * 64 B of LDS each) */

// gfx950's 8-bit LDS transpose read (semantics established by tools/tr8_probe.hip): the 16 lanes
// of a group each supply the address of an 8-byte chunk; lane j receives, as element e = 0..7, byte
// (j & 7) of the chunk supplied by lane 2e + (j >> 3).  With lane i supplying the 8 patch bytes
// [8 (i & 1), 8 (i & 1) + 8) of pixel (i >> 1), lane j gets the byte of patch element j at the 8
// pixels: the MFMA operand of the weight gradient (reduction index = pixel) in ONE LDS
// instruction instead of eight ds_read_u8.  (The compiler builtin, not inline assembly: the
// compiler then counts the read among its outstanding LDS operations and places the waits.)
typedef int aa_i32x2_t __attribute__((ext_vector_type(2)));
__device__ static inline aa_i32x2_t aa_ds_read_tr8(const unsigned char* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr8_b64_v2i32(
      (__attribute__((address_space(3))) aa_i32x2_t*)lds_ptr);
}

template <int NW, bool TR8>   // waves per workgroup; wave w owns the patch-row chunks {w, w + NW, ...}
__global__ void __launch_bounds__(NW * 64)
aa_conv_u8_dw_bf16x3_kernel(GemmP p, int n_img, int frame_bytes, int want_colsum) {
  constexpr int NT = NW * 64;
  constexpr int MT = AA_CU8_DW_MAX_CH / NW;
  extern __shared__ __attribute__((aligned(16))) uint4 dyn[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform branches
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int nch = p.M >> 5;                  // patch-row chunks (M = patch elements)
  const int R = p.seg >> 5;
  const int OHW = p.OHW;
  const int n_oct = ((OHW + 15) >> 4) << 1;  // pixel octets per frame, padded to whole chunks
  uint4* zfrag = dyn;                                      // [3][n_oct][32]
  int* origin = reinterpret_cast<int*>(dyn + 3 * n_oct * 32);   // [n_oct * 8] patch origins
  unsigned char* frame = reinterpret_cast<unsigned char*>(origin + n_oct * 8);
  // patch origin (byte offset inside a frame) of every output pixel; padding pixels alias the
  // last real one (their dZ pieces are zero)
  for (int i = tid; i < n_oct * 8; i += NT) {
    const int pix = i < OHW ? i : OHW - 1;
    const int oy = pix / p.OW;
    origin[i] = oy * p.stride * p.rowpitch + (pix - oy * p.OW) * p.stride * p.Cin;
  }
  const unsigned char* A = reinterpret_cast<const unsigned char*>(p.A);

  int toff[MT];                    // byte offset of this lane's patch element
  bool tlive[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int ch = wave + NW * t;
    tlive[t] = ch < nch;
    const int chc = tlive[t] ? ch : 0;
    const int ky = chc / R;
    toff[t] = ky * p.rowpitch + ((chc - ky * R) << 5) + r;
  }
  f32x16 acc[MT][3];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][s][e] = 0.f;
  float csum = 0.f;   // this thread's share of the bias gradient (staging items are filter-major)

  for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
    __syncthreads();   // the previous frame's readers are done
    // (a) frame bytes -> LDS, 16 bytes per load
    const uint4* fsrc = reinterpret_cast<const uint4*>(A + (size_t)img * p.imgpitch);
    uint4* fdst = reinterpret_cast<uint4*>(frame);
    const int n16 = frame_bytes >> 4;
    // four loads per thread in flight per trip, every thread (clamped indices: the threads whose
    // fourth item fell off the end used to take the remainder loop -- one dependent round trip
    // per item, three for more than half of the workgroup at the Atari frame size)
    for (int i0 = tid; i0 < n16; i0 += 4 * NT) {
      // four requests per thread in flight, pinned in front of the stores (native vectors through
      // an empty asm): in the transpose-read variant the compiler otherwise sinks two of them
      // into the conditions below -- one dependent round trip each -- and parks the array on the
      // stack (80 bytes of scratch per lane, tools/kernel_resources.py)
      typedef unsigned aa_u32x4_t __attribute__((ext_vector_type(4)));
      const aa_u32x4_t* vsrc = reinterpret_cast<const aa_u32x4_t*>(fsrc);
      aa_u32x4_t* vdst = reinterpret_cast<aa_u32x4_t*>(fdst);
      const int i1 = i0 + NT, i2 = i0 + 2 * NT, i3 = i0 + 3 * NT;
      aa_u32x4_t t0 = vsrc[i0 < n16 ? i0 : n16 - 1], t1 = vsrc[i1 < n16 ? i1 : n16 - 1];
      aa_u32x4_t t2 = vsrc[i2 < n16 ? i2 : n16 - 1], t3 = vsrc[i3 < n16 ? i3 : n16 - 1];
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
      if (i0 < n16) vdst[i0] = t0;
      if (i1 < n16) vdst[i1] = t1;
      if (i2 < n16) vdst[i2] = t2;
      if (i3 < n16) vdst[i3] = t3;
    }
    // (b) dZ rows of the frame -> three bf16 planes in fragment order (4 items = 32 loads in flight)
    const float* dz = p.B + (size_t)img * OHW * p.ldb;
    const int n_item = n_oct * 32;
    for (int item0 = tid; item0 < n_item; item0 += NT * 4) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int item = item0 + u * NT;
        const int c = item & 31, o = item >> 5;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int pix = o * 8 + e;
          v[u][e] = (item < n_item && pix < OHW) ? dz[(size_t)pix * p.ldb + c] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int item = item0 + u * NT;
        if (item >= n_item) continue;
        const int c = item & 31, o = item >> 5;
        unsigned pc[3][4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float r0 = v[u][e], r1 = v[u][e + 1];
          csum += r0 + r1;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const unsigned pk = aa_pk_bf16(r0, r1);
            pc[s][e >> 1] = pk;
            if (s < 2) {
              r0 -= __uint_as_float(pk << 16);
              r1 -= __uint_as_float(pk & 0xffff0000u);
            }
          }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
          zfrag[(s * n_oct + o) * 32 + c] = make_uint4(pc[s][0], pc[s][1], pc[s][2], pc[s][3]);
      }
    }
    __syncthreads();

    if constexpr (TR8) {
      // one transpose read per fragment: this lane supplies 8 patch bytes of pixel (lane & 15) >> 1
      // of its octet, at elements [16 ((lane >> 4) & 1) + 8 (lane & 1), + 8) of the wave's chunk
      const int sub = (lane & 15) >> 1;
      const int eoff = (((lane >> 4) & 1) << 4) + ((lane & 1) << 3) - r;   // toff holds + r
      // software pipeline: the fragment of step s + 1 is requested -- its pixel origin was loaded
      // a step earlier -- before the MFMAs of step s, the origin of step s + 2 behind it
      const int last_o = n_oct - 2 + h;
      auto org_of = [&](int o) { return origin[(o <= last_o ? o : last_o) * 8 + sub]; };
      aa_i32x2_t fa[MT], fb[MT];   // two register sets: a step never copies a fragment
      int org_nxt = org_of(h + 2);
      {
        const int org = org_of(h);
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = aa_ds_read_tr8(frame + org + toff[t] + eoff);
      }
      auto step = [&](int o0, const aa_i32x2_t (&c)[MT], aa_i32x2_t (&n)[MT]) {
        const int o = o0 + h;
        AaFrag bf[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) bf[s].q = zfrag[(s * n_oct + o) * 32 + r];
#pragma unroll
        for (int t = 0; t < MT; ++t) n[t] = aa_ds_read_tr8(frame + org_nxt + toff[t] + eoff);
        org_nxt = org_of(o + 4);
        // (the scheduler otherwise sinks the requests above behind the MFMAs below, and the
        // next step then starts by waiting for its fragment)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          // (a wave without a live chunk computes on chunk 0 and stores nothing: no branch in
          // the loop)
          AaFrag af;
          aa_u8x4_to_bf16((unsigned)c[t].x, af.q.x, af.q.y);
          aa_u8x4_to_bf16((unsigned)c[t].y, af.q.z, af.q.w);
#pragma unroll
          for (int s = 0; s < 3; ++s)
            acc[t][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.v, bf[s].v, acc[t][s], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // (nor the next step's address arithmetic above them)
      };
      int o0 = 0;
      for (; o0 + 2 < n_oct; o0 += 4) {
        step(o0, fa, fb);
        step(o0 + 2, fb, fa);
      }
      if (o0 < n_oct) step(o0, fa, fb);
    } else
    for (int o0 = 0; o0 < n_oct; o0 += 2) {
      const int o = o0 + h;
      int base[8];   // this lane's 8 pixels
      {
        const int4 b0 = *reinterpret_cast<const int4*>(origin + o * 8);
        const int4 b1 = *reinterpret_cast<const int4*>(origin + o * 8 + 4);
        base[0] = b0.x; base[1] = b0.y; base[2] = b0.z; base[3] = b0.w;
        base[4] = b1.x; base[5] = b1.y; base[6] = b1.z; base[7] = b1.w;
      }
      AaFrag bf[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) bf[s].q = zfrag[(s * n_oct + o) * 32 + r];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (!tlive[t]) continue;
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const unsigned f0 = __float_as_uint((float)frame[base[e] + toff[t]]);
          const unsigned f1 = __float_as_uint((float)frame[base[e + 1] + toff[t]]);
          w[e >> 1] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
        }
        AaFrag af;
        af.q = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
        for (int s = 0; s < 3; ++s)
          acc[t][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.v, bf[s].v, acc[t][s], 0, 0, 0);
      }
    }
  }

  // slab g: [M][32], then the column-sum rows [splits][32] behind all slabs
  const int g = blockIdx.x;
  float* slab = p.C + (size_t)g * p.M * 32;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (!tlive[t]) continue;
    const int ch = wave + NW * t;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      const float sum = (acc[t][2][e] + acc[t][1][e]) + acc[t][0][e];
      const float q0 = sum * p.a_rcp;
      slab[(size_t)(ch * 32 + row) * 32 + r] =
          __builtin_fmaf(__builtin_fmaf(-p.a_div, q0, sum), p.a_rcp, q0);
    }
  }
  if (want_colsum) {
    // thread tid staged items tid, tid+256, ...: always filter (tid & 31); fixed-order sum of the
    // 8 threads of a filter
    __syncthreads();
    float* red = reinterpret_cast<float*>(dyn);
    red[tid] = csum;
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < NT / 32; ++j) t += red[j * 32 + tid];
      p.C[(size_t)gridDim.x * p.M * 32 + (size_t)g * 32 + tid] = t;
    }
  }
}

static bool aa_conv_u8_dw_bf16_ok(const aa_gemm_desc* d) {
  const int seg = d->KW * d->Cin;
  if (!(d->a_mode == AA_A_PATCH_T_U8 && d->b_mode == AA_B_ROW && d->N == 32 && seg % 32 == 0 &&
        d->M == d->KH * seg && d->M <= 32 * AA_CU8_DW_MAX_CH && d->mask_src == nullptr &&
        d->bias == nullptr && d->act == AA_ACT_NONE && d->n_img >= 1 && d->stride >= 1))
    return false;
  const int64_t frame = (int64_t)d->H * d->W * d->Cin;
  const int OH = (d->H - d->KH) / d->stride + 1, OW = (d->W - d->KW) / d->stride + 1;
  return frame % 16 == 0 && frame <= AA_CU8_DW_MAX_FRAME && OH >= 1 && OW >= 1 &&
         OH * OW <= AA_CU8_DW_MAX_OHW && (int64_t)d->n_img * OH * OW == d->K;
}
// slabs of the split reduce = workgroups; each takes frames g, g + groups, ...
static int aa_conv_u8_dw_groups(int n_img) {
  int g = n_img > 256 ? 256 : n_img;
  return g < 2 ? 2 : g;
}
static size_t aa_conv_u8_dw_lds(const GemmP& p, int frame_bytes) {
  const int n_oct = ((p.OHW + 15) >> 4) << 1;
  return (size_t)3 * n_oct * 32 * sizeof(uint4) + (size_t)n_oct * 8 * sizeof(int) +
         (size_t)frame_bytes;
}
static int aa_conv_u8_dw_bf16_launch(const GemmP& p, int n_img, int frame_bytes, int groups,
                                     hipStream_t st) {
  const size_t smem = aa_conv_u8_dw_lds(p, frame_bytes);
  constexpr int NW = AA_CU8_DW_WAVES;
  // the transpose-read loop needs 8-byte aligned chunk addresses: pixel origins and patch rows
  const bool tr8 = (p.stride * p.Cin) % 8 == 0 && p.rowpitch % 8 == 0;
  static size_t lds_limit[AA_MAX_DEVICES][2] = {{0}};   // dynamic LDS above 64 KiB: granted per
  const int dv = aa_device_ordinal();                   // kernel and device
  if (dv < 0) return AA_ERR_LAUNCH;
  if (smem > lds_limit[dv][tr8]) {
    const void* fn = tr8 ? (const void*)aa_conv_u8_dw_bf16x3_kernel<NW, true>
                         : (const void*)aa_conv_u8_dw_bf16x3_kernel<NW, false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess)
      return AA_ERR_LAUNCH;
    lds_limit[dv][tr8] = smem;
  }
  if (tr8)
    hipLaunchKernelGGL((aa_conv_u8_dw_bf16x3_kernel<NW, true>), dim3(groups), dim3(NW * 64), smem,
                       st, p, n_img, frame_bytes, p.colsum_out != nullptr ? 1 : 0);
  else
    hipLaunchKernelGGL((aa_conv_u8_dw_bf16x3_kernel<NW, false>), dim3(groups), dim3(NW * 64), smem,
                       st, p, n_img, frame_bytes, p.colsum_out != nullptr ? 1 : 0);
  return aa_launch_status();
}
