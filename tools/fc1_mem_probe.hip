// What bounds the fc1 forward's k loop?  The operand traffic of aa_gemm_dma_kernel<0,3,32,64,...>
// (X[256][3136] row tiles + W[3136][512] column tiles, split-K) replayed WITHOUT the LDS operand
// fetch, with / without MFMAs, through LDS-DMA or plain register loads, at several ring depths.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fc1_mem_probe.hip -o tools/_bin/fc1_mem_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define OOB 0x80000000u

__device__ static inline i32x4 make_desc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 d;
  d.x = (int)(unsigned)(a & 0xffffffffull);
  d.y = (int)(unsigned)((a >> 32) & 0xffffull);
  d.z = (int)bytes;
  d.w = 0x00020000;
  return d;
}
__device__ static inline void dma16(unsigned voff, i32x4 desc, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(desc), "s"(lds_byte) : "memory");
}
template <int N> __device__ static inline void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct P {
  const float* X; const float* W; float* out;
  int M, N, K, kps, splits, gx, gy;
  int mfma_per_tile;   // per wave
};

// XCD-aware order: block L -> XCD c = L & 7 takes K-split c's (x, y) tiles when splits == 8;
// generally: i = (L & 7) * per + (L >> 3), z slowest
__device__ static inline void block_of(const P& p, int& x, int& y, int& z) {
  const int n = p.gx * p.gy * p.splits, per = (n + 7) >> 3;
  const int L = blockIdx.x;
  const int i = (L & 7) * per + (L >> 3);
  z = i / (p.gx * p.gy);
  const int r = i - z * (p.gx * p.gy);
  y = r / p.gx;
  x = r - y * p.gx;
}

// MODE 0: LDS-DMA ring.  MODE 1: register ring (global loads), consumed by an OR chain.
template <int MODE, int NS, int BM, int BN>
__global__ void __launch_bounds__(256) mem_kernel(P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int bx, by, bz;
  block_of(p, bx, by, bz);
  if (bz >= p.splits) return;
  const int m0 = bx * BM, n0 = by * BN;
  const int k_begin = bz * p.kps;
  int k_end = k_begin + p.kps; if (k_end > p.K) k_end = p.K;
  const int nk = (k_end - k_begin + 31) / 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const i32x4 rA = make_desc(p.X, (unsigned)((size_t)p.M * p.K * 4));
  const i32x4 rB = make_desc(p.W, (unsigned)((size_t)p.K * p.N * 4));
  // granules of a stage: A = BM rows x 8 granules (32 k), B = 32 k rows x BN/4 granules
  constexpr int GA = BM * 8, GB = 32 * (BN / 4);
  constexpr int NA = (GA + 255) / 256, NB = (GB + 255) / 256;
  constexpr int STAGE = (NA + NB) * 4096;
  unsigned baseA[NA], baseB[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int G = (q * 4 + wave) * 64 + lane;
    const int x = G >> 3, kg = G & 7;
    baseA[q] = (G < GA && m0 + x < p.M) ? 4u * ((unsigned)(m0 + x) * p.K + 4 * kg) : OOB;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int G = (q * 4 + wave) * 64 + lane;
    const int k = G / (BN / 4), xg = G - k * (BN / 4);
    baseB[q] = (G < GB) ? 4u * ((unsigned)k * p.N + n0 + 4 * xg) : OOB;
  }
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)smem;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float a_op = (float)lane, b_op = 1.0f;

  if constexpr (MODE == 0) {
    auto issue = [&](int t) {
      const unsigned st = lds0 + (unsigned)(t % NS) * STAGE;
      const int k0 = k_begin + t * 32;
      const bool ok = t < nk;
#pragma unroll
      for (int q = 0; q < NA; ++q)
        dma16(ok ? baseA[q] + 4u * k0 : OOB, rA, st + (unsigned)(q * 4 + wave) * 1024u);
#pragma unroll
      for (int q = 0; q < NB; ++q)
        dma16(ok ? baseB[q] + 4u * (unsigned)k0 * p.N : OOB, rB,
              st + (unsigned)((NA + q) * 4 + wave) * 1024u);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t);
    for (int t = 0; t < nk; ++t) {
      wait_vm<(NS - 2) * (NA + NB)>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(t + NS - 1);
      for (int j = 0; j < p.mfma_per_tile; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op, b_op, acc, 0, 0, 0);
    }
    wait_vm<0>();
  } else {
    f32x4 ring[NS][NA + NB];
    const __amdgpu_buffer_rsrc_t qA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.X), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t qB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.W), 0, (int)((size_t)p.K * p.N * 4), 0x00020000);
    auto load = [&](int t, int s) {
      const int k0 = k_begin + t * 32;
      const bool ok = t < nk;
#pragma unroll
      for (int q = 0; q < NA; ++q)
        ring[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            qA, ok ? baseA[q] + 4u * k0 : OOB, 0, 0));
#pragma unroll
      for (int q = 0; q < NB; ++q)
        ring[s][NA + q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            qB, ok ? baseB[q] + 4u * (unsigned)k0 * p.N : OOB, 0, 0));
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) load(t, t);
    const int nk_r = (nk + NS - 1) / NS * NS;
    for (int t0 = 0; t0 < nk_r; t0 += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        load(t0 + s + NS - 1, (s + NS - 1) % NS);
#pragma unroll
        for (int q = 0; q < NA + NB; ++q) a_op += ring[s][q].x;
        for (int j = 0; j < p.mfma_per_tile; ++j)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op, b_op, acc, 0, 0, 0);
      }
    }
  }
  float s = a_op;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += acc[e];
  if (s == 12345.678f) p.out[blockIdx.x * 256 + threadIdx.x] = s;   // keep everything alive
}

template <int MODE, int NS, int BM, int BN>
static void bench(const char* tag, P p, int splits, int mfma) {
  p.splits = splits;
  p.kps = ((p.K + splits - 1) / splits + 7) / 8 * 8;
  p.gx = (p.M + BM - 1) / BM; p.gy = (p.N + BN - 1) / BN;
  p.mfma_per_tile = mfma;
  const int n = p.gx * p.gy * splits;
  const int grid = (n + 7) / 8 * 8;
  constexpr int GA = BM * 8, GB = 32 * (BN / 4);
  constexpr int NA = (GA + 255) / 256, NB = (GB + 255) / 256;
  const size_t smem = MODE == 0 ? (size_t)NS * (NA + NB) * 4096 : 0;
  hipFuncSetAttribute((const void*)mem_kernel<MODE, NS, BM, BN>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mem_kernel<MODE, NS, BM, BN>), dim3(grid), dim3(256), smem, 0, p);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((mem_kernel<MODE, NS, BM, BN>), dim3(grid), dim3(256), smem, 0, p);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)n * ((double)(p.kps + 31) / 32) * (GA + GB) * 16;
  printf("%-28s %dx%d tile, %2d splits (%4d WGs), ring %d, %d mfma/tile/wave: %6.2f us  (%.1f MB moved, %.2f TB/s)%s\n",
         tag, BM, BN, splits, n, NS, mfma, ms * 1000 / 40, bytes / 1e6, bytes / (ms / 40 * 1e-3) / 1e12,
         hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
  P p{};
  p.M = 256; p.N = 512; p.K = 3136;
  float *X, *W, *out;
  hipMalloc(&X, (size_t)p.M * p.K * 4); hipMalloc(&W, (size_t)p.K * p.N * 4); hipMalloc(&out, 1 << 22);
  hipMemset(X, 0, (size_t)p.M * p.K * 4); hipMemset(W, 0, (size_t)p.K * p.N * 4);
  p.X = X; p.W = W; p.out = out;
  // empty launch cost for reference
  bench<0, 4, 32, 64>("lds-dma, no mfma", p, 8, 0);
  bench<0, 4, 32, 64>("lds-dma + mfma", p, 8, 8);
  bench<0, 4, 32, 64>("lds-dma, mfma only-ish", p, 8, 16);
  bench<0, 6, 32, 64>("lds-dma, no mfma", p, 8, 0);
  bench<0, 6, 32, 64>("lds-dma + mfma", p, 8, 8);
  bench<0, 8, 32, 64>("lds-dma, no mfma", p, 8, 0);
  bench<0, 8, 32, 64>("lds-dma + mfma", p, 8, 8);
  bench<0, 4, 32, 64>("lds-dma, no mfma", p, 4, 0);
  bench<0, 4, 32, 64>("lds-dma + mfma", p, 4, 8);
  bench<0, 8, 32, 64>("lds-dma, no mfma", p, 4, 0);
  bench<0, 8, 32, 64>("lds-dma + mfma", p, 4, 8);
  bench<0, 12, 32, 64>("lds-dma + mfma", p, 4, 8);
  bench<0, 4, 32, 64>("lds-dma, no mfma", p, 16, 0);
  bench<0, 4, 32, 64>("lds-dma + mfma", p, 16, 8);
  bench<0, 4, 64, 64>("lds-dma, no mfma", p, 8, 0);
  bench<0, 4, 64, 64>("lds-dma + mfma", p, 8, 16);
  bench<0, 6, 64, 64>("lds-dma + mfma", p, 8, 16);
  bench<0, 4, 64, 128>("lds-dma, no mfma", p, 16, 0);
  bench<0, 4, 64, 128>("lds-dma + mfma", p, 16, 32);
  bench<1, 2, 32, 64>("reg ring, no mfma", p, 8, 0);
  bench<1, 4, 32, 64>("reg ring, no mfma", p, 8, 0);
  bench<1, 4, 32, 64>("reg ring + mfma", p, 8, 8);
  bench<1, 8, 32, 64>("reg ring, no mfma", p, 8, 0);
  bench<1, 8, 32, 64>("reg ring + mfma", p, 8, 8);
  bench<1, 8, 32, 64>("reg ring + mfma", p, 4, 8);
  bench<1, 4, 64, 64>("reg ring + mfma", p, 8, 16);
  // MFMA alone: K walk without loads is not expressible here; report mfma-heavy instead
  bench<0, 4, 32, 64>("lds-dma + 2x mfma", p, 8, 16);
  return 0;
}
