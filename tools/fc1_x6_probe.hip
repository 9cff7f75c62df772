// Feasibility probe for the fc1 forward on the bf16 matrix cores with the weight operand PRE-SPLIT
// (three bf16 planes in the natural [k][n] layout, as an optimizer would maintain them) and the
// 64-row activation operand split while its tile is staged: the instruction mix and the memory
// traffic of the planned kernel, on dummy data (no result is checked).
//   workgroup = 64 x 128 output tile over K / splits, 4 waves of 32 x 64 (2 x 4 MFMA tiles),
//   per 32-k step: LDS-DMA of A (fp32, 8 KB) + B planes (24 KB) three steps ahead, split pass
//   A fp32 -> three bf16 planes in LDS, A fragments ds_read_b128, B fragments
//   ds_read_b64_tr_b16 (transposing read of the [k][n] planes), 48 x v_mfma_f32_16x16x32_bf16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iagents_amd/csrc tools/fc1_x6_probe.hip -o tools/_bin/fc1_x6_probe
#include <hip/hip_runtime.h>
#include "x6_common.h"

#include <cstdio>

typedef int i32x4 __attribute__((ext_vector_type(4)));
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
__device__ static inline uint2 tr_read(uint32_t lds_addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
  return v;
}

struct P {
  const float* X;          // [M][K] fp32
  const uint16_t* Wp;      // [3][K][N] bf16 planes
  float* slab;             // [splits][M][N]
  int M, N, K, kps, splits, gx, gy;
  int mode;                // bit 0: skip the split pass, bit 1: skip MFMAs, bit 2: skip fragment reads
};

constexpr int BM = 64, BN = 128, NS = 3;
constexpr int A_STAGE = BM * 32 * 4;            // 8 KB fp32
constexpr int B_PLANE = 32 * BN * 2;            // 8 KB per plane
constexpr int STAGE = A_STAGE + 3 * B_PLANE;    // 32 KB
constexpr int AP_PITCH = 80, AP_PLANE = BM * AP_PITCH, AP_BUF = 3 * AP_PLANE;

__global__ void __launch_bounds__(256) x6_fwd_probe(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int n_all = p.gx * p.gy * p.splits, per = (n_all + 7) >> 3;
  const int L = blockIdx.x, bi = (L & 7) * per + (L >> 3);
  if (bi >= n_all) return;
  const int bz = bi / (p.gx * p.gy), rem = bi - bz * (p.gx * p.gy);
  const int by = rem / p.gx, bx = rem - by * p.gx;
  const int m0 = bx * BM, n0 = by * BN;
  const int k_begin = bz * p.kps;
  int k_end = k_begin + p.kps; if (k_end > p.K) k_end = p.K;
  const int nk = (k_end - k_begin + 31) / 32;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave & 1, wn = wave >> 1;
  const i32x4 rA = make_desc(p.X, (unsigned)((size_t)p.M * p.K * 4));
  const i32x4 rB = make_desc(p.Wp, (unsigned)((size_t)3 * p.K * p.N * 2));
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds;
  char* ap = lds + NS * STAGE;
  // DMA roles: A = 512 granules (row = G / 8, k granule = G % 8), 2 per thread; B = 3 planes x
  // 32 rows x 16 granules = 1536, 6 per thread; the B granule position is XOR-swizzled with the row
  unsigned srcA[2], srcB[6];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int G = (q * 4 + wave) * 64 + lane, row = G >> 3, kg = G & 7;
    srcA[q] = 4u * ((unsigned)(m0 + row) * p.K + 4 * kg);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int G = (q * 4 + wave) * 64 + lane;      // 0 .. 1535
    const int pl = G / 512, g2 = G - pl * 512, k = g2 >> 4, o = (g2 & 15) ^ (2 * (k & 7));
    srcB[q] = 2u * ((unsigned)pl * p.K * p.N + (unsigned)k * p.N + n0 + 8 * o);
  }
  auto issue = [&](int t) {
    const unsigned st = lds0 + (unsigned)(t % NS) * STAGE;
    const bool ok = t < nk;
    const int k0 = k_begin + t * 32;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      dma16(ok ? srcA[q] + 4u * k0 : OOB, rA, st + (unsigned)(q * 4 + wave) * 1024u);
#pragma unroll
    for (int q = 0; q < 6; ++q)
      dma16(ok ? srcB[q] + 2u * (unsigned)k0 * p.N : OOB, rB,
            st + A_STAGE + (unsigned)(q * 4 + wave) * 1024u);
  };
  // fragment addresses (stage-relative): A rows of this wave's two row tiles; B tr reads of its
  // four column tiles (row k = 8 g + 4 h + (j >> 2), swizzled granule, quad j & 3)
  const int lr = lane & 15, lg = lane >> 4;
  unsigned a_off[2], b_off[4][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) a_off[rt] = (unsigned)((wm * 32 + rt * 16 + lr) * AP_PITCH + lg * 16);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 8 * lg + 4 * h + (lr >> 2), quad = (wn * 4 + c) * 4 + (lr & 3);
      const int o = (quad >> 1) ^ (2 * (k & 7));
      b_off[c][h] = (unsigned)(A_STAGE + k * (BN * 2) + o * 16 + (quad & 1) * 8);
    }
  cx_f32x4 big[4][2], small[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { big[c][rt] = cx_f32x4{0, 0, 0, 0}; small[c][rt] = cx_f32x4{0, 0, 0, 0}; }

  auto split = [&](int t) {      // A tile of step t: fp32 stage -> three bf16 planes in ap[t & 1]
    if (p.mode & 1) return;
    const char* sa = lds + (t % NS) * STAGE;
    const int row = tid >> 2, kq = tid & 3;
    const float4 u0 = *reinterpret_cast<const float4*>(sa + row * 128 + kq * 32);
    const float4 u1 = *reinterpret_cast<const float4*>(sa + row * 128 + kq * 32 + 16);
    const float a[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
    uint4 f[3];
    cx_split8(a, f);
    char* d = ap + (t & 1) * AP_BUF + row * AP_PITCH + kq * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(d + pl * AP_PLANE) = f[pl];
  };

#pragma unroll
  for (int t = 0; t < NS - 1; ++t) issue(t);
  wait_vm<(NS - 2) * 8>();
  __builtin_amdgcn_s_barrier();
  issue(NS - 1);
  split(0);
  for (int t = 0; t < nk; ++t) {
    wait_vm<(NS - 2) * 8>();           // tile t+1 landed (this wave's part)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // ... for everybody; ap[t & 1] complete; ap[(t+1)&1] free
    asm volatile("" ::: "memory");
    issue(t + NS);
    // fragments of step t
    const unsigned apb = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)(ap + (t & 1) * AP_BUF);
    const unsigned stb = lds0 + (unsigned)(t % NS) * STAGE;
    CxFrag fa[2][3], fb[4][3];
    if (!(p.mode & 4)) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fa[rt][pl].q = *reinterpret_cast<const uint4*>(ap + (t & 1) * AP_BUF + pl * AP_PLANE + a_off[rt]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint2 r0 = tr_read(stb + pl * B_PLANE + b_off[c][0]);
          const uint2 r1 = tr_read(stb + pl * B_PLANE + b_off[c][1]);
          fb[c][pl].q = make_uint4(r0.x, r0.y, r1.x, r1.y);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[rt][pl].q = make_uint4(lane, t, pl, rt);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[c][pl].q = make_uint4(lane, t, pl, c);
    }
    (void)apb;
    split(t + 1);                      // next step's A planes: VALU under the MFMAs below
    if (!(p.mode & 2)) {
#pragma unroll
      for (int c = 0; c < 4; ++c) cx_mma6<2>(fa, fb[c], big[c], small[c]);
    }
  }
  wait_vm<0>();
  // epilogue: the slab tile (D layout: lane (n = lr, rows 4 lg .. 4 lg + 3))
  float* out = p.slab + (size_t)bz * p.M * p.N;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 32 + rt * 16 + 4 * lg + e, n = n0 + (wn * 4 + c) * 16 + lr;
        out[(size_t)m * p.N + n] = big[c][rt][e] + small[c][rt][e];
      }
}

static void bench(const char* tag, P p, int splits, int mode) {
  p.splits = splits; p.mode = mode;
  p.kps = ((p.K + splits - 1) / splits + 31) / 32 * 32;
  p.gx = p.M / BM; p.gy = p.N / BN;
  const int n = p.gx * p.gy * splits, grid = (n + 7) / 8 * 8;
  const size_t smem = (size_t)NS * STAGE + 2 * AP_BUF;
  hipFuncSetAttribute((const void*)x6_fwd_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(x6_fwd_probe, dim3(grid), dim3(256), smem, 0, p);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(x6_fwd_probe, dim3(grid), dim3(256), smem, 0, p);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %2d splits (%4d WGs, %d k-steps each): %6.2f us per launch%s\n", tag, splits, n,
         (p.kps + 31) / 32, ms * 1000 / 40, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
  P p{};
  p.M = 256; p.N = 512; p.K = 3136;
  float* X; uint16_t* W; float* slab;
  hipMalloc(&X, (size_t)p.M * p.K * 4); hipMalloc(&W, (size_t)3 * p.K * p.N * 2);
  hipMalloc(&slab, (size_t)32 * p.M * p.N * 4);
  hipMemset(X, 0, (size_t)p.M * p.K * 4); hipMemset(W, 0, (size_t)3 * p.K * p.N * 2);
  p.X = X; p.Wp = W; p.slab = slab;
  for (int splits : {16, 8, 32}) {
    bench("full (dma + split + fragments + 48 mfma)", p, splits, 0);
    bench("no split pass", p, splits, 1);
    bench("no mfma", p, splits, 2);
    bench("no fragment reads (mfma on registers)", p, splits, 4);
    bench("dma + barriers only", p, splits, 7);
  }
  return 0;
}
