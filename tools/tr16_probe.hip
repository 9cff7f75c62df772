// Semantics probe for gfx950's LDS transpose read `ds_read_b64_tr_b16` (no documentation in the
// image; /opt/skills/guides only says "each 16-lane group reads one [4-key][16-col] block, 4
// contiguous bf16 per lane at its own 8-byte-aligned address").  Two parts:
//   1. mapping dump: LDS word i holds the value i; lane l reads at byte address addr(l) for a few
//      address patterns; prints, per lane, which four 16-bit words came back.
//   2. self-check of the intended use (next round's per-frame conv weight gradient on the bf16
//      cores): D[16x16] = A^T-style fragments built from a ROW-major [k][m]
//      image with tr reads, against a host reference, through v_mfma_f32_16x16x32_bf16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tr16_probe.hip -o tools/_bin/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short b8 __attribute__((ext_vector_type(8)));
typedef short b4 __attribute__((ext_vector_type(4)));

__device__ static inline b4 tr_read(uint32_t lds_byte_addr) {
  b4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_byte_addr) : "memory");
  return v;
}

// pattern 0: addr = lane * 8 (512 contiguous bytes)
// pattern 1: 16-lane group g: lane j -> row (j / 4) of a [4][16] block, 8 bytes at column 4 (j % 4);
//            row pitch 32 B, block g at g * 128 B              (the guide's description)
// pattern 2: same block shape, row pitch 64 B, block g at g * 256 B
// pattern 3: lane j -> row (j % 4), column group (j / 4); row pitch 32 B
__global__ void dump(int pattern, uint16_t* out) {
  __shared__ uint16_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, j = l & 15;
  uint32_t a = 0;
  if (pattern == 0) a = l * 8;
  if (pattern == 1) a = g * 128 + (j >> 2) * 32 + (j & 3) * 8;
  if (pattern == 2) a = g * 256 + (j >> 2) * 64 + (j & 3) * 8;
  if (pattern == 3) a = g * 128 + (j & 3) * 32 + (j >> 2) * 8;
  const uint32_t base = (uint32_t)(uintptr_t)lds;   // LDS aperture offset (low 32 bits)
  const b4 v = tr_read(base + a);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (uint16_t)v[e];
}

// ---- part 2: C[m][n] = sum_k X[k][m] * Y[k][n], X and Y row-major in k (the "pixel" index),
// m = 16 channels of x, n = 16 channels of dz, K = 32.  Lane (r = l & 15, g = l >> 4) of the MFMA
// A operand needs X[8g .. 8g+7][r]; with the image [k][16] (32 bytes per k row) that is two tr
// reads whose 16-lane group g covers rows 8g..8g+3 and 8g+4..8g+7.  Which lane must supply which
// row / column-quad address is what part 1 tells; `variant` selects the two candidate assignments.
__device__ static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  return __uint_as_float(u);
}

__global__ void mfma_check(int variant, const uint16_t* X, const uint16_t* Y, float* C) {
  __shared__ uint16_t sx[32 * 16], sy[32 * 16];
  for (int i = threadIdx.x; i < 512; i += 64) { sx[i] = X[i]; sy[i] = Y[i]; }
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, j = l & 15;
  const uint32_t bx = (uint32_t)(uintptr_t)sx, by = (uint32_t)(uintptr_t)sy;
  b8 a, b;
  for (int h = 0; h < 2; ++h) {
    int row, cq;
    if (variant == 0) { row = 8 * g + 4 * h + (j >> 2); cq = j & 3; }
    else { row = 8 * g + 4 * h + (j & 3); cq = j >> 2; }
    const uint32_t off = row * 32 + cq * 8;
    const b4 va = tr_read(bx + off), vb = tr_read(by + off);
    for (int e = 0; e < 4; ++e) { a[4 * h + e] = va[e]; b[4 * h + e] = vb[e]; }
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  // D layout of 16x16 f32: lane (n = l & 15, rows 4 (l >> 4) .. + 3)
  for (int e = 0; e < 4; ++e) C[(4 * g + e) * 16 + j] = acc[e];
}

int main() {
  uint16_t* d_out;
  hipMalloc(&d_out, 64 * 4 * 2);
  std::vector<uint16_t> h(256);
  for (int p = 0; p < 4; ++p) {
    hipLaunchKernelGGL(dump, dim3(1), dim3(64), 0, 0, p, d_out);
    hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (16-bit word indices returned per lane)\n", p);
    for (int l = 0; l < 64; ++l) {
      printf("  l%02d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      if ((l & 3) == 3) printf("\n");
    }
  }
  // part 2
  std::vector<uint16_t> X(512), Y(512);
  std::vector<float> xf(512), yf(512);
  srand(1);
  for (int i = 0; i < 512; ++i) {
    const int a = rand() % 17 - 8, b = rand() % 13 - 6;   // small integers: exact in bf16
    xf[i] = (float)a; yf[i] = (float)b;
    uint32_t ua, ub;
    memcpy(&ua, &xf[i], 4); memcpy(&ub, &yf[i], 4);
    X[i] = (uint16_t)(ua >> 16); Y[i] = (uint16_t)(ub >> 16);
  }
  uint16_t *dX, *dY; float* dC;
  hipMalloc(&dX, 1024); hipMalloc(&dY, 1024); hipMalloc(&dC, 1024);
  hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(dY, Y.data(), 1024, hipMemcpyHostToDevice);
  std::vector<float> C(256), R(256, 0.f);
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n)
      for (int k = 0; k < 32; ++k) R[m * 16 + n] += xf[k * 16 + m] * yf[k * 16 + n];
  for (int v = 0; v < 2; ++v) {
    hipLaunchKernelGGL(mfma_check, dim3(1), dim3(64), 0, 0, v, dX, dY, dC);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0, bad_t = 0;
    for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) bad_t += C[n * 16 + m] != R[m * 16 + n];
    printf("mfma via tr reads, variant %d: %d / 256 mismatches (%d against the transpose)\n", v, bad, bad_t);
  }
  return 0;
}
