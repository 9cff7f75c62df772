// Mapping dump for gfx950's 8-bit LDS transpose read `ds_read_b64_tr_b8` (no documentation in the
// image beyond its name): LDS byte i holds i (low byte in pass 0, high byte in pass 1), lane l reads
// at byte address addr(l) for a few address patterns; prints, per lane, which eight bytes came back.
// Intended use: the uint8 conv1 weight gradient (conv_u8_bf16.h), whose MFMA operand needs the byte
// of ONE patch element at 8 consecutive pixels (8 `ds_read_u8` per fragment today).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tr8_probe.hip -o tools/_bin/tr8_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int i2 __attribute__((ext_vector_type(2)));

__device__ static inline i2 tr8_read(uint32_t a) {
  i2 v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// pattern 0: addr = lane * 8 (512 contiguous bytes)
// pattern 1: 16-lane group g, lane j: row j of a [16][8]-byte block, row pitch 8 B, block g at g * 128
// pattern 2: row j, row pitch 16 B, block g at g * 256
// pattern 3: lane j -> row (j >> 1), 8-byte half (j & 1) of a [8][16]-byte block, pitch 16 B, block g * 128
// pattern 4: row j, row pitch 64 B, block g at g * 8 (four interleaved column blocks)
__global__ void dump(int pattern, int pass, uint8_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = pass == 0 ? (uint8_t)i : (uint8_t)(i >> 8);
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, j = l & 15;
  uint32_t a = 0;
  if (pattern == 0) a = l * 8;
  if (pattern == 1) a = g * 128 + j * 8;
  if (pattern == 2) a = g * 256 + j * 16;
  if (pattern == 3) a = g * 128 + (j >> 1) * 16 + (j & 1) * 8;
  if (pattern == 4) a = g * 8 + j * 64;
  const uint32_t base = (uint32_t)(uintptr_t)lds;
  const i2 v = tr8_read(base + a);
  const uint32_t w0 = (uint32_t)v[0], w1 = (uint32_t)v[1];
  for (int e = 0; e < 4; ++e) {
    out[l * 8 + e] = (uint8_t)(w0 >> (8 * e));
    out[l * 8 + 4 + e] = (uint8_t)(w1 >> (8 * e));
  }
}

int main() {
  uint8_t *d, lo[512], hi[512];
  if (hipMalloc(&d, 512) != hipSuccess) return 1;
  for (int p = 0; p < 5; ++p) {
    dump<<<1, 64>>>(p, 0, d);
    hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
    dump<<<1, 64>>>(p, 1, d);
    hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("pattern %d: launch failed\n", p); return 1; }
    printf("pattern %d (lane: the LDS byte offsets of its 8 result bytes, element 0 first)\n", p);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int e = 0; e < 8; ++e) printf(" %4d", (int)lo[l * 8 + e] | ((int)hi[l * 8 + e] << 8));
      printf("\n");
    }
  }
  return 0;
}
