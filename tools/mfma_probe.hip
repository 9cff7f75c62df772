// Issue-rate calibration of v_mfma_f32_16x16x32_bf16 / 32x32x16 chains on gfx950: NACC accumulators
// used round-robin (dependency distance NACC), W waves per workgroup, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o tools/_bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short b8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BIG>
__global__ void __launch_bounds__(512) k(float* out, int iters, long long* cyc) {
  b8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  f4 acc[NACC];
  f16v accb[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = f4{0, 0, 0, 0};
    for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
  }
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 18 / NACC * NACC; ++r) {
      if (BIG) accb[r % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accb[r % NACC], 0, 0, 0);
      else acc[r % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[r % NACC], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) { s += acc[i][0]; s += accb[i][0]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, bool BIG>
void run(int waves, const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, BIG><<<256, waves * 64>>>(out, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, BIG><<<256, waves * 64>>>(out, iters, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * (18 / NACC * NACC);
  const double per_simd = waves / 4.0;
  printf("%-10s nacc=%d waves=%d: %.1f clock64-cycles per MFMA per wave, %.2f ns per MFMA per SIMD, "
         "%.0f TFLOP/s\n", name, NACC, waves, c / n, ms * 1e6 / (n * per_simd),
         n * waves * 256 * (BIG ? 32768.0 : 16384.0) / (ms * 1e-3) / 1e12);
}

int main() {
  run<3, false>(8, "16x16x32"); run<3, false>(4, "16x16x32");
  run<6, false>(8, "16x16x32"); run<6, false>(4, "16x16x32");
  run<2, false>(8, "16x16x32"); run<1, false>(8, "16x16x32"); run<9, false>(4, "16x16x32");
  run<3, true>(8, "32x32x16"); run<3, true>(4, "32x32x16"); run<1, true>(8, "32x32x16");
  run<2, true>(4, "32x32x16");
  return 0;
}
