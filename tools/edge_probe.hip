// What does a dependency between two kernels cost on this runtime -- as an edge of a replayed HIP
// graph, and as two launches in stream order?  (Round 5: a PPO collect body of ~20 launches of a few
// microseconds each replays SLOWER as a graph than the eager loop issues it.)
//   chain of N kernels of ~T us each: (a) captured into a graph and replayed, (b) launched back to
//   back into one stream from C; GPU time = events around the chain; host time = the issue calls.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/edge_probe tools/edge_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                             \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                           \
    }                                                                     \
  } while (0)

// one wave per CU-ish grid, spins for `ticks` of the 100 MHz wall clock, touches memory
__global__ void work_kernel(float* p, long long ticks, int n) {
  const long long t0 = wall_clock64();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? p[i] : 0.f;
  while (wall_clock64() - t0 < ticks) v = v * 1.0001f + 1.f;
  if (i < n) p[i] = v;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const int n = 256 * 256;
  float* buf;
  CK(hipMalloc(&buf, n * sizeof(float)));
  CK(hipMemset(buf, 0, n * sizeof(float)));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("chain of N dependent kernels (256 workgroups x 256 lanes, each spinning T us)\n");
  printf("%4s %6s | %12s %12s | %12s %12s | per-edge graph - stream\n", "N", "T us", "graph GPU us",
         "graph host us", "stream GPU us", "stream host us");
  for (int T : {2, 5, 10}) {
    for (int N : {4, 12, 20}) {
      const long long ticks = (long long)T * 100;
      // (a) graph
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < N; ++k)
        hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, st, buf, ticks, n);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      double g_gpu = 1e30, g_host = 1e30, s_gpu = 1e30, s_host = 1e30;
      for (int rep = 0; rep < 12; ++rep) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const double h0 = now_us();
        CK(hipGraphLaunch(ge, st));
        const double h1 = now_us();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) {
          if (ms * 1e3 < g_gpu) g_gpu = ms * 1e3;
          if (h1 - h0 < g_host) g_host = h1 - h0;
        }
      }
      // (b) stream order
      for (int rep = 0; rep < 12; ++rep) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const double h0 = now_us();
        for (int k = 0; k < N; ++k)
          hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, st, buf, ticks, n);
        const double h1 = now_us();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) {
          if (ms * 1e3 < s_gpu) s_gpu = ms * 1e3;
          if (h1 - h0 < s_host) s_host = h1 - h0;
        }
      }
      printf("%4d %6d | %12.1f %12.1f | %12.1f %12.1f | %6.2f us\n", N, T, g_gpu, g_host, s_gpu,
             s_host, (g_gpu - s_gpu) / N);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  }
  // (c) the same chain issued while the GPU is BUSY (the stream already holds work): is the
  // stream-order chain limited by the host's issue rate?
  {
    const int N = 12;
    const long long ticks = 500;
    CK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, st, buf, (long long)20000, n);  // 200 us
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < N; ++k)
      hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, st, buf, ticks, n);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("12 x 5 us in stream order, queued behind 200 us of work (host far ahead): %.1f us\n",
           ms * 1e3);
  }
  return 0;
}
