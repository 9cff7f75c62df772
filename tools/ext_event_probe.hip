// Probe: do EXTERNAL event nodes work in captured HIP graphs on this ROCm, and what do they cost?
//   hipStreamWaitEvent(capturing stream, ev, hipEventWaitExternal)     -> event-wait node
//   hipEventRecordWithFlags(ev, capturing stream, hipEventRecordExternal) -> event-record node
// They would let (a) the optimizer launch become the last node of the gradient graph (its one
// dependency outside the graph is the collect lane's event) and (b) a lane start work at a chosen
// node INSIDE a running graph.  Every kernel stamps wall_clock64 (100 MHz) at start and end.
//   hipcc --offload-arch=gfx950 -O2 tools/ext_event_probe.hip -o tools/_bin/ext_event_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void spin(long long* out, int slot, long long ticks) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2 * slot] = t0;
  while (wall_clock64() - t0 < ticks) {
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2 * slot + 1] = wall_clock64();
}

static double med(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  long long* d;
  CK(hipMalloc(&d, 64 * sizeof(long long)));
  long long h[64];
  hipStream_t s1, s2, cap;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  CK(hipStreamCreate(&cap));
  hipEvent_t ev_in, ev_out;
  CK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
  const long long US = 100;   // ticks per microsecond
  const int REPS = 30;

  // --- graphs -------------------------------------------------------------------------------
  hipGraph_t g;
  hipGraphExec_t gA, gAB, gWait, gRec;
  // gA: {A 30us}
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  spin<<<1, 64, 0, cap>>>(d, 0, 30 * US);
  CK(hipStreamEndCapture(cap, &g));
  CK(hipGraphInstantiate(&gA, g, nullptr, nullptr, 0));
  // gAB: {A 30us -> B 5us}
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  spin<<<1, 64, 0, cap>>>(d, 0, 30 * US);
  spin<<<1, 64, 0, cap>>>(d, 1, 5 * US);
  CK(hipStreamEndCapture(cap, &g));
  CK(hipGraphInstantiate(&gAB, g, nullptr, nullptr, 0));
  // gWait: {A 30us -> wait(ev_in, external) -> B 5us}
  bool wait_ok = true, rec_ok = true;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  spin<<<1, 64, 0, cap>>>(d, 0, 30 * US);
  {
    hipError_t e = hipStreamWaitEvent(cap, ev_in, hipEventWaitExternal);
    if (e != hipSuccess) {
      printf("external WAIT in capture: %s\n", hipGetErrorString(e));
      wait_ok = false;
    }
  }
  spin<<<1, 64, 0, cap>>>(d, 1, 5 * US);
  {
    hipError_t e = hipStreamEndCapture(cap, &g);
    if (e != hipSuccess) {
      printf("end capture (wait graph): %s\n", hipGetErrorString(e));
      wait_ok = false;
    } else {
      size_t n = 0;
      hipGraphGetNodes(g, nullptr, &n);
      printf("wait graph: %zu nodes (3 = kernel, event wait, kernel)\n", n);
      CK(hipGraphInstantiate(&gWait, g, nullptr, nullptr, 0));
    }
  }
  // gRec: {A 30us -> record(ev_out, external) -> B 60us}
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  spin<<<1, 64, 0, cap>>>(d, 0, 30 * US);
  {
    hipError_t e = hipEventRecordWithFlags(ev_out, cap, hipEventRecordExternal);
    if (e != hipSuccess) {
      printf("external RECORD in capture: %s\n", hipGetErrorString(e));
      rec_ok = false;
    }
  }
  spin<<<1, 64, 0, cap>>>(d, 1, 60 * US);
  {
    hipError_t e = hipStreamEndCapture(cap, &g);
    if (e != hipSuccess) {
      printf("end capture (record graph): %s\n", hipGetErrorString(e));
      rec_ok = false;
    } else {
      size_t n = 0;
      hipGraphGetNodes(g, nullptr, &n);
      printf("record graph: %zu nodes (3 = kernel, event record, kernel)\n", n);
      CK(hipGraphInstantiate(&gRec, g, nullptr, nullptr, 0));
    }
  }
  CK(hipDeviceSynchronize());

  auto us = [&](int a, int b) { return (double)(h[a] - h[b]) / US; };
  std::vector<double> v1, v2, v3, v3c, v4, v4b;

  // 1. graph {A} then an eager kernel on the same stream
  for (int r = 0; r < REPS; ++r) {
    CK(hipGraphLaunch(gA, s1));
    spin<<<1, 64, 0, s1>>>(d, 1, 5 * US);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    v1.push_back(us(2, 1));
  }
  printf("1. graph{A} ; eager B (same stream):        B.start - A.end = %6.1f us\n", med(v1));
  // 2. A -> B inside one graph
  for (int r = 0; r < REPS; ++r) {
    CK(hipGraphLaunch(gAB, s1));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    v2.push_back(us(2, 1));
  }
  printf("2. graph{A -> B}:                            B.start - A.end = %6.1f us\n", med(v2));
  // 3. external wait: C (100 us) on s2, record ev_in, then the graph on s1: B must start after C
  if (wait_ok) {
    for (int r = 0; r < REPS; ++r) {
      spin<<<1, 64, 0, s2>>>(d, 2, 100 * US);
      CK(hipEventRecord(ev_in, s2));
      CK(hipGraphLaunch(gWait, s1));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      v3.push_back(us(2, 5));    // B.start - C.end  (>= 0 if the wait node works)
      // and with the event long complete: the cost of the node on the chain
      CK(hipEventRecord(ev_in, s2));
      CK(hipStreamSynchronize(s2));
      CK(hipGraphLaunch(gWait, s1));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      v3c.push_back(us(2, 1));   // B.start - A.end
    }
    printf("3. graph{A -> wait(ext) -> B}, C=100us on s2: B.start - C.end = %6.1f us (must be >= 0)\n",
           med(v3));
    printf("   same graph, event already complete:       B.start - A.end = %6.1f us\n", med(v3c));
  }
  // 4. external record inside the graph; s2 waits for it (wait issued AFTER the graph launch)
  if (rec_ok) {
    for (int r = 0; r < REPS; ++r) {
      CK(hipGraphLaunch(gRec, s1));
      CK(hipStreamWaitEvent(s2, ev_out, 0));
      spin<<<1, 64, 0, s2>>>(d, 3, 5 * US);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      v4.push_back(us(6, 1));    // D.start - A.end : >= 0 if s2 really waited for the node
      v4b.push_back(us(3, 6));   // B.end - D.start : > 0 if D ran while the graph was still busy
    }
    printf("4. graph{A -> record(ext) -> B 60us}; s2 waits: D.start - A.end = %6.1f us (>= 0: waited),"
           " B.end - D.start = %6.1f us (> 0: mid-graph)\n", med(v4), med(v4b));
  }
  return 0;
}
