// What bounds a one-shot random row gather on MI355X?  (round 3, verdict item 8: get_next at
// BASELINE configs[1] moves 512 rows of 28,224 B = 14.45 MB in and 14.45 MB out and ran at 0.27
// of the HBM peak in isolation.)  The probe times variants of "copy n_rows spans of `span` bytes
// from random places of a big table into a dense batch":
//   threads per workgroup x vectors in flight per lane, rows per workgroup (a sample's two
//   consecutive rows are contiguous in the table), random vs sequential source rows (TLB),
//   read-only / write-only halves, and the per-workgroup prologue (counter read + barrier +
//   arrival atomic) the library's kernel carries.
// Every variant is captured 50x into one HIP graph; time = graph replay / 50.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_probe.hip -o tools/_bin/gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0 copy, 1 read only, 2 write only.  PRO 0 none, 1 counter read through LDS + barrier,
// 2 = 1 + arrival atomic on 8 shards + last-arriver logic.
template <int THREADS, int INFLIGHT, int MODE, int PRO>
__global__ void __launch_bounds__(THREADS)
gather(const char* __restrict__ table, char* __restrict__ out, const int64_t* __restrict__ rows,
       int64_t span, int64_t* counter, unsigned long long* arrival, u32x4* sink) {
  __shared__ int64_t s_v;
  unsigned long long ticket = 0;
  int64_t bias = 0;
  if (PRO >= 1) {
    if (threadIdx.x == 0) {
      s_v = __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (PRO >= 2)
        ticket = __hip_atomic_fetch_add(arrival + (blockIdx.x & 7) * 16, 1ull, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    bias = s_v & 0;  // the value is "used"
  }
  const int64_t r = blockIdx.x;
  const u32x4* s = reinterpret_cast<const u32x4*>(table + (rows[r] + bias) * span);
  u32x4* d = reinterpret_cast<u32x4*>(out + r * span);
  const int n = (int)(span / 16);
  u32x4 acc = {0, 0, 0, 0};
  for (int base = threadIdx.x; base < n; base += INFLIGHT * THREADS) {
    u32x4 v[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const int i = base + u * THREADS;
      if (MODE != 2) { if (i < n) v[u] = __builtin_nontemporal_load(s + i); }
      else v[u] = u32x4{(unsigned)i, 1u, 2u, 3u};
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      const int i = base + u * THREADS;
      if (i < n) {
        if (MODE != 1) d[i] = v[u];
        else acc ^= v[u];
      }
    }
  }
  if (MODE == 1 && acc.x == 0x12345u && acc.y == 0x54321u) sink[0] = acc;
  if (PRO >= 2 && threadIdx.x == 0) {
    const unsigned k = blockIdx.x & 7u;
    const unsigned long long in_shard = (gridDim.x + 7u - k) >> 3;
    if (ticket == in_shard - 1ull) {
      __hip_atomic_store(arrival + k * 16, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long p2 = __hip_atomic_fetch_add(arrival + 8 * 16, 1ull, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
      if (p2 == 7ull) {
        __hip_atomic_store(arrival + 8 * 16, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *counter += 1;
      }
    }
  }
}

struct Ctx {
  char* table; char* out; int64_t* rows_rand; int64_t* rows_seq; int64_t* rows_rand2; int64_t* rows_seq2;
  int64_t* counter; unsigned long long* arrival; u32x4* sink; hipStream_t st;
};

template <typename F>
static float time_graph(hipStream_t st, F launch, int reps = 50) {
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipStreamEndCapture(st, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(a, st)); CHECK(hipGraphLaunch(ge, st)); CHECK(hipEventRecord(b, st));
    CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms / reps);
  }
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  return best * 1e3f;
}

template <int THREADS, int INFLIGHT, int MODE, int PRO>
static void run(const Ctx& c, const char* name, bool seq, int rows_per_wg, int64_t row_bytes, int n_rows) {
  const int64_t span = row_bytes * rows_per_wg;
  const int grid = n_rows / rows_per_wg;
  const int64_t* rows = rows_per_wg == 2 ? (seq ? c.rows_seq2 : c.rows_rand2) : (seq ? c.rows_seq : c.rows_rand);
  const float us = time_graph(c.st, [&] {
    hipLaunchKernelGGL((gather<THREADS, INFLIGHT, MODE, PRO>), dim3(grid), dim3(THREADS), 0, c.st,
                       c.table, c.out, rows, span, c.counter, c.arrival, c.sink);
  });
  const double bytes = (MODE == 0 ? 2.0 : 1.0) * row_bytes * n_rows;
  printf("%-34s thr=%4d inflight=%2d rows/wg=%d %s  %7.2f us  %6.0f GB/s  frac=%.3f\n", name, THREADS,
         INFLIGHT, rows_per_wg, seq ? "seq " : "rand", us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
}

__global__ void empty_kernel() {}

int main(int argc, char** argv) {
  const int64_t row_bytes = 28224;
  const int n_rows = 512;
  const int64_t capacity = argc > 1 ? atoll(argv[1]) : 1000000;  // rows in the table (28 GB default)
  Ctx c;
  CHECK(hipStreamCreate(&c.st));
  CHECK(hipMalloc(&c.table, capacity * row_bytes));
  CHECK(hipMemset(c.table, 1, capacity * row_bytes));
  CHECK(hipMalloc(&c.out, n_rows * row_bytes));
  std::vector<int64_t> rr(n_rows), rs(n_rows), rr2(n_rows / 2), rs2(n_rows / 2);
  srand(7);
  for (int i = 0; i < n_rows / 2; ++i) {
    const int64_t id = (((int64_t)rand() << 20) ^ rand()) % (capacity - 2);
    rr[2 * i] = id; rr[2 * i + 1] = id + 1;  // a sample = two consecutive rows
    rr2[i] = id / 2;                         // spans of two rows (aligned to even ids: same bytes moved)
    rs[2 * i] = 2 * i; rs[2 * i + 1] = 2 * i + 1;
    rs2[i] = i;
  }
  auto up = [&](std::vector<int64_t>& v) {
    int64_t* d; CHECK(hipMalloc(&d, v.size() * 8));
    CHECK(hipMemcpy(d, v.data(), v.size() * 8, hipMemcpyHostToDevice)); return d; };
  c.rows_rand = up(rr); c.rows_seq = up(rs); c.rows_rand2 = up(rr2); c.rows_seq2 = up(rs2);
  CHECK(hipMalloc(&c.counter, 8)); CHECK(hipMemset(c.counter, 0, 8));
  CHECK(hipMalloc(&c.arrival, 9 * 16 * 8)); CHECK(hipMemset(c.arrival, 0, 9 * 16 * 8));
  CHECK(hipMalloc(&c.sink, 16));
  CHECK(hipDeviceSynchronize());
  printf("table %.1f GB, %d rows of %lld B per launch\n", capacity * row_bytes / 1e9, n_rows, (long long)row_bytes);
  const float e = time_graph(c.st, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, c.st); });
  printf("empty kernel in the same graph shape: %.2f us per launch\n", e);

  run<256, 8, 0, 0>(c, "copy", false, 1, row_bytes, n_rows);
  run<256, 8, 0, 0>(c, "copy", true, 1, row_bytes, n_rows);
  run<256, 8, 0, 1>(c, "copy +counter read", false, 1, row_bytes, n_rows);
  run<256, 8, 0, 2>(c, "copy +counter +arrival", false, 1, row_bytes, n_rows);
  run<256, 8, 1, 0>(c, "read only", false, 1, row_bytes, n_rows);
  run<256, 8, 1, 0>(c, "read only", true, 1, row_bytes, n_rows);
  run<256, 8, 2, 0>(c, "write only", false, 1, row_bytes, n_rows);
  run<512, 4, 0, 0>(c, "copy", false, 1, row_bytes, n_rows);
  run<1024, 2, 0, 0>(c, "copy", false, 1, row_bytes, n_rows);
  run<256, 4, 0, 0>(c, "copy (two passes)", false, 1, row_bytes, n_rows);
  run<256, 16, 0, 0>(c, "copy", false, 2, row_bytes, n_rows);
  run<512, 8, 0, 0>(c, "copy", false, 2, row_bytes, n_rows);
  run<1024, 4, 0, 0>(c, "copy", false, 2, row_bytes, n_rows);
  run<512, 8, 0, 2>(c, "copy +counter +arrival", false, 2, row_bytes, n_rows);
  run<1024, 4, 0, 2>(c, "copy +counter +arrival", false, 2, row_bytes, n_rows);
  run<512, 8, 1, 0>(c, "read only", false, 2, row_bytes, n_rows);
  run<1024, 4, 1, 0>(c, "read only", false, 2, row_bytes, n_rows);
  return 0;
}
