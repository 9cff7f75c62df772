// In-kernel timeline of the dense bf16x6 GEMM (csrc/gemm_x6d.h) on the fc1 shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iagents_amd/csrc tools/x6d_probe.hip -o tools/_bin/x6d_probe
#define AA_X6D_STAMPS 1
#include "../agents_amd/csrc/gemm.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

static void run(const char* name, int M, int N, int K, int a_mode, int b_mode, bool mask) {
  float *A, *B, *C, *Y, *ws;
  const size_t na = (size_t)M * K, nb = (size_t)K * N, nc = (size_t)M * N;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4); hipMalloc(&Y, nc * 4);
  hipMalloc(&ws, 64u << 20);
  std::vector<float> h(std::max(na, std::max(nb, nc)));
  unsigned s = 1u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
  hipMemcpy(Y, h.data(), nc * 4, hipMemcpyHostToDevice);
  aa_gemm_desc d{};
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K;
  d.a_mode = a_mode; d.b_mode = b_mode;
  d.lda = a_mode == AA_A_ROW ? K : M;
  d.ldb = b_mode == AA_B_ROW ? N : K;
  d.ldc = N;
  if (mask) { d.mask_src = Y; d.ldm = N; d.mask_kind = AA_ACT_RELU; }
  long long* st;
  const size_t n_st = 4096 * 8;
  hipMalloc(&st, n_st * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(d_x6d_stamps), &st, sizeof(st));
  std::vector<long long> hs(n_st);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st, 0, n_st * 8);
    int rc = aa_gemm_f32(&d, ws, 64 << 20, nullptr);
    hipDeviceSynchronize();
    if (rc != 0) { printf("%s rc %d\n", name, rc); return; }
  }
  hipMemcpy(hs.data(), st, n_st * 8, hipMemcpyDeviceToHost);
  long long t0 = 1LL << 62, t1 = 0;
  int n = 0;
  for (size_t g = 0; g < 4096; ++g)
    if (hs[g * 8] != 0) { t0 = std::min(t0, hs[g * 8]); t1 = std::max(t1, hs[g * 8 + 3]); ++n; }
  printf("%s: %d workgroups, span %.2f us\n", name, n, (t1 - t0) * 0.01);
  const char* seg[3] = {"prologue", "k loop", "epilogue"};
  for (int i = 0; i < 3; ++i) {
    std::vector<double> v;
    for (size_t g = 0; g < 4096; ++g)
      if (hs[g * 8] != 0) v.push_back((hs[g * 8 + i + 1] - hs[g * 8 + i]) * 0.01);
    std::sort(v.begin(), v.end());
    printf("   %-9s median %.2f p10 %.2f p90 %.2f max %.2f us\n", seg[i], v[v.size() / 2],
           v[v.size() / 10], v[v.size() * 9 / 10], v.back());
  }
  std::vector<double> st_, en;
  for (size_t g = 0; g < 4096; ++g)
    if (hs[g * 8] != 0) { st_.push_back((hs[g * 8] - t0) * 0.01); en.push_back((hs[g * 8 + 3] - t0) * 0.01); }
  std::sort(st_.begin(), st_.end()); std::sort(en.begin(), en.end());
  printf("   start median %.2f max %.2f; end median %.2f max %.2f us\n", st_[st_.size() / 2],
         st_.back(), en[en.size() / 2], en.back());
}

int main() {
  run("fc1.fwd (A row, B row)", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false);
  run("fc1.dX  (A row, B col)", 256, 3136, 512, AA_A_ROW, AA_B_COL, true);
  run("fc1.dW  (A col, B row)", 3136, 512, 256, AA_A_COL, AA_B_ROW, false);
  return 0;
}
