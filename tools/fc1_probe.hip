// In-kernel timeline + launch time of the fp32 MFMA GEMM (csrc/gemm_dma.h) on the fc1 shapes, for
// the default plan and forced (cfg, splits) variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iagents_amd/csrc tools/fc1_probe.hip -o tools/_bin/fc1_probe
#define AA_GD_STAMPS 1
#include "../agents_amd/csrc/gemm.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

static float *A_, *B_, *C_, *Y_, *ws_;
static long long* st_;
static const size_t N_ST = 4096 * 8;

static void run(const char* name, int M, int N, int K, int a_mode, int b_mode, bool mask,
                int cfg, int splits, bool defer) {
  aa_gemm_desc d{};
  d.A = A_; d.B = B_; d.C = C_; d.M = M; d.N = N; d.K = K;
  d.a_mode = a_mode; d.b_mode = b_mode;
  d.lda = a_mode == AA_A_ROW ? K : M;
  d.ldb = b_mode == AA_B_ROW ? N : K;
  d.ldc = N;
  d.force_cfg = cfg; d.force_splits = splits;
  if (mask) { d.mask_src = Y_; d.ldm = N; d.mask_kind = AA_ACT_RELU; }
  int sp = 0;
  auto call = [&]() {
    return defer ? aa_gemm_f32_slabs(&d, ws_, 64 << 20, &sp, nullptr)
                 : aa_gemm_f32(&d, ws_, 64 << 20, nullptr);
  };
  std::vector<long long> hs(N_ST);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st_, 0, N_ST * 8);
    int rc = call();
    hipDeviceSynchronize();
    if (rc != 0) { printf("%s cfg %d splits %d: rc %d\n", name, cfg, splits, rc); return; }
  }
  hipMemcpy(hs.data(), st_, N_ST * 8, hipMemcpyDeviceToHost);
  // launch time: 40 back-to-back launches between two events (includes the reduce unless deferred)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  long long* none = nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(d_gd_stamps), &none, sizeof(none));
  for (int i = 0; i < 5; ++i) call();
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < 40; ++i) call();
  hipEventRecord(e1, nullptr);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpyToSymbol(HIP_SYMBOL(d_gd_stamps), &st_, sizeof(st_));
  long long t0 = 1LL << 62, t1 = 0;
  int n = 0;
  for (size_t g = 0; g < 4096; ++g)
    if (hs[g * 8] != 0) { t0 = std::min(t0, hs[g * 8]); t1 = std::max(t1, hs[g * 8 + 3]); ++n; }
  printf("%s cfg %d splits %d (%s): %.2f us per call; %d workgroups, in-kernel span %.2f us\n", name,
         cfg, defer ? sp : splits, defer ? "slabs left for the consumer" : "incl. reduce",
         ms * 1000.0 / 40, n, (t1 - t0) * 0.01);
  if (n == 0) return;
  const char* seg[3] = {"first tile", "k loop", "epilogue"};
  for (int i = 0; i < 3; ++i) {
    std::vector<double> v;
    for (size_t g = 0; g < 4096; ++g)
      if (hs[g * 8] != 0) v.push_back((hs[g * 8 + i + 1] - hs[g * 8 + i]) * 0.01);
    std::sort(v.begin(), v.end());
    printf("   %-10s median %.2f p10 %.2f p90 %.2f max %.2f us\n", seg[i], v[v.size() / 2],
           v[v.size() / 10], v[v.size() * 9 / 10], v.back());
  }
  std::vector<double> sv, en;
  for (size_t g = 0; g < 4096; ++g)
    if (hs[g * 8] != 0) { sv.push_back((hs[g * 8] - t0) * 0.01); en.push_back((hs[g * 8 + 3] - t0) * 0.01); }
  std::sort(sv.begin(), sv.end()); std::sort(en.begin(), en.end());
  printf("   start median %.2f p90 %.2f max %.2f; end median %.2f max %.2f us\n", sv[sv.size() / 2],
         sv[sv.size() * 9 / 10], sv.back(), en[en.size() / 2], en.back());
}

int main() {
  const size_t na = (size_t)3136 * 512;
  hipMalloc(&A_, na * 4); hipMalloc(&B_, na * 4); hipMalloc(&C_, na * 4); hipMalloc(&Y_, na * 4);
  hipMalloc(&ws_, 64u << 20);
  std::vector<float> h(na);
  unsigned s = 1u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  hipMemcpy(A_, h.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(B_, h.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(Y_, h.data(), na * 4, hipMemcpyHostToDevice);
  hipMalloc(&st_, N_ST * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(d_gd_stamps), &st_, sizeof(st_));
  // cfg numbering = force_cfg: 1 128x64, 2 128x32, 3 64x64, 4 128x128, 5 64x32, 6 32x64
  {
    run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 0, 0, true);
    run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 3, 8, true);
    run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 6, 4, true);
    run("fc1.dX ", 256, 3136, 512, AA_A_ROW, AA_B_COL, true, 0, 0, false);
    run("fc1.dX ", 256, 3136, 512, AA_A_ROW, AA_B_COL, true, 3, 1, false);
    run("fc1.dW ", 3136, 512, 256, AA_A_COL, AA_B_ROW, false, 0, 0, false);
    run("fc1.dW ", 3136, 512, 256, AA_A_COL, AA_B_ROW, false, 3, 1, false);
    run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 3, 16, true);
    run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 1, 16, true);
    run("fc1.dX ", 256, 3136, 512, AA_A_ROW, AA_B_COL, true, 1, 1, false);
  }
  return 0;
  run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 0, 0, true);
  run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, 0, 0, false);
  const int cfgs[] = {6, 5, 3};
  const int spl[] = {4, 8, 16};
  for (int c : cfgs)
    for (int sp : spl) run("fc1.fwd", 256, 512, 3136, AA_A_ROW, AA_B_ROW, false, c, sp, true);
  run("fc1.dX ", 256, 3136, 512, AA_A_ROW, AA_B_COL, true, 0, 0, false);
  for (int c : cfgs) run("fc1.dX ", 256, 3136, 512, AA_A_ROW, AA_B_COL, true, c, 1, false);
  run("fc1.dX ", 256, 3136, 512, AA_A_ROW, AA_B_COL, true, 3, 2, false);
  run("fc1.dW ", 3136, 512, 256, AA_A_COL, AA_B_ROW, false, 0, 0, false);
  for (int c : cfgs) run("fc1.dW ", 3136, 512, 256, AA_A_COL, AA_B_ROW, false, c, 1, false);
  run("fc1.dW ", 3136, 512, 256, AA_A_COL, AA_B_ROW, false, 1, 1, false);
  return 0;
}
