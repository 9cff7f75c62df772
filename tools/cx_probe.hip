// In-kernel timeline of the bf16x6 conv pair (csrc/conv_pair_x6.hip) on the Atari shapes: where a
// workgroup's time goes (staging+split | conv2 | conv3), per wave, from wall_clock64 stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iagents_amd/csrc tools/cx_probe.hip -o tools/_bin/cx_probe
#define AA_CX_STAMPS 1
#include "../agents_amd/csrc/conv_pair_x6.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;
  const int H = 20, W = 20, C = 32;
  std::vector<float> hx((size_t)B * H * W * C), hw1(4 * 4 * 32 * 64), hw2(3 * 3 * 64 * 64);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw1) v = rnd() * 0.1f;
  for (auto& v : hw2) v = rnd() * 0.1f;
  float *x, *w1, *w2, *y1, *y2;
  void* ws;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&w1, hw1.size() * 4); hipMalloc(&w2, hw2.size() * 4);
  hipMalloc(&y1, (size_t)B * 81 * 64 * 4); hipMalloc(&y2, (size_t)B * 49 * 64 * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice);
  aa_conv_layer_desc a{w1, nullptr, y1, 4, 4, 2, 64, AA_ACT_RELU}, b{w2, nullptr, y2, 3, 3, 1, 64, AA_ACT_RELU};
  const int64_t wsb = aa_conv_pair_x6_workspace_bytes(B, H, W, C, &a, &b);
  hipMalloc(&ws, wsb);
  const size_t n_st = (size_t)512 * 8 * 8;
  hipMalloc(&g_cx_stamps, n_st * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(d_cx_stamps), &g_cx_stamps, sizeof(long long*));
  std::vector<long long> st(n_st);
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(g_cx_stamps, 0, n_st * 8);
    int rc = aa_conv_pair_x6_forward(x, 0, B, H, W, C, &a, &b, ws, wsb, nullptr);
    hipDeviceSynchronize();
    if (rc != 0) { printf("rc %d\n", rc); return 1; }
  }
  hipMemcpy(st.data(), g_cx_stamps, n_st * 8, hipMemcpyDeviceToHost);
  const int nwg = std::min(B, 512);
  long long t0 = 1LL << 62, t1 = 0;
  for (int g = 0; g < nwg; ++g)
    for (int w = 0; w < 8; ++w) {
      const long long* p = &st[((size_t)g * 8 + w) * 8];
      if (p[0] == 0) continue;
      t0 = std::min(t0, p[0]); t1 = std::max(t1, p[5]);
    }
  printf("kernel span (first wave start -> last wave end): %.2f us\n", (t1 - t0) * 0.01);
  const char* names[5] = {"stage+split", "barrier", "conv2", "barrier", "conv3"};
  for (int i = 0; i < 5; ++i) {
    std::vector<double> d;
    for (int g = 0; g < nwg; ++g)
      for (int w = 0; w < 8; ++w) {
        const long long* p = &st[((size_t)g * 8 + w) * 8];
        if (p[0] != 0) d.push_back((p[i + 1] - p[i]) * 0.01);
      }
    std::sort(d.begin(), d.end());
    printf("  %-12s median %.2f  p10 %.2f  p90 %.2f  max %.2f us\n", names[i], d[d.size() / 2],
           d[d.size() / 10], d[d.size() * 9 / 10], d.back());
  }
  {
    const int seg[3][2] = {{2, 6}, {6, 7}, {7, 3}};
    const char* sn[3] = {"conv2 prologue", "conv2 k loop", "conv2 epilogue"};
    for (int i = 0; i < 3; ++i) {
      std::vector<double> d;
      for (int g = 0; g < nwg; ++g)
        for (int w = 0; w < 8; ++w) {
          const long long* p = &st[((size_t)g * 8 + w) * 8];
          if (p[0] != 0) d.push_back((p[seg[i][1]] - p[seg[i][0]]) * 0.01);
        }
      std::sort(d.begin(), d.end());
      printf("    %-15s median %.2f  p10 %.2f  p90 %.2f us\n", sn[i], d[d.size() / 2],
             d[d.size() / 10], d[d.size() * 9 / 10]);
    }
  }
  std::vector<double> starts, ends;
  for (int g = 0; g < nwg; ++g) {
    const long long* p = &st[((size_t)g * 8) * 8];
    if (p[0] != 0) { starts.push_back((p[0] - t0) * 0.01); ends.push_back((p[5] - t0) * 0.01); }
  }
  std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
  printf("  workgroup start: median %.2f  max %.2f us;  end: median %.2f  max %.2f us\n",
         starts[starts.size() / 2], starts.back(), ends[ends.size() / 2], ends.back());
  return 0;
}
