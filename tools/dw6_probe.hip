// Standalone check + timing of the per-frame bf16x6 conv weight gradient (csrc/conv_dw_frame_x6.hip)
// on the DQN shapes, against a float64 reference on the host: max relative error with random
// operands, bit-exactness with small-integer operands, microseconds per call (kernel + reduce).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iagents_amd/csrc \
//         tools/dw6_probe.hip -o tools/_bin/dw6_probe
#define AA_DW6_DEBUG 1
#include "../agents_amd/csrc/conv_dw_frame_x6.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static void run(const char* name, int n, int H, int W, int Cin, int K, int s, int Cout, bool ints) {
  const int OH = (H - K) / s + 1, OW = (W - K) / s + 1;
  const size_t nx = (size_t)n * H * W * Cin, nz = (size_t)n * OH * OW * Cout;
  const size_t nw = (size_t)K * K * Cin * Cout;
  std::vector<float> x(nx), dz(nz);
  srand(7);
  for (auto& v : x) v = ints ? (float)(rand() % 7 - 3) : (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : dz) v = ints ? (float)(rand() % 5 - 2) : (float)rand() / RAND_MAX * 2.f - 1.f;
  if (!ints) for (size_t i = 0; i < nx; i += 3) x[i] = x[i] > 0 ? x[i] : 0.f;   // relu-like zeros
  std::vector<double> ref(nw, 0.0), rdb(Cout, 0.0);
  for (int b = 0; b < n; ++b)
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox) {
        const float* z = &dz[(((size_t)b * OH + oy) * OW + ox) * Cout];
        for (int co = 0; co < Cout; ++co) rdb[co] += z[co];
        for (int ky = 0; ky < K; ++ky)
          for (int kx = 0; kx < K; ++kx) {
            const float* xp = &x[(((size_t)b * H + oy * s + ky) * W + ox * s + kx) * Cin];
            double* r = &ref[((size_t)(ky * K + kx) * Cin) * Cout];
            for (int ci = 0; ci < Cin; ++ci) {
              const double xv = xp[ci];
              if (xv == 0.0) continue;
              for (int co = 0; co < Cout; ++co) r[(size_t)ci * Cout + co] += xv * z[co];
            }
          }
      }
  float *dx, *ddz, *dw, *db;
  hipMalloc(&dx, nx * 4); hipMalloc(&ddz, nz * 4); hipMalloc(&dw, nw * 4); hipMalloc(&db, Cout * 4);
  hipMemcpy(dx, x.data(), nx * 4, hipMemcpyHostToDevice);
  hipMemcpy(ddz, dz.data(), nz * 4, hipMemcpyHostToDevice);
  aa_conv_dx_desc d = {};
  d.dz = ddz; d.n_img = n; d.H = H; d.W = W; d.Cin = Cin; d.KH = K; d.KW = K; d.stride = s;
  d.Cout = Cout;
  const int64_t wsb = aa_conv_dw_frame_x6_workspace_bytes(&d);
  if (wsb <= 0) { printf("%s: shape refused\n", name); return; }
  void* ws;
  hipMalloc(&ws, wsb);
  hipMemset(dw, 0xff, nw * 4);
  int rc = aa_conv_dw_frame_x6(&d, dx, dw, db, ws, wsb, nullptr);
  hipDeviceSynchronize();
  std::vector<float> got(nw), gdb(Cout);
  hipMemcpy(got.data(), dw, nw * 4, hipMemcpyDeviceToHost);
  hipMemcpy(gdb.data(), db, Cout * 4, hipMemcpyDeviceToHost);
  double maxref = 0, maxerr = 0, dberr = 0;
  size_t exact_bad = 0;
  for (size_t i = 0; i < nw; ++i) {
    maxref = std::fmax(maxref, std::fabs(ref[i]));
    maxerr = std::fmax(maxerr, std::fabs((double)got[i] - ref[i]));
    exact_bad += (double)got[i] != ref[i];
  }
  for (int c = 0; c < Cout; ++c) dberr = std::fmax(dberr, std::fabs((double)gdb[c] - rdb[c]));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) aa_conv_dw_frame_x6(&d, dx, dw, db, ws, wsb, nullptr);
  hipEventRecord(e0, nullptr);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) aa_conv_dw_frame_x6(&d, dx, dw, db, ws, wsb, nullptr);
  hipEventRecord(e1, nullptr);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%s%s: rc %d  max|err| / max|ref| = %.3g (max|ref| %.3g)  db err %.3g  inexact %zu / %zu  "
         "%.1f us per call (kernel + reduce)\n", name, ints ? " [ints]" : "", rc, maxerr / maxref,
         maxref, dberr, exact_bad, nw, ms * 1e3 / reps);
  hipFree(dx); hipFree(ddz); hipFree(dw); hipFree(db); hipFree(ws);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  if (argc > 2 && argv[2][0] == 'p') {   // LDS pitch sweep: pad bytes per pixel row (x, dZ)
    for (int xp : {-1, 0, 16, 32, 48})
      for (int zp : {-1}) {
        g_dw6_xpad = xp; g_dw6_zpad = zp; g_dw6_dbg = 0;
        printf("xpad %d zpad %d: ", xp, zp);
        run("conv2.dW", n, 20, 20, 32, 4, 2, 64, true);
        printf("xpad %d zpad %d: ", xp, zp);
        run("conv3.dW", n, 9, 9, 64, 3, 1, 64, true);
      }
    for (int zp : {0, 16, 32, 48}) {
      g_dw6_xpad = -1; g_dw6_zpad = zp;
      printf("xpad auto zpad %d: ", zp);
      run("conv2.dW", n, 20, 20, 32, 4, 2, 64, true);
      printf("xpad auto zpad %d: ", zp);
      run("conv3.dW", n, 9, 9, 64, 3, 1, 64, true);
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'l') {   // round 5: image-row pad of the x planes / dZ pair pad, on / off
    for (int v : {0, 1, 2, 3, 0, 1, 2, 3}) {
      g_dw6_xrowpad = (v & 1) ? 0 : -1;      // 0: rows at W x pitch (rounds 2-4)
      g_dw6_zpad = (v & 2) ? 64 : -1;        // 64: a pair = 2 x 160 bytes (the old per-pixel pitch)
      printf("x rows %s, dZ pair pad %s: ", (v & 1) ? "unpadded" : "padded",
             (v & 2) ? "64" : "32");
      run("conv2.dW", n, 20, 20, 32, 4, 2, 64, true);
      printf("x rows %s, dZ pair pad %s: ", (v & 1) ? "unpadded" : "padded",
             (v & 2) ? "64" : "32");
      run("conv3.dW", n, 9, 9, 64, 3, 1, 64, true);
    }
    return 0;
  }
  if (argc > 2) {   // ablation timings (results are wrong by construction)
    for (int m : {0, 1, 2, 3, 4, 7}) {
      g_dw6_dbg = m;
      printf("dbg=%d (1 = no staging, 2 = no multiply, 4 = no slab store): ", m);
      run("conv2.dW", n, 20, 20, 32, 4, 2, 64, true);
      printf("dbg=%d: ", m);
      run("conv3.dW", n, 9, 9, 64, 3, 1, 64, true);
    }
    return 0;
  }
  run("conv2.dW 20x20x32 k4 s2", n, 20, 20, 32, 4, 2, 64, true);
  run("conv3.dW 9x9x64 k3 s1", n, 9, 9, 64, 3, 1, 64, true);
  run("conv2.dW 20x20x32 k4 s2", n, 20, 20, 32, 4, 2, 64, false);
  run("conv3.dW 9x9x64 k3 s1", n, 9, 9, 64, 3, 1, 64, false);
  run("odd: 11x13x16 k4 s1 n=37", 37, 11, 13, 16, 4, 1, 64, true);
  return 0;
}
