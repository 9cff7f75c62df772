// Small HBM-bound helpers around the GEMM: bias gradients (column sums), col2im for the
// Conv2D input gradient, device counters.
//   keras Dense/Conv2D backward under tf.GradientTape (agents/dqn/dqn_agent.py:412-426).
#include "common.h"
#include "agents_amd.h"

__device__ static inline float aa_actgrad2(float y, int kind) {
  if (kind == AA_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (kind == AA_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

__global__ void aa_counter_add_kernel(int64_t* c, int64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

// One-thread no-op: a named dispatch that profilers show in launch order (bench.py brackets its
// timed region and every per-kernel case with it, then reads rocprofv3's kernel trace by position).
__global__ void aa_marker_kernel(int32_t id, int32_t* sink) {
  if (sink != nullptr && threadIdx.x == 0) *sink = id;
}

// ---- column sum, two deterministic stages ---------------------------------------------------
// stage 1: block (bx, by) sums rows [bx*rpb, (bx+1)*rpb) of columns [by*ncol, (by+1)*ncol).
// Threads are laid out (ty, tx) with tx over columns so that row reads are coalesced.
__global__ void __launch_bounds__(256)
aa_colsum_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int64_t N,
                         int64_t rows_per_block, int ncol, float* __restrict__ partial) {
  __shared__ float red[256];
  const int tx = threadIdx.x % ncol, ty = threadIdx.x / ncol, nty = 256 / ncol;
  const int64_t n = (int64_t)blockIdx.y * ncol + tx;
  const int64_t m_lo = (int64_t)blockIdx.x * rows_per_block;
  int64_t m_hi = m_lo + rows_per_block;
  if (m_hi > M) m_hi = M;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains keep loads in flight
  if (n < N) {
    int64_t m = m_lo + ty;
    for (; m + 3 * nty < m_hi; m += 4 * nty) {
      s0 += x[m * ld + n];
      s1 += x[(m + nty) * ld + n];
      s2 += x[(m + 2 * nty) * ld + n];
      s3 += x[(m + 3 * nty) * ld + n];
    }
    for (; m < m_hi; m += nty) s0 += x[m * ld + n];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && n < N) {
    float t = 0.f;
    for (int j = 0; j < nty; ++j) t += red[j * ncol + tx];
    partial[(int64_t)blockIdx.x * N + n] = t;
  }
}
// stage 2: 16 x 64 threads; thread (ty, tx) sums partial rows ty, ty+16, ... of column tx, then
// the 16 row-groups are combined in fixed order.
__global__ void __launch_bounds__(1024)
aa_colsum_final_kernel(const float* __restrict__ partial, int64_t P, int64_t N,
                       float* __restrict__ out) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  float s = 0.f;
  if (n < N)
    for (int64_t p = ty; p < P; p += 16) s += partial[p * N + n];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    float t = 0.f;
    for (int j = 0; j < 16; ++j) t += red[j][tx];
    out[n] = t;
  }
}

static void aa_colsum_plan(int64_t M, int64_t N, int64_t* P, int64_t* rpb, int* ncol) {
  int nc = 1;
  while (nc < N && nc < 256) nc <<= 1;
  *ncol = nc;
  // about 512 workgroups in total, at least 64 rows each
  const int64_t col_blocks = (N + nc - 1) / nc;
  int64_t p = 512 / col_blocks;
  if (p < 1) p = 1;
  int64_t r = (M + p - 1) / p;
  if (r < 64) r = 64;
  p = (M + r - 1) / r;
  if (p < 1) p = 1;
  *P = p;
  *rpb = r;
}

// ---- col2im ---------------------------------------------------------------------------------
// One thread per (input pixel, 4 channels).  dcol is [n_img*OH*OW, KH*KW*Cin] row-major.
__global__ void __launch_bounds__(256)
aa_col2im_kernel(const float* __restrict__ dcol, int n_img, int H, int W, int Cin, int KH, int KW,
                 int stride, int OH, int OW, float* __restrict__ dx,
                 const float* __restrict__ mask_src, int mask_kind) {
  const int c4n = Cin / 4;
  const int64_t total = (int64_t)n_img * H * W * c4n;
  const int Kp = KH * KW * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t pix = i / c4n;
    const int ix = (int)(pix % W);
    pix /= W;
    const int iy = (int)(pix % H);
    const int b = (int)(pix / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < KH; ++ky) {
      const int ty = iy - ky;
      if (ty < 0 || ty % stride != 0) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int tx = ix - kx;
        if (tx < 0 || tx % stride != 0) continue;
        const int ox = tx / stride;
        if (ox >= OW) continue;
        const int64_t row = ((int64_t)b * OH + oy) * OW + ox;
        const float4 v =
            *reinterpret_cast<const float4*>(dcol + row * Kp + (ky * KW + kx) * Cin + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    const int64_t o = (((int64_t)b * H + iy) * W + ix) * Cin + c;
    if (mask_kind != 0) {
      const float4 y = *reinterpret_cast<const float4*>(mask_src + o);
      acc.x *= aa_actgrad2(y.x, mask_kind);
      acc.y *= aa_actgrad2(y.y, mask_kind);
      acc.z *= aa_actgrad2(y.z, mask_kind);
      acc.w *= aa_actgrad2(y.w, mask_kind);
    }
    *reinterpret_cast<float4*>(dx + o) = acc;
  }
}

// dz = dy * act'(y)   (gradient through a trailing activation, expressed via its output y)
__global__ void __launch_bounds__(256)
aa_act_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y, int kind,
                       int64_t n, float* __restrict__ dz) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dz[i] = dy[i] * aa_actgrad2(y[i], kind);
}

// sumsq_out[0] = sum x^2 over n elements (keras l2 regulariser / tf.nn.l2_loss), one workgroup
__global__ void __launch_bounds__(1024)
aa_sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[i] * x[i];
  const float t = aa_block_sum(s, red);
  if (threadIdx.x == 0) out[0] = t;
}

extern "C" {

int aa_abi_version(void) { return AA_ABI_VERSION; }

int aa_act_backward(const float* dy, const float* y, int32_t act, int64_t n, float* dz,
                    void* stream) {
  if (!dy || !y || !dz || n <= 0) return AA_ERR_INVALID;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(aa_act_backward_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, dy, y, (int)act, n, dz);
  return aa_launch_status();
}

int aa_sumsq_f32(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_sumsq_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
  return aa_launch_status();
}

int aa_counter_add(int64_t* counter_dev, int64_t inc, void* stream) {
  if (counter_dev == nullptr) return AA_ERR_INVALID;
  hipLaunchKernelGGL(aa_counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     counter_dev, inc);
  return aa_launch_status();
}

int aa_marker(int32_t id, void* stream) {
  hipLaunchKernelGGL(aa_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, id,
                     (int32_t*)nullptr);
  return aa_launch_status();
}

int64_t aa_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return -1;
  int64_t P, rpb;
  int ncol;
  aa_colsum_plan(M, N, &P, &rpb, &ncol);
  return P * N * (int64_t)sizeof(float);
}

int aa_colsum_f32(const float* x, int64_t ld, int64_t M, int64_t N, float* out, void* workspace,
                  int64_t workspace_bytes, void* stream) {
  if (x == nullptr || out == nullptr || M <= 0 || N <= 0 || ld < N) return AA_ERR_INVALID;
  int64_t P, rpb;
  int ncol;
  aa_colsum_plan(M, N, &P, &rpb, &ncol);
  if (workspace == nullptr || workspace_bytes < P * N * (int64_t)sizeof(float)) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)P, (unsigned)((N + ncol - 1) / ncol));
  hipLaunchKernelGGL(aa_colsum_partial_kernel, grid, dim3(256), 0, st, x, ld, M, N, rpb, ncol,
                     (float*)workspace);
  hipLaunchKernelGGL(aa_colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, st,
                     (const float*)workspace, P, N, out);
  return aa_launch_status();
}

int aa_col2im_f32(const float* dcol, int32_t n_img, int32_t H, int32_t W, int32_t Cin, int32_t KH,
                  int32_t KW, int32_t stride, float* dx, const float* mask_src, int32_t mask_kind,
                  void* stream) {
  if (dcol == nullptr || dx == nullptr || n_img <= 0 || H <= 0 || W <= 0 || Cin <= 0 || KH <= 0 ||
      KW <= 0 || stride <= 0 || Cin % 4 != 0)
    return AA_ERR_INVALID;
  const int OH = (H - KH) / stride + 1, OW = (W - KW) / stride + 1;
  if (OH <= 0 || OW <= 0) return AA_ERR_INVALID;
  if (mask_src == nullptr) mask_kind = 0;
  const int64_t total = (int64_t)n_img * H * W * (Cin / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(aa_col2im_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     dcol, n_img, H, W, Cin, KH, KW, stride, OH, OW, dx, mask_src, mask_kind);
  return aa_launch_status();
}

}  // extern "C"
