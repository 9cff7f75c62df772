// Fused forward / backward of small MLPs (every Dense layer <= 64 wide, <= 4 layers): the PPO
// actor and value networks (64, 64; tf_agents/agents/ppo/ppo_actor_network.py:42-113,
// networks/value_network.py), the CartPole Q-network (100 is too wide; 64-unit variants fit).
//
// At these sizes one Dense layer is 0.5 MFLOP per sample tile: an MFMA GEMM launch per layer (plus
// split-K reduce, plus a column-sum for the bias gradient) is ~10 launches of pure latency per
// network and direction -- 45 launches per PPO minibatch step.  Here a workgroup owns 64 samples and
// walks ALL layers with the activations in LDS and the layer's weights staged in LDS:
//   forward : 256 threads = 64 samples x 4 column quarters; thread (s, q) computes outputs
//             16q..16q+15 of sample s: for k: h_in[s][k] (one LDS read) times W[k][16q..] (four
//             broadcast-friendly ds_read_b128) -> 16 FMAs.  Outputs go to LDS for the next layer and
//             to the per-layer activation buffers in HBM (the backward pass and the heads read them).
//   backward: per layer, (i) dW partial = H_in^T G over the tile's 64 samples as a 64x64 block
//             product (each thread a 4x4 patch, operands by ds_read_b128), written to this
//             workgroup's slab; (ii) bias partial = column sums of G; (iii) G_in = (G W^T) * act'(h_in).
//   reduce  : slabs summed over workgroups in index order (deterministic) into the flat gradient
//             buffer.
// Plain fp32 FMAs on the VALU: no MFMA tile shape fits a 17-wide input or a 6-wide head without
// mostly multiplying zeros, and at 64 samples x 64 x 64 per layer the VALU is not the limit.
#include "common.h"
#include "agents_amd.h"

#include "mlp_small_common.h"

struct AaMlpOut {
  float* y[AA_MLP_MAX_LAYERS];
};

// TILE samples per workgroup, QN = 256 / TILE threads per sample, OPT = 64 / QN outputs per thread
// (TILE 16 / 32 / 64 -> 4 / 8 / 16 outputs): small batches use small tiles so that every CU gets a
// workgroup (a 4,096-sample PPO minibatch on 64-sample tiles kept 192 of 256 CUs idle).
template <int TILE>
__global__ void __launch_bounds__(256)
aa_mlp_small_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ params,
                        AaMlpDesc d, int64_t B, AaMlpOut out) {
  constexpr int QN = 256 / TILE, OPT = AA_MLP_MAXW / QN;
  __shared__ __attribute__((aligned(16))) float Ws[AA_MLP_MAXW * AA_MLP_MAXW];
  __shared__ __attribute__((aligned(16))) float bs[AA_MLP_MAXW];
  __shared__ __attribute__((aligned(16))) float h[2][TILE][AA_MLP_MAXW + 4];
  const int s = threadIdx.x / QN, q = threadIdx.x % QN;
  const int64_t b = (int64_t)blockIdx.x * TILE + s;
  // input tile -> h[0]
  for (int i = threadIdx.x; i < TILE * AA_MLP_MAXW; i += blockDim.x) {
    const int ss = i >> 6, k = i & 63;
    const int64_t bb = (int64_t)blockIdx.x * TILE + ss;
    h[0][ss][k] = (bb < B && k < d.dims[0]) ? x[bb * ldx + k] : 0.f;
  }
  int cur = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    const int n_in = d.dims[l], n_out = d.dims[l + 1];
    __syncthreads();   // previous layer's readers of Ws / writers of h[cur] are done
    aa_mlp_stage_w(params, d.k_off[l], d.b_off[l], n_in, n_out, Ws, bs);
    __syncthreads();
    float acc[OPT];
#pragma unroll
    for (int j = 0; j < OPT; ++j) acc[j] = bs[OPT * q + j];
    for (int k = 0; k < n_in; ++k) {
      const float hk = h[cur][s][k];
      const float4* wr = reinterpret_cast<const float4*>(Ws + k * AA_MLP_MAXW + OPT * q);
#pragma unroll
      for (int v = 0; v < OPT / 4; ++v) {
        const float4 w = wr[v];
        acc[4 * v + 0] = fmaf(hk, w.x, acc[4 * v + 0]);
        acc[4 * v + 1] = fmaf(hk, w.y, acc[4 * v + 1]);
        acc[4 * v + 2] = fmaf(hk, w.z, acc[4 * v + 2]);
        acc[4 * v + 3] = fmaf(hk, w.w, acc[4 * v + 3]);
      }
    }
    float* yo = out.y[l];
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
      const int col = OPT * q + j;
      const float v = aa_mlp_act(acc[j], d.acts[l]);
      h[cur ^ 1][s][col] = col < n_out ? v : 0.f;
      if (b < B && col < n_out) yo[b * n_out + col] = v;
    }
    cur ^= 1;
  }
}

// Backward.  dy = d loss / d (last layer's OUTPUT, after its activation) [B, n_L].
// slabs: [gridDim.x][total_params] partial gradients (same layout as the flat parameter buffer).
template <int TILE>
__global__ void __launch_bounds__(256)
aa_mlp_small_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ params,
                        AaMlpDesc d, int64_t B, AaMlpOut act, const float* __restrict__ dy,
                        float* __restrict__ slabs, int64_t total_params,
                        float* __restrict__ dx_out /* nullable [B, n0] */) {
  __shared__ __attribute__((aligned(16))) float Ws[AA_MLP_MAXW * AA_MLP_MAXW];
  __shared__ __attribute__((aligned(16))) float bs[AA_MLP_MAXW];
  constexpr int QN = 256 / TILE, OPT = AA_MLP_MAXW / QN;
  __shared__ __attribute__((aligned(16))) float G[TILE][AA_MLP_MAXW + 4];   // [s][j]
  __shared__ __attribute__((aligned(16))) float H[TILE][AA_MLP_MAXW + 4];   // [s][k]
  const int s = threadIdx.x / QN, q = threadIdx.x % QN;
  const int64_t b0 = (int64_t)blockIdx.x * TILE;
  float* slab = slabs + (int64_t)blockIdx.x * total_params;
  const int L = d.n_layers;
  // G <- dy * act'(y_L)
  {
    const int n_out = d.dims[L];
    for (int i = threadIdx.x; i < TILE * AA_MLP_MAXW; i += blockDim.x) {
      const int ss = i >> 6, j = i & 63;
      const int64_t bb = b0 + ss;
      float g = 0.f;
      if (bb < B && j < n_out) {
        g = dy[bb * n_out + j];
        if (d.acts[L - 1] != AA_ACT_NONE)
          g *= aa_mlp_actgrad(act.y[L - 1][bb * n_out + j], d.acts[L - 1]);
      }
      G[ss][j] = g;
    }
  }
  for (int l = L - 1; l >= 0; --l) {
    const int n_in = d.dims[l], n_out = d.dims[l + 1];
    // H <- input activations of layer l (x for l == 0)
    const float* hin = l == 0 ? x : act.y[l - 1];
    const int64_t ldh = l == 0 ? ldx : n_in;
    __syncthreads();   // G complete; previous users of H / Ws done
    for (int i = threadIdx.x; i < TILE * AA_MLP_MAXW; i += blockDim.x) {
      const int ss = i >> 6, k = i & 63;
      const int64_t bb = b0 + ss;
      H[ss][k] = (bb < B && k < n_in) ? hin[bb * ldh + k] : 0.f;
    }
    aa_mlp_stage_w(params, d.k_off[l], d.b_off[l], n_in, n_out, Ws, bs);
    __syncthreads();
    // (i) dW[k][j] = sum_s H[s][k] G[s][j]: thread -> 4x4 patch (k0.., j0..)
    {
      const int k0 = (threadIdx.x >> 4) * 4, j0 = (threadIdx.x & 15) * 4;
      float a[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) a[u][v] = 0.f;
      for (int ss = 0; ss < TILE; ++ss) {
        const float4 hv = *reinterpret_cast<const float4*>(&H[ss][k0]);
        const float4 gv = *reinterpret_cast<const float4*>(&G[ss][j0]);
        const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) a[u][v] = fmaf(hh[u], gg[v], a[u][v]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (k0 + u < n_in && j0 + v < n_out)
            slab[d.k_off[l] + (int64_t)(k0 + u) * n_out + j0 + v] = a[u][v];
    }
    // (ii) db[j] = sum_s G[s][j]
    if (threadIdx.x < n_out) {
      float sum = 0.f;
      for (int ss = 0; ss < TILE; ++ss) sum += G[ss][threadIdx.x];
      slab[d.b_off[l] + threadIdx.x] = sum;
    }
    // (iii) G_in[s][k] = (sum_j G[s][j] W[k][j]) * act'(h_in[s][k]) for k in the thread's quarter
    if (l > 0 || dx_out != nullptr) {
      float gin[OPT];
#pragma unroll
      for (int kk = 0; kk < OPT; ++kk) gin[kk] = 0.f;
      for (int j = 0; j < n_out; j += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(&G[s][j]);
#pragma unroll
        for (int kk = 0; kk < OPT; ++kk) {
          const float4 w = *reinterpret_cast<const float4*>(Ws + (OPT * q + kk) * AA_MLP_MAXW + j);
          gin[kk] = fmaf(gv.x, w.x, gin[kk]);
          gin[kk] = fmaf(gv.y, w.y, gin[kk]);
          gin[kk] = fmaf(gv.z, w.z, gin[kk]);
          gin[kk] = fmaf(gv.w, w.w, gin[kk]);
        }
      }
      __syncthreads();   // everybody is done reading G (dW, db and this product)
      if (l > 0) {
#pragma unroll
        for (int kk = 0; kk < OPT; ++kk) {
          const int k = OPT * q + kk;
          G[s][k] = k < n_in ? gin[kk] * aa_mlp_actgrad(H[s][k], d.acts[l - 1]) : 0.f;
        }
      } else {
        const int64_t bb = b0 + s;
#pragma unroll
        for (int kk = 0; kk < OPT; ++kk) {
          const int k = OPT * q + kk;
          if (bb < B && k < n_in) dx_out[bb * n_in + k] = gin[kk];
        }
      }
    }
  }
}

// grads[i] = sum over slabs: 16 "z-lanes" per parameter each sum slabs zl, zl+16, ... in that order
// (four loads in flight) and the 16 partials are added in lane order through LDS -- a fixed
// association for a given slab count, so the result is reproducible (a single thread walking 256
// slabs took 21 us for a 10 K-parameter network).
__global__ void __launch_bounds__(256)
aa_mlp_slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int64_t total,
                          float* __restrict__ grads, AaMlpDesc d) {
  __shared__ float part[16][16];
  const int it = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + it;
  // the alignment padding between the layers' segments is never written by the backward kernel:
  // its gradient is zero by definition and the (uninitialised) slab entries are not read -- this
  // replaces a memset of all slabs per call (two 5 us fill launches per PPO train step)
  bool live = false;
  for (int l = 0; l < d.n_layers; ++l) {
    const int64_t nk = (int64_t)d.dims[l] * d.dims[l + 1];
    live = live || (i >= d.k_off[l] && i < d.k_off[l] + nk) ||
           (i >= d.b_off[l] && i < d.b_off[l] + d.dims[l + 1]);
  }
  float v = 0.f;
  if (i < total && live) {
    int z = zl;
    for (; z + 48 < n_slabs; z += 64) {
      const float t0 = slabs[(int64_t)z * total + i], t1 = slabs[(int64_t)(z + 16) * total + i];
      const float t2 = slabs[(int64_t)(z + 32) * total + i];
      const float t3 = slabs[(int64_t)(z + 48) * total + i];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; z < n_slabs; z += 16) v += slabs[(int64_t)z * total + i];
  }
  part[zl][it] = v;
  __syncthreads();
  if (zl == 0 && i < total) {
    float r = part[0][it];
#pragma unroll
    for (int j = 1; j < 16; ++j) r += part[j][it];
    grads[i] = r;
  }
}

// samples per workgroup: the smallest tile that still gives every CU a workgroup
extern "C" {

int aa_mlp_small_forward(const float* x, int64_t ldx, const float* params, int32_t n_layers,
                         const int32_t* dims, const int32_t* acts, const int64_t* k_off,
                         const int64_t* b_off, int64_t B, float* const* y_out_h, void* stream) {
  if (!x || !params || !y_out_h || B <= 0) return AA_ERR_INVALID;
  AaMlpDesc d;
  int rc = aa_mlp_fill(d, n_layers, dims, acts, k_off, b_off);
  if (rc != AA_OK) return rc;
  if (ldx < dims[0]) return AA_ERR_INVALID;
  AaMlpOut out;
  for (int i = 0; i < n_layers; ++i) {
    if (!y_out_h[i]) return AA_ERR_INVALID;
    out.y[i] = y_out_h[i];
  }
  const int tile = aa_mlp_tile(B);
  const int64_t grid = (B + tile - 1) / tile;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  if (tile == 16)
    hipLaunchKernelGGL(aa_mlp_small_fwd_kernel<16>, dim3((unsigned)grid), dim3(256), 0, st, x, ldx,
                       params, d, B, out);
  else if (tile == 32)
    hipLaunchKernelGGL(aa_mlp_small_fwd_kernel<32>, dim3((unsigned)grid), dim3(256), 0, st, x, ldx,
                       params, d, B, out);
  else
    hipLaunchKernelGGL(aa_mlp_small_fwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, st, x, ldx,
                       params, d, B, out);
  return aa_launch_status();
}

int64_t aa_mlp_small_workspace_bytes(int64_t B, int64_t total_params) {
  if (B <= 0 || total_params <= 0) return -1;
  const int tile = aa_mlp_tile(B);
  return ((B + tile - 1) / tile) * total_params * (int64_t)sizeof(float);
}

int aa_mlp_small_backward(const float* x, int64_t ldx, const float* params, int32_t n_layers,
                          const int32_t* dims, const int32_t* acts, const int64_t* k_off,
                          const int64_t* b_off, int64_t B, float* const* y_h, const float* dy,
                          float* grads, int64_t total_params, float* dx_out, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  if (!x || !params || !y_h || !dy || !grads || !workspace || B <= 0 || total_params <= 0)
    return AA_ERR_INVALID;
  AaMlpDesc d;
  int rc = aa_mlp_fill(d, n_layers, dims, acts, k_off, b_off);
  if (rc != AA_OK) return rc;
  if (ldx < dims[0]) return AA_ERR_INVALID;
  AaMlpOut act;
  for (int i = 0; i < n_layers; ++i) {
    if (!y_h[i]) return AA_ERR_INVALID;
    act.y[i] = y_h[i];
  }
  const int tile = aa_mlp_tile(B);
  const int64_t grid = (B + tile - 1) / tile;
  if (grid > 0x7fffffffLL) return AA_ERR_RANGE;
  if (workspace_bytes < grid * total_params * (int64_t)sizeof(float)) return AA_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  if (tile == 16)
    hipLaunchKernelGGL(aa_mlp_small_bwd_kernel<16>, dim3((unsigned)grid), dim3(256), 0, st, x, ldx,
                       params, d, B, act, dy, (float*)workspace, total_params, dx_out);
  else if (tile == 32)
    hipLaunchKernelGGL(aa_mlp_small_bwd_kernel<32>, dim3((unsigned)grid), dim3(256), 0, st, x, ldx,
                       params, d, B, act, dy, (float*)workspace, total_params, dx_out);
  else
    hipLaunchKernelGGL(aa_mlp_small_bwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, st, x, ldx,
                       params, d, B, act, dy, (float*)workspace, total_params, dx_out);
  hipLaunchKernelGGL(aa_mlp_slab_reduce_kernel, dim3((unsigned)((total_params + 15) / 16)),
                     dim3(256), 0, st, (const float*)workspace, (int)grid, total_params, grads, d);
  return aa_launch_status();
}

}  // extern "C"
