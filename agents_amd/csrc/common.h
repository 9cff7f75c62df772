// Shared device/host helpers for the agents_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AA_OK 0
#define AA_ERR_INVALID (-22)   // EINVAL-style: bad argument
#define AA_ERR_RANGE (-34)     // ERANGE-style: size out of supported range
#define AA_ERR_LAUNCH (-5)     // EIO-style: HIP launch failure

#define AA_WAVE 64
#define AA_MAX_ARRIVAL_GROUPS 16   // see aa_advance_when_all_done

// Returns AA_ERR_LAUNCH if the preceding launch failed (does not synchronise).
static inline int aa_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AA_OK : AA_ERR_LAUNCH;
}

// ----------------------------------------------------------------------------
// Philox4x32-10 counter RNG (Salmon et al., SC'11).  This is the canonical
// random stream of this package (the reference's tf.random stream
// is unseeded and therefore unpinned).  oracle/philox.py restates it in numpy and
// is checked against the published Random123 known-answer vectors.
// ----------------------------------------------------------------------------
struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ static inline uint32_t aa_mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

__host__ __device__ static inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                        uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = aa_mulhi32(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = aa_mulhi32(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.x = c0; o.y = c1; o.z = c2; o.w = c3;
  return o;
}

// Uniform float in [0,1) from 32 random bits: top 24 bits * 2^-24 (exact in fp32).
__host__ __device__ static inline float aa_u01(uint32_t bits) {
  return (float)(bits >> 8) * (1.0f / 16777216.0f);
}

// Wave-level (64-lane) sum via DPP-free shuffles.
__device__ static inline float aa_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-level sum for blockDim.x <= 1024 (multiple of 64).  Result valid in thread 0.
// Deterministic: fixed tree order.
__device__ static inline float aa_block_sum(float v, float* smem /* >= 16 floats */) {
  v = aa_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = 0.f;
  if (threadIdx.x == 0) {
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}

// A device counter that every workgroup of a launch reads at its start (Philox call counter, env
// step counter, replay last_id) is advanced by the LAST workgroup to finish instead of by a second
// one-thread launch: *arrival counts finished workgroups (zero before the launch, reset to zero by
// the last one), so by the time the counter changes every group has consumed the old value.
__device__ static inline void aa_advance_when_all_done(int64_t* counter, int64_t* arrival,
                                                       int64_t inc, unsigned n_groups) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long prev = __hip_atomic_fetch_add(
        reinterpret_cast<unsigned long long*>(arrival), 1ull, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (unsigned long long)n_groups - 1ull) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(arrival), 0ull, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      *counter += inc;
    }
  }
}

#define AA_RB_ARRIVAL_STRIDE 16  // int64 words between arrival counters: one 128-byte line each
#define AA_RB_ARRIVAL_WORDS (9 * AA_RB_ARRIVAL_STRIDE)   // 8 shards + 1 top

// aa_advance_when_all_done (common.h) for grids of hundreds to thousands of workgroups: arrivals
// are counted on 8 words (blockIdx & 7 -- one per XCD under the usual round-robin placement), the
// last arriver of each shard reports to a ninth: no word sees more than n/8 (+8) atomics, where
// one word serialised ~2,000 of them at ~11 ns each.  The nine words sit on nine different
// 128-byte lines (atomics on one line serialise in its L2 channel whatever the word) and are zero
// between launches.
__device__ static inline void aa_advance_sharded(int64_t* counter, int64_t* arrival, int64_t inc,
                                                 unsigned n_groups) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned k = blockIdx.x & 7u;
    const unsigned long long in_shard = (n_groups + 7u - k) >> 3;
    unsigned long long* a = reinterpret_cast<unsigned long long*>(arrival);
    unsigned long long* mine = a + k * AA_RB_ARRIVAL_STRIDE;
    unsigned long long* top = a + 8 * AA_RB_ARRIVAL_STRIDE;
    const unsigned long long prev =
        __hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == in_shard - 1ull) {
      __hip_atomic_store(mine, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long n_shards = n_groups < 8u ? n_groups : 8u;
      const unsigned long long p2 =
          __hip_atomic_fetch_add(top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p2 == n_shards - 1ull) {
        __hip_atomic_store(top, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *counter += inc;
      }
    }
  }
}


// hipFuncSetAttribute (the grant of more than 64 KiB of dynamic LDS) applies to the function object
// of the CURRENT device: the launchers' "largest size granted so far" tables are kept per device
// ordinal, so that a process that drives a second GPU grants there too.
#define AA_MAX_DEVICES 16
static inline int aa_device_ordinal() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < AA_MAX_DEVICES) ? d : -1;
}
