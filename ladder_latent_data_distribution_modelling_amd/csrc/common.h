// Shared helpers for the gfx950 LaDDer kernels (internal; the public surface is include/ladder_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ladder_hip.h"

#define LADDER_CHECK_LAUNCH()                                   \
  do {                                                          \
    if (hipGetLastError() != hipSuccess) return LADDER_E_LAUNCH; \
  } while (0)

static inline bool ladder_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float ladder_act_fn(float v, int act) {
  switch (act) {
    case LADDER_ACT_LEAKY: return v > 0.f ? v : 0.2f * v;
    case LADDER_ACT_RELU: return v > 0.f ? v : 0.f;
    case LADDER_ACT_TANH: return tanhf(v);
    default: return v;
  }
}
// derivative expressed through the activation OUTPUT y (sign-preserving for leaky/relu)
__device__ __forceinline__ float ladder_act_grad_from_out(float y, int act) {
  switch (act) {
    case LADDER_ACT_LEAKY: return y > 0.f ? 1.f : 0.2f;
    case LADDER_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case LADDER_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// 64-lane wavefront reductions (DPP/shuffle based)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Bijective XCD-aware remap: hardware places block b on XCD b%8; give every XCD a contiguous
// run of tile ids so that neighbouring tiles (shared operand panels) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// ---- absolute-maximum records ---------------------------------------------------------------------------------------------------
// A tensor's max |x| travels as a RECORD of 16 slots, 128 bytes apart (LADDER_ABSMAX_FLOATS floats): every producing workgroup
// folds its block maximum into slot (block id % 16) with one atomic max on the bit pattern (non-negative floats order like their
// bit patterns: exact, order-independent), consumers take the maximum of the 16 slots.  One shared counter would serialise all
// workgroups of a launch in the L2 (~12 ns per same-address atomic: 100 us for 8192 workgroups); 16 lines keep it under 2 us.
constexpr int AMAX_SLOTS = 16, AMAX_STRIDE = 32;
static_assert(AMAX_SLOTS * AMAX_STRIDE == LADDER_ABSMAX_FLOATS, "record size of include/ladder_hip.h");

__device__ __forceinline__ float amax_load(const float* rec) {
  float m = 0.f;
#pragma unroll
  for (int s = 0; s < AMAX_SLOTS; ++s) m = fmaxf(m, rec[s * AMAX_STRIDE]);
  return m;
}

// Clears a record from inside a kernel that runs BEFORE the record's producer on the same stream (the finalize kernels of the norm
// layers): block 0 writes the LADDER_ABSMAX_FLOATS zeros, which saves the separate memset launch per record.
__device__ __forceinline__ void amax_clear_by_block0(float* __restrict__ rec) {
  if (rec != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < AMAX_SLOTS * AMAX_STRIDE; i += blockDim.x) rec[i] = 0.f;
}
// Block-wide: EVERY thread of the workgroup must call it (contains a barrier); `m` = the thread's running max of |values written|.
__device__ __forceinline__ void amax_commit_block(float m, float* rec) {
  __shared__ float amax_red[16];
  m = wave_max(m);
  const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  if ((tid & 63) == 0) amax_red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    const int nw = (blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
    float b = 0.f;
    for (int w = 0; w < nw; ++w) b = fmaxf(b, amax_red[w]);
    const unsigned slot = (blockIdx.x + 7u * blockIdx.y + 3u * blockIdx.z) % AMAX_SLOTS;
    atomicMax(reinterpret_cast<unsigned*>(rec) + slot * AMAX_STRIDE, __builtin_bit_cast(unsigned, b));
  }
}
