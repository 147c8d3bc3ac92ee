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
