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

// Output-channel tile of a CLASS-structured halo-kernel launch (the four 128-channel tiles of a pixel patch are parity classes with
// different tap counts: 9 / 6 / 6 / 4 for the upsample-fused forward, 4 / 2 / 2 / 1 for a stride-2 backward-data): position `c` of patch
// `mt` computes class (c + mt) % 4.  Measured in round 4 (csrc/convf32.hip, conv2d_7 upsample-fused forward at batch 128): with the classes
// in a FIXED order the launch took as long as if every tile had the longest class's taps (4535 us with tap masks, 4573 us without) --
// consecutive workgroups of an XCD are handed to its shader engines in a static round-robin (workgroup k -> engine k % 4), so each engine
// received ONE class only; uniform masks scale as expected (4 taps: 2143 us, 1 tap: 724 us).  Rotating the class with the patch index gives
// every engine all four classes: 3363 us (ideal 25 / 36: 3180).  A bijection inside each patch's four consecutive workgroups, so their
// shared input halo still meets in one L2.
__device__ __forceinline__ int class_tile(int c, int mt, int tiles_n, bool class_mode) {
  return (class_mode && tiles_n == 4) ? ((c + mt) & 3) : c;
}

// ---- absolute-maximum records ---------------------------------------------------------------------------------------------------
// A tensor's max |x| travels as a RECORD of LADDER_ABSMAX_FLOATS = 512 floats = 16 lines of 128 bytes, in one of two layouts:
//   mode 0 (rec[1] == 0)  ONE bound for the whole tensor: every producing workgroup folds its block maximum into slot (block id % 16) =
//                         float 0 of line (block id % 16) with one atomic max on the bit pattern (non-negative floats order like their
//                         bit patterns: exact, order-independent); the bound is the maximum of the 16 slots.  One shared counter would
//                         serialise all workgroups of a launch in the L2 (~12 ns per same-address atomic: 100 us for 8192 workgroups).
//   mode 1 (rec[1] != 0)  one bound PER SAMPLE (NHWC tensors; round 3): sample n folds into float 2 + (n / 16) % 30 of line n % 16
//                         (480 distinct samples, larger batches share slots -- still valid bounds).  Consumers whose accumulations stay
//                         inside one sample (convolution forward / backward-data) scale each sample by ITS OWN maximum: the fp32-like
//                         relative precision of the f16x3 format then holds per sample, not only within 2^16 of the tensor's largest
//                         element.  The tensor-wide bound is the maximum over all floats but the flag.
// Every value in a record is an UPPER bound: a producer that cannot attribute its maxima to samples writes a mode-0 record.
constexpr int AMAX_SLOTS = 16, AMAX_STRIDE = 32, AMAX_MODE_IDX = 1, AMAX_PS_FIRST = 2, AMAX_PS_PER_LINE = 30;
static_assert(AMAX_SLOTS * AMAX_STRIDE == LADDER_ABSMAX_FLOATS, "record size of include/ladder_hip.h");
// per-sample scales are capped at 2^AMAX_PS_CAP above the tensor-wide scale (filter-gradient kernels re-scale their accumulators when
// the reduction crosses into another sample: the ratio of two scale products stays below 2^(2 * cap), far inside the fp32 range)
constexpr int AMAX_PS_CAP = 24;

__device__ __forceinline__ int amax_ps_index(int n) { return (n & (AMAX_SLOTS - 1)) * AMAX_STRIDE + AMAX_PS_FIRST + ((n >> 4) % AMAX_PS_PER_LINE); }

// Tensor-wide bound of a record in either mode.  COOPERATIVE: all 64 lanes of the calling wavefront must be active (call it at kernel
// start, before any divergent exit); each lane reads 8 of the 512 floats.
__device__ __forceinline__ float amax_load(const float* rec) {
  const int lane = threadIdx.x & 63;
  const float4 a = reinterpret_cast<const float4*>(rec)[2 * lane], b = reinterpret_cast<const float4*>(rec)[2 * lane + 1];
  float m = fmaxf(fmaxf(a.x, lane == 0 ? 0.f : a.y), fmaxf(a.z, a.w));          // (float 1 = the mode flag)
  m = fmaxf(m, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
  // (every lane holds the maximum: hand it to the scalar unit, so that the scales derived from it live in SGPRs like the 16 scalar
  // loads of the round-2 record did -- two more VGPRs spill in the 128-register halo kernels)
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wave_max(m))));
}
__device__ __forceinline__ bool amax_per_sample(const float* rec) { return rec[AMAX_MODE_IDX] != 0.f; }
// Bound for sample n: its own slot of a mode-1 record when `per_sample`, else the tensor-wide bound `tmax` (= amax_load(rec)).
__device__ __forceinline__ float amax_sample(const float* rec, int n, bool per_sample, float tmax) {
  return per_sample ? rec[amax_ps_index(n)] : tmax;
}

// wave_max for kernel EPILOGUES: the lane id is re-derived from an opaque copy of the thread id at the point of use, so that the shuffle
// addresses ((lane ^ o) * 4 for ds_bpermute) are computed here and not hoisted above a register-starved main loop (they were parked in
// scratch across the loop of the 128-register halo convolution kernels).
__device__ __forceinline__ float wave_max_late(float v) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int lane = t & 63;                                         // (1-D workgroups of whole wavefronts)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ o) << 2, __builtin_bit_cast(int, v))));
  return v;
}

// One slot update: fire-and-forget (no returned value: the wavefront does not wait for the L2).  A read-before-update filter was tried
// and was slower: the workgroup then waits a full L2 round trip for the value before it can retire.
__device__ __forceinline__ void amax_fold(unsigned* slot, float b) { atomicMax(slot, __builtin_bit_cast(unsigned, b)); }

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
    amax_fold(reinterpret_cast<unsigned*>(rec) + slot * AMAX_STRIDE, b);
  }
}
// The per-sample form (mode 1): the whole workgroup's values belong to sample n.  EVERY thread must call it.  The record must have been
// cleared before the launch; workgroup (0,0,0) sets the mode flag (65536 workgroups storing to that one address stretched the fused
// instance-norm + resize kernel by 40 %).  RESYNC: the workgroup commits more than once (the reduction array is reused).
template <bool RESYNC = false>
__device__ __forceinline__ void amax_commit_block_sample(float m, float* rec, int n) {
  __shared__ float amax_red_ps[16];
  m = wave_max_late(m);                              // (1-D workgroups: every caller)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  if ((tid & 63) == 0) amax_red_ps[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    float b = 0.f;
    for (int w = 0; w < nw; ++w) b = fmaxf(b, amax_red_ps[w]);
    amax_fold(reinterpret_cast<unsigned*>(rec) + amax_ps_index(n), b);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) rec[AMAX_MODE_IDX] = 1.f;
  }
  if (RESYNC) __syncthreads();
}
