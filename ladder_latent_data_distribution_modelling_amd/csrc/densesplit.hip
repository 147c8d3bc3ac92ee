// Small dense layers (batch-sized GEMMs: M = minibatch <= a few hundred rows) on the bf16 matrix cores, fp32-class (bf16x6: three bf16
// planes per fp32 operand, six plane products, fp32 accumulation -- split16.h; no tensor scale needed, so no absmax pass).
//
// The mapping MLP, the style projections, the inner VAE and the encoder heads are ~110 dependent GEMMs of 128 x 512 x 512 per
// iteration.  On the fp32 MFMA path each was two launches (split-K gather kernel + reduction pass, ~18 + 6 us) because a 32x32 tile
// over K = 512 alone is 256 fp32 MFMAs x 64 cycles = 7 us for one wavefront.  With the bf16 MFMA the same tile is 32 K-steps x 6 MFMAs x
// 32 cycles = 2.6 us, so ONE launch without split-K suffices: one workgroup per 32x32 output tile, its 4 wavefronts take every 4th
// K-step, load BOTH operand fragments straight from global memory in MFMA layout (8 consecutive reduction elements per lane: two
// 16-byte loads when the reduction index is contiguous in memory, 8 coalesced 4-byte loads when it is strided), split them in
// registers, and combine their partial tiles through LDS in a fixed order.  One generic kernel serves the three calls of a dense layer:
//     C[i][j] = sum_r A(i, r) * B(r, j),   A(i, r) at a + i * a_is + r * a_rs,   B(r, j) at b + r * b_rs + j * b_js
//   forward        y  = x w        : i = m, j = n, r = k    A = x  (a_is = K, a_rs = 1)   B = w  (b_rs = N, b_js = 1)
//   backward-data  dx = dy w^T     : i = m, j = k, r = n    A = dy (a_is = N, a_rs = 1)   B = w  (b_rs = 1, b_js = N)  -- no transposed copy
//   backward-weight dw = x^T dy    : i = k, j = n, r = m    A = x  (a_is = 1, a_rs = K)   B = dy (b_rs = N, b_js = 1), db = column sums of dy
//
// Round 4: the same kernels in STRICT fp32 (template parameter F32; entry points ladder_dense_*_small_f32, used when matmul_precision is
// "f32").  The 8 reduction elements a lane holds per step feed 8 fp32 MFMAs -- v_mfma_f32_32x32x2_f32 (lane half lh supplies reduction
// index 8 lh + t in MFMA t) resp. v_mfma_f32_16x16x4_f32 (lane quarter lq supplies 8 lq + t): the reduction order inside a step is a
// permutation of the indices, identical for both operands -- bit-exact fp32 FMA chains, no splitting.  One launch per call instead of the
// gather kernel + split-K second pass of the round-1 path (18 + 6 us -> ~6 us), no transposed copy of the weights for backward-data.
#include <cstdlib>
#include "split16.h"

namespace {

struct GemmSmall {
  int I, J, R;
  long a_is, a_rs, b_rs, b_js;
  int act, gate_act;
};

// 8 consecutive reduction elements r0..r0+7 of one operand row/column, zero beyond R
__device__ __forceinline__ void load8(const float* __restrict__ p, long rs, int r0, int R, bool live, float (&v)[8]) {
  if (live && rs == 1 && r0 + 8 <= R && ((reinterpret_cast<uintptr_t>(p + r0) & 15u) == 0)) {
    const float4 a = *reinterpret_cast<const float4*>(p + r0), b = *reinterpret_cast<const float4*>(p + r0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (live && r0 + e < R) ? p[(long)(r0 + e) * rs] : 0.f;
  }
}

template <bool F32>
__device__ __forceinline__ void mma_step(const float (&av)[8], const float (&bv)[8], f32x16& acc) {
  if (F32) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    return;
  }
  uint2 alo[3], ahi[3], blo[3], bhi[3];
  split4<3, false>(make_float4(av[0], av[1], av[2], av[3]), alo);
  split4<3, false>(make_float4(av[4], av[5], av[6], av[7]), ahi);
  split4<3, false>(make_float4(bv[0], bv[1], bv[2], bv[3]), blo);
  split4<3, false>(make_float4(bv[4], bv[5], bv[6], bv[7]), bhi);
  uint4 af[3], bf[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    af[p] = make_uint4(alo[p].x, alo[p].y, ahi[p].x, ahi[p].y);
    bf[p] = make_uint4(blo[p].x, blo[p].y, bhi[p].x, bhi[p].y);
  }
#pragma unroll
  for (int sum = 2; sum >= 0; --sum)
#pragma unroll
    for (int pa = 0; pa <= sum; ++pa) acc = mfma16<false>(af[pa], bf[sum - pa], acc);
}

// NW wavefronts per 32x32 tile; wavefront w takes K-steps w, w + NW, ... two at a time (both steps' loads are issued before the first
// split: the kernel is bound by the load -> split -> MFMA latency chain, ~0.8 us per round, not by throughput)
// One problem of a launch (blockIdx.z selects; up to two independent GEMMs share a launch: the backward-data and backward-weight
// calls of a dense layer -- a launch costs as much as the arithmetic of one of these)
struct GemmProb {
  const float* a;
  const float* b;
  const float* bias;
  float* c;
  const float* gate;
  float* colsum;
  GemmSmall g;
};

template <int NW, bool PAIR, bool F32>
__global__ __launch_bounds__(NW * 64) void gemm_small_split_kernel(const GemmProb p0, const GemmProb p1) {
  const GemmProb& pr = blockIdx.z ? p1 : p0;
  const GemmSmall& g = pr.g;
  const int tiles_j = (g.J + 31) / 32, tile_i = (int)blockIdx.x / tiles_j, tile_j = (int)blockIdx.x - tile_i * tiles_j;
  if (tile_i * 32 >= g.I) return;                             // (the grid has as many workgroups as the larger problem has tiles)
  const float* __restrict__ a = pr.a;
  const float* __restrict__ b = pr.b;
  const float* __restrict__ bias = pr.bias;
  float* __restrict__ c = pr.c;
  const float* __restrict__ gate = pr.gate;
  float* __restrict__ colsum = pr.colsum;
  __shared__ float red[NW][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int i0 = tile_i * 32, j0 = tile_j * 32;
  const int i = i0 + l31, j = j0 + l31;
  const bool i_ok = i < g.I, j_ok = j < g.J;
  const float* ap = a + (long)(i_ok ? i : 0) * g.a_is;
  const float* bp = b + (long)(j_ok ? j : 0) * g.b_js;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float csum = 0.f;                                           // column sum of B (bias gradient), this lane's column and r-octets
  const int nsteps = (g.R + 15) / 16;
  for (int ks = wv; ks < nsteps; ks += (PAIR ? 2 : 1) * NW) {
    const bool two = PAIR && ks + NW < nsteps;                // (wave-uniform; the 1024-thread variant is held to 128 VGPRs: one step)
    float a0[8], b0[8], a1[8], b1[8];
    load8(ap, g.a_rs, ks * 16 + 8 * lh, g.R, i_ok, a0);
    load8(bp, g.b_rs, ks * 16 + 8 * lh, g.R, j_ok, b0);
    load8(ap, g.a_rs, (ks + NW) * 16 + 8 * lh, g.R, i_ok && two, a1);
    load8(bp, g.b_rs, (ks + NW) * 16 + 8 * lh, g.R, j_ok && two, b1);
    if (colsum != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) csum += b0[e] + b1[e];
    }
    mma_step<F32>(a0, b0, acc);
    if (two) mma_step<F32>(a1, b1, acc);
  }
  // partial tiles of the NW wavefronts -> LDS (row-major 32 x 33), fixed-order sum, epilogue with 128-byte row segments
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wv][((e & 3) + 8 * (e >> 2) + 4 * lh) * 33 + l31] = acc[e];
  __syncthreads();
  if (tid < 256) {
    const int col = tid & 31, rq = tid >> 5;                 // 8 row groups x 32 columns
    const int jj = j0 + col;
    const float bval = (bias != nullptr && jj < g.J) ? bias[jj] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = rq + 8 * rr, ii = i0 + row;
      if (ii < g.I && jj < g.J) {
        float v = red[0][row * 33 + col];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][row * 33 + col];
        v = ladder_act_fn(v + bval, g.act);
        const long o = (long)ii * g.J + jj;
        if (gate != nullptr) v *= ladder_act_grad_from_out(gate[o], g.gate_act);
        c[o] = v;
      }
    }
  }
  if (colsum != nullptr && tile_i == 0) {                 // db[j] = sum_r B(r, j): lanes (l31, lh) x NW wavefronts, fixed order
    __syncthreads();
    csum += __shfl_xor(csum, 32, 64);
    if (lh == 0) red[wv][l31] = csum;
    __syncthreads();
    if (tid < 32 && j0 + tid < g.J) {
      float v = red[0][tid];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += red[w][tid];
      colsum[j0 + tid] = v;
    }
  }
}

// ---- 16x16 tiles (v_mfma_f32_16x16x32_bf16) for launches that would leave most of the chip idle -------------------------------------------
// A 128 x 512 output is only 64 tiles of 32x32 on 256 CUs; with 16x16 tiles it is 256 workgroups, each with a quarter of the work on its
// latency chain (load -> split -> MFMA -> reduce).  Fragment layout of the 16x16x32 instruction: lane l holds row / column l % 16 and the
// K-octet l / 16 (4 octets = 32); the accumulator (4 registers) holds column l % 16, rows 4 (l / 16) + 0..3.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool F32>
__device__ __forceinline__ void mma_step16(const float (&av)[8], const float (&bv)[8], f32x4v& acc) {
  if (F32) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], acc, 0, 0, 0);
    return;
  }
  uint2 alo[3], ahi[3], blo[3], bhi[3];
  split4<3, false>(make_float4(av[0], av[1], av[2], av[3]), alo);
  split4<3, false>(make_float4(av[4], av[5], av[6], av[7]), ahi);
  split4<3, false>(make_float4(bv[0], bv[1], bv[2], bv[3]), blo);
  split4<3, false>(make_float4(bv[4], bv[5], bv[6], bv[7]), bhi);
#pragma unroll
  for (int sum = 2; sum >= 0; --sum)
#pragma unroll
    for (int pa = 0; pa <= sum; ++pa) {
      const int pb = sum - pa;
      const uint4 af = make_uint4(alo[pa].x, alo[pa].y, ahi[pa].x, ahi[pa].y), bf = make_uint4(blo[pb].x, blo[pb].y, bhi[pb].x, bhi[pb].y);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
    }
}

template <int NW, bool F32>
__global__ __launch_bounds__(NW * 64) void gemm_small16_split_kernel(const GemmProb p0, const GemmProb p1) {
  const GemmProb& pr = blockIdx.z ? p1 : p0;
  const GemmSmall& g = pr.g;
  const int tiles_j = (g.J + 15) / 16, tile_i = (int)blockIdx.x / tiles_j, tile_j = (int)blockIdx.x - tile_i * tiles_j;
  if (tile_i * 16 >= g.I) return;
  __shared__ float red[NW][16 * 17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int i0 = tile_i * 16, j0 = tile_j * 16;
  const int i = i0 + l15, j = j0 + l15;
  const bool i_ok = i < g.I, j_ok = j < g.J;
  const float* ap = pr.a + (long)(i_ok ? i : 0) * g.a_is;
  const float* bp = pr.b + (long)(j_ok ? j : 0) * g.b_js;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  const int nsteps = (g.R + 31) / 32;
  for (int ks = wv; ks < nsteps; ks += NW) {
    float av[8], bv[8];
    load8(ap, g.a_rs, ks * 32 + 8 * lq, g.R, i_ok, av);
    load8(bp, g.b_rs, ks * 32 + 8 * lq, g.R, j_ok, bv);
    if (pr.colsum != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) csum += bv[e];
    }
    mma_step16<F32>(av, bv, acc);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wv][(4 * lq + e) * 17 + l15] = acc[e];
  __syncthreads();
  for (int t = tid; t < 256; t += NW * 64) {                  // (NW = 2: two outputs per thread)
    const int col = t & 15, row = t >> 4;
    const int ii = i0 + row, jj = j0 + col;
    if (ii < g.I && jj < g.J) {
      float v = red[0][row * 17 + col];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += red[w][row * 17 + col];
      v = ladder_act_fn(v + (pr.bias != nullptr ? pr.bias[jj] : 0.f), g.act);
      const long o = (long)ii * g.J + jj;
      if (pr.gate != nullptr) v *= ladder_act_grad_from_out(pr.gate[o], g.gate_act);
      pr.c[o] = v;
    }
  }
  if (pr.colsum != nullptr && tile_i == 0) {                  // db[j]: 4 K-octet lanes x NW wavefronts per column, fixed order
    __syncthreads();
    csum += __shfl_xor(csum, 16, 64);
    csum += __shfl_xor(csum, 32, 64);
    if (lq == 0) red[wv][l15] = csum;
    __syncthreads();
    if (tid < 16 && j0 + tid < g.J) {
      float v = red[0][tid];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += red[w][tid];
      pr.colsum[j0 + tid] = v;
    }
  }
}

template <bool F32>
int launch_gemm_probs(const GemmProb& p0, const GemmProb* p1, hipStream_t st) {
  const GemmSmall& g = p0.g;
  if (g.I <= 0 || g.J <= 0 || g.R <= 0) return LADDER_E_SHAPE;
  long tiles = (long)((g.J + 31) / 32) * ((g.I + 31) / 32);
  long tiles16 = (long)((g.J + 15) / 16) * ((g.I + 15) / 16);
  int steps32 = (g.R + 31) / 32;
  // fewer tiles than compute units and a reduction of >= 32 steps: 16 wavefronts per tile (a fixed 16-way K split inside the workgroup)
  bool wide = tiles <= 128 && (g.R + 15) / 16 >= 32;
  dim3 grid(1, 1, 1);
  if (p1 != nullptr) {
    const GemmSmall& h = p1->g;
    if (h.I <= 0 || h.J <= 0 || h.R <= 0) return LADDER_E_SHAPE;
    tiles = max(tiles, (long)((h.J + 31) / 32) * ((h.I + 31) / 32));
    tiles16 = max(tiles16, (long)((h.J + 15) / 16) * ((h.I + 15) / 16));
    steps32 = max(steps32, (h.R + 31) / 32);
    grid.z = 2;
    wide = true;                                            // (a short reduction just leaves the upper wavefronts idle)
  }
  const GemmProb& q = p1 != nullptr ? *p1 : p0;
  static const bool no16 = getenv("LADDER_DISABLE_GEMM16") != nullptr;
  if (!no16 && tiles * grid.z < 256 && tiles16 <= 4096) {   // the chip would be under-filled by 32x32 tiles: 16x16 tiles, K split over <= 8 wavefronts
    grid.x = (unsigned)tiles16;
    if (steps32 >= 8) hipLaunchKernelGGL((gemm_small16_split_kernel<8, F32>), grid, dim3(512), 0, st, p0, q);
    else if (steps32 >= 4) hipLaunchKernelGGL((gemm_small16_split_kernel<4, F32>), grid, dim3(256), 0, st, p0, q);
    else hipLaunchKernelGGL((gemm_small16_split_kernel<2, F32>), grid, dim3(128), 0, st, p0, q);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  grid.x = (unsigned)tiles;
  if (wide) hipLaunchKernelGGL((gemm_small_split_kernel<16, false, F32>), grid, dim3(1024), 0, st, p0, q);
  else hipLaunchKernelGGL((gemm_small_split_kernel<4, true, F32>), grid, dim3(256), 0, st, p0, q);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

template <bool F32>
int launch_gemm_small(const float* a, const float* b, const float* bias, float* c, const float* gate, float* colsum, const GemmSmall& g,
                      hipStream_t st) {
  const GemmProb p{a, b, bias, c, gate, colsum, g};
  return launch_gemm_probs<F32>(p, nullptr, st);
}

template <bool F32>
int dense_fwd_small(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, hipStream_t stream) {
  if (!ladder_dense_small_eligible(M, K, N)) return LADDER_E_SHAPE;
  const GemmSmall g{M, N, K, K, 1, N, 1, act, 0};
  return launch_gemm_small<F32>(x, w, bias, y, nullptr, nullptr, g, stream);
}

template <bool F32>
int dense_bwd_data_small(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act, hipStream_t stream) {
  if (!ladder_dense_small_eligible(M, K, N)) return LADDER_E_SHAPE;
  const GemmSmall g{M, K, N, N, 1, 1, N, LADDER_ACT_NONE, gate_act};
  return launch_gemm_small<F32>(dy, w, nullptr, dx, gate_y, nullptr, g, stream);
}

template <bool F32>
int dense_bwd_weight_small(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, hipStream_t stream) {
  if (!ladder_dense_small_eligible(M, K, N)) return LADDER_E_SHAPE;
  const GemmSmall g{K, N, M, 1, K, N, 1, LADDER_ACT_NONE, 0};
  return launch_gemm_small<F32>(x, dy, nullptr, dw, nullptr, db, g, stream);
}

template <bool F32>
int dense_bwd_small(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int M, int K, int N, const float* gate_y,
                    int gate_act, hipStream_t stream) {
  if (!ladder_dense_small_eligible(M, K, N) || dx == nullptr || dw == nullptr) return LADDER_E_SHAPE;
  const GemmProb pd{dy, w, nullptr, dx, gate_y, nullptr, GemmSmall{M, K, N, N, 1, 1, N, LADDER_ACT_NONE, gate_act}};
  const GemmProb pw{x, dy, nullptr, dw, nullptr, db, GemmSmall{K, N, M, 1, K, N, 1, LADDER_ACT_NONE, 0}};
  return launch_gemm_probs<F32>(pd, &pw, stream);
}

}  // namespace

extern "C" {

// Worth it while every output of the layer's three GEMMs is at most a few thousand 32x32 tiles: the forward / backward-data outputs are
// M x N and M x K (batch-sized M), the backward-weight output is K x N with the reduction over M -- a wide layer (K * N beyond 2 M
// elements, e.g. a 16 K-feature flatten into 512 units) would launch thousands of tiles of M / 16 strided 4-byte-load steps each and is
// better served by the split-K gather filter-gradient kernel, so such layers stay on the tiled path as a whole.
int ladder_dense_small_eligible(int M, int K, int N) {
  return (M > 0 && K > 0 && N > 0 && M <= 512 && (long)M * N <= (1L << 20) && (long)M * K <= (1L << 21) && (long)K * N <= (1L << 21)) ? 1 : 0;
}

int ladder_dense_fwd_small(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                           ladder_stream_t stream) {
  return dense_fwd_small<false>(x, w, bias, y, M, K, N, act, stream);
}

int ladder_dense_bwd_data_small(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act,
                                ladder_stream_t stream) {
  return dense_bwd_data_small<false>(dy, w, dx, M, K, N, gate_y, gate_act, stream);
}

int ladder_dense_bwd_weight_small(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, ladder_stream_t stream) {
  return dense_bwd_weight_small<false>(x, dy, dw, db, M, K, N, stream);
}

// Both gradient GEMMs of a dense layer in ONE launch (dx = dy w^T with the optional activation gate, dw = x^T dy, db = column sums of dy).
int ladder_dense_bwd_small(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int M, int K, int N,
                           const float* gate_y, int gate_act, ladder_stream_t stream) {
  return dense_bwd_small<false>(x, dy, w, dx, dw, db, M, K, N, gate_y, gate_act, stream);
}

// ---- the same four calls in strict fp32 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: bit-exact fp32 FMA chains) ---------------------
int ladder_dense_fwd_small_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                               ladder_stream_t stream) {
  return dense_fwd_small<true>(x, w, bias, y, M, K, N, act, stream);
}

int ladder_dense_bwd_data_small_f32(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act,
                                    ladder_stream_t stream) {
  return dense_bwd_data_small<true>(dy, w, dx, M, K, N, gate_y, gate_act, stream);
}

int ladder_dense_bwd_weight_small_f32(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, ladder_stream_t stream) {
  return dense_bwd_weight_small<true>(x, dy, dw, db, M, K, N, stream);
}

int ladder_dense_bwd_small_f32(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int M, int K, int N,
                               const float* gate_y, int gate_act, ladder_stream_t stream) {
  return dense_bwd_small<true>(x, dy, w, dx, dw, db, M, K, N, gate_y, gate_act, stream);
}

}  // extern "C"
