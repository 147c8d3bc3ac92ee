// Convolution / dense kernels for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain at the fp32 vector rate, 157.3 TFLOP/s peak).  Four kernel families:
//
//   conv3x3_halo_kernel     3x3 / stride 1 / SAME forward and backward-data of large maps: 8x32-pixel x 128-channel tile,
//                           the input halo of each 16-channel slab staged once in LDS for all 9 taps (dominant kernel, ~134 TF).
//   wgrad3x3_halo_kernel    filter gradient of the same layers (Cin % 64 == 0), all 9 taps fused over LDS-DMA-staged 1x32-pixel patches,
//                           12 balanced wavefronts x 6 MFMA tiles (~120 TF; ablation: 126 without staging, 133 without the
//                           per-patch barrier as well).
//   igemm_fwd_kernel<...>   everything else that is forward-shaped (conv fwd / bwd_data of any size, stride, kernel; dense):
//                           C[M x N] = A_gather[M x K] * B[K x N], M = pixels, N = Cout, K = (tap, ci).  A is gathered on the
//                           fly from the NHWC input (no im2col buffer); B is the HWIO filter bank viewed as a row-major
//                           [K x Cout] matrix -- exactly the reference's checkpoint layout.  Tap table + per-row validity
//                           bitmask on the fast path, split-K when the grid cannot fill the chip, stride-2 backward-data
//                           decomposed into its four output-parity classes.
//   igemm_wgrad_kernel<...> every other filter / weight gradient: dW[K x N] = A_gather^T[K x M] * dY[M x N], split over M with a
//                           fixed-order second-stage reduction (bit-reproducible), bias gradient fused.
//
// Common structure: 16-deep K chunks staged through double-buffered LDS (global -> registers issued before the MFMA block of
// the current chunk, registers -> LDS after it, one barrier per chunk); NHWC keeps ci contiguous, so every global load is a
// 16-byte access; padding lanes read a 16-byte zero buffer so loads stay unconditional.
#include <cstdlib>
#include "split16.h"
#include "convf32.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FastDiv {   // exact unsigned 32-bit division by a runtime constant (Granlund-Montgomery round-up method)
  uint32_t m, s1, s2;
};
struct IgemmDesc {
  int N, H, W, Cin;      // gathered tensor
  int Ho, Wo, Cout;      // GEMM-M spatial extent and GEMM-N
  int KH, KW, stride, ups, pad_t, pad_l;
  int M, K;
  int act;
  FastDiv div_howo, div_wo;
  // fast path (Cin % 16 == 0): explicit tap table.  K order = channel slab outer, taps inner; tap t reads the input at
  // (base_h + tap_dh[t], base_w + tap_dw[t]) and the filter tap tap_w[t] (= r*KW + s of the HWIO bank).
  int ntaps;
  signed char tap_dh[28], tap_dw[28], tap_w[28];
  // output pixel (ho, wo) of the M space is stored at y[n, out_h0 + ho*out_sh, out_w0 + wo*out_sw, :] of an OH x OW map
  int out_sh, out_sw, out_h0, out_w0, OH, OW;
};

// Fills the tap table / output mapping of an ordinary convolution (all KH*KW taps, dense output).
inline bool set_conv_taps(IgemmDesc& d) {
  d.out_sh = d.out_sw = 1;
  d.out_h0 = d.out_w0 = 0;
  d.OH = d.Ho;
  d.OW = d.Wo;
  d.ntaps = 0;
  if (d.ups != 1 || d.KH * d.KW > 28) return false;
  for (int r = 0; r < d.KH; ++r)
    for (int sx = 0; sx < d.KW; ++sx) {
      d.tap_dh[d.ntaps] = (signed char)r;
      d.tap_dw[d.ntaps] = (signed char)sx;
      d.tap_w[d.ntaps] = (signed char)(r * d.KW + sx);
      ++d.ntaps;
    }
  return true;
}

namespace {

constexpr int kThreads = 256;
#ifndef IGEMM_BK
#define IGEMM_BK 16
#endif
#ifndef IGEMM_MINW
#define IGEMM_MINW 1
#endif
#ifndef IGEMM_HALO
#define IGEMM_HALO 1
#endif
#ifndef IGEMM_WH_BLOCKS     // workgroups per CU (x rounds) the halo filter-gradient plan aims for
#define IGEMM_WH_BLOCKS 2
#endif
#ifndef IGEMM_FWD_MINW
#define IGEMM_FWD_MINW 4
#endif
constexpr int BK = IGEMM_BK;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Source of out-of-image / out-of-range operands: loads stay unconditional (no exec-mask branches around VMEM in the
// K loop); lanes that would read padding fetch these 16 zero bytes instead.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  const uint32_t t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
inline FastDiv make_fastdiv(uint32_t d) {
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  FastDiv f;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 1 ? l - 1 : 0;
  return f;
}

// One gathered element of A: returns pointer offset or -1 when the tap falls outside / between samples.
__device__ __forceinline__ long gather_off(const IgemmDesc& d, long img, int bh, int bw, int r, int s, int ci) {
  int nh = bh + r, nw = bw + s;
  if (nh < 0 || nw < 0) return -1;
  if (d.ups > 1) {
    if ((nh % d.ups) | (nw % d.ups)) return -1;
    nh /= d.ups;
    nw /= d.ups;
  }
  if (nh >= d.H || nw >= d.W) return -1;
  return img + ((long)nh * d.W + nw) * d.Cin + ci;
}

template <int BM, int BN, int WM, int WN, bool VECA, bool VECB>
__global__ __launch_bounds__(kThreads, (BM * BN >= 128 * 128) ? IGEMM_FWD_MINW : IGEMM_MINW) void igemm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             const IgemmDesc d, const int tiles_n, const bool fast,
                                                             float* __restrict__ part, const int chunks_per_split,
                                                             const float* __restrict__ gate, const int gate_act,
                                                             float* __restrict__ stats_part) {
  constexpr int LDA = BK + 1;  // odd row stride: conflict-free ds_read_b32 of A fragments
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int A_UNITS = BM * (BK / 4), B_UNITS = BK * (BN / 4);
  constexpr int AU = (A_UNITS + kThreads - 1) / kThreads, BU = (B_UNITS + kThreads - 1) / kThreads;
  static_assert(WM * WN == 4, "4 wavefronts");
  __shared__ float As[2][BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int HoWo = d.Ho * d.Wo;

  // per-thread A rows (fixed for the whole K loop): base pointer of tap (0,0) and a bitmask of the taps that
  // fall inside the image, so the K loop only adds a wave-uniform tap offset and tests one bit.
  constexpr bool FASTA = VECA;          // needs KH*KW <= 32 and ups == 1 (checked by the launcher via `fast`)
  int a_row[AU], a_kq[AU], a_bh[AU], a_bw[AU];
  long a_img[AU];
  bool a_ok[AU];
  const float* a_ptr[AU];
  uint32_t a_mask[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = tid + i * kThreads;
    a_row[i] = u / (BK / 4);
    a_kq[i] = u % (BK / 4);
    const int m = m0 + a_row[i];
    a_ok[i] = (u < A_UNITS) && (m < d.M);
    const uint32_t mm = a_ok[i] ? m : 0;
    const uint32_t n_img = fdiv(mm, d.div_howo), rem = mm - n_img * HoWo;
    const uint32_t ho = fdiv(rem, d.div_wo), wo = rem - ho * d.Wo;
    a_bh[i] = (int)ho * d.stride - d.pad_t;
    a_bw[i] = (int)wo * d.stride - d.pad_l;
    a_img[i] = (long)n_img * d.H * d.W * d.Cin;
    a_ptr[i] = x + a_img[i] + ((long)a_bh[i] * d.W + a_bw[i]) * d.Cin + a_kq[i] * 4;
    uint32_t msk = 0;
    if (FASTA && fast && a_ok[i]) {
      for (int t = 0; t < d.ntaps; ++t) {
        const int nh = a_bh[i] + d.tap_dh[t], nw = a_bw[i] + d.tap_dw[t];
        if (nh >= 0 && nh < d.H && nw >= 0 && nw < d.W) msk |= 1u << t;
      }
    }
    a_mask[i] = msk;
  }
  int b_kr[BU], b_nq[BU];
#pragma unroll
  for (int i = 0; i < BU; ++i) {
    const int u = tid + i * kThreads;
    b_kr[i] = u / (BN / 4);
    b_nq[i] = u % (BN / 4);
  }

  float4 ra0[AU], rb0[BU];
  auto load_chunk = [&](int c, float4 (&ra)[AU], float4 (&rb)[BU]) {
    // K order on the fast path: channel slab outer, taps inner -> the taps of one 16-channel slab re-read the same cache lines
    int k0 = c * BK;     // row of the [K x Cout] filter matrix this chunk starts at
    if (VECA) {  // Cin % BK == 0: the whole chunk lies inside one filter tap (wave-uniform tap, ci0)
      if (fast) {
        const int t = c % d.ntaps, ci0 = (c / d.ntaps) * BK;
        k0 = (int)d.tap_w[t] * d.Cin + ci0;
        const long coff = ((long)d.tap_dh[t] * d.W + d.tap_dw[t]) * d.Cin + ci0;
#pragma unroll
        for (int i = 0; i < AU; ++i) ra[i] = ld4(((a_mask[i] >> t) & 1u) ? a_ptr[i] + coff : g_zero16);
      } else {
        const int rs = k0 / d.Cin, ci0 = k0 - rs * d.Cin;
        const int r = rs / d.KW, s = rs - r * d.KW;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
          long off = a_ok[i] ? gather_off(d, a_img[i], a_bh[i], a_bw[i], r, s, ci0 + a_kq[i] * 4) : -1;
          ra[i] = ld4(off >= 0 ? x + off : g_zero16);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < AU; ++i) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + a_kq[i] * 4 + j;
          v[j] = 0.f;
          if (a_ok[i] && k < d.K) {
            const int rs = k / d.Cin, ci = k - rs * d.Cin;
            const int r = rs / d.KW, s = rs - r * d.KW;
            const long off = gather_off(d, a_img[i], a_bh[i], a_bw[i], r, s, ci);
            if (off >= 0) v[j] = x[off];
          }
        }
        ra[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      const int k = k0 + b_kr[i], n = n0 + b_nq[i] * 4;
      const bool inb = (tid + i * kThreads < B_UNITS) && k < d.KH * d.KW * d.Cin;
      if (VECB) {
        rb[i] = ld4((inb && n < d.Cout) ? w + (long)k * d.Cout + n : g_zero16);
      } else {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (inb && n + j < d.Cout) ? w[(long)k * d.Cout + n + j] : 0.f;
        rb[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto store_chunk = [&](int buf, const float4 (&ra)[AU], const float4 (&rb)[BU]) {
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      if (tid + i * kThreads < A_UNITS) {
        float* p = &As[buf][a_row[i] * LDA + a_kq[i] * 4];
        p[0] = ra[i].x; p[1] = ra[i].y; p[2] = ra[i].z; p[3] = ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      if (tid + i * kThreads < B_UNITS)
        *reinterpret_cast<float4*>(&Bs[buf][b_kr[i] * BN + b_nq[i] * 4]) = rb[i];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // split-K: blockIdx.y owns the chunk range [c_begin, nchunks)
  const int c_begin = blockIdx.y * chunks_per_split;
  const int nchunks = min((d.K + BK - 1) / BK, c_begin + chunks_per_split);

  auto mma_chunk = [&](int buf) {
    const float* Ab = &As[buf][(wm * TM + l31) * LDA + lh];
    const float* Bb = &Bs[buf][lh * BN + wn * TN + l31];
    // all fragment reads of the chunk are issued before the MFMA block (one LDS round trip per chunk, not per k-step)
    float af[BK / 2][MI], bf[BK / 2][NI];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[ks][mi] = Ab[mi * 32 * LDA + 2 * ks];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ks][ni] = Bb[2 * ks * BN + ni * 32];
    }
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks][mi], bf[ks][ni], acc[mi][ni], 0, 0, 0);
  };

  load_chunk(c_begin, ra0, rb0);
  store_chunk(0, ra0, rb0);
  __syncthreads();
  for (int c = c_begin; c < nchunks; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < nchunks) load_chunk(c + 1, ra0, rb0);
    mma_chunk(buf);
    if (c + 1 < nchunks) store_chunk(buf ^ 1, ra0, rb0);
    __syncthreads();
  }

  if (part != nullptr) {   // split-K partial: raw accumulators, bias/activation applied by splitk_epilogue_kernel
    float* o = part + (size_t)blockIdx.y * d.M * d.Cout;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * TN + ni * 32 + l31;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * TM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < d.M && n < d.Cout) o[(long)m * d.Cout + n] = acc[mi][ni][e];
        }
    }
    return;
  }
  // epilogue: bias + activation, 128-byte row segments per half-wave
  const bool dense_out = d.out_sh == 1 && d.out_sw == 1 && d.OH == d.Ho && d.OW == d.Wo;
  // batch-norm statistics of what is written (128x128 tiles only; lane = output channel: the column sums are local):
  // stats_part [tile_m][4][Cout] = sum | sum of squares | min | max per tile, reduced by ladder_bn_stats_minmax_from_partials
  constexpr bool STATS = (BM == 128 && BN == 128 && WM == 2);
  float st0[NI], st1[NI], smn[NI], smx[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) { st0[ni] = 0.f; st1[ni] = 0.f; smn[ni] = INFINITY; smx[ni] = -INFINITY; }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * TN + ni * 32 + l31;
    const float bv = (bias != nullptr && n < d.Cout) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * TM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < d.M && n < d.Cout) {
          long row = m;
          if (!dense_out) {   // phased backward-data: the M space is one parity class of the output map
            const uint32_t n_img = fdiv((uint32_t)m, d.div_howo), rem = (uint32_t)m - n_img * HoWo;
            const uint32_t ho = fdiv(rem, d.div_wo), wo = rem - ho * d.Wo;
            row = ((long)n_img * d.OH + d.out_h0 + (long)ho * d.out_sh) * d.OW + d.out_w0 + (long)wo * d.out_sw;
          }
          float v = ladder_act_fn(acc[mi][ni][e] + bv, d.act);
          // optional gate: multiply by act'(gate) evaluated from the producer layer's activation OUTPUT at the same element
          // (fuses the previous layer's activation backward into this backward-data pass)
          if (gate != nullptr) v *= ladder_act_grad_from_out(gate[row * d.Cout + n], gate_act);
          y[row * d.Cout + n] = v;
          if (STATS && stats_part != nullptr) {        // (wave-uniform: a plain forward / backward-data launch carries no statistics work -- ADVICE r4)
            st0[ni] += v;
            st1[ni] += v * v;
            smn[ni] = fminf(smn[ni], v);
            smx[ni] = fmaxf(smx[ni], v);
          }
        }
      }
    }
  }
  if (STATS && stats_part != nullptr) {
    float* sred = &Bs[0][0];                                 // [which 4][wm 2][128 channels] (the operand tiles are dead)
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float a0 = st0[ni] + __shfl_xor(st0[ni], 32, 64), a1 = st1[ni] + __shfl_xor(st1[ni], 32, 64);
      const float a2 = fminf(smn[ni], __shfl_xor(smn[ni], 32, 64)), a3 = fmaxf(smx[ni], __shfl_xor(smx[ni], 32, 64));
      if (lh == 0) {
        const int cc = wn * TN + ni * 32 + l31;
        sred[(0 * 2 + wm) * 128 + cc] = a0;
        sred[(1 * 2 + wm) * 128 + cc] = a1;
        sred[(2 * 2 + wm) * 128 + cc] = a2;
        sred[(3 * 2 + wm) * 128 + cc] = a3;
      }
    }
    __syncthreads();
    for (int t = tid; t < 4 * 128; t += kThreads) {
      const int which = t >> 7, c = t & 127;
      const float v0 = sred[(which * 2 + 0) * 128 + c], v1 = sred[(which * 2 + 1) * 128 + c];
      const float r = which < 2 ? v0 + v1 : (which == 2 ? fminf(v0, v1) : fmaxf(v0, v1));
      if (n0 + c < d.Cout) stats_part[((size_t)(tile / tiles_n) * 4 + which) * d.Cout + n0 + c] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The gather kernel on split operands (precision formats of split16.h / convsplit.hip): 128x128 tile, 4 wavefronts of 64x64, 32-channel
// K chunks = two 32x32x16 MFMA K-steps.  A is gathered exactly like igemm_fwd_kernel's fast path (tap table + validity mask) but from the
// PRE-SPLIT 16-bit planes of the input (ladder_presplit: an element is re-read once per tap and per output-channel tile, 18-36 times;
// splitting it in this kernel each time cost 38 % of its run time) straight into LDS ([plane][k-step][channel octet][row][8 x 16 bit]:
// a fragment = 32 consecutive 16-byte slots); B comes pre-split from ladder_filter_pack_split, whose blocks already have that layout.
constexpr int GS_BM = 128, GS_BN = 128, GS_BK = 32;

// ladder_presplit leaves behind the planes: 16 zero bytes (the source of padding taps), then a 16-byte header whose first word says
// whether the planes carry one scale per SAMPLE (f16x3, mode-1 absmax record) or one for the tensor.
__device__ __forceinline__ bool planes_per_sample(const uint16_t* planes, size_t plane_elems, int ns) {
  return reinterpret_cast<const uint32_t*>(planes + (size_t)ns * plane_elems)[4] != 0u;
}
constexpr int GS_PLANE = 2 * 2 * 128 * 16;      // bytes per plane of either operand tile (8192)

template <int PREC>
__global__ __launch_bounds__(kThreads, 2) void igemm_fwd_split_kernel(const uint16_t* __restrict__ xp, const size_t plane_elems,
                                                                     const uint4* __restrict__ wp,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     const IgemmDesc d, const int tiles_n, float* __restrict__ part,
                                                                     const int chunks_per_split, const float* __restrict__ gate,
                                                                     const int gate_act, const float* __restrict__ xamax,
                                                                     const float* __restrict__ wamax, float* __restrict__ stats_part) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  constexpr int OPB = NS * GS_PLANE;                       // bytes of one operand tile
  constexpr int AU = GS_BM * (GS_BK / 8) / kThreads;       // 2 (row, channel octet) units per thread, NS 16-byte loads each
  constexpr int B_CHUNKS = 2 * NS * 256;                   // 16-byte chunks of the B tile (two packed blocks)
  constexpr int BU = B_CHUNKS / kThreads;                  // 4 (NS=2) / 6 (NS=3)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * OPB];   // [buffer][A | B]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * GS_BM, cot = tile % tiles_n, n0 = cot * GS_BN;
  const int HoWo = d.Ho * d.Wo;
  const int nslabs16 = d.Cin / 16;
  // f16x3: the planes were scaled by ladder_presplit -- with ONE scale for the tensor or with one per sample (flag word behind the
  // planes); a GEMM row is one output pixel = one sample, so the un-scale factor is a per-row constant (table in LDS for the epilogue)
  float tmax = 0.f, cw = 1.f;
  bool x_ps = false;
  if (F16) {
    tmax = amax_load(xamax);
    cw = scale_from_absmax(amax_load(wamax));
    x_ps = planes_per_sample(xp, plane_elems, NS);
  }
  // (thread r < 128 fetches the scale of tile row r now: the load has the whole main loop to arrive)
  float my_row_scale = 1.f;
  if (F16 && tid < GS_BM) my_row_scale = scale_for_sample(xamax, (int)fdiv((uint32_t)min(m0 + tid, d.M - 1), d.div_howo), x_ps, tmax);

  long a_off[AU];                                          // element offset of the unit at tap (0,0), channel slab 0
  uint32_t a_mask[AU];
  int a_dst[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = tid + i * kThreads;
    const int row = u >> 2, oct = u & 3;                   // oct = k-step * 2 + channel octet of the 32-channel chunk
    const int m = m0 + row;
    const bool ok = m < d.M;
    const uint32_t mm = ok ? m : 0;
    const uint32_t n_img = fdiv(mm, d.div_howo), rem = mm - n_img * HoWo;
    const uint32_t ho = fdiv(rem, d.div_wo), wo = rem - ho * d.Wo;
    const int bh = (int)ho * d.stride - d.pad_t, bw = (int)wo * d.stride - d.pad_l;
    a_off[i] = (long)n_img * d.H * d.W * d.Cin + ((long)bh * d.W + bw) * d.Cin + oct * 8;
    uint32_t msk = 0;
    if (ok)
      for (int t = 0; t < d.ntaps; ++t) {
        const int nh = bh + d.tap_dh[t], nw = bw + d.tap_dw[t];
        if (nh >= 0 && nh < d.H && nw >= 0 && nw < d.W) msk |= 1u << t;
      }
    a_mask[i] = msk;
    a_dst[i] = (oct * 128 + row) * 16;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  static_assert(AU == 2, "two A units per thread");
  uint4 ra00, ra01, ra02, ra10, ra11, ra12;                // [unit][plane]  (named: hipcc demotes a uint4 array captured by two
  uint4 rb0, rb1, rb2, rb3, rb4, rb5;                      //  lambdas to LDS)
  static_assert(BU == 4 || BU == 6, "B staging rounds");
  auto load_chunk = [&](int c) {
    const int t = c % d.ntaps, ci0 = (c / d.ntaps) * GS_BK;
    const long coff = ((long)d.tap_dh[t] * d.W + d.tap_dw[t]) * d.Cin + ci0;
#define GS_ASRC(i_, p_) *reinterpret_cast<const uint4*>(((a_mask[i_] >> t) & 1u) ? (const void*)(xp + (size_t)(p_) * plane_elems + a_off[i_] + coff) \
                                                                             : (const void*)g_zero16)
    ra00 = GS_ASRC(0, 0); ra01 = GS_ASRC(0, 1); ra10 = GS_ASRC(1, 0); ra11 = GS_ASRC(1, 1);
    if (NS > 2) { ra02 = GS_ASRC(0, 2); ra12 = GS_ASRC(1, 2); }
#undef GS_ASRC
    const uint4* bs = wp + (((size_t)d.tap_w[t] * nslabs16 + ci0 / 16) * tiles_n + cot) * (NS * 256);
    // unit i = (k-step i / NS, plane i % NS), 256 threads x 16 B each
#define GS_BSRC(i_) bs[(size_t)((i_) / NS) * tiles_n * (NS * 256) + ((i_) % NS) * 256 + tid]
    rb0 = GS_BSRC(0); rb1 = GS_BSRC(1); rb2 = GS_BSRC(2); rb3 = GS_BSRC(3);
    if (BU > 4) { rb4 = GS_BSRC(4); rb5 = GS_BSRC(5); }
#undef GS_BSRC
  };
  auto store_chunk = [&](int buf) {
    unsigned char* Ab = lds + buf * 2 * OPB;
#define GS_ADST(i_, p_) *reinterpret_cast<uint4*>(Ab + (p_) * GS_PLANE + a_dst[i_])
    GS_ADST(0, 0) = ra00; GS_ADST(0, 1) = ra01; GS_ADST(1, 0) = ra10; GS_ADST(1, 1) = ra11;
    if (NS > 2) { GS_ADST(0, 2) = ra02; GS_ADST(1, 2) = ra12; }
#undef GS_ADST
#define GS_BDST(i_) *reinterpret_cast<uint4*>(Ab + OPB + ((i_) % NS) * GS_PLANE + (((i_) / NS) * 256 + tid) * 16)
    GS_BDST(0) = rb0; GS_BDST(1) = rb1; GS_BDST(2) = rb2; GS_BDST(3) = rb3;
    if (BU > 4) { GS_BDST(4) = rb4; GS_BDST(5) = rb5; }
#undef GS_BDST
  };
  auto mma_chunk = [&](int buf) {
    const unsigned char* Ab = lds + buf * 2 * OPB + (lh * 128 + wm * 64 + l31) * 16;
    const unsigned char* Bb = lds + buf * 2 * OPB + OPB + (lh * 128 + wn * 64 + l31) * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 a[2][NS], b[2][NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi][p] = *reinterpret_cast<const uint4*>(Ab + p * GS_PLANE + ks * 4096 + mi * 512);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni][p] = *reinterpret_cast<const uint4*>(Bb + p * GS_PLANE + ks * 4096 + ni * 512);
      }
#pragma unroll
      for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
        for (int pa = 0; pa <= sum; ++pa) {
          const int pb = sum - pa;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16<F16>(a[mi][pa], b[ni][pb], acc[mi][ni]);
        }
    }
  };

  const int c_begin = blockIdx.y * chunks_per_split;
  const int nchunks = min(d.ntaps * (d.Cin / GS_BK), c_begin + chunks_per_split);
  load_chunk(c_begin);
  store_chunk(0);
  __syncthreads();
  for (int c = c_begin; c < nchunks; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    mma_chunk(buf);
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  float* const row_unscale = reinterpret_cast<float*>(lds);   // (the operand tiles are dead after the last barrier of the main loop)
  if (F16) {
    if (tid < GS_BM) row_unscale[tid] = 1.f / (my_row_scale * cw);       // exact: powers of two
    __syncthreads();
  }
  if (part != nullptr) {   // split-K partial: un-scaled accumulators, bias/activation applied by splitk_epilogue_kernel
    float* o = part + (size_t)blockIdx.y * d.M * d.Cout;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + l31;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < d.M && n < d.Cout) o[(long)m * d.Cout + n] = F16 ? acc[mi][ni][e] * row_unscale[m - m0] : acc[mi][ni][e];
        }
    }
    return;
  }
  const bool dense_out = d.out_sh == 1 && d.out_sw == 1 && d.OH == d.Ho && d.OW == d.Wo;
  // batch-norm statistics of what is written (lane = output channel: the column sums are local), stats_part [tile_m][4][Cout] =
  // sum | sum of squares | min | max per tile, reduced by ladder_bn_stats_minmax_from_partials
  float st0[2] = {0.f, 0.f}, st1[2] = {0.f, 0.f}, smn[2] = {INFINITY, INFINITY}, smx[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    const float bv = (bias != nullptr && n < d.Cout) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < d.M && n < d.Cout) {
          long row = m;
          if (!dense_out) {
            const uint32_t n_img = fdiv((uint32_t)m, d.div_howo), rem = (uint32_t)m - n_img * HoWo;
            const uint32_t ho = fdiv(rem, d.div_wo), wo = rem - ho * d.Wo;
            row = ((long)n_img * d.OH + d.out_h0 + (long)ho * d.out_sh) * d.OW + d.out_w0 + (long)wo * d.out_sw;
          }
          float v = ladder_act_fn((F16 ? acc[mi][ni][e] * row_unscale[m - m0] : acc[mi][ni][e]) + bv, d.act);
          if (gate != nullptr) v *= ladder_act_grad_from_out(gate[row * d.Cout + n], gate_act);
          y[row * d.Cout + n] = v;
          st0[ni] += v;
          st1[ni] += v * v;
          smn[ni] = fminf(smn[ni], v);
          smx[ni] = fmaxf(smx[ni], v);
        }
      }
    }
  }
  if (stats_part != nullptr) {
    float* sred = reinterpret_cast<float*>(lds);             // [which 4][wm 2][128 channels] (the operand tiles are dead)
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const float a0 = st0[ni] + __shfl_xor(st0[ni], 32, 64), a1 = st1[ni] + __shfl_xor(st1[ni], 32, 64);
      const float a2 = fminf(smn[ni], __shfl_xor(smn[ni], 32, 64)), a3 = fmaxf(smx[ni], __shfl_xor(smx[ni], 32, 64));
      if (lh == 0) {
        const int cc = wn * 64 + ni * 32 + l31;
        sred[(0 * 2 + wm) * 128 + cc] = a0;
        sred[(1 * 2 + wm) * 128 + cc] = a1;
        sred[(2 * 2 + wm) * 128 + cc] = a2;
        sred[(3 * 2 + wm) * 128 + cc] = a3;
      }
    }
    __syncthreads();
    for (int t = tid; t < 4 * 128; t += kThreads) {
      const int which = t >> 7, c = t & 127;
      const float v0 = sred[(which * 2 + 0) * 128 + c], v1 = sred[(which * 2 + 1) * 128 + c];
      const float r = which < 2 ? v0 + v1 : (which == 2 ? fminf(v0, v1) : fmaxf(v0, v1));
      if (n0 + c < d.Cout) stats_part[((size_t)(tile / tiles_n) * 4 + which) * d.Cout + n0 + c] = r;
    }
  }
}

// dW[K x N] (+= over pixel split) = A_gather^T * dY.  GEMM-M here is the filter index k' = (r,s,ci).
#ifndef IGEMM_WG_MINW
#define IGEMM_WG_MINW 3
#endif
template <int BM, int BN, int WM, int WN, bool VECA, bool VECB>
__global__ __launch_bounds__(kThreads, (BM * BN >= 128 * 128) ? IGEMM_WG_MINW : 1) void igemm_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ out, float* __restrict__ bias_part,
                                                               const IgemmDesc d, const int tiles_n, const int m_per_split) {
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int A_UNITS = BK * (BM / 4), B_UNITS = BK * (BN / 4);
  constexpr int AU = (A_UNITS + kThreads - 1) / kThreads, BU = (B_UNITS + kThreads - 1) / kThreads;
  __shared__ __attribute__((aligned(16))) float At[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bt[2][BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid / WN, wn = wid % WN;
  const int tile = blockIdx.x;
  const int k0t = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int HoWo = d.Ho * d.Wo;
  const int p_begin = blockIdx.y * m_per_split;
  const int p_end = min(d.M, p_begin + m_per_split);

  // per-thread filter coordinates (fixed) for the A tile
  int a_pr[AU], a_kq[AU], a_r[AU][VECA ? 1 : 4], a_s[AU][VECA ? 1 : 4], a_ci[AU][VECA ? 1 : 4];
  bool a_kok[AU][VECA ? 1 : 4];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = tid + i * kThreads;
    a_pr[i] = u / (BM / 4);
    a_kq[i] = u % (BM / 4);
#pragma unroll
    for (int j = 0; j < (VECA ? 1 : 4); ++j) {
      const int k = k0t + a_kq[i] * 4 + j;
      a_kok[i][j] = (u < A_UNITS) && k < d.K;
      const int kk = a_kok[i][j] ? k : 0;
      const int rs = kk / d.Cin;
      a_ci[i][j] = kk - rs * d.Cin;
      a_r[i][j] = rs / d.KW;
      a_s[i][j] = rs - a_r[i][j] * d.KW;
    }
  }
  int b_pr[BU], b_nq[BU];
#pragma unroll
  for (int i = 0; i < BU; ++i) {
    const int u = tid + i * kThreads;
    b_pr[i] = u / (BN / 4);
    b_nq[i] = u % (BN / 4);
  }

  // bias gradient (column sums of dY) rides along in the k'-tile-0 workgroups: they already stream every dY row
  const bool do_bias = (bias_part != nullptr) && (tile / tiles_n == 0);
  float4 bsum[BU];
#pragma unroll
  for (int i = 0; i < BU; ++i) bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ra[AU], rb[BU];
  auto load_chunk = [&](int c) {
    const int pc = p_begin + c * BK;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const int p = pc + a_pr[i];
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (p < p_end) {
        const uint32_t n_img = fdiv((uint32_t)p, d.div_howo), rem = (uint32_t)p - n_img * HoWo;
        const int ho = (int)fdiv(rem, d.div_wo), wo = (int)rem - ho * d.Wo;
        const long img = (long)n_img * d.H * d.W * d.Cin;
        const int bh = ho * d.stride - d.pad_t, bw = wo * d.stride - d.pad_l;
        if (VECA) {
          const long off = a_kok[i][0] ? gather_off(d, img, bh, bw, a_r[i][0], a_s[i][0], a_ci[i][0]) : -1;
          const float4 t = ld4(off >= 0 ? x + off : g_zero16);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (a_kok[i][VECA ? 0 : j]) {
              const long off = gather_off(d, img, bh, bw, a_r[i][VECA ? 0 : j], a_s[i][VECA ? 0 : j], a_ci[i][VECA ? 0 : j]);
              if (off >= 0) v[j] = x[off];
            }
          }
        }
      }
      ra[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      const int p = pc + b_pr[i], n = n0 + b_nq[i] * 4;
      const bool inb = (tid + i * kThreads < B_UNITS) && p < p_end;
      if (VECB) {
        rb[i] = ld4((inb && n < d.Cout) ? dy + (long)p * d.Cout + n : g_zero16);
      } else {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (inb && n + j < d.Cout) ? dy[(long)p * d.Cout + n + j] : 0.f;
        rb[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (do_bias) {
        bsum[i].x += rb[i].x; bsum[i].y += rb[i].y; bsum[i].z += rb[i].z; bsum[i].w += rb[i].w;
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AU; ++i)
      if (tid + i * kThreads < A_UNITS) *reinterpret_cast<float4*>(&At[buf][a_pr[i] * BM + a_kq[i] * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BU; ++i)
      if (tid + i * kThreads < B_UNITS) *reinterpret_cast<float4*>(&Bt[buf][b_pr[i] * BN + b_nq[i] * 4]) = rb[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nchunks = (p_end - p_begin + BK - 1) / BK;
  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) load_chunk(c + 1);
    const float* Ab = &At[buf][lh * BM + wm * TM + l31];
    const float* Bb = &Bt[buf][lh * BN + wn * TN + l31];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = Ab[kk * BM + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = Bb[kk * BN + ni * 32];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
    if (c + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  float* o = out + (size_t)blockIdx.y * d.K * d.Cout;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * TN + ni * 32 + l31;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0t + wm * TM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (k < d.K && n < d.Cout) o[(long)k * d.Cout + n] = acc[mi][ni][e];
      }
    }
  }
  if (do_bias) {   // fixed-order reduction of the per-thread column sums through LDS (all MFMA reads of Bt are done)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BU; ++i)
      if (tid + i * kThreads < B_UNITS) *reinterpret_cast<float4*>(&Bt[0][b_pr[i] * BN + b_nq[i] * 4]) = bsum[i];
    __syncthreads();
    if (tid < BN && n0 + tid < d.Cout) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < BK; ++r) sacc += Bt[0][r * BN + tid];
      bias_part[(size_t)blockIdx.y * d.Cout + n0 + tid] = sacc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Filter / weight gradient on split operands: dW[k' x n] = sum_pixels A[pixel][k'] * dY[pixel][n], k' = (tap, ci).  The reduction index
// is the pixel while both operands are channel-contiguous in memory, so the LDS images stay [32-channel block][pixel][32 ch x 16 bit]
// (written by plain 8-byte stores of split float4s) and the MFMA fragments -- 8 consecutive pixels of one channel per lane -- are
// fetched with the transposing LDS read (ds_read_b64_tr_b16; see wgrad3x3_split_kernel in convsplit.hip for the layout argument).
// 128 x 128 tile (one filter tap x 128 input channels, Cin % 128 == 0), 4 wavefronts of 64x64, 32-pixel chunks, pixel range split
// over gridDim.y with a fixed-order second stage.
constexpr int GW_BLK = 32 * 64;                 // bytes of one 32-channel block of a 32-pixel chunk (2048)
constexpr int GW_PLANE = 4 * GW_BLK;            // 128 channels (8192)
typedef short gw_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gw_tr_frag(const unsigned char* p) {
  const gw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gw_s16x4*)(p));
  const gw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gw_s16x4*)(p + 4 * 64));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}
template <bool F16>
__device__ __forceinline__ float gw_frag_sum(const uint4 f) {
  const uint32_t w[4] = {f.x, f.y, f.z, f.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (F16) {
      const f32x2 v = __builtin_convertvector(__builtin_bit_cast(f16x2, w[i]), f32x2);
      s += v.x + v.y;
    } else {
      s += __builtin_bit_cast(float, w[i] << 16) + __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
  }
  return s;
}

template <int PREC>
__global__ __launch_bounds__(kThreads, 2) void igemm_wgrad_split_kernel(const uint16_t* __restrict__ xp, const size_t x_plane_elems,
                                                                       const uint16_t* __restrict__ dyp, const size_t dy_plane_elems,
                                                                       float* __restrict__ out, float* __restrict__ bias_part,
                                                                       const IgemmDesc d, const int tiles_n, const int m_per_split,
                                                                       const float* __restrict__ xamax, const float* __restrict__ damax) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  constexpr int OPB = NS * GW_PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * OPB];     // [buffer][A | dY]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = blockIdx.x;
  const int k0t = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
  const int HoWo = d.Ho * d.Wo;
  const int p_begin = blockIdx.y * m_per_split;
  const int p_end = min(d.M, p_begin + m_per_split);
  const int tap = k0t / d.Cin, ci0 = k0t - tap * d.Cin;       // the tile lies inside one tap (Cin % 128 == 0)
  const int tr = tap / d.KW, ts = tap - tr * d.KW;
  // f16x3 scales: either operand's planes may carry one scale per sample (ladder_presplit).  The reduction runs over the pixels of ALL
  // samples, so the accumulators hold s * (partial sum) with s = cx(n) * cd(n) of the sample being reduced and are multiplied by
  // s_new / s_old (a power of two: exact) whenever a 32-pixel chunk starts in another sample.  Needs chunks inside one sample.
  // The scales of the samples this workgroup reduces over are gathered ONCE into an LDS table (vector loads, counted by vmcnt like the chunk
  // loads they precede): scalar loads between the segments made the first LDS wait of every segment stall on them (lgkmcnt counts both).
  constexpr int GW_NSC = 256;
  __shared__ float2 seg_scale[GW_NSC];                       // [sample - n_first] = {cx * cd, cd}
  float xt = 0.f, dt = 0.f, s_cur = 1.f, cd_cur = 1.f;
  bool x_ps = false, d_ps = false;
  int n_cur = 0, n_first = 0;
  if (F16) {
    xt = amax_load(xamax);
    dt = amax_load(damax);
    x_ps = planes_per_sample(xp, x_plane_elems, NS);
    d_ps = planes_per_sample(dyp, dy_plane_elems, NS);
    if ((x_ps || d_ps) && ((HoWo % 32) != 0 || (p_begin % 32) != 0)) __builtin_trap();     // the caller asked for per-sample planes on a map it must not
    n_first = n_cur = (int)fdiv((uint32_t)min(p_begin, d.M - 1), d.div_howo);
    // (one entry when both operands carry a single scale; else m_per_split / HoWo + 1 samples per workgroup: far below the table size)
    const int n_tab = (x_ps || d_ps) ? (int)fdiv((uint32_t)(max(p_end, p_begin + 1) - 1), d.div_howo) - n_first : 0;
    if (n_tab >= GW_NSC) __builtin_trap();
    if (tid <= n_tab) {
      const float cdn = scale_for_sample(damax, n_first + tid, d_ps, dt);
      seg_scale[tid] = make_float2(scale_for_sample(xamax, n_first + tid, x_ps, xt) * cdn, cdn);
    }
  }
  const bool do_bias = (bias_part != nullptr) && (tile / tiles_n == 0) && wm == 0;
  float bsum[2] = {0.f, 0.f};

  // staging units: 32 pixels x 16 channel octets per operand = 512 units = 2 per thread and operand, NS 16-byte plane loads each;
  // unit u -> pixel u >> 4, octet u & 15 (both operands arrive pre-split: ladder_presplit)
  const int s_pix = tid >> 4, s_oct = tid & 15;              // + 16 pixels for the second round
  const int s_dst = (s_oct >> 2) * GW_BLK + s_pix * 64 + (s_oct & 3) * 16;   // + round * 16 * 64
  const bool n_ok = (n0 + s_oct * 8) < d.Cout;
  const int frag_lane = (8 * lh + (l15 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (l15 & 3)) * 2;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  uint4 ra00, ra01, ra02, ra10, ra11, ra12, rd00, rd01, rd02, rd10, rd11, rd12;     // [round][plane] (named: see igemm_fwd_split_kernel)
  auto unit_offsets = [&](int pc, int round, long& ao, long& dof) {
    const int p = pc + s_pix + round * 16;
    ao = dof = -1;
    if (p < p_end) {
      const uint32_t n_img = fdiv((uint32_t)p, d.div_howo), rem = (uint32_t)p - n_img * HoWo;
      const int ho = (int)fdiv(rem, d.div_wo), wo = (int)rem - ho * d.Wo;
      const int hi = ho * d.stride - d.pad_t + tr, wi = wo * d.stride - d.pad_l + ts;
      if (hi >= 0 && hi < d.H && wi >= 0 && wi < d.W) ao = (((long)n_img * d.H + hi) * d.W + wi) * d.Cin + ci0 + s_oct * 8;
      if (n_ok) dof = (long)p * d.Cout + n0 + s_oct * 8;
    }
  };
#define GW_LD(base_, elems_, off_, p_) *reinterpret_cast<const uint4*>((off_) >= 0 ? (const void*)((base_) + (size_t)(p_) * (elems_) + (off_)) \
                                                                                : (const void*)g_zero16)
  auto load_chunk = [&](int c) {
    const int pc = p_begin + c * 32;
    long ao, dof;
    unit_offsets(pc, 0, ao, dof);
    ra00 = GW_LD(xp, x_plane_elems, ao, 0); ra01 = GW_LD(xp, x_plane_elems, ao, 1);
    rd00 = GW_LD(dyp, dy_plane_elems, dof, 0); rd01 = GW_LD(dyp, dy_plane_elems, dof, 1);
    if (NS > 2) { ra02 = GW_LD(xp, x_plane_elems, ao, 2); rd02 = GW_LD(dyp, dy_plane_elems, dof, 2); }
    unit_offsets(pc, 1, ao, dof);
    ra10 = GW_LD(xp, x_plane_elems, ao, 0); ra11 = GW_LD(xp, x_plane_elems, ao, 1);
    rd10 = GW_LD(dyp, dy_plane_elems, dof, 0); rd11 = GW_LD(dyp, dy_plane_elems, dof, 1);
    if (NS > 2) { ra12 = GW_LD(xp, x_plane_elems, ao, 2); rd12 = GW_LD(dyp, dy_plane_elems, dof, 2); }
  };
#undef GW_LD
  auto store_chunk = [&](int buf) {
    unsigned char* Ab = lds + buf * 2 * OPB;
#define GW_ST(op_, round_, p_) *reinterpret_cast<uint4*>(Ab + (op_) * OPB + (p_) * GW_PLANE + s_dst + (round_) * 16 * 64)
    GW_ST(0, 0, 0) = ra00; GW_ST(0, 0, 1) = ra01; GW_ST(0, 1, 0) = ra10; GW_ST(0, 1, 1) = ra11;
    GW_ST(1, 0, 0) = rd00; GW_ST(1, 0, 1) = rd01; GW_ST(1, 1, 0) = rd10; GW_ST(1, 1, 1) = rd11;
    if (NS > 2) { GW_ST(0, 0, 2) = ra02; GW_ST(0, 1, 2) = ra12; GW_ST(1, 0, 2) = rd02; GW_ST(1, 1, 2) = rd12; }
#undef GW_ST
  };
  auto mma_chunk = [&](int buf) {
    const unsigned char* Ab = lds + buf * 2 * OPB + frag_lane + wm * 2 * GW_BLK;
    const unsigned char* Bb = lds + buf * 2 * OPB + OPB + frag_lane + wn * 2 * GW_BLK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 a[2][NS], b[2][NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi][p] = gw_tr_frag(Ab + p * GW_PLANE + mi * GW_BLK + ks * 16 * 64);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni][p] = gw_tr_frag(Bb + p * GW_PLANE + ni * GW_BLK + ks * 16 * 64);
      }
#pragma unroll
      for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
        for (int pa = 0; pa <= sum; ++pa) {
          const int pb = sum - pa;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16<F16>(a[mi][pa], b[ni][pb], acc[mi][ni]);
        }
      if (do_bias) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int p = NS - 1; p >= 0; --p) bsum[ni] += gw_frag_sum<F16>(b[ni][p]);
      }
    }
  };

  const int nchunks = (p_end - p_begin + 31) / 32;
  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  // chunk loop in SEGMENTS of chunks that share their scales (per-sample planes: the chunks of one sample; else one segment): the inner
  // loop is the plain pipelined loop, all re-scaling sits between segments (a test inside the chunk loop cost the kernel 19 %)
  if (F16) {                                                 // (after the barrier above: the table is complete)
    s_cur = seg_scale[0].x;
    cd_cur = seg_scale[0].y;
  }
  const bool any_ps = F16 && (x_ps || d_ps);
  const int cps = any_ps ? HoWo / 32 : nchunks;             // chunks per sample (HoWo % 32 == 0 and p_begin % 32 == 0 were checked)
  int c = 0;
  int seg_end = any_ps ? min(nchunks, cps - (int)((uint32_t)(p_begin / 32) - (uint32_t)n_cur * (uint32_t)cps)) : nchunks;
  while (c < nchunks) {
    // the NEXT segment's scales are fetched now and used after this segment's chunks
    float s_nxt = s_cur, cd_nxt = cd_cur;
    if (any_ps && seg_end < nchunks) {
      const float2 e = seg_scale[n_cur + 1 - n_first];
      s_nxt = e.x;
      cd_nxt = e.y;
    }
    for (; c < seg_end; ++c) {
      const int buf = c & 1;
      if (c + 1 < nchunks) load_chunk(c + 1);
      mma_chunk(buf);
      if (c + 1 < nchunks) store_chunk(buf ^ 1);
      __syncthreads();
    }
    if (c < nchunks) {                                       // entering the next sample: re-scale the accumulators (powers of two: exact)
      if (s_nxt != s_cur || cd_nxt != cd_cur) {
        const float ratio = s_nxt / s_cur, bratio = cd_nxt / cd_cur;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] *= ratio;
        bsum[0] *= bratio;
        bsum[1] *= bratio;
      }
      ++n_cur; s_cur = s_nxt; cd_cur = cd_nxt;
      seg_end = min(nchunks, seg_end + cps);
    }
  }

  const float unscale = F16 ? 1.f / s_cur : 1.f;
  const float cd = cd_cur;
  float* o = out + (size_t)blockIdx.y * d.K * d.Cout;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0t + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (k < d.K && n < d.Cout) o[(long)k * d.Cout + n] = acc[mi][ni][e] * unscale;
      }
    }
    if (do_bias) {
      const float v = (bsum[ni] + __shfl_down(bsum[ni], 32, 64)) * (F16 ? 1.f / cd : 1.f);
      if (lh == 0 && n < d.Cout) bias_part[(size_t)blockIdx.y * d.Cout + n] = v;
    }
  }
}

__global__ void splitk_epilogue_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                       int S, size_t MN, int N, int act, const float* __restrict__ gate, int gate_act) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MN) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += part[(size_t)z * MN + i];   // fixed order
  if (bias != nullptr) s += bias[i % N];
  s = ladder_act_fn(s, act);
  if (gate != nullptr) s *= ladder_act_grad_from_out(gate[i], gate_act);
  y[i] = s;
}

// The same second pass for a call whose M space is a strided subset of the output map (one parity class of a stride-2 backward-data):
// partials are dense in the class-local pixel index m, the result goes to y[n, out_h0 + ho*out_sh, out_w0 + wo*out_sw, :].
__global__ void splitk_epilogue_strided_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                               int S, size_t MN, int act, const float* __restrict__ gate, int gate_act, const IgemmDesc d) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MN) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += part[(size_t)z * MN + i];   // fixed order
  const uint32_t m = (uint32_t)(i / d.Cout), n = (uint32_t)(i - (size_t)m * d.Cout);
  const uint32_t n_img = fdiv(m, d.div_howo), rem = m - n_img * (uint32_t)(d.Ho * d.Wo);
  const uint32_t ho = fdiv(rem, d.div_wo), wo = rem - ho * d.Wo;
  const size_t o = ((((size_t)n_img * d.OH + d.out_h0 + (size_t)ho * d.out_sh) * d.OW + d.out_w0 + (size_t)wo * d.out_sw)) * d.Cout + n;
  if (bias != nullptr) s += bias[n];
  s = ladder_act_fn(s, act);
  if (gate != nullptr) s *= ladder_act_grad_from_out(gate[o], gate_act);
  y[o] = s;
}

__global__ void reduce_splits_kernel(const float* __restrict__ ws, float* __restrict__ out, int S, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += ws[(size_t)z * n + i];  // fixed order
  out[i] = s;
}

// Many partials of a SHORT vector (filter gradients of the image-side convs: up to 1024 splits of a few thousand floats, and every
// fused bias gradient): 16 threads share one output, each summing every 16th partial, then a fixed-order LDS tree.  The serial
// version above needs `S` dependent loads per thread with only n threads in flight (235 us for enc.conv2d's 768 x 3456 partials).
__global__ __launch_bounds__(256) void reduce_splits_wide_kernel(const float* __restrict__ ws, float* __restrict__ out, int S, size_t n) {
  __shared__ float part[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + o;
  float s = 0.f;
  if (i < n)
    for (int z = g; z < S; z += 16) s += ws[(size_t)z * n + i];
  part[g][o] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][o];   // fixed order
    out[i] = t;
  }
}

// Block partials with trailing bias slots, ws[S][n+nb]: out[i] = sum_s ws[s][i] (i < n), bias_out[j] = sum_s ws[s][n+j].
// 16 threads per output, each summing every 16th partial, then a fixed-order LDS combine (as reduce_splits_wide_kernel).
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                    float* __restrict__ bias_out, int S, int n, int nb) {
  __shared__ float part[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + o;
  float a = 0.f;
  if (i < n + nb)
    for (int z = g; z < S; z += 16) a += ws[(size_t)z * (n + nb) + i];
  part[g][o] = a;
  __syncthreads();
  if (g == 0 && i < n + nb) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][o];
    if (i < n) out[i] = t;
    else if (bias_out != nullptr) bias_out[i - n] = t;
  }
}

static inline void launch_reduce_splits(const float* ws, float* out, int S, size_t n, hipStream_t st) {
  if (S >= 32 && n * 16 <= ((size_t)1 << 24))
    hipLaunchKernelGGL(reduce_splits_wide_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, ws, out, S, n);
  else
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, out, S, n);
}

__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wT, int KH, int KW, int Cin, int Cout) {
  // wT[KH-1-r][KW-1-s][co][ci] = w[r][s][ci][co]; 32x32 LDS tile transpose per tap
  __shared__ float t[32][33];
  const int tap = blockIdx.z, r = tap / KW, s = tap % KW;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const float* src = w + (size_t)tap * Cin * Cout;
  float* dst = wT + (size_t)((KH - 1 - r) * KW + (KW - 1 - s)) * Cin * Cout;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int ci = ci0 + i, co = co0 + threadIdx.x;
    t[i][threadIdx.x] = (ci < Cin && co < Cout) ? src[(size_t)ci * Cout + co] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int co = co0 + i, ci = ci0 + threadIdx.x;
    if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = t[threadIdx.x][i];
  }
}

struct TileCfg { int bm, bn; };


// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / SAME convolution with an LDS-staged INPUT HALO tile (the decoder's dominant layers and their
// backward-data passes).  One workgroup = 8 wavefronts = an 8x32-pixel output patch x 128 output channels.
// For every 16-channel input slab the (8+2)x(32+2) halo patch is fetched from HBM/L2 ONCE and serves all 9 filter
// taps straight out of LDS (the gather kernel above re-fetches it per tap); the filter slab [16 x 128] of each tap is
// double-buffered.  Per 9 K-chunks a thread issues 12 global loads instead of 36, and the 256-pixel patch halves the
// filter bytes per MFMA.  Wavefront (wm, wn) owns patch rows {2wm, 2wm+1} x channels [64wn, 64wn+64): an MFMA A
// fragment is 32 consecutive pixels of one patch row, i.e. consecutive LDS rows of odd stride 17 -> conflict-free.
constexpr int HT_H = 8, HT_W = 32, HT_PW = HT_W + 2, HT_PH = HT_H + 2, HT_LDA = BK + 1;
constexpr int HT_THREADS = 512, HT_BN = 128;
constexpr int HT_HALO_UNITS = HT_PH * HT_PW * (BK / 4);                      // float4 units per slab (1360)
constexpr int HT_AU = (HT_HALO_UNITS + HT_THREADS - 1) / HT_THREADS;         // 3

__global__ __launch_bounds__(HT_THREADS, 2) void conv3x3_halo_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     const int N, const int H, const int W, const int Cin,
                                                                     const int Cout, const int act, const int tiles_n) {
  static_assert(BK == 16, "halo kernel is written for 16-channel slabs");
  __shared__ float Ah[2][HT_PH * HT_PW * HT_LDA];
  __shared__ __attribute__((aligned(16))) float Bh[2][BK * HT_BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / tiles_n, n0 = (tile % tiles_n) * HT_BN;
  const int tw_n = W / HT_W, th_n = H / HT_H;
  const int img = mt / (tw_n * th_n), rem = mt - img * (tw_n * th_n);
  const int h0 = (rem / tw_n) * HT_H, w0 = (rem % tw_n) * HT_W;

  // halo load units (fixed per workgroup): source pointer at channel 0 of the slab, or the zero buffer
  const float* hsrc[HT_AU];
  int hdst[HT_AU];
#pragma unroll
  for (int i = 0; i < HT_AU; ++i) {
    const int u = tid + i * HT_THREADS;
    const int pix = u >> 2, kq = u & 3;
    const int hr = pix / HT_PW, hc = pix - hr * HT_PW;
    const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
    const bool ok = (u < HT_HALO_UNITS) && hi >= 0 && hi < H && wi >= 0 && wi < W;
    hsrc[i] = ok ? x + (((long)img * H + hi) * W + wi) * Cin + kq * 4 : nullptr;
    hdst[i] = pix * HT_LDA + kq * 4;
  }
  const int b_kr = tid >> 5, b_nq = tid & 31;                                // 16 rows x 32 float4 = 512 units
  const bool b_ok = (n0 + b_nq * 4) < Cout;
  const float* bsrc = w + (long)b_kr * Cout + n0 + b_nq * 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ha[HT_AU], rb;
  auto load_halo = [&](int slab) {
#pragma unroll
    for (int i = 0; i < HT_AU; ++i) ha[i] = ld4(hsrc[i] != nullptr ? hsrc[i] + slab * BK : g_zero16);
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < HT_AU; ++i)
      if (tid + i * HT_THREADS < HT_HALO_UNITS) {
        float* p = &Ah[buf][hdst[i]];
        p[0] = ha[i].x; p[1] = ha[i].y; p[2] = ha[i].z; p[3] = ha[i].w;
      }
  };
  auto load_b = [&](int slab, int tap) { rb = ld4(b_ok ? bsrc + ((long)tap * Cin + slab * BK) * Cout : g_zero16); };
  auto store_b = [&](int buf) { *reinterpret_cast<float4*>(&Bh[buf][b_kr * HT_BN + b_nq * 4]) = rb; };

  const int nslabs = Cin / BK;
  load_halo(0);
  load_b(0, 0);
  store_halo(0);
  store_b(0);
  __syncthreads();
  int bbuf = 0;
  for (int slab = 0; slab < nslabs; ++slab) {
    const int hb = slab & 1;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      // prefetch: next filter slab every tap; next input halo once per slab (issued at tap 0, written at tap 4)
      const bool last = (slab + 1 == nslabs) && (tap == 8);
      if (!last) load_b(tap == 8 ? slab + 1 : slab, tap == 8 ? 0 : tap + 1);
      if (tap == 0 && slab + 1 < nslabs) load_halo(slab + 1);
      const int r = tap / 3, sft = tap - 3 * r;
      const float* Ab = &Ah[hb][((2 * wm + r) * HT_PW + sft + l31) * HT_LDA + lh];
      const float* Bb = &Bh[bbuf][lh * HT_BN + wn * 64 + l31];
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        float a[2], b[2];
        a[0] = Ab[2 * ks];
        a[1] = Ab[HT_PW * HT_LDA + 2 * ks];
        b[0] = Bb[2 * ks * HT_BN];
        b[1] = Bb[2 * ks * HT_BN + 32];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
      if (tap == 4 && slab + 1 < nslabs) store_halo(hb ^ 1);
      if (!last) store_b(bbuf ^ 1);
      __syncthreads();
      bbuf ^= 1;
    }
  }

#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const long rowbase = (((long)img * H + h0 + 2 * wm + mi) * W + w0) * Cout;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (n < Cout) y[rowbase + (long)px * Cout + n] = ladder_act_fn(acc[mi][ni][e] + bv, act);
      }
    }
  }
}

// Debug / validation switch: LADDER_DISABLE_HALO=1 in the environment routes every convolution through the generic gather
// kernels (used by the tests to cross-check the halo kernels in situ at full size).  Read on every call: no cached state.
inline bool halo_disabled() {
  const char* e = getenv("LADDER_DISABLE_HALO");
  return e != nullptr && e[0] == '1';
}

bool halo_eligible(const IgemmDesc& d) {
  if (halo_disabled()) return false;
#if IGEMM_HALO
  return d.KH == 3 && d.KW == 3 && d.stride == 1 && d.ups == 1 && d.pad_t == 1 && d.pad_l == 1 && d.Ho == d.H && d.Wo == d.W &&
         (d.Cin % BK) == 0 && (d.Cout % 4) == 0 && (d.W % HT_W) == 0 && (d.H % HT_H) == 0 && d.Cout >= 64 &&
         (long)d.N * (d.H / HT_H) * (d.W / HT_W) * ((d.Cout + HT_BN - 1) / HT_BN) >= 512;
#else
  return false;
#endif
}

int launch_halo(const float* x, const float* w, const float* bias, float* y, const IgemmDesc& d, hipStream_t st) {
  const int tiles_n = (d.Cout + HT_BN - 1) / HT_BN;
  const int tiles_m = d.N * (d.H / HT_H) * (d.W / HT_W);
  hipLaunchKernelGGL(conv3x3_halo_kernel, dim3(tiles_m * tiles_n), dim3(HT_THREADS), 0, st, x, w, bias, y, d.N, d.H, d.W, d.Cin,
                     d.Cout, d.act, tiles_n);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

struct SplitPlan { int splits, cps; };
// split-K only when the output tiling cannot fill the chip and K is long enough to amortise the second pass
SplitPlan plan_splitk(long M, int K, int Cout, int bm, int bn) {
  const long tiles = ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn);
  const int nchunks = (K + BK - 1) / BK;
  SplitPlan p{1, nchunks};
  if (tiles >= 640 || nchunks < 16) return p;
  long s = tiles < 128 ? (384 + tiles - 1) / tiles : 1024 / tiles;   // small grids: >=1.5 wg/CU; mid grids: one full round of 4 wg/CU
  if (s > nchunks / 8) s = nchunks / 8;
  if (tiles < 128 && s < nchunks / 4 && s * tiles < 256) s = nchunks / 4;
  if (s > 32) s = 32;
  if (s < 2) return p;
  p.cps = (int)((nchunks + s - 1) / s);
  p.splits = (nchunks + p.cps - 1) / p.cps;
  return p;
}

template <int BM, int BN, int WM, int WN>
int launch_fwd(const float* x, const float* w, const float* bias, float* y, const IgemmDesc& d, void* ws, size_t ws_bytes,
               hipStream_t st, const float* gate = nullptr, int gate_act = 0, float* stats_part = nullptr) {
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.Cout + BN - 1) / BN;
  const bool veca = (d.Cin % BK) == 0, vecb = (d.Cout % 4) == 0;
  const bool fast = d.ntaps > 0;
  const bool dense_out = d.out_sh == 1 && d.out_sw == 1 && d.OH == d.Ho && d.OW == d.Wo;
  // (round 4: also the parity classes of a stride-2 backward-data -- a quarter of the pixels, 1 ... 4 of the 9 taps: 32 tiles on 256 CUs
  // for the 8x8 <- 4x4 layer -- through splitk_epilogue_strided_kernel: 241 -> 96 us for its four launches)
  // (only the small ones: at 512 tiles the strided second pass costs more than the idle CUs -- 16x16 <- 8x8: 174 us single pass, 195 split)
  SplitPlan sp = (dense_out || (long)tiles_m * tiles_n <= 256) ? plan_splitk(d.M, d.K, d.Cout, BM, BN) : SplitPlan{1, (d.K + BK - 1) / BK};
  const size_t need = (size_t)sp.splits * d.M * d.Cout * sizeof(float);
  if (sp.splits > 1 && (ws == nullptr || ws_bytes < need)) sp = SplitPlan{1, (d.K + BK - 1) / BK};
  float* part = sp.splits > 1 ? (float*)ws : nullptr;
  if (stats_part != nullptr && (part != nullptr || !dense_out || BM != 128 || BN != 128)) return LADDER_E_SHAPE;   // statistics come from the single-pass 128x128 epilogue
  dim3 grid(tiles_m * tiles_n, sp.splits), block(kThreads);
  if (veca && vecb) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, st, x, w, bias, y, d, tiles_n, fast, part, sp.cps, part ? nullptr : gate, gate_act, stats_part);
  else if (veca) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, st, x, w, bias, y, d, tiles_n, fast, part, sp.cps, part ? nullptr : gate, gate_act, stats_part);
  else if (vecb) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, st, x, w, bias, y, d, tiles_n, fast, part, sp.cps, part ? nullptr : gate, gate_act, stats_part);
  else hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, st, x, w, bias, y, d, tiles_n, fast, part, sp.cps, part ? nullptr : gate, gate_act, stats_part);
  if (part != nullptr) {
    const size_t mn = (size_t)d.M * d.Cout;
    if (dense_out)
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, y,
                         sp.splits, mn, d.Cout, d.act, gate, gate_act);
    else
      hipLaunchKernelGGL(splitk_epilogue_strided_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, y,
                         sp.splits, mn, d.act, gate, gate_act, d);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// tile selection: BM*1000 + BN
int select_fwd_tile(long M, int Cout) {
  const long big_tiles = ((M + 127) / 128) * ((Cout + 127) / 128);
  if (Cout > 64) {
    if (big_tiles >= 192) return 128128;
    if (M > 32) return 64064;
    return 32128;
  }
  if (Cout > 32) return ((M + 127) / 128 >= 192) ? 128064 : 64064;
  return 128032;
}


// `small_ok`: a few 128x128 tiles can still fill the chip through split-K (the 8x8 ... 1x1 ends of the CelebA encoder / decoder:
// M = 128 ... 8192 pixels, K = 256 ... 8192; strided outputs -- the parity classes of a stride-2 backward-data -- go through
// splitk_epilogue_strided_kernel)
bool split_gather_ok(const IgemmDesc& d, bool small_ok = false) {
  if (!(d.ntaps > 0 && d.ntaps <= 28 && d.ups == 1 && (d.Cin % GS_BK) == 0 && d.M > 0)) return false;
  if (select_fwd_tile(d.M, d.Cout) == 128128) return true;
  return small_ok && d.M >= 128 && d.Cout >= 128;
}

// Split-K plan of the split gather kernel (32-deep chunks).  Measured at B = 128: with >= 384 tiles (two workgroups per CU, 75 % full)
// the partial traffic costs more than the idle CUs (dec.conv4, 512 tiles: 252 TF with 2 splits, 267 without; the stride-2 encoder layer
// 172 vs 218); around 256 tiles a LONG reduction still gains from 4 splits (dec.conv3, K = 4608: 237 vs 189 TF) while a short one (a
// parity class of a stride-2 backward-data, K <= 1024) does not; a split keeps at least 8 chunks (K = 256).
SplitPlan plan_splitk32(const IgemmDesc& d) {
  const int nchunks = d.ntaps * (d.Cin / GS_BK);
  const long tiles = (long)((d.M + GS_BM - 1) / GS_BM) * ((d.Cout + GS_BN - 1) / GS_BN);
  SplitPlan p{1, nchunks};
  if (tiles >= 384 || nchunks < 16 || (tiles >= 192 && nchunks < 64)) return p;
  long s = (1024 + tiles - 1) / tiles;
  if (s > nchunks / 8) s = nchunks / 8;
  if (s > 32) s = 32;
  if (s < 2) return p;
  p.cps = (int)((nchunks + s - 1) / s);
  p.splits = (nchunks + p.cps - 1) / p.cps;
  return p;
}

size_t fwd_split_ws_bytes(const IgemmDesc& d) {
  const SplitPlan sp = plan_splitk32(d);
  return sp.splits > 1 ? (size_t)sp.splits * d.M * d.Cout * sizeof(float) : 0;
}

int launch_fwd_split(const void* x, const float* xamax, const void* packed, const float* bias, float* y, const IgemmDesc& d, int prec,
                     void* ws, size_t ws_bytes, hipStream_t st, const float* gate, int gate_act, float* stats_part = nullptr) {
  // `x` = the pre-split planes of the gathered tensor (ladder_presplit), plane-major, d.N*d.H*d.W*d.Cin elements per plane
  const bool dense_out = d.out_sh == 1 && d.out_sw == 1 && d.OH == d.Ho && d.OW == d.Wo;
  if (!split_gather_ok(d, true) || !prec_ok(prec)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(packed) || !ladder_aligned16(y)) return LADDER_E_ALIGN;
  const size_t plane_elems = (size_t)d.N * d.H * d.W * d.Cin;
  if (prec == LADDER_PREC_F16X3 && xamax == nullptr) return LADDER_E_SHAPE;
  const int tiles_m = (d.M + GS_BM - 1) / GS_BM, tiles_n = (d.Cout + GS_BN - 1) / GS_BN;
  SplitPlan sp = plan_splitk32(d);
  const size_t need = (size_t)sp.splits * d.M * d.Cout * sizeof(float);
  if (sp.splits > 1 && (ws == nullptr || ws_bytes < need)) sp = SplitPlan{1, d.ntaps * (d.Cin / GS_BK)};
  float* part = sp.splits > 1 ? (float*)ws : nullptr;
  if (stats_part != nullptr && (part != nullptr || !dense_out)) return LADDER_E_SHAPE;   // statistics come from the single-pass epilogue
  const float* wamax = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed) + pack_payload_bytes(d.KH * d.KW, d.Cin, d.Cout, prec));
  const dim3 grid(tiles_m * tiles_n, sp.splits), block(kThreads);
#define LADDER_GS_LAUNCH(P_) \
  hipLaunchKernelGGL(igemm_fwd_split_kernel<P_>, grid, block, 0, st, (const uint16_t*)x, plane_elems, (const uint4*)packed, bias, y, d, tiles_n, part, sp.cps, \
                     part ? nullptr : gate, gate_act, xamax, wamax, stats_part)
  if (prec == LADDER_PREC_F16X3) LADDER_GS_LAUNCH(LADDER_PREC_F16X3);
  else if (prec == LADDER_PREC_BF16X6) LADDER_GS_LAUNCH(LADDER_PREC_BF16X6);
  else LADDER_GS_LAUNCH(LADDER_PREC_BF16X3);
#undef LADDER_GS_LAUNCH
  if (part != nullptr) {
    const size_t mn = (size_t)d.M * d.Cout;
    if (dense_out)
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, y,
                         sp.splits, mn, d.Cout, d.act, gate, gate_act);
    else
      hipLaunchKernelGGL(splitk_epilogue_strided_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, (const float*)part, bias, y,
                         sp.splits, mn, d.act, gate, gate_act, d);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// ---- direct convolution from a 3-channel tensor: backward-data of the CelebA output conv (1x1, 3 -> 128 channels) ----------------
// K = 3 is far too short for an MFMA K-loop; the call is bound by reading the activation-derivative gate and writing dx (1.07 GB
// each at B=128): the generic gather kernel ran it at 2.2 TB/s, this one at 4.6 TB/s.  One thread owns 4 output channels, holds
// their K x 4 filter taps in registers for the whole launch and walks output pixels grid-stride; the lanes of a pixel read the same
// input words (broadcast, L1-resident) and store one contiguous Cout*4-byte row.  Accumulation order = (kh, kw, ci).
// (A 3x3 / Cin=3 instantiation for enc.conv2d forward measured 176 us against 133 us for the MFMA gather kernel -- VALU-bound on
// its 27 broadcast loads + 108 FMAs per thread -- so the forward keeps the gather kernel.)
template <int KH_, int KW_, int CIN, bool GATE>
__global__ __launch_bounds__(256) void conv_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            const float* __restrict__ gate, int H, int W, int Wo, int Cout,
                                                            int stride, int pad_t, int pad_l, int act, int M, FastDiv div_howo,
                                                            FastDiv div_wo, int HoWo) {
  constexpr int K = KH_ * KW_ * CIN;
  const int cq = Cout >> 2;                         // channel quads per pixel (power of two, 4..64)
  const int q = threadIdx.x & (cq - 1);
  const int ppb = 256 / cq;                         // pixels per block step
  const int pl = threadIdx.x / cq;
  float4 wr[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (size_t)k * Cout + q * 4);
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias != nullptr) b4 = *reinterpret_cast<const float4*>(bias + q * 4);
  for (int m = blockIdx.x * ppb + pl; m < M; m += gridDim.x * ppb) {
    const int n = (int)fdiv((uint32_t)m, div_howo);
    const int r = m - n * HoWo;
    const int oh = (int)fdiv((uint32_t)r, div_wo), ow = r - oh * Wo;
    const int ih0 = oh * stride - pad_t, iw0 = ow * stride - pad_l;
    const float* xn = x + (size_t)n * H * W * CIN;
    float4 acc = b4;
#pragma unroll
    for (int kh = 0; kh < KH_; ++kh) {
      const int ih = ih0 + kh;
      const bool okh = (unsigned)ih < (unsigned)H;
#pragma unroll
      for (int kw = 0; kw < KW_; ++kw) {
        const int iw = iw0 + kw;
        const bool ok = okh && (unsigned)iw < (unsigned)W;
        const float* px = xn + ((ok ? ih : 0) * W + (ok ? iw : 0)) * CIN;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float v = ok ? px[ci] : 0.f;
          const float4 ww = wr[(kh * KW_ + kw) * CIN + ci];
          acc.x = fmaf(v, ww.x, acc.x);
          acc.y = fmaf(v, ww.y, acc.y);
          acc.z = fmaf(v, ww.z, acc.z);
          acc.w = fmaf(v, ww.w, acc.w);
        }
      }
    }
    const size_t o = (size_t)m * Cout + q * 4;
    if (GATE) {                                     // dx *= act'(y_prev) evaluated from the layer-below's OUTPUT
      const float4 g = *reinterpret_cast<const float4*>(gate + o);
      acc.x *= ladder_act_grad_from_out(g.x, act);
      acc.y *= ladder_act_grad_from_out(g.y, act);
      acc.z *= ladder_act_grad_from_out(g.z, act);
      acc.w *= ladder_act_grad_from_out(g.w, act);
    } else {
      acc.x = ladder_act_fn(acc.x, act);
      acc.y = ladder_act_fn(acc.y, act);
      acc.z = ladder_act_fn(acc.z, act);
      acc.w = ladder_act_fn(acc.w, act);
    }
    *reinterpret_cast<float4*>(y + o) = acc;
  }
}

bool smallcin_eligible(int Cin, int Cout, int KH, int KW) {
  static const bool off = getenv("LADDER_DISABLE_SMALLCIN") != nullptr;
  const int cq = Cout >> 2;
  return !off && KH == 1 && KW == 1 && Cin == 3 && (Cout & 3) == 0 && cq >= 4 && cq <= 64 && (cq & (cq - 1)) == 0;
}

int launch_smallcin(const float* x, const float* w, const float* bias, float* y, const float* gate, int N, int H, int W, int Cin,
                    int Ho, int Wo, int Cout, int KH, int stride, int pad_t, int pad_l, int act, hipStream_t st) {
  if (!ladder_aligned16(w) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias)) ||
      (gate != nullptr && !ladder_aligned16(gate)))
    return LADDER_E_ALIGN;
  const long Ml = (long)N * Ho * Wo;
  if (Ml >= (1L << 31)) return LADDER_E_SHAPE;
  const int M = (int)Ml, ppb = 256 / (Cout >> 2);
  long blocks = (Ml + ppb - 1) / ppb;
  if (blocks > 256L * 16) blocks = 256L * 16;        // two residency rounds of the chip, grid-stride beyond that
  const FastDiv dhw = make_fastdiv(Ho * Wo), dw = make_fastdiv(Wo);
#define LADDER_SMALLCIN(KH_, KW_, CIN_, G_)                                                                                      \
  hipLaunchKernelGGL((conv_smallcin_kernel<KH_, KW_, CIN_, G_>), dim3((unsigned)blocks), dim3(256), 0, st, x, w, bias, y, gate, H, W, \
                     Wo, Cout, stride, pad_t, pad_l, act, M, dhw, dw, Ho * Wo)
  if (KH != 1 || Cin != 3) return LADDER_E_SHAPE;
  if (gate != nullptr) LADDER_SMALLCIN(1, 1, 3, true); else LADDER_SMALLCIN(1, 1, 3, false);
#undef LADDER_SMALLCIN
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// ---- single-output-channel 5x5 VALID convolution: the MNIST decoders' output layer (codes/models.py:141-148, 308-315) -----------
// [B,32,32,Cin] -> [B,28,28,1] is a per-pixel dot product of length 25*Cin: as a GEMM it wastes 31/32 of an N tile (329 us forward,
// 381 us filter gradient at B = 256, Cin = 64 -- a fifth of the MNIST-fashion iteration).  Direct form: lane = input channel, a
// wavefront walks one output row (64/Cin rows when Cin < 64) left to right keeping the KH x KW input window in registers -- each
// step loads ONE new column (KH loads) instead of KH*KW -- and reduces over the channel lanes with shuffles.  The filter gradient
// uses the same sliding window with dy as the broadcast scalar and 25 accumulators per lane.
template <int KH_, int KW_, int CL, bool WGRAD>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         const float* __restrict__ dy, float* __restrict__ part, int H, int W, int Ho,
                                                         int Wo, int act, int rows_total) {
  constexpr int RPW = 64 / CL, KT = KH_ * KW_;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ci = lane % CL, rg = lane / CL;
  float wr[KT], acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    wr[t] = WGRAD ? 0.f : w[t * CL + ci];
    acc[t] = 0.f;
  }
  const float b = (!WGRAD && bias != nullptr) ? bias[0] : 0.f;
  float bsum = 0.f;
  for (int rb = (blockIdx.x * 4 + wv) * RPW; rb < rows_total; rb += gridDim.x * 4 * RPW) {
    const int row = rb + rg;
    const bool valid = row < rows_total;
    const int rc = valid ? row : rows_total - 1;
    const int n = rc / Ho, oh = rc - n * Ho;
    const float* xr = x + ((size_t)n * H + oh) * W * CL + ci;         // input row oh (pad 0), column 0
    float win[KH_][KW_];
#pragma unroll
    for (int kh = 0; kh < KH_; ++kh)
#pragma unroll
      for (int kw = 0; kw < KW_ - 1; ++kw) win[kh][kw] = xr[((size_t)kh * W + kw) * CL];
    for (int ow0 = 0; ow0 < Wo; ow0 += KW_) {
#pragma unroll
      for (int p = 0; p < KW_; ++p) {
        const int ow = ow0 + p;
        if (ow < Wo) {                                                 // wave-uniform
#pragma unroll
          for (int kh = 0; kh < KH_; ++kh) win[kh][(p + KW_ - 1) % KW_] = xr[((size_t)kh * W + ow + KW_ - 1) * CL];
          if (WGRAD) {
            const float g = valid ? dy[(size_t)rc * Wo + ow] : 0.f;
#pragma unroll
            for (int kh = 0; kh < KH_; ++kh)
#pragma unroll
              for (int kw = 0; kw < KW_; ++kw) acc[kh * KW_ + kw] = fmaf(win[kh][(p + kw) % KW_], g, acc[kh * KW_ + kw]);
            bsum += g;
          } else {
            float s = 0.f;
#pragma unroll
            for (int kh = 0; kh < KH_; ++kh)
#pragma unroll
              for (int kw = 0; kw < KW_; ++kw) s = fmaf(win[kh][(p + kw) % KW_], wr[kh * KW_ + kw], s);
#pragma unroll
            for (int o = CL / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (ci == 0 && valid) y[(size_t)row * Wo + ow] = ladder_act_fn(s + b, act);
          }
        }
      }
    }
  }
  if (WGRAD) {
    // block partial [KT][CL] (+ bias): sum the 4 waves x RPW row groups in LDS in a fixed order
    __shared__ float red[4][64];
    float* out = part + (size_t)blockIdx.x * (KT * CL + 1);
    for (int t = 0; t <= KT; ++t) {
      red[wv][lane] = t < KT ? acc[t] : bsum;
      __syncthreads();
      if (threadIdx.x < CL) {
        float v = 0.f;
        for (int q = 0; q < 4; ++q)
          for (int g2 = 0; g2 < RPW; ++g2) v += red[q][g2 * CL + threadIdx.x];
        if (t < KT) out[t * CL + threadIdx.x] = v;
        else if (threadIdx.x == 0) out[KT * CL] = v;
      }
      __syncthreads();
    }
  }
}

bool cout1_eligible(int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int H, int W, int Ho, int Wo) {
  static const bool off = getenv("LADDER_DISABLE_COUT1") != nullptr;
  return !off && Cout == 1 && KH == 5 && KW == 5 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == H - 4 && Wo == W - 4 &&
         (Cin == 16 || Cin == 32 || Cin == 64);
}
int cout1_blocks(int rows, int Cin) {
  const int rpb = 4 * (64 / Cin);
  int b = (rows + rpb - 1) / rpb;
  return b > 1024 ? 1024 : b;
}
size_t cout1_wgrad_ws_bytes(int N, int Ho, int Cin) { return (size_t)cout1_blocks(N * Ho, Cin) * (25 * Cin + 1) * sizeof(float); }

// ---- 1x1 convolution to <= 4 channels over a wide map: the CelebA output layer (codes/models.py:580-586) -----------------------
// [B,128,128,128] -> 3 channels is HBM-bound (1.07 GB in, 25 MB out).  Forward: 32 lanes share a pixel (lane = channel quad, one
// coalesced 512-byte row), each lane keeps its 4 x COUT filter taps in registers, partial dot products are combined with shuffles.
// Backward: ONE pass over x produces all three results -- dx = (dy . W^T) * act'(x) (x is the producing layer's output, so the
// activation-derivative gate costs no extra read), dW += x^T dy and db += dy -- where separate backward-data and filter-gradient
// kernels each streamed the 1.07 GB tensor (0.44 + 0.32 ms -> 0.45 ms).
template <int COUT, bool BWD>
__global__ __launch_bounds__(256) void conv1x1_smallcout_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                const float* __restrict__ dy, float* __restrict__ dx,
                                                                float* __restrict__ part, int Cin, int act, int M,
                                                                float* __restrict__ dxamax, int rows_per_sample) {
  float dmax = 0.f;                                 // max |dx| written by this thread (backward, optional absmax record)
  // per-sample record of dx (rows_per_sample > 0): gridDim.x = N * bps workgroups, the bps workgroups of a sample sweep ITS pixels
  // grid-stride (128 concurrent 64 KB stripes at batch 128; one contiguous run per workgroup measured 13 % slower), one commit each
  const bool ps_rec = BWD && dxamax != nullptr && rows_per_sample > 0;
  const int bps = ps_rec ? gridDim.x / (M / rows_per_sample) : 1;
  const int ps_n = ps_rec ? blockIdx.x / bps : 0, ps_bx = ps_rec ? blockIdx.x - ps_n * bps : 0;
  const int cq = Cin >> 2;                          // channel quads per pixel (power of two, 4..64)
  const int q = threadIdx.x & (cq - 1), pl = threadIdx.x / cq, ppb = 256 / cq;
  float4 wr[COUT];                                  // wr[o] = w[4q..4q+3][o]
#pragma unroll
  for (int o = 0; o < COUT; ++o)
    wr[o] = make_float4(w[(size_t)(4 * q + 0) * COUT + o], w[(size_t)(4 * q + 1) * COUT + o], w[(size_t)(4 * q + 2) * COUT + o],
                        w[(size_t)(4 * q + 3) * COUT + o]);
  float4 gw[COUT];
  float gb[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    gw[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb[o] = 0.f;
  }
  constexpr int U = 1;                              // (U = 4 pixels in flight per thread measured slower: 553 vs 443 us backward)
  const int per_round = (ps_rec ? bps : gridDim.x) * ppb;
  const int steps = ((ps_rec ? rows_per_sample : M) + per_round * U - 1) / (per_round * U);
  for (int it = 0; it < steps; ++it) {              // uniform trip count: the shuffles below need whole wavefronts
    int mm[U];
    float4 xu[U];

#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ps_rec) {
        const int r = (it * bps + ps_bx) * ppb + pl;               // pixel inside the sample
        mm[u] = r < rows_per_sample ? ps_n * rows_per_sample + r : M;
      } else {
        mm[u] = ((it * U + u) * gridDim.x + blockIdx.x) * ppb + pl;
      }
      xu[u] = mm[u] < M ? *reinterpret_cast<const float4*>(x + (size_t)mm[u] * Cin + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mm[u];
      const bool ok = m < M;
      const float4 xv = xu[u];
      if (!BWD) {
        float s[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          s[o] = fmaf(xv.x, wr[o].x, fmaf(xv.y, wr[o].y, fmaf(xv.z, wr[o].z, xv.w * wr[o].w)));
          for (int d = cq >> 1; d > 0; d >>= 1) s[o] += __shfl_xor(s[o], d, 64);
        }
        if (q == 0 && ok) {
#pragma unroll
          for (int o = 0; o < COUT; ++o) y[(size_t)m * COUT + o] = ladder_act_fn(s[o] + (bias != nullptr ? bias[o] : 0.f), act);
        }
      } else {
        float g[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) g[o] = ok ? dy[(size_t)m * COUT + o] : 0.f;
        float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          d4.x = fmaf(g[o], wr[o].x, d4.x); d4.y = fmaf(g[o], wr[o].y, d4.y);
          d4.z = fmaf(g[o], wr[o].z, d4.z); d4.w = fmaf(g[o], wr[o].w, d4.w);
          gw[o].x = fmaf(xv.x, g[o], gw[o].x); gw[o].y = fmaf(xv.y, g[o], gw[o].y);
          gw[o].z = fmaf(xv.z, g[o], gw[o].z); gw[o].w = fmaf(xv.w, g[o], gw[o].w);
          gb[o] += g[o];
        }
        if (ok && dx != nullptr) {
          d4.x *= ladder_act_grad_from_out(xv.x, act); d4.y *= ladder_act_grad_from_out(xv.y, act);
          d4.z *= ladder_act_grad_from_out(xv.z, act); d4.w *= ladder_act_grad_from_out(xv.w, act);
          *reinterpret_cast<float4*>(dx + (size_t)m * Cin + q * 4) = d4;
          dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(d4.x), fabsf(d4.y))), fmaxf(fabsf(d4.z), fabsf(d4.w)));
        }
      }
    }
  }
  if (BWD) {
    // block partial: [Cin][COUT] filter gradient + [COUT] bias gradient, pixel lanes combined in LDS in a fixed order
    __shared__ float red[256];
    float* out = part + (size_t)blockIdx.x * ((size_t)Cin * COUT + COUT);
#pragma unroll
    for (int t = 0; t < 4 * COUT + COUT; ++t) {
      float v;
      if (t < 4 * COUT) {
        const float4 gv = gw[t >> 2];
        const int e = t & 3;
        v = e == 0 ? gv.x : (e == 1 ? gv.y : (e == 2 ? gv.z : gv.w));
      } else {
        v = (q == 0) ? gb[t - 4 * COUT] : 0.f;
      }
      red[threadIdx.x] = v;
      __syncthreads();
      if (pl == 0) {
        float a = 0.f;
        for (int r = 0; r < ppb; ++r) a += red[r * cq + q];
        if (t < 4 * COUT) out[(size_t)(4 * q + (t & 3)) * COUT + (t >> 2)] = a;
        else if (q == 0) out[(size_t)Cin * COUT + (t - 4 * COUT)] = a;
      }
      __syncthreads();
    }
    if (ps_rec) amax_commit_block_sample(dmax, dxamax, ps_n);
    else if (dxamax != nullptr) amax_commit_block(dmax, dxamax);
  }
}

bool smallcout_eligible(int Cin, int Cout, int KH, int KW, int stride, long M) {
  static const bool off = getenv("LADDER_DISABLE_SMALLCOUT") != nullptr;
  const int cq = Cin >> 2;
  return !off && KH == 1 && KW == 1 && stride == 1 && Cout >= 1 && Cout <= 4 && (Cin & 3) == 0 && cq >= 4 && cq <= 64 &&
         (cq & (cq - 1)) == 0 && M >= 65536 && M < (1L << 31);
}
int smallcout_blocks(long M, int Cin) {
  const long ppb = 256 / (Cin >> 2);
  long b = (M + ppb - 1) / ppb;
  return (int)(b > 2048 ? 2048 : b);
}

int dispatch_fwd(const float* x, const float* w, const float* bias, float* y, const IgemmDesc& d, void* ws, size_t ws_bytes,
                 hipStream_t st, const float* gate = nullptr, int gate_act = 0) {
  if (d.M <= 0 || d.K <= 0 || d.Cout <= 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(w) || !ladder_aligned16(y)) return LADDER_E_ALIGN;
  if (gate == nullptr && halo_eligible(d)) return launch_halo(x, w, bias, y, d, st);
  switch (select_fwd_tile(d.M, d.Cout)) {
    case 128128: return launch_fwd<128, 128, 2, 2>(x, w, bias, y, d, ws, ws_bytes, st, gate, gate_act);
    case 64064: return launch_fwd<64, 64, 2, 2>(x, w, bias, y, d, ws, ws_bytes, st, gate, gate_act);
    case 32128: return launch_fwd<32, 128, 1, 4>(x, w, bias, y, d, ws, ws_bytes, st, gate, gate_act);
    case 128064: return launch_fwd<128, 64, 4, 1>(x, w, bias, y, d, ws, ws_bytes, st, gate, gate_act);
    default: return launch_fwd<128, 32, 4, 1>(x, w, bias, y, d, ws, ws_bytes, st, gate, gate_act);
  }
}

size_t fwd_ws_bytes(long M, int K, int Cout) {
  const int t = select_fwd_tile(M, Cout);
  const SplitPlan sp = plan_splitk(M, K, Cout, t / 1000, t % 1000);
  size_t need = sp.splits > 1 ? (size_t)sp.splits * M * Cout * sizeof(float) : 0;
  // the call may be a stride-2 backward-data: its parity classes (a quarter of the pixels, at most 4 of 9 taps) split on their own
  if ((K % 9) == 0 && M >= 4) {
    const long Mc = (M + 3) / 4;
    const int Kc = K / 9 * 4, tc = select_fwd_tile(Mc, Cout);
    const SplitPlan sc = plan_splitk(Mc, Kc, Cout, tc / 1000, tc % 1000);
    const size_t nc = sc.splits > 1 ? (size_t)sc.splits * Mc * Cout * sizeof(float) : 0;
    if (nc > need) need = nc;
  }
  return need;
}

// ---- wgrad planning (shared by the workspace query and the launcher)
struct WgradPlan { int bm, bn, tiles_k, tiles_n, splits, m_per_split; };
WgradPlan plan_wgrad(long M, int K, int Cout) {
  WgradPlan p;
  p.bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
  p.bm = (K > 64 || p.bn != 64) ? 128 : 64;
  if (p.bn == 128 && K <= 64) { p.bm = 64; p.bn = 64; }
  p.tiles_k = (K + p.bm - 1) / p.bm;
  p.tiles_n = (Cout + p.bn - 1) / p.bn;
  const long tiles = (long)p.tiles_k * p.tiles_n;
  // fill whole rounds of the chip: 256 CUs x 3 resident workgroups; two rounds when the reduction is long enough
  const long slots = 256 * 3;
  long s = (2 * slots) / tiles;
  const long max_s = (M + 255) / 256;              // at least 256 pixels (16 chunks) per split
  if (s > max_s) s = max_s;
  if (s * tiles > slots && s * tiles < 2 * slots) s = slots / tiles;   // avoid a partially filled second round
  // the partial sums are written and re-read once: keep them <= ~48 MB (one round of the chip is enough for wide filters)
  if (s * tiles > slots && (size_t)s * K * Cout * sizeof(float) > ((size_t)48 << 20)) s = slots / tiles > 0 ? slots / tiles : 1;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  long mps = (M + s - 1) / s;
  mps = (mps + BK - 1) / BK * BK;
  p.m_per_split = (int)mps;
  p.splits = (int)((M + mps - 1) / mps);
  return p;
}

template <int BM, int BN, int WM, int WN>
int launch_wgrad(const float* x, const float* dy, float* out, float* bias_part, const IgemmDesc& d, const WgradPlan& p, hipStream_t st) {
  const bool veca = (d.Cin % 4) == 0, vecb = (d.Cout % 4) == 0;
  dim3 grid(p.tiles_k * p.tiles_n, p.splits), block(kThreads);
  if (veca && vecb) hipLaunchKernelGGL((igemm_wgrad_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, st, x, dy, out, bias_part, d, p.tiles_n, p.m_per_split);
  else if (veca) hipLaunchKernelGGL((igemm_wgrad_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, st, x, dy, out, bias_part, d, p.tiles_n, p.m_per_split);
  else if (vecb) hipLaunchKernelGGL((igemm_wgrad_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, st, x, dy, out, bias_part, d, p.tiles_n, p.m_per_split);
  else hipLaunchKernelGGL((igemm_wgrad_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, st, x, dy, out, bias_part, d, p.tiles_n, p.m_per_split);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Filter gradient of a 3x3 / stride 1 / SAME convolution with an LDS-staged input halo patch (tap-fused).
// One workgroup = 12 wavefronts (a balanced 3 per SIMD) owns a (64 input channels x 128 output channels) slab of dW for all 9
// taps = 72 MFMA tiles of 32x32, reduced over a range of 1x32-pixel patches.  Wavefront -> (filter row r, input-channel half cb,
// output-channel pair njp); its 6 tiles are the taps (r, 0..2) x the 2 channel blocks of the pair: per 2-pixel k-step it reads
// 3 halo fragments (one per column shift) and 2 dY fragments and issues 6 MFMAs -- 0.83 LDS reads per MFMA (the first version
// of this kernel: 2.0, 111 TF/s; shared-dY assignment: 1.33, 119 TF/s).  Per patch the 3x(32+2) input halo (64 channels, 26 KB)
// and the 32x128 dY tile (16 KB) are brought into LDS ONCE by LDS-DMA (global_load_lds, no staging registers) and serve all 9
// taps; the generic kernel streams both operands once per tap.  A fragment = halo shifted by (r,s), lane = ci; B = dY rows.
// Patch shapes (round 4): 1 x 32 pixels for maps at least 32 wide; 2 x 16 and 4 x 8 for the 16 x 16 and 8 x 8 decoder maps (the halo of a
// squarer patch is smaller: 72 / 60 halo pixels per 32 against 102) -- the generic kernel streamed both operands once per tap there
// (~88 TF against ~125 TF here).  The pixel index of a k-step maps to (row, column) of the patch at compile time (PW is a power of two).
constexpr int WH_CI = 64, WH_CO = 128, WH_PIX = 32;
constexpr int WH_THREADS = 768;
// Stride 2 (S = 2; SAME padding of an even map = one zero line below / right, none above / left): the input footprint of a patch is
// (2 PH + 1) x (2 PW + 1) pixels, tap (r, s) of output pixel (i, j) reads footprint pixel (2 i + r, 2 j + s) -- the same kernel with a
// pixel stride of 2 in the fragment addresses (encoder conv2d_1 ... 3: 522 / 251 / 138 us on the generic kernel at batch 128).
template <int PH, int PW, int S> struct WhGeom {
  static constexpr int HW = S * PW + 3 - S, HH = S * PH + 3 - S;
  static constexpr int XU = HH * HW * (WH_CI / 4), DU = WH_PIX * (WH_CO / 4);             // float4 units per patch
  static constexpr int XN = (XU + WH_THREADS - 1) / WH_THREADS, DN = (DU + WH_THREADS - 1) / WH_THREADS;   // 3 (2 for the small patches; 5 / 4 / 4 at stride 2), 2
  static constexpr int XF = HH * HW * WH_CI, DF = WH_PIX * WH_CO;                          // floats per buffer
};
#ifndef IGEMM_WH_MINW
#define IGEMM_WH_MINW 3
#endif
template <int PH, int PW, int S>
__global__ __launch_bounds__(WH_THREADS, IGEMM_WH_MINW) void wgrad3x3_halo_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ out, float* __restrict__ bias_part,
                                                                  const int N, const int H, const int W, const int Cin,
                                                                  const int Cout, const int tiles_co, const int patches_per_split) {
  // H, W: the OUTPUT map (= the input map at stride 1; the input map is S H x S W)
  using G = WhGeom<PH, PW, S>;
  static_assert(PH * PW == WH_PIX && (PW & (PW - 1)) == 0 && (PW % 2) == 0, "32-pixel patches, rows of a power of two");
  constexpr int WH_HW = G::HW, WH_XF = G::XF, WH_DF = G::DF;
  // ONE LDS object (hipcc serialises LDS-DMA against ds_reads when several __shared__ objects exist): [buf][halo | dY]
  __shared__ __attribute__((aligned(16))) float lds[2 * (WH_XF + WH_DF)];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;            // 12 wavefronts
  const int l31 = lane & 31, lh = lane >> 5;
  const int ci0 = (blockIdx.x / tiles_co) * WH_CI, co0 = (blockIdx.x % tiles_co) * WH_CO;
  const int WP = W / PW, HP = H / PH;
  const int q_total = N * HP * WP;
  const int q0 = blockIdx.y * patches_per_split, q1 = min(q_total, q0 + patches_per_split);
  const int r = wv >> 2, cb = (wv >> 1) & 1, njp = wv & 1;
  const int a_off = (r * WH_HW + S * lh) * WH_CI + cb * 32 + l31;      // + s * WH_CI for the column shift s
  const int b_off = WH_XF + lh * WH_CO + njp * 64 + l31;               // + j * 32 for the second block of the pair
  // bias: the centre-tap tiles (r = 1, s = 1) of the cb = 0 wavefronts cover every output channel exactly once.
  const bool do_bias = (bias_part != nullptr) && (ci0 == 0) && r == 1 && cb == 0;

  f32x16 acc[6];                                       // [j][s]
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float bsum[2] = {0.f, 0.f};

  auto stage_patch = [&](int q, int buf) {
    const int n = q / (HP * WP), rem = q - n * (HP * WP);
    const int hp = rem / WP, h0 = hp * PH, w0 = (rem - hp * WP) * PW;
    float* xb = &lds[buf * (WH_XF + WH_DF)];
    float* db = xb + WH_XF;
#pragma unroll
    for (int i = 0; i < G::XN; ++i) {
      const int u = tid + i * WH_THREADS;
      if (u < G::XU) {
        const int pix = u >> 4, q4 = u & 15;
        const int hr = pix / WH_HW, hc = pix - hr * WH_HW;
        const int hi = S * h0 - (2 - S) + hr, wi = S * w0 - (2 - S) + hc;
        const bool ok = hi >= 0 && hi < S * H && wi >= 0 && wi < S * W;
        const float* src = ok ? x + (((long)n * (S * H) + hi) * (S * W) + wi) * Cin + ci0 + q4 * 4 : g_zero16;
        __builtin_amdgcn_global_load_lds(src, xb + u * 4, 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < G::DN; ++i) {
      const int u = tid + i * WH_THREADS;
      if (u < G::DU) {
        const int p = u >> 5, q4 = u & 31;
        const bool ok = (co0 + q4 * 4) < Cout;
        const float* src = ok ? dy + (((long)n * H + h0 + p / PW) * W + w0 + (p & (PW - 1))) * Cout + co0 + q4 * 4 : g_zero16;
        __builtin_amdgcn_global_load_lds(src, db + u * 4, 16, 0, 0);
      }
    }
  };

  if (q0 < q1) stage_patch(q0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int q = q0; q < q1; ++q) {
    const int buf = (q - q0) & 1;
    if (q + 1 < q1) stage_patch(q + 1, buf ^ 1);
    const float* base = &lds[buf * (WH_XF + WH_DF)];
#pragma unroll 4
    for (int ks = 0; ks < WH_PIX / 2; ++ks) {
      float a[3], b[2];
      const int arow = S * (((2 * ks) / PW) * WH_HW + ((2 * ks) & (PW - 1))) * WH_CI;     // pixel 2ks(+lh) of the patch -> halo row/col
#pragma unroll
      for (int s = 0; s < 3; ++s) a[s] = base[a_off + s * WH_CI + arow];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = base[b_off + j * 32 + 2 * ks * WH_CO];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[j * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[j], acc[j * 3 + s], 0, 0, 0);
      if (do_bias) {
        bsum[0] += b[0];
        bsum[1] += b[1];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's LDS-DMA pieces of the next patch have landed
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = co0 + njp * 64 + j * 32 + l31;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float* o = out + ((size_t)blockIdx.y * 9 + (3 * r + s)) * Cin * Cout;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (n < Cout) o[(size_t)ci * Cout + n] = acc[j * 3 + s][e];
      }
    }
    if (do_bias) {
      const float v = bsum[j] + __shfl_down(bsum[j], 32, 64);   // odd + even pixels of the k-step pairs
      if (lh == 0 && n < Cout) bias_part[(size_t)blockIdx.y * Cout + n] = v;
    }
  }
}

struct WgradHaloPlan { bool ok; int tiles_ci, tiles_co, splits, pps, pw, stride; };    // pw: patch width 32 / 16 / 8 (patch = 32 / pw rows)
WgradHaloPlan plan_wgrad_halo(const IgemmDesc& d) {
  WgradHaloPlan p{false, 0, 0, 1, 0, 0, 1};
  if (halo_disabled()) return p;
#if IGEMM_HALO
  static const int small_env = getenv("LADDER_WGRAD_SMALL_PATCHES") ? atoi(getenv("LADDER_WGRAD_SMALL_PATCHES")) : 1;   // (test switch)
  const int pw = (d.Wo % 32) == 0 ? 32 : (small_env && d.Wo == 16 && (d.Ho % 2) == 0) ? 16 : (small_env && d.Wo == 8 && (d.Ho % 4) == 0) ? 8 : 0;
  const bool s1 = d.stride == 1 && d.pad_t == 1 && d.pad_l == 1 && d.Ho == d.H && d.Wo == d.W;
  const bool s2 = small_env && d.stride == 2 && d.pad_t == 0 && d.pad_l == 0 && d.H == 2 * d.Ho && d.W == 2 * d.Wo;
  if (!(d.KH == 3 && d.KW == 3 && (s1 || s2) && d.ups == 1 && (d.Cin % WH_CI) == 0 && (d.Cout % 4) == 0 && d.Cout >= 64 && pw != 0))
    return p;
  p.stride = s2 ? 2 : 1;
  const long q_total = (long)d.N * d.Ho * d.Wo / WH_PIX;
  if (pw == 32 && s1 ? q_total < 4096 : q_total * (d.Cin / WH_CI) * ((d.Cout + WH_CO - 1) / WH_CO) < 4096) return p;   // small launches stay on the generic kernel
  p.pw = pw;
  p.tiles_ci = d.Cin / WH_CI;
  p.tiles_co = (d.Cout + WH_CO - 1) / WH_CO;
  const long pairs = (long)p.tiles_ci * p.tiles_co;
  long s = ((pw < 32 || s2 ? 1 : IGEMM_WH_BLOCKS) * 256L) / pairs;   // whole rounds of the chip (small maps: one -- 690 / 718 / 765 / 792 us for 1 ... 4 rounds on 16x16 512 -> 256 at batch 128: the partial tiles cost more than the tail)
  if (s > q_total / 16) s = q_total / 16;
  if (s < 1) s = 1;
  p.pps = (int)((q_total + s - 1) / s);
  p.splits = (int)((q_total + p.pps - 1) / p.pps);
  p.ok = true;
#endif
  return p;
}

size_t wgrad_ws_bytes(long M, int K, int Cout) {
  const WgradPlan p = plan_wgrad(M, K, Cout);
  const size_t a = p.splits > 1 ? (size_t)p.splits * K * Cout * sizeof(float) : 0;
  return a + (size_t)p.splits * Cout * sizeof(float);   // + per-split bias partials
}

size_t wgrad_ws_bytes_desc(const IgemmDesc& d) {
  const WgradHaloPlan hp = plan_wgrad_halo(d);
  size_t need = wgrad_ws_bytes(d.M, d.K, d.Cout);
  if (hp.ok) {
    const size_t h = ((hp.splits > 1 ? (size_t)hp.splits * d.K * d.Cout : 0) + (size_t)hp.splits * d.Cout) * sizeof(float);
    if (h > need) need = h;
  }
  return need;
}

int run_wgrad(const float* x, const float* dy, float* dw, float* db, const IgemmDesc& d, void* ws, size_t ws_bytes, hipStream_t st) {
  if (d.M <= 0 || d.K <= 0 || d.Cout <= 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(dy) || !ladder_aligned16(dw)) return LADDER_E_ALIGN;
  if (ws_bytes < wgrad_ws_bytes_desc(d) || ws == nullptr) return LADDER_E_WORKSPACE;
  const WgradHaloPlan hp = plan_wgrad_halo(d);
  if (hp.ok) {
    const size_t kn = (size_t)d.K * d.Cout;
    float* out = hp.splits > 1 ? (float*)ws : dw;
    float* bias_part = db != nullptr ? (float*)ws + (hp.splits > 1 ? (size_t)hp.splits * kn : 0) : nullptr;
    const dim3 grid(hp.tiles_ci * hp.tiles_co, hp.splits), block(WH_THREADS);
#define LADDER_WH_LAUNCH(PH_, PW_, S_) \
  hipLaunchKernelGGL((wgrad3x3_halo_kernel<PH_, PW_, S_>), grid, block, 0, st, x, dy, out, bias_part, d.N, d.Ho, d.Wo, d.Cin, d.Cout, hp.tiles_co, hp.pps)
    if (hp.stride == 1) {
      if (hp.pw == 32) LADDER_WH_LAUNCH(1, 32, 1); else if (hp.pw == 16) LADDER_WH_LAUNCH(2, 16, 1); else LADDER_WH_LAUNCH(4, 8, 1);
    } else {
      if (hp.pw == 32) LADDER_WH_LAUNCH(1, 32, 2); else if (hp.pw == 16) LADDER_WH_LAUNCH(2, 16, 2); else LADDER_WH_LAUNCH(4, 8, 2);
    }
#undef LADDER_WH_LAUNCH
    if (hp.splits > 1)
      launch_reduce_splits((const float*)ws, dw, hp.splits, kn, st);
    if (db != nullptr)
      launch_reduce_splits((const float*)bias_part, db,
                         hp.splits, (size_t)d.Cout, st);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  const WgradPlan p = plan_wgrad(d.M, d.K, d.Cout);
  const size_t kn = (size_t)d.K * d.Cout;
  float* out = p.splits > 1 ? (float*)ws : dw;
  float* bias_part = db != nullptr ? (float*)ws + (p.splits > 1 ? (size_t)p.splits * kn : 0) : nullptr;
  int rc;
  if (p.bm == 128 && p.bn == 128) rc = launch_wgrad<128, 128, 2, 2>(x, dy, out, bias_part, d, p, st);
  else if (p.bm == 128 && p.bn == 64) rc = launch_wgrad<128, 64, 4, 1>(x, dy, out, bias_part, d, p, st);
  else if (p.bm == 128 && p.bn == 32) rc = launch_wgrad<128, 32, 4, 1>(x, dy, out, bias_part, d, p, st);
  else rc = launch_wgrad<64, 64, 2, 2>(x, dy, out, bias_part, d, p, st);
  if (rc != LADDER_OK) return rc;
  if (p.splits > 1)
    launch_reduce_splits((const float*)ws, dw, p.splits, kn, st);
  if (db != nullptr)
    launch_reduce_splits((const float*)bias_part, db,
                       p.splits, (size_t)d.Cout, st);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// ---- split filter gradient: planning shared by the workspace query and the launcher
bool wgrad_split_ok(const IgemmDesc& d) {
  if (d.ups != 1 || (d.Cin % 128) != 0 || d.Cout <= 64 || (d.Cout % 8) != 0 || d.M < 128) return false;
  const WgradPlan p = plan_wgrad(d.M, d.K, d.Cout);
  return p.bm == 128 && p.bn == 128;
}
WgradPlan plan_wgrad_split32(const IgemmDesc& d) {
  WgradPlan p = plan_wgrad(d.M, d.K, d.Cout);
  // this kernel keeps 2 workgroups per CU resident: exactly ONE full round of the chip (512) measured best on every layer (dec.conv4
  // 347 us against 377 us with the fp32 kernel's 768-slot plan, enc.conv2 186 / 223 us); at least 256 pixels per split
  {
    const long tiles = (long)p.tiles_k * p.tiles_n;
    long s = 512 / tiles;
    const long max_s = (d.M + 255) / 256;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    p.m_per_split = (int)((d.M + s - 1) / s);
  }
  long mps = (p.m_per_split + 31) / 32 * 32;           // 32-pixel chunks
  p.m_per_split = (int)mps;
  p.splits = (int)((d.M + mps - 1) / mps);
  return p;
}
size_t wgrad_split_ws_bytes(const IgemmDesc& d) {
  const WgradPlan p = plan_wgrad_split32(d);
  return ((size_t)p.splits * d.K * d.Cout + (size_t)p.splits * d.Cout) * sizeof(float);
}
int run_wgrad_split(const void* x, const float* xamax, const void* dy, const float* damax, float* dw, float* db, const IgemmDesc& d,
                    int prec, void* ws, size_t ws_bytes, hipStream_t st) {
  // x, dy = pre-split planes (ladder_presplit) of the layer input [N,H,W,Cin] and of the output gradient [N,Ho,Wo,Cout]
  if (!wgrad_split_ok(d) || !prec_ok(prec)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(dy) || !ladder_aligned16(dw)) return LADDER_E_ALIGN;
  if (prec == LADDER_PREC_F16X3 && (xamax == nullptr || damax == nullptr)) return LADDER_E_SHAPE;
  if (ws == nullptr || ws_bytes < wgrad_split_ws_bytes(d)) return LADDER_E_WORKSPACE;
  const WgradPlan p = plan_wgrad_split32(d);
  const size_t kn = (size_t)d.K * d.Cout;
  float* part = (float*)ws;
  float* bias_part = db != nullptr ? part + (size_t)p.splits * kn : nullptr;
  const dim3 grid(p.tiles_k * p.tiles_n, p.splits), block(kThreads);
#define LADDER_GW_LAUNCH(P_) \
  hipLaunchKernelGGL(igemm_wgrad_split_kernel<P_>, grid, block, 0, st, (const uint16_t*)x, (size_t)d.N * d.H * d.W * d.Cin, (const uint16_t*)dy, \
                     (size_t)d.M * d.Cout, part, bias_part, d, p.tiles_n, p.m_per_split, xamax, damax)
  if (prec == LADDER_PREC_F16X3) LADDER_GW_LAUNCH(LADDER_PREC_F16X3);
  else if (prec == LADDER_PREC_BF16X6) LADDER_GW_LAUNCH(LADDER_PREC_BF16X6);
  else LADDER_GW_LAUNCH(LADDER_PREC_BF16X3);
#undef LADDER_GW_LAUNCH
  launch_reduce_splits((const float*)part, dw, p.splits, kn, st);
  if (db != nullptr) launch_reduce_splits((const float*)bias_part, db, p.splits, (size_t)d.Cout, st);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

}  // namespace

extern "C" {

int ladder_abi_version(void) { return 2; }   // == LADDER_ABI_VERSION (include/ladder_hip.h)

int ladder_dense_fwd_is_persistent(long M, int K, int N) { return dense_f32_big_ok(M, K, N) ? 1 : 0; }
int ladder_dense_bwd_weight_is_persistent(long M, int K, int N) { return dense_wgrad_f32_ok(M, K, N) ? 1 : 0; }

int ladder_igemm_fwd_tile(long M, int Cin, int Cout) {
  // BM*1000+BN of the kernel instantiation ladder_conv2d_fwd / _bwd_data / ladder_dense_* dispatch to; negative when the
  // vectorised (Cin%16==0 && Cout%4==0) instantiation is not used.  Lets a profiler attribute a launch to a kernel name.
  const int t = select_fwd_tile(M, Cout);
  return ((Cin % BK) == 0 && (Cout % 4) == 0) ? t : -t;   // (the 3x3 halo kernel is reported by ladder_conv3x3_uses_halo)
}

int ladder_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                      int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int act,
                      void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return LADDER_E_SHAPE;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, act, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  set_conv_taps(d);
  if (smallcout_eligible(Cin, Cout, KH, KW, stride, (long)N * Ho * Wo) && ladder_aligned16(x)) {
    const int M = N * Ho * Wo, blocks = smallcout_blocks(M, Cin);
#define LADDER_SCO_FWD(CO_) hipLaunchKernelGGL((conv1x1_smallcout_kernel<CO_, false>), dim3(blocks), dim3(256), 0, stream, x, w, bias, y, \
                                               (const float*)nullptr, (float*)nullptr, (float*)nullptr, Cin, act, M, (float*)nullptr, 0)
    switch (Cout) { case 1: LADDER_SCO_FWD(1); break; case 2: LADDER_SCO_FWD(2); break; case 3: LADDER_SCO_FWD(3); break; default: LADDER_SCO_FWD(4); }
#undef LADDER_SCO_FWD
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  if (cout1_eligible(Cin, Cout, KH, KW, stride, pad_t, pad_l, H, W, Ho, Wo)) {
    const int rows = N * Ho, blocks = cout1_blocks(rows, Cin);
#define LADDER_COUT1_FWD(CL_) hipLaunchKernelGGL((conv_cout1_kernel<5, 5, CL_, false>), dim3(blocks), dim3(256), 0, stream, x, w, bias, y, \
                                                 (const float*)nullptr, (float*)nullptr, H, W, Ho, Wo, act, rows)
    if (Cin == 64) LADDER_COUT1_FWD(64); else if (Cin == 32) LADDER_COUT1_FWD(32); else LADDER_COUT1_FWD(16);
#undef LADDER_COUT1_FWD
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  return dispatch_fwd(x, w, bias, y, d, ws, ws_bytes, stream);
}

size_t ladder_igemm_fwd_workspace_bytes(long M, int K, int Cout) { return fwd_ws_bytes(M, K, Cout); }

int ladder_igemm_fwd_splits(long M, int K, int Cout) {
  const int t = select_fwd_tile(M, Cout);
  return plan_splitk(M, K, Cout, t / 1000, t % 1000).splits;
}

// ---- strict-fp32 convolution whose epilogue also emits the batch-norm statistics of its output (round 4: the fp32 form of
// ladder_conv2d_fwd_split_bnstats): 128x128-tile gather launches without split-K.  0 bytes = not available for this geometry.
static bool fwd_bnstats_ok(const IgemmDesc& d) {
  if (getenv("LADDER_DISABLE_BNSTATS") != nullptr) return false;       // (test-only switch: the statistics then come from the separate pass)
  // (Cin % 16: the image-side conv, Cin = 3, is bound by writing its output -- the statistics epilogue costs it the 90 us the separate pass takes: measured, no gain)
  return (d.Cin % BK) == 0 && (d.Cout % 4) == 0 && d.ntaps > 0 && !halo_eligible(d) && select_fwd_tile(d.M, d.Cout) == 128128 &&
         plan_splitk(d.M, d.K, d.Cout, 128, 128).splits == 1 && !smallcout_eligible(d.Cin, d.Cout, d.KH, d.KW, d.stride, d.M);
}

size_t ladder_conv2d_fwd_bnstats_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                                 int pad_l) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return 0;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  set_conv_taps(d);
  if (!fwd_bnstats_ok(d)) return 0;
  return (((size_t)d.M + 127) / 128) * 4 * (size_t)Cout * sizeof(float);
}

int ladder_conv2d_fwd_bnstats(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                              int KH, int KW, int stride, int pad_t, int pad_l, int act, float* sums4, void* stats_ws, size_t stats_ws_bytes,
                              ladder_stream_t stream) {
  const size_t need = ladder_conv2d_fwd_bnstats_workspace_bytes(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l);
  if (need == 0 || sums4 == nullptr) return LADDER_E_SHAPE;
  if (stats_ws == nullptr || stats_ws_bytes < need) return LADDER_E_WORKSPACE;
  if (!ladder_aligned16(x) || !ladder_aligned16(w) || !ladder_aligned16(y)) return LADDER_E_ALIGN;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, act, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  set_conv_taps(d);
  const int rc = launch_fwd<128, 128, 2, 2>(x, w, bias, y, d, nullptr, 0, stream, nullptr, 0, (float*)stats_ws);
  if (rc != LADDER_OK) return rc;
  return ladder_bn_stats_minmax_from_partials((const float*)stats_ws, (int)(((size_t)d.M + 127) / 128), sums4, Cout, stream);
}

int ladder_conv2d_fwd_kernel_id(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int ups,
                                int pad_t, int pad_l) {
  // 256128 = conv3x3_halo_kernel (8x32-pixel x 128-channel LDS-halo tile); otherwise the code of ladder_igemm_fwd_tile
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, ups, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(1), make_fastdiv(1)};
  set_conv_taps(d);
  if (halo_eligible(d)) return 256128;
  if (ups == 1 && stride == 1 && smallcin_eligible(Cin, Cout, KH, KW)) return 9003;     // conv_smallcin_kernel (direct, output-write bound)
  return ladder_igemm_fwd_tile(d.M, Cin, Cout);
}

int ladder_reduce_splits(const float* ws, float* out, int splits, size_t n, ladder_stream_t stream) {
  if (splits <= 0 || n == 0) return LADDER_E_SHAPE;
  launch_reduce_splits(ws, out, splits, n, stream);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_filter_flip_transpose(const float* w, float* wT, int KH, int KW, int Cin, int Cout, ladder_stream_t stream) {
  if (KH <= 0 || KW <= 0 || Cin <= 0 || Cout <= 0) return LADDER_E_SHAPE;
  dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, KH * KW), block(32, 8);
  hipLaunchKernelGGL(flip_transpose_kernel, grid, block, 0, stream, w, wT, KH, KW, Cin, Cout);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_conv2d_bwd_data(const float* dy, const float* wT, float* dx, int N, int H, int W, int Cin, int Ho, int Wo,
                           int Cout, int KH, int KW, int stride, int pad_t, int pad_l, const float* gate_y, int gate_act,
                           void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return LADDER_E_SHAPE;
  // dx[hi] gathers dy[(hi + pad_t - r)/stride] = dy[(hi + r' - (KH-1-pad_t))/stride] with the flipped tap r'.
  IgemmDesc d{N, Ho, Wo, Cout, H, W, Cin, KH, KW, 1, stride, KH - 1 - pad_t, KW - 1 - pad_l, N * H * W, KH * KW * Cout, LADDER_ACT_NONE, make_fastdiv(H * W), make_fastdiv(W)};
  set_conv_taps(d);   // stride 1: full tap table; stride > 1: dense output mapping only (ntaps = 0 -> generic gather)
  if (stride == 1 && smallcin_eligible(Cout, Cin, KH, KW))             // 1x1 from a 3-channel dy (CelebA output conv), gate fused
    return launch_smallcin(dy, wT, nullptr, dx, gate_y, N, Ho, Wo, Cout, H, W, Cin, KH, 1, 0, 0, gate_y != nullptr ? gate_act : 0, stream);
  if (stride == 1) return dispatch_fwd(dy, wT, nullptr, dx, d, ws, ws_bytes, stream, gate_y, gate_act);
  if (stride == 2 && (Cout % BK) == 0 && KH * KW <= 28) {
    // Parity-class decomposition of the transposed convolution: output pixels with (hi, wi) parity (ch, cw) only ever see the
    // flipped taps r' = (pad' + ch) mod 2 (+2), so each class is a dense stride-1 gather over ~1/4 of the taps written to every
    // other output pixel -- no zero taps enter the MFMA pipeline.
    const int padh = KH - 1 - pad_t, padw = KW - 1 - pad_l;
    for (int ch = 0; ch < 2; ++ch)
      for (int cw = 0; cw < 2; ++cw) {
        const int Hc = (H - ch + 1) / 2, Wc = (W - cw + 1) / 2;
        if (Hc <= 0 || Wc <= 0) continue;
        IgemmDesc c = d;
        c.Ho = Hc; c.Wo = Wc; c.stride = 1; c.ups = 1; c.pad_t = 0; c.pad_l = 0;
        c.M = N * Hc * Wc;
        c.div_howo = make_fastdiv(Hc * Wc);
        c.div_wo = make_fastdiv(Wc);
        c.out_sh = c.out_sw = 2; c.out_h0 = ch; c.out_w0 = cw; c.OH = H; c.OW = W;
        c.ntaps = 0;
        for (int r = 0; r < KH; ++r) {
          if (((r + padh + ch) & 1) != 0) continue;
          for (int sx = 0; sx < KW; ++sx) {
            if (((sx + padw + cw) & 1) != 0) continue;
            c.tap_dh[c.ntaps] = (signed char)((ch + r - padh) / 2);     // exact: numerator is even
            c.tap_dw[c.ntaps] = (signed char)((cw + sx - padw) / 2);
            c.tap_w[c.ntaps] = (signed char)(r * KW + sx);
            ++c.ntaps;
          }
        }
        if (c.ntaps == 0) return dispatch_fwd(dy, wT, nullptr, dx, d, ws, ws_bytes, stream, gate_y, gate_act);   // degenerate: legacy path writes the zeros
        c.K = c.ntaps * Cout;
        const int rc = dispatch_fwd(dy, wT, nullptr, dx, c, ws, ws_bytes, stream, gate_y, gate_act);   // (the classes run one after another: one workspace)
        if (rc != LADDER_OK) return rc;
      }
    return LADDER_OK;
  }
  return dispatch_fwd(dy, wT, nullptr, dx, d, ws, ws_bytes, stream, gate_y, gate_act);   // generic strided gather (ups > 1)
}

// ---- split-precision variants of the forward-type convolution calls (gather kernel; the 3x3 halo layers have their own entry points
// in convsplit.hip).  `dry` = only report whether every launch of the call would run on the split kernel.
static int conv2d_fwd_split_impl(const void* x, const float* x_absmax, const void* packed, const float* bias, float* y, int N, int H,
                                 int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int act,
                                 int prec, void* ws, size_t ws_bytes, ladder_stream_t stream, bool dry, size_t* ws_need) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return LADDER_E_SHAPE;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, act, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  set_conv_taps(d);
  if (!split_gather_ok(d, true) || halo_eligible(d)) return LADDER_E_SHAPE;
  if (ws_need != nullptr) *ws_need = fwd_split_ws_bytes(d);
  if (dry) return LADDER_OK;
  return launch_fwd_split(x, x_absmax, packed, bias, y, d, prec, ws, ws_bytes, stream, nullptr, 0);
}

static int conv2d_bwd_data_split_impl(const void* dy, const float* dy_absmax, const void* packed, float* dx, int N, int H, int W, int Cin,
                                      int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, const float* gate_y,
                                      int gate_act, int prec, void* ws, size_t ws_bytes, ladder_stream_t stream, bool dry, size_t* ws_need) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return LADDER_E_SHAPE;
  IgemmDesc d{N, Ho, Wo, Cout, H, W, Cin, KH, KW, 1, stride, KH - 1 - pad_t, KW - 1 - pad_l, N * H * W, KH * KW * Cout, LADDER_ACT_NONE, make_fastdiv(H * W), make_fastdiv(W)};
  set_conv_taps(d);
  if (ws_need != nullptr) *ws_need = 0;
  if (stride == 1) {
    if (!split_gather_ok(d, true) || (gate_y == nullptr && halo_eligible(d))) return LADDER_E_SHAPE;
    if (ws_need != nullptr) *ws_need = fwd_split_ws_bytes(d);
    if (dry) return LADDER_OK;
    return launch_fwd_split(dy, dy_absmax, packed, nullptr, dx, d, prec, ws, ws_bytes, stream, gate_y, gate_act);
  }
  if (stride != 2 || KH * KW > 28) return LADDER_E_SHAPE;
  const int padh = KH - 1 - pad_t, padw = KW - 1 - pad_l;
  for (int pass = 0; pass < 2; ++pass) {                 // pass 0: every parity class must be eligible; pass 1: launch
    for (int ch = 0; ch < 2; ++ch)
      for (int cw = 0; cw < 2; ++cw) {
        const int Hc = (H - ch + 1) / 2, Wc = (W - cw + 1) / 2;
        if (Hc <= 0 || Wc <= 0) return LADDER_E_SHAPE;
        IgemmDesc c = d;
        c.Ho = Hc; c.Wo = Wc; c.stride = 1; c.ups = 1; c.pad_t = 0; c.pad_l = 0;
        c.M = N * Hc * Wc;
        c.div_howo = make_fastdiv(Hc * Wc);
        c.div_wo = make_fastdiv(Wc);
        c.out_sh = c.out_sw = 2; c.out_h0 = ch; c.out_w0 = cw; c.OH = H; c.OW = W;
        c.ntaps = 0;
        for (int r = 0; r < KH; ++r) {
          if (((r + padh + ch) & 1) != 0) continue;
          for (int sx = 0; sx < KW; ++sx) {
            if (((sx + padw + cw) & 1) != 0) continue;
            c.tap_dh[c.ntaps] = (signed char)((ch + r - padh) / 2);
            c.tap_dw[c.ntaps] = (signed char)((cw + sx - padw) / 2);
            c.tap_w[c.ntaps] = (signed char)(r * KW + sx);
            ++c.ntaps;
          }
        }
        c.K = c.ntaps * Cout;
        if (pass == 0) {
          if (c.ntaps == 0 || !split_gather_ok(c, true)) return LADDER_E_SHAPE;
          if (ws_need != nullptr && fwd_split_ws_bytes(c) > *ws_need) *ws_need = fwd_split_ws_bytes(c);   // (classes run one after another)
        } else {
          const int rc = launch_fwd_split(dy, dy_absmax, packed, nullptr, dx, c, prec, ws, ws_bytes, stream, gate_y, gate_act);
          if (rc != LADDER_OK) return rc;
        }
      }
    if (dry) return LADDER_OK;
  }
  return LADDER_OK;
}

int ladder_conv2d_fwd_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                     int pad_l) {
  return conv2d_fwd_split_impl(nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, 0,
                               LADDER_PREC_F16X3, nullptr, 0, nullptr, true, nullptr) == LADDER_OK ? 1 : 0;
}

size_t ladder_conv2d_fwd_split_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                               int pad_t, int pad_l) {
  size_t need = 0;
  conv2d_fwd_split_impl(nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, 0,
                        LADDER_PREC_F16X3, nullptr, 0, nullptr, true, &need);
  return need;
}

int ladder_conv2d_fwd_split(const void* x_planes, const float* x_absmax, const void* packed, const float* bias, float* y, int N, int H, int W,
                            int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int act, int prec,
                            void* ws, size_t ws_bytes, ladder_stream_t stream) {
  return conv2d_fwd_split_impl(x_planes, x_absmax, packed, bias, y, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, act, prec, ws,
                               ws_bytes, stream, false, nullptr);
}

// Forward + the batch-norm statistics of its output (sums4 = the statistics record, minmax form: 2 Cout doubles sum | sum of squares, then min | max, as
// ladder_bn_fwd_stats_minmax computes them from a second pass over y) from the epilogue's per-tile column statistics.  Available when the
// call runs without split-K (workspace query > 0): the partial sums of a split reduction never see the finished values.
size_t ladder_conv2d_fwd_split_bnstats_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                                       int pad_t, int pad_l) {
  if (!ladder_conv2d_fwd_split_eligible(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l)) return 0;
  if (ladder_conv2d_fwd_split_workspace_bytes(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l) != 0) return 0;
  const size_t tiles_m = ((size_t)N * Ho * Wo + GS_BM - 1) / GS_BM;
  return tiles_m * 4 * (size_t)Cout * sizeof(float);
}

int ladder_conv2d_fwd_split_bnstats(const void* x_planes, const float* x_absmax, const void* packed, const float* bias, float* y, int N, int H,
                                    int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int act,
                                    int prec, float* sums4, void* stats_ws, size_t stats_ws_bytes, ladder_stream_t stream) {
  const size_t need = ladder_conv2d_fwd_split_bnstats_workspace_bytes(N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l);
  if (need == 0 || sums4 == nullptr) return LADDER_E_SHAPE;
  if (stats_ws == nullptr || stats_ws_bytes < need) return LADDER_E_WORKSPACE;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, act, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  set_conv_taps(d);
  const int rc = launch_fwd_split(x_planes, x_absmax, packed, bias, y, d, prec, nullptr, 0, stream, nullptr, 0, (float*)stats_ws);
  if (rc != LADDER_OK) return rc;
  return ladder_bn_stats_minmax_from_partials((const float*)stats_ws, (int)(((size_t)N * Ho * Wo + GS_BM - 1) / GS_BM), sums4, Cout, stream);
}

int ladder_conv2d_bwd_data_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                          int pad_l, int gated) {
  static const float dummy = 0.f;
  return conv2d_bwd_data_split_impl(nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l,
                                    gated ? &dummy : nullptr, 0, LADDER_PREC_F16X3, nullptr, 0, nullptr, true, nullptr) == LADDER_OK ? 1 : 0;
}

size_t ladder_conv2d_bwd_data_split_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                                    int pad_t, int pad_l) {
  size_t need = 0;
  static const float dummy = 0.f;
  conv2d_bwd_data_split_impl(nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, &dummy, 0,
                             LADDER_PREC_F16X3, nullptr, 0, nullptr, true, &need);
  return need;
}

int ladder_conv2d_bwd_data_split(const void* dy_planes, const float* dy_absmax, const void* packed_T, float* dx, int N, int H, int W, int Cin,
                                 int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, const float* gate_y,
                                 int gate_act, int prec, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  return conv2d_bwd_data_split_impl(dy_planes, dy_absmax, packed_T, dx, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l, gate_y, gate_act,
                                    prec, ws, ws_bytes, stream, false, nullptr);
}

size_t ladder_conv2d_bwd_filter_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW) {
  // upper bound over the stride/pad variants of this geometry (the halo path needs stride 1 / pad 1 / Ho==H, or stride 2 / pad 0 / H==2Ho)
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, 1, 1, 1, 1, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(1), make_fastdiv(1)};
  size_t need = wgrad_ws_bytes_desc(d);
  if (H == 2 * Ho && W == 2 * Wo) {
    IgemmDesc d2{N, H, W, Cin, Ho, Wo, Cout, KH, KW, 2, 1, 0, 0, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(1), make_fastdiv(1)};
    const size_t n2 = wgrad_ws_bytes_desc(d2);
    if (n2 > need) need = n2;
  }
  if (cout1_eligible(Cin, Cout, KH, KW, 1, 0, 0, H, W, Ho, Wo) && cout1_wgrad_ws_bytes(N, Ho, Cin) > need) need = cout1_wgrad_ws_bytes(N, Ho, Cin);
  return need;
}

int ladder_conv2d_bwd_filter_kernel_id(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                       int pad_l) {
  // 9128 = wgrad3x3_halo_kernel (tap-fused, LDS-DMA staged); 0 = any other filter-gradient path
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(1), make_fastdiv(1)};
  return plan_wgrad_halo(d).ok ? 9128 : 0;
}

int ladder_conv2d_bwd_filter(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Ho,
                             int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, void* ws, size_t ws_bytes,
                             ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return LADDER_E_SHAPE;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, LADDER_ACT_NONE, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  if (cout1_eligible(Cin, Cout, KH, KW, stride, pad_t, pad_l, H, W, Ho, Wo)) {
    if (ws == nullptr || ws_bytes < cout1_wgrad_ws_bytes(N, Ho, Cin)) return LADDER_E_WORKSPACE;
    const int rows = N * Ho, blocks = cout1_blocks(rows, Cin), kn = 25 * Cin;
    float* part = (float*)ws;
#define LADDER_COUT1_WG(CL_) hipLaunchKernelGGL((conv_cout1_kernel<5, 5, CL_, true>), dim3(blocks), dim3(256), 0, stream, x, (const float*)nullptr, \
                                                (const float*)nullptr, (float*)nullptr, dy, part, H, W, Ho, Wo, 0, rows)
    if (Cin == 64) LADDER_COUT1_WG(64); else if (Cin == 32) LADDER_COUT1_WG(32); else LADDER_COUT1_WG(16);
#undef LADDER_COUT1_WG
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((kn + 1 + 15) / 16), dim3(256), 0, stream, (const float*)part, dw, db, blocks, kn, 1);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  return run_wgrad(x, dy, dw, db, d, ws, ws_bytes, stream);
}

int ladder_conv2d_bwd_filter_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                            int pad_l) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return 0;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  return (wgrad_split_ok(d) && !plan_wgrad_halo(d).ok) ? 1 : 0;
}

size_t ladder_conv2d_bwd_filter_split_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW) {
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, 1, 1, 0, 0, N * Ho * Wo, KH * KW * Cin, 0, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  return wgrad_split_ws_bytes(d);
}

int ladder_conv2d_bwd_filter_split(const void* x_planes, const float* x_absmax, const void* dy_planes, const float* dy_absmax, float* dw, float* db,
                                   int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l,
                                   int prec, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return LADDER_E_SHAPE;
  IgemmDesc d{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, 1, pad_t, pad_l, N * Ho * Wo, KH * KW * Cin, LADDER_ACT_NONE, make_fastdiv(Ho * Wo), make_fastdiv(Wo)};
  return run_wgrad_split(x_planes, x_absmax, dy_planes, dy_absmax, dw, db, d, prec, ws, ws_bytes, stream);
}

int ladder_conv1x1_smallcout_eligible(long M, int Cin, int Cout) { return smallcout_eligible(Cin, Cout, 1, 1, 1, M) ? 1 : 0; }

size_t ladder_conv1x1_smallcout_bwd_workspace_bytes(long M, int Cin, int Cout) {
  return (size_t)smallcout_blocks(M, Cin) * ((size_t)Cin * Cout + Cout) * sizeof(float);
}

int ladder_conv1x1_smallcout_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long M, int Cin,
                                 int Cout, int gate_act, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  return ladder_conv1x1_smallcout_bwd_absmax(x, dy, w, dx, dw, db, M, Cin, Cout, gate_act, ws, ws_bytes, nullptr, 0, stream);
}

int ladder_conv1x1_smallcout_bwd_absmax(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long M, int Cin,
                                        int Cout, int gate_act, void* ws, size_t ws_bytes, float* dx_absmax, long rows_per_sample,
                                        ladder_stream_t stream) {
  if (!smallcout_eligible(Cin, Cout, 1, 1, 1, M)) return LADDER_E_SHAPE;
  int rps = (rows_per_sample > 0 && rows_per_sample < (1L << 30) && (M % rows_per_sample) == 0) ? (int)rows_per_sample : 0;
  if (dx_absmax != nullptr && (dx == nullptr || hipMemsetAsync(dx_absmax, 0, LADDER_ABSMAX_FLOATS * sizeof(float), stream) != hipSuccess))
    return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || (dx != nullptr && !ladder_aligned16(dx))) return LADDER_E_ALIGN;
  if (ws == nullptr || ws_bytes < ladder_conv1x1_smallcout_bwd_workspace_bytes(M, Cin, Cout)) return LADDER_E_WORKSPACE;
  int blocks = smallcout_blocks(M, Cin);
  const int kn = Cin * Cout;
  if (rps > 0) {                                              // per-sample record: N * bps workgroups (<= the partial-sum slots of the workspace)
    const long nsmp = M / rps, ppb = 256 / (Cin >> 2);
    long bps = blocks / nsmp;
    if (bps > (rps + ppb - 1) / ppb) bps = (rps + ppb - 1) / ppb;
    if (nsmp > blocks || bps < 1) rps = 0; else blocks = (int)(nsmp * bps);
  }
  float* part = (float*)ws;
#define LADDER_SCO_BWD(CO_) hipLaunchKernelGGL((conv1x1_smallcout_kernel<CO_, true>), dim3(blocks), dim3(256), 0, stream, x, w, (const float*)nullptr, \
                                               (float*)nullptr, dy, dx, part, Cin, gate_act, (int)M, dx_absmax, rps)
  switch (Cout) { case 1: LADDER_SCO_BWD(1); break; case 2: LADDER_SCO_BWD(2); break; case 3: LADDER_SCO_BWD(3); break; default: LADDER_SCO_BWD(4); }
#undef LADDER_SCO_BWD
  // partial layout per block: [Cin*Cout filter gradient | Cout bias gradient]
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((kn + Cout + 15) / 16), dim3(256), 0, stream, (const float*)part, dw, db, blocks, kn, Cout);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_dense_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                     void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (dense_f32_big_ok(M, K, N)) return dense_f32_big_launch(x, w, bias, y, nullptr, 0, M, K, N, act, stream);     // (csrc/densef32.hip)
  IgemmDesc d{M, 1, 1, K, 1, 1, N, 1, 1, 1, 1, 0, 0, M, K, act, make_fastdiv(1), make_fastdiv(1)};
  set_conv_taps(d);
  return dispatch_fwd(x, w, bias, y, d, ws, ws_bytes, stream);
}

int ladder_dense_fwd_nt(const float* x, const float* wT, const float* bias, float* y, int M, int K, int N, int act, ladder_stream_t stream) {
  return dense_f32_nt_launch(x, wT, bias, y, nullptr, 0, M, K, N, act, stream);                    // (csrc/densef32.hip; LADDER_E_SHAPE when not eligible)
}

int ladder_dense_bwd_data_nt(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act, ladder_stream_t stream) {
  return dense_f32_nt_launch(dy, w, nullptr, dx, gate_y, gate_act, M, N, K, LADDER_ACT_NONE, stream);
}

int ladder_dense_bwd_data(const float* dy, const float* wT, float* dx, int M, int K, int N, const float* gate_y, int gate_act,
                          void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (dense_f32_big_ok(M, N, K)) return dense_f32_big_launch(dy, wT, nullptr, dx, gate_y, gate_act, M, N, K, LADDER_ACT_NONE, stream);
  IgemmDesc d{M, 1, 1, N, 1, 1, K, 1, 1, 1, 1, 0, 0, M, N, LADDER_ACT_NONE, make_fastdiv(1), make_fastdiv(1)};
  set_conv_taps(d);
  return dispatch_fwd(dy, wT, nullptr, dx, d, ws, ws_bytes, stream, gate_y, gate_act);
}

size_t ladder_dense_bwd_weight_workspace_bytes(int M, int K, int N) {
  const size_t a = wgrad_ws_bytes(M, K, N), b = dense_wgrad_f32_ok(M, K, N) ? dense_wgrad_f32_ws_bytes(M, K, N) : 0;
  return a > b ? a : b;
}

int ladder_dense_bwd_weight(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, void* ws,
                            size_t ws_bytes, ladder_stream_t stream) {
  if (dense_wgrad_f32_ok(M, K, N)) {                             // (csrc/densef32.hip: the filter gradients of the projected decoder pairs)
    if (ws == nullptr || ws_bytes < dense_wgrad_f32_ws_bytes(M, K, N)) return LADDER_E_WORKSPACE;
    if (!ladder_aligned16(dw)) return LADDER_E_ALIGN;
    int splits, mps;
    dense_wgrad_f32_plan(M, K, N, &splits, &mps);
    const size_t kn = (size_t)K * N;
    float* part = (float*)ws;
    float* bias_part = db != nullptr ? part + (size_t)splits * kn : nullptr;
    const int rc = dense_wgrad_f32_launch(x, dy, part, bias_part, M, K, N, splits, mps, stream);
    if (rc != LADDER_OK) return rc;
    launch_reduce_splits(part, dw, splits, kn, stream);
    if (db != nullptr) launch_reduce_splits(bias_part, db, splits, (size_t)N, stream);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  IgemmDesc d{M, 1, 1, K, 1, 1, N, 1, 1, 1, 1, 0, 0, M, K, LADDER_ACT_NONE, make_fastdiv(1), make_fastdiv(1)};
  return run_wgrad(x, dy, dw, db, d, ws, ws_bytes, stream);
}


}  // extern "C"
