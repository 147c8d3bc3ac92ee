// "Project, then upsample" (round 5): resize x2 -> 3x3 / SAME convolution with 9 / 36 of the reference's products, strict fp32.
//
// tf.image.resize_images (TF1 legacy bilinear, factor 2) is linear and acts on every channel separately, so it commutes with the channel
// contraction of the convolution behind it (decoder conv2d_4 ... conv2d_7, reference codes/models.py:544-578):
//
//     conv3x3(up(x))[p, q] = b + sum_{r,s} up(Z_rs)[p + r - 1, q + s - 1],      Z_rs = x . w[r][s]   (nine 1x1 convolutions at LOW resolution)
//
// with up(Z) taken as zero outside the high-resolution map (the convolution's zero padding).  The contraction work is 9 products per
// low-resolution pixel = 9 / 36 of the direct form on the upsampled map (the tap-folding of rounds 3-5: 25 / 36); what is added is an elementwise
// pass over the nine low-resolution maps -- exact on every pixel, so the edge-line / border-line / line-gradient helpers of the tap-folded form
// have no counterpart here.  Three GEMM-shaped calls on the dense kernels (igemm.hip) and three elementwise kernels (this file):
//
//   forward        Z [M, 9 Cout] = x [M, Cin] . Wcat [Cin, 9 Cout]            (ladder_dense_fwd; Wcat = orientation 6 of filterbank.h, ladder_filter_pack_split)
//                  y = act(b + sum_rs shift_rs(up(Z_rs)))                      ladder_up2proj_fwd_combine (+ the fused 1x1 projection of the last layer)
//   backward-data  D_rs = (shift_rs o up)^T dy   [M, 9 Cout]                   ladder_up2proj_bwd_combine
//                  dx [M, Cin] = D . WcatT [9 Cout, Cin]                       (ladder_dense_fwd; orientation 7)
//   filter grad.   dWcat [Cin, 9 Cout] = x^T D, db9 = column sums of D        (ladder_dense_bwd_weight), then
//                  dw[t][ci][co] = dWcat[ci][t Cout + co], db = db9[4 Cout ..] ladder_up2proj_wgrad_unpack  (the centre tap's column sums ARE sum dy)
//
// Per axis (low-resolution length L, u = p + r - 1 the position on the upsampled line, valid for 0 <= u < 2L):  up(Z)[2i] = Z[i],
// up(Z)[2i+1] = (Z[i] + Z[min(i+1, L-1)]) / 2.  Its transpose: D_r[i] = sum_{alpha=0..2} omega_alpha dy[2i - r + alpha], omega = (1/2 [i >= 1], 1,
// 1/2 -- or 1 on the last line, where the clamp folds both halves onto it), every term gated by 0 <= 2i - r + alpha < 2L.
#include "common.h"
#include <type_traits>

namespace {

// forward weights of tap r for output parity a at low-resolution line i: rows (lo, hi) = (i-1, i) for r = 0, (i, min(i+1, L-1)) for r = 1, 2
struct AxisW { int lo, hi; float wlo[2], whi[2]; };

__device__ __forceinline__ AxisW up2_axis(int r, int i, int L) {
  AxisW a;
  if (r == 0) {
    a.lo = i - 1; a.hi = i;
    a.wlo[0] = i >= 1 ? 0.5f : 0.f; a.whi[0] = i >= 1 ? 0.5f : 0.f;      // u = 2i - 1: the zero padding above / left of the map at i = 0
    a.wlo[1] = 0.f; a.whi[1] = 1.f;                                       // u = 2i
    if (i < 1) a.lo = i;                                                  // (never read with a non-zero weight; keeps the address inside the map)
  } else if (r == 1) {
    a.lo = i; a.hi = min(i + 1, L - 1);
    a.wlo[0] = 1.f; a.whi[0] = 0.f;                                       // u = 2i
    a.wlo[1] = 0.5f; a.whi[1] = 0.5f;                                     // u = 2i + 1 (clamped)
  } else {
    a.lo = i; a.hi = min(i + 1, L - 1);
    a.wlo[0] = 0.5f; a.whi[0] = 0.5f;                                     // u = 2i + 1 (clamped)
    a.wlo[1] = 0.f; a.whi[1] = (i + 1 <= L - 1) ? 1.f : 0.f;              // u = 2i + 2: the zero padding below / right of the map on the last line
  }
  return a;
}

// streaming store of a tensor that is written once and read by a LATER kernel (y, D: GBs): non-temporal, so that it does not displace the lines the
// neighbouring threads are about to re-read (Z resp. dy are re-read 2x / 6x through L2 inside these kernels).  LADDER_UPPROJ_NT=0 (build-time -D) turns it off.
#ifndef LADDER_UPPROJ_NT
#define LADDER_UPPROJ_NT 1
#endif
typedef float up_f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if LADDER_UPPROJ_NT
  up_f32x4 t;
  t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<up_f32x4*>(p));
#else
  *p = v;
#endif
}

__device__ __forceinline__ float4 f4_fma(float s, float4 v, float4 a) {
  return make_float4(fmaf(s, v.x, a.x), fmaf(s, v.y, a.y), fmaf(s, v.z, a.z), fmaf(s, v.w, a.w));
}

// one step of a reduce-scatter over the lanes of a half-wave: a lane keeps the HALF values its bit selects and adds the partner's copy of them
// (bitwise selects: both operands stay in registers -- a `?:` on array elements becomes a 16-way select chain on the index)
template <int HALF>
__device__ __forceinline__ void rs_step(float (&pv)[16], const int cq, const int bit) {
  const uint32_t m = (cq & bit) ? 0xffffffffu : 0u;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const uint32_t a = __float_as_uint(pv[k]), b = __float_as_uint(pv[k + HALF]);
    const float send = __uint_as_float((a & m) | (b & ~m)), keep = __uint_as_float((b & m) | (a & ~m));
    pv[k] = keep + __shfl_xor(send, bit, 64);
  }
}

// y [N, 2H, 2W, C] (may be NULL) and / or pout [N, 2H, 2W, pco] = the 1x1 projection of the ACTIVATED value (pw [C][pco], pb [pco]; needs C == 128:
// the 32 lanes of a half-wave hold one pixel's channels).  One thread = one low-resolution pixel x 4 channels = a 2x2 output block.
template <bool PROJ>
__global__ __launch_bounds__(256) void up2proj_fwd_combine_kernel(const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ y,
                                                                  const float* __restrict__ pw, const float* __restrict__ pb,
                                                                  float* __restrict__ pout, const int pco, const int N, const int H, const int W,
                                                                  const int C, const int act) {
  const int CV = C >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)N * H * W * CV;
  const bool live = t < total;
  const long tt = live ? t : 0;
  const int cq = (int)(tt % CV);
  const long pix = tt / CV;
  const int j = (int)(pix % W), i = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
  const float4* zb = reinterpret_cast<const float4*>(z) + (long)n * H * W * 9 * CV + cq;       // plane t at + t * CV, pixel stride 9 CV
  float4 acc[2][2];
  const float4 bv = bias != nullptr ? reinterpret_cast<const float4*>(bias)[cq] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = bv;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const AxisW R = up2_axis(r, i, H);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const AxisW S = up2_axis(s, j, W);
      const float4* zp = zb + (r * 3 + s) * CV;
      const float4 v00 = zp[((long)R.lo * W + S.lo) * 9 * CV], v01 = zp[((long)R.lo * W + S.hi) * 9 * CV];
      const float4 v10 = zp[((long)R.hi * W + S.lo) * 9 * CV], v11 = zp[((long)R.hi * W + S.hi) * 9 * CV];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float4 q = acc[a][b];
          q = f4_fma(R.wlo[a] * S.wlo[b], v00, q);
          q = f4_fma(R.wlo[a] * S.whi[b], v01, q);
          q = f4_fma(R.whi[a] * S.wlo[b], v10, q);
          q = f4_fma(R.whi[a] * S.whi[b], v11, q);
          acc[a][b] = q;
        }
    }
  }
  float4 pwv[4];                                             // rows of the projection matrix for this thread's 4 channels (pco <= 4 columns, zero padded)
  if (PROJ) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float* pr = pw + (long)(cq * 4 + c4) * pco;
      pwv[c4] = make_float4(pr[0], pco > 1 ? pr[1] : 0.f, pco > 2 ? pr[2] : 0.f, pco > 3 ? pr[3] : 0.f);
    }
  }
  float pv[16];                                              // projection partials: [pixel a * 2 + b][output column]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float4 v = acc[a][b];
      v = make_float4(ladder_act_fn(v.x, act), ladder_act_fn(v.y, act), ladder_act_fn(v.z, act), ladder_act_fn(v.w, act));
      const long opix = ((long)n * 2 * H + 2 * i + a) * 2 * W + 2 * j + b;
      if (live && y != nullptr) reinterpret_cast<float4*>(y)[opix * CV + cq] = v;
      if (PROJ) {
        float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
        pr = f4_fma(v.x, pwv[0], pr);
        pr = f4_fma(v.y, pwv[1], pr);
        pr = f4_fma(v.z, pwv[2], pr);
        pr = f4_fma(v.w, pwv[3], pr);
        pv[(a * 2 + b) * 4 + 0] = pr.x; pv[(a * 2 + b) * 4 + 1] = pr.y; pv[(a * 2 + b) * 4 + 2] = pr.z; pv[(a * 2 + b) * 4 + 3] = pr.w;
      }
    }
  if (PROJ) {
    // C == 128: the 32 lanes cq = 0 .. 31 of a half-wave hold one low-resolution pixel.  Fixed-order reduce-scatter over them: at every step a
    // lane keeps the half of its values its bit selects and adds the partner's copy of that half (8 + 4 + 2 + 1 + 1 = 16 exchanges for 16 sums).
    rs_step<8>(pv, cq, 16);
    rs_step<4>(pv, cq, 8);
    rs_step<2>(pv, cq, 4);
    rs_step<1>(pv, cq, 2);
    pv[0] += __shfl_xor(pv[0], 1, 64);
    const int idx = cq >> 1, pixel = idx >> 2, o = idx & 3;          // this lane's sum: output column o of pixel (a, b) = (pixel >> 1, pixel & 1)
    if (live && (cq & 1) == 0 && o < pco) {
      const long opix = ((long)n * 2 * H + 2 * i + (pixel >> 1)) * 2 * W + 2 * j + (pixel & 1);
      pout[opix * pco + o] = pv[0] + (pb != nullptr ? pb[o] : 0.f);
    }
  }
}

// The same combination with one thread walking `seg` consecutive low-resolution rows of its (column, 4 channels): of the two Z rows a plane needs
// per step, the upper one is the lower one of the step before and stays in registers -- 18 instead of 36 loads per 2x2 output block (the 4x re-read
// of Z through L2, not HBM, bounded the one-pixel kernel above: 892 us for conv2d_7's 2.4 GB without a y write).
template <bool PROJ>
__global__ __launch_bounds__(256) void up2proj_fwd_combine_rows_kernel(const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ y,
                                                                       const float* __restrict__ pw, const float* __restrict__ pb,
                                                                       float* __restrict__ pout, const int pco, const int N, const int H, const int W,
                                                                       const int C, const int act, const int seg) {
  const int CV = C >> 2, nseg = (H + seg - 1) / seg;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)N * nseg * W * CV;
  const bool live = t < total;
  const long tt = live ? t : 0;
  const int cq = (int)(tt % CV);
  const long rest = tt / CV;
  const int j = (int)(rest % W), sg = (int)((rest / W) % nseg), n = (int)(rest / ((long)W * nseg));
  const int i0 = sg * seg, i1 = min(H, i0 + seg);
  const float4* zb = reinterpret_cast<const float4*>(z) + (long)n * H * W * 9 * CV + cq;
  const float4 bv = bias != nullptr ? reinterpret_cast<const float4*>(bias)[cq] : make_float4(0.f, 0.f, 0.f, 0.f);
  AxisW S[3];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) S[sx] = up2_axis(sx, j, W);
  float4 pwv[4];
  if (PROJ) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float* pr = pw + (long)(cq * 4 + c4) * pco;
      pwv[c4] = make_float4(pr[0], pco > 1 ? pr[1] : 0.f, pco > 2 ? pr[2] : 0.f, pco > 3 ? pr[3] : 0.f);
    }
  }
  // lo[plane][column]: the plane's LOWER-index row of the current step: row i - 1 for the planes of tap row r = 0, row i for r = 1, 2
  float4 lo[9][2], hi[9][2];
  auto load_row = [&](float4 (&dst)[9][2], int r, int row) {
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const float4* zp = zb + (r * 3 + sx) * CV + (long)row * W * 9 * CV;
      dst[r * 3 + sx][0] = zp[(long)S[sx].lo * 9 * CV];
      dst[r * 3 + sx][1] = zp[(long)S[sx].hi * 9 * CV];
    }
  };
  load_row(lo, 0, max(i0 - 1, 0));
  load_row(lo, 1, i0);
  load_row(lo, 2, i0);
  for (int i = i0; i < i1; ++i) {
    const int up = min(i + 1, H - 1);
    load_row(hi, 0, i);
    load_row(hi, 1, up);
    load_row(hi, 2, up);
    float4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = bv;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const AxisW R = up2_axis(r, i, H);
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) {
        const int pl = r * 3 + sx;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float4 q = acc[a][b];
            q = f4_fma(R.wlo[a] * S[sx].wlo[b], lo[pl][0], q);
            q = f4_fma(R.wlo[a] * S[sx].whi[b], lo[pl][1], q);
            q = f4_fma(R.whi[a] * S[sx].wlo[b], hi[pl][0], q);
            q = f4_fma(R.whi[a] * S[sx].whi[b], hi[pl][1], q);
            acc[a][b] = q;
          }
      }
    }
    float pv[16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float4 v = acc[a][b];
        v = make_float4(ladder_act_fn(v.x, act), ladder_act_fn(v.y, act), ladder_act_fn(v.z, act), ladder_act_fn(v.w, act));
        const long opix = ((long)n * 2 * H + 2 * i + a) * 2 * W + 2 * j + b;
        if (live && y != nullptr) st_stream(reinterpret_cast<float4*>(y) + opix * CV + cq, v);
        if (PROJ) {
          float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
          pr = f4_fma(v.x, pwv[0], pr);
          pr = f4_fma(v.y, pwv[1], pr);
          pr = f4_fma(v.z, pwv[2], pr);
          pr = f4_fma(v.w, pwv[3], pr);
          pv[(a * 2 + b) * 4 + 0] = pr.x; pv[(a * 2 + b) * 4 + 1] = pr.y; pv[(a * 2 + b) * 4 + 2] = pr.z; pv[(a * 2 + b) * 4 + 3] = pr.w;
        }
      }
    if (PROJ) {
      rs_step<8>(pv, cq, 16);
      rs_step<4>(pv, cq, 8);
      rs_step<2>(pv, cq, 4);
      rs_step<1>(pv, cq, 2);
      pv[0] += __shfl_xor(pv[0], 1, 64);
      const int idx = cq >> 1, pixel = idx >> 2, o = idx & 3;
      if (live && (cq & 1) == 0 && o < pco) {
        const long opix = ((long)n * 2 * H + 2 * i + (pixel >> 1)) * 2 * W + 2 * j + (pixel & 1);
        pout[opix * pco + o] = pv[0] + (pb != nullptr ? pb[o] : 0.f);
      }
    }
#pragma unroll
    for (int pl = 0; pl < 9; ++pl) { lo[pl][0] = hi[pl][0]; lo[pl][1] = hi[pl][1]; }
  }
}

// D [N, H, W, 9 C] from dy [N, 2H, 2W, C]: D_rs[i, j] = sum_{alpha, beta} omega_r[alpha] omega_c[beta] dy[2i - r + alpha, 2j - s + beta]
__global__ __launch_bounds__(256) void up2proj_bwd_combine_kernel(const float* __restrict__ dy, float* __restrict__ d, const int N, const int H,
                                                                  const int W, const int C) {
  const int CV = C >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)N * H * W * CV) return;
  const int cq = (int)(t % CV);
  const long pix = t / CV;
  const int j = (int)(pix % W), i = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
  const float4* gy = reinterpret_cast<const float4*>(dy) + (long)n * 4 * H * W * CV + cq;
  // the 5 x 5 neighbourhood rows 2i-2 .. 2i+2, columns 2j-2 .. 2j+2 (zero outside the map)
  float4 g[5][5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int p = 2 * i - 2 + a, q = 2 * j - 2 + b;
      g[a][b] = (p >= 0 && p < 2 * H && q >= 0 && q < 2 * W) ? gy[((long)p * 2 * W + q) * CV] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  const float wr[3] = {i >= 1 ? 0.5f : 0.f, 1.f, i == H - 1 ? 1.f : 0.5f}, wc[3] = {j >= 1 ? 0.5f : 0.f, 1.f, j == W - 1 ? 1.f : 0.5f};
  float4* dp = reinterpret_cast<float4*>(d) + pix * 9 * CV + cq;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int al = 0; al < 3; ++al)
#pragma unroll
        for (int be = 0; be < 3; ++be) acc = f4_fma(wr[al] * wc[be], g[2 - r + al][2 - s + be], acc);     // row 2i - r + al = (2i - 2) + (2 - r + al)
      st_stream(dp + (r * 3 + s) * CV, acc);
    }
}


// ---- round 6: the backward combination as a WALK DOWN THE ROWS, optionally straight from the gradient of the 1x1 projection behind the pair ----------------
// up2proj_bwd_combine_kernel loads the 5 x 5 neighbourhood of every low-resolution pixel (25 float4 per thread, every dy value fetched 6.25 times through
// L1 / L2) and needs dy itself in HBM.  The combination is separable -- D_rs[i, j] = sum_alpha omega_r[alpha] C_s[2i - r + alpha], C_s[p] = sum_beta
// omega_c[beta] dy[p, 2j - s + beta] -- so a thread that owns (image, column j, channel quad) and walks down a segment of rows folds every high-resolution
// row ONCE along the columns (5 loads -> 3 folded values) and keeps the folded rows of the 5-row window in registers: 10 loads per low-resolution pixel.
// PCO > 0 (the last pair: conv2d_7 under the 1x1 conv2d_8): dy is never materialised -- it is formed where it is consumed,
//   dy[p, q, c] = act'(y[p, q, c]) * sum_o dyp[p, q, o] pw[c][o]
// from the pair's activated output y and the PCO-channel gradient dyp, and the walk also owns the 2 x 2 pixels under its low-resolution pixel for the
// projection's filter / bias gradient (per-block partials in the layout of conv1x1_smallcout_kernel, summed in a fixed order): the separate pass that
// read y (1.07 GB at batch 128) and wrote dy (1.07 GB), and this kernel's read of dy, become ONE read of y.
template <int PCO>
__global__ __launch_bounds__(256) void up2proj_bwd_combine_walk_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ dyp,
                                                                       const float* __restrict__ pw, float* __restrict__ d, float* __restrict__ part,
                                                                       const int N, const int H, const int W, const int C, const int seg, const int act) {
  constexpr int PC = PCO > 0 ? PCO : 1;
  const int CV = C >> 2, nseg = (H + seg - 1) / seg;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < (long)N * nseg * W * CV;
  const int cq = (int)(t % CV);
  const long u = t / CV;
  const int j = (int)(u % W), sg = (int)((u / W) % nseg), n = live ? (int)(u / ((long)W * nseg)) : 0;
  const float4* gy = reinterpret_cast<const float4*>(PCO > 0 ? y : dy) + (long)n * 4 * H * W * CV + cq;
  const float* gp = PCO > 0 ? dyp + (long)n * 4 * H * W * PCO : nullptr;
  float4 wq[PC], gw[PC];
  float gb[PC];
#pragma unroll
  for (int o = 0; o < PC; ++o) {
    wq[o] = PCO > 0 ? make_float4(pw[(4 * cq + 0) * PC + o], pw[(4 * cq + 1) * PC + o], pw[(4 * cq + 2) * PC + o], pw[(4 * cq + 3) * PC + o]) : make_float4(0.f, 0.f, 0.f, 0.f);
    gw[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb[o] = 0.f;
  }
  const float wc0 = j >= 1 ? 0.5f : 0.f, wc2 = j == W - 1 ? 1.f : 0.5f;
  // (PCO > 0) the 5 x PCO gradient values of a row are the same for the CV <= 64 lanes that share the pixel: lane k < 5 PCO of the group fetches ONE of
  // them (consecutive addresses), the group passes them round through its LDS slot -- one load, one ds_write and <= 5 broadcast ds_read_b128 per row instead of
  // 5 PCO loads per lane (which made the texture-address path, not HBM, the limit: 1 549 us against the two launches' 1 117, profiles/r06_bwd_walk_probe.txt).
  // The lanes of a group sit in ONE wavefront, whose LDS operations complete in order: no s_barrier, only the compiler's convergence point.
  __shared__ __attribute__((aligned(16))) float gsl[PCO > 0 ? 256 / 4 * 40 : 4];
  float* slot = gsl + (PCO > 0 ? (threadIdx.x / CV) * 40 : 0);                      // (two rows of 20 floats per group)
  const float neg = act == LADDER_ACT_LEAKY ? 0.2f : (act == LADDER_ACT_RELU ? 0.f : 1.f);      // act'(y) = y > 0 ? 1 : neg
  constexpr int GL = 1;                                                             // gradient values a lane fetches per row: CV >= 5 PCO (eligibility)
  // the global loads of one high-resolution row p: five float4 of y (or dy) and this lane's share of the row's 5 x PCO projection-gradient values
  auto load_row = [&](const int p, float4 (&v)[5], float (&gl)[GL]) __attribute__((always_inline)) {
    const bool rok = live && p >= 0 && p < 2 * H;
#pragma unroll
    for (int b5 = 0; b5 < 5; ++b5) {
      const int q = 2 * j - 2 + b5;
      v[b5] = (rok && q >= 0 && q < 2 * W) ? gy[((long)p * 2 * W + q) * CV] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (PCO > 0) {
#pragma unroll
      for (int tr = 0; tr < GL; ++tr) {
        const int k = cq + tr * CV, q = 2 * j - 2 + k / PC;
        gl[tr] = (tr * CV < 5 * PC && k < 5 * PC && rok && q >= 0 && q < 2 * W) ? gp[((long)p * 2 * W + q) * PC + (k - (k / PC) * PC)] : 0.f;
      }
    }
  };
  // ... and the row folded along the columns: c[s] = sum_beta omega_c[beta] dy[p, 2j - s + beta]
  auto fold_row = [&](const float4 (&v)[5], const float (&gl)[GL], float* sl, const bool own, float4 (&c)[3]) __attribute__((always_inline)) {
    float4 g[5];
    float go[5 * PC];
    if (PCO > 0) {
#pragma unroll
      for (int tr = 0; tr < GL; ++tr) {
        const int k = cq + tr * CV;
        if (tr * CV < 5 * PC && k < 5 * PC) sl[k] = gl[tr];
      }
      // (without a convergence point the compiler sinks the reads into both sides of the branch above, and the lanes that do not write may read first; the barrier
      // is convergent but not a memory operation, so a compiler-level memory clobber on either side keeps the LDS accesses on their side of it)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k4 = 0; k4 < (5 * PC + 3) / 4; ++k4) {
        const float4 t4 = *reinterpret_cast<const float4*>(sl + 4 * k4);
        go[4 * k4] = t4.x;
        if (4 * k4 + 1 < 5 * PC) go[4 * k4 + 1] = t4.y;
        if (4 * k4 + 2 < 5 * PC) go[4 * k4 + 2] = t4.z;
        if (4 * k4 + 3 < 5 * PC) go[4 * k4 + 3] = t4.w;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int b5 = 0; b5 < 5; ++b5) {
      if (PCO == 0) {
        g[b5] = v[b5];
      } else {
        float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < PC; ++o) {
          d4 = f4_fma(go[b5 * PC + o], wq[o], d4);
          if ((b5 == 2 || b5 == 3) && own) {                                      // the 2 x 2 pixels under (i, j) belong to this thread
            gw[o] = f4_fma(go[b5 * PC + o], v[b5], gw[o]);
            gb[o] += go[b5 * PC + o];
          }
        }
        g[b5] = make_float4(d4.x * (v[b5].x > 0.f ? 1.f : neg), d4.y * (v[b5].y > 0.f ? 1.f : neg), d4.z * (v[b5].z > 0.f ? 1.f : neg),
                            d4.w * (v[b5].w > 0.f ? 1.f : neg));
      }
    }
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) c[sft] = f4_fma(wc0, g[2 - sft], f4_fma(wc2, g[4 - sft], g[3 - sft]));
  };
  const int i0 = sg * seg, i1 = min(H, i0 + seg);
  float4 cw[5][3];                                                                   // folded rows 2i - 2 ... 2i + 2
  float4 va[5], vb[5], na[5];
  float ga[GL], gb2[GL], nga[GL];
  load_row(2 * i0 - 2, va, ga);
  load_row(2 * i0 - 1, vb, gb2);
  load_row(2 * i0, na, nga);
  fold_row(va, ga, slot, false, cw[2]);
  fold_row(vb, gb2, slot + 20, false, cw[3]);
  fold_row(na, nga, slot, true, cw[4]);
  if (PCO > 0) {
    load_row(2 * i0 + 1, va, ga);
    load_row(2 * i0 + 2, vb, gb2);
  }
  for (int i = i0; i < i1; ++i) {
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) { cw[0][sft] = cw[2][sft]; cw[1][sft] = cw[3][sft]; cw[2][sft] = cw[4][sft]; }
    // PCO > 0 (two waves a SIMD at best): the rows of the NEXT step are requested while this one is folded -- its first row before the first fold, its second
    // into the registers the first fold has just released (880 against 1 300 us at conv2d_7; the plain form has the occupancy to hide the latency: no gain)
    const bool more = PCO > 0 && i + 1 < i1;
    if (more) load_row(2 * i + 3, na, nga);
    if (PCO == 0) load_row(2 * i + 1, va, ga);
    fold_row(va, ga, slot, true, cw[3]);
    if (more) load_row(2 * i + 4, va, ga);
    if (PCO == 0) {
      load_row(2 * i + 2, va, ga);
      fold_row(va, ga, slot, false, cw[4]);
    } else {
      fold_row(vb, gb2, slot + 20, i + 1 < i1, cw[4]);                               // (row 2 i1 belongs to the walk of the next segment)
    }
    if (live) {
      const float wr0 = i >= 1 ? 0.5f : 0.f, wr2 = i == H - 1 ? 1.f : 0.5f;
      float4* dp = reinterpret_cast<float4*>(d) + (((long)n * H + i) * W + j) * 9 * CV + cq;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
          st_stream(dp + (r * 3 + sft) * CV, f4_fma(wr0, cw[2 - r][sft], f4_fma(wr2, cw[4 - r][sft], cw[3 - r][sft])));
    }
    if (PCO > 0) {
#pragma unroll
      for (int b5 = 0; b5 < 5; ++b5) { vb[b5] = va[b5]; va[b5] = na[b5]; }
#pragma unroll
      for (int tr = 0; tr < GL; ++tr) { gb2[tr] = ga[tr]; ga[tr] = nga[tr]; }
    }
  }
  if (PCO > 0) {
    // block partial [C][PCO] filter gradient + [PCO] bias gradient: the 256 / CV pixel lanes of a channel quad in a fixed order
    __shared__ float red[256];
    const int pl = threadIdx.x / CV, ppb = 256 / CV;
    float* out = part + (size_t)blockIdx.x * ((size_t)C * PC + PC);
#pragma unroll
    for (int k = 0; k < 5 * PC; ++k) {
      float v;
      if (k < 4 * PC) {
        const float4 gv = gw[k >> 2];
        const int e = k & 3;
        v = e == 0 ? gv.x : (e == 1 ? gv.y : (e == 2 ? gv.z : gv.w));
      } else {
        v = cq == 0 ? gb[k - 4 * PC] : 0.f;
      }
      red[threadIdx.x] = v;
      __syncthreads();
      if (pl == 0) {
        float a = 0.f;
        for (int r = 0; r < ppb; ++r) a += red[r * CV + cq];
        if (k < 4 * PC) out[(size_t)(4 * cq + (k & 3)) * PC + (k >> 2)] = a;
        else if (cq == 0) out[(size_t)C * PC + (k - 4 * PC)] = a;
      }
      __syncthreads();
    }
  }
}

// out[i] = sum over the S block partials: one wavefront per output, lane l adds partials l, l + 64, ... in order, then a fixed xor tree over the lanes
// (bit-reproducible; 25 workgroups of 16 serial runs took 24 us for the 1 024 x 387 partials of conv2d_7)
__global__ __launch_bounds__(256) void up2proj_part_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, const int S,
                                                                  const int n, const int nb) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n + nb) return;                                                           // (whole wavefronts leave together)
  float a = 0.f;
  for (int z = lane; z < S; z += 64) a += part[(size_t)z * (n + nb) + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) {
    if (i < n) dw[i] = a;
    else if (db != nullptr) db[i - n] = a;
  }
}

// ---- round 6: the forward pair in ONE launch -- Z never leaves the CU ------------------------------------------------------------------------------------
// ladder_up2proj_fwd_combine reads back the nine planes Z [M][9 Cout] that the projection GEMM has just written: 2 x 2.4 GB per forward of conv2d_7 at
// batch 128, 23.7 GB per iteration over all pairs.  Here a workgroup owns (a group of G = 64 / W images, a slab of 16 output
// channels) and STREAMS DOWN THE ROWS of the low-resolution map:
//   per row k:  Zrow [64 px][9 x 16] = X_k [64 px][Cin] . wslab [Cin][9 x 16]   -- v_mfma_f32_16x16x4_f32, K in chunks of 32 through a 3-stage LDS pipeline
//               (64 px = row k of G images side by side: W in {8, 16, 32, 64}, so a row step never needs a column halo)
//               Zrow -> ONE row of the nine planes in LDS (36 KB); the combination threads read it once, fold it along the columns into six values per
//               (pixel, 4 channels) -- H_r^b = sum_s up-weights x Z_rs, tap row r, output-column parity b -- and keep the folded rows k - 1, k - 2 in
//               REGISTERS: output rows 2(k-1), 2(k-1) + 1 are sums of <= 6 of those (18 LDS reads per row and thread instead of the 54 of a version
//               that re-read three rows of a ring, and the row halo costs no LDS at all)
// so the only HBM traffic is x (re-read by the Cout / 16 slabs of a group, which sit next to each other on ONE XCD and share its L2), the weight slab
// (L2-resident: 9 x 16 x Cin floats) and y.  No halo in either direction: the MFMA work is exactly the projection GEMM's (9 of 36 products).
//
// ROLES (one workgroup per CU, 12 waves).  Waves 0-3 = one per SIMD -- do nothing but LDS fragment reads and MFMAs: wave w owns pixel tile w (16 px) and
// all nine taps (one x fragment feeds 9 MFMAs per k-step; 9 independent accumulators keep the matrix pipe issuing back to back; the fragments of the
// next 16-deep group -- across chunk and row boundaries -- are requested before the 36 MFMAs of the current one).  Waves 4-7 stage the x / weight chunks
// (global -> registers three chunks ahead -> LDS, three stages); waves 8-11 combine.  The three sides never meet at an s_barrier: they hand stages over
// through LDS counters (full / empty per stage, row written / row read for the plane row).  Measured on the way (profiles/r06_fused_*.txt, conv2d_7):
// one __syncthreads per chunk + combination behind the row's last chunk 2 115 us (two launches: 2 052); specialised waves but the stage released after
// its last MFMA 2 030 (a hand-over is two LDS round trips through a busy queue, ~1 000 cycles a hop: the chain was serial with the MFMAs); released as
// soon as its last fragment read has returned 1 971; the 3-row ring and its 54 reads per thread cost 350 us of that.
// Lane roles: the A operand of the MFMA is the WEIGHT slab (rows = 16 channels of one tap), the B operand the pixels, so a lane ends with 4 consecutive
// channels of one pixel = one 128-bit LDS write.  K order inside a 16-deep group is the permutation of gemm_nt16_f32_kernel (both fragments one
// ds_read_b128 per 4 k-steps).
// The 1x1 output conv of the last pair (conv2d_8, 128 -> 3) needs all 128 channels of a pixel: every slab writes its PARTIAL projection
// [slab][pco][pixel] and up2proj_proj_reduce_kernel sums the slabs in a fixed order (+ bias): bit-reproducible, 0.2 GB instead of the 1 GB activation.
//
// UF_LD = 40: the fragment read of lane (r16, kq) is a 16-byte piece of row r16 at float offset 16 u + 4 kq; ds_read_b128 is serviced in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): eight rows of one kq and the OTHER eight rows of the next.  With 40-float rows the first eight
// land on even 16-byte bank groups and the second eight on odd ones (r x 10 mod 16 is even): conflict-free; 36-float rows are 2-way conflicted in 7 of 8.
constexpr int UF_PX = 64, UF_CS = 16, UF_N = 9 * UF_CS, UF_K = 32, UF_LD = UF_K + 8, UF_THREADS = 768, UF_PLANE = UF_PX * UF_CS;
constexpr int UF_STAGES = 3;
// WRES (Cin <= 128): the whole weight slab [144][Cin] stays in LDS for the lifetime of the workgroup -- rows of Cin + 24 floats (conflict-free like UF_LD:
// (Cin + 24) / 4 = 6 or 14 mod 16) -- and only x is staged per chunk.  Measured (profiles/r06_fused_ablation_v8_loads.txt, conv2d_7): the weight chunks,
// re-fetched from L2 for every row (18 of the 26 KB a chunk stages), cost 280 us of the 1 760, the x chunks 170.
constexpr int UF_WRES_CIN = 128, UF_WRES_PAD = 24;
constexpr int UF_WS_FLOATS = UF_N * (UF_WRES_CIN + UF_WRES_PAD) > UF_STAGES * UF_N * UF_LD ? UF_N * (UF_WRES_CIN + UF_WRES_PAD) : UF_STAGES * UF_N * UF_LD;
constexpr int UF_MW = 4, UF_SW = 4, UF_CW = 4;                                           // MFMA / staging / combination waves
// hand-over words in LDS, four per event (one per wave of the publishing role; int4-aligned): FULL[stage] / EMPTY[stage] / ZFULL / CDONE
enum { UF_FULL0 = 0, UF_EMPTY0 = 4 * UF_STAGES, UF_ZFULL = 8 * UF_STAGES, UF_CDONE = 8 * UF_STAGES + 4, UF_WREADY = 8 * UF_STAGES + 8, UF_NFLAGS = 8 * UF_STAGES + 12 };

// A hand-over is a PROGRESS WORD per publishing wave (monotonic: rounds of a stage, rows written, rows read), written with a plain LDS store once the
// wave's own LDS reads / writes so far have completed (DS operations of a wave are processed in order), and read four at a time (one ds_read_b128,
// all lanes the same address: a broadcast): no atomics.
__device__ __forceinline__ void uf_publish(int* word, const int value) {
  // (the fence pins the MFMAs of the block in front of the hand-over: they are not memory operations, and the compiler otherwise sinks them below the
  // wait -- which then sits in front of the matrix block and exposes the latency of the fragment reads just issued for the next chunk)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // (lane 0 only: 64 lanes storing to ONE address are serialised by the LDS -- SQ_LDS_BANK_CONFLICT counted 218 cycles per chunk for the all-lane form)
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ int uf_peek(const int* words) {                            // the slowest of the four publishing waves
  typedef int uf_i32x4 __attribute__((ext_vector_type(4)));
  asm volatile("" ::: "memory");                                                      // (a fresh read every time: the clobbers keep it out of registers)
  const uf_i32x4 v = *reinterpret_cast<const uf_i32x4*>(words);
  asm volatile("" ::: "memory");
  return min(min(v[0], v[1]), min(v[2], v[3]));
}
// SLEEP: units of 64 cycles between polls.  Every poll is an LDS read in the queue the MFMA waves' fragment reads wait in: the staging / combination
// waves, which have chunks of slack, poll rarely; the MFMA waves (normally never waiting) poll tightly.
template <int SLEEP = 1>
__device__ __forceinline__ void uf_wait(const int* words, const int target) {
  int spins = 0;
  while (uf_peek(words) < target) {
    __builtin_amdgcn_s_sleep(SLEEP);
    if (++spins > (1 << 22)) __builtin_trap();                                        // (a hand-over that never comes is a bug: abort the launch, never hang the device)
  }
}

template <bool PROJ, bool WRES>
__global__ __launch_bounds__(UF_THREADS, 3) void up2proj_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wT, const float* __restrict__ bias,
                                                                          float* __restrict__ y, const float* __restrict__ pw, float* __restrict__ ppart,
                                                                          const int pco, const int N, const int H, const int W, const int wshift, const int Cin,
                                                                          const int Cout, const int act, const int dbg) {
  // dbg (ablation switches of profiles/tools/r6_fused_probe.py, 0 in production): 1 = the combination waves only hand the row over (no reads, no stores),
  // 2 = the staging waves issue no global loads (8: none of x, 16: none of the weights), 32 = no y stores
  __shared__ __attribute__((aligned(16))) float Xs[UF_STAGES][UF_PX * UF_LD];
  __shared__ __attribute__((aligned(16))) float Ws[UF_WS_FLOATS];                      // [stage][144][UF_LD], or (WRES) the whole slab [144][Cin + 24]
  __shared__ __attribute__((aligned(16))) float Zr[9 * UF_PLANE];
  __shared__ __attribute__((aligned(16))) int flags[UF_NFLAGS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int G = UF_PX >> wshift, ngroups = N / G, nslab = Cout / UF_CS;
  // workgroup -> (group, slab): the slabs of one group are consecutive workgroups of ONE XCD (workgroup b runs on XCD b % 8), so the group's x rows
  // are fetched into that L2 once and the two 64-byte halves of every y line are written from the same L2
  int group, slab;
  if ((ngroups & 7) == 0) {
    const int xcd = (int)(blockIdx.x & 7), l = (int)(blockIdx.x >> 3);
    group = (l / nslab) * 8 + xcd;
    slab = l % nslab;
  } else {
    group = (int)blockIdx.x / nslab;
    slab = (int)blockIdx.x % nslab;
  }
  const int n0 = group * G, c0 = slab * UF_CS;
  const int cpt = Cin / UF_K, total = H * cpt;
  if (tid < UF_NFLAGS) flags[tid] = 0;
  __syncthreads();                                                                    // (the only barrier of the kernel)

  if (wid < UF_MW) {
    // ------------------------------------------------------------------------------------------------ MFMA waves
    __builtin_amdgcn_s_setprio(2);                                                    // their LDS reads and MFMAs win the issue arbitration of their SIMD
    const int pt = wid;
    f32x4_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    float4 fx0, fx1, fw0[9], fw1[9];
    const int x_off = (pt * 16 + r16) * UF_LD + 4 * kq, w_off = r16 * UF_LD + 4 * kq;
    // plane row: pixel p, channel quad q at float offset 16 p + 4 (q ^ ((p >> 1) & 3)).  ds_write_b128 is serviced in groups of 8 lanes = 8 consecutive
    // pixels of ONE quad here: unswizzled they start on two bank groups only (4-way conflicts: 875 conflict cycles per row, profiles/r06_fused_pmc_v7*.txt);
    // the swizzle spreads them over all eight, and a pixel's four quads stay inside its own 64 bytes (the combination's reads do not change banks)
    const int z_px = pt * 16 + r16, z_off = z_px * UF_CS + 4 * (kq ^ ((z_px >> 1) & 3));
    const int ldw = Cin + UF_WRES_PAD, w_off_res = r16 * ldw + 4 * kq;
    auto rd = [&](float4& fx, float4 (&fw)[9], const int sg, const int u, const int cc) __attribute__((always_inline)) {   // cc: the chunk's index in its row (WRES)
      fx = *reinterpret_cast<const float4*>(&Xs[sg][x_off + 16 * u]);
      if (WRES) {
        const float* wp = &Ws[w_off_res + cc * UF_K + 16 * u];
#pragma unroll
        for (int t = 0; t < 9; ++t) fw[t] = *reinterpret_cast<const float4*>(wp + t * 16 * ldw);
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) fw[t] = *reinterpret_cast<const float4*>(&Ws[sg * UF_N * UF_LD + w_off + t * 16 * UF_LD + 16 * u]);
      }
    };
    auto mmj = [&](const float4& fx, const float4 (&fw)[9], const int j) __attribute__((always_inline)) {   // one k-step: 9 independent MFMAs
      const float b = j == 0 ? fx.x : (j == 1 ? fx.y : (j == 2 ? fx.z : fx.w));
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float a = j == 0 ? fw[t].x : (j == 1 ? fw[t].y : (j == 2 ? fw[t].z : fw[t].w));
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    };
    if (WRES) uf_wait(&flags[UF_WREADY], 1);
    uf_wait(&flags[UF_FULL0], 1);
    rd(fx0, fw0, 0, 0, 0);
    int k = 0, c = 0, round = 0;                                                      // chunk g lives in stage g % 3, its round = g / 3 (published values: round + 1)
    // One chunk, the stage index a compile-time constant (three copies of the body: every LDS offset an immediate, no address arithmetic between the
    // MFMAs).  The stage is released as soon as its last fragment read has RETURNED (18 MFMAs into the chunk): a hand-over costs two LDS round trips
    // through a busy queue, and with three stages the staging waves then run up to two chunks ahead.
    auto chunk = [&](auto SG, const int g) __attribute__((always_inline)) {
      constexpr int sg = decltype(SG)::value, sn = (sg + 1) % UF_STAGES;
      const int round_n = sn == 0 ? round + 1 : round;
      rd(fx1, fw1, sg, 1, c);
      const int nxt_target = (g + 1 < total) ? round_n + 1 : 0;                       // (last chunk: nothing to wait for)
      const int seen = uf_peek(&flags[UF_FULL0 + 4 * sn]);                            // asked for ahead of the MFMAs: normally the staging waves are ahead
      __builtin_amdgcn_sched_barrier(0);                                              // (the eleven reads are ISSUED here, 18 MFMAs ahead of the wait in uf_publish)
      mmj(fx0, fw0, 0); mmj(fx0, fw0, 1);
      uf_publish(&flags[UF_EMPTY0 + 4 * sg + wid], round + 1);                        // every fragment of this stage is in registers
      if (seen < nxt_target) uf_wait(&flags[UF_FULL0 + 4 * sn], nxt_target);
      mmj(fx0, fw0, 2); mmj(fx0, fw0, 3);
      rd(fx0, fw0, sn, 0, c == cpt - 1 ? 0 : c + 1);                                  // (unconditional: behind the last chunk it reads a stale stage, unused)
      __builtin_amdgcn_sched_barrier(0);                                              // (... and these 36 MFMAs ahead of their first use)
      mmj(fx1, fw1, 0); mmj(fx1, fw1, 1); mmj(fx1, fw1, 2); mmj(fx1, fw1, 3);
      round = round_n;
      if (c == cpt - 1) {
        // row k of the nine planes -> LDS (lane: pixel pt 16 + r16, channels 4 kq .. + 3 of tap t) once the combination waves have read row k - 1
        if (k >= 1) uf_wait(&flags[UF_CDONE], k);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          *reinterpret_cast<float4*>(&Zr[t * UF_PLANE + z_off]) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
        }
        ++k;
        uf_publish(&flags[UF_ZFULL + wid], k);                                        // rows 0 .. k - 1 of this wave's pixel tile are written
        c = 0;
      } else {
        ++c;
      }
    };
    for (int g = 0; g < total; g += 3) {
      chunk(std::integral_constant<int, 0>{}, g);
      if (g + 1 < total) chunk(std::integral_constant<int, 1>{}, g + 1);
      if (g + 2 < total) chunk(std::integral_constant<int, 2>{}, g + 2);
    }
    return;
  }

  if (wid < UF_MW + UF_SW) {
    // ------------------------------------------------------------------------------------------------ staging waves: global -> registers -> LDS
    // x row k of the group = 64 pixels x 32 floats per chunk = 512 quads (item e: pixel e / 8, k quad e % 8); weight slab chunk = 144 rows (tap t,
    // channel c0 + c) x 32 floats = 1152 quads.  256 threads: items st + 256 i -- 2 of x, 4.5 of the weights.  FOUR register sets: the chunk stored in
    // an iteration was requested three iterations earlier (78 KB in flight per CU; one set ahead was latency-bound).
    const int st = tid - UF_MW * 64, sq = st & 7, sr = st >> 3;                       // k quad / row of item 0 (rows sr + 32 i)
    const float* xb[2];
    const float* wb[5];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = sr + 32 * i;
      xb[i] = x + ((long)(n0 + (m >> wshift)) * H * W + (m & (W - 1))) * Cin + sq * 4; // + (k W) Cin + c 32
    }
    const bool w4_on = st < 128;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int nl = (i < 4 || w4_on) ? sr + 32 * i : 0;
      wb[i] = wT + ((long)(nl >> 4) * Cout + c0 + (nl & 15)) * Cin + sq * 4;
    }
    float4 s0[7], s1[7], s2[7], s3[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) s0[i] = s1[i] = s2[i] = s3[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](const int g, float4 (&r)[7]) __attribute__((always_inline)) {
      const int kk = g / cpt, cc = g - kk * cpt;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (!(dbg & 8)) r[i] = *reinterpret_cast<const float4*>(xb[i] + (long)kk * W * Cin + cc * UF_K);
      if (!WRES) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!(dbg & 16)) r[2 + i] = *reinterpret_cast<const float4*>(wb[i] + cc * UF_K);
        if (w4_on && !(dbg & 16)) r[6] = *reinterpret_cast<const float4*>(wb[4] + cc * UF_K);
      }
    };
    auto store = [&](const int sg, const float4 (&r)[7]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&Xs[sg][(sr + 32 * i) * UF_LD + sq * 4]) = r[i];
      if (!WRES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Ws[sg * UF_N * UF_LD + (sr + 32 * i) * UF_LD + sq * 4]) = r[2 + i];
        if (w4_on) *reinterpret_cast<float4*>(&Ws[sg * UF_N * UF_LD + (sr + 128) * UF_LD + sq * 4]) = r[6];
      }
    };
    if (WRES) {
      // the whole slab, once: 144 rows x Cin / 4 quads over the 256 staging threads
      const int qpr = Cin >> 2, ldw = Cin + UF_WRES_PAD;
      for (int e = st; e < UF_N * qpr; e += UF_SW * 64) {
        const int row = e / qpr, q4 = e - row * qpr;
        const float4 v = *reinterpret_cast<const float4*>(wT + ((long)(row >> 4) * Cout + c0 + (row & 15)) * Cin + q4 * 4);
        *reinterpret_cast<float4*>(&Ws[row * ldw + q4 * 4]) = v;
      }
      uf_publish(&flags[UF_WREADY + (wid - UF_MW)], 1);
      // ... then only x per chunk (2 quads per thread): EIGHT register sets, the chunk stored in an iteration was requested seven iterations earlier
      // (x comes from HBM for the first slab of a group to ask; three chunks ahead left the MFMA waves waiting: 160 us on conv2d_7)
      float4 q0[2], q1[2], q2[2], q3[2], q4[2], q5[2], q6[2], q7[2];
      auto xload = [&](const int g, float4 (&r)[2]) __attribute__((always_inline)) {
        const int kk = g / cpt, cc = g - kk * cpt;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (!(dbg & 8)) r[i] = *reinterpret_cast<const float4*>(xb[i] + (long)kk * W * Cin + cc * UF_K);
      };
      int sgx = 0, roundx = 0;
      auto xstep = [&](const int g, const float4 (&cur)[2], float4 (&nxt)[2]) __attribute__((always_inline)) {
        if (g + 7 < total) xload(g + 7, nxt);
        if (roundx >= 1) uf_wait<4>(&flags[UF_EMPTY0 + 4 * sgx], roundx);
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&Xs[sgx][(sr + 32 * i) * UF_LD + sq * 4]) = cur[i];
        uf_publish(&flags[UF_FULL0 + 4 * sgx + (wid - UF_MW)], roundx + 1);
        if (++sgx == UF_STAGES) { sgx = 0; ++roundx; }
      };
#pragma unroll
      for (int i = 0; i < 2; ++i) q0[i] = q1[i] = q2[i] = q3[i] = q4[i] = q5[i] = q6[i] = q7[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      xload(0, q0);
      if (total > 1) xload(1, q1);
      if (total > 2) xload(2, q2);
      if (total > 3) xload(3, q3);
      if (total > 4) xload(4, q4);
      if (total > 5) xload(5, q5);
      if (total > 6) xload(6, q6);
      for (int g = 0; g < total; g += 8) {
        xstep(g, q0, q7);
        if (g + 1 < total) xstep(g + 1, q1, q0);
        if (g + 2 < total) xstep(g + 2, q2, q1);
        if (g + 3 < total) xstep(g + 3, q3, q2);
        if (g + 4 < total) xstep(g + 4, q4, q3);
        if (g + 5 < total) xstep(g + 5, q5, q4);
        if (g + 6 < total) xstep(g + 6, q6, q5);
        if (g + 7 < total) xstep(g + 7, q7, q6);
      }
      return;
    }
    int sg = 0, round = 0;
    auto step = [&](const int g, const float4 (&cur)[7], float4 (&nxt)[7]) __attribute__((always_inline)) {
      if (g + 3 < total && !(dbg & 2)) load(g + 3, nxt);                              // into the set the previous iteration has just stored
      if (round >= 1) uf_wait<4>(&flags[UF_EMPTY0 + 4 * sg], round);                     // chunk g - 3 (same stage) has been consumed by all four MFMA waves
      store(sg, cur);
      uf_publish(&flags[UF_FULL0 + 4 * sg + (wid - UF_MW)], round + 1);
      if (++sg == UF_STAGES) { sg = 0; ++round; }
    };
    load(0, s0);
    if (total > 1) load(1, s1);
    if (total > 2) load(2, s2);
    for (int g = 0; g < total; g += 4) {
      step(g, s0, s3);
      if (g + 1 < total) step(g + 1, s1, s0);
      if (g + 2 < total) step(g + 2, s2, s1);
      if (g + 3 < total) step(g + 3, s3, s2);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- combination waves: plane row -> y
  // thread = (pixel m, channel quad q).  Per low-resolution row k it reads its two columns of each of the nine planes (18 x 16 bytes), folds them along
  // the columns (H_r^b: tap row r, output-column parity b) and releases the row; output rows 2i, 2i + 1 (i = k - 1) are then
  //   bias + sum_r  wlo_r[a] H_r^b[row lo_r] + whi_r[a] H_r^b[row hi_r],   (lo, hi) = (i - 1, i) for r = 0, (i, min(i + 1, H - 1)) for r = 1, 2
  // from the folded rows k (fresh), k - 1 and -- tap row 0 only -- k - 2, which live in registers.
  const int ct = tid - (UF_MW + UF_SW) * 64;                                          // 0 .. 255
  const int cq = ct & 3, cm = ct >> 2;
  const int cg = cm >> wshift, cj = cm & (W - 1);
  AxisW S[3];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) S[sx] = up2_axis(sx, cj, W);
  const float4 bv = bias != nullptr ? *reinterpret_cast<const float4*>(bias + c0 + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pwv[4];
  if (PROJ) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float* pr = pw + (long)(c0 + cq * 4 + c4) * pco;
      pwv[c4] = make_float4(pr[0], pco > 1 ? pr[1] : 0.f, pco > 2 ? pr[2] : 0.f, pco > 3 ? pr[3] : 0.f);
    }
  }
  const float4* Zq = reinterpret_cast<const float4*>(Zr);                             // + plane base / 4 + zoff(column)
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 hn[3][2];                                                                    // folded row k (fresh)
  float4 hc[3][2] = {{zero4, zero4}, {zero4, zero4}, {zero4, zero4}};                 // folded row k - 1
  float4 h0m[2] = {zero4, zero4};                                                     // tap row 0 of row k - 2
  auto emit = [&](const int i, const float4 (&n1)[2], const float4 (&n2)[2]) __attribute__((always_inline)) {   // block row i from h0m, hc and the rows below (n1, n2)
    const AxisW R0 = up2_axis(0, i, H), R1 = up2_axis(1, i, H), R2 = up2_axis(2, i, H);
#pragma unroll
    for (int ca = 0; ca < 2; ++ca) {
      float4 o[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float4 v = bv;
        v = f4_fma(R0.wlo[ca], h0m[b], v); v = f4_fma(R0.whi[ca], hc[0][b], v);
        v = f4_fma(R1.wlo[ca], hc[1][b], v); v = f4_fma(R1.whi[ca], n1[b], v);
        v = f4_fma(R2.wlo[ca], hc[2][b], v); v = f4_fma(R2.whi[ca], n2[b], v);
        o[b] = make_float4(ladder_act_fn(v.x, act), ladder_act_fn(v.y, act), ladder_act_fn(v.z, act), ladder_act_fn(v.w, act));
      }
      const long opix = ((long)(n0 + cg) * 2 * H + 2 * i + ca) * 2 * W + 2 * cj;        // output pixel of column parity 0; parity 1 is the next one
      if (y != nullptr && !(dbg & 32)) {
        float4* yp = reinterpret_cast<float4*>(y + opix * Cout + c0) + cq;
        st_stream(yp, o[0]);
        st_stream(yp + (Cout >> 2), o[1]);
      }
      if (PROJ) {
        float4 p0 = zero4, p1 = zero4;
        p0 = f4_fma(o[0].x, pwv[0], p0); p0 = f4_fma(o[0].y, pwv[1], p0); p0 = f4_fma(o[0].z, pwv[2], p0); p0 = f4_fma(o[0].w, pwv[3], p0);
        p1 = f4_fma(o[1].x, pwv[0], p1); p1 = f4_fma(o[1].y, pwv[1], p1); p1 = f4_fma(o[1].z, pwv[2], p1); p1 = f4_fma(o[1].w, pwv[3], p1);
        float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {                                                 // the 4 quads of a pixel are 4 consecutive lanes: fixed-order sum
          pv[e] += __shfl_xor(pv[e], 1, 64);
          pv[e] += __shfl_xor(pv[e], 2, 64);
        }
        // every lane of the quad holds all sums: lane cq = o stores output channel o of the pixel pair (columns 2j, 2j + 1) as ONE 8-byte piece into
        // the partial plane [slab][o][pixel] -- 16 pixels of a wave make 128 contiguous bytes (12-byte pieces per pixel cost 250 us on conv2d_7)
        if (cq < pco) {
          const float a0 = cq == 0 ? pv[0] : (cq == 1 ? pv[1] : (cq == 2 ? pv[2] : pv[3]));
          const float a1 = cq == 0 ? pv[4] : (cq == 1 ? pv[5] : (cq == 2 ? pv[6] : pv[7]));
          *reinterpret_cast<float2*>(ppart + ((long)slab * pco + cq) * ((long)N * 4 * H * W) + opix) = make_float2(a0, a1);
        }
      }
    }
  };
  // float4 offsets of this thread's quad in its two columns of a plane (swizzled: see the MFMA waves' z_off)
  auto zoff = [&](const int col) __attribute__((always_inline)) { const int p_ = (cg << wshift) + col; return p_ * 4 + (cq ^ ((p_ >> 1) & 3)); };
  const int plo[3] = {zoff(S[0].lo), zoff(S[1].lo), zoff(S[2].lo)};
  const int phi[3] = {zoff(S[0].hi), zoff(S[1].hi), zoff(S[2].hi)};
  for (int k = 0; k < H; ++k) {
    uf_wait<6>(&flags[UF_ZFULL], k + 1);
    if (dbg & 1) { uf_publish(&flags[UF_CDONE + (wid - UF_MW - UF_SW)], k + 1); continue; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float4 h0 = zero4, h1 = zero4;
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) {
        const float4* zp = Zq + ((r * 3 + sx) * UF_PLANE >> 2);
        const float4 v0 = zp[plo[sx]], v1 = zp[phi[sx]];
        h0 = f4_fma(S[sx].wlo[0], v0, h0); h0 = f4_fma(S[sx].whi[0], v1, h0);
        h1 = f4_fma(S[sx].wlo[1], v0, h1); h1 = f4_fma(S[sx].whi[1], v1, h1);
      }
      hn[r][0] = h0;
      hn[r][1] = h1;
    }
    uf_publish(&flags[UF_CDONE + (wid - UF_MW - UF_SW)], k + 1);                      // row k has been read: the MFMA waves may write row k + 1
    if (k >= 1) emit(k - 1, hn[1], hn[2]);
    h0m[0] = hc[0][0]; h0m[1] = hc[0][1];
#pragma unroll
    for (int r = 0; r < 3; ++r) { hc[r][0] = hn[r][0]; hc[r][1] = hn[r][1]; }
  }
  if (!(dbg & 1)) emit(H - 1, hc[1], hc[2]);                                          // the last block row: its "row below" is the clamped last row itself
}

// ---- the same kernel with a 32-PIXEL wave tile (Cin > 128: the weight slab is re-staged for every row) -----------------------------------------------------------
// A row step covers 128 pixels (row k of 2 x 64 / W images): an MFMA wave owns TWO pixel tiles, so one weight fragment feeds 8 MFMAs instead of 4 -- half the
// fragment reads per MFMA and half the slab traffic per flop of the 16-pixel tile (profiles/r06_fused_probe.txt: conv2d_6 / conv2d_5 ran at 0.60 of peak there).
// LDS: two x rows per stage and a 128-pixel plane row leave room for TWO stages (160.9 KB); registers: 72 accumulators, so the fragments are not double-buffered
// as a set but ROLLED -- tap t's weight fragment of the NEXT 16-deep group is requested right after tap t's eight MFMAs of the current one, into the register those
// MFMAs have just released (at any time nine weight fragments are live).  A stage is released five tap blocks into its second group with an explicit
// `s_waitcnt lgkmcnt(2)`: LDS returns in order, the stage's last read is at least seven requests old by then.  No RGB projection (that pair has Cin = 128).
constexpr int UF2_PX = 128, UF2_PLANE = UF2_PX * UF_CS, UF2_STAGES = 2;
enum { UF2_FULL0 = 0, UF2_EMPTY0 = 4 * UF2_STAGES, UF2_ZFULL = 8 * UF2_STAGES, UF2_CDONE = 8 * UF2_STAGES + 4, UF2_NFLAGS = 8 * UF2_STAGES + 8 };

__global__ __launch_bounds__(UF_THREADS, 3) void up2proj_fused2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wT, const float* __restrict__ bias,
                                                                           float* __restrict__ y, const int N, const int H, const int W, const int wshift,
                                                                           const int Cin, const int Cout, const int act, const int poll) {
  __shared__ __attribute__((aligned(16))) float Xs[UF2_STAGES][UF2_PX * UF_LD];
  __shared__ __attribute__((aligned(16))) float Ws[UF2_STAGES][UF_N * UF_LD];
  __shared__ __attribute__((aligned(16))) float Zr[9 * UF2_PLANE];
  __shared__ __attribute__((aligned(16))) int flags[UF2_NFLAGS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int G = UF2_PX >> wshift, ngroups = N / G, nslab = Cout / UF_CS;
  int group, slab;                                                                    // (as in the 64-pixel kernel: the slabs of a group on one XCD)
  if ((ngroups & 7) == 0) {
    const int xcd = (int)(blockIdx.x & 7), l = (int)(blockIdx.x >> 3);
    group = (l / nslab) * 8 + xcd;
    slab = l % nslab;
  } else {
    group = (int)blockIdx.x / nslab;
    slab = (int)blockIdx.x % nslab;
  }
  const int n0 = group * G, c0 = slab * UF_CS;
  const int cpt = Cin / UF_K, total = H * cpt;
  if (tid < UF2_NFLAGS) flags[tid] = 0;
  __syncthreads();

  if (wid < UF_MW) {
    // ------------------------------------------------------------------------------------------------ MFMA waves: pixel tiles wid and wid + 4
    __builtin_amdgcn_s_setprio(2);
    const int pt = wid;
    f32x4_t acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int p_ = 0; p_ < 2; ++p_)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][p_][e] = 0.f;
    float4 fx[2], nfx[2], fw[9];
    const int x_off0 = (pt * 16 + r16) * UF_LD + 4 * kq, x_off1 = x_off0 + 64 * UF_LD, w_off = r16 * UF_LD + 4 * kq;
    const int z_px0 = pt * 16 + r16, z_off0 = z_px0 * UF_CS + 4 * (kq ^ ((z_px0 >> 1) & 3));     // (pixel + 64 has the same swizzle)
    auto rdx = [&](float4 (&d)[2], const int sg, const int u) __attribute__((always_inline)) {
      d[0] = *reinterpret_cast<const float4*>(&Xs[sg][x_off0 + 16 * u]);
      d[1] = *reinterpret_cast<const float4*>(&Xs[sg][x_off1 + 16 * u]);
    };
    auto rdw = [&](const int t, const int sg, const int u) __attribute__((always_inline)) {
      fw[t] = *reinterpret_cast<const float4*>(&Ws[sg][w_off + t * 16 * UF_LD + 16 * u]);
    };
    auto mmt = [&](const int t) __attribute__((always_inline)) {                      // tap t of the current group: 4 k-steps x 2 pixel tiles
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = j == 0 ? fw[t].x : (j == 1 ? fw[t].y : (j == 2 ? fw[t].z : fw[t].w));
#pragma unroll
        for (int p_ = 0; p_ < 2; ++p_) {
          const float b = j == 0 ? fx[p_].x : (j == 1 ? fx[p_].y : (j == 2 ? fx[p_].z : fx[p_].w));
          acc[t][p_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t][p_], 0, 0, 0);
        }
      }
    };
    uf_wait(&flags[UF2_FULL0], 1);
    rdx(fx, 0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) rdw(t, 0, 0);
    int k = 0, c = 0, round = 0;                                                      // chunk g lives in stage g & 1, its round = g >> 1
    auto chunk = [&](auto SG, const int g) __attribute__((always_inline)) {
      constexpr int sg = decltype(SG)::value, sn = sg ^ 1;
      const int round_n = sn == 0 ? round + 1 : round;
      // group 0: its fragments are in registers; the requests issued beside its MFMAs are those of group 1 of the same stage
      const int seen = uf_peek(&flags[UF2_FULL0 + 4 * sn]);
      rdx(nfx, sg, 1);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        mmt(t);
        rdw(t, sg, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      fx[0] = nfx[0]; fx[1] = nfx[1];
      // group 1: the requests beside its MFMAs go to the NEXT stage
      const int nxt_target = (g + 1 < total) ? round_n + 1 : 0;
      if (seen < nxt_target) uf_wait(&flags[UF2_FULL0 + 4 * sn], nxt_target);
      rdx(nfx, sn, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        mmt(t);
        rdw(t, sn, 0);
        if (t == 4) {
          // every fragment read of stage sg is >= 7 requests old (2 of x + 5 of the weights of the next stage since): release it without waiting for the newest two
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
          if (lane == 0) __hip_atomic_store(&flags[UF2_EMPTY0 + 4 * sg + wid], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      fx[0] = nfx[0]; fx[1] = nfx[1];
      round = round_n;
      if (c == cpt - 1) {
        if (k >= 1) uf_wait(&flags[UF2_CDONE], k);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int p_ = 0; p_ < 2; ++p_) {
            *reinterpret_cast<float4*>(&Zr[t * UF2_PLANE + p_ * 64 * UF_CS + z_off0]) = make_float4(acc[t][p_][0], acc[t][p_][1], acc[t][p_][2], acc[t][p_][3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][p_][e] = 0.f;
          }
        ++k;
        uf_publish(&flags[UF2_ZFULL + wid], k);
        c = 0;
      } else {
        ++c;
      }
    };
    for (int g = 0; g < total; g += 2) {
      chunk(std::integral_constant<int, 0>{}, g);
      if (g + 1 < total) chunk(std::integral_constant<int, 1>{}, g + 1);
    }
    return;
  }

  if (wid < UF_MW + UF_SW) {
    // ------------------------------------------------------------------------------------------------ staging waves: 128 x rows + 144 weight rows per chunk
    const int st = tid - UF_MW * 64, sq = st & 7, sr = st >> 3;
    const float* xb[4];
    const float* wb[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = sr + 32 * i;
      xb[i] = x + ((long)(n0 + (m >> wshift)) * H * W + (m & (W - 1))) * Cin + sq * 4;
    }
    const bool w4_on = st < 128;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int nl = (i < 4 || w4_on) ? sr + 32 * i : 0;
      wb[i] = wT + ((long)(nl >> 4) * Cout + c0 + (nl & 15)) * Cin + sq * 4;
    }
    float4 s0[9], s1[9], s2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) s0[i] = s1[i] = s2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](const int g, float4 (&r)[9]) __attribute__((always_inline)) {
      const int kk = g / cpt, cc = g - kk * cpt;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(xb[i] + (long)kk * W * Cin + cc * UF_K);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[4 + i] = *reinterpret_cast<const float4*>(wb[i] + cc * UF_K);
      if (w4_on) r[8] = *reinterpret_cast<const float4*>(wb[4] + cc * UF_K);
    };
    int sg = 0, round = 0;
    auto step = [&](const int g, const float4 (&cur)[9], float4 (&nxt)[9]) __attribute__((always_inline)) {
      if (g + 2 < total) load(g + 2, nxt);                                            // (a chunk is 4 608 MFMA cycles here: two ahead covers what four covered at 64 pixels)
      if (round >= 1) {
        int spins = 0;
        while (uf_peek(&flags[UF2_EMPTY0 + 4 * sg]) < round) {
          for (int q = 0; q < poll; ++q) __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) __builtin_trap();
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Xs[sg][(sr + 32 * i) * UF_LD + sq * 4]) = cur[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Ws[sg][(sr + 32 * i) * UF_LD + sq * 4]) = cur[4 + i];
      if (w4_on) *reinterpret_cast<float4*>(&Ws[sg][(sr + 128) * UF_LD + sq * 4]) = cur[8];
      uf_publish(&flags[UF2_FULL0 + 4 * sg + (wid - UF_MW)], round + 1);
      if (++sg == UF2_STAGES) { sg = 0; ++round; }
    };
    load(0, s0);
    if (total > 1) load(1, s1);
    for (int g = 0; g < total; g += 3) {
      step(g, s0, s2);
      if (g + 1 < total) step(g + 1, s1, s0);
      if (g + 2 < total) step(g + 2, s2, s1);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- combination waves: two (pixel, quad) items per thread
  const int ct = tid - (UF_MW + UF_SW) * 64;
  const int cq = ct & 3, cm = ct >> 2;                                                // items: pixel cm and pixel cm + 64 (same column, 64 / W images further)
  const int cg = cm >> wshift, cj = cm & (W - 1), gstep = 64 >> wshift;
  AxisW S[3];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) S[sx] = up2_axis(sx, cj, W);
  const float4 bv = bias != nullptr ? *reinterpret_cast<const float4*>(bias + c0 + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* Zq = reinterpret_cast<const float4*>(Zr);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Per item the row halo is carried as the PARTIAL outputs of the open block row (everything that does not depend on the row below: 2 x 2 x 4 values) and
  // the folded tap-row-0 values of the previous row -- 24 registers instead of the 56 of three folded rows (two items per thread have to fit 168 VGPRs).
  float4 pp[2][2][2], h0p[2][2];                                                       // [item][output-row parity][column parity], [item][column parity]
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    h0p[it][0] = h0p[it][1] = zero4;
#pragma unroll
    for (int ca = 0; ca < 2; ++ca) pp[it][ca][0] = pp[it][ca][1] = zero4;
  }
  auto zoff = [&](const int it, const int col) __attribute__((always_inline)) { const int p_ = ((cg + it * gstep) << wshift) + col; return p_ * 4 + (cq ^ ((p_ >> 1) & 3)); };
  // block row i of item `it` is complete: add the terms of the row below (folded tap rows 1, 2: n1, n2), activate, store
  auto finish = [&](const int i, const int it, const float4 (&n1)[2], const float4 (&n2)[2]) __attribute__((always_inline)) {
    const AxisW R1 = up2_axis(1, i, H), R2 = up2_axis(2, i, H);
#pragma unroll
    for (int ca = 0; ca < 2; ++ca) {
      float4 o[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float4 v = pp[it][ca][b];
        v = f4_fma(R1.whi[ca], n1[b], v);
        v = f4_fma(R2.whi[ca], n2[b], v);
        o[b] = make_float4(ladder_act_fn(v.x, act), ladder_act_fn(v.y, act), ladder_act_fn(v.z, act), ladder_act_fn(v.w, act));
      }
      const long opix = ((long)(n0 + cg + it * gstep) * 2 * H + 2 * i + ca) * 2 * W + 2 * cj;
      float4* yp = reinterpret_cast<float4*>(y + opix * Cout + c0) + cq;
      st_stream(yp, o[0]);
      st_stream(yp + (Cout >> 2), o[1]);
    }
  };
  for (int k = 0; k < H; ++k) {
    uf_wait<6>(&flags[UF2_ZFULL], k + 1);
    const AxisW R0 = up2_axis(0, k, H), R1 = up2_axis(1, k, H), R2 = up2_axis(2, k, H);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float4 hn[3][2];                                                                // row k of item `it`, folded along the columns
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float4 h0 = zero4, h1 = zero4;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
          const float4* zp = Zq + ((r * 3 + sx) * UF2_PLANE >> 2);
          const float4 v0 = zp[zoff(it, S[sx].lo)], v1 = zp[zoff(it, S[sx].hi)];
          h0 = f4_fma(S[sx].wlo[0], v0, h0); h0 = f4_fma(S[sx].whi[0], v1, h0);
          h1 = f4_fma(S[sx].wlo[1], v0, h1); h1 = f4_fma(S[sx].whi[1], v1, h1);
        }
        hn[r][0] = h0;
        hn[r][1] = h1;
      }
      if (it == 1) uf_publish(&flags[UF2_CDONE + (wid - UF_MW - UF_SW)], k + 1);     // both items have read row k: the MFMA waves may write row k + 1
      if (k >= 1) finish(k - 1, it, hn[1], hn[2]);
      // open block row k: bias + tap row 0 (rows k - 1, k) + the own-row terms of tap rows 1, 2
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float4 v = bv;
          v = f4_fma(R0.wlo[ca], h0p[it][b], v); v = f4_fma(R0.whi[ca], hn[0][b], v);
          v = f4_fma(R1.wlo[ca], hn[1][b], v);
          v = f4_fma(R2.wlo[ca], hn[2][b], v);
          pp[it][ca][b] = v;
        }
      h0p[it][0] = hn[0][0]; h0p[it][1] = hn[0][1];
      if (k == H - 1) finish(H - 1, it, hn[1], hn[2]);                                // the last block row: its "row below" is the clamped last row itself
      __builtin_amdgcn_sched_barrier(0);                                              // (one item after the other: their temporaries must not be live together)
    }
  }
}

// pout [P][pco] = pb + sum over the slabs (in order) of the partial projections [nslab][pco][P]; one thread per pixel
__global__ __launch_bounds__(256) void up2proj_proj_reduce_kernel(const float* __restrict__ ppart, const float* __restrict__ pb, float* __restrict__ pout,
                                                                  const long P, const int pco, const int nslab) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  for (int o = 0; o < pco; ++o) {
    float sacc = pb != nullptr ? pb[o] : 0.f;
    for (int sl = 0; sl < nslab; ++sl) sacc += ppart[((long)sl * pco + o) * P + t];
    pout[t * pco + o] = sacc;
  }
}

// ... four pixels per thread (P is a multiple of 4): every partial plane is read as float4, the 4 x PCO results leave as PCO float4 -- all loads of a thread
// independent of each other (the scalar form above chains pco x nslab dependent 4-byte loads: 67 us for 0.23 GB at conv2d_7; this one ~45).  Same order of additions.
template <int PCO>
__global__ __launch_bounds__(256) void up2proj_proj_reduce4_kernel(const float* __restrict__ ppart, const float* __restrict__ pb, float* __restrict__ pout,
                                                                   const long P, const int nslab) {
  const long t4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (t4 * 4 >= P) return;
  float4 acc[PCO];
#pragma unroll
  for (int o = 0; o < PCO; ++o) {
    const float b = pb != nullptr ? pb[o] : 0.f;
    acc[o] = make_float4(b, b, b, b);
  }
#pragma unroll 4
  for (int sl = 0; sl < nslab; ++sl)
#pragma unroll
    for (int o = 0; o < PCO; ++o) {
      const float4 v = *reinterpret_cast<const float4*>(ppart + ((long)sl * PCO + o) * P + t4 * 4);
      acc[o].x += v.x; acc[o].y += v.y; acc[o].z += v.z; acc[o].w += v.w;
    }
  float flat[4 * PCO];                                                                 // [pixel][o]
#pragma unroll
  for (int o = 0; o < PCO; ++o) { flat[o] = acc[o].x; flat[PCO + o] = acc[o].y; flat[2 * PCO + o] = acc[o].z; flat[3 * PCO + o] = acc[o].w; }
  float4* op = reinterpret_cast<float4*>(pout + t4 * 4 * PCO);
#pragma unroll
  for (int k = 0; k < PCO; ++k) op[k] = make_float4(flat[4 * k], flat[4 * k + 1], flat[4 * k + 2], flat[4 * k + 3]);
}

// dw [3][3][Cin][Cout] from dWcat [Cin][9 Cout]; db [Cout] (may be NULL) = the centre tap's column sums of D = sum over all pixels of dy
__global__ __launch_bounds__(256) void up2proj_wgrad_unpack_kernel(const float* __restrict__ dwcat, const float* __restrict__ db9, float* __restrict__ dw,
                                                                   float* __restrict__ db, const int Cin, const int Cout) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)9 * Cin * Cout;
  if (t < total) {
    const int co = (int)(t % Cout), ci = (int)((t / Cout) % Cin), tap = (int)(t / ((long)Cout * Cin));
    dw[t] = dwcat[(long)ci * 9 * Cout + (long)tap * Cout + co];
  }
  if (db != nullptr && t < Cout) db[t] = db9[4 * Cout + t];
}


// ---- any integer resize factor F (decoder conv2d_3: the 2x2 -> 8x8 resize, F = 4; codes/models.py:536-542).  Per axis up(Z)[F i + k] = (1 - k/F) Z[i] + (k/F) Z[min(i+1, L-1)]
// (TF1 legacy bilinear: source position = destination / F, no half-pixel offset).  Small maps: one thread per OUTPUT pixel x 4 channels forward,
// one per low-resolution pixel x 4 channels backward; the operands live in L2.
template <int F>
__global__ __launch_bounds__(256) void upfproj_fwd_combine_kernel(const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ y,
                                                                  const int N, const int H, const int W, const int C, const int act) {
  const int CV = C >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)N * F * H * F * W * CV) return;
  const int cq = (int)(t % CV);
  const long pix = t / CV;
  const int q = (int)(pix % (F * W)), p = (int)((pix / (F * W)) % (F * H)), n = (int)(pix / ((long)F * W * F * H));
  const float4* zb = reinterpret_cast<const float4*>(z) + (long)n * H * W * 9 * CV + cq;
  float4 acc = bias != nullptr ? reinterpret_cast<const float4*>(bias)[cq] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int u = p + r - 1;
    if (u < 0 || u >= F * H) continue;
    const int i0 = u / F, kr = u - i0 * F, i1 = min(i0 + 1, H - 1);
    const float wr1 = (float)kr * (1.f / F), wr0 = 1.f - wr1;
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const int v = q + sx - 1;
      if (v < 0 || v >= F * W) continue;
      const int j0 = v / F, kc = v - j0 * F, j1 = min(j0 + 1, W - 1);
      const float wc1 = (float)kc * (1.f / F), wc0 = 1.f - wc1;
      const float4* zp = zb + (r * 3 + sx) * CV;
      acc = f4_fma(wr0 * wc0, zp[((long)i0 * W + j0) * 9 * CV], acc);
      if (kc) acc = f4_fma(wr0 * wc1, zp[((long)i0 * W + j1) * 9 * CV], acc);
      if (kr) acc = f4_fma(wr1 * wc0, zp[((long)i1 * W + j0) * 9 * CV], acc);
      if (kr && kc) acc = f4_fma(wr1 * wc1, zp[((long)i1 * W + j1) * 9 * CV], acc);
    }
  }
  acc = make_float4(ladder_act_fn(acc.x, act), ladder_act_fn(acc.y, act), ladder_act_fn(acc.z, act), ladder_act_fn(acc.w, act));
  reinterpret_cast<float4*>(y)[pix * CV + cq] = acc;
}

// coefficient of Z[i] in up(Z)[u] on an axis of low-resolution length L (u inside [0, F L))
template <int F>
__device__ __forceinline__ float upf_coef(int u, int i, int L) {
  const int i0 = u / F, k = u - i0 * F, i1 = min(i0 + 1, L - 1);
  const float w1 = (float)k * (1.f / F);
  return (i0 == i ? 1.f - w1 : 0.f) + (i1 == i ? w1 : 0.f);
}

template <int F>
__global__ __launch_bounds__(256) void upfproj_bwd_combine_kernel(const float* __restrict__ dy, float* __restrict__ d, const int N, const int H,
                                                                  const int W, const int C) {
  const int CV = C >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)N * H * W * CV) return;
  const int cq = (int)(t % CV);
  const long pix = t / CV;
  const int j = (int)(pix % W), i = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
  const float4* gy = reinterpret_cast<const float4*>(dy) + (long)n * F * H * F * W * CV + cq;
  float4* dp = reinterpret_cast<float4*>(d) + pix * 9 * CV + cq;
  // positions of the upsampled line that read Z[i]: u in (F (i - 1), F (i + 1)); the gradient there through tap r is dy[u - r + 1]
  constexpr int NU = 2 * F - 1;
  float cu[NU], cv[NU];
#pragma unroll
  for (int a = 0; a < NU; ++a) {
    const int u = F * (i - 1) + 1 + a, v = F * (j - 1) + 1 + a;
    cu[a] = (u >= 0 && u < F * H) ? upf_coef<F>(u, i, H) : 0.f;
    cv[a] = (v >= 0 && v < F * W) ? upf_coef<F>(v, j, W) : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // E[b] = sum_u cu[u] dy[u - r + 1][q_b] for the NU + 2 columns q_b = F (j - 1) + b  (b = v_index + 1 - s + ... see below)
    float4 e[NU + 2];
#pragma unroll
    for (int b = 0; b < NU + 2; ++b) {
      const int q = F * (j - 1) + b;                     // q = v - s + 1 with v = F (j - 1) + 1 + a  ->  b = a + 2 - s
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q >= 0 && q < F * W) {
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          const int p = F * (i - 1) + 1 + a - r + 1;
          if (cu[a] != 0.f && p >= 0 && p < F * H) acc = f4_fma(cu[a], gy[((long)p * F * W + q) * CV], acc);
        }
      }
      e[b] = acc;
    }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int a = 0; a < NU; ++a) acc = f4_fma(cv[a], e[a + 2 - sx], acc);
      dp[(r * 3 + sx) * CV] = acc;
    }
  }
}

}  // namespace

extern "C" {

int ladder_up2proj_eligible(int N, int H, int W, int Cin, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_UP2PROJ") != nullptr;
  return (!off && N > 0 && H >= 1 && W >= 1 && Cin > 0 && Cout > 0 && (Cin % 16) == 0 && (Cout % 16) == 0 && (long)N * H * W < (1L << 30)) ? 1 : 0;
}

int ladder_up2proj_fwd_combine(const float* z, const float* bias, float* y, const float* proj_w, const float* proj_b, float* proj_out, int proj_cout,
                               int N, int H, int W, int C, int act, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) != 0 || (y == nullptr && proj_out == nullptr)) return LADDER_E_SHAPE;
  if (proj_out != nullptr && (proj_w == nullptr || proj_cout < 1 || proj_cout > 4 || C != 128)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(z) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias))) return LADDER_E_ALIGN;
  static const int seg_env = getenv("LADDER_UP2PROJ_SEG") != nullptr ? atoi(getenv("LADDER_UP2PROJ_SEG")) : -1;
  // rows per thread: 16, fewer while the launch would not fill the chip with >= 4 waves per SIMD (256 K threads); 0 (env) = the one-pixel kernel
  int seg = 16;
  while (seg > 2 && (long)N * ((H + seg - 1) / seg) * W * (C / 4) < (1L << 18)) seg >>= 1;
  if (seg_env >= 0) seg = seg_env;
  if (seg >= 2 && H >= 2) {
    const long total = (long)N * ((H + seg - 1) / seg) * W * (C / 4);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (proj_out != nullptr)
      hipLaunchKernelGGL(up2proj_fwd_combine_rows_kernel<true>, dim3(grid), dim3(256), 0, stream, z, bias, y, proj_w, proj_b, proj_out, proj_cout, N, H, W, C, act, seg);
    else
      hipLaunchKernelGGL(up2proj_fwd_combine_rows_kernel<false>, dim3(grid), dim3(256), 0, stream, z, bias, y, (const float*)nullptr, (const float*)nullptr,
                         (float*)nullptr, 0, N, H, W, C, act, seg);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  const long total = (long)N * H * W * (C / 4);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (proj_out != nullptr)
    hipLaunchKernelGGL(up2proj_fwd_combine_kernel<true>, dim3(grid), dim3(256), 0, stream, z, bias, y, proj_w, proj_b, proj_out, proj_cout, N, H, W, C, act);
  else
    hipLaunchKernelGGL(up2proj_fwd_combine_kernel<false>, dim3(grid), dim3(256), 0, stream, z, bias, y, (const float*)nullptr, (const float*)nullptr,
                       (float*)nullptr, 0, N, H, W, C, act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// rows a thread of the walking form covers (LADDER_UP2BWD_SEG overrides)
static int up2bwd_walk_seg(int N, int H, int W, int C) {
  static const int forced = getenv("LADDER_UP2BWD_SEG") != nullptr ? atoi(getenv("LADDER_UP2BWD_SEG")) : 0;
  if (forced > 0) return forced < H ? forced : H;
  // (measured, batch 128, profiles/r06_bwd_walk_probe.txt: long segments win -- the 3-row warm-up is the cost -- as long as every CU has two workgroups;
  // conv2d_7 under the projection: whole columns 717 us, 16 rows 815, 8 rows 920: r06_bwd_walk_probe_seg.txt)
  int seg = H;
  while (seg > 4 && (long)N * ((H + seg - 1) / seg) * W * (C / 4) < 256L * 512) seg = (seg + 1) >> 1;   // two workgroups of 256 threads per CU
  return seg;
}

// the walking form of ladder_up2proj_bwd_combine (same arguments; rows_per_thread 0 = chosen here)
int ladder_up2proj_bwd_combine_walk(const float* dy, float* d, int N, int H, int W, int C, int rows_per_thread, ladder_stream_t stream) {
  if (N <= 0 || H < 4 || W <= 0 || C <= 0 || (C % 4) != 0 || rows_per_thread < 0 || (long)N * H * W * 4 >= (1L << 30)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(dy) || !ladder_aligned16(d)) return LADDER_E_ALIGN;
  const int seg = rows_per_thread > 0 ? (rows_per_thread < H ? rows_per_thread : H) : up2bwd_walk_seg(N, H, W, C);
  const long total = (long)N * ((H + seg - 1) / seg) * W * (C / 4);
  hipLaunchKernelGGL(up2proj_bwd_combine_walk_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dy, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, d, (float*)nullptr, N, H, W, C, seg, 0);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_up2proj_bwd_combine(const float* dy, float* d, int N, int H, int W, int C, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) != 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(dy) || !ladder_aligned16(d)) return LADDER_E_ALIGN;
  // the walking form wherever it applies (conv2d_6 173 -> 133 us, conv2d_5 95 -> 74, conv2d_4 27 -> 21 at batch 128); LADDER_UP2BWD_WALK=0: the neighbourhood kernel
  static const bool walk = getenv("LADDER_UP2BWD_WALK") == nullptr || atoi(getenv("LADDER_UP2BWD_WALK")) != 0;
  if (walk && H >= 4 && (long)N * H * W * 4 < (1L << 30)) return ladder_up2proj_bwd_combine_walk(dy, d, N, H, W, C, 0, stream);
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(up2proj_bwd_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dy, d, N, H, W, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// D [N, H, W, 9 C] of a pair whose activated output y [N, 2H, 2W, C] feeds a 1x1 convolution to pco <= 4 channels (pw [C][pco]), straight from the gradient
// dyp [N, 2H, 2W, pco] of that convolution's output: dy = act'(y) * (dyp . pw^T) is formed inside and never stored; dpw [C][pco] / dpb [pco] (may be NULL)
// receive the 1x1 convolution's filter / bias gradient (overwritten).  Replaces ladder_conv1x1_smallcout_bwd + ladder_up2proj_bwd_combine.
int ladder_up2proj_bwd_combine_proj_eligible(int N, int H, int W, int C, int pco) {
  static const bool off = getenv("LADDER_DISABLE_UP2BWD_PROJ") != nullptr;
  const int cv = C / 4;
  return (!off && N > 0 && H >= 4 && W > 0 && C > 0 && (C % 4) == 0 && cv <= 64 && (cv & (cv - 1)) == 0 && pco >= 1 && pco <= 4 && cv >= 5 * pco &&
          (long)N * H * W * 4 < (1L << 30)) ? 1 : 0;
}

static long up2bwd_proj_blocks(int N, int H, int W, int C) {
  const int seg = up2bwd_walk_seg(N, H, W, C);
  return ((long)N * ((H + seg - 1) / seg) * W * (C / 4) + 255) / 256;
}

size_t ladder_up2proj_bwd_combine_proj_workspace_bytes(int N, int H, int W, int C, int pco) {
  if (!ladder_up2proj_bwd_combine_proj_eligible(N, H, W, C, pco)) return 0;
  return (size_t)up2bwd_proj_blocks(N, H, W, C) * ((size_t)C * pco + pco) * sizeof(float);
}

int ladder_up2proj_bwd_combine_proj(const float* y, const float* dyp, const float* pw, float* d, float* dpw, float* dpb, int pco, int N, int H, int W, int C,
                                    int act, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (!ladder_up2proj_bwd_combine_proj_eligible(N, H, W, C, pco) || y == nullptr || dyp == nullptr || pw == nullptr || d == nullptr || dpw == nullptr ||
      (act != LADDER_ACT_NONE && act != LADDER_ACT_LEAKY && act != LADDER_ACT_RELU))
    return LADDER_E_SHAPE;
  if (!ladder_aligned16(y) || !ladder_aligned16(d)) return LADDER_E_ALIGN;
  if (ws == nullptr || ws_bytes < ladder_up2proj_bwd_combine_proj_workspace_bytes(N, H, W, C, pco)) return LADDER_E_WORKSPACE;
  const int seg = up2bwd_walk_seg(N, H, W, C);
  const unsigned blocks = (unsigned)up2bwd_proj_blocks(N, H, W, C);
  float* part = (float*)ws;
#define LADDER_UP2BWD_PROJ(P_) hipLaunchKernelGGL(up2proj_bwd_combine_walk_kernel<P_>, dim3(blocks), dim3(256), 0, stream, (const float*)nullptr, y, dyp, pw, d, \
                                                  part, N, H, W, C, seg, act)
  switch (pco) { case 1: LADDER_UP2BWD_PROJ(1); break; case 2: LADDER_UP2BWD_PROJ(2); break; case 3: LADDER_UP2BWD_PROJ(3); break; default: LADDER_UP2BWD_PROJ(4); }
#undef LADDER_UP2BWD_PROJ
  const int kn = C * pco;
  hipLaunchKernelGGL(up2proj_part_reduce_kernel, dim3((unsigned)((kn + pco + 3) / 4)), dim3(256), 0, stream, (const float*)part, dpw, dpb, (int)blocks, kn, pco);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_up2proj_wgrad_unpack(const float* dwcat, const float* db9, float* dw, float* db, int Cin, int Cout, ladder_stream_t stream) {
  if (Cin <= 0 || Cout <= 0 || dwcat == nullptr || dw == nullptr || (db != nullptr && db9 == nullptr)) return LADDER_E_SHAPE;
  const long total = (long)9 * Cin * Cout;
  hipLaunchKernelGGL(up2proj_wgrad_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dwcat, db9, dw, db, Cin, Cout);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// Any resize factor of {2, 4}: y [N, F H, F W, C] = act(bias + sum_rs shift_rs(up_F(Z_rs))) and d = (shift o up_F)^T dy.  Factor 2 takes the kernels above.
int ladder_upfproj_eligible(int factor, int N, int H, int W, int Cin, int Cout) {
  return ((factor == 2 || factor == 4) && ladder_up2proj_eligible(N, H, W, Cin, Cout) && (long)N * H * W * factor * factor < (1L << 30)) ? 1 : 0;
}

int ladder_upfproj_fwd_combine(const float* z, const float* bias, float* y, int factor, int N, int H, int W, int C, int act, ladder_stream_t stream) {
  if (factor == 2) return ladder_up2proj_fwd_combine(z, bias, y, nullptr, nullptr, nullptr, 0, N, H, W, C, act, stream);
  if (factor != 4 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) != 0 || y == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(z) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias))) return LADDER_E_ALIGN;
  const long total = (long)N * 16 * H * W * (C / 4);
  hipLaunchKernelGGL(upfproj_fwd_combine_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, bias, y, N, H, W, C, act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_upfproj_bwd_combine(const float* dy, float* d, int factor, int N, int H, int W, int C, ladder_stream_t stream) {
  if (factor == 2) return ladder_up2proj_bwd_combine(dy, d, N, H, W, C, stream);
  if (factor != 4 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) != 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(dy) || !ladder_aligned16(d)) return LADDER_E_ALIGN;
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(upfproj_bwd_combine_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dy, d, N, H, W, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// ---- the fused forward of a pair (round 6) -------------------------------------------------------------------------------------------------------------
// y [N, 2H, 2W, Cout] = act(bias + sum_rs shift_rs(up(x . w_rs))) straight from x [N, H, W, Cin] and wcatT [9 Cout][Cin] (orientation 7 of filterbank.h):
// the nine planes live only in LDS.  proj_out != NULL: also the 1x1 conv of the activated value (proj_w [Cout][proj_cout], proj_cout <= 4) through
// per-slab partials in `ws`; y may then be NULL (forward-only runs).
int ladder_up2proj_fused_eligible(int N, int H, int W, int Cin, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_UP2FUSE") != nullptr;
  if (off || N <= 0 || H < 2 || Cin < 2 * UF_K || (Cin % UF_K) != 0 || (Cout % UF_CS) != 0) return 0;
  if (W != 8 && W != 16 && W != 32 && W != 64) return 0;
  if (N % (UF_PX / W) != 0 || (long)N * H * W * 4 >= (1L << 30)) return 0;
  return 1;
}

// ... and PREFERRED over the two-call form as an isolated launch (measured, batch 128, profiles/r06_fused_probe.txt): where the weight slab stays in LDS (Cin <= 128:
// conv2d_7 1 559 against 2 022 us, with the RGB projection 1 809 / 1 670 against 2 062 / 1 950), on the 8-pixel-wide maps (conv2d_4: 199 against 226 us) and where the
// 128-pixel row step applies (conv2d_6 705 against 799 us, conv2d_5 340 against 398).  (The engine's default fuses every ELIGIBLE pair.)
int ladder_up2proj_fused_wide_tile(int N, int H, int W, int Cin, int Cout);
int ladder_up2proj_fused_preferred(int N, int H, int W, int Cin, int Cout) {
  return (ladder_up2proj_fused_eligible(N, H, W, Cin, Cout) && (Cin <= UF_WRES_CIN || W <= 8 || ladder_up2proj_fused_wide_tile(N, H, W, Cin, Cout))) ? 1 : 0;
}

// 1 when ladder_up2proj_fused_fwd (without projection) runs the 128-pixel row step / 32-pixel wave tile kernel: the slab cannot stay resident (Cin > 128), two
// 64-pixel groups exist per row step (W 16 or 32) and the grid still covers the chip
int ladder_up2proj_fused_wide_tile(int N, int H, int W, int Cin, int Cout) {
  static const bool pt2_off = getenv("LADDER_UP2FUSE_NO_PT2") != nullptr;
  static const bool any = getenv("LADDER_UP2FUSE_PT2_ANY") != nullptr;                // (experiment switch: also where the slab would fit, and on 64-pixel-wide maps)
  return (!pt2_off && ladder_up2proj_fused_eligible(N, H, W, Cin, Cout) && (any || (Cin > UF_WRES_CIN && (W == 16 || W == 32))) && W <= 64 && N % (UF2_PX / W) == 0 &&
          (long)(N / (UF2_PX / W)) * (Cout / UF_CS) >= 256) ? 1 : 0;
}

size_t ladder_up2proj_fused_workspace_bytes(int N, int H, int W, int Cout, int proj_cout) {
  return proj_cout > 0 ? (size_t)(Cout / UF_CS) * N * 4 * H * W * proj_cout * sizeof(float) : 0;
}

int ladder_up2proj_fused_fwd(const float* x, const float* wcatT, const float* bias, float* y, const float* proj_w, const float* proj_b, float* proj_out,
                             int proj_cout, int N, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (!ladder_up2proj_fused_eligible(N, H, W, Cin, Cout) || x == nullptr || wcatT == nullptr || (y == nullptr && proj_out == nullptr)) return LADDER_E_SHAPE;
  if (proj_out != nullptr && (proj_w == nullptr || proj_cout < 1 || proj_cout > 4)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(wcatT) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias))) return LADDER_E_ALIGN;
  if (proj_out != nullptr && (ws == nullptr || ws_bytes < ladder_up2proj_fused_workspace_bytes(N, H, W, Cout, proj_cout))) return LADDER_E_WORKSPACE;
  int wshift = 3;
  while ((1 << wshift) < W) ++wshift;
  static const int dbg = getenv("LADDER_UP2FUSE_DBG") != nullptr ? atoi(getenv("LADDER_UP2FUSE_DBG")) : 0;
  const unsigned grid = (unsigned)(N / (UF_PX / W)) * (unsigned)(Cout / UF_CS);
  // 128-pixel row steps (32-pixel wave tiles) where the slab cannot stay resident, two images fill the row step and the grid still covers the chip
  if (proj_out == nullptr && y != nullptr && dbg == 0 && ladder_up2proj_fused_wide_tile(N, H, W, Cin, Cout)) {
    const unsigned grid2 = (unsigned)(N / (UF2_PX / W)) * (unsigned)(Cout / UF_CS);
    static const int poll = getenv("LADDER_UP2FUSE_POLL") != nullptr ? atoi(getenv("LADDER_UP2FUSE_POLL")) : 4;   // staging waves: x 64 cycles between polls
    hipLaunchKernelGGL(up2proj_fused2_fwd_kernel, dim3(grid2), dim3(UF_THREADS), 0, stream, x, wcatT, bias, y, N, H, W, wshift, Cin, Cout, act, poll);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  static const bool wres_off = getenv("LADDER_UP2FUSE_NO_WRES") != nullptr;
  const bool wres = Cin <= UF_WRES_CIN && !wres_off;                                   // the weight slab fits LDS beside the x stages and the plane row
  float* pp = proj_out != nullptr ? (float*)ws : nullptr;
  const float* pwp = proj_out != nullptr ? proj_w : nullptr;
  const int pco = proj_out != nullptr ? proj_cout : 0;
#define UF_LAUNCH(P, R) hipLaunchKernelGGL((up2proj_fused_fwd_kernel<P, R>), dim3(grid), dim3(UF_THREADS), 0, stream, x, wcatT, bias, y, pwp, pp, pco, N, H, W, wshift, Cin, Cout, act, dbg)
  if (proj_out != nullptr) {
    if (wres) UF_LAUNCH(true, true); else UF_LAUNCH(true, false);
    const long P = (long)N * 4 * H * W;
    if ((P & 3) == 0 && ladder_aligned16(ws) && ladder_aligned16(proj_out)) {
      const unsigned rb = (unsigned)((P / 4 + 255) / 256);
#define UF_REDUCE4(C_) hipLaunchKernelGGL(up2proj_proj_reduce4_kernel<C_>, dim3(rb), dim3(256), 0, stream, (const float*)ws, proj_b, proj_out, P, Cout / UF_CS)
      switch (proj_cout) { case 1: UF_REDUCE4(1); break; case 2: UF_REDUCE4(2); break; case 3: UF_REDUCE4(3); break; default: UF_REDUCE4(4); }
#undef UF_REDUCE4
    } else {
      hipLaunchKernelGGL(up2proj_proj_reduce_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, (const float*)ws, proj_b, proj_out, P, proj_cout,
                         Cout / UF_CS);
    }
  } else {
    if (wres) UF_LAUNCH(false, true); else UF_LAUNCH(false, false);
  }
#undef UF_LAUNCH
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

}  // extern "C"
