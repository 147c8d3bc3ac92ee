// HBM-bound passes of the LaDDer path for gfx950: batch-norm (train mode, two-phase so the host can
// all-reduce the 2C statistics), instance-norm + style modulation + leaky-ReLU, TF1-legacy bilinear
// resize, depth-to-space, symmetric pad, activation backward.  All tensors NHWC fp32: the channel
// axis is contiguous, so a wavefront always touches >=256 contiguous bytes per row.
#include "split16.h"

namespace {

// ----------------------------------------------------------------------------- column statistics
// x viewed as [rows, C].  256 threads = 64 channels x 4 row-lanes; grid = (ceil(C/64), row-blocks).
// MODE 1: (sum x, sum x^2)    MODE 2: BN backward (sum dp, sum dp*xhat)
template <int MODE>
__global__ __launch_bounds__(256) void colstats_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ ws, size_t rows,
                                                       int C, size_t rows_per_blk, int act) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const size_t r0 = (size_t)blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    if (MODE == 1) {
      for (size_t r = r0 + rl; r < r1; r += 4) {
        const float v = a[r * C + c];
        s0 += v;
        s1 += v * v;
      }
    } else {
      const float mu = mean_rstd[c], rs = mean_rstd[C + c], g = gamma[c], be = beta[c];
      for (size_t r = r0 + rl; r < r1; r += 4) {
        const float xh = (b[r * C + c] - mu) * rs;
        const float dp = a[r * C + c] * ladder_act_grad_from_out(g * xh + be, act);
        s0 += dp;
        s1 += dp * xh;
      }
    }
  }
  __shared__ float sm[2][4][64];
  sm[0][rl][cl] = s0;
  sm[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    ws[((size_t)blockIdx.y * 2 + 0) * C + c] = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    ws[((size_t)blockIdx.y * 2 + 1) * C + c] = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void colstats_stage1_v4(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ ws, size_t rows,
                                                          int C, size_t rows_per_blk, int act) {
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;       // 16 float4 channel groups x 16 row lanes
  const int c = blockIdx.x * 64 + cq * 4;
  const size_t r0 = (size_t)blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (c < C) {
    if (MODE == 1) {
      size_t r = r0 + rl;
      for (; r + 7 * 16 < r1; r += 8 * 16) {        // eight loads in flight, accumulated in row order (bit-identical to the plain loop)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(a + (r + u * 16) * C + c);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s0.x += v[u].x; s0.y += v[u].y; s0.z += v[u].z; s0.w += v[u].w;
          s1.x += v[u].x * v[u].x; s1.y += v[u].y * v[u].y; s1.z += v[u].z * v[u].z; s1.w += v[u].w * v[u].w;
        }
      }
      for (; r < r1; r += 16) {
        const float4 v = *reinterpret_cast<const float4*>(a + r * C + c);
        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        s1.x += v.x * v.x; s1.y += v.y * v.y; s1.z += v.z * v.z; s1.w += v.w * v.w;
      }
    } else {
      const float4 mu = *reinterpret_cast<const float4*>(mean_rstd + c), rs = *reinterpret_cast<const float4*>(mean_rstd + C + c);
      const float4 g = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
      auto row = [&](const float4 xv, const float4 dv) {
        float xh, dp;
        xh = (xv.x - mu.x) * rs.x; dp = dv.x * ladder_act_grad_from_out(g.x * xh + be.x, act); s0.x += dp; s1.x += dp * xh;
        xh = (xv.y - mu.y) * rs.y; dp = dv.y * ladder_act_grad_from_out(g.y * xh + be.y, act); s0.y += dp; s1.y += dp * xh;
        xh = (xv.z - mu.z) * rs.z; dp = dv.z * ladder_act_grad_from_out(g.z * xh + be.z, act); s0.z += dp; s1.z += dp * xh;
        xh = (xv.w - mu.w) * rs.w; dp = dv.w * ladder_act_grad_from_out(g.w * xh + be.w, act); s0.w += dp; s1.w += dp * xh;
      };
      size_t r = r0 + rl;
      for (; r + 7 * 16 < r1; r += 8 * 16) {        // the loads of eight rows in flight, accumulated in row order (bit-identical to the plain loop)
        float4 xv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xv[u] = *reinterpret_cast<const float4*>(b + (r + u * 16) * C + c);
          dv[u] = *reinterpret_cast<const float4*>(a + (r + u * 16) * C + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) row(xv[u], dv[u]);
      }
      for (; r < r1; r += 16) row(*reinterpret_cast<const float4*>(b + r * C + c), *reinterpret_cast<const float4*>(a + r * C + c));
    }
  }
  __shared__ float4 sm[2][16][16];
  sm[0][rl][cq] = s0;
  sm[1][rl][cq] = s1;
  __syncthreads();
  if (rl < 2 && c < C) {        // rl 0 -> sums, rl 1 -> second statistic; fixed-order tree over the 16 row lanes
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 v = sm[rl][k][cq];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(ws + ((size_t)blockIdx.y * 2 + rl) * C + c) = t;
  }
}

// ---- batch-norm FORWARD statistics in fp64 (round 5) ------------------------------------------------------------------------------------
// TF's fused batch norm is two-pass (reference codes/models.py:398-460: tf.layers.batch_normalization): the variance is formed about the
// mean.  The single-pass form var = E[x^2] - mean^2 from fp32 sums loses eps_fp32 x (1 + mean^2 / var) of relative accuracy -- invisible at
// initialisation (|mean| ~ std), 1.5e-4 on a channel whose mean sits 50 standard deviations off zero.  Rounds 1-4 accumulated and STORED the
// two sums in fp32 (and all-reduced those: C2).  Now: every element enters an fp64 accumulator (x^2 is exact in fp64), partials and the record
// are fp64, C2 all-reduces 2C doubles, and E[x^2] - mean^2 in fp64 carries 1e-16 x (1 + mean^2 / var).  These passes are HBM-bound (3 fp64
// operations per element against 78 TFLOP/s of fp64 vector rate): their duration does not change.
// RECORD layout (the `sums` argument of every ladder_bn_* entry point): 2C doubles = sum x | sum x^2, i.e. the first 4C floats of the buffer;
// the "minmax" form appends min x | max x as 2C floats (6C floats in all).
__global__ __launch_bounds__(256) void colstats64_stage1(const float* __restrict__ a, double* __restrict__ ws, size_t rows, int C,
                                                         size_t rows_per_blk) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const size_t r0 = (size_t)blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  double s0 = 0.0, s1 = 0.0;
  if (c < C)
    for (size_t r = r0 + rl; r < r1; r += 4) {
      const double v = (double)a[r * C + c];
      s0 += v;
      s1 = fma(v, v, s1);
    }
  __shared__ double sm[2][4][64];
  sm[0][rl][cl] = s0;
  sm[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    ws[((size_t)blockIdx.y * 2 + 0) * C + c] = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    ws[((size_t)blockIdx.y * 2 + 1) * C + c] = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
  }
}

// 16 float4 channel groups x 16 row lanes; MINMAX: also the per-channel extremes.  Partials: ws [nblk][2][C] doubles, then (MINMAX) [nblk][2][C]
// floats behind all of them.
template <bool MINMAX>
__global__ __launch_bounds__(256) void colstats64_stage1_v4(const float* __restrict__ a, double* __restrict__ ws, float* __restrict__ wsmm,
                                                            size_t rows, int C, size_t rows_per_blk) {
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const size_t r0 = (size_t)blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  if (c < C) {
    auto row = [&](const float4 v) {
      const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0[j] += d[j];
        s1[j] = fma(d[j], d[j], s1[j]);
      }
      if (MINMAX) {
        mn.x = fminf(mn.x, v.x); mn.y = fminf(mn.y, v.y); mn.z = fminf(mn.z, v.z); mn.w = fminf(mn.w, v.w);
        mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
      }
    };
    size_t r = r0 + rl;
    for (; r + 7 * 16 < r1; r += 8 * 16) {          // eight loads in flight, accumulated in row order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(a + (r + u * 16) * C + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) row(v[u]);
    }
    for (; r < r1; r += 16) row(*reinterpret_cast<const float4*>(a + r * C + c));
  }
  __shared__ double sm[2][16][16][4];
  __shared__ float4 smm[2][16][16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[0][rl][cq][j] = s0[j];
    sm[1][rl][cq][j] = s1[j];
  }
  if (MINMAX) {
    smm[0][rl][cq] = mn;
    smm[1][rl][cq] = mx;
  }
  __syncthreads();
  if (rl < 2 && c < C) {          // rl = which sum; fixed-order combination of the 16 row lanes
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double t = sm[rl][0][cq][j];
#pragma unroll
      for (int k = 1; k < 16; ++k) t += sm[rl][k][cq][j];
      ws[((size_t)blockIdx.y * 2 + rl) * C + c + j] = t;
    }
  } else if (MINMAX && rl < 4 && c < C) {
    const int w = rl - 2;
    float4 t = smm[w][0][cq];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = smm[w][k][cq];
      if (w == 0) { t.x = fminf(t.x, v.x); t.y = fminf(t.y, v.y); t.z = fminf(t.z, v.z); t.w = fminf(t.w, v.w); }
      else { t.x = fmaxf(t.x, v.x); t.y = fmaxf(t.y, v.y); t.z = fmaxf(t.z, v.z); t.w = fmaxf(t.w, v.w); }
    }
    *reinterpret_cast<float4*>(wsmm + ((size_t)blockIdx.y * 2 + w) * C + c) = t;
  }
}

// second stage over fp64 partials [nblk][2][C] (+ float extremes [nblk][2][C]): the record (2C doubles, + 2C floats); fixed order
__global__ __launch_bounds__(1024) void colstats64_stage2(const double* __restrict__ ws, const float* __restrict__ wsmm, float* __restrict__ rec,
                                                          int nblk, int C) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;       // rl 0 .. 15
  const int i = blockIdx.x * 64 + cl;                             // over 2C sums, then (wsmm != NULL) 2C extremes
  const int total = wsmm != nullptr ? 4 * C : 2 * C;
  const bool is_sum = i < 2 * C;
  const int which = is_sum ? i / C : (i - 2 * C) / C, c = is_sum ? i - which * C : i - 2 * C - which * C;
  double s = is_sum ? 0.0 : (which == 0 ? (double)INFINITY : -(double)INFINITY);
  if (i < total) {
    for (int b = rl; b < nblk; b += 16) {
      if (is_sum) s += ws[((size_t)b * 2 + which) * C + c];
      else {
        const double v = (double)wsmm[((size_t)b * 2 + which) * C + c];
        s = which == 0 ? fmin(s, v) : fmax(s, v);
      }
    }
  }
  __shared__ double sm[16][64];
  sm[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && i < total) {
    double t = sm[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) t = is_sum ? t + sm[k][cl] : (which == 0 ? fmin(t, sm[k][cl]) : fmax(t, sm[k][cl]));
    if (is_sum) reinterpret_cast<double*>(rec)[i] = t;
    else rec[4 * C + (i - 2 * C)] = (float)t;
  }
}

// second stage for the fp32 [nblk][4][C] partials a convolution epilogue emits (sum, sum of squares, min, max per tile): sums in fp64 (fixed
// order), extremes exactly; out = the fp64 record (2C doubles sum | sum of squares, then min | max as 2C floats).
// 16 columns x 64 row lanes per workgroup, four independent accumulators per lane (round 4: with 16 row lanes and one dependent fp64 chain per
// lane the 4 096 partial rows of the image-side conv took 47 us -- a third of the conv itself)
__global__ __launch_bounds__(1024) void colstats_minmax_stage2(const float* __restrict__ ws, float* __restrict__ out, int nblk, int C) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;          // rl 0 .. 63
  const int i = blockIdx.x * 16 + cl;   // over 4C
  const int which = i < 4 * C ? i / C : 0, c = i < 4 * C ? i - which * C : 0;
  const double init = which == 2 ? (double)INFINITY : (which == 3 ? -(double)INFINITY : 0.0);
  double s4[4] = {init, init, init, init};
  if (i < 4 * C) {
    int b = rl;
    for (; b + 3 * 64 < nblk; b += 4 * 64) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (double)ws[((size_t)(b + u * 64) * 4 + which) * C + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] = which < 2 ? s4[u] + v[u] : (which == 2 ? fmin(s4[u], v[u]) : fmax(s4[u], v[u]));
    }
    for (int u = 0; b < nblk; b += 64, ++u) {
      const double v = (double)ws[((size_t)b * 4 + which) * C + c];
      s4[u] = which < 2 ? s4[u] + v : (which == 2 ? fmin(s4[u], v) : fmax(s4[u], v));
    }
  }
  const double s = which < 2 ? (s4[0] + s4[1]) + (s4[2] + s4[3]) : (which == 2 ? fmin(fmin(s4[0], s4[1]), fmin(s4[2], s4[3])) : fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3])));
  __shared__ double sm[64][17];
  sm[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && i < 4 * C) {
    double t = sm[0][cl];
#pragma unroll 8
    for (int k = 1; k < 64; ++k) t = which < 2 ? t + sm[k][cl] : (which == 2 ? fmin(t, sm[k][cl]) : fmax(t, sm[k][cl]));
    // the fp64 record (see colstats64_stage1): 2C doubles = sum | sum of squares, then min | max as floats
    if (which < 2) reinterpret_cast<double*>(out)[i] = t;
    else out[4 * C + (i - 2 * C)] = (float)t;
  }
}

// stage 2: one workgroup per 64 (which,channel) columns; 16 row-lanes stride over the stage-1 partials, combined in a
// fixed order (fp64) -> deterministic and ~nblk/16 dependent adds instead of nblk (round 4: 4 row-lanes -> 16: 11 -> 6 us, 16 calls per iteration).
__global__ __launch_bounds__(1024) void colstats_stage2(const float* __restrict__ ws, float* __restrict__ out, int nblk, int C) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;       // rl 0 .. 15
  const int i = blockIdx.x * 64 + cl;   // over 2C
  double s = 0.0;
  if (i < 2 * C) {
    const int which = i / C, c = i - which * C;
    for (int b = rl; b < nblk; b += 16) s += (double)ws[((size_t)b * 2 + which) * C + c];
  }
  __shared__ double sm[16][64];
  sm[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && i < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][cl];
    out[i] = (float)t;
  }
}

// stage 2 for MANY partials (the per-tile column sums a convolution epilogue emits: thousands of blocks): 16 columns x 16 row lanes per
// workgroup, every lane sums each 16th partial in fp64, fixed-order LDS tree.
__global__ __launch_bounds__(256) void colstats_stage2_wide(const float* __restrict__ ws, float* __restrict__ out, int nblk, int C) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;   // over 2C
  double s = 0.0;
  if (i < 2 * C) {
    const int which = i / C, c = i - which * C;
    for (int b = rl; b < nblk; b += 16) s += (double)ws[((size_t)b * 2 + which) * C + c];
  }
  __shared__ double sm[16][17];
  sm[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && i < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][cl];
    reinterpret_cast<double*>(out)[i] = t;          // the fp64 record (forward statistics only: ladder_bn_stats_from_partials)
  }
}

size_t stats_nblk(size_t rows) {
  // (1024 rows per block.  Smaller blocks on the small encoder maps are 2-3x faster -- a few workgroups walk 64 dependent loads each there -- but
  // change the summation partition, and with it which side of zero the numerically-zero gradients of iteration 0 fall on: the golden two-iteration
  // fixture then deviates by 1e-4 ... 3e-3 behind Adam's first step.  The loads are batched instead: same order, same bits.)
  size_t nblk = (rows + 1023) / 1024;
  if (nblk > 512) nblk = 512;
  if (nblk < 1) nblk = 1;
  return nblk;
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, double count, float eps, float* __restrict__ mean_rstd, int C,
                                   float* __restrict__ clear_rec) {
  amax_clear_by_block0(clear_rec);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double* s64 = reinterpret_cast<const double*>(sums);      // the fp64 record: sum | sum of squares
  const double mean = s64[c] / count;
  double var = s64[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  mean_rstd[c] = (float)mean;
  mean_rstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, size_t n, int C, int act, float* __restrict__ yamax) {
  // C % 4 == 0 path: float4 per thread, grid-stride; the 4 channels' coefficients are loaded once when the stride is a multiple of C
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool fixed = ((stride * 4) % (size_t)C) == 0;
  int c = (int)((i0 * 4) % C);
  float4 mu = *reinterpret_cast<const float4*>(mean_rstd + c), rs = *reinterpret_cast<const float4*>(mean_rstd + C + c);
  float4 g = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  float ymax = 0.f;
  for (size_t i = i0; i < n / 4; i += stride) {
    if (!fixed) {
      c = (int)((i * 4) % C);
      mu = *reinterpret_cast<const float4*>(mean_rstd + c); rs = *reinterpret_cast<const float4*>(mean_rstd + C + c);
      g = *reinterpret_cast<const float4*>(gamma + c); be = *reinterpret_cast<const float4*>(beta + c);
    }
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = ladder_act_fn(g.x * ((v.x - mu.x) * rs.x) + be.x, act);
    o.y = ladder_act_fn(g.y * ((v.y - mu.y) * rs.y) + be.y, act);
    o.z = ladder_act_fn(g.z * ((v.z - mu.z) * rs.z) + be.z, act);
    o.w = ladder_act_fn(g.w * ((v.w - mu.w) * rs.w) + be.w, act);
    reinterpret_cast<float4*>(y)[i] = o;
    ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (yamax != nullptr) amax_commit_block(ymax, yamax);     // max|y| for the split contraction that consumes y
}
// ... with the finalisation INSIDE (strict fp32, no absmax record): a thread derives the mean / 1 / sd of its four channels from the fp64 record itself -- the
// expressions of bn_finalize_kernel, so the values are the same bits -- and workgroup 0 writes mean_rstd for the backward pass.  One launch less per batch
// norm (a 5 us kernel + its gap on the serial chain, twelve times an iteration).  Host side: only when the stride is a multiple of C.
__global__ __launch_bounds__(256) void bn_apply_fin_kernel(const float* __restrict__ x, const float* __restrict__ sums, const double count, const float eps,
                                                           float* __restrict__ mean_rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ y, size_t n, int C, int act) {
  const double* s64 = reinterpret_cast<const double*>(sums);      // the fp64 record: sum | sum of squares
  auto coef = [&](const int c, float& mu, float& rs) __attribute__((always_inline)) {
    const double mean = s64[c] / count;
    double var = s64[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mu = (float)mean;
    rs = (float)(1.0 / sqrt(var + (double)eps));
  };
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += 256) {
      float mu, rs;
      coef(c, mu, rs);
      mean_rstd[c] = mu;
      mean_rstd[C + c] = rs;
    }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)((i0 * 4) % C);
  float mu[4], rs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) coef(c + j, mu[j], rs[j]);
  const float4 g = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  for (size_t i = i0; i < n / 4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = ladder_act_fn(g.x * ((v.x - mu[0]) * rs[0]) + be.x, act);
    o.y = ladder_act_fn(g.y * ((v.y - mu[1]) * rs[1]) + be.y, act);
    o.z = ladder_act_fn(g.z * ((v.z - mu[2]) * rs[2]) + be.z, act);
    o.w = ladder_act_fn(g.w * ((v.w - mu[3]) * rs[3]) + be.w, act);
    reinterpret_cast<float4*>(y)[i] = o;
  }
}
__global__ void bn_apply_scalar_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ y, size_t n, int C, int act) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    y[i] = ladder_act_fn(gamma[c] * ((x[i] - mean_rstd[c]) * mean_rstd[C + c]) + beta[c], act);
  }
}

// ---- batch-norm apply that emits the fp16 PLANES of y (the operand images of the split gather kernels, ladder_presplit layout) ----------
// max|y| must be known before y is split (the f16x3 scale); per channel y = act(gamma * xhat + beta) is monotone in x, so it is attained
// at the channel's minimum or maximum: the finalize kernel evaluates those 2C values with the apply kernel's own expression and writes
// the record.  One workgroup (C <= a few thousand channels).
__global__ __launch_bounds__(256) void bn_finalize_minmax_kernel(const float* __restrict__ sums4, double count, float eps,
                                                                 float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int C, int act, float* __restrict__ rec) {
  float bmax = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double* s64 = reinterpret_cast<const double*>(sums4);   // the fp64 record: sum | sum of squares (2C doubles), then min | max (floats)
    const double mean = s64[c] / count;
    double var = s64[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
    mean_rstd[c] = mu;
    mean_rstd[C + c] = rs;
    const float g = gamma[c], be = beta[c];
    const float ylo = ladder_act_fn(g * ((sums4[4 * C + c] - mu) * rs) + be, act), yhi = ladder_act_fn(g * ((sums4[5 * C + c] - mu) * rs) + be, act);
    bmax = fmaxf(bmax, fmaxf(fabsf(ylo), fabsf(yhi)));
  }
  __shared__ float red[4];
  bmax = wave_max(bmax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bmax;
  __syncthreads();
  for (int i = threadIdx.x; i < AMAX_SLOTS * AMAX_STRIDE; i += blockDim.x) rec[i] = 0.f;
  __syncthreads();
  if (threadIdx.x == 0) rec[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void bn_apply_planes_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, uint16_t* __restrict__ planes, size_t n, int C, int act,
                                                              const float* __restrict__ rec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool fixed = ((stride * 4) % (size_t)C) == 0;
  const float cs = scale_from_absmax(amax_load(rec));
  int c = (int)((i0 * 4) % C);
  float4 mu = *reinterpret_cast<const float4*>(mean_rstd + c), rs = *reinterpret_cast<const float4*>(mean_rstd + C + c);
  float4 g = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  if (i0 == 0) {
    *reinterpret_cast<uint4*>(planes + 2 * n) = make_uint4(0u, 0u, 0u, 0u);      // the zero pad behind the last plane
    *reinterpret_cast<uint4*>(planes + 2 * n + 8) = make_uint4(0u, 0u, 0u, 0u);  // header (ladder_presplit): ONE scale for the tensor
  }
  for (size_t i = i0; i < n / 4; i += stride) {
    if (!fixed) {
      c = (int)((i * 4) % C);
      mu = *reinterpret_cast<const float4*>(mean_rstd + c); rs = *reinterpret_cast<const float4*>(mean_rstd + C + c);
      g = *reinterpret_cast<const float4*>(gamma + c); be = *reinterpret_cast<const float4*>(beta + c);
    }
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = ladder_act_fn(g.x * ((v.x - mu.x) * rs.x) + be.x, act);
    o.y = ladder_act_fn(g.y * ((v.y - mu.y) * rs.y) + be.y, act);
    o.z = ladder_act_fn(g.z * ((v.z - mu.z) * rs.z) + be.z, act);
    o.w = ladder_act_fn(g.w * ((v.w - mu.w) * rs.w) + be.w, act);
    if (y != nullptr) reinterpret_cast<float4*>(y)[i] = o;
    uint2 pl[2];
    split4<2, true>(make_float4(o.x * cs, o.y * cs, o.z * cs, o.w * cs), pl);
    reinterpret_cast<uint2*>(planes)[i] = pl[0];
    reinterpret_cast<uint2*>(planes + n)[i] = pl[1];
  }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ dsums, float inv_count,
                                    float* __restrict__ dx, size_t n, int C, int act) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    const float mu = mean_rstd[c], rs = mean_rstd[C + c], g = gamma[c];
    const float xh = (x[i] - mu) * rs;
    const float dp = dy[i] * ladder_act_grad_from_out(g * xh + beta[c], act);
    dx[i] = g * rs * (dp - dsums[c] * inv_count - xh * dsums[C + c] * inv_count);
  }
}
// float4 per thread, grid-stride.  When the stride is a multiple of C (always for power-of-two channel counts) a thread meets the same 4
// channels in every iteration: their 6 coefficients are loaded once -- per-iteration parameter gathers (24 dword loads beside the two
// 16-byte data loads) held this kernel to ~2 TB/s.
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ dsums,
                                                              float inv_count, float* __restrict__ dx, size_t n, int C, int act,
                                                              float* __restrict__ dxamax, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  if (dgamma != nullptr && blockIdx.x == 0)                       // (the parameter gradients ARE the reduced sums: no launch of their own)
    for (int c = threadIdx.x; c < C; c += 256) {
      dbeta[c] = dsums[c];
      dgamma[c] = dsums[C + c];
    }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool fixed = ((stride * 4) % (size_t)C) == 0;
  float dmax = 0.f;
  float mu[4], rs[4], g[4], be[4], s1[4], s2[4];
  auto coeffs = [&](int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = mean_rstd[c + j]; rs[j] = mean_rstd[C + c + j]; g[j] = gamma[c + j]; be[j] = beta[c + j];
      s1[j] = dsums[c + j] * inv_count; s2[j] = dsums[C + c + j] * inv_count;
    }
  };
  if (fixed) coeffs((int)((i0 * 4) % C));
  for (size_t i = i0; i < n / 4; i += stride) {
    if (!fixed) coeffs((int)((i * 4) % C));
    const float4 xv = reinterpret_cast<const float4*>(x)[i], dv = reinterpret_cast<const float4*>(dy)[i];
    float o[4];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mu[j]) * rs[j];
      const float dp = ds[j] * ladder_act_grad_from_out(g[j] * xh + be[j], act);
      o[j] = g[j] * rs[j] * (dp - s1[j] - xh * s2[j]);
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
    dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (dxamax != nullptr) amax_commit_block(dmax, dxamax);
}
__global__ void bn_param_grad_kernel(const float* __restrict__ dsums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                     float* __restrict__ clear_rec) {
  amax_clear_by_block0(clear_rec);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = dsums[c];
  dgamma[c] = dsums[C + c];
}

// ----------------------------------------------------------------------------- instance norm + style + act
// one workgroup per (sample n, 64-channel group); 4 row-lanes stride over HW.
__device__ __forceinline__ float block_rowlane_sum(float v, float (*sm)[64], int rl, int cl) {
  __syncthreads();
  sm[rl][cl] = v;
  __syncthreads();
  return (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
}

__global__ __launch_bounds__(256) void in_style_fwd_kernel(const float* __restrict__ x, const float* __restrict__ style,
                                                           float* __restrict__ y, float* __restrict__ mean_rstd, int HW, int C,
                                                           float eps, int act) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.y, c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  const float* xp = x + (size_t)n * HW * C;
  float* yp = y + (size_t)n * HW * C;
  __shared__ float sm[4][64];
  float s = 0.f;
  if (ok)
    for (int r = rl; r < HW; r += 4) s += xp[(size_t)r * C + c];
  const float mean = block_rowlane_sum(s, sm, rl, cl) / (float)HW;
  s = 0.f;
  if (ok)
    for (int r = rl; r < HW; r += 4) {
      const float dlt = xp[(size_t)r * C + c] - mean;
      s += dlt * dlt;
    }
  const float var = block_rowlane_sum(s, sm, rl, cl) / (float)HW;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!ok) return;
  const float s0 = style[(size_t)n * 2 * C + c] + 1.f, s1 = style[(size_t)n * 2 * C + C + c];
  if (rl == 0) {
    mean_rstd[(size_t)n * 2 * C + c] = mean;
    mean_rstd[(size_t)n * 2 * C + C + c] = rstd;
  }
  for (int r = rl; r < HW; r += 4) {
    const size_t i = (size_t)r * C + c;
    yp[i] = ladder_act_fn((xp[i] - mean) * rstd * s0 + s1, act);
  }
}

__global__ __launch_bounds__(256) void in_style_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ style, const float* __restrict__ mean_rstd,
                                                           float* __restrict__ dx, float* __restrict__ dstyle, int HW, int C, int act) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.y, c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  const size_t base = (size_t)n * HW * C;
  __shared__ float sm[4][64];
  float mean = 0.f, rstd = 0.f, s0 = 0.f, s1 = 0.f;
  if (ok) {
    mean = mean_rstd[(size_t)n * 2 * C + c];
    rstd = mean_rstd[(size_t)n * 2 * C + C + c];
    s0 = style[(size_t)n * 2 * C + c] + 1.f;
    s1 = style[(size_t)n * 2 * C + C + c];
  }
  float a0 = 0.f, a1 = 0.f;
  if (ok)
    for (int r = rl; r < HW; r += 4) {
      const size_t i = base + (size_t)r * C + c;
      const float xh = (x[i] - mean) * rstd;
      const float dp = dy[i] * ladder_act_grad_from_out(xh * s0 + s1, act);
      a0 += dp * xh;
      a1 += dp;
    }
  const float ds0 = block_rowlane_sum(a0, sm, rl, cl);
  const float ds1 = block_rowlane_sum(a1, sm, rl, cl);
  if (!ok) return;
  if (rl == 0) {
    dstyle[(size_t)n * 2 * C + c] = ds0;
    dstyle[(size_t)n * 2 * C + C + c] = ds1;
  }
  const float inv = 1.f / (float)HW;
  for (int r = rl; r < HW; r += 4) {
    const size_t i = base + (size_t)r * C + c;
    const float xh = (x[i] - mean) * rstd;
    const float dp = dy[i] * ladder_act_grad_from_out(xh * s0 + s1, act);
    dx[i] = rstd * s0 * (dp - ds1 * inv - xh * ds0 * inv);
  }
}

// Vectorised instance-norm kernels (C % 4 == 0).  256 threads = 16 channel quads (64 channels) x 16 row lanes.
// The H*W axis of one (sample, 64-channel) slab is split over `split` workgroups so that large maps expose enough
// parallelism to stream at HBM rate; moments are accumulated about a per-channel PIVOT, which keeps the single-pass variance
// var = E[(x - p)^2] - (E[x - p])^2 exact enough for eps = 1e-6 -- as long as the pivot lies within a few standard deviations of the mean
// (the relative error of the variance is eps_fp32 x (1 + (p - mean)^2 / var)).  Round 4: the pivot is the average of 16 pixels spread
// evenly over the instance (in_pivot).  Rounds 1-3 took the instance's FIRST pixel: on the small decoder maps (8x8, 16x16) that is a
// zero-padded corner of the producing convolution, which can sit tens of standard deviations off a low-variance channel's mean -- found
// by the full-resolution data-parallel test against the float64 oracle (batch 16: 1-3 % error on the gradients behind such a channel,
// i.e. 40x the fp32 CPU restatement's deviation, while batch 8 happened to pass).  Partials are combined in a fixed order.
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
constexpr int IN_PIVOT_SAMPLES = 16;
// average of pixels floor(k HW / 16), k = 0..15 (min(16, HW) of them) of channel(s) c of one instance: the same fp32 operations in the
// same order for a scalar and for a float4 caller, so every workgroup of an instance and the finalize kernel agree bit for bit
__device__ __forceinline__ float in_pivot(const float* __restrict__ xp, int HW, int C, int c) {
  const int cnt = HW < IN_PIVOT_SAMPLES ? HW : IN_PIVOT_SAMPLES;
  float s = 0.f;
  for (int k = 0; k < cnt; ++k) s += xp[(size_t)((long)k * HW / cnt) * C + c];
  return s * (1.f / (float)cnt);
}
__device__ __forceinline__ float4 in_pivot4(const float* __restrict__ xp, int HW, int C, int c) {
  const int cnt = HW < IN_PIVOT_SAMPLES ? HW : IN_PIVOT_SAMPLES;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < cnt; ++k) s = f4add(s, *reinterpret_cast<const float4*>(xp + (size_t)((long)k * HW / cnt) * C + c));
  const float r = 1.f / (float)cnt;
  return make_float4(s.x * r, s.y * r, s.z * r, s.w * r);
}

__global__ __launch_bounds__(256) void in_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C, int split) {
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.y, c = blockIdx.x * 64 + cq * 4, sp = blockIdx.z;
  const float* xp = x + (size_t)n * HW * C;
  const int per = (HW + split - 1) / split, r0 = sp * per, r1 = min(HW, r0 + per);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (c < C) {
    const float4 pv = in_pivot4(xp, HW, C, c);
    for (int r = r0 + rl; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(xp + (size_t)r * C + c);
      const float dx = v.x - pv.x, dy = v.y - pv.y, dz = v.z - pv.z, dw = v.w - pv.w;
      s0.x += dx; s0.y += dy; s0.z += dz; s0.w += dw;
      s1.x += dx * dx; s1.y += dy * dy; s1.z += dz * dz; s1.w += dw * dw;
    }
  }
  __shared__ float4 sm[2][16][16];
  sm[0][rl][cq] = s0;
  sm[1][rl][cq] = s1;
  __syncthreads();
  if (rl < 2 && c < C) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) t = f4add(t, sm[rl][k][cq]);
    *reinterpret_cast<float4*>(part + (((size_t)n * split + sp) * 2 + rl) * C + c) = t;
  }
}

__global__ void in_finalize_kernel(const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean_rstd,
                                   int N, int HW, int C, int split, float eps, float* __restrict__ clear_rec) {
  amax_clear_by_block0(clear_rec);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double s0 = 0.0, s1 = 0.0;
  for (int sp = 0; sp < split; ++sp) {
    s0 += (double)part[(((size_t)n * split + sp) * 2 + 0) * C + c];
    s1 += (double)part[(((size_t)n * split + sp) * 2 + 1) * C + c];
  }
  const double pv = (double)in_pivot(x + (size_t)n * HW * C, HW, C, c);
  const double md = s0 / HW;
  double var = s1 / HW - md * md;
  if (var < 0.0) var = 0.0;
  mean_rstd[(size_t)n * 2 * C + c] = (float)(pv + md);
  mean_rstd[(size_t)n * 2 * C + C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ x, const float* __restrict__ style,
                                                       const float* __restrict__ mean_rstd, float* __restrict__ y, int HW, int C,
                                                       int act, int split, float* __restrict__ yamax) {
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.y, sp = blockIdx.z;
  int c = blockIdx.x * 64 + cq * 4;
  const bool live = c < C;
  if (!live) {
    if (yamax == nullptr) return;
    c = 0;                                       // (keeps the loads in range; the thread stores nothing and contributes max 0)
  }
  float ymax = 0.f;
  const size_t base = (size_t)n * HW * C;
  const int per = (HW + split - 1) / split, r0 = sp * per, r1 = live ? min(HW, r0 + per) : r0;
  const float4 mu = *reinterpret_cast<const float4*>(mean_rstd + (size_t)n * 2 * C + c);
  const float4 rs = *reinterpret_cast<const float4*>(mean_rstd + (size_t)n * 2 * C + C + c);
  float4 s0 = *reinterpret_cast<const float4*>(style + (size_t)n * 2 * C + c);
  const float4 s1 = *reinterpret_cast<const float4*>(style + (size_t)n * 2 * C + C + c);
  s0.x += 1.f; s0.y += 1.f; s0.z += 1.f; s0.w += 1.f;
  for (int r = r0 + rl; r < r1; r += 16) {
    const size_t i = base + (size_t)r * C + c;
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    float4 o;
    o.x = ladder_act_fn((v.x - mu.x) * rs.x * s0.x + s1.x, act);
    o.y = ladder_act_fn((v.y - mu.y) * rs.y * s0.y + s1.y, act);
    o.z = ladder_act_fn((v.z - mu.z) * rs.z * s0.z + s1.z, act);
    o.w = ladder_act_fn((v.w - mu.w) * rs.w * s0.w + s1.w, act);
    *reinterpret_cast<float4*>(y + i) = o;
    ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (yamax != nullptr) amax_commit_block_sample(ymax, yamax, n);      // (blockIdx.y = sample: a per-sample record)
}

// backward statistics: part[.,0] = sum dp*xhat, part[.,1] = sum dp
__global__ __launch_bounds__(256) void in_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ style, const float* __restrict__ mean_rstd,
                                                           float* __restrict__ part, int HW, int C, int act, int split) {
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.y, c = blockIdx.x * 64 + cq * 4, sp = blockIdx.z;
  const size_t base = (size_t)n * HW * C;
  const int per = (HW + split - 1) / split, r0 = sp * per, r1 = min(HW, r0 + per);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4*>(mean_rstd + (size_t)n * 2 * C + c);
    const float4 rs = *reinterpret_cast<const float4*>(mean_rstd + (size_t)n * 2 * C + C + c);
    float4 s0 = *reinterpret_cast<const float4*>(style + (size_t)n * 2 * C + c);
    const float4 s1 = *reinterpret_cast<const float4*>(style + (size_t)n * 2 * C + C + c);
    s0.x += 1.f; s0.y += 1.f; s0.z += 1.f; s0.w += 1.f;
    for (int r = r0 + rl; r < r1; r += 16) {
      const size_t i = base + (size_t)r * C + c;
      const float4 xv = *reinterpret_cast<const float4*>(x + i), dv = *reinterpret_cast<const float4*>(dy + i);
      float xh, dp;
      xh = (xv.x - mu.x) * rs.x; dp = dv.x * ladder_act_grad_from_out(xh * s0.x + s1.x, act); a0.x += dp * xh; a1.x += dp;
      xh = (xv.y - mu.y) * rs.y; dp = dv.y * ladder_act_grad_from_out(xh * s0.y + s1.y, act); a0.y += dp * xh; a1.y += dp;
      xh = (xv.z - mu.z) * rs.z; dp = dv.z * ladder_act_grad_from_out(xh * s0.z + s1.z, act); a0.z += dp * xh; a1.z += dp;
      xh = (xv.w - mu.w) * rs.w; dp = dv.w * ladder_act_grad_from_out(xh * s0.w + s1.w, act); a0.w += dp * xh; a1.w += dp;
    }
  }
  __shared__ float4 sm[2][16][16];
  sm[0][rl][cq] = a0;
  sm[1][rl][cq] = a1;
  __syncthreads();
  if (rl < 2 && c < C) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) t = f4add(t, sm[rl][k][cq]);
    *reinterpret_cast<float4*>(part + (((size_t)n * split + sp) * 2 + rl) * C + c) = t;
  }
}
__global__ void in_bwd_finalize_kernel(const float* __restrict__ part, float* __restrict__ dstyle, int N, int C, int split,
                                       float* __restrict__ clear_rec) {
  amax_clear_by_block0(clear_rec);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  double s0 = 0.0, s1 = 0.0;
  for (int sp = 0; sp < split; ++sp) {
    s0 += (double)part[(((size_t)n * split + sp) * 2 + 0) * C + c];
    s1 += (double)part[(((size_t)n * split + sp) * 2 + 1) * C + c];
  }
  dstyle[(size_t)n * 2 * C + c] = (float)s0;
  dstyle[(size_t)n * 2 * C + C + c] = (float)s1;
}
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ style, const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ dstyle, float* __restrict__ dx, int HW, int C,
                                                           int act, int split, float* __restrict__ dxamax) {
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.y, sp = blockIdx.z;
  int c = blockIdx.x * 64 + cq * 4;
  const bool live = c < C;
  if (!live) {
    if (dxamax == nullptr) return;
    c = 0;
  }
  float omax = 0.f;
  const size_t base = (size_t)n * HW * C;
  const int per = (HW + split - 1) / split, r0 = sp * per, r1 = live ? min(HW, r0 + per) : r0;
  const float inv = 1.f / (float)HW;
  float mu[4], rs[4], s0[4], s1[4], d0[4], d1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean_rstd[(size_t)n * 2 * C + c + j];
    rs[j] = mean_rstd[(size_t)n * 2 * C + C + c + j];
    s0[j] = style[(size_t)n * 2 * C + c + j] + 1.f;
    s1[j] = style[(size_t)n * 2 * C + C + c + j];
    d0[j] = dstyle[(size_t)n * 2 * C + c + j] * inv;
    d1[j] = dstyle[(size_t)n * 2 * C + C + c + j] * inv;
  }
  for (int r = r0 + rl; r < r1; r += 16) {
    const size_t i = base + (size_t)r * C + c;
    const float4 xv = *reinterpret_cast<const float4*>(x + i), dv = *reinterpret_cast<const float4*>(dy + i);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mu[j]) * rs[j];
      const float dp = ds[j] * ladder_act_grad_from_out(xh * s0[j] + s1[j], act);
      o[j] = rs[j] * s0[j] * (dp - d1[j] - xh * d0[j]);
    }
    *reinterpret_cast<float4*>(dx + i) = make_float4(o[0], o[1], o[2], o[3]);
    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (dxamax != nullptr) amax_commit_block_sample(omax, dxamax, n);
}

inline int in_split(int N, int HW, int C) {
  const long groups = (long)N * ((C + 63) / 64);
  int sp = (int)((2048 + groups - 1) / groups);     // aim for ~8 workgroups per CU
  const int max_sp = (HW + 63) / 64;                // at least 64 pixels (4 per row lane) per workgroup
  if (sp > max_sp) sp = max_sp;
  if (sp < 1) sp = 1;
  return sp;
}

// ----------------------------------------------------------------------------- legacy bilinear resize (integer factors)
// Source-centric mapping: `bpr` consecutive blocks cover one source row n*H + iy, walking (ix, channel vector).  A thread
// loads the 4 corner vectors of its source pixel once and writes the fy x fx output pixels whose lower corner it is, so the
// forward reads every source vector ~4x from L1/L2 and once from HBM, and all index arithmetic is 32-bit.  V = 4 uses 16-byte
// accesses (C % 4 == 0 and 16-byte aligned bases).
template <int V> struct ResizeVec;
template <> struct ResizeVec<4> { using T = float4; };
template <> struct ResizeVec<1> { using T = float; };
__device__ __forceinline__ float4 rz_lerp(const float4& a, const float4& b, float t) {
  return make_float4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
__device__ __forceinline__ float rz_lerp(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ void rz_fma(float4& acc, float w, const float4& q) {
  acc.x += w * q.x; acc.y += w * q.y; acc.z += w * q.z; acc.w += w * q.w;
}
__device__ __forceinline__ void rz_fma(float& acc, float w, float q) { acc += w * q; }
__device__ __forceinline__ void rz_zero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void rz_zero(float& a) { a = 0.f; }
__device__ __forceinline__ void rz_gate(float4& a, const float4& y, int act) {
  a.x *= ladder_act_grad_from_out(y.x, act); a.y *= ladder_act_grad_from_out(y.y, act);
  a.z *= ladder_act_grad_from_out(y.z, act); a.w *= ladder_act_grad_from_out(y.w, act);
}
__device__ __forceinline__ void rz_gate(float& a, float y, int act) { a *= ladder_act_grad_from_out(y, act); }

template <int V>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int CV,
                                                         int fy, int fx, int bpr) {
  using T = typename ResizeVec<V>::T;
  const int row = blockIdx.x / bpr;                 // n*H + iy (wave-uniform)
  const int j = (blockIdx.x - row * bpr) * blockDim.x + threadIdx.x;
  if (j >= W * CV) return;
  const int n = row / H, iy = row - n * H;
  const int ix = j / CV, cv = j - ix * CV;
  const int yhi = min(iy + 1, H - 1), xhi = min(ix + 1, W - 1);
  const T* r0 = reinterpret_cast<const T*>(x) + (size_t)row * W * CV;
  const T* r1 = reinterpret_cast<const T*>(x) + ((size_t)n * H + yhi) * W * CV;
  const T tl = r0[ix * CV + cv], tr = r0[xhi * CV + cv], bl = r1[ix * CV + cv], br = r1[xhi * CV + cv];
  const int OW = W * fx;
  T* out = reinterpret_cast<T*>(y) + (((size_t)row * fy) * OW + (size_t)ix * fx) * CV + cv;
  for (int dy = 0; dy < fy; ++dy) {
    const float yl = (float)dy / (float)fy;
    for (int dx = 0; dx < fx; ++dx) {
      const float xl = (float)dx / (float)fx;
      const T top = rz_lerp(tl, tr, xl), bot = rz_lerp(bl, br, xl);
      out[((size_t)dy * OW + dx) * CV] = rz_lerp(top, bot, yl);
    }
  }
}

// instance-norm apply FUSED with the factor-2 legacy-bilinear resize that follows it in the decoder (models.py:528-538, 554-561, 571-578):
// up[n, 2i + a, 2j + b, :] = mean over the (1 + a) x (1 + b) block of y[n, i.., j.., :] with the neighbour indices clamped to the map, y the
// normalised, styled, activated tensor -- which is never written.  One thread = one low-resolution pixel x 4 channels: it normalises its
// own value and its right / lower / diagonal neighbours (re-read through L1) and writes the 2x2 output block; the call is bound by
// writing `up` (4x the input).  The record receives max|y| (a bilinear interpolation is a convex combination).
__global__ __launch_bounds__(256) void in_apply_resize2x_kernel(const float* __restrict__ x, const float* __restrict__ style,
                                                                const float* __restrict__ mean_rstd, float* __restrict__ up, int H, int W,
                                                                int C, int act, float* __restrict__ yamax, float* __restrict__ lo) {
  const int CV = C >> 2;
  const int n = blockIdx.y;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;          // over H * W * CV
  float ymax = 0.f;
  if (j < (long)H * W * CV) {
    const int cv = (int)(j % CV);
    const int pix = (int)(j / CV), iy = pix / W, ix = pix - iy * W;
    const int c = cv * 4;
    const float4 mu = *reinterpret_cast<const float4*>(mean_rstd + (size_t)n * 2 * C + c);
    const float4 rs = *reinterpret_cast<const float4*>(mean_rstd + (size_t)n * 2 * C + C + c);
    float4 s0 = *reinterpret_cast<const float4*>(style + (size_t)n * 2 * C + c);
    const float4 s1 = *reinterpret_cast<const float4*>(style + (size_t)n * 2 * C + C + c);
    s0.x += 1.f; s0.y += 1.f; s0.z += 1.f; s0.w += 1.f;
    const int yh = min(iy + 1, H - 1), xh = min(ix + 1, W - 1);
    const float* xb = x + (size_t)n * H * W * C + c;
    auto norm = [&](int r, int q) -> float4 {
      const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)r * W + q) * C);
      float4 o;
      o.x = ladder_act_fn((v.x - mu.x) * rs.x * s0.x + s1.x, act);
      o.y = ladder_act_fn((v.y - mu.y) * rs.y * s0.y + s1.y, act);
      o.z = ladder_act_fn((v.z - mu.z) * rs.z * s0.z + s1.z, act);
      o.w = ladder_act_fn((v.w - mu.w) * rs.w * s0.w + s1.w, act);
      return o;
    };
    const float4 tl = norm(iy, ix), tr = norm(iy, xh), bl = norm(yh, ix), br = norm(yh, xh);
    ymax = fmaxf(fmaxf(fabsf(tl.x), fabsf(tl.y)), fmaxf(fabsf(tl.z), fabsf(tl.w)));
    // the same arithmetic as resize_fwd_kernel (lerp along x, then along y, weights 0 and 1/2)
    const float4 top = rz_lerp(tl, tr, 0.5f), bot = rz_lerp(bl, br, 0.5f);
    const int OW = 2 * W;
    float* o = up + (((size_t)n * 2 * H + 2 * iy) * OW + 2 * ix) * C + c;
    *reinterpret_cast<float4*>(o) = tl;
    if (lo != nullptr) *reinterpret_cast<float4*>(lo + (((size_t)n * H + iy) * W + ix) * C + c) = tl;      // the normalised tensor itself (= up[2i][2j])
    *reinterpret_cast<float4*>(o + C) = top;
    *reinterpret_cast<float4*>(o + (size_t)OW * C) = rz_lerp(tl, bl, 0.5f);
    *reinterpret_cast<float4*>(o + (size_t)OW * C + C) = rz_lerp(top, bot, 0.5f);
  }
  if (yamax != nullptr) amax_commit_block_sample(ymax, yamax, n);
}

// transpose of the map above in gather form (no atomics): input pixel (iy,ix) collects every output
// pixel whose lo or hi index equals it, in (oy, ox) order.
template <int V>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int CV,
                                                         int fy, int fx, int bpr) {
  using T = typename ResizeVec<V>::T;
  const int row = blockIdx.x / bpr;
  const int j = (blockIdx.x - row * bpr) * blockDim.x + threadIdx.x;
  if (j >= W * CV) return;
  const int n = row / H, iy = row - n * H;
  const int ix = j / CV, cv = j - ix * CV;
  const int OH = H * fy, OW = W * fx;
  T acc;
  rz_zero(acc);
  const int oy0 = max(0, (iy - 1) * fy), oy1 = min(OH, (iy + 1) * fy);
  const int ox0 = max(0, (ix - 1) * fx), ox1 = min(OW, (ix + 1) * fx);
  const T* base = reinterpret_cast<const T*>(dy) + (size_t)n * OH * OW * CV + cv;
  for (int oy = oy0; oy < oy1; ++oy) {
    const int ylo = oy / fy, yhi = min(ylo + 1, H - 1);
    const float yl = (float)(oy - ylo * fy) / (float)fy;
    const float wy = (ylo == iy ? 1.f - yl : 0.f) + (yhi == iy ? yl : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox0; ox < ox1; ++ox) {
      const int xlo = ox / fx, xhi = min(xlo + 1, W - 1);
      const float xl = (float)(ox - xlo * fx) / (float)fx;
      const float wx = (xlo == ix ? 1.f - xl : 0.f) + (xhi == ix ? xl : 0.f);
      if (wx == 0.f) continue;
      rz_fma(acc, wy * wx, base[((size_t)oy * OW + ox) * CV]);
    }
  }
  reinterpret_cast<T*>(dx)[((size_t)row * W + ix) * CV + cv] = acc;
}

// The factor-2 case of the transpose (every resize of the CelebA decoder but one): output row 2i carries weight 1 to input row i, rows
// 2i-1 and 2i+1 weight 1/2 (row 2H-1 weight 1: its upper neighbour is clamped to H-1) -- a 3x3 footprint with constant weights.  All
// nine loads are issued before the first add (the generic kernel walks a 4x4 window behind integer divisions, one load at a time),
// the sum runs in the same (oy, ox) order.
template <int V>
__global__ __launch_bounds__(256) void resize_bwd_x2_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int CV,
                                                            int bpr, const float* __restrict__ gate_y, int gate_act) {
  using T = typename ResizeVec<V>::T;
  const int row = blockIdx.x / bpr;
  const int j = (blockIdx.x - row * bpr) * blockDim.x + threadIdx.x;
  if (j >= W * CV) return;
  const int n = row / H, iy = row - n * H;
  const int ix = j / CV, cv = j - ix * CV;
  const int OW = 2 * W;
  const T* base = reinterpret_cast<const T*>(dy) + (size_t)n * (2 * H) * OW * CV + cv;
  const float wy[3] = {iy > 0 ? 0.5f : 0.f, 1.f, iy == H - 1 ? 1.f : 0.5f};
  const float wx[3] = {ix > 0 ? 0.5f : 0.f, 1.f, ix == W - 1 ? 1.f : 0.5f};
  T v[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int oy = max(2 * iy - 1 + a, 0), ox = max(2 * ix - 1 + b, 0);        // (clamped taps carry weight 0)
      v[a][b] = base[((size_t)oy * OW + ox) * CV];
    }
  T acc;
  rz_zero(acc);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (wy[a] != 0.f && wx[b] != 0.f) rz_fma(acc, wy[a] * wx[b], v[a][b]);
  const size_t o = ((size_t)row * W + ix) * CV + cv;
  if (gate_y != nullptr) rz_gate(acc, reinterpret_cast<const T*>(gate_y)[o], gate_act);   // the producer's activation backward, fused
  reinterpret_cast<T*>(dx)[o] = acc;
}

// ----------------------------------------------------------------------------- minibatch assembly from an HBM-resident data set
// out[b, :] = scale * float(src[idx[b], :]): the reference's input pipeline (uint8 CelebA pixels * 1/255, models.py:354-371;
// MNIST floats, data_loader.py:19-33) for a data set that lives in device memory (CelebA train split as uint8 = 8.8 GB of the
// 288 GB): shuffle by index, gather and normalise in one pass, no host work per step.  16 source bytes -> 16 floats per thread.
__global__ __launch_bounds__(256) void gather_rows_u8_kernel(const unsigned char* __restrict__ src, const long long* __restrict__ idx,
                                                             float* __restrict__ out, int B, long long D, float scale) {
  const long long D16 = D >> 4;
  const long long total = (long long)B * D16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / D16);
    const long long c = i - (long long)b * D16;
    const uint4 v = *reinterpret_cast<const uint4*>(src + idx[b] * D + c * 16);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float4* o = reinterpret_cast<float4*>(out + (long long)b * D + c * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = make_float4((float)(w[k] & 0xFF) * scale, (float)((w[k] >> 8) & 0xFF) * scale, (float)((w[k] >> 16) & 0xFF) * scale,
                         (float)(w[k] >> 24) * scale);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_scalar_kernel(const T* __restrict__ src, const long long* __restrict__ idx,
                                                                 float* __restrict__ out, int B, long long D, float scale) {
  const long long total = (long long)B * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / D);
    const long long c = i - (long long)b * D;
    out[i] = (float)src[idx[b] * D + c] * scale;
  }
}

// forward: y[n, h*r+i, w*r+j, c'] = x[n, h, w, (i*r+j)*C' + c'] (DCR) ; inverse swaps the roles.  For a fixed output row (n, h*r+i) the
// r*C' floats of every w are one contiguous segment in BOTH layouts (source offset w*C + i*r*C', output offset w*r*C'), so the op is
// a row-wise copy of W segments: `bpr` blocks per output row, V floats per thread, 32-bit index arithmetic.
template <int V>
__global__ __launch_bounds__(256) void d2s_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int r,
                                                  int inverse, int bpr) {
  using T = typename ResizeVec<V>::T;
  const int row = blockIdx.x / bpr;                  // n*H*r + oh
  const int e = ((blockIdx.x - row * bpr) * blockDim.x + threadIdx.x) * V;
  const int seg = C / r;                             // r * C'
  if (e >= W * seg) return;
  const int nh = row / r, i = row - nh * r;          // nh = n*H + h
  const int w = e / seg, q = e - w * seg;
  const size_t small = ((size_t)nh * W + w) * C + (size_t)i * seg + q;
  const size_t big = (size_t)row * W * seg + e;
  if (inverse) *reinterpret_cast<T*>(y + small) = *reinterpret_cast<const T*>(x + big);
  else *reinterpret_cast<T*>(y + big) = *reinterpret_cast<const T*>(x + small);
}

__global__ void pad_sym_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int p) {
  const int OH = H + 2 * p, OW = W + 2 * p;
  const size_t total = (size_t)N * OH * OW * C;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int c = (int)(o % C);
    size_t t = o / C;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    int h = oh - p, w = ow - p;
    h = h < 0 ? -h - 1 : (h >= H ? 2 * H - 1 - h : h);
    w = w < 0 ? -w - 1 : (w >= W ? 2 * W - 1 - w : w);
    y[o] = x[(((size_t)n * H + h) * W + w) * C + c];
  }
}

// transpose of the SYMMETRIC pad in gather form: an input pixel collects its own position and its mirror images (p <= H, W).
__global__ void pad_sym_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C, int p) {
  const int OH = H + 2 * p, OW = W + 2 * p;
  const size_t total = (size_t)N * H * W * C;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int c = (int)(o % C);
    size_t t = o / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    // padded rows that read input row h: p + h, and the mirrors p - 1 - h (top, if h < p) and p + 2H - 1 - h (bottom, if h >= H - p)
    int rows[3], cols[3], nr = 0, nc = 0;
    rows[nr++] = p + h;
    if (h < p) rows[nr++] = p - 1 - h;
    if (h >= H - p) rows[nr++] = p + 2 * H - 1 - h;
    cols[nc++] = p + w;
    if (w < p) cols[nc++] = p - 1 - w;
    if (w >= W - p) cols[nc++] = p + 2 * W - 1 - w;
    float a = 0.f;
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) a += dy[(((size_t)n * OH + rows[i]) * OW + cols[j]) * C + c];
    dx[o] = a;
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n, int act) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = g.x * ladder_act_grad_from_out(v.x, act);
    o.y = g.y * ladder_act_grad_from_out(v.y, act);
    o.z = g.z * ladder_act_grad_from_out(v.z, act);
    o.w = g.w * ladder_act_grad_from_out(v.w, act);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = dy[i] * ladder_act_grad_from_out(y[i], act);
}

// (in == out is a documented use -- in-place scaling --, so neither pointer is __restrict__)
__global__ void axpy_kernel(const float* in, float* out, size_t n, float scale, int accumulate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = accumulate == 2 ? scale : (accumulate ? out[i] + scale * in[i] : scale * in[i]);      // 2: fill with `scale` (in is not read)
}

inline unsigned ew_grid(size_t work_items) {
  size_t g = (work_items + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;   // ~8 workgroups per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

int ladder_bn_fwd_stats(const float* x, float* sums, size_t rows, int C, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (rows == 0 || C <= 0) return LADDER_E_SHAPE;
  const size_t nblk = stats_nblk(rows);
  if (ws_bytes < nblk * 2 * (size_t)C * sizeof(double)) return LADDER_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(sums) & 7u) != 0 || (reinterpret_cast<uintptr_t>(ws) & 7u) != 0) return LADDER_E_ALIGN;
  const size_t rpb = (rows + nblk - 1) / nblk;
  if (C % 4 == 0 && ladder_aligned16(x))
    hipLaunchKernelGGL(colstats64_stage1_v4<false>, dim3((C + 63) / 64, (unsigned)nblk), dim3(256), 0, stream, x, (double*)ws, (float*)nullptr, rows, C, rpb);
  else
    hipLaunchKernelGGL(colstats64_stage1, dim3((C + 63) / 64, (unsigned)nblk), dim3(256), 0, stream, x, (double*)ws, rows, C, rpb);
  hipLaunchKernelGGL(colstats64_stage2, dim3((2 * C + 63) / 64), dim3(1024), 0, stream, (const double*)ws, (const float*)nullptr, sums, (int)nblk, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// sums[0:C] = sum over blocks of partials[b][0][:], sums[C:2C] = ... of partials[b][1][:]  (partials [nblk][2][C]; fixed order, fp64)
int ladder_bn_stats_from_partials(const float* partials, int nblk, float* sums, int C, ladder_stream_t stream) {
  if (nblk <= 0 || C <= 0) return LADDER_E_SHAPE;
  hipLaunchKernelGGL(colstats_stage2_wide, dim3((2 * C + 15) / 16), dim3(256), 0, stream, partials, sums, nblk, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_bn_fwd_apply(const float* x, const float* sums, double count, const float* gamma, const float* beta, float* y,
                        float* mean_rstd, size_t rows, int C, float eps, int act, ladder_stream_t stream) {
  return ladder_bn_fwd_apply_absmax(x, sums, count, gamma, beta, y, mean_rstd, rows, C, eps, act, nullptr, stream);
}

int ladder_bn_fwd_apply_absmax(const float* x, const float* sums, double count, const float* gamma, const float* beta, float* y,
                               float* mean_rstd, size_t rows, int C, float eps, int act, float* y_absmax, ladder_stream_t stream) {
  if (rows == 0 || C <= 0 || count <= 0) return LADDER_E_SHAPE;
  if (y_absmax != nullptr) {                     // the record is produced by the vectorised kernel only
    if (!(C % 4 == 0 && ladder_aligned16(x) && ladder_aligned16(y))) return LADDER_E_SHAPE;
  }
  const size_t n = rows * (size_t)C;
  static const bool fold = getenv("LADDER_DISABLE_BN_FOLD") == nullptr;      // (A / B switch: the finalisation as a launch of its own)
  if (fold && y_absmax == nullptr && C % 4 == 0 && ladder_aligned16(x) && ladder_aligned16(y) && ladder_aligned16(gamma) && ladder_aligned16(beta) &&
      ((size_t)ew_grid(n / 4) * 256 * 4) % (size_t)C == 0) {
    hipLaunchKernelGGL(bn_apply_fin_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, x, sums, count, eps, mean_rstd, gamma, beta, y, n, C, act);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  // (the finalize kernel clears the record: no separate memset launch)
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, count, eps, mean_rstd, C, y_absmax);
  if (C % 4 == 0 && ladder_aligned16(x) && ladder_aligned16(y))
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, x, mean_rstd, gamma, beta, y, n, C, act, y_absmax);
  else
    hipLaunchKernelGGL(bn_apply_scalar_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, mean_rstd, gamma, beta, y, n, C, act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// sums4 = the record in its minmax form (2C doubles sum x | sum x^2, then min x | max x as floats); workspace: 2 x ladder_bn_workspace_bytes
int ladder_bn_fwd_stats_minmax(const float* x, float* sums4, size_t rows, int C, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (rows == 0 || C <= 0 || (C % 4) != 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x)) return LADDER_E_ALIGN;
  const size_t nblk = stats_nblk(rows);
  if (ws_bytes < nblk * (2 * (size_t)C * sizeof(double) + 2 * (size_t)C * sizeof(float))) return LADDER_E_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(sums4) & 7u) != 0 || (reinterpret_cast<uintptr_t>(ws) & 7u) != 0) return LADDER_E_ALIGN;
  const size_t rpb = (rows + nblk - 1) / nblk;
  float* wsmm = reinterpret_cast<float*>(reinterpret_cast<double*>(ws) + nblk * 2 * (size_t)C);
  hipLaunchKernelGGL(colstats64_stage1_v4<true>, dim3((C + 63) / 64, (unsigned)nblk), dim3(256), 0, stream, x, (double*)ws, wsmm, rows, C, rpb);
  hipLaunchKernelGGL(colstats64_stage2, dim3((4 * C + 63) / 64), dim3(1024), 0, stream, (const double*)ws, (const float*)wsmm, sums4, (int)nblk, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// the second stage alone, for [nblk][4][C] partials emitted by a convolution epilogue
int ladder_bn_stats_minmax_from_partials(const float* partials, int nblk, float* sums4, int C, ladder_stream_t stream) {
  if (nblk <= 0 || C <= 0) return LADDER_E_SHAPE;
  hipLaunchKernelGGL(colstats_minmax_stage2, dim3((4 * C + 15) / 16), dim3(1024), 0, stream, partials, sums4, nblk, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// Apply that writes the two fp16 planes of y (ladder_presplit layout for LADDER_PREC_F16X3: plane-major, 16 zero bytes behind) and the
// record of max|y| it scaled them with; the fp32 tensor itself is optional (y == NULL: never written).
int ladder_bn_fwd_apply_planes(const float* x, const float* sums4, double count, const float* gamma, const float* beta, float* y, void* y_planes,
                               float* mean_rstd, size_t rows, int C, float eps, int act, float* y_absmax, ladder_stream_t stream) {
  if (rows == 0 || C <= 0 || count <= 0 || (C % 4) != 0 || y_planes == nullptr || y_absmax == nullptr) return LADDER_E_SHAPE;
  const size_t n = rows * (size_t)C;
  if ((n % 8) != 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(y_planes) || (y != nullptr && !ladder_aligned16(y))) return LADDER_E_ALIGN;
  hipLaunchKernelGGL(bn_finalize_minmax_kernel, dim3(1), dim3(256), 0, stream, sums4, count, eps, mean_rstd, gamma, beta, C, act, y_absmax);
  hipLaunchKernelGGL(bn_apply_planes_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, x, (const float*)mean_rstd, gamma, beta, y,
                     (uint16_t*)y_planes, n, C, act, (const float*)y_absmax);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_bn_bwd_stats(const float* dy, const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                        float* dsums, size_t rows, int C, int act, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (rows == 0 || C <= 0) return LADDER_E_SHAPE;
  const size_t nblk = stats_nblk(rows);
  if (ws_bytes < nblk * 2 * (size_t)C * sizeof(float)) return LADDER_E_WORKSPACE;
  const size_t rpb = (rows + nblk - 1) / nblk;
  if (C % 4 == 0 && ladder_aligned16(x) && ladder_aligned16(dy))
    hipLaunchKernelGGL(colstats_stage1_v4<2>, dim3((C + 63) / 64, (unsigned)nblk), dim3(256), 0, stream, dy, x, mean_rstd, gamma, beta,
                       (float*)ws, rows, C, rpb, act);
  else
    hipLaunchKernelGGL(colstats_stage1<2>, dim3((C + 63) / 64, (unsigned)nblk), dim3(256), 0, stream, dy, x, mean_rstd, gamma, beta,
                       (float*)ws, rows, C, rpb, act);
  hipLaunchKernelGGL(colstats_stage2, dim3((2 * C + 63) / 64), dim3(1024), 0, stream, (const float*)ws, dsums, (int)nblk, C);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_bn_bwd_apply(const float* dy, const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                        const float* dsums, double count, float* dx, float* dgamma, float* dbeta, size_t rows, int C, int act,
                        ladder_stream_t stream) {
  return ladder_bn_bwd_apply_absmax(dy, x, mean_rstd, gamma, beta, dsums, count, dx, dgamma, dbeta, rows, C, act, nullptr, stream);
}

int ladder_bn_bwd_apply_absmax(const float* dy, const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                               const float* dsums, double count, float* dx, float* dgamma, float* dbeta, size_t rows, int C, int act,
                               float* dx_absmax, ladder_stream_t stream) {
  if (rows == 0 || C <= 0 || count <= 0) return LADDER_E_SHAPE;
  const size_t n = rows * (size_t)C;
  if (dx_absmax != nullptr) {
    if (!(dx != nullptr && C % 4 == 0 && ladder_aligned16(x) && ladder_aligned16(dy) && ladder_aligned16(dx))) return LADDER_E_SHAPE;
  }
  // the parameter-gradient kernel runs first and clears the record on its way (no memset launch); without parameter gradients: memset
  const bool pgrad = dgamma != nullptr && dbeta != nullptr;
  const bool v4 = dx != nullptr && C % 4 == 0 && ladder_aligned16(x) && ladder_aligned16(dy) && ladder_aligned16(dx);
  static const bool fold = getenv("LADDER_DISABLE_BN_FOLD") == nullptr;
  const bool pgrad_rides = fold && pgrad && v4 && dx_absmax == nullptr;   // strict fp32: workgroup 0 of the apply kernel copies them
  if (pgrad && !pgrad_rides)
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, dsums, dgamma, dbeta, C, dx_absmax);
  else if (!pgrad && dx_absmax != nullptr && hipMemsetAsync(dx_absmax, 0, LADDER_ABSMAX_FLOATS * sizeof(float), stream) != hipSuccess)
    return LADDER_E_LAUNCH;
  if (dx != nullptr) {
    if (v4)
      hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, dy, x, mean_rstd, gamma, beta, dsums,
                         (float)(1.0 / count), dx, n, C, act, dx_absmax, pgrad_rides ? dgamma : (float*)nullptr, pgrad_rides ? dbeta : (float*)nullptr);
    else
      hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, dy, x, mean_rstd, gamma, beta, dsums,
                         (float)(1.0 / count), dx, n, C, act);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

size_t ladder_bn_workspace_bytes(size_t rows, int C) { return stats_nblk(rows) * 2 * (size_t)C * sizeof(double); }   // (fp64 partials of the forward statistics)

size_t ladder_in_style_workspace_bytes(int N, int HW, int C) { return (size_t)N * in_split(N, HW, C) * 2 * C * sizeof(float); }

int ladder_in_style_fwd(const float* x, const float* style, float* y, float* mean_rstd, int N, int HW, int C, float eps, int act,
                        void* ws, size_t ws_bytes, ladder_stream_t stream) {
  return ladder_in_style_fwd_absmax(x, style, y, mean_rstd, N, HW, C, eps, act, ws, ws_bytes, nullptr, stream);
}

int ladder_in_style_fwd_absmax(const float* x, const float* style, float* y, float* mean_rstd, int N, int HW, int C, float eps, int act,
                               void* ws, size_t ws_bytes, float* y_absmax, ladder_stream_t stream) {
  if (N <= 0 || HW <= 0 || C <= 0) return LADDER_E_SHAPE;
  if (y_absmax != nullptr) {
    if (!(C % 4 == 0 && ws != nullptr && ws_bytes >= ladder_in_style_workspace_bytes(N, HW, C) && ladder_aligned16(x) && ladder_aligned16(y)))
      return LADDER_E_SHAPE;                     // the record is produced by the vectorised three-kernel path only (its finalize kernel clears it)
  }
  if (C % 4 == 0 && ws != nullptr && ws_bytes >= ladder_in_style_workspace_bytes(N, HW, C) && ladder_aligned16(x) && ladder_aligned16(y)) {
    const int sp = in_split(N, HW, C);
    dim3 grid((C + 63) / 64, N, sp);
    hipLaunchKernelGGL(in_stats_kernel, grid, dim3(256), 0, stream, x, (float*)ws, HW, C, sp);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, stream, x, (const float*)ws, mean_rstd, N, HW, C, sp, eps, y_absmax);
    hipLaunchKernelGGL(in_apply_kernel, grid, dim3(256), 0, stream, x, style, (const float*)mean_rstd, y, HW, C, act, sp, y_absmax);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  hipLaunchKernelGGL(in_style_fwd_kernel, dim3((C + 63) / 64, N), dim3(256), 0, stream, x, style, y, mean_rstd, HW, C, eps, act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// Forward + the factor-2 bilinear resize of the decoder in one pass over x: `up` [N, 2H, 2W, C]; the normalised tensor itself is not
// written.  Needs the vectorised path (C % 4 == 0, workspace, 16-byte alignment).
int ladder_in_style_fwd_resize2x(const float* x, const float* style, float* up, float* mean_rstd, int N, int H, int W, int C, float eps,
                                 int act, void* ws, size_t ws_bytes, float* up_absmax, ladder_stream_t stream) {
  return ladder_in_style_fwd_resize2x_keep(x, style, up, nullptr, mean_rstd, N, H, W, C, eps, act, ws, ws_bytes, up_absmax, stream);
}

// ... and, optionally, the normalised tensor y [N, H, W, C] beside it (a quarter of `up`, written in the same pass): the input of the
// upsample-fused convolution of a training forward (ladder_conv3x3_up2_split), which would otherwise gather it out of the 4x larger tensor.
int ladder_in_style_fwd_resize2x_keep(const float* x, const float* style, float* up, float* y, float* mean_rstd, int N, int H, int W, int C,
                                      float eps, int act, void* ws, size_t ws_bytes, float* up_absmax, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return LADDER_E_SHAPE;
  if (y != nullptr && !ladder_aligned16(y)) return LADDER_E_ALIGN;
  const int HW = H * W;
  if (!(C % 4 == 0 && ws != nullptr && ws_bytes >= ladder_in_style_workspace_bytes(N, HW, C) && ladder_aligned16(x) && ladder_aligned16(up)))
    return LADDER_E_SHAPE;
  const int sp = in_split(N, HW, C);
  dim3 grid((C + 63) / 64, N, sp);
  hipLaunchKernelGGL(in_stats_kernel, grid, dim3(256), 0, stream, x, (float*)ws, HW, C, sp);
  hipLaunchKernelGGL(in_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, stream, x, (const float*)ws, mean_rstd, N, HW, C, sp, eps, up_absmax);
  const long units = (long)HW * (C / 4);
  hipLaunchKernelGGL(in_apply_resize2x_kernel, dim3((unsigned)((units + 255) / 256), N), dim3(256), 0, stream, x, style,
                     (const float*)mean_rstd, up, H, W, C, act, up_absmax, y);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_in_style_bwd(const float* dy, const float* x, const float* style, const float* mean_rstd, float* dx, float* dstyle,
                        int N, int HW, int C, int act, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  return ladder_in_style_bwd_absmax(dy, x, style, mean_rstd, dx, dstyle, N, HW, C, act, ws, ws_bytes, nullptr, stream);
}

int ladder_in_style_bwd_absmax(const float* dy, const float* x, const float* style, const float* mean_rstd, float* dx, float* dstyle,
                               int N, int HW, int C, int act, void* ws, size_t ws_bytes, float* dx_absmax, ladder_stream_t stream) {
  if (N <= 0 || HW <= 0 || C <= 0) return LADDER_E_SHAPE;
  if (dx_absmax != nullptr) {
    if (!(C % 4 == 0 && ws != nullptr && ws_bytes >= ladder_in_style_workspace_bytes(N, HW, C) && ladder_aligned16(x) && ladder_aligned16(dy) &&
          ladder_aligned16(dx)))
      return LADDER_E_SHAPE;
  }
  if (C % 4 == 0 && ws != nullptr && ws_bytes >= ladder_in_style_workspace_bytes(N, HW, C) && ladder_aligned16(x) && ladder_aligned16(dy) &&
      ladder_aligned16(dx)) {
    const int sp = in_split(N, HW, C);
    dim3 grid((C + 63) / 64, N, sp);
    hipLaunchKernelGGL(in_bwd_stats_kernel, grid, dim3(256), 0, stream, dy, x, style, mean_rstd, (float*)ws, HW, C, act, sp);
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, stream, (const float*)ws, dstyle, N, C, sp, dx_absmax);
    hipLaunchKernelGGL(in_bwd_apply_kernel, grid, dim3(256), 0, stream, dy, x, style, mean_rstd, (const float*)dstyle, dx, HW, C, act, sp, dx_absmax);
    LADDER_CHECK_LAUNCH();
    return LADDER_OK;
  }
  hipLaunchKernelGGL(in_style_bwd_kernel, dim3((C + 63) / 64, N), dim3(256), 0, stream, dy, x, style, mean_rstd, dx, dstyle, HW, C, act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

static int resize_launch(bool fwd, const float* a, float* b, int N, int H, int W, int C, int OH, int OW, hipStream_t stream,
                         const float* gate_y = nullptr, int gate_act = 0) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH < H || OW < W || OH % H || OW % W) return LADDER_E_SHAPE;
  if (gate_y != nullptr && (fwd || OH != 2 * H || OW != 2 * W)) return LADDER_E_SHAPE;      // the fused gate exists for the factor-2 transpose
  const bool v4 = (C % 4 == 0) && ladder_aligned16(a) && ladder_aligned16(b) && (gate_y == nullptr || ladder_aligned16(gate_y));
  const int CV = v4 ? C / 4 : C;
  const long bprl = ((long)W * CV + 255) / 256;
  if (bprl * N * H >= (1L << 31)) return LADDER_E_SHAPE;
  const int bpr = (int)bprl;
  const dim3 grid((unsigned)(bprl * N * H)), block(256);
  const int fy = OH / H, fx = OW / W;
  if (fwd) {
    if (v4) hipLaunchKernelGGL(resize_fwd_kernel<4>, grid, block, 0, stream, a, b, H, W, CV, fy, fx, bpr);
    else hipLaunchKernelGGL(resize_fwd_kernel<1>, grid, block, 0, stream, a, b, H, W, CV, fy, fx, bpr);
  } else {
    if (fy == 2 && fx == 2) {
      if (v4) hipLaunchKernelGGL(resize_bwd_x2_kernel<4>, grid, block, 0, stream, a, b, H, W, CV, bpr, gate_y, gate_act);
      else hipLaunchKernelGGL(resize_bwd_x2_kernel<1>, grid, block, 0, stream, a, b, H, W, CV, bpr, gate_y, gate_act);
    } else if (v4) hipLaunchKernelGGL(resize_bwd_kernel<4>, grid, block, 0, stream, a, b, H, W, CV, fy, fx, bpr);
    else hipLaunchKernelGGL(resize_bwd_kernel<1>, grid, block, 0, stream, a, b, H, W, CV, fy, fx, bpr);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_resize_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, ladder_stream_t stream) {
  return resize_launch(true, x, y, N, H, W, C, OH, OW, stream);
}

int ladder_resize_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, ladder_stream_t stream) {
  return resize_launch(false, dy, dx, N, H, W, C, OH, OW, stream);
}

// factor-2 transpose with the activation backward of the layer that produced the resized tensor fused in: dx *= act'(gate_y), gate_y = that
// layer's OUTPUT [N,H,W,C] (saves the separate read-modify-write pass of ladder_act_bwd)
int ladder_resize_bilinear_bwd_gated(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, const float* gate_y, int gate_act,
                                     ladder_stream_t stream) {
  if (gate_y == nullptr) return LADDER_E_SHAPE;
  return resize_launch(false, dy, dx, N, H, W, C, OH, OW, stream, gate_y, gate_act);
}

int ladder_gather_rows(const void* src, int src_is_u8, const int64_t* idx, float* out, int B, int64_t D, float scale,
                       ladder_stream_t stream) {
  if (B <= 0 || D <= 0) return LADDER_E_SHAPE;
  const long long total = (long long)B * D;
  if (src_is_u8 && (D % 16) == 0 && ladder_aligned16(src) && ladder_aligned16(out)) {
    hipLaunchKernelGGL(gather_rows_u8_kernel, dim3(ew_grid((size_t)(total / 16))), dim3(256), 0, stream, (const unsigned char*)src,
                       (const long long*)idx, out, B, (long long)D, scale);
  } else if (src_is_u8) {
    hipLaunchKernelGGL(gather_rows_scalar_kernel<unsigned char>, dim3(ew_grid((size_t)total)), dim3(256), 0, stream,
                       (const unsigned char*)src, (const long long*)idx, out, B, (long long)D, scale);
  } else {
    hipLaunchKernelGGL(gather_rows_scalar_kernel<float>, dim3(ew_grid((size_t)total)), dim3(256), 0, stream, (const float*)src,
                       (const long long*)idx, out, B, (long long)D, scale);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_depth_to_space(const float* x, float* y, int N, int H, int W, int C, int r, int inverse, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0 || C % (r * r)) return LADDER_E_SHAPE;
  const int seg = C / r;                              // contiguous run shared by both layouts
  const bool v4 = (seg % 4 == 0) && ladder_aligned16(x) && ladder_aligned16(y);
  const long per_row = (long)W * seg / (v4 ? 4 : 1);
  const long bpr = (per_row + 255) / 256, rows = (long)N * H * r;
  if (bpr * rows >= (1L << 31) || (long)W * seg >= (1L << 31)) return LADDER_E_SHAPE;
  if (v4) hipLaunchKernelGGL(d2s_kernel<4>, dim3((unsigned)(bpr * rows)), dim3(256), 0, stream, x, y, H, W, C, r, inverse, (int)bpr);
  else hipLaunchKernelGGL(d2s_kernel<1>, dim3((unsigned)(bpr * rows)), dim3(256), 0, stream, x, y, H, W, C, r, inverse, (int)bpr);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_pad_symmetric(const float* x, float* y, int N, int H, int W, int C, int p, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || p < 0 || p > H || p > W) return LADDER_E_SHAPE;
  hipLaunchKernelGGL(pad_sym_kernel, dim3(ew_grid((size_t)N * (H + 2 * p) * (W + 2 * p) * C)), dim3(256), 0, stream, x, y, N, H, W, C, p);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_pad_symmetric_bwd(const float* dy, float* dx, int N, int H, int W, int C, int p, ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || p < 0 || p > H || p > W) return LADDER_E_SHAPE;
  hipLaunchKernelGGL(pad_sym_bwd_kernel, dim3(ew_grid((size_t)N * H * W * C)), dim3(256), 0, stream, dy, dx, N, H, W, C, p);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, ladder_stream_t stream) {
  if (n == 0) return LADDER_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, stream, dy, y, dx, n, act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_axpy(const float* in, float* out, size_t n, float scale, int accumulate, ladder_stream_t stream) {
  if (n == 0) return LADDER_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, in, out, n, scale, accumulate);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

}  // extern "C"
