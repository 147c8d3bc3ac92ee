// The image-side convolution of the CelebA encoder: 3x3, stride 2, SAME, 3 -> Cout channels over a 128x128 RGB batch (codes/models.py:398-405).
// K = 27 is far too short for the generic gather kernels (they pad it to 32 fp32 MFMA K-steps and spend their time on address
// arithmetic: 140 us forward, 143 us filter gradient at batch 128 -- the call is bound by WRITING the 268 MB activation, ~60 us).
//
// Forward (conv_rgb_s2_fwd_kernel): one workgroup = an 8x32 output patch x 128 channels.
//   1. the 17 x 65 x 3 input patch goes to LDS as fp32 (coalesced row segments; 13 KB);
//   2. the im2col matrix A[256 pixels][32] (27 taps + zero padding) and the filter B[32][128] are built in LDS as TWO fp16 planes
//      (split16.h: f16x3, three MFMAs per product, fp32-class) with a PER-WORKGROUP power-of-two scale from the patch's / the filter's
//      own absolute maximum -- the scale only has to be constant inside one accumulation, so no tensor-wide absmax pass is needed;
//   3. 2 K-steps x 3 plane products of v_mfma_f32_32x32x16_f16 per 32x32 tile (lane = channel, registers = pixels), un-scale + bias +
//      activation, stores of whole 128-byte lines.
// Filter gradient (conv_rgb_s2_wgrad_kernel): dW[27][Cout] = sum_pixels A[pixel][27] * dY[pixel][Cout]; the reduction index is the
// pixel, so both operands are read with the transposing LDS read (ds_read_b64_tr_b16) from pixel-major fp16 plane images exactly as in
// wgrad3x3_split_kernel; each workgroup reduces a run of patches, partials are summed in a fixed order by ladder_reduce_splits.
#include "split16.h"

namespace {

constexpr int RGB_TH = 8, RGB_TW = 32, RGB_PIX = RGB_TH * RGB_TW;                 // output patch
constexpr int RGB_PH = 2 * RGB_TH + 1, RGB_PW = 2 * RGB_TW + 1;                   // input patch 17 x 65 (x 3 channels)
constexpr int RGB_PATCH = RGB_PH * RGB_PW * 3;                                    // 3315 floats
constexpr int RGB_THREADS = 512;
constexpr int RGB_A_PLANE = 2 * 2 * RGB_PIX * 16;                                 // [k-step][channel octet][pixel][8 x 16 bit] = 16 KB
constexpr int RGB_B_PLANE = 2 * 2 * 128 * 16;                                     // 8 KB

__device__ __forceinline__ float block_max_512(float m, float* red) {             // all 512 threads; red = 8 floats of LDS
  m = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  float b = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) b = fmaxf(b, red[w]);
  return b;
}

// element k (0..26) of a pixel's im2col row, taken from the fp32 patch in LDS: k = (r*3 + s)*3 + ci
__device__ __forceinline__ float rgb_tap(const float* __restrict__ patch, int pr, int pc, int k) {
  const int tap = k / 3, ci = k - tap * 3, r = tap / 3, s = tap - r * 3;
  return patch[((2 * pr + r) * RGB_PW + 2 * pc + s) * 3 + ci];
}

// F32 (round 4: strict fp32): the im2col matrix A[256 pixels][28] (row stride 29: conflict-free fragment reads) and the filter B[28][128] are
// built in LDS as fp32 and multiplied by 14 K-steps of v_mfma_f32_32x32x2_f32 -- no scales, no planes; everything else (patch load, tile
// shape, lane = channel epilogue with whole-line stores, statistics) is shared.
constexpr int RGB_K32 = 28, RGB_LDA32 = RGB_K32 + 1;
template <bool F32>
__global__ __launch_bounds__(RGB_THREADS, 4) void conv_rgb_s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                                        const int N, const int H, const int W, const int Cout,
                                                                        const int act, const int tiles_n, float* __restrict__ stats_part) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * RGB_A_PLANE + 2 * RGB_B_PLANE];   // 48 KB; the fp32 patch aliases the B region + tail
  __shared__ __attribute__((aligned(16))) float patch[RGB_PATCH + 5];
  __shared__ float red[8];
  unsigned char* const Ab = lds;
  unsigned char* const Bb = lds + 2 * RGB_A_PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5, wm = wid >> 1, wn = wid & 1;
  const int Ho = H / 2, Wo = W / 2;
  const int tile = blockIdx.x, mt = tile / tiles_n, cot = tile - mt * tiles_n, n0 = cot * 128;
  const int tw_n = Wo / RGB_TW, th_n = Ho / RGB_TH;
  const int img = mt / (tw_n * th_n), rem = mt - img * (tw_n * th_n);
  const int h0 = (rem / tw_n) * RGB_TH, w0 = (rem % tw_n) * RGB_TW;

  // 1. input patch -> LDS (rows of 195 consecutive floats; the bottom row / right column of the image border are the SAME padding)
  float xmax = 0.f;
  for (int u = tid; u < RGB_PATCH; u += RGB_THREADS) {
    const int r = u / (RGB_PW * 3), c3 = u - r * (RGB_PW * 3);
    const int hi = 2 * h0 + r, wi3 = 2 * w0 * 3 + c3;
    float v = 0.f;
    if (hi < H && wi3 < W * 3) v = x[((long)img * H + hi) * W * 3 + wi3];
    patch[u] = v;
    xmax = fmaxf(xmax, fabsf(v));
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  float cx = 1.f, cw = 1.f;
  if constexpr (F32) {
    static_assert(RGB_PIX * RGB_LDA32 * 4 <= 2 * RGB_A_PLANE && RGB_K32 * 128 * 4 <= 2 * RGB_B_PLANE, "the fp32 operands fit the plane buffers");
    float* const A32 = reinterpret_cast<float*>(Ab);                 // [pixel][29]
    float* const B32 = reinterpret_cast<float*>(Bb);                 // [k][128]
    {
      const int co = tid & 127, kq = tid >> 7;                       // thread = (channel, 7 taps)
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int k = kq * 7 + j;
        B32[k * 128 + co] = (k < 27 && n0 + co < Cout) ? w[(long)k * Cout + n0 + co] : 0.f;
      }
    }
    __syncthreads();                                                 // (the patch is complete)
    {
      const int p = tid & 255, kh = tid >> 8, pr = p >> 5, pc = p & 31;   // thread = (pixel, K half)
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const int k = kh * 14 + j;
        A32[p * RGB_LDA32 + k] = k < 27 ? rgb_tap(patch, pr, pc, k) : 0.f;
      }
    }
    __syncthreads();
    const float* Afr = A32 + (wm * 64 + l31) * RGB_LDA32 + lh;
    const float* Bfr = B32 + lh * 128 + wn * 64 + l31;
#pragma unroll
    for (int ks = 0; ks < RGB_K32 / 2; ++ks) {
      float a[2], b[2];
      a[0] = Afr[2 * ks];
      a[1] = Afr[32 * RGB_LDA32 + 2 * ks];
      b[0] = Bfr[2 * ks * 128];
      b[1] = Bfr[2 * ks * 128 + 32];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // lane = channel
    }
  } else {
  // 2. filter block -> registers (thread = channel co, K octet kq), its maximum
  const int co = tid & 127, kq = tid >> 7;
  float wv[8], wmax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kq * 8 + j;
    wv[j] = (k < 27 && n0 + co < Cout) ? w[(long)k * Cout + n0 + co] : 0.f;
    wmax = fmaxf(wmax, fabsf(wv[j]));
  }
  cx = scale_from_absmax(block_max_512(xmax, red));
  cw = scale_from_absmax(block_max_512(wmax, red));     // (also orders the patch writes before the reads below)
  {
    uint2 lo[2], hi[2];
    split4<2, true>(make_float4(wv[0] * cw, wv[1] * cw, wv[2] * cw, wv[3] * cw), lo);
    split4<2, true>(make_float4(wv[4] * cw, wv[5] * cw, wv[6] * cw, wv[7] * cw), hi);
#pragma unroll
    for (int p = 0; p < 2; ++p)   // [plane][k-step kq >> 1][octet kq & 1][co][8 x 16 bit]
      *reinterpret_cast<uint4*>(Bb + p * RGB_B_PLANE + (((kq >> 1) * 2 + (kq & 1)) * 128 + co) * 16) = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
  }
  // 3. im2col rows: thread = (pixel, K half)
  {
    const int p = tid & 255, kh = tid >> 8, pr = p >> 5, pc = p & 31;
    float av[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = kh * 16 + j;
      av[j] = k < 27 ? rgb_tap(patch, pr, pc, k) * cx : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      uint2 lo[2], hi[2];
      split4<2, true>(make_float4(av[8 * o], av[8 * o + 1], av[8 * o + 2], av[8 * o + 3]), lo);
      split4<2, true>(make_float4(av[8 * o + 4], av[8 * o + 5], av[8 * o + 6], av[8 * o + 7]), hi);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        *reinterpret_cast<uint4*>(Ab + pl * RGB_A_PLANE + ((kh * 2 + o) * RGB_PIX + p) * 16) = make_uint4(lo[pl].x, lo[pl].y, hi[pl].x, hi[pl].y);
    }
  }
  __syncthreads();

#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 a[2][2], b[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        a[mi][p] = *reinterpret_cast<const uint4*>(Ab + p * RGB_A_PLANE + ((ks * 2 + lh) * RGB_PIX + wm * 64 + mi * 32 + l31) * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        b[ni][p] = *reinterpret_cast<const uint4*>(Bb + p * RGB_B_PLANE + ((ks * 2 + lh) * 128 + wn * 64 + ni * 32 + l31) * 16);
    }
#pragma unroll
    for (int sum = 1; sum >= 0; --sum)
#pragma unroll
      for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16<true>(a[mi][pa], b[ni][sum - pa], acc[mi][ni]);   // lane = channel, registers = pixels
  }

  }

  // lane = channel: every store instruction writes whole 128-byte lines (32 consecutive channels of one pixel per half-wave) -- the call
  // is bound by writing y, and 16-byte pieces of four different lines per lane (the transposed layout of the halo kernels) ran at 2 TB/s
  const float unscale = 1.f / (cx * cw);                             // exact: powers of two
  float st0[2] = {0.f, 0.f}, st1[2] = {0.f, 0.f};                    // per-channel sum / sum of squares / extremes of what is written
  float smn[2] = {INFINITY, INFINITY}, smx[2] = {-INFINITY, -INFINITY};   // (batch-norm statistics; the extremes give max|BN(y)| in advance)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wn * 64 + ni * 32 + l31;
    const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float* yp = y + (((long)img * Ho + h0 + 2 * wm + mi) * Wo + w0) * Cout + n;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float v = ladder_act_fn(acc[mi][ni][e] * unscale + bv, act);
        if (n < Cout) yp[(long)px * Cout] = v;
        st0[ni] += v;
        st1[ni] += v * v;
        smn[ni] = fminf(smn[ni], v);
        smx[ni] = fmaxf(smx[ni], v);
      }
    }
  }
  if (stats_part != nullptr) {
    // lane = channel makes the column sums local: 32 pixels per lane, the other half-wave holds the other 32 of this wavefront's 64,
    // the four wm wavefronts the rest of the patch; fixed order throughout.  partials[tile_m][4][Cout] = sum | sum of squares | min | max
    float* sred = reinterpret_cast<float*>(lds);                     // [which 4][wm 4][128 channels] (the plane images are dead)
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const float a0 = st0[ni] + __shfl_xor(st0[ni], 32, 64), a1 = st1[ni] + __shfl_xor(st1[ni], 32, 64);
      const float a2 = fminf(smn[ni], __shfl_xor(smn[ni], 32, 64)), a3 = fmaxf(smx[ni], __shfl_xor(smx[ni], 32, 64));
      if (lh == 0) {
        const int cc = wn * 64 + ni * 32 + l31;
        sred[(0 * 4 + wm) * 128 + cc] = a0;
        sred[(1 * 4 + wm) * 128 + cc] = a1;
        sred[(2 * 4 + wm) * 128 + cc] = a2;
        sred[(3 * 4 + wm) * 128 + cc] = a3;
      }
    }
    __syncthreads();
    {
      const int which = tid >> 7, c = tid & 127;                    // 512 threads = 4 statistics x 128 channels
      const float v0 = sred[(which * 4 + 0) * 128 + c], v1 = sred[(which * 4 + 1) * 128 + c], v2 = sred[(which * 4 + 2) * 128 + c],
                  v3 = sred[(which * 4 + 3) * 128 + c];
      const float t = which < 2 ? (v0 + v1) + (v2 + v3) : (which == 2 ? fminf(fminf(v0, v1), fminf(v2, v3)) : fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
      if (n0 + c < Cout) stats_part[((size_t)mt * 4 + which) * Cout + n0 + c] = t;
    }
  }
}

// ---- filter gradient ------------------------------------------------------------------------------------------------------------
// Workgroup = 4 wavefronts, walks a run of 2x32-pixel output patches (64 pixels = 4 MFMA K-steps of 16); wavefront w owns the 32 output
// channels [32w, 32w + 32) of the 128-channel slab (one accumulator tile, no cross-wavefront reduction).  Per patch:
// the 5 x 65 x 3 input patch and the 64 x 128 dY tile (prefetched into registers during the previous patch) are written to LDS -- dY as
// two fp16 planes in [32-channel block][pixel][32 ch x 16 bit] rows of 64 bytes, the im2col matrix [pixel][32 taps x 16 bit] likewise
// -- and every wavefront multiplies A = im2col^T (rows = tap index k) with its B = dY block, fragments through the transposing LDS
// read.  The scales are TENSOR-wide here (absmax records of x and dy): one accumulator sums over many patches.
constexpr int RW_PH = 2, RW_PW = 32, RW_PIX = RW_PH * RW_PW;
constexpr int RW_XH = 2 * RW_PH + 1, RW_XW = 2 * RW_PW + 1, RW_PATCH = RW_XH * RW_XW * 3;       // 5 x 65 x 3 = 975 floats
constexpr int RW_THREADS = 256;
constexpr int RW_ABLK = RW_PIX * 64 + 64, RW_DBLK = RW_PIX * 64 + 64;                          // bytes per 32-channel block image
constexpr int RW_APLANE = RW_ABLK, RW_DPLANE = 4 * RW_DBLK;
constexpr int RW_XU = (RW_PATCH + RW_THREADS - 1) / RW_THREADS;                                // 4 patch floats per thread
constexpr int RW_DU = RW_PIX * 32 / RW_THREADS;                                                // 8 dY float4 units per thread

typedef short rgb_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 rgb_tr_frag(const unsigned char* p) {
  const rgb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rgb_s16x4*)(p));
  const rgb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rgb_s16x4*)(p + 4 * 64));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ float rgb_frag_sum(const uint4 f) {      // sum of the 8 fp16 values of a fragment, in fp32
  const uint32_t w[4] = {f.x, f.y, f.z, f.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = __builtin_convertvector(__builtin_bit_cast(f16x2, w[i]), f32x2);
    s += v.x + v.y;
  }
  return s;
}

__global__ __launch_bounds__(RW_THREADS, 3) void conv_rgb_s2_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                         float* __restrict__ part, float* __restrict__ bias_part,
                                                                         const int N, const int H, const int W, const int Cout,
                                                                         const int tiles_co, const int patches_per_split,
                                                                         const float* __restrict__ xamax, const float* __restrict__ damax) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * RW_APLANE + 2 * RW_DPLANE];
  __shared__ __attribute__((aligned(16))) float patch[RW_PATCH + 1];
  unsigned char* const Ab = lds;
  unsigned char* const Db = lds + 2 * RW_APLANE;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15;
  const int split = blockIdx.x / tiles_co, co0 = (blockIdx.x - split * tiles_co) * 128;
  const int Ho = H / 2, Wo = W / 2, WP = Wo / RW_PW, HP = Ho / RW_PH;
  const int q_total = N * HP * WP;
  const int q0 = split * patches_per_split, q1 = min(q_total, q0 + patches_per_split);
  const float cx = scale_from_absmax(amax_load(xamax)), cd = scale_from_absmax(amax_load(damax));
  const int frag_lane = (8 * lh + (l15 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (l15 & 3)) * 2;

  f32x16 acc;                                                  // wavefront wv owns output channels [32 wv, 32 wv + 32) for all 4 K-steps
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float bsum = 0.f;

  float xr[RW_XU];
  float4 d0, d1, d2, d3, d4, d5, d6, d7;                       // (named: see convsplit.hip on arrays of vectors captured by lambdas)
  auto load_patch = [&](int q) {
    const int img = q / (HP * WP), rem = q - img * (HP * WP), hp = rem / WP;
    const int h0 = hp * RW_PH, w0 = (rem - hp * WP) * RW_PW;
#pragma unroll
    for (int i = 0; i < RW_XU; ++i) {
      const int v = tid + i * RW_THREADS;
      const int r = v / (RW_XW * 3), c3 = v - r * (RW_XW * 3);
      const int hi = 2 * h0 + r, wi3 = 2 * w0 * 3 + c3;
      xr[i] = (v < RW_PATCH && hi < H && wi3 < W * 3) ? x[((long)img * H + hi) * W * 3 + wi3] : 0.f;
    }
    const float* dbase = dy + (((long)img * Ho + h0) * Wo + w0) * Cout + co0;
#define RW_DSRC(i_) [&]() -> float4 {                                                                                       \
      const int u = tid + (i_) * RW_THREADS, px = u >> 5, q4 = u & 31;                                                        \
      if (co0 + q4 * 4 >= Cout) return make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
      return *reinterpret_cast<const float4*>(dbase + ((long)(px >> 5) * Wo + (px & 31)) * Cout + q4 * 4); }()
    d0 = RW_DSRC(0); d1 = RW_DSRC(1); d2 = RW_DSRC(2); d3 = RW_DSRC(3); d4 = RW_DSRC(4); d5 = RW_DSRC(5); d6 = RW_DSRC(6); d7 = RW_DSRC(7);
#undef RW_DSRC
  };
  auto store_d = [&](int i, float4 v) {
    const int u = tid + i * RW_THREADS, px = u >> 5, q4 = u & 31;
    uint2 pl[2];
    split4<2, true>(make_float4(v.x * cd, v.y * cd, v.z * cd, v.w * cd), pl);
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<uint2*>(Db + p * RW_DPLANE + (q4 >> 3) * RW_DBLK + px * 64 + (q4 & 7) * 8) = pl[p];
  };

  if (q0 < q1) load_patch(q0);
  for (int q = q0; q < q1; ++q) {
    // registers -> LDS: fp32 input patch, dY planes
#pragma unroll
    for (int i = 0; i < RW_XU; ++i)
      if (tid + i * RW_THREADS < RW_PATCH) patch[tid + i * RW_THREADS] = xr[i];
    store_d(0, d0); store_d(1, d1); store_d(2, d2); store_d(3, d3); store_d(4, d4); store_d(5, d5); store_d(6, d6); store_d(7, d7);
    __syncthreads();
    if (tid < 2 * RW_PIX) {                                     // im2col rows: thread = (pixel, K half)
      const int p = tid & (RW_PIX - 1), kh = tid >> 6, pr = p >> 5, pc = p & 31;
      float av[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = kh * 16 + j;
        const int tap = k / 3, ci = k - tap * 3, r = tap / 3, s = tap - r * 3;
        av[j] = k < 27 ? patch[((2 * pr + r) * RW_XW + 2 * pc + s) * 3 + ci] * cx : 0.f;
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        uint2 pl[2];
        split4<2, true>(make_float4(av[4 * o], av[4 * o + 1], av[4 * o + 2], av[4 * o + 3]), pl);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) *reinterpret_cast<uint2*>(Ab + pp * RW_APLANE + p * 64 + kh * 32 + o * 8) = pl[pp];
      }
    }
    __syncthreads();
    if (q + 1 < q1) load_patch(q + 1);                          // in flight across the MFMAs and the next patch's barriers
#pragma unroll
    for (int ks = 0; ks < RW_PIX / 16; ++ks) {
      const int pb = ks * 16;
      const uint4 a0 = rgb_tr_frag(Ab + pb * 64 + frag_lane), a1 = rgb_tr_frag(Ab + RW_APLANE + pb * 64 + frag_lane);
      const uint4 b0 = rgb_tr_frag(Db + wv * RW_DBLK + pb * 64 + frag_lane), b1 = rgb_tr_frag(Db + RW_DPLANE + wv * RW_DBLK + pb * 64 + frag_lane);
      acc = mfma16<true>(a1, b0, acc);
      acc = mfma16<true>(a0, b1, acc);
      acc = mfma16<true>(a0, b0, acc);
      bsum += rgb_frag_sum(b1) + rgb_frag_sum(b0);
    }
    __syncthreads();
  }

  const float unscale = 1.f / (cx * cd);
  float* o = part + (size_t)split * 27 * Cout;
  const int n = co0 + wv * 32 + l31;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = (e & 3) + 8 * (e >> 2) + 4 * lh;
    if (k < 27 && n < Cout) o[(size_t)k * Cout + n] = acc[e] * unscale;
  }
  bsum += __shfl_down(bsum, 32, 64);                           // pixels 0-7 + 8-15 of every K-step
  if (bias_part != nullptr && lh == 0 && n < Cout) bias_part[(size_t)split * Cout + n] = bsum * (1.f / cd);
}

// Strict-fp32 filter gradient (round 4): the same patch walk (2 x 32 output pixels per step, wavefront wv owns channels [32 wv, 32 wv + 32) of the
// 128-channel slab), fp32 throughout.  A = im2col^T (rows = tap index k < 27 of 32, reduction index = pixel) is read from the fp32 im2col image
// A[pixel][33] in LDS (lane = tap: consecutive words, no transposing read needed at 32 bits); B = dY is read ONCE, by the one wavefront that owns
// its channel block, so its fragments go from global memory straight into registers (lane = channel: 128-byte segments) a whole patch ahead.
// 32 K-steps of v_mfma_f32_32x32x2_f32 per patch and wavefront; the call is bound by reading dY (268 MB at batch 128).
constexpr int RW_LDA32 = 33;
__global__ __launch_bounds__(RW_THREADS, 4) void conv_rgb_s2_wgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                             float* __restrict__ part, float* __restrict__ bias_part,
                                                                             const int N, const int H, const int W, const int Cout,
                                                                             const int tiles_co, const int patches_per_split) {
  __shared__ float A32[RW_PIX * RW_LDA32];
  __shared__ float patch[RW_PATCH + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x / tiles_co, co0 = (blockIdx.x - split * tiles_co) * 128;
  const int Ho = H / 2, Wo = W / 2, WP = Wo / RW_PW, HP = Ho / RW_PH;
  const int q_total = N * HP * WP;
  const int q0 = split * patches_per_split, q1 = min(q_total, q0 + patches_per_split);
  const int co = co0 + wv * 32 + l31;
  const bool co_ok = co < Cout;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float bsum = 0.f;
  float xr[RW_XU];                                             // the NEXT patch's input floats of this thread (one patch ahead)
  auto decode = [&](int q, int& img, int& h0, int& w0) {
    img = q / (HP * WP);
    const int rem = q - img * (HP * WP), hp = rem / WP;
    h0 = hp * RW_PH;
    w0 = (rem - hp * WP) * RW_PW;
  };
  auto load_x = [&](int q) {
    int img, h0, w0;
    decode(q, img, h0, w0);
#pragma unroll
    for (int i = 0; i < RW_XU; ++i) {
      const int v = tid + i * RW_THREADS;
      const int r = v / (RW_XW * 3), c3 = v - r * (RW_XW * 3);
      const int hi = 2 * h0 + r, wi3 = 2 * w0 * 3 + c3;
      xr[i] = (v < RW_PATCH && hi < H && wi3 < W * 3) ? x[((long)img * H + hi) * W * 3 + wi3] : 0.f;
    }
  };

  if (q0 < q1) load_x(q0);
  for (int q = q0; q < q1; ++q) {
    // this patch's dY fragments: requested first, they travel while the input patch and the im2col image go through LDS (several workgroups
    // per CU -- 12 KB of LDS, ~90 registers -- cover the rest of the latency)
    float b[RW_PIX / 2];
    {
      int img, h0, w0;
      decode(q, img, h0, w0);
      // pixel 2 ks + lh of the patch (row ks / 16, column 2 (ks % 16) + lh): a wave-uniform base per k-step + ONE per-lane offset (channel and
      // pixel parity), so the 32 loads take scalar bases instead of 32 vector address pairs
      const float* du = dy + (((long)img * Ho + h0) * Wo + w0) * Cout;
      const unsigned loff = (unsigned)(co + lh * Cout);
#pragma unroll
      for (int ks = 0; ks < RW_PIX / 2; ++ks) {
        const float* pk = du + ((long)(ks >> 4) * Wo + 2 * (ks & 15)) * Cout;
        b[ks] = co_ok ? pk[loff] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < RW_XU; ++i)
      if (tid + i * RW_THREADS < RW_PATCH) patch[tid + i * RW_THREADS] = xr[i];
    __syncthreads();
    if (q + 1 < q1) load_x(q + 1);
    {
      const int p = tid & (RW_PIX - 1), kq = tid >> 6, pr = p >> 5, pc = p & 31;   // thread = (pixel, 8 taps)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kq * 8 + j;
        const int tap = k / 3, ci = k - tap * 3, r = tap / 3, sx = tap - r * 3;
        A32[p * RW_LDA32 + k] = k < 27 ? patch[((2 * pr + r) * RW_XW + 2 * pc + sx) * 3 + ci] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < RW_PIX / 2; ++ks) {
      const float a = A32[(2 * ks + lh) * RW_LDA32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[ks], acc, 0, 0, 0);
      bsum += b[ks];
    }
    __syncthreads();                                           // (the im2col image and the patch are rewritten next)
  }

  float* o = part + (size_t)split * 27 * Cout;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = (e & 3) + 8 * (e >> 2) + 4 * lh;
    if (k < 27 && co_ok) o[(size_t)k * Cout + co] = acc[e];
  }
  if (bias_part != nullptr) {
    const float v = bsum + __shfl_down(bsum, 32, 64);          // even + odd pixels of the k-step pairs
    if (lh == 0 && co_ok) bias_part[(size_t)split * Cout + co] = v;
  }
}

struct RgbWgradPlan { int tiles_co, splits, pps; };
RgbWgradPlan plan_rgb_wgrad(int N, int H, int W, int Cout) {
  RgbWgradPlan p;
  p.tiles_co = (Cout + 127) / 128;
  const long q_total = (long)N * (H / 2 / RW_PH) * (W / 2 / RW_PW);
  long s = (3 * 256L) / p.tiles_co;                            // three workgroups per CU
  if (s > q_total / 4) s = q_total / 4;
  if (s < 1) s = 1;
  p.pps = (int)((q_total + s - 1) / s);
  p.splits = (int)((q_total + p.pps - 1) / p.pps);
  return p;
}

bool rgb_s2_ok(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l) {
  return N > 0 && Cin == 3 && KH == 3 && KW == 3 && stride == 2 && pad_t == 0 && pad_l == 0 && (H % 2) == 0 && (W % 2) == 0 &&
         ((H / 2) % RGB_TH) == 0 && ((W / 2) % RGB_TW) == 0 && Cout >= 32 && (Cout % 4) == 0;
}

}  // namespace

extern "C" {

int ladder_conv_rgb_s2_eligible(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l) {
  return rgb_s2_ok(N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l) ? 1 : 0;
}

static int rgb_fwd_launch(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act, float* stats_part,
                          ladder_stream_t stream, bool f32 = false) {
  if (!rgb_s2_ok(N, H, W, 3, Cout, 3, 3, 2, 0, 0)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias))) return LADDER_E_ALIGN;
  const int tiles_n = (Cout + 127) / 128;
  const long tiles = (long)N * (H / 2 / RGB_TH) * (W / 2 / RGB_TW) * tiles_n;
  if (tiles >= (1L << 31)) return LADDER_E_SHAPE;
  if (f32)
    hipLaunchKernelGGL(conv_rgb_s2_fwd_kernel<true>, dim3((unsigned)tiles), dim3(RGB_THREADS), 0, stream, x, w, bias, y, N, H, W, Cout, act, tiles_n,
                       stats_part);
  else
    hipLaunchKernelGGL(conv_rgb_s2_fwd_kernel<false>, dim3((unsigned)tiles), dim3(RGB_THREADS), 0, stream, x, w, bias, y, N, H, W, Cout, act, tiles_n,
                       stats_part);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_conv_rgb_s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                           ladder_stream_t stream) {
  return rgb_fwd_launch(x, w, bias, y, N, H, W, Cout, act, nullptr, stream);
}

size_t ladder_conv_rgb_s2_fwd_bnstats_workspace_bytes(int N, int H, int W, int Cout) {
  if (!rgb_s2_ok(N, H, W, 3, Cout, 3, 3, 2, 0, 0)) return 0;
  return (size_t)N * (H / 2 / RGB_TH) * (W / 2 / RGB_TW) * 4 * Cout * sizeof(float);
}

// The forward call + the batch-norm statistics of its output (sums4 = the statistics record in its minmax form, 6 Cout floats: per-channel sum | sum of squares (doubles), min | max of y: what
// ladder_bn_fwd_stats_minmax would compute from a second pass over y), from per-patch column statistics of the epilogue.
int ladder_conv_rgb_s2_fwd_bnstats(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                                   float* sums, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  const size_t need = ladder_conv_rgb_s2_fwd_bnstats_workspace_bytes(N, H, W, Cout);
  if (need == 0 || sums == nullptr) return LADDER_E_SHAPE;
  if (ws == nullptr || ws_bytes < need) return LADDER_E_WORKSPACE;
  const int rc = rgb_fwd_launch(x, w, bias, y, N, H, W, Cout, act, (float*)ws, stream);
  if (rc != LADDER_OK) return rc;
  return ladder_bn_stats_minmax_from_partials((const float*)ws, N * (H / 2 / RGB_TH) * (W / 2 / RGB_TW), sums, Cout, stream);
}

// Strict-fp32 instantiations of the two forward calls (fp32 MFMA, no operand scaling): same arguments, same workspace.
int ladder_conv_rgb_s2_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                               ladder_stream_t stream) {
  return rgb_fwd_launch(x, w, bias, y, N, H, W, Cout, act, nullptr, stream, true);
}

int ladder_conv_rgb_s2_fwd_bnstats_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                                       float* sums, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  const size_t need = ladder_conv_rgb_s2_fwd_bnstats_workspace_bytes(N, H, W, Cout);
  if (need == 0 || sums == nullptr) return LADDER_E_SHAPE;
  if (ws == nullptr || ws_bytes < need) return LADDER_E_WORKSPACE;
  const int rc = rgb_fwd_launch(x, w, bias, y, N, H, W, Cout, act, (float*)ws, stream, true);
  if (rc != LADDER_OK) return rc;
  return ladder_bn_stats_minmax_from_partials((const float*)ws, N * (H / 2 / RGB_TH) * (W / 2 / RGB_TW), sums, Cout, stream);
}

// Strict-fp32 filter (and bias) gradient of the same layer: no absmax records, same workspace as ladder_conv_rgb_s2_bwd_filter.
int ladder_conv_rgb_s2_bwd_filter_f32(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cout, void* ws,
                                      size_t ws_bytes, ladder_stream_t stream) {
  if (!rgb_s2_ok(N, H, W, 3, Cout, 3, 3, 2, 0, 0)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(dw)) return LADDER_E_ALIGN;
  const RgbWgradPlan p = plan_rgb_wgrad(N, H, W, Cout);
  if (ws == nullptr || ws_bytes < (size_t)p.splits * (27 * (size_t)Cout + Cout) * sizeof(float)) return LADDER_E_WORKSPACE;
  float* part = (float*)ws;
  float* bias_part = db != nullptr ? part + (size_t)p.splits * 27 * Cout : nullptr;
  hipLaunchKernelGGL(conv_rgb_s2_wgrad_f32_kernel, dim3(p.tiles_co * p.splits), dim3(RW_THREADS), 0, stream, x, dy, part, bias_part, N, H, W, Cout,
                     p.tiles_co, p.pps);
  LADDER_CHECK_LAUNCH();
  int rc = ladder_reduce_splits(part, dw, p.splits, (size_t)27 * Cout, stream);
  if (rc != LADDER_OK) return rc;
  if (db != nullptr) rc = ladder_reduce_splits(bias_part, db, p.splits, (size_t)Cout, stream);
  return rc;
}

size_t ladder_conv_rgb_s2_bwd_filter_workspace_bytes(int N, int H, int W, int Cout) {
  if (!rgb_s2_ok(N, H, W, 3, Cout, 3, 3, 2, 0, 0)) return 0;
  const RgbWgradPlan p = plan_rgb_wgrad(N, H, W, Cout);
  return (size_t)p.splits * (27 * (size_t)Cout + Cout) * sizeof(float);
}

int ladder_conv_rgb_s2_bwd_filter(const float* x, const float* x_absmax, const float* dy, const float* dy_absmax, float* dw, float* db, int N,
                                  int H, int W, int Cout, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (!rgb_s2_ok(N, H, W, 3, Cout, 3, 3, 2, 0, 0) || x_absmax == nullptr || dy_absmax == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(dy) || !ladder_aligned16(dw)) return LADDER_E_ALIGN;
  if (ws == nullptr || ws_bytes < ladder_conv_rgb_s2_bwd_filter_workspace_bytes(N, H, W, Cout)) return LADDER_E_WORKSPACE;
  const RgbWgradPlan p = plan_rgb_wgrad(N, H, W, Cout);
  float* part = (float*)ws;
  float* bias_part = db != nullptr ? part + (size_t)p.splits * 27 * Cout : nullptr;
  hipLaunchKernelGGL(conv_rgb_s2_wgrad_kernel, dim3(p.tiles_co * p.splits), dim3(RW_THREADS), 0, stream, x, dy, part, bias_part, N, H, W, Cout,
                     p.tiles_co, p.pps, x_absmax, dy_absmax);
  LADDER_CHECK_LAUNCH();
  int rc = ladder_reduce_splits(part, dw, p.splits, (size_t)27 * Cout, stream);
  if (rc != LADDER_OK) return rc;
  if (db != nullptr) rc = ladder_reduce_splits(bias_part, db, p.splits, (size_t)Cout, stream);
  return rc;
}

}  // extern "C"
