// fp32 convolutions on the 16-bit matrix cores by operand splitting (gfx950, MI355X).
//
// gfx950 has no TF32/xf32 MFMA; its f32-input MFMA runs at the f32 vector rate (157 TFLOP/s) while the 16-bit-input
// v_mfma_f32_32x32x16_{bf16,f16} run 16x faster (2.5 PFLOP/s nominal; ~1.5-1.8 PFLOP/s sustained on random operands, where the chip
// is power-limited -- see DESIGN.md).  Every fp32 operand is split into NS 16-bit "planes"
//     x = x0 + x1 (+ x2) + residual,   x0 = rn16(x), x1 = rn16(x - x0), x2 = rn16(x - x0 - x1)
// (each subtraction is exact in fp32) and the product a*b is evaluated as the sum of the plane products whose weight is above
// the fp32 rounding level, all accumulated in ONE fp32 MFMA accumulator (16-bit x 16-bit products are exact in fp32):
//
//   LADDER_PREC_F16X3   2 fp16 planes (11 + 11 = 22 significand bits), a0b0 + a0b1 + a1b0; dropped a1b1 <= 2^-22 |ab|
//                       -> fp32-class result from 3 MFMAs.  fp16 has a 5-bit exponent, so each operand TENSOR is scaled by a
//                       power of two taken from its absolute maximum (|x| c < 2^14; ladder_absmax / the pack kernel): an element
//                       is then represented to max(2^-23 |x|, 2^-38 max|x|), i.e. exactly like fp32 down to 2^-16 of the
//                       tensor's largest magnitude and with an absolute floor of 4e-12 of it below; the result is un-scaled
//                       in the epilogue.
//   LADDER_PREC_BF16X6  3 bf16 planes (24 bits), a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1; dropped terms <= 2^-23 |ab|
//                       -> fp32-class result from 6 MFMAs, no scaling (bf16 has the fp32 exponent).
//   LADDER_PREC_BF16X3  2 bf16 planes (16 bits), 3 MFMAs; dropped terms <= 3 * 2^-17 |ab|  (between TF32 and fp32).
//
// Kernels here:
//   absmax_kernel               max |x| of a tensor into a device scalar (order-independent atomic max on the bit pattern).
//   filter_pack_kernel          once per weight update: HWIO fp32 filter bank (optionally flipped + transposed for backward-data)
//                               -> split planes laid out exactly as the conv kernel's LDS image (one contiguous 16-byte
//                               chunk per thread, no transposition in the hot loop).
//   conv3x3_halo_split_kernel   3x3 / stride 1 / SAME forward and backward-data of the large decoder maps: the 8x32-pixel x
//                               128-channel LDS-halo tiling of conv3x3_halo_kernel (igemm.hip) with the input halo split into
//                               planes while it is staged into LDS.
#include <cstdlib>
#include "split16.h"
#include "filterbank.h"
#include "convf32.h"

namespace {

__device__ __attribute__((aligned(16))) float gs_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// ---- absolute maximum ------------------------------------------------------------------------------------------------------
// Non-negative floats order like their bit patterns, so an unsigned atomic max is exact and order-independent (deterministic).
// 512 workgroups x 8 float4 loads in flight per thread; block maxima are folded into the 16-slot record (split16.h).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n4, size_t n, float* __restrict__ rec) {
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float m = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x4[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0)
    for (size_t t = n4 * 4 + threadIdx.x; t < n; t += 256) m = fmaxf(m, fabsf(x[t]));
  amax_commit_block(m, rec);
}

// The per-sample form (mode-1 record, common.h): blockIdx.y = sample, its blocks stride over the sample's elements.
__global__ __launch_bounds__(256) void absmax_samples_kernel(const float* __restrict__ x, size_t per_sample, float* __restrict__ rec) {
  const float* xs = x + (size_t)blockIdx.y * per_sample;
  const size_t n4 = per_sample / 4;                             // (per_sample % 4 == 0 and x 16-byte aligned: checked by the launcher)
  const float4* x4 = reinterpret_cast<const float4*>(xs);
  float m = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  amax_commit_block_sample(m, rec, blockIdx.y);
}

// ---- operand pre-splitting ------------------------------------------------------------------------------------------------------
// planes[p][i] = plane p of x[i] (16-bit, scaled for the fp16 format), plane-major.  The gather kernels re-read every input element
// once per filter tap and per output-channel tile (18-36 times): splitting it there each time cost 38 % of the kernel (VALU + LDS
// stores); one elementwise pass per tensor (4 B read + 4 B written per element) lets them stage planes by LDS-DMA.  16 zero bytes
// (+ a 16-byte header: word 0 != 0 when the planes carry one f16x3 scale per SAMPLE instead of one for the tensor)
// follow the last plane: the source of out-of-image taps (an offset INTO THE SAME BUFFER, so that the DMA source is selected by a
// conditional move on an integer -- a pointer select against another object compiles to exec-masked branches, and a partially
// masked global_load_lds takes its LDS base from the first ACTIVE lane).
template <int PREC>
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ x, const float* __restrict__ xamax,
                                                       uint4* __restrict__ planes, size_t n8, unsigned sample8) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  // sample8 = 16-byte units (8 elements) per sample when the consumer scales per sample (and the record carries per-sample bounds), else 0
  const float tmax = F16 ? amax_load(xamax) : 0.f;
  const bool ps = F16 && sample8 != 0u && amax_per_sample(xamax);
  float c = F16 ? scale_from_absmax(tmax) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    planes[(size_t)NS * n8] = make_uint4(0u, 0u, 0u, 0u);               // the 16 zero bytes behind the planes
    planes[(size_t)NS * n8 + 1] = make_uint4(ps ? 1u : 0u, 0u, 0u, 0u);  // header: one scale per sample / one for the tensor
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    if (ps) c = scale_for_sample(xamax, (int)(i / sample8), true, tmax);
    if (F16) {
      a = make_float4(a.x * c, a.y * c, a.z * c, a.w * c);
      b = make_float4(b.x * c, b.y * c, b.z * c, b.w * c);
    }
    uint2 lo[NS], hi[NS];
    split4<NS, F16>(a, lo);
    split4<NS, F16>(b, hi);
#pragma unroll
    for (int p = 0; p < NS; ++p) planes[(size_t)p * n8 + i] = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
  }
}

// ---- filter packing ----------------------------------------------------------------------------------------------------
// Logical filter F[tap][ci][co] (tap = r*KW+s < ntaps, ci < Cin, co < Cout):
//   transpose_flip = 0:  F[tap][ci][co] = w[tap][ci][co]               (w is the HWIO bank [ntaps][Cin][Cout])              forward
//   transpose_flip = 1:  F[tap][ci][co] = w[ntaps - 1 - tap][co][ci]   (w is the HWIO bank [ntaps][Cout][Cin] of the layer)  backward-data
//   transpose_flip = 2:  the four parity classes of a 3x3 / stride-2 backward-data as output-channel blocks (see filter_pack_element)
// Packed image (16-bit elements): P[tap][slab = ci/16][cot = co/128][plane][kg = (ci%16)/8][co % 128][ci % 8]; one (tap, slab, cot)
// block is the contiguous NS * 4096 bytes a workgroup stages per K-step.  The last 16 bytes of the buffer hold the filter's
// absolute maximum (f16x3 only; the kernels derive the power-of-two scale from it).
constexpr int SP_BN = 128;

template <int PREC>
__device__ __forceinline__ void filter_pack_element(const float* __restrict__ w, uint4* __restrict__ out, int ntaps, int Cin, int Cout,
                                                    int transpose_flip, int i, float c) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  const int cots = (Cout + SP_BN - 1) / SP_BN, nslabs = Cin / 16;
  const int col = i % SP_BN;
  int t = i / SP_BN;
  const int kg = t & 1;
  t >>= 1;
  const int cot = t % cots;
  t /= cots;
  const int slab = t % nslabs, tap = t / nslabs;
  const int co = cot * SP_BN + col, ci0 = slab * 16 + kg * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // (the logical bank F[tap][ci][co] of the orientation: filterbank.h -- shared with the fp32 banks of convf32.hip)
    const float f = co < Cout ? filter_bank_element(w, ntaps, Cin, Cout, transpose_flip, tap, ci0 + j, co) : 0.f;
    v[j] = f * c;
  }
  uint2 lo[NS], hi[NS];
  split4<NS, F16>(make_float4(v[0], v[1], v[2], v[3]), lo);
  split4<NS, F16>(make_float4(v[4], v[5], v[6], v[7]), hi);
  const size_t blk = ((size_t)(tap * nslabs + slab) * cots + cot) * (NS * 256);   // uint4 units
#pragma unroll
  for (int p = 0; p < NS; ++p) out[blk + (size_t)p * 256 + kg * 128 + col] = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
}

__global__ void record_times4_kernel(float* __restrict__ rec) {
  for (int k = threadIdx.x; k < AMAX_SLOTS * AMAX_STRIDE; k += 256) rec[k] *= 4.f;
}

template <int PREC>
__global__ __launch_bounds__(256) void filter_pack_kernel(const float* __restrict__ w, uint4* __restrict__ out, int ntaps, int Cin,
                                                          int Cout, int transpose_flip, int total, const float* __restrict__ wamax) {
  const int i = blockIdx.x * 256 + threadIdx.x;     // one thread per (tap, slab, cot, kg, co)
  const float c = Fmt<PREC>::F16 ? scale_from_absmax(amax_load(wamax)) : 1.f;   // (cooperative load: before the divergent exit)
  if (i >= total) return;
  filter_pack_element<PREC>(w, out, ntaps, Cin, Cout, transpose_flip, i, c);
}

// ---- all filter banks of a model in two launches -----------------------------------------------------------------------------------------
// After an optimiser step every convolution needs its split image again (forward orientation and the flipped / transposed one for
// backward-data): ~25 banks on the CelebA nets, each of them memset + absmax + pack = three 5 us launches on the lazy path.  Here a job
// table in device memory describes all of them: kernel A leaves 64 partial maxima per job in a scratch array (plain stores: nothing to
// clear; 64 blocks per bank, 16-byte loads), kernel B derives each job's scale from them, packs, and writes the bank's absmax record behind its payload.
constexpr int PK_PARTS = 64;
__global__ __launch_bounds__(256) void filter_absmax_multi_kernel(const ladder_pack_job_t* __restrict__ jobs, float* __restrict__ partial) {
  const ladder_pack_job_t j = jobs[blockIdx.y];
  const size_t n = (size_t)j.ntaps * (j.transpose_flip == 4 ? j.Cin / 4 : j.Cin) * ((j.transpose_flip == 2 || j.transpose_flip == 3) ? j.Cout / 4 : j.Cout);   // Cin % 16 == 0: a multiple of 4; banks are 16-byte aligned views of the flat store
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(j.w) & 15u) == 0) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (size_t)PK_PARTS * 256) {
      const float4 v = reinterpret_cast<const float4*>(j.w)[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)PK_PARTS * 256) m = fmaxf(m, fabsf(j.w[i]));
  }
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * PK_PARTS + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int PREC>
__global__ __launch_bounds__(256) void filter_pack_multi_kernel(const ladder_pack_job_t* __restrict__ jobs, int njobs,
                                                                const float* __restrict__ partial) {
  // job of this block: block_begin is ascending
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ladder_pack_job_t j = jobs[lo];
  const int total = j.ntaps * (j.Cin / 16) * ((j.Cout + SP_BN - 1) / SP_BN) * 2 * SP_BN;
  const int i = ((int)blockIdx.x - j.block_begin) * 256 + threadIdx.x;
  __shared__ float amax_s;
  if (threadIdx.x < 64) {                                            // PK_PARTS = 64 partial maxima of this bank: one per lane of wavefront 0
    const float m = wave_max(partial[lo * PK_PARTS + threadIdx.x]);
    if (threadIdx.x == 0) amax_s = m;
  }
  __syncthreads();
  const float amax = j.transpose_flip >= 3 ? 4.f * amax_s : amax_s;     // (upsample-fused bank: |W_eff| <= 2 x 2 x max|w|)
  float* rec = reinterpret_cast<float*>(static_cast<unsigned char*>(j.packed) + pack_payload_bytes_dev(j.ntaps, j.Cin, j.Cout, Fmt<PREC>::NS));
  if ((int)blockIdx.x == j.block_begin)                              // the bank's absmax record: slot 0 = the maximum, the rest 0
    for (int k = threadIdx.x; k < AMAX_SLOTS * AMAX_STRIDE; k += 256) rec[k] = k == 0 ? amax : 0.f;
  if (i >= total) return;
  filter_pack_element<PREC>(j.w, (uint4*)j.packed, j.ntaps, j.Cin, j.Cout, j.transpose_flip, i, Fmt<PREC>::F16 ? scale_from_absmax(amax) : 1.f);
}

// ---- 3x3 halo convolution on split operands -------------------------------------------------------------------------------
// Workgroup = 8 wavefronts = an 8x32-pixel output patch x 128 output channels; wavefront (wm, wn) owns patch rows {2wm, 2wm+1} x
// channels [64wn, 64wn+64) = 2x2 MFMA tiles of 32x32.  K-step = (16-channel input slab, filter tap) = ONE 32x32x16 MFMA K.
// LDS image of the input halo, per plane: [kg = channel octet 0/1][halo pixel 10x34][8 x 16 bit] -> the A fragment of a patch row
// at column shift s is 32 consecutive 16-byte slots (conflict-free ds_read_b128); the filter block per plane is
// [kg][128 channels][8 x 16 bit], same property.  The halo of the next slab is fetched as fp32 at tap 0 and split + written at
// tap 4; the packed filter block of the next K-step is double-buffered every tap.
constexpr int SP_H = 8, SP_W = 32, SP_PW = SP_W + 2, SP_PH = SP_H + 2, SP_NPIX = SP_PH * SP_PW;   // 340 halo pixels
constexpr int SP_THREADS = 512;
constexpr int SP_A_PLANE = 2 * SP_NPIX * 16;            // bytes per plane (10880)
constexpr int SP_B_PLANE = 2 * SP_BN * 16;              // bytes per plane (4096)
constexpr int SP_HALO_UNITS = SP_NPIX * 4;              // float4 units per slab (1360)
constexpr int SP_AU = (SP_HALO_UNITS + SP_THREADS - 1) / SP_THREADS;   // 3

// (two-plane formats: 128 registers and < 80 KB of LDS, so that TWO workgroups share a CU and one's staging / barrier phase hides behind
// the other's MFMAs; the 9 taps are fully unrolled: fragment addresses become immediates, ~4 VALU instructions per tap are left)
// PROJ: transposed accumulators (lane = pixel) for the fused projection; otherwise lane = channel and whole-line stores (see the 16-wave kernel)
template <int PREC, bool PROJ, int UPM = 0>
__global__ __launch_bounds__(SP_THREADS, Fmt<PREC>::NS == 2 ? 4 : 2) void conv3x3_halo_split_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                                          const int N, const int H, const int W, const int Cin,
                                                                          const int Cout, const int act, const int tiles_n,
                                                                          const float* __restrict__ xamax,
                                                                          const float* __restrict__ wamax, float* __restrict__ yamax,
                                                                          const float* __restrict__ pw, const float* __restrict__ pb,
                                                                          float* __restrict__ pout, const int pco,
                                                                          const unsigned long long tap_masks, const int s2_out) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  constexpr int A_BUF = NS * SP_A_PLANE, B_BUF = NS * SP_B_PLANE;
  constexpr int B_CHUNKS = NS * 256;                                       // 16-byte chunks per filter block
  constexpr int BU = (B_CHUNKS + SP_THREADS - 1) / SP_THREADS;             // 2 (NS=3: 1.5) / 1 (NS=2)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * A_BUF + 2 * B_BUF];
  unsigned char* const Abase = lds;
  unsigned char* const Bbase = lds + 2 * A_BUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / tiles_n, cot = class_tile(tile % tiles_n, mt, tiles_n, UPM != 2 && s2_out != 0), n0 = cot * SP_BN;   // (class tiles rotate with the patch: common.h)
  const int tw_n = W / SP_W, th_n = H / SP_H;
  const int img = mt / (tw_n * th_n), rem = mt - img * (tw_n * th_n);
  const int h0 = (rem / tw_n) * SP_H, w0 = (rem % tw_n) * SP_W;
  const int nslabs = Cin / 16;
  float cx = 1.f, unscale = 1.f;
  if (F16) {                                                           // the patch lies inside ONE image: that sample's own scale
    cx = uniform_f(scale_for_sample(xamax, img, amax_per_sample(xamax), amax_load(xamax)));
    unscale = uniform_f(1.f / (cx * scale_from_absmax(amax_load(wamax))));      // exact: powers of two
  }

  const float* hsrc[SP_AU];
  int hdst[SP_AU];
#pragma unroll
  for (int i = 0; i < SP_AU; ++i) {
    const int u = tid + i * SP_THREADS;
    const int pix = u >> 2, kq = u & 3;
    const int hr = pix / SP_PW, hc = pix - hr * SP_PW;
    const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
    // s2_out == 2 (upsample-fused bank): the halo outside the map is the CLAMPED pixel, negated above / left of the map (the zero padding
    // of the high-resolution convolution, exactly) and plain below / right of it (exact but for the last output row / column, which
    // ladder_conv3x3_up2_split recomputes); bit 0 of hdst carries the sign (the offsets are multiples of 8)
    const bool inside = hi >= 0 && hi < H && wi >= 0 && wi < W, up2 = UPM == 1;
    const bool ok = (u < SP_HALO_UNITS) && (inside || up2);
    const int hs = up2 ? min(max(hi, 0), H - 1) : hi, ws_ = up2 ? min(max(wi, 0), W - 1) : wi;
    // (s2_out == 3: x is the even-row / even-column sub-grid of an ALREADY upsampled [N, 2H, 2W, Cin] tensor -- up[2i][2j] = x[i][j] -- that a
    // training forward keeps for its backward pass)
    const int sm = (UPM == 1 && s2_out == 3) ? 2 : 1;
    hsrc[i] = ok ? x + (((long)img * (H * sm) + hs * sm) * (W * sm) + ws_ * sm) * Cin + kq * 4 : nullptr;
    hdst[i] = (((kq >> 1) * SP_NPIX + pix) * 16 + (kq & 1) * 8) | ((up2 && ((hi < 0) != (wi < 0))) ? 1 : 0);
  }
  const uint4* const bsrc = wp + (size_t)cot * B_CHUNKS + tid;             // + (tap*nslabs + slab) * tiles_n * B_CHUNKS

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ha[SP_AU];
  uint4 rb0, rb1;                                                         // (named: an array here is demoted to LDS by hipcc)
  const bool b_second = BU > 1 && tid + SP_THREADS < B_CHUNKS;
  auto load_halo = [&](int slab) {
#pragma unroll
    for (int i = 0; i < SP_AU; ++i) ha[i] = *reinterpret_cast<const float4*>(hsrc[i] != nullptr ? hsrc[i] + slab * 16 : gs_zero16);
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < SP_AU; ++i)
      if (tid + i * SP_THREADS < SP_HALO_UNITS) {
        uint2 pl[NS];
        float4 v = ha[i];
        const float cm = (UPM == 1 && (hdst[i] & 1)) ? -cx : cx;
        if (F16 || UPM == 1) v = make_float4(v.x * cm, v.y * cm, v.z * cm, v.w * cm);
        split4<NS, F16>(v, pl);
#pragma unroll
        for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(Abase + buf * A_BUF + p * SP_A_PLANE + (UPM == 1 ? (hdst[i] & ~1) : hdst[i])) = pl[p];
      }
  };
  auto load_b = [&](int slab, int tap) {
    const uint4* s = bsrc + (size_t)(tap * nslabs + slab) * tiles_n * B_CHUNKS;
    rb0 = s[0];
    if (b_second) rb1 = s[SP_THREADS];
  };
  auto store_b = [&](int buf) {
    *reinterpret_cast<uint4*>(Bbase + buf * B_BUF + tid * 16) = rb0;
    if (b_second) *reinterpret_cast<uint4*>(Bbase + buf * B_BUF + (tid + SP_THREADS) * 16) = rb1;
  };

  // taps this output-channel tile uses (all 9 for a convolution; the parity classes of a stride-2 backward-data bank use 1, 2 or 4:
  // ladder_conv3x3_s2_bwd_data_split) -- a masked tap costs its barrier, nothing else: no filter block staged, no fragment read, no MFMA
  const unsigned tmask = (unsigned)(tap_masks >> (9 * cot)) & 0x1ffu;
  load_halo(0);
  if (tmask & 1u) load_b(0, 0);
  store_halo(0);
  if (tmask & 1u) store_b(0);
  __syncthreads();
  int bbuf = 0;
  for (int slab = 0; slab < nslabs; ++slab) {
    const int hb = slab & 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const bool last = (slab + 1 == nslabs) && (tap == 8);
      const bool next_used = !last && ((tmask >> (tap == 8 ? 0 : tap + 1)) & 1u);
      if (next_used) load_b(tap == 8 ? slab + 1 : slab, tap == 8 ? 0 : tap + 1);
      if (tap == 0 && slab + 1 < nslabs) load_halo(slab + 1);
      if ((tmask >> tap) & 1u) {
      const int r = tap / 3, sft = tap - 3 * r;
      const unsigned char* Ab = Abase + hb * A_BUF + (lh * SP_NPIX + (2 * wm + r) * SP_PW + sft + l31) * 16;
      const unsigned char* Bb = Bbase + bbuf * B_BUF + (lh * SP_BN + wn * 64 + l31) * 16;
      uint4 a[2][NS], b[2][NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi][p] = *reinterpret_cast<const uint4*>(Ab + p * SP_A_PLANE + mi * SP_PW * 16);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[ni][p] = *reinterpret_cast<const uint4*>(Bb + p * SP_B_PLANE + ni * 32 * 16);
      }
      // plane products, smallest weights first: (pa, pb) with pa + pb < NS
#pragma unroll
      for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
        for (int pa = 0; pa <= sum; ++pa) {
          const int pb = sum - pa;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = PROJ ? mfma16<F16>(b[ni][pb], a[mi][pa], acc[mi][ni])       // D^T: lane = pixel, registers = channels
                                 : mfma16<F16>(a[mi][pa], b[ni][pb], acc[mi][ni]);      // lane = channel, registers = pixels
        }
      }
      if (tap == 4 && slab + 1 < nslabs) store_halo(hb ^ 1);
      if (next_used) store_b(bbuf ^ 1);
      __syncthreads();
      bbuf ^= 1;
    }
  }

  {  // ---- epilogue scope
  // Epilogue coordinates are RE-DERIVED from opaque copies of the thread / workgroup ids: computed before the main loop they were live
  // across it, and at 128 registers the compiler parked them in scratch (12-56 bytes per lane).
  int e_tid = threadIdx.x, e_bid = blockIdx.x;
  asm volatile("" : "+v"(e_tid));
  asm volatile("" : "+s"(e_bid));
  const int e_lane = e_tid & 63, e_wid = e_tid >> 6;
  const int l31 = e_lane & 31, lh = e_lane >> 5, wm = e_wid >> 1, wn = e_wid & 1;
  const int e_tile = xcd_remap(e_bid, gridDim.x);
  const int e_mt = e_tile / tiles_n, e_cot = class_tile(e_tile % tiles_n, e_mt, tiles_n, UPM != 2 && s2_out != 0), n0 = e_cot * SP_BN;
  const int e_twn = W / SP_W, e_thn = H / SP_H;
  const int img = e_mt / (e_twn * e_thn), e_rem = e_mt - img * (e_twn * e_thn);
  const int h0 = (e_rem / e_twn) * SP_H, w0 = (e_rem % e_twn) * SP_W;
  if (!PROJ) {
    float ymax = 0.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + l31;
      const float bv = (bias != nullptr && n < Cout) ? bias[s2_out ? n - n0 : n] : 0.f;    // (parity-class tiles share the layer's 128 channels)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        // s2_out: this output-channel tile is one PARITY CLASS (ph, pw) = (cot >> 1, cot & 1) of a stride-2 backward-data result: pixel
        // (h, w) of the class is dx[2h + ph, 2w + pw], channels = the tile's 128 (the 4 class tiles interleave into [N, 2H, 2W, 128])
        float* yp = s2_out ? y + (((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (e_cot >> 1)) * 2 * W + 2 * w0 + (e_cot & 1)) * SP_BN + (n - n0)
                           : y + (((long)img * H + h0 + 2 * wm + mi) * W + w0) * Cout + n;
        const long pstride = s2_out ? 2 * SP_BN : Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
          float v = acc[mi][ni][e];
          if (F16) v *= unscale;
          v = ladder_act_fn(v + bv, act);
          if (n < Cout) yp[(long)px * pstride] = v;
          ymax = fmaxf(ymax, n < Cout ? fabsf(v) : 0.f);
        }
      }
    }
    if (yamax != nullptr) amax_commit_block_sample(ymax, yamax, img);
    return;
  }
  // Transposed accumulators (the filter fragment is the MFMA's A operand): lane l31 = pixel of the patch row, register e -> channel
  // (e & 3) + 8 (e >> 2) + 4 lh of the 32-channel tile, i.e. four consecutive channels per register quad = one 16-byte store, and the
  // channel sum of the fused 1x1 projection below stays inside the lane.
  float ymax = 0.f;
  float pacc[2][4];                                          // fused projection: partial sums of this lane's channels, per patch row
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int o = 0; o < 4; ++o) pacc[mi][o] = 0.f;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      // (s2_out: this tile is one output-parity class of a [N, 2H, 2W, 128] map and owns all its 128 channels, see the lane = channel branch)
      const long opix = s2_out ? ((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (e_cot >> 1)) * 2 * W + 2 * (w0 + l31) + (e_cot & 1)
                               : ((long)img * H + h0 + 2 * wm + mi) * W + w0 + l31;
      const int coff = s2_out ? n0 : 0;
      float* yp = y != nullptr ? y + opix * (s2_out ? SP_BN : Cout) - coff : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * lh;
        if (n < Cout) {                                      // Cout % 4 == 0: a channel quad is inside or outside as a whole
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + n - coff);
          float4 v = make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
          if (F16) v = make_float4(v.x * unscale, v.y * unscale, v.z * unscale, v.w * unscale);
          v = make_float4(ladder_act_fn(v.x + bv.x, act), ladder_act_fn(v.y + bv.y, act), ladder_act_fn(v.z + bv.z, act),
                          ladder_act_fn(v.w + bv.w, act));
          if (yp != nullptr) *reinterpret_cast<float4*>(yp + n) = v;
          ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
          if (pout != nullptr) {                             // 1x1 projection: pw[Cout][pco], pco <= 4
            const float vv[4] = {v.x, v.y, v.z, v.w};
            if (pco == 3) {                                  // (the RGB output conv) rows n..n+3 = 12 consecutive floats, 16-byte aligned
              const float4* q = reinterpret_cast<const float4*>(pw + (size_t)(n - coff) * 3);
              const float4 q0 = q[0], q1 = q[1], q2 = q[2];
              const float wq[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int o = 0; o < 3; ++o) pacc[mi][o] = fmaf(vv[c4], wq[c4 * 3 + o], pacc[mi][o]);
            } else {
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4)
                for (int o = 0; o < pco; ++o) pacc[mi][o] = fmaf(vv[c4], pw[(size_t)(n - coff + c4) * pco + o], pacc[mi][o]);
            }
          }
        }
      }
    }
  }
  if (yamax != nullptr) amax_commit_block_sample(ymax, yamax, img);     // the output's absolute maximum for the next split contraction
  if (pout != nullptr) {
    // combine the two half-waves (lh) in registers, the two channel halves (wn) through LDS (free after the main loop), fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);             // [wm 4][mi 2][pixel 32][4]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int o = 0; o < 4; ++o) pacc[mi][o] += __shfl_xor(pacc[mi][o], 32, 64);
    if (wn == 1 && lh == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        *reinterpret_cast<float4*>(red + (((wm * 2 + mi) * 32 + l31) * 4)) = make_float4(pacc[mi][0], pacc[mi][1], pacc[mi][2], pacc[mi][3]);
    }
    __syncthreads();
    if (wn == 0 && lh == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float4 other = *reinterpret_cast<const float4*>(red + (((wm * 2 + mi) * 32 + l31) * 4));
        const float t[4] = {pacc[mi][0] + other.x, pacc[mi][1] + other.y, pacc[mi][2] + other.z, pacc[mi][3] + other.w};
        const long opix = s2_out ? ((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (e_cot >> 1)) * 2 * W + 2 * (w0 + l31) + (e_cot & 1)
                                 : ((long)img * H + h0 + 2 * wm + mi) * W + w0 + l31;
        float* op = pout + opix * pco;
        for (int o = 0; o < pco; ++o) op[o] = t[o] + (pb != nullptr ? pb[o] : 0.f);
      }
    }
  }
  }  // epilogue scope
}

// ---- the same kernel with a 16x32-pixel patch per workgroup (two-plane formats) ----------------------------------------------------
// 16 wavefronts (one workgroup per CU instead of two of 8): the filter block a tap needs is fetched ONCE for 512 pixels instead of once
// per 256 -- the L2 -> LDS filter traffic (590 KB per workgroup at 128 -> 128 channels, 4.8 GB per launch) was the largest single item
// of the 8-wave kernel's non-MFMA time (ablation: +13..15 % without it) -- and the halo overhead drops from 340/256 to 612/512.  The
// filter blocks of a whole filter ROW (3 taps, 24 KB) are staged at once: one barrier per 3 taps instead of one per tap.
constexpr int F_H = 16, F_PH = F_H + 2, F_NPIX = F_PH * SP_PW;                 // 612 halo pixels
constexpr int F_THREADS = 1024;
constexpr int F_A_PLANE = 2 * F_NPIX * 16;                                     // bytes per plane (19584)
constexpr int F_HALO_UNITS = F_NPIX * 4;                                       // float4 units per slab (2448)
constexpr int F_AU = (F_HALO_UNITS + F_THREADS - 1) / F_THREADS;               // 3
// PROJ: transposed accumulators (lane = pixel; needed by the fused 1x1 projection).  Without it the accumulators are lane = channel and
// every store instruction writes whole 128-byte lines (32 consecutive channels of a pixel per half-wave): +2 ... +5 % on the layers at
// batch 128 against the 16-byte pieces of the transposed layout.
template <int PREC, bool PROJ, int UPM = 0>
__global__ __launch_bounds__(F_THREADS) void conv3x3_halo_split16_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                                          const int N, const int H, const int W, const int Cin,
                                                                          const int Cout, const int act, const int tiles_n,
                                                                          const float* __restrict__ xamax,
                                                                          const float* __restrict__ wamax, float* __restrict__ yamax,
                                                                          const float* __restrict__ pw, const float* __restrict__ pb,
                                                                          float* __restrict__ pout, const int pco,
                                                                          const unsigned long long tap_masks, const int s2_out) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  static_assert(NS == 2, "two-plane formats only");
  constexpr int A_BUF = NS * F_A_PLANE, B_BUF = NS * SP_B_PLANE, B_STAGE = 3 * B_BUF;
  constexpr int B_CHUNKS = NS * 256;                                       // 16-byte chunks per filter block
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * A_BUF + 2 * B_STAGE];
  unsigned char* const Abase = lds;
  unsigned char* const Bbase = lds + 2 * A_BUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / tiles_n, cot = class_tile(tile % tiles_n, mt, tiles_n, UPM != 2 && s2_out != 0), n0 = cot * SP_BN;   // (class tiles rotate with the patch: common.h)
  const int tw_n = W / SP_W, th_n = H / F_H;
  const int img = mt / (tw_n * th_n), rem = mt - img * (tw_n * th_n);
  const int h0 = (rem / tw_n) * F_H, w0 = (rem % tw_n) * SP_W;
  const int nslabs = Cin / 16;
  float cx = 1.f, unscale = 1.f;
  if (F16) {                                                           // the patch lies inside ONE image: that sample's own scale
    cx = uniform_f(scale_for_sample(xamax, img, amax_per_sample(xamax), amax_load(xamax)));
    unscale = uniform_f(1.f / (cx * scale_from_absmax(amax_load(wamax))));      // exact: powers of two
  }

  const float* hsrc[F_AU];
  int hdst[F_AU];
#pragma unroll
  for (int i = 0; i < F_AU; ++i) {
    const int u = tid + i * F_THREADS;
    const int pix = u >> 2, kq = u & 3;
    const int hr = pix / SP_PW, hc = pix - hr * SP_PW;
    const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
    // s2_out == 2 (upsample-fused bank): the halo outside the map is the CLAMPED pixel, negated above / left of the map (the zero padding
    // of the high-resolution convolution, exactly) and plain below / right of it (exact but for the last output row / column, which
    // ladder_conv3x3_up2_split recomputes); bit 0 of hdst carries the sign (the offsets are multiples of 8)
    const bool inside = hi >= 0 && hi < H && wi >= 0 && wi < W, up2 = UPM == 1;
    const bool ok = (u < F_HALO_UNITS) && (inside || up2);
    const int hs = up2 ? min(max(hi, 0), H - 1) : hi, ws_ = up2 ? min(max(wi, 0), W - 1) : wi;
    // (s2_out == 3: x is the even-row / even-column sub-grid of an ALREADY upsampled [N, 2H, 2W, Cin] tensor -- up[2i][2j] = x[i][j] -- that a
    // training forward keeps for its backward pass)
    // (s2_out == 4, backward-data of the upsample-fused pair: x = dy [N, 2H, 2W, Cin / 4]; the kernel walks its four pixel-parity classes as
    // four groups of input-channel slabs -- pixel (hi, wi) of class (a, b) is dy[2 hi + a][2 wi + b]; the class offset is added per slab)
    const int sm = (UPM == 2 || (UPM == 1 && s2_out == 3)) ? 2 : 1;
    const int cpp = (UPM == 2) ? (Cin >> 2) : Cin;            // channels per pixel of the tensor behind x
    hsrc[i] = ok ? x + (((long)img * (H * sm) + hs * sm) * (W * sm) + ws_ * sm) * cpp + kq * 4 : nullptr;
    hdst[i] = (((kq >> 1) * F_NPIX + pix) * 16 + (kq & 1) * 8) | ((up2 && ((hi < 0) != (wi < 0))) ? 1 : 0);
  }
  // filter stage = the blocks of taps 3r .. 3r+2: 1536 chunks of 16 bytes, thread tid takes chunk tid (tap 3r + tid/512) and, the first
  // half of the workgroup, chunk 1024 + tid (tap 3r + 2)
  const size_t tap_stride = (size_t)nslabs * tiles_n * B_CHUNKS;
  const uint4* const bsrc = wp + (size_t)cot * B_CHUNKS + (tid & 511) + (size_t)(tid >> 9) * tap_stride;   // + (3r*nslabs + slab) * tiles_n * B_CHUNKS

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // staging registers are the scarce resource (128 per lane at 16 wavefronts): ONE halo unit and ONE filter chunk are in flight at a
  // time -- halo round i of the next slab rides on filter row i of this one, the second filter chunk re-uses the first one's register
  float4 ha;
  uint4 rb0;
  const bool b_second = tid < 512;
  auto load_halo = [&](int slab, int i) {
    int soff = slab * 16;
    if (UPM == 2) {                                                    // slab -> (parity class, 16-channel slab of dy)
      const int spc = nslabs >> 2, cls = slab / spc;
      soff = ((cls >> 1) * 2 * W + (cls & 1)) * (Cin >> 2) + (slab - cls * spc) * 16;
    }
    ha = *reinterpret_cast<const float4*>(hsrc[i] != nullptr ? hsrc[i] + soff : gs_zero16);
  };
  auto store_halo = [&](int buf, int i) {
    if (tid + i * F_THREADS < F_HALO_UNITS) {
      uint2 pl[NS];
      float4 v = ha;
      const float cm = (UPM == 1 && (hdst[i] & 1)) ? -cx : cx;
      if (F16 || UPM == 1) v = make_float4(v.x * cm, v.y * cm, v.z * cm, v.w * cm);
      split4<NS, F16>(v, pl);
#pragma unroll
      for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(Abase + buf * A_BUF + p * F_A_PLANE + (UPM == 1 ? (hdst[i] & ~1) : hdst[i])) = pl[p];
    }
  };
  auto load_b = [&](int slab, int r, int part) {                          // part 0: chunk tid, part 1: chunk 1024 + tid (first half only)
    const uint4* s = bsrc + (size_t)(3 * r * nslabs + slab) * tiles_n * B_CHUNKS;
    const unsigned tm = (unsigned)(tap_masks >> (9 * ((UPM == 2) ? slab / (nslabs >> 2) : cot))) & 0x1ffu;
    if (part == 0) { if ((tm >> (3 * r + (tid >> 9))) & 1u) rb0 = s[0]; }
    else if (b_second && ((tm >> (3 * r + 2)) & 1u)) rb0 = s[2 * tap_stride];
  };
  auto store_b = [&](int buf, int part) {
    if (part == 0) *reinterpret_cast<uint4*>(Bbase + buf * B_STAGE + tid * 16) = rb0;
    else if (b_second) *reinterpret_cast<uint4*>(Bbase + buf * B_STAGE + (tid + F_THREADS) * 16) = rb0;
  };

  // (see the 8-wave kernel: taps this output-channel tile uses; a thread's filter chunk of a stage belongs to tap 3r + (tid >> 9) resp. 3r + 2)
  const unsigned tmask0 = (unsigned)(tap_masks >> (9 * cot)) & 0x1ffu;
#pragma unroll
  for (int i = 0; i < F_AU; ++i) {
    load_halo(0, i);
    store_halo(0, i);
  }
  load_b(0, 0, 0);
  store_b(0, 0);
  load_b(0, 0, 1);
  store_b(0, 1);
  __syncthreads();
  int bbuf = 0;
  for (int slab = 0; slab < nslabs; ++slab) {
    const int hb = slab & 1;
    // (s2_out == 4: the taps depend on the parity class of the INPUT slab, not on the output-channel tile)
    const unsigned tmask = (UPM == 2) ? (unsigned)(tap_masks >> (9 * (slab / (nslabs >> 2)))) & 0x1ffu : tmask0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool last = (slab + 1 == nslabs) && (r == 2);
      const int nslab = r == 2 ? slab + 1 : slab, nr = r == 2 ? 0 : r + 1;   // the filter stage being fetched
      if (slab + 1 < nslabs) load_halo(slab + 1, r);
#pragma unroll
      for (int sft = 0; sft < 3; ++sft) {
        if (!last && sft < 2) load_b(nslab, nr, sft);
        if ((tmask >> (3 * r + sft)) & 1u) {
        const unsigned char* Ab = Abase + hb * A_BUF + (lh * F_NPIX + (2 * wm + r) * SP_PW + sft + l31) * 16;
        const unsigned char* Bb = Bbase + bbuf * B_STAGE + sft * B_BUF + (lh * SP_BN + wn * 64 + l31) * 16;
        uint4 a[2][NS], b[2][NS];
#pragma unroll
        for (int p = 0; p < NS; ++p) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a[mi][p] = *reinterpret_cast<const uint4*>(Ab + p * F_A_PLANE + mi * SP_PW * 16);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) b[ni][p] = *reinterpret_cast<const uint4*>(Bb + p * SP_B_PLANE + ni * 32 * 16);
        }
        // plane products, smallest weights first: (pa, pb) with pa + pb < NS
#pragma unroll
        for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
          for (int pa = 0; pa <= sum; ++pa) {
            const int pb = sum - pa;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = PROJ ? mfma16<F16>(b[ni][pb], a[mi][pa], acc[mi][ni])     // D^T: lane = pixel, registers = channels
                                   : mfma16<F16>(a[mi][pa], b[ni][pb], acc[mi][ni]);    // lane = channel, registers = pixels
          }
        }
        if (!last && sft < 2) store_b(bbuf ^ 1, sft);
      }
      if (slab + 1 < nslabs) store_halo(hb ^ 1, r);
      __syncthreads();
      bbuf ^= 1;
    }
  }

  {  // ---- epilogue scope
  const int s2o = (UPM == 2) ? 0 : s2_out;               // (mode 4 writes the plain [N, H, W, Cout] layout)
  // Epilogue coordinates are RE-DERIVED from opaque copies of the thread / workgroup ids: computed before the main loop they were live
  // across it, and at 128 registers the compiler parked them in scratch (12-56 bytes per lane).
  int e_tid = threadIdx.x, e_bid = blockIdx.x;
  asm volatile("" : "+v"(e_tid));
  asm volatile("" : "+s"(e_bid));
  const int e_lane = e_tid & 63, e_wid = e_tid >> 6;
  const int l31 = e_lane & 31, lh = e_lane >> 5, wm = e_wid >> 1, wn = e_wid & 1;
  const int e_tile = xcd_remap(e_bid, gridDim.x);
  const int e_mt = e_tile / tiles_n, e_cot = class_tile(e_tile % tiles_n, e_mt, tiles_n, UPM != 2 && s2_out != 0), n0 = e_cot * SP_BN;
  const int e_twn = W / SP_W, e_thn = H / F_H;
  const int img = e_mt / (e_twn * e_thn), e_rem = e_mt - img * (e_twn * e_thn);
  const int h0 = (e_rem / e_twn) * F_H, w0 = (e_rem % e_twn) * SP_W;
  if (!PROJ) {
    float ymax = 0.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + l31;
      const float bv = (bias != nullptr && n < Cout) ? bias[s2o ? n - n0 : n] : 0.f;    // (parity-class tiles share the layer's 128 channels)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        // s2o: this output-channel tile is one PARITY CLASS (ph, pw) = (cot >> 1, cot & 1) of a stride-2 backward-data result: pixel
        // (h, w) of the class is dx[2h + ph, 2w + pw], channels = the tile's 128 (the 4 class tiles interleave into [N, 2H, 2W, 128])
        float* yp = s2o ? y + (((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (e_cot >> 1)) * 2 * W + 2 * w0 + (e_cot & 1)) * SP_BN + (n - n0)
                           : y + (((long)img * H + h0 + 2 * wm + mi) * W + w0) * Cout + n;
        const long pstride = s2o ? 2 * SP_BN : Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
          float v = acc[mi][ni][e];
          if (F16) v *= unscale;
          v = ladder_act_fn(v + bv, act);
          if (n < Cout) yp[(long)px * pstride] = v;
          ymax = fmaxf(ymax, n < Cout ? fabsf(v) : 0.f);
        }
      }
    }
    if (yamax != nullptr) amax_commit_block_sample(ymax, yamax, img);
    return;
  }
  // Transposed accumulators (the filter fragment is the MFMA's A operand): lane l31 = pixel of the patch row, register e -> channel
  // (e & 3) + 8 (e >> 2) + 4 lh of the 32-channel tile, i.e. four consecutive channels per register quad = one 16-byte store, and the
  // channel sum of the fused 1x1 projection below stays inside the lane.
  float ymax = 0.f;
  float pacc[2][4];                                          // fused projection: partial sums of this lane's channels, per patch row
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int o = 0; o < 4; ++o) pacc[mi][o] = 0.f;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      // (s2o: this tile is one output-parity class of a [N, 2H, 2W, 128] map and owns all its 128 channels, see the lane = channel branch)
      const long opix = s2o ? ((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (e_cot >> 1)) * 2 * W + 2 * (w0 + l31) + (e_cot & 1)
                               : ((long)img * H + h0 + 2 * wm + mi) * W + w0 + l31;
      const int coff = s2o ? n0 : 0;
      float* yp = y != nullptr ? y + opix * (s2o ? SP_BN : Cout) - coff : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * lh;
        if (n < Cout) {                                      // Cout % 4 == 0: a channel quad is inside or outside as a whole
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + n - coff);
          float4 v = make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
          if (F16) v = make_float4(v.x * unscale, v.y * unscale, v.z * unscale, v.w * unscale);
          v = make_float4(ladder_act_fn(v.x + bv.x, act), ladder_act_fn(v.y + bv.y, act), ladder_act_fn(v.z + bv.z, act),
                          ladder_act_fn(v.w + bv.w, act));
          if (yp != nullptr) *reinterpret_cast<float4*>(yp + n) = v;
          ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
          if (pout != nullptr) {                             // 1x1 projection: pw[Cout][pco], pco <= 4
            const float vv[4] = {v.x, v.y, v.z, v.w};
            if (pco == 3) {                                  // (the RGB output conv) rows n..n+3 = 12 consecutive floats, 16-byte aligned
              const float4* q = reinterpret_cast<const float4*>(pw + (size_t)(n - coff) * 3);
              const float4 q0 = q[0], q1 = q[1], q2 = q[2];
              const float wq[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int o = 0; o < 3; ++o) pacc[mi][o] = fmaf(vv[c4], wq[c4 * 3 + o], pacc[mi][o]);
            } else {
#pragma unroll
              for (int c4 = 0; c4 < 4; ++c4)
                for (int o = 0; o < pco; ++o) pacc[mi][o] = fmaf(vv[c4], pw[(size_t)(n - coff + c4) * pco + o], pacc[mi][o]);
            }
          }
        }
      }
    }
  }
  if (yamax != nullptr) amax_commit_block_sample(ymax, yamax, img);     // the output's absolute maximum for the next split contraction
  if (pout != nullptr) {
    // combine the two half-waves (lh) in registers, the two channel halves (wn) through LDS (free after the main loop), fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);             // [wm 8][mi 2][pixel 32][4]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int o = 0; o < 4; ++o) pacc[mi][o] += __shfl_xor(pacc[mi][o], 32, 64);
    if (wn == 1 && lh == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        *reinterpret_cast<float4*>(red + (((wm * 2 + mi) * 32 + l31) * 4)) = make_float4(pacc[mi][0], pacc[mi][1], pacc[mi][2], pacc[mi][3]);
    }
    __syncthreads();
    if (wn == 0 && lh == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float4 other = *reinterpret_cast<const float4*>(red + (((wm * 2 + mi) * 32 + l31) * 4));
        const float t[4] = {pacc[mi][0] + other.x, pacc[mi][1] + other.y, pacc[mi][2] + other.z, pacc[mi][3] + other.w};
        const long opix = s2o ? ((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (e_cot >> 1)) * 2 * W + 2 * (w0 + l31) + (e_cot & 1)
                                 : ((long)img * H + h0 + 2 * wm + mi) * W + w0 + l31;
        float* op = pout + opix * pco;
        for (int o = 0; o < pco; ++o) op[o] = t[o] + (pb != nullptr ? pb[o] : 0.f);
      }
    }
  }
  }  // epilogue scope
}

// ---- 3x3 filter gradient on split operands -----------------------------------------------------------------------------------
// dW[tap][ci][co] = sum_pixels x[pix + tap][ci] * dy[pix][co]: the reduction index of the MFMA is the PIXEL, so both fragments need
// 8 consecutive pixels of one channel per lane while NHWC memory (and the LDS image a coalesced staging pass writes) has the
// channels contiguous.  gfx950's transposing LDS read closes the gap: ds_read_b64_tr_b16 lets every 16-lane group read a
// [4 pixels][16 channels] block of a channel-contiguous image and hands each lane the 4 pixels of ITS channel, so
//   * the LDS images stay [32-channel block][pixel][32 channels x 16 bit] (64-byte pixel rows: the 4 rows x 64 B a half-wave
//     touches are one contiguous 256-byte bank row -> conflict-free), written by plain 8-byte stores of split float4s;
//   * a filter tap is a whole-pixel address offset into the halo image (no misaligned or shifted copies).
// One workgroup = 8 wavefronts owns a 64 ci x 128 co slab of dW for all 9 taps (72 MFMA tiles); wavefront (cb, nq) owns ALL 9 taps
// of a 32 ci x 32 co block (9 tiles, 144 accumulator registers): per 16-pixel K-step one dy fragment serves 27 MFMAs and the three
// shifted x fragments of a filter row 9 -- 20 fragment reads per 27 MFMAs (the first version, 12 wavefronts x 6 tiles as in
// wgrad3x3_halo_kernel of igemm.hip, needed 20 per 18: 983 KB of LDS reads per patch against 6912 MFMA cycles).  2 wavefronts per SIMD
// leave 256 registers per lane.  The workgroup walks 2x32-pixel patches (K = 4 x 16 pixels; 4x34 halo) DOWN a 32-pixel column of an
// image, double-buffered in LDS; partial sums over the patch split are reduced in a fixed order by the caller.
// Memory side (measured, conv7 at batch 64): walking along W with the slabs of one pixel split on different XCDs read 2.14 GB from HBM
// per launch (8 % L2 hits); the column walk (consecutive patches share two of four halo rows) + placing the slabs of a split on one
// XCD brought that to 1.10 GB = the operands once.  Time barely moved -- the kernel is bound by the per-unit ISSUE cost of staging
// (address arithmetic, split, LDS stores: no-staging ablation 433 TF), which the per-thread offset tables below cut by a third.
constexpr int WS_CI = 64, WS_CO = 128, WS_PH = 2, WS_PW = 32, WS_HW = WS_PW + 2, WS_HH = WS_PH + 2;
constexpr int WS_THREADS = 512;
constexpr int WS_XPIX = WS_HH * WS_HW, WS_DPIX = WS_PH * WS_PW;      // 136 halo pixels, 64 output pixels per patch
constexpr int WS_XBLK = WS_XPIX * 64 + 64, WS_DBLK = WS_DPIX * 64 + 64;   // bytes per 32-channel block (+64: blocks land on different banks)
constexpr int WS_XPLANE = 2 * WS_XBLK, WS_DPLANE = 4 * WS_DBLK;
constexpr int WS_XU = WS_XPIX * (WS_CI / 4), WS_DU = WS_DPIX * (WS_CO / 4);                     // float4 units per patch: 2176, 2048
constexpr int WS_ROUNDS = (WS_XU + WS_DU + WS_THREADS - 1) / WS_THREADS;                        // 9 staging rounds (the last one 1/4 full)
constexpr int WS_KSTEPS = WS_DPIX / 16;                                                          // 4 MFMA K-steps per patch
static_assert(WS_ROUNDS == 9 && WS_KSTEPS == 4, "staging schedule below is written for 2x32 patches");

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_tr_frag(const unsigned char* p) {
  // 8 consecutive pixels (rows of 64 B) of this lane's channel: two transposing reads of 4 pixels each
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 64));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}

// sum of the 8 16-bit values of a fragment register quad, in fp32
template <bool F16>
__device__ __forceinline__ float frag_sum(const uint4 f) {
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
__global__ __launch_bounds__(WS_THREADS, 2) void wgrad3x3_split_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       float* __restrict__ out, float* __restrict__ bias_part,
                                                                       const int N, const int H, const int W, const int Cin,
                                                                       const int Cout, const int tiles_ci, const int tiles_co,
                                                                       const int nsplits, const int patches_per_split,
                                                                       const float* __restrict__ xamax, const float* __restrict__ damax) {
  constexpr int NS = Fmt<PREC>::NS;
  constexpr bool F16 = Fmt<PREC>::F16;
  constexpr int BUF = NS * (WS_XPLANE + WS_DPLANE);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;            // 8 wavefronts
  const int l31 = lane & 31, lh = lane >> 5, l15 = lane & 15;
  // 1-D grid: the workgroups that walk the SAME pixel split (one per ci x co slab) get linear ids 8 apart -- same XCD, dispatched
  // together -- so that the x / dy rows they share are fetched from HBM once and hit in that XCD's L2 for the others
  const int slabs = tiles_ci * tiles_co;
  const int grp = blockIdx.x / (8 * slabs), within = blockIdx.x - grp * (8 * slabs);
  const int slab = within >> 3, split = grp * 8 + (within & 7);
  if (split >= nsplits) return;
  const int ci0 = (slab / tiles_co) * WS_CI, co0 = (slab % tiles_co) * WS_CO;
  const int WP = W / WS_PW, HP = H / WS_PH;
  const int q_total = N * HP * WP;
  const int q0 = split * patches_per_split, q1 = min(q_total, q0 + patches_per_split);
  const int cb = wv >> 2, nq = wv & 3;                 // ci half, co quarter
  float cx = 1.f, cd = 1.f;
  if (F16) {
    cx = scale_from_absmax(amax_load(xamax));
    cd = scale_from_absmax(amax_load(damax));
  }
  const int frag_lane = (8 * lh + (l15 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (l15 & 3)) * 2;
  const int a_off = cb * WS_XBLK + frag_lane;                             // + plane*WS_XPLANE + (halo pixel of the k-step and tap) * 64
  const int b_off = NS * WS_XPLANE + nq * WS_DBLK + frag_lane;            // + plane*WS_DPLANE + (pixel of the k-step) * 64
  const bool do_bias = (bias_part != nullptr) && (ci0 == 0) && cb == 0;   // every output channel once: the cb = 0 wavefronts

  f32x16 acc[9];                                       // [tap]
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float bsum = 0.f;

  float4 r0, r1, r2;                                   // one batch = three staging rounds, in flight across one K-step
  // Staging units: everything that does not depend on the patch is computed ONCE per thread -- the element offset of the unit from the
  // patch's origin pixel, its LDS offset, and which image borders would invalidate it (bit 0 top, 1 bottom, 2 left, 3 right halo).  Per
  // patch a unit costs: border test, one load off a wave-uniform base, four selects (the per-patch address arithmetic -- a division,
  // bounds tests and a 64-bit multiply-add chain per unit -- was a quarter of the kernel's time).
  int goff[WS_ROUNDS], loff[WS_ROUNDS];
  uint64_t bmask = 0;                                  // 4 bits per round
#pragma unroll
  for (int i = 0; i < WS_ROUNDS; ++i) {
    const int u = tid + i * WS_THREADS;
    if (u < WS_XU) {
      const int pix = u >> 4, q4 = u & 15;
      const int hr = pix / WS_HW, hc = pix - hr * WS_HW;
      goff[i] = ((hr - 1) * W + (hc - 1)) * Cin + q4 * 4;
      loff[i] = (q4 >> 3) * WS_XBLK + pix * 64 + (q4 & 7) * 8;
      bmask |= (uint64_t)((hr == 0 ? 1 : 0) | (hr == WS_HH - 1 ? 2 : 0) | (hc == 0 ? 4 : 0) | (hc == WS_HW - 1 ? 8 : 0)) << (4 * i);
    } else if (u < WS_XU + WS_DU) {
      const int v = u - WS_XU, pp = v >> 5, q4 = v & 31;
      goff[i] = ((pp >> 5) * W + (pp & 31)) * Cout + q4 * 4;
      loff[i] = NS * WS_XPLANE + (q4 >> 3) * WS_DBLK + pp * 64 + (q4 & 7) * 8;
      if (co0 + q4 * 4 >= Cout) bmask |= (uint64_t)15 << (4 * i);            // channel tail of the last co slab: never valid
    } else {
      goff[i] = 0;
      loff[i] = -1;                                    // idle slot of the last round
    }
  }
  int pn = q0 / (HP * WP), ph0, pw0;                   // the patch being staged (the next one), advanced incrementally
  {
    const int rem = q0 - pn * (HP * WP), wp = rem / HP;
    pw0 = wp * WS_PW;
    ph0 = (rem - wp * HP) * WS_PH;
  }
  // (patches are walked DOWN a 32-pixel column of an image: consecutive patches share two of their four halo rows -> L2 hits)
  auto next_patch = [&]() {
    ph0 += WS_PH;
    if (ph0 >= H) {
      ph0 = 0;
      pw0 += WS_PW;
      if (pw0 >= W) {
        pw0 = 0;
        ++pn;
      }
    }
  };
  const float* xb = x;                                  // wave-uniform bases of the staged patch: its origin pixel, first channel of the slab
  const float* db = dy;
  unsigned pmask = 0;
  auto set_patch = [&]() {
    const long pix0 = ((long)pn * H + ph0) * W + pw0;
    xb = x + pix0 * Cin + ci0;
    db = dy + pix0 * Cout + co0;
    pmask = (ph0 == 0 ? 1u : 0u) | (ph0 + WS_PH >= H ? 2u : 0u) | (pw0 == 0 ? 4u : 0u) | (pw0 + WS_PW >= W ? 8u : 0u);
  };
  auto load_unit = [&](int i) -> float4 {
    const int u = tid + i * WS_THREADS;
    const bool isx = u < WS_XU;                          // (compile-time for all rounds but the mixed one)
    // x units: invalid when one of their border bits meets the patch's; dy units carry 0 (valid) or 15 (channel tail)
    const unsigned m = (unsigned)(bmask >> (4 * i)) & 15u;
    const bool bad = isx ? ((m & pmask) != 0u) : (m == 15u);
    const int off = bad ? 0 : goff[i];
    float4 v = *reinterpret_cast<const float4*>((isx ? xb : db) + off);
    if (bad) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
  };
  auto store_unit = [&](int buf, int i, float4 v) {
    if (loff[i] < 0) return;
    const bool isx = (tid + i * WS_THREADS) < WS_XU;
    const float c = isx ? cx : cd;
    if (F16) v = make_float4(v.x * c, v.y * c, v.z * c, v.w * c);
    uint2 pl[NS];
    split4<NS, F16>(v, pl);
#pragma unroll
    for (int p = 0; p < NS; ++p) *reinterpret_cast<uint2*>(lds + buf * BUF + loff[i] + p * (isx ? WS_XPLANE : WS_DPLANE)) = pl[p];
  };
  auto mma_step = [&](int buf, int ks) {
    const int prow = ks >> 1, pcol = (ks & 1) * 16;                                   // 16 pixels of patch row prow starting at pcol
    const unsigned char* base = lds + buf * BUF;
    uint4 b0 = lds_tr_frag(base + b_off + (prow * WS_PW + pcol) * 64);
    uint4 b1 = lds_tr_frag(base + b_off + WS_DPLANE + (prow * WS_PW + pcol) * 64);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      uint4 a0[3], a1[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        a0[s] = lds_tr_frag(base + a_off + ((prow + r) * WS_HW + pcol + s) * 64);
        a1[s] = lds_tr_frag(base + a_off + WS_XPLANE + ((prow + r) * WS_HW + pcol + s) * 64);
      }
      // plane products, smallest weights first: lo x hi, hi x lo, hi x hi
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[r * 3 + s] = mfma16<F16>(a1[s], b0, acc[r * 3 + s]);
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[r * 3 + s] = mfma16<F16>(a0[s], b1, acc[r * 3 + s]);
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[r * 3 + s] = mfma16<F16>(a0[s], b0, acc[r * 3 + s]);
    }
    if (do_bias) bsum += frag_sum<F16>(b1) + frag_sum<F16>(b0);
  };
  static_assert(NS == 2, "two-plane formats only");

  set_patch();
  if (q0 < q1) {
#pragma unroll
    for (int i = 0; i < WS_ROUNDS; i += 3) {
      r0 = load_unit(i); r1 = load_unit(i + 1); r2 = load_unit(i + 2);
      store_unit(0, i, r0); store_unit(0, i + 1, r1); store_unit(0, i + 2, r2);
    }
  }
  __syncthreads();
  for (int q = q0; q < q1; ++q) {
    const int buf = (q - q0) & 1;
    const bool more = q + 1 < q1;
    next_patch();
    if (more) set_patch();
    // (K-steps unrolled by hand: the staging rounds must be compile-time indices into the per-thread offset tables)
    if (more) { r0 = load_unit(0); r1 = load_unit(1); r2 = load_unit(2); }
    mma_step(buf, 0);
    if (more) { store_unit(buf ^ 1, 0, r0); store_unit(buf ^ 1, 1, r1); store_unit(buf ^ 1, 2, r2); }
    if (more) { r0 = load_unit(3); r1 = load_unit(4); r2 = load_unit(5); }
    mma_step(buf, 1);
    if (more) { store_unit(buf ^ 1, 3, r0); store_unit(buf ^ 1, 4, r1); store_unit(buf ^ 1, 5, r2); }
    if (more) { r0 = load_unit(6); r1 = load_unit(7); r2 = load_unit(8); }
    mma_step(buf, 2);
    if (more) { store_unit(buf ^ 1, 6, r0); store_unit(buf ^ 1, 7, r1); store_unit(buf ^ 1, 8, r2); }
    mma_step(buf, 3);
    __syncthreads();
  }

  const float unscale = F16 ? 1.f / (cx * cd) : 1.f;
  const int n = co0 + nq * 32 + l31;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float* o = out + ((size_t)split * 9 + t) * Cin * Cout;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ci = ci0 + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (n < Cout) o[(size_t)ci * Cout + n] = acc[t][e] * unscale;
    }
  }
  if (do_bias) {
    const float v = (bsum + __shfl_down(bsum, 32, 64)) * (F16 ? 1.f / cd : 1.f);   // pixels 0-7 + 8-15 of every k-step
    if (lh == 0 && n < Cout) bias_part[(size_t)split * Cout + n] = v;
  }
}

struct WgradSplitPlan { bool ok; int tiles_ci, tiles_co, splits, pps; };
WgradSplitPlan plan_wgrad_split(int N, int H, int W, int Cin, int Cout) {
  WgradSplitPlan p{false, 0, 0, 1, 0};
  if (!(N > 0 && (Cin % WS_CI) == 0 && (Cout % 4) == 0 && Cout >= 64 && (W % WS_PW) == 0 && (H % WS_PH) == 0)) return p;
  const long q_total = (long)N * (H / WS_PH) * (W / WS_PW);
  if (q_total * WS_PH < 4096) return p;                        // small maps stay on the generic kernel
  p.tiles_ci = Cin / WS_CI;
  p.tiles_co = (Cout + WS_CO - 1) / WS_CO;
  const long pairs = (long)p.tiles_ci * p.tiles_co;
  long s = 256L / pairs;                                       // ONE round of the chip (one workgroup per CU): every further round doubles the
                                                               // partial-sum traffic for nothing (conv5: 360 / 340 / 309 / 292 TF at 1 / 2 / 3 / 4 rounds)
  if (s > q_total / 16) s = q_total / 16;
  if (s < 1) s = 1;
  p.pps = (int)((q_total + s - 1) / s);
  p.splits = (int)((q_total + p.pps - 1) / p.pps);
  p.ok = true;
  return p;
}

// ---- the last output row / column of the upsample-fused convolution (ladder_conv3x3_up2_edges) -----------------------------------------
// Output row 2H-1 sees up-rows 2H-2 and 2H-1 (both = x[H-1] upsampled along W: the resize clamps) and the zero padding below:
//   y[2H-1, X] = act(b + sum_s (w[0][s] + w[1][s]) . uL[X + s - 1]),   uL = 1-D legacy upsample of the last row of x, zero outside [0, 2W)
// and likewise the last column with (w[r][0] + w[r][1]) and the 1-D upsample vL of the last column of x: a 1x3 convolution over the line
// tensor uL [N, 1, 2W, Cin] and a 3x1 convolution over vL [N, 2H, 1, Cin] (SAME padding) on the fp32 matrix cores.  This kernel writes the
// two lines (float4 units) and the two summed filter banks.
__global__ __launch_bounds__(256) void up2_edge_operands_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ u_row,
                                                                float* __restrict__ v_col, float* __restrict__ w_row, float* __restrict__ w_col,
                                                                int N, int H, int W, int Cin, int C, int sm) {
  const int CV = Cin >> 2;                                        // sm = 2: x is the even sub-grid of an upsampled [N, 2H, 2W, Cin] tensor
  const long n_row = (long)N * 2 * W * CV, n_col = (long)N * 2 * H * CV, n_w = (long)3 * Cin * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_row + n_col + 2 * n_w; i += (long)gridDim.x * 256) {
    if (i < n_row + n_col) {
      const bool row = i < n_row;
      const long j = row ? i : i - n_row;
      const int L = row ? W : H;                                  // low-resolution length of the line
      const int cv = (int)(j % CV);
      const int q = (int)((j / CV) % (2 * L));                    // position on the upsampled line
      const int n = (int)(j / ((long)CV * 2 * L));
      const int lo = q >> 1, hi = min(lo + 1, L - 1);
      const float4* base = reinterpret_cast<const float4*>(row ? x + (((long)n * (H * sm) + (H - 1) * sm) * (W * sm)) * Cin
                                                               : x + (((long)n * (H * sm)) * (W * sm) + (W - 1) * sm) * Cin) + cv;
      const long stride = row ? (long)sm * CV : (long)sm * (W * sm) * CV;
      const float4 xl = base[lo * stride];
      float4 v = xl;
      if (q & 1) {                                                // the arithmetic of resize_fwd_kernel (lerp, weight 1/2)
        const float4 xh = base[hi * stride];
        v = make_float4(xl.x + (xh.x - xl.x) * 0.5f, xl.y + (xh.y - xl.y) * 0.5f, xl.z + (xh.z - xl.z) * 0.5f, xl.w + (xh.w - xl.w) * 0.5f);
      }
      reinterpret_cast<float4*>(row ? u_row : v_col)[j] = v;
    } else {
      const long j = i - n_row - n_col;
      const bool row = j < n_w;
      const long e = row ? j : j - n_w;
      const int co = (int)(e % C), ci = (int)((e / C) % Cin), t = (int)(e / ((long)C * Cin));
      // row bank [1][3][Cin][C]: taps (r = 0, s = t) + (r = 1, s = t); column bank [3][1][Cin][C]: taps (r = t, s = 0) + (r = t, s = 1)
      const float v0 = row ? w[(((long)0 * 3 + t) * Cin + ci) * C + co] : w[(((long)t * 3 + 0) * Cin + ci) * C + co];
      const float v1 = row ? w[(((long)1 * 3 + t) * Cin + ci) * C + co] : w[(((long)t * 3 + 1) * Cin + ci) * C + co];
      (row ? w_row : w_col)[e] = v0 + v1;
    }
  }
}

// e_row [N, 2W, C], e_col [N, 2H, C] (bias and activation applied by the GEMM) -> the edge pixels of y [N, 2H, 2W, C] and, optionally, of the
// fused 1x1 projection pout [N, 2H, 2W, pco]; one wavefront per pixel (lane = channel pair), grid (pixels / 4, N)
__global__ __launch_bounds__(256) void up2_edge_scatter_kernel(const float* __restrict__ e_row, const float* __restrict__ e_col, float* __restrict__ y,
                                                               const float* __restrict__ pw, const float* __restrict__ pb, float* __restrict__ pout,
                                                               int pco, int H, int W, int C, float* __restrict__ yamax) {
  const int n = blockIdx.y, lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);            // 0 .. 2W-1: last row; 2W .. 2W+2H-2: last column (its last pixel is the row's)
  const int OH = 2 * H, OW = 2 * W;
  float m = 0.f;
  if (pix < OW + OH - 1) {
    const bool row = pix < OW;
    const int oy = row ? OH - 1 : pix - OW, ox = row ? pix : OW - 1;
    const float* src = row ? e_row + ((long)n * OW + ox) * C : e_col + ((long)n * OH + oy) * C;
    const long o = ((long)n * OH + oy) * OW + ox;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < C; c += 64) {
      const float v = src[c];
      if (y != nullptr) y[o * C + c] = v;
      m = fmaxf(m, fabsf(v));
      if (pout != nullptr)
        for (int q = 0; q < pco; ++q) acc[q] = fmaf(v, pw[(long)c * pco + q], acc[q]);
    }
    if (pout != nullptr) {
      for (int q = 0; q < pco; ++q) {
        float a = acc[q];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) pout[o * pco + q] = a + (pb != nullptr ? pb[q] : 0.f);
      }
    }
  }
  if (yamax != nullptr) amax_commit_block_sample(m, yamax, n);
}

// ---- border lines of the fused low-resolution backward-data (ladder_conv3x3_up2_bwd_border) -------------------------------------------------
// d_up = the plain backward-data on a strip of dy next to an image border: n_up = 2 lines (first border: rows / columns 0, 1) or 3 lines (last
// border: 2L-3 .. 2L-1).  The border line of dx_lo is the resize transpose across the strip -- t = d0 + d1 / 2 resp. d0 / 2 + d1 + d2 -- followed by
// the 1-D legacy resize transpose ALONG the line: lo[q] = t[2q] + t[2q-1] / 2 (q > 0) + t[2q+1] x (1/2, or 1 at the clamped end).
// One thread = one (sample, position q, channel quad); grid (ceil(L * C / 4 / 256), N).
__global__ __launch_bounds__(256) void up2_bwd_border_kernel(const float* __restrict__ dup, float* __restrict__ dx, float* __restrict__ rec,
                                                             int H, int W, int C, int axis, int first) {
  const int n = blockIdx.y, CV = C >> 2;
  const int Lq = axis == 1 ? W : H, n_up = first ? 2 : 3;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float m = 0.f;
  if (i < Lq * CV) {
    const int cv = i % CV, q = i / CV;
    // element (line k, position X on the upsampled line) of the strip: rows strip [N][n_up][2W][C], columns strip [N][2H][n_up][C]
    auto at = [&](int k, int X) -> float4 {
      const size_t pix = axis == 1 ? ((size_t)n * n_up + k) * (2 * W) + X : ((size_t)n * (2 * H) + X) * n_up + k;
      return reinterpret_cast<const float4*>(dup)[pix * CV + cv];
    };
    auto tline = [&](int X) -> float4 {
      const float4 a = at(0, X), b = at(1, X);
      if (first) return make_float4(a.x + 0.5f * b.x, a.y + 0.5f * b.y, a.z + 0.5f * b.z, a.w + 0.5f * b.w);
      const float4 c = at(2, X);
      return make_float4(0.5f * a.x + b.x + c.x, 0.5f * a.y + b.y + c.y, 0.5f * a.z + b.z + c.z, 0.5f * a.w + b.w + c.w);
    };
    float4 v = tline(2 * q);
    if (q > 0) {
      const float4 u = tline(2 * q - 1);
      v = make_float4(v.x + 0.5f * u.x, v.y + 0.5f * u.y, v.z + 0.5f * u.z, v.w + 0.5f * u.w);
    }
    const float4 u = tline(2 * q + 1);
    const float wt = q < Lq - 1 ? 0.5f : 1.f;
    v = make_float4(v.x + wt * u.x, v.y + wt * u.y, v.z + wt * u.z, v.w + wt * u.w);
    const size_t opix = axis == 1 ? ((size_t)n * H + (first ? 0 : H - 1)) * W + q : ((size_t)n * H + q) * W + (first ? 0 : W - 1);
    reinterpret_cast<float4*>(dx)[opix * CV + cv] = v;
    m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (rec != nullptr) amax_commit_block_sample(m, rec, n);
}

// the 16-row patch variant: two-plane formats, enough patches for two rounds of the chip
bool split_halo16_ok(int N, int H, int W, int Cin, int Cout, int prec) {
  static const bool off = getenv("LADDER_DISABLE_HALO16") != nullptr;
  return !off && prec_planes(prec) == 2 && (H % F_H) == 0 && (long)N * (H / F_H) * (W / SP_W) * ((Cout + SP_BN - 1) / SP_BN) >= 512;
}

bool split_halo_ok(int N, int H, int W, int Cin, int Cout) {
  return N > 0 && (Cin % 16) == 0 && (Cout % 4) == 0 && Cout >= 64 && (W % SP_W) == 0 && (H % SP_H) == 0 &&
         (long)N * (H / SP_H) * (W / SP_W) * ((Cout + SP_BN - 1) / SP_BN) >= 512;
}

}  // namespace

extern "C" {

int ladder_absmax(const float* x, size_t n, float* out, ladder_stream_t stream) {
  if (n == 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x)) return LADDER_E_ALIGN;
  if (hipMemsetAsync(out, 0, LADDER_ABSMAX_FLOATS * sizeof(float), stream) != hipSuccess) return LADDER_E_LAUNCH;
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n4, n, out);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_absmax_samples(const float* x, int n_samples, size_t per_sample, float* out, ladder_stream_t stream) {
  if (n_samples <= 0 || per_sample == 0 || (per_sample % 4) != 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(out)) return LADDER_E_ALIGN;
  if (hipMemsetAsync(out, 0, LADDER_ABSMAX_FLOATS * sizeof(float), stream) != hipSuccess) return LADDER_E_LAUNCH;
  size_t bx = (per_sample / 4 + 256 * 4 - 1) / (256 * 4);
  const size_t cap = (size_t)(2048 / n_samples) > 0 ? (size_t)(2048 / n_samples) : 1;          // ~2048 workgroups in total
  bx = bx < 1 ? 1 : (bx > cap ? cap : bx);
  hipLaunchKernelGGL(absmax_samples_kernel, dim3((unsigned)bx, (unsigned)n_samples), dim3(256), 0, stream, x, per_sample, out);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

size_t ladder_presplit_bytes(size_t n, int prec) { return (prec_ok(prec) && n % 8 == 0) ? (size_t)prec_planes(prec) * n * 2 + 32 : 0; }

int ladder_presplit(const float* x, const float* x_absmax, void* planes, size_t n, int n_samples, int prec, ladder_stream_t stream) {
  if (n == 0 || (n % 8) != 0 || !prec_ok(prec) || n_samples < 0) return LADDER_E_SHAPE;
  unsigned sample8 = 0;                                          // per-sample scales: n_samples > 0 and whole 16-byte units per sample
  if (n_samples > 0) {
    if ((n % (size_t)n_samples) != 0 || ((n / (size_t)n_samples) % 8) != 0 || (n / (size_t)n_samples) / 8 > 0xffffffffull) return LADDER_E_SHAPE;
    sample8 = (unsigned)((n / (size_t)n_samples) / 8);
  }
  if (!ladder_aligned16(x) || !ladder_aligned16(planes)) return LADDER_E_ALIGN;
  if (prec == LADDER_PREC_F16X3 && x_absmax == nullptr) return LADDER_E_SHAPE;
  const size_t n8 = n / 8;
  size_t blocks = (n8 + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  const dim3 grid((unsigned)blocks), block(256);
  if (prec == LADDER_PREC_F16X3) hipLaunchKernelGGL(presplit_kernel<LADDER_PREC_F16X3>, grid, block, 0, stream, x, x_absmax, (uint4*)planes, n8, sample8);
  else if (prec == LADDER_PREC_BF16X6) hipLaunchKernelGGL(presplit_kernel<LADDER_PREC_BF16X6>, grid, block, 0, stream, x, x_absmax, (uint4*)planes, n8, sample8);
  else hipLaunchKernelGGL(presplit_kernel<LADDER_PREC_BF16X3>, grid, block, 0, stream, x, x_absmax, (uint4*)planes, n8, sample8);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

size_t ladder_filter_pack_split_bytes(int ntaps, int Cin, int Cout, int prec) {
  if (prec == LADDER_PREC_F32) return (ntaps > 0 && Cin > 0 && Cout > 0 && (Cin % 16) == 0) ? filter_pack_f32_bytes(ntaps, Cin, Cout) : 0;
  if (ntaps <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 16) != 0 || !prec_ok(prec)) return 0;
  return pack_payload_bytes(ntaps, Cin, Cout, prec) + LADDER_ABSMAX_FLOATS * sizeof(float);
}

int ladder_filter_pack_split(const float* w, void* packed, int ntaps, int Cin, int Cout, int transpose_flip, int prec,
                             ladder_stream_t stream) {
  if (prec == LADDER_PREC_F32) {                                 // strict fp32: the fp32 bank [ntaps][Cin][Cout] of the orientation (convf32.hip)
    if (ntaps <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 16) != 0 || transpose_flip < 0 || transpose_flip > 7) return LADDER_E_SHAPE;
    if (transpose_flip >= 6 && (ntaps != 1 || ((transpose_flip == 6 ? Cout : Cin) % 9) != 0)) return LADDER_E_SHAPE;
    if (!ladder_aligned16(packed) || !ladder_aligned16(w)) return LADDER_E_ALIGN;
    return filter_pack_f32(w, (float*)packed, ntaps, Cin, Cout, transpose_flip, stream);
  }
  if (ntaps <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 16) != 0 || !prec_ok(prec)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(packed) || !ladder_aligned16(w)) return LADDER_E_ALIGN;
  const long total_l = (long)ntaps * (Cin / 16) * ((Cout + SP_BN - 1) / SP_BN) * 2 * SP_BN;
  if (total_l >= (1L << 31)) return LADDER_E_SHAPE;
  const int total = (int)total_l;
  const dim3 grid((total + 255) / 256), block(256);
  float* wamax = reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + pack_payload_bytes(ntaps, Cin, Cout, prec));
  if (prec == LADDER_PREC_F16X3) {
    const int rc = ladder_absmax(w, (size_t)ntaps * (transpose_flip == 4 ? Cin / 4 : Cin) * ((transpose_flip == 2 || transpose_flip == 3) ? Cout / 4 : Cout), wamax, stream);
    if (rc != LADDER_OK) return rc;
    if (transpose_flip >= 3) hipLaunchKernelGGL(record_times4_kernel, dim3(1), dim3(256), 0, stream, wamax);   // |W_eff| <= 4 max|w|
    hipLaunchKernelGGL(filter_pack_kernel<LADDER_PREC_F16X3>, grid, block, 0, stream, w, (uint4*)packed, ntaps, Cin, Cout, transpose_flip, total, wamax);
  } else if (prec == LADDER_PREC_BF16X6) {
    hipLaunchKernelGGL(filter_pack_kernel<LADDER_PREC_BF16X6>, grid, block, 0, stream, w, (uint4*)packed, ntaps, Cin, Cout, transpose_flip, total, wamax);
  } else {
    hipLaunchKernelGGL(filter_pack_kernel<LADDER_PREC_BF16X3>, grid, block, 0, stream, w, (uint4*)packed, ntaps, Cin, Cout, transpose_flip, total, wamax);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_filter_pack_job_blocks(int ntaps, int Cin, int Cout) {
  if (ntaps <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 16) != 0) return 0;
  const long total = (long)ntaps * (Cin / 16) * ((Cout + SP_BN - 1) / SP_BN) * 2 * SP_BN;
  return total >= (1L << 31) ? 0 : (int)((total + 255) / 256);
}

size_t ladder_filter_pack_split_multi_scratch_bytes(int njobs) { return njobs > 0 ? (size_t)njobs * PK_PARTS * sizeof(float) : 0; }

int ladder_filter_pack_split_multi(const ladder_pack_job_t* jobs_dev, int njobs, int total_blocks, int prec, void* scratch, size_t scratch_bytes,
                                   ladder_stream_t stream) {
  if (prec == LADDER_PREC_F32) {                                 // (no scales: no scratch needed)
    if (jobs_dev == nullptr || njobs <= 0 || total_blocks <= 0) return LADDER_E_SHAPE;
    return filter_pack_f32_multi(jobs_dev, njobs, total_blocks, stream);
  }
  if (jobs_dev == nullptr || njobs <= 0 || total_blocks <= 0 || !prec_ok(prec)) return LADDER_E_SHAPE;
  if (scratch == nullptr || scratch_bytes < ladder_filter_pack_split_multi_scratch_bytes(njobs)) return LADDER_E_WORKSPACE;
  hipLaunchKernelGGL(filter_absmax_multi_kernel, dim3(PK_PARTS, njobs), dim3(256), 0, stream, jobs_dev, (float*)scratch);
  const dim3 grid(total_blocks), block(256);
  if (prec == LADDER_PREC_F16X3) hipLaunchKernelGGL(filter_pack_multi_kernel<LADDER_PREC_F16X3>, grid, block, 0, stream, jobs_dev, njobs, (const float*)scratch);
  else if (prec == LADDER_PREC_BF16X6) hipLaunchKernelGGL(filter_pack_multi_kernel<LADDER_PREC_BF16X6>, grid, block, 0, stream, jobs_dev, njobs, (const float*)scratch);
  else hipLaunchKernelGGL(filter_pack_multi_kernel<LADDER_PREC_BF16X3>, grid, block, 0, stream, jobs_dev, njobs, (const float*)scratch);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_conv3x3_split_eligible(int N, int H, int W, int Cin, int Cout) { return split_halo_ok(N, H, W, Cin, Cout) ? 1 : 0; }

static int conv3x3_split_launch(const float* x, const float* x_absmax, const void* packed, const float* bias, float* y, float* y_absmax,
                                const float* pw, const float* pb, float* pout, int pco, int N, int H, int W, int Cin, int Cout, int act,
                                int prec, ladder_stream_t stream, unsigned long long tap_masks = ~0ull, int s2_out = 0) {
  if (prec == LADDER_PREC_F32)       // strict fp32 (v_mfma_f32_32x32x2_f32): `packed` = the fp32 bank, no absmax records
    return conv3x3_f32_launch(x, (const float*)packed, bias, y, pw, pb, pout, pco, N, H, W, Cin, Cout, act, stream, tap_masks, s2_out);
  if (!split_halo_ok(N, H, W, Cin, Cout) || !prec_ok(prec)) return LADDER_E_SHAPE;
  if (pout != nullptr && (pw == nullptr || pco < 1 || pco > 4 || (Cout > SP_BN && s2_out < 2) || !ladder_aligned16(pw))) return LADDER_E_SHAPE;
  if (pout == nullptr && y == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(packed) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias)))
    return LADDER_E_ALIGN;
  if (prec == LADDER_PREC_F16X3 && x_absmax == nullptr) return LADDER_E_SHAPE;
  const int tiles_n = (Cout + SP_BN - 1) / SP_BN;
  const int tiles_m = N * (H / SP_H) * (W / SP_W);
  const dim3 grid(tiles_m * tiles_n), block(SP_THREADS);
  const float* wamax = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed) + pack_payload_bytes(9, Cin, Cout, prec));
  if (y_absmax != nullptr && hipMemsetAsync(y_absmax, 0, LADDER_ABSMAX_FLOATS * sizeof(float), stream) != hipSuccess) return LADDER_E_LAUNCH;
#define LADDER_SPLIT_LAUNCH__(P_, PROJ_, UP2_) \
  hipLaunchKernelGGL((conv3x3_halo_split_kernel<P_, PROJ_, UP2_>), grid, block, 0, stream, x, (const uint4*)packed, bias, y, N, H, W, Cin, Cout, act, tiles_n, x_absmax, wamax, y_absmax, \
                     pw, pb, pout, pco, tap_masks, s2_out)
#define LADDER_SPLIT_LAUNCH_(P_, PROJ_) do { if (s2_out == 2 || s2_out == 3) LADDER_SPLIT_LAUNCH__(P_, PROJ_, 1); else LADDER_SPLIT_LAUNCH__(P_, PROJ_, 0); } while (0)
#define LADDER_SPLIT_LAUNCH(P_) do { if (pout != nullptr) LADDER_SPLIT_LAUNCH_(P_, true); else LADDER_SPLIT_LAUNCH_(P_, false); } while (0)
#define LADDER_SPLIT16_LAUNCH_(P_, PROJ_, UP2_) \
  hipLaunchKernelGGL((conv3x3_halo_split16_kernel<P_, PROJ_, UP2_>), dim3(N * (H / F_H) * (W / SP_W) * tiles_n), dim3(F_THREADS), 0, stream, x, (const uint4*)packed, bias, y, N, H, W, Cin, Cout, act, tiles_n, x_absmax, wamax, y_absmax, \
                     pw, pb, pout, pco, tap_masks, s2_out)
#define LADDER_SPLIT16_LAUNCH(P_, PROJ_) do { if (s2_out == 2 || s2_out == 3) LADDER_SPLIT16_LAUNCH_(P_, PROJ_, 1); else LADDER_SPLIT16_LAUNCH_(P_, PROJ_, 0); } while (0)
  if (s2_out == 4) {                                       // backward-data of the upsample-fused pair: 16-wave kernel, plain epilogue
    if (pout != nullptr || !split_halo16_ok(N, H, W, Cin, Cout, prec)) return LADDER_E_SHAPE;
    if (prec == LADDER_PREC_F16X3) LADDER_SPLIT16_LAUNCH_(LADDER_PREC_F16X3, false, 2); else LADDER_SPLIT16_LAUNCH_(LADDER_PREC_BF16X3, false, 2);
  } else if (split_halo16_ok(N, H, W, Cin, Cout, prec)) {
    if (pout != nullptr) {
      if (prec == LADDER_PREC_F16X3) LADDER_SPLIT16_LAUNCH(LADDER_PREC_F16X3, true); else LADDER_SPLIT16_LAUNCH(LADDER_PREC_BF16X3, true);
    } else {
      if (prec == LADDER_PREC_F16X3) LADDER_SPLIT16_LAUNCH(LADDER_PREC_F16X3, false); else LADDER_SPLIT16_LAUNCH(LADDER_PREC_BF16X3, false);
    }
  } else if (prec == LADDER_PREC_F16X3) LADDER_SPLIT_LAUNCH(LADDER_PREC_F16X3);
  else if (prec == LADDER_PREC_BF16X6) LADDER_SPLIT_LAUNCH(LADDER_PREC_BF16X6);
  else LADDER_SPLIT_LAUNCH(LADDER_PREC_BF16X3);
#undef LADDER_SPLIT_LAUNCH
#undef LADDER_SPLIT_LAUNCH_
#undef LADDER_SPLIT_LAUNCH__
#undef LADDER_SPLIT16_LAUNCH_
#undef LADDER_SPLIT16_LAUNCH
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_conv3x3_split(const float* x, const float* x_absmax, const void* packed, const float* bias, float* y, float* y_absmax, int N,
                         int H, int W, int Cin, int Cout, int act, int prec, ladder_stream_t stream) {
  return conv3x3_split_launch(x, x_absmax, packed, bias, y, y_absmax, nullptr, nullptr, nullptr, 0, N, H, W, Cin, Cout, act, prec, stream);
}

int ladder_conv3x3_split_proj(const float* x, const float* x_absmax, const void* packed, const float* bias, float* y, const float* proj_w,
                              const float* proj_b, float* proj_out, int proj_cout, int N, int H, int W, int Cin, int Cout, int act,
                              int prec, ladder_stream_t stream) {
  if (proj_out == nullptr) return LADDER_E_SHAPE;
  return conv3x3_split_launch(x, x_absmax, packed, bias, y, nullptr, proj_w, proj_b, proj_out, proj_cout, N, H, W, Cin, Cout, act, prec,
                              stream);
}

// ---- backward-data of a 3x3 / stride-2 / SAME convolution (even maps: pad_t = pad_l = 0) as ONE launch of the halo kernel ---------------
// dx[h, w] = sum over (r, s) with h - r, w - s even of dy[(h - r) / 2, (w - s) / 2] . w[r, s]^T splits into the four output-parity classes
// (h % 2, w % 2), each a 3x3 / stride-1 correlation over dy that uses 4 / 2 / 2 / 1 of the nine taps.  The four classes are the four
// 128-channel OUTPUT TILES of one halo-kernel launch over dy (filter bank [9][Cout][4 * 128], ladder_filter_pack_split with
// transpose_flip = 2); a tile issues only its class's taps (tap mask) and its epilogue writes to the interleaved pixels of dx.  The
// gather kernel ran the classes as four launches with K = 128 ... 512 each -- 345 us on enc.conv1 at batch 128 (1.43 GB moved at
// 4.1 TB/s: bandwidth-bound on re-reading dy and the filter per tap); here dy is staged once per slab for all taps of a class.
static unsigned long long s2_tap_masks() { return filter_bank_tap_masks(2); }

int ladder_conv3x3_s2_bwd_data_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                              int pad_l) {
  static const bool off = getenv("LADDER_DISABLE_S2HALO") != nullptr;          // (test-only switch, see include/ladder_hip.h)
  return (!off && KH == 3 && KW == 3 && stride == 2 && pad_t == 0 && pad_l == 0 && H == 2 * Ho && W == 2 * Wo && Cin == SP_BN &&
          split_halo_ok(N, Ho, Wo, Cout, 4 * SP_BN)) ? 1 : 0;
}

int ladder_conv3x3_s2_bwd_data_split(const float* dy, const float* dy_absmax, const void* packed_s2, float* dx, float* dx_absmax, int N, int H,
                                     int W, int Cin, int Ho, int Wo, int Cout, int prec, ladder_stream_t stream) {
  if (prec == LADDER_PREC_F32) {     // strict fp32: any class width the fp32 tilings take (csrc/convf32.hip, csrc/convf32s.hip)
    if (!ladder_conv3x3_s2_bwd_data_f32_eligible(N, H, W, Cin, Ho, Wo, Cout)) return LADDER_E_SHAPE;
    return conv3x3_f32_launch(dy, (const float*)packed_s2, nullptr, dx, nullptr, nullptr, nullptr, 0, N, Ho, Wo, Cout, 4 * Cin, LADDER_ACT_NONE, stream,
                              s2_tap_masks(), 1);
  }
  if (!ladder_conv3x3_s2_bwd_data_split_eligible(N, H, W, Cin, Ho, Wo, Cout, 3, 3, 2, 0, 0)) return LADDER_E_SHAPE;
  return conv3x3_split_launch(dy, dy_absmax, packed_s2, nullptr, dx, dx_absmax, nullptr, nullptr, nullptr, 0, N, Ho, Wo, Cout, 4 * SP_BN,
                              LADDER_ACT_NONE, prec, stream, s2_tap_masks(), 1);
}

// ---- a 3x3 / SAME convolution of the factor-2 legacy-bilinear UPSAMPLE of x, without the upsampled tensor ---------------------------------
// (decoder: tf.image.resize_images(x, 2H x 2W) followed by conv2d, codes/models.py:554-578).  up[2i] = x[i], up[2i+1] = (x[i] + x[i+1]) / 2
// (index clamped), so each output-parity class (row % 2, col % 2) of the convolution is a 3x3 correlation over x itself with effective
// taps W_eff (ladder_filter_pack_split, transpose_flip = 3) -- 9 / 6 / 6 / 4 taps = 25 low-resolution tap products per 2x2 output block
// instead of 36, and the 4x larger tensor is neither written nor read.  One launch of the halo kernel: the classes are its four
// output-channel tiles (tap masks, interleaving epilogue, as for the stride-2 backward-data above), the halo outside the map is the
// clamped pixel with the sign that reproduces the zero padding of the high-resolution convolution.  That is exact everywhere but in the
// LAST output row and column (there the clamp of the resize and the padding of the convolution cannot both be expressed by one halo value):
// those 2H + 2W - 1 pixels per image are recomputed from the last row / column of x by ladder_conv3x3_up2_edges.
static unsigned long long up2_tap_masks() { return filter_bank_tap_masks(3); }

int ladder_conv3x3_up2_split_eligible(int N, int H, int W, int Cin, int Cout, int prec) {
  static const bool off = getenv("LADDER_DISABLE_UP2") != nullptr;              // (test-only switch, see include/ladder_hip.h)
  if (prec == LADDER_PREC_F32) return (!off && Cout > 0 && conv3x3_f32_any_ok(N, H, W, Cin, 4 * Cout, 2)) ? 1 : 0;   // (any class width: convf32s.hip)
  return (!off && (prec_ok(prec) || prec == LADDER_PREC_F32) && Cout == SP_BN && split_halo_ok(N, H, W, Cin, 4 * SP_BN)) ? 1 : 0;
}

int ladder_conv3x3_up2_split(const float* x, const float* x_absmax, const void* packed_up2, const float* bias, float* y, float* y_absmax,
                             int N, int H, int W, int Cin, int Cout, int act, int prec, int x_upsampled, ladder_stream_t stream) {
  if (!ladder_conv3x3_up2_split_eligible(N, H, W, Cin, Cout, prec)) return LADDER_E_SHAPE;
  return conv3x3_split_launch(x, x_absmax, packed_up2, bias, y, y_absmax, nullptr, nullptr, nullptr, 0, N, H, W, Cin, 4 * Cout, act, prec,
                              stream, up2_tap_masks(), x_upsampled ? 3 : 2);
}

// ... with the 1x1 projection of ladder_conv3x3_split_proj fused behind it (y may be NULL in forward-only runs)
int ladder_conv3x3_up2_split_proj(const float* x, const float* x_absmax, const void* packed_up2, const float* bias, float* y, const float* proj_w,
                                  const float* proj_b, float* proj_out, int proj_cout, int N, int H, int W, int Cin, int Cout, int act, int prec,
                                  int x_upsampled, ladder_stream_t stream) {
  if (!ladder_conv3x3_up2_split_eligible(N, H, W, Cin, Cout, prec) || proj_out == nullptr || Cout != SP_BN || (prec != LADDER_PREC_F32 && prec_planes(prec) != 2)) return LADDER_E_SHAPE;
  return conv3x3_split_launch(x, x_absmax, packed_up2, bias, y, nullptr, proj_w, proj_b, proj_out, proj_cout, N, H, W, Cin, 4 * SP_BN, act, prec,
                              stream, up2_tap_masks(), x_upsampled ? 3 : 2);
}

// Backward-data of the pair (factor-2 legacy-bilinear resize -> 3x3 / SAME convolution), dy [N, 2H, 2W, C] -> dx_lo [N, H, W, Cout], as ONE launch:
// the composite transpose is a zero-padded 5x5 / stride-2 correlation over dy (25 tap products per low-resolution pixel instead of 36 +
// the resize transpose, and the [N, 2H, 2W, Cout] intermediate is never written), exact everywhere but on the FOUR border lines of dx_lo
// (rows 0 and H-1, columns 0 and W-1: there the resize's clamp and the convolution's zero padding change the coefficients) -- the caller
// recomputes those from strips of dy with the plain backward-data + resize-transpose kernels.  packed_up2t =
// ladder_filter_pack_split(w, ., 9, 4 * C, Cout, transpose_flip = 4, prec) from the layer's HWIO bank [3][3][Cout][C].  16-wave kernel only.
static unsigned long long up2t_tap_masks() { return filter_bank_tap_masks(4); }

int ladder_conv3x3_up2_bwd_data_split_eligible(int N, int H, int W, int C, int Cout, int prec) {
  static const bool off = getenv("LADDER_DISABLE_UP2") != nullptr;
  if (prec == LADDER_PREC_F32) return (!off && (C % 16) == 0 && conv3x3_f32_any_ok(N, H, W, 4 * C, Cout, 4)) ? 1 : 0;      // (8-wave fp32 kernels)
  return (!off && prec_ok(prec) && (C % 16) == 0 && split_halo_ok(N, H, W, 4 * C, Cout) && split_halo16_ok(N, H, W, 4 * C, Cout, prec)) ? 1 : 0;
}

int ladder_conv3x3_up2_bwd_data_split(const float* dy, const float* dy_absmax, const void* packed_up2t, float* dx, float* dx_absmax, int N, int H,
                                      int W, int C, int Cout, int prec, ladder_stream_t stream) {
  if (!ladder_conv3x3_up2_bwd_data_split_eligible(N, H, W, C, Cout, prec)) return LADDER_E_SHAPE;
  return conv3x3_split_launch(dy, dy_absmax, packed_up2t, nullptr, dx, dx_absmax, nullptr, nullptr, nullptr, 0, N, H, W, 4 * C, Cout,
                              LADDER_ACT_NONE, prec, stream, up2t_tap_masks(), 4);
}

// One border line of dx_lo [N, H, W, C] from the plain backward-data on the adjoining strip of dy (d_up: rows strip [N, n_up, 2W, C] for axis 1,
// columns strip [N, 2H, n_up, C] for axis 2; n_up = 2 at the first border, 3 at the last): see the kernel.  dx_absmax (per-sample record of the
// main launch) is raised where a border value exceeds it.
int ladder_conv3x3_up2_bwd_border(const float* d_up, float* dx, float* dx_absmax, int N, int H, int W, int C, int axis, int first,
                                  ladder_stream_t stream) {
  if (N <= 0 || H <= 1 || W <= 1 || C <= 0 || (C % 4) != 0 || (axis != 1 && axis != 2)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(d_up) || !ladder_aligned16(dx)) return LADDER_E_ALIGN;
  const int Lq = axis == 1 ? W : H;
  hipLaunchKernelGGL(up2_bwd_border_kernel, dim3((Lq * (C / 4) + 255) / 256, N), dim3(256), 0, stream, d_up, dx, dx_absmax, H, W, C, axis, first ? 1 : 0);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

static size_t up2_align(size_t b) { return (b + 255) & ~(size_t)255; }

size_t ladder_conv3x3_up2_edges_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const size_t K = (size_t)3 * Cin, mr = (size_t)N * 2 * W, mc = (size_t)N * 2 * H;
  const size_t g = ladder_igemm_fwd_workspace_bytes((long)(mr > mc ? mr : mc), (int)K, Cout);
  return up2_align(mr * Cin * 4) + up2_align(mc * Cin * 4) + 2 * up2_align(K * Cout * 4) + up2_align(mr * Cout * 4) + up2_align(mc * Cout * 4) + up2_align(g) + 256;
}

// Recomputes the last output row and column of ladder_conv3x3_up2_split(_proj) from x and the layer's HWIO bank w [3][3][Cin][Cout] in
// fp32 (a 1x3 and a 3x1 convolution over the 1-D upsampled last row / column on the fp32 matrix cores + operand / scatter kernels); y and / or
// pout (with pw [Cout][pco], pb [pco]) receive them, the per-sample record y_absmax (already written by the main launch) is raised where an
// edge value exceeds it.
int ladder_conv3x3_up2_edges(const float* x, const float* w, const float* bias, float* y, float* y_absmax, const float* pw, const float* pb,
                             float* pout, int pco, int N, int H, int W, int Cin, int Cout, int act, int x_upsampled, void* ws, size_t ws_bytes,
                             ladder_stream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 4) != 0 || Cout <= 0 || (y == nullptr && pout == nullptr)) return LADDER_E_SHAPE;
  if (pout != nullptr && (pw == nullptr || pco < 1 || pco > 4)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x)) return LADDER_E_ALIGN;
  if (ws == nullptr || ws_bytes < ladder_conv3x3_up2_edges_workspace_bytes(N, H, W, Cin, Cout)) return LADDER_E_WORKSPACE;
  const size_t K = (size_t)3 * Cin, mr = (size_t)N * 2 * W, mc = (size_t)N * 2 * H;
  char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* u_row = (float*)p; p += up2_align(mr * Cin * 4);
  float* v_col = (float*)p; p += up2_align(mc * Cin * 4);
  float* w_row = (float*)p; p += up2_align(K * Cout * 4);
  float* w_col = (float*)p; p += up2_align(K * Cout * 4);
  float* e_row = (float*)p; p += up2_align(mr * Cout * 4);
  float* e_col = (float*)p; p += up2_align(mc * Cout * 4);
  const size_t g = ws_bytes - (size_t)(p - (char*)ws);
  if (up2_edge_lines_f32_ok(N, H, W, Cin, Cout) && ladder_aligned16(w)) {     // round 5: both lines in one launch (csrc/convf32s.hip)
    const int rc = up2_edge_lines_f32(x, w, bias, e_row, e_col, N, H, W, Cin, Cout, act, x_upsampled, stream);
    if (rc != LADDER_OK) return rc;
  } else {
    hipLaunchKernelGGL(up2_edge_operands_kernel, dim3(1024), dim3(256), 0, stream, x, w, u_row, v_col, w_row, w_col, N, H, W, Cin, Cout, x_upsampled ? 2 : 1);
    int rc = ladder_conv2d_fwd(u_row, w_row, bias, e_row, N, 1, 2 * W, Cin, 1, 2 * W, Cout, 1, 3, 1, 0, 1, act, p, g, stream);
    if (rc != LADDER_OK) return rc;
    rc = ladder_conv2d_fwd(v_col, w_col, bias, e_col, N, 2 * H, 1, Cin, 2 * H, 1, Cout, 3, 1, 1, 1, 0, act, p, g, stream);
    if (rc != LADDER_OK) return rc;
  }
  hipLaunchKernelGGL(up2_edge_scatter_kernel, dim3((2 * W + 2 * H - 1 + 3) / 4, N), dim3(256), 0, stream, (const float*)e_row, (const float*)e_col, y,
                     pw, pb, pout, pco, H, W, Cout, y_absmax);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// (two-plane formats only: the double-buffered 2x32-pixel patch images of a three-plane format exceed the 160 KB of LDS)
int ladder_conv3x3_wgrad_split_eligible(int N, int H, int W, int Cin, int Cout, int prec) {
  return (prec_ok(prec) && prec_planes(prec) == 2 && plan_wgrad_split(N, H, W, Cin, Cout).ok) ? 1 : 0;
}

size_t ladder_conv3x3_wgrad_split_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  const WgradSplitPlan p = plan_wgrad_split(N, H, W, Cin, Cout);
  if (!p.ok) return 0;
  return ((size_t)p.splits * 9 * Cin * Cout + (size_t)p.splits * Cout) * sizeof(float);
}

int ladder_conv3x3_wgrad_split(const float* x, const float* x_absmax, const float* dy, const float* dy_absmax, float* dw, float* db,
                               int N, int H, int W, int Cin, int Cout, int prec, void* ws, size_t ws_bytes, ladder_stream_t stream) {
  const WgradSplitPlan p = plan_wgrad_split(N, H, W, Cin, Cout);
  if (!p.ok || !prec_ok(prec) || prec_planes(prec) != 2) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(dy) || !ladder_aligned16(dw)) return LADDER_E_ALIGN;
  if (prec == LADDER_PREC_F16X3 && (x_absmax == nullptr || dy_absmax == nullptr)) return LADDER_E_SHAPE;
  if (ws == nullptr || ws_bytes < ladder_conv3x3_wgrad_split_workspace_bytes(N, H, W, Cin, Cout)) return LADDER_E_WORKSPACE;
  const size_t kn = (size_t)9 * Cin * Cout;
  float* part = (float*)ws;
  float* bias_part = db != nullptr ? part + (size_t)p.splits * kn : nullptr;
  const dim3 grid(p.tiles_ci * p.tiles_co * ((p.splits + 7) / 8) * 8), block(WS_THREADS);
#define LADDER_WS_LAUNCH(P_) \
  hipLaunchKernelGGL(wgrad3x3_split_kernel<P_>, grid, block, 0, stream, x, dy, part, bias_part, N, H, W, Cin, Cout, p.tiles_ci, p.tiles_co, p.splits, p.pps, x_absmax, dy_absmax)
  if (prec == LADDER_PREC_F16X3) LADDER_WS_LAUNCH(LADDER_PREC_F16X3);
  else LADDER_WS_LAUNCH(LADDER_PREC_BF16X3);
#undef LADDER_WS_LAUNCH
  LADDER_CHECK_LAUNCH();
  int rc = ladder_reduce_splits(part, dw, p.splits, kn, stream);
  if (rc != LADDER_OK) return rc;
  if (db != nullptr) rc = ladder_reduce_splits(bias_part, db, p.splits, (size_t)Cout, stream);
  return rc;
}

}  // extern "C"
