// Strict-fp32 forms of the FUSED 3x3 halo convolutions (gfx950, MI355X): every product on v_mfma_f32_32x32x2_f32, i.e. bit-exact fp32 FMA
// chains at the fp32 matrix / vector rate (157.3 TFLOP/s peak).  Round 4: everything rounds 2-3 built for the 16-bit split kernels of
// convsplit.hip that REMOVES work rather than narrowing operands, brought to the precision the reference computes in (fp32 end to end,
// codes/models.py:348,388; codes/base.py:457-517):
//
//   * resize x2 -> 3x3 conv as ONE convolution over the low-resolution map (decoder conv2d_6 / conv2d_7, codes/models.py:554-578): four
//     output-parity classes with effective taps (filterbank.h, orientation 3), 25 instead of 36 tap products per 2x2 output block, the 4x
//     larger upsampled tensor neither written nor read in forward-only runs;
//   * its backward-data as a 5x5 / stride-2 correlation over dy = a 3x3 correlation over dy's four pixel-parity classes (orientation 4);
//   * the 1x1 RGB output conv (codes/models.py:573-586) fused into the epilogue of the last 3x3 conv (transposed accumulators: lane =
//     pixel, so the 128 -> 3 channel sum stays inside a lane); forward-only runs never write the 1.07 GB activation;
//   * backward-data of a 3x3 / stride-2 conv (encoder conv2d_1, codes/models.py:409-418) as ONE launch: the four output-parity classes are
//     four output-channel tiles with 4 / 2 / 2 / 1 taps (orientation 2).
//
// The tiling is conv3x3_halo_kernel's (igemm.hip): one workgroup = 8 wavefronts = an 8x32-pixel patch x 128 output channels; per 16-channel
// input slab the (8+2)x(32+2) halo is staged in LDS ONCE for all taps (rows of odd stride 17 -> conflict-free ds_read_b32 fragments); the
// [16 x 128] filter slab of the next tap is double-buffered.  New here: a tile walks only the taps its class uses (the tap mask is scanned on
// the scalar unit: no barrier, no staging and no MFMA for an absent tap), the halo outside the map can be the signed clamped pixel, the
// input can be a parity class of a 2x larger tensor, and the epilogue can interleave class tiles into a [N, 2H, 2W, 128] map.
#include <cstdlib>
#include "convf32.h"
#include "filterbank.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FK = 16;                                                       // channels per input slab
constexpr int FH_H = 8, FH_W = 32, FH_PW = FH_W + 2, FH_PH = FH_H + 2, FH_LDA = FK + 1;
constexpr int FH_THREADS = 512, FH_BN = 128;
constexpr int FH_NPIX = FH_PH * FH_PW;                                       // 340 halo pixels
constexpr int FH_HALO_UNITS = FH_NPIX * (FK / 4);                            // float4 units per slab (1360)
constexpr int FH_AU = (FH_HALO_UNITS + FH_THREADS - 1) / FH_THREADS;         // 3

__device__ __attribute__((aligned(16))) float f32_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// PROJ: transposed accumulators (the filter fragment is the MFMA's A operand: lane = pixel, registers = channels) + fused 1x1 projection.
// UPM 0: the halo outside the map is zero; 1: upsample-fused forward (signed clamped halo; s2_out 2 / 3); 2: backward-data of the
// upsample-fused pair (x = dy [N, 2H, 2W, Cin / 4], input slabs grouped by pixel-parity class; s2_out 4).
template <bool PROJ, int UPM>
__global__ __launch_bounds__(FH_THREADS, 2) void conv3x3_halo_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                                         const int N, const int H, const int W, const int Cin,
                                                                         const int Cout, const int act, const int tiles_n,
                                                                         const float* __restrict__ pw, const float* __restrict__ pb,
                                                                         float* __restrict__ pout, const int pco,
                                                                         const unsigned long long tap_masks, const int s2_out) {
  __shared__ float Ah[2][FH_NPIX * FH_LDA];
  __shared__ __attribute__((aligned(16))) float Bh[2][FK * FH_BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / tiles_n;
  // class tiles (s2_out 1 / 2 / 3) rotate with the patch index: see class_tile (common.h) for the measurement behind it
  const int cot = class_tile(tile % tiles_n, mt, tiles_n, UPM != 2 && s2_out != 0), n0 = cot * FH_BN;
  const int tw_n = W / FH_W, th_n = H / FH_H;
  const int img = mt / (tw_n * th_n), rem_t = mt - img * (tw_n * th_n);
  const int h0 = (rem_t / tw_n) * FH_H, w0 = (rem_t % tw_n) * FH_W;
  const int nslabs = Cin / FK;
  const int spc = nslabs >> 2;                                               // UPM 2: 16-channel slabs per parity class of dy

  // halo load units (fixed per workgroup): source pointer at channel 0 of the slab (nullptr: zero), LDS offset, sign
  const float* hsrc[FH_AU];
  int hdst[FH_AU];
  unsigned hneg = 0u;
#pragma unroll
  for (int i = 0; i < FH_AU; ++i) {
    const int u = tid + i * FH_THREADS;
    const int pix = u >> 2, kq = u & 3;
    const int hr = pix / FH_PW, hc = pix - hr * FH_PW;
    const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
    // UPM 1: the halo outside the map is the CLAMPED pixel, negated above / left of the map (= the zero padding of the high-resolution
    // convolution, exactly: up[-1] = (x[-1] + x[0]) / 2 must vanish) and plain below / right of it (exact but for the last output row /
    // column, which ladder_conv3x3_up2_edges recomputes)
    const bool inside = hi >= 0 && hi < H && wi >= 0 && wi < W, up2 = UPM == 1;
    const bool ok = (u < FH_HALO_UNITS) && (inside || up2);
    const int hs = up2 ? min(max(hi, 0), H - 1) : hi, ws_ = up2 ? min(max(wi, 0), W - 1) : wi;
    // s2_out 3: x is the even-row / even-column sub-grid of an ALREADY upsampled [N, 2H, 2W, Cin] tensor (up[2i][2j] = x[i][j]);
    // UPM 2: pixel (hi, wi) of class (a, b) is dy[2 hi + a][2 wi + b], the class offset is added per slab
    const int sm = (UPM == 2 || (UPM == 1 && s2_out == 3)) ? 2 : 1;
    const int cpp = (UPM == 2) ? (Cin >> 2) : Cin;                           // channels per pixel of the tensor behind x
    hsrc[i] = ok ? x + (((long)img * (H * sm) + hs * sm) * (W * sm) + ws_ * sm) * cpp + kq * 4 : nullptr;
    hdst[i] = pix * FH_LDA + kq * 4;
    if (up2 && ((hi < 0) != (wi < 0))) hneg |= 1u << i;
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

  float4 ha[FH_AU], rb;
  auto load_halo = [&](int slab) {
    int soff = slab * FK;
    if (UPM == 2) {                                                          // slab -> (parity class, 16-channel slab of dy)
      const int cls = slab / spc;
      soff = ((cls >> 1) * 2 * W + (cls & 1)) * (Cin >> 2) + (slab - cls * spc) * FK;
    }
#pragma unroll
    for (int i = 0; i < FH_AU; ++i) ha[i] = *reinterpret_cast<const float4*>(hsrc[i] != nullptr ? hsrc[i] + soff : f32_zero16);
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < FH_AU; ++i)
      if (tid + i * FH_THREADS < FH_HALO_UNITS) {
        float4 v = ha[i];
        if (UPM == 1 && ((hneg >> i) & 1u)) v = make_float4(-v.x, -v.y, -v.z, -v.w);
        float* p = &Ah[buf][hdst[i]];
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
      }
  };
  auto load_b = [&](int slab, int tap) {
    rb = *reinterpret_cast<const float4*>(b_ok ? bsrc + ((long)tap * Cin + slab * FK) * Cout : f32_zero16);
  };
  auto store_b = [&](int buf) { *reinterpret_cast<float4*>(&Bh[buf][b_kr * FH_BN + b_nq * 4]) = rb; };

  // taps this tile issues: all 9 for a plain convolution; a class tile of a stride-2 backward-data / upsample-fused bank 1 ... 9 of them
  // (per OUTPUT tile; UPM 2: per input-slab class).  The masks live in scalar registers; an absent tap costs nothing at all.
  const unsigned tmask0 = (unsigned)(tap_masks >> (9 * cot)) & 0x1ffu;
  auto mask_of = [&](int slab) -> unsigned { return (UPM == 2) ? ((unsigned)(tap_masks >> (9 * (slab / spc))) & 0x1ffu) : tmask0; };

  load_halo(0);
  load_b(0, __builtin_ctz(mask_of(0)));
  store_halo(0);
  store_b(0);
  __syncthreads();
  int bbuf = 0;
  for (int slab = 0; slab < nslabs; ++slab) {
    const int hb = slab & 1;
    unsigned rem = mask_of(slab);
    const int half = __popc(rem) >> 1;
    const bool more = slab + 1 < nslabs;
    int i = 0;
#pragma unroll 1
    while (rem != 0u) {
      const int tap = __builtin_ctz(rem);
      rem &= rem - 1u;
      // prefetch: the filter slab of the next issued tap every step; the next input halo once per slab (fetched at the slab's first
      // tap, written at its middle one)
      int ntap = -1, nslab = slab;
      if (rem != 0u) ntap = __builtin_ctz(rem);
      else if (more) { nslab = slab + 1; ntap = __builtin_ctz(mask_of(slab + 1)); }
      if (ntap >= 0) load_b(nslab, ntap);
      if (i == 0 && more) load_halo(slab + 1);
      const int r = tap / 3, sft = tap - 3 * r;
      const float* Ab = &Ah[hb][((2 * wm + r) * FH_PW + sft + l31) * FH_LDA + lh];
      const float* Bb = &Bh[bbuf][lh * FH_BN + wn * 64 + l31];
#pragma unroll
      for (int ks = 0; ks < FK / 2; ++ks) {
        float a[2], b[2];
        a[0] = Ab[2 * ks];
        a[1] = Ab[FH_PW * FH_LDA + 2 * ks];
        b[0] = Bb[2 * ks * FH_BN];
        b[1] = Bb[2 * ks * FH_BN + 32];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = PROJ ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni], a[mi], acc[mi][ni], 0, 0, 0)    // D^T: lane = pixel
                               : __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // lane = channel
      }
      if (i == half && more) store_halo(hb ^ 1);
      if (ntap >= 0) store_b(bbuf ^ 1);
      __syncthreads();
      bbuf ^= 1;
      ++i;
    }
  }

  const int s2o = (UPM == 2) ? 0 : s2_out;                 // (mode 4 writes the plain [N, H, W, Cout] layout)
  if (!PROJ) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + l31;
      const float bv = (bias != nullptr && n < Cout) ? bias[s2o ? n - n0 : n] : 0.f;      // (class tiles share the layer's 128 channels)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        // s2o: this output-channel tile is one PARITY CLASS (ph, pw) = (cot >> 1, cot & 1) of a [N, 2H, 2W, 128] map: pixel (h, w) of the
        // class is out[2h + ph, 2w + pw], channels = the tile's 128
        float* yp = s2o ? y + (((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (cot >> 1)) * 2 * W + 2 * w0 + (cot & 1)) * FH_BN + (n - n0)
                        : y + (((long)img * H + h0 + 2 * wm + mi) * W + w0) * Cout + n;
        const long pstride = s2o ? 2 * FH_BN : Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (n < Cout) yp[(long)px * pstride] = ladder_act_fn(acc[mi][ni][e] + bv, act);
        }
      }
    }
    return;
  }
  // Transposed accumulators: lane l31 = pixel of the patch row, register e -> channel (e & 3) + 8 (e >> 2) + 4 lh of the 32-channel tile,
  // i.e. four consecutive channels per register quad = one 16-byte store, and the channel sum of the 1x1 projection stays inside the lane.
  float pacc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int o = 0; o < 4; ++o) pacc[mi][o] = 0.f;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const long opix = s2o ? ((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (cot >> 1)) * 2 * W + 2 * (w0 + l31) + (cot & 1)
                            : ((long)img * H + h0 + 2 * wm + mi) * W + w0 + l31;
      const int coff = s2o ? n0 : 0;
      float* yp = y != nullptr ? y + opix * (s2o ? FH_BN : Cout) - coff : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * lh;
        if (n < Cout) {                                      // Cout % 4 == 0: a channel quad is inside or outside as a whole
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + n - coff);
          const float4 v = make_float4(ladder_act_fn(acc[mi][ni][4 * g] + bv.x, act), ladder_act_fn(acc[mi][ni][4 * g + 1] + bv.y, act),
                                       ladder_act_fn(acc[mi][ni][4 * g + 2] + bv.z, act), ladder_act_fn(acc[mi][ni][4 * g + 3] + bv.w, act));
          if (yp != nullptr) *reinterpret_cast<float4*>(yp + n) = v;
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
  if (pout != nullptr) {
    // combine the two half-waves (lh) in registers, the two channel halves (wn) through LDS (free after the main loop), fixed order
    float* red = &Ah[0][0];                                  // [wm 4][mi 2][pixel 32][4]
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
        const long opix = s2o ? ((long)img * 2 * H + 2 * (h0 + 2 * wm + mi) + (cot >> 1)) * 2 * W + 2 * (w0 + l31) + (cot & 1)
                              : ((long)img * H + h0 + 2 * wm + mi) * W + w0 + l31;
        float* op = pout + opix * pco;
        for (int o = 0; o < pco; ++o) op[o] = t[o] + (pb != nullptr ? pb[o] : 0.f);
      }
    }
  }
}

// ---- fp32 banks of the logical filter (filterbank.h) ------------------------------------------------------------------------------
// bank[tap][ci][co], ci < Cin, co < Cout (LOGICAL dimensions of the orientation); thread index decoded as in filter_pack_element of
// convsplit.hip, so that a job table's block counts (ladder_filter_pack_job_blocks) serve both precisions:
//   i -> (tap, 16-channel slab, 128-column tile, channel octet kg, column) ; a thread writes 8 rows (ci) of one column
__device__ __forceinline__ void filter_pack_f32_element(const float* __restrict__ w, float* __restrict__ out, int ntaps, int Cin, int Cout,
                                                        int transpose_flip, int i) {
  const int cots = (Cout + FH_BN - 1) / FH_BN, nslabs = Cin / 16;
  const int col = i % FH_BN;
  int t = i / FH_BN;
  const int kg = t & 1;
  t >>= 1;
  const int cot = t % cots;
  t /= cots;
  const int slab = t % nslabs, tap = t / nslabs;
  const int co = cot * FH_BN + col, ci0 = slab * 16 + kg * 8;
  if (co >= Cout) return;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    out[((size_t)tap * Cin + ci0 + j) * Cout + co] = filter_bank_element(w, ntaps, Cin, Cout, transpose_flip, tap, ci0 + j, co);
}

__global__ __launch_bounds__(256) void filter_pack_f32_kernel(const float* __restrict__ w, float* __restrict__ out, int ntaps, int Cin,
                                                              int Cout, int transpose_flip, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total) filter_pack_f32_element(w, out, ntaps, Cin, Cout, transpose_flip, i);
}

__global__ __launch_bounds__(256) void filter_pack_f32_multi_kernel(const ladder_pack_job_t* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;                                // job of this block: block_begin is ascending
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ladder_pack_job_t j = jobs[lo];
  const int total = j.ntaps * (j.Cin / 16) * ((j.Cout + FH_BN - 1) / FH_BN) * 2 * FH_BN;
  const int i = ((int)blockIdx.x - j.block_begin) * 256 + threadIdx.x;
  if (i < total) filter_pack_f32_element(j.w, (float*)j.packed, j.ntaps, j.Cin, j.Cout, j.transpose_flip, i);
}

}  // namespace

bool conv3x3_f32_halo_ok(int N, int H, int W, int Cin, int Cout) {
  return N > 0 && (Cin % FK) == 0 && (Cout % 4) == 0 && Cout >= 64 && (W % FH_W) == 0 && (H % FH_H) == 0 &&
         (long)N * (H / FH_H) * (W / FH_W) * ((Cout + FH_BN - 1) / FH_BN) >= 512;
}

size_t filter_pack_f32_bytes(int ntaps, int Cin, int Cout) { return (size_t)ntaps * Cin * Cout * sizeof(float); }

int filter_pack_f32(const float* w, float* bank, int ntaps, int Cin, int Cout, int transpose_flip, hipStream_t stream) {
  const long total_l = (long)ntaps * (Cin / 16) * ((Cout + FH_BN - 1) / FH_BN) * 2 * FH_BN;
  if (total_l <= 0 || total_l >= (1L << 31)) return LADDER_E_SHAPE;
  const int total = (int)total_l;
  hipLaunchKernelGGL(filter_pack_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, bank, ntaps, Cin, Cout, transpose_flip, total);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int filter_pack_f32_multi(const ladder_pack_job_t* jobs_dev, int njobs, int total_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(filter_pack_f32_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, jobs_dev, njobs);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int conv3x3_f32_launch(const float* x, const float* bank, const float* bias, float* y, const float* pw, const float* pb, float* pout,
                       int pco, int N, int H, int W, int Cin, int Cout, int act, hipStream_t stream, unsigned long long tap_masks,
                       int s2_out) {
  if (!conv3x3_f32_halo_ok(N, H, W, Cin, Cout)) return LADDER_E_SHAPE;
  if (pout != nullptr && (pw == nullptr || pco < 1 || pco > 4 || (Cout > FH_BN && s2_out < 2) || !ladder_aligned16(pw))) return LADDER_E_SHAPE;
  if (pout == nullptr && y == nullptr) return LADDER_E_SHAPE;
  if (s2_out == 4 && (pout != nullptr || (Cin % (4 * FK)) != 0)) return LADDER_E_SHAPE;
  if ((s2_out >= 1 && s2_out <= 3) && Cout != 4 * FH_BN) return LADDER_E_SHAPE;          // class tiles = the layer's 128 channels each
  if (!ladder_aligned16(x) || !ladder_aligned16(bank) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias)))
    return LADDER_E_ALIGN;
  const int tiles_n = (Cout + FH_BN - 1) / FH_BN;
  const int tiles_m = N * (H / FH_H) * (W / FH_W);
  const dim3 grid(tiles_m * tiles_n), block(FH_THREADS);
#define LADDER_F32_LAUNCH(PROJ_, UPM_) \
  hipLaunchKernelGGL((conv3x3_halo_f32_kernel<PROJ_, UPM_>), grid, block, 0, stream, x, bank, bias, y, N, H, W, Cin, Cout, act, tiles_n, pw, pb, pout, pco, tap_masks, s2_out)
  if (s2_out == 4) LADDER_F32_LAUNCH(false, 2);
  else if (s2_out == 2 || s2_out == 3) { if (pout != nullptr) LADDER_F32_LAUNCH(true, 1); else LADDER_F32_LAUNCH(false, 1); }
  else { if (pout != nullptr) LADDER_F32_LAUNCH(true, 0); else LADDER_F32_LAUNCH(false, 0); }
#undef LADDER_F32_LAUNCH
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}
