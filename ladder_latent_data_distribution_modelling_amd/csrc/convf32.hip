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
__global__ __launch_bounds__(FH_THREADS, 4) void conv3x3_halo_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                                         const int N, const int H, const int W, const int Cin,
                                                                         const int Cout, const int act, const int tiles_n,
                                                                         const float* __restrict__ pw, const float* __restrict__ pb,
                                                                         float* __restrict__ pout, const int pco,
                                                                         const unsigned long long tap_masks, const int s2_out) {
  __shared__ float Ah[2][FH_NPIX * FH_LDA];
  __shared__ __attribute__((aligned(16))) float Bh[2][FK * FH_BN];
  // PROJ: the projection matrix (rows padded to 4 outputs) and the bias of the tile's 128 channels, staged once: the epilogue reads them
  // as LDS broadcasts (from global memory its 64 dependent 16-byte loads per lane cost ~6 us per tile)
  __shared__ __attribute__((aligned(16))) float Pw[PROJ ? FH_BN * 4 + FH_BN : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  // cls_loop (bit 8 of s2_out; class launches only): ONE workgroup computes all four class tiles of its patch one after another -- every
  // workgroup of the launch then carries the same 25 taps per slab (no class mix to balance, whole rounds of the chip) and the index
  // arithmetic of the patch is done once; the grid is the patch count.  Else one class tile per workgroup, rotated with the patch index
  // (class_tile, common.h).
  const int s2_mode = s2_out & 0xff;
  const bool cls_loop = (s2_out & 0x100) != 0;
  const int mt = cls_loop ? tile : tile / tiles_n;
  const int cot_first = cls_loop ? (mt & 3) : class_tile(tile % tiles_n, mt, tiles_n, UPM != 2 && s2_mode != 0);
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
    const int sm = (UPM == 2 || (UPM == 1 && s2_mode == 3)) ? 2 : 1;
    const int cpp = (UPM == 2) ? (Cin >> 2) : Cin;                           // channels per pixel of the tensor behind x
    hsrc[i] = ok ? x + (((long)img * (H * sm) + hs * sm) * (W * sm) + ws_ * sm) * cpp + kq * 4 : nullptr;
    hdst[i] = pix * FH_LDA + kq * 4;
    if (up2 && ((hi < 0) != (wi < 0))) hneg |= 1u << i;
  }
  const int b_kr = tid >> 5, b_nq = tid & 31;                                // 16 rows x 32 float4 = 512 units
  if (PROJ) {                                                                // (channels of a projecting tile are local: Cout <= 128 or class tiles)
    const int c = tid >> 2, o = tid & 3;
    const int cmax = min(Cout, FH_BN);
    Pw[tid] = (pw != nullptr && c < cmax && o < pco) ? pw[c * pco + o] : 0.f;
    if (tid < FH_BN) Pw[FH_BN * 4 + tid] = (bias != nullptr && tid < cmax) ? bias[tid] : 0.f;   // (visible after the first barrier of the tap loop)
  }
  for (int cc = 0; cc < (cls_loop ? 4 : 1); ++cc) {
  const int cot = (cot_first + cc) & (cls_loop ? 3 : 0x7fffffff), n0 = cot * FH_BN;
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

  const int s2o = (UPM == 2) ? 0 : s2_mode;                // (mode 4 writes the plain [N, H, W, Cout] layout)
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
        if constexpr (UPM == 2) {
          // backward-data of an upsample-fused pair: `pw` / `pco` of a non-projecting launch carry an optional GATE -- the activation OUTPUT of the
          // layer that produced the low-resolution tensor and its activation id: dx *= act'(gate), that layer's activation backward (round 5: the
          // separate ladder_act_bwd pass over conv2d_5's output is gone)
          const float* gp = (pw != nullptr && n < Cout) ? pw + (yp - y) : nullptr;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
            float v = ladder_act_fn(acc[mi][ni][e] + bv, act);
            if (gp != nullptr) v *= ladder_act_grad_from_out(gp[(long)px * pstride], pco);
            if (n < Cout) yp[(long)px * pstride] = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (n < Cout) yp[(long)px * pstride] = ladder_act_fn(acc[mi][ni][e] + bv, act);
          }
        }
      }
    }
    continue;                                                // (next class of the patch, or done)
  }
  // Transposed accumulators: lane l31 = pixel of the patch row, register e -> channel (e & 3) + 8 (e >> 2) + 4 lh of the 32-channel tile,
  // i.e. four consecutive channels per register quad = one 16-byte store, and the channel sum of the 1x1 projection stays inside the lane.
  float pacc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int o = 0; o < 4; ++o) pacc[mi][o] = 0.f;
  // (bias quad + four projection rows per channel quad as 16-byte LDS broadcasts: a broadcast still returns 1 KB per instruction, so these
  // 80 reads per lane, not the 256 FMAs, are what this epilogue costs -- ~2 us per tile; sharing the rows between the lane's two pixel rows
  // was tried and lost to the spills it caused)
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
          const int nl = n - n0;                             // local channel of the tile
          const float4 bv = *reinterpret_cast<const float4*>(&Pw[FH_BN * 4 + nl]);
          const float4 v = make_float4(ladder_act_fn(acc[mi][ni][4 * g] + bv.x, act), ladder_act_fn(acc[mi][ni][4 * g + 1] + bv.y, act),
                                       ladder_act_fn(acc[mi][ni][4 * g + 2] + bv.z, act), ladder_act_fn(acc[mi][ni][4 * g + 3] + bv.w, act));
          if (yp != nullptr) *reinterpret_cast<float4*>(yp + n) = v;
          if (pout != nullptr) {                             // 1x1 projection: pw[Cout][pco], pco <= 4 (rows zero-padded to 4 in LDS)
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
              const float4 q = *reinterpret_cast<const float4*>(&Pw[(nl + c4) * 4]);
              pacc[mi][0] = fmaf(vv[c4], q.x, pacc[mi][0]);
              pacc[mi][1] = fmaf(vv[c4], q.y, pacc[mi][1]);
              pacc[mi][2] = fmaf(vv[c4], q.z, pacc[mi][2]);
              pacc[mi][3] = fmaf(vv[c4], q.w, pacc[mi][3]);
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
    __syncthreads();                                         // (the reduction scratch is the halo buffer the next class stages into)
  }
  }  // class loop
}

// ---- fp32 banks of the logical filter (filterbank.h) ------------------------------------------------------------------------------
// bank[tap][ci][co], ci < Cin, co < Cout (LOGICAL dimensions of the orientation); thread index decoded as in filter_pack_element of
// convsplit.hip, so that a job table's block counts (ladder_filter_pack_job_blocks) serve both precisions:
//   i -> (tap, 16-channel slab, 128-column tile, channel octet kg, column) ; a thread writes 8 rows (ci) of one column
// per-axis coefficients A_a[d][0..2] of the upsample algebra (filterbank.h) as three selects -- a thread decodes its tap and class ONCE; the
// table lookups of filter_bank_element (runtime-indexed local arrays = scratch memory) cost the all-bank re-pack 265 us per iteration in round 5's
// first build (18 banks, 130 MB written at 0.5 TB/s)
__device__ __forceinline__ void up2_axis_coef(int a, int d, float c[3]) {
  if (a == 0) {
    c[0] = d <= 1 ? 0.5f : 0.f;
    c[1] = d == 1 ? 1.f : 0.f;
    c[2] = d >= 1 ? 0.5f : 0.f;
  } else {
    c[0] = d == 1 ? 1.f : 0.f;
    c[1] = d >= 1 ? 0.5f : 0.f;
    c[2] = d == 2 ? 1.f : 0.f;
  }
}

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
  if ((transpose_flip == 3 && ((Cout >> 2) % 4) == 0) || (transpose_flip == 4 && ((Cin >> 2) % 8) == 0)) {
    // the effective taps of the upsample-fused forward (3) / backward-data (4): the SAME sums in the SAME order as filter_bank_element (r outer,
    // s inner, zero coefficients skipped, the exact product of two powers of two as the weight), with the class / tap decode hoisted
    const int dr = tap / 3, dc = tap - 3 * dr;
    float cr[3], cs[3];
    size_t base, rstride;                                     // address of term (r, s), row j: base + (r * 3 + s) * rstride + j * jstride
    int jstride;
    if (transpose_flip == 3) {
      const int C = Cout >> 2, cls = co / C, cc = co - cls * C;
      up2_axis_coef(cls >> 1, dr, cr);
      up2_axis_coef(cls & 1, dc, cs);
      base = (size_t)ci0 * C + cc; rstride = (size_t)Cin * C; jstride = C;
    } else {
      const int C = Cin >> 2, cls = ci0 / C, cc0 = ci0 - cls * C;            // (8 rows of one thread: one class, C % 8 == 0)
      up2_axis_coef(cls >> 1, 2 - dr, cr);
      up2_axis_coef(cls & 1, 2 - dc, cs);
      base = (size_t)co * C + cc0; rstride = (size_t)Cout * C; jstride = 1;
    }
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (cr[r] == 0.f) continue;
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) {
        if (cs[sx] == 0.f) continue;
        const float cf = cr[r] * cs[sx];
        const float* src = w + base + (size_t)(r * 3 + sx) * rstride;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += cf * src[(size_t)j * jstride];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[((size_t)tap * Cin + ci0 + j) * Cout + co] = f[j];
    return;
  }
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

// the 8x32-pixel x 128-channel tiling takes this call (class launches: classes of exactly 128 channels)
static bool conv3x3_f32_big_ok(int N, int H, int W, int Cin, int Cout, int s2_out) {
  if (!conv3x3_f32_halo_ok(N, H, W, Cin, Cout)) return false;
  if ((s2_out == 4 || s2_out == 5) && (Cin % (4 * FK)) != 0) return false;
  if ((s2_out >= 1 && s2_out <= 3) && Cout != 4 * FH_BN) return false;
  return true;
}

bool conv3x3_f32_any_ok(int N, int H, int W, int Cin, int Cout, int s2_out) {
  return conv3x3_f32_big_ok(N, H, W, Cin, Cout, s2_out) || conv3x3_f32s_ok(N, H, W, Cin, Cout, s2_out);
}

int conv3x3_f32_launch(const float* x, const float* bank, const float* bias, float* y, const float* pw, const float* pb, float* pout,
                       int pco, int N, int H, int W, int Cin, int Cout, int act, hipStream_t stream, unsigned long long tap_masks,
                       int s2_out) {
  if (!conv3x3_f32_big_ok(N, H, W, Cin, Cout, s2_out)) {     // small maps / wide classes: convf32s.hip (no fused projection there)
    if (pout != nullptr || pw != nullptr) return LADDER_E_SHAPE;        // (no fused projection / gate there)
    return conv3x3_f32s_launch(x, bank, bias, y, N, H, W, Cin, Cout, act, stream, tap_masks, s2_out);
  }
  if (pout != nullptr && (pw == nullptr || pco < 1 || pco > 4 || (Cout > FH_BN && s2_out < 2) || !ladder_aligned16(pw))) return LADDER_E_SHAPE;
  if (pout == nullptr && y == nullptr) return LADDER_E_SHAPE;
  if ((s2_out == 4 || s2_out == 5) && pout != nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(bank) || !ladder_aligned16(y) || (bias != nullptr && !ladder_aligned16(bias)))
    return LADDER_E_ALIGN;
  const int tiles_n = (Cout + FH_BN - 1) / FH_BN;
  const int tiles_m = N * (H / FH_H) * (W / FH_W);
  // class launches: all four class tiles of a patch in one workgroup when the patches alone fill whole rounds of the chip
  static const int cls_env = getenv("LADDER_CLASS_LOOP") ? atoi(getenv("LADDER_CLASS_LOOP")) : -1;
  const bool cls_loop = (s2_out >= 1 && s2_out <= 3) && (cls_env >= 0 ? cls_env != 0 : tiles_m >= 512);
  const dim3 grid(cls_loop ? tiles_m : tiles_m * tiles_n), block(FH_THREADS);
  const int s2x = s2_out | (cls_loop ? 0x100 : 0);
#define LADDER_F32_LAUNCH(PROJ_, UPM_) \
  hipLaunchKernelGGL((conv3x3_halo_f32_kernel<PROJ_, UPM_>), grid, block, 0, stream, x, bank, bias, y, N, H, W, Cin, Cout, act, tiles_n, pw, pb, pout, pco, tap_masks, s2x)
  if (s2_out == 4 || s2_out == 5) LADDER_F32_LAUNCH(false, 2);
  else if (s2_out == 2 || s2_out == 3) { if (pout != nullptr) LADDER_F32_LAUNCH(true, 1); else LADDER_F32_LAUNCH(false, 1); }
  else { if (pout != nullptr) LADDER_F32_LAUNCH(true, 0); else LADDER_F32_LAUNCH(false, 0); }
#undef LADDER_F32_LAUNCH
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------------
// Filter gradient of the pair (factor-2 legacy-bilinear resize -> 3x3 / SAME convolution) over the LOW-resolution map (round 4).
//
// Forward (filterbank.h, orientation 3): y[2i+a, 2j+b] = sum_{dr,dc} W_eff[a,b][dr,dc] . x~[i+dr-1, j+dc-1] with x~ = the low-resolution x
// with the signed clamped halo and W_eff[a,b][dr,dc] = sum_{r,s} A_a[dr][r] A_b[dc][s] w[r][s] -- exact on every output pixel but the last
// row / column.  Hence
//     dL/dw[r][s] = sum_{a,b} sum_{dr,dc} A_a[dr][r] A_b[dc][s] . G_ab[dr][dc],    G_ab[dr][dc] = sum_{n,i,j} x~[n, i+dr-1, j+dc-1] (x) dy[n, 2i+a, 2j+b]
// over all output pixels off the last row / column: 9 / 6 / 6 / 4 = 25 tap tiles of Cin x Cout instead of the 36 (4 pixel classes x 9 taps)
// the direct filter gradient on the upsampled map accumulates -- the same 25 / 36 as the forward.  The last output row / column are a 1x3 /
// 3x1 convolution of the 1-D upsampled last row / column of x with the taps w[0][s] + w[1][s] / w[r][0] + w[r][1] (convsplit.hip:
// ladder_conv3x3_up2_edges), so their contribution is a 1x3 / 3x1 filter gradient E_row[s] / E_col[r] added to dw[0][s], dw[1][s] / dw[r][0],
// dw[r][1].  Kernel = wgrad3x3_halo_kernel's structure (igemm.hip: 12 wavefronts, 64 ci x 128 co slab, 1x32-pixel patches, LDS-DMA staging,
// double buffered) on ONE parity class per workgroup:
//   classes (0, b): wavefront -> (filter row dr = 0..2, ci half, co pair); 3 - b column shifts x 2 co blocks = 6 / 4 MFMAs per k-step
//   (patches are 2 x 32 low-resolution pixels -- the halo overhead halves and one LDS-DMA round trip / barrier carries twice the MFMA work:
//   conv2d_7 at batch 128 5.8 -> 4.4 ms against 5.3 ms for the direct kernel under rocprofv3; the next patch is staged from inside the MFMA loop)
//   classes (1, b): rows dr = 1, 2 only -> wavefront -> (dr, ci half, column shift); 4 co blocks = 4 MFMAs per k-step (b = 1: 8 of the
//                   12 wavefronts, two per SIMD)
// i.e. 18 / 12 / 12 / 8 MFMA slots per SIMD and k-step against 4 x 18 for the direct kernel = 25 / 36.  The pixel split of a class is sized
// in proportion to that cost, so all workgroups of a launch run equally long (a fixed class order would hand every shader engine one class:
// see class_tile in common.h).  The signs of the halo (negated above / left of the map) are applied to the A fragments after the LDS read:
// the DMA staging cannot negate.
// A staged patch is 64 low-resolution pixels: 2 x 32 (maps of width % 32 == 0: 2 rows halve the halo overhead of a 1-row patch and double
// the MFMA work behind one LDS-DMA round trip), 4 x 16 or 8 x 8 (round 5: the 16x16 / 8x8 maps of decoder conv2d_5 / conv2d_4).
constexpr int WU_CI = 64, WU_CO = 128, WU_THREADS = 768, WU_PIX = 64;
constexpr int WU_DU = WU_PIX * (WU_CO / 4), WU_DN = (WU_DU + WU_THREADS - 1) / WU_THREADS, WU_DF = WU_PIX * WU_CO;     // dY: float4 units per patch (1024), per lane (2), floats
template <int PW> struct WuGeo {
  static constexpr int PH = WU_PIX / PW, HW = PW + 2, HH = PH + 2;
  static constexpr int XU = HH * HW * (WU_CI / 4), XN = (XU + WU_THREADS - 1) / WU_THREADS, XF = HH * HW * WU_CI;        // input halo: float4 units (2176 / 1728 / 1600), per lane (3), floats
  static constexpr int LDS_FLOATS = 2 * (XF + WU_DF) + XN * WU_THREADS;
};
constexpr int WU_NTAPS[4] = {9, 6, 6, 4}, WU_TAP0[4] = {0, 9, 15, 21};                    // taps per class, first combination index

struct WgradUp2Plan { bool ok; int pw, tiles_ci, tiles_co, ns[4], s0[4], total; };               // splits per class, first workgroup of each class (per slab pair)

template <int CLS, int WU_PW, bool BIAS>
__device__ __forceinline__ void wgrad_up2_body(float* __restrict__ lds, const float* __restrict__ x, const float* __restrict__ dy,
                                               float* __restrict__ out, float* __restrict__ bias_part, const int N, const int H, const int W,
                                               const int Cin, const int Cout, const int ci0, const int co0, const int split, const int nsplits,
                                               const int xs) {
  constexpr int A = CLS >> 1, B = CLS & 1, NS = 3 - B;
  constexpr bool MODEB = A == 1;
  using WG_ = WuGeo<WU_PW>;
  constexpr int WU_PH = WG_::PH, WU_HW = WG_::HW, WU_XU = WG_::XU, WU_XN = WG_::XN, WU_XF = WG_::XF;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;            // 12 wavefronts
  const int l31 = lane & 31, lh = lane >> 5;
  const int Hc = H - A, WP = W / WU_PW;                // (a = 1: the last low-resolution row maps to the last output row: edge path)
  const int Hg = (Hc + WU_PH - 1) / WU_PH;             // row groups of a patch column
  const int q_total = N * Hg * WP;
  const int pps = (q_total + nsplits - 1) / nsplits;
  const int q0 = split * pps, q1 = min(q_total, q0 + pps);
  // wavefront -> tiles
  int dr, cb, sh = 0, njp = 0;
  bool active = true;
  if (!MODEB) {
    dr = wv >> 2; cb = (wv >> 1) & 1; njp = wv & 1;
  } else if (B == 0) {
    dr = 1 + wv / 6; cb = (wv % 6) / 3; sh = (wv % 6) % 3;
  } else {
    active = wv < 8; dr = 1 + ((wv >> 2) & 1); cb = (wv >> 1) & 1; sh = 1 + (wv & 1);
  }
  const int a_off = (dr * WU_HW + lh) * WU_CI + cb * 32 + l31;          // + s * WU_CI (column shift) + 2 ks * WU_CI (pixel)
  const int b_off = WU_XF + lh * WU_CO + l31;                           // + co block * 32 + 2 ks * WU_CO
  // (BIAS: a compile-time switch since round 5 -- the layers in front of a norm have no bias gradient, and the sums + selects of the runtime form
  // were ~4 VALU instructions per k-step in every wavefront of every launch)
  const bool do_bias = BIAS && (bias_part != nullptr) && (ci0 == 0) && dr == 1 && cb == 0 && active && (!MODEB || sh == 1);
  constexpr int NACC = MODEB ? 4 : 6;
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  auto decode = [&](int q, int& n, int& i, int& j0) {            // i = first low-resolution row of the patch
    n = q / (Hg * WP);
    const int rem = q - n * (Hg * WP);
    const int g = rem / WP;
    i = g * WU_PH;
    j0 = (rem - g * WP) * WU_PW;
  };
  // one staging unit (16 bytes per lane by LDS-DMA) of patch (n, i, j0) into buffer `buf`: units 0 .. WU_XN-1 = input halo, then dY.
  // A lane's halo pixel of unit k is fixed: (row, column) = xpk[k], packed once (the division by the halo width is not repeated per patch),
  // and the source address is a 32-bit offset from the image's base -- a unit costs ~15 VALU instructions where the straightforward index
  // arithmetic cost ~50 (x 5 units per patch against 192 MFMAs: staging was 9 % of the kernel).
  // (kept in LDS behind the staging buffers, same object: registers are what this kernel is short of)
  int* const xpk = reinterpret_cast<int*>(lds + 2 * (WU_XF + WU_DF));
#pragma unroll
  for (int k = 0; k < WU_XN; ++k) {
    const int u = tid + k * WU_THREADS, pix = u >> 4;
    const int hr = pix / WU_HW, hc = pix - hr * WU_HW;
    xpk[u] = u < WU_XU ? ((hr << 8) | hc) : -1;            // (read back by the same lane only)
  }
  const int xq4 = (tid & 15) * 4, dq4 = (tid & 31) * 4;
  const bool d_cok = (co0 + dq4) < Cout;
  auto stage_unit = [&](int k, int n, int i, int j0, int buf) {
    float* xb = &lds[buf * (WU_XF + WU_DF)];
    if (k < WU_XN) {
      const int pk = xpk[tid + k * WU_THREADS];
      if (pk >= 0) {
        const int hs = min(max(i - 1 + (pk >> 8), 0), H - 1), ws_ = min(max(j0 - 1 + (pk & 255), 0), W - 1);   // clamped: always inside the map
        const unsigned off = ((unsigned)(hs * xs) * (unsigned)(W * xs) + (unsigned)(ws_ * xs)) * (unsigned)Cin + (unsigned)xq4;
        const float* xn = x + (size_t)n * (H * xs) * (W * xs) * Cin + ci0;
        __builtin_amdgcn_global_load_lds(xn + off, xb + (tid + k * WU_THREADS) * 4, 16, 0, 0);
      }
    } else {
      const int u = tid + (k - WU_XN) * WU_THREADS;
      if (u < WU_DU) {
        const int p = u >> 5;
        const int jj = j0 + (p & (WU_PW - 1)), ii = i + (p / WU_PW);
        const bool ok = d_cok && ii < Hc && !(B == 1 && jj == W - 1);                               // (a ragged last row group; the last output column: edge path)
        const unsigned off = ((unsigned)(2 * ii + A) * (unsigned)(2 * W) + (unsigned)(2 * jj + B)) * (unsigned)Cout + (unsigned)(co0 + dq4);
        const float* dn = dy + (size_t)n * (2 * H) * (2 * W) * Cout;
        __builtin_amdgcn_global_load_lds(ok ? dn + off : f32_zero16, xb + WU_XF + u * 4, 16, 0, 0);
      }
    }
  };
  constexpr int NUNITS = WU_XN + WU_DN;

  if (q0 < q1) {
    int n, i, j0;
    decode(q0, n, i, j0);
#pragma unroll
    for (int k = 0; k < NUNITS; ++k) stage_unit(k, n, i, j0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int q = q0; q < q1; ++q) {
    const int buf = (q - q0) & 1;
    int n, i, j0, nn = 0, ni = 0, nj0 = 0;
    decode(q, n, i, j0);
    const bool has_next = q + 1 < q1;
    if (has_next) decode(q + 1, nn, ni, nj0);
    // halo signs: the row above the map (dr = 0 at i = 0) and the column left of it (halo column 0 at j0 = 0) are the NEGATED clamped pixels
    const unsigned rsign0 = (dr == 0 && i == 0) ? 0x80000000u : 0u;   // (first pixel row of the patch only: its dr = 0 tap reads halo row 0)
    const unsigned csign = (j0 == 0 && lh == 0) ? 0x80000000u : 0u;
    const float* base = &lds[buf * (WU_XF + WU_DF)];
    if (active) {
      // software pipeline of depth one: the fragments of k-step ks + 1 are requested BEFORE the MFMAs of k-step ks are issued, so the LDS
      // latency runs behind the wavefront's own 6 (4) MFMAs instead of in front of them
      constexpr int NA = MODEB ? 1 : 3, NB = MODEB ? 4 : 2;
      float ac[NA], bc[NB], an[NA], bn[NB];
      auto load_frag = [&](const int ks, float* af, float* bf) {
        const int arow = (((2 * ks) / WU_PW) * WU_HW + ((2 * ks) & (WU_PW - 1))) * WU_CI;
        if (!MODEB) {
#pragma unroll
          for (int s = B; s < 3; ++s) af[s] = base[a_off + s * WU_CI + arow];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = base[b_off + (njp * 2 + j) * 32 + 2 * ks * WU_CO];
        } else {
          af[0] = base[a_off + sh * WU_CI + arow];
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[j] = base[b_off + j * 32 + 2 * ks * WU_CO];
        }
      };
      // the NEXT patch is requested in one block in front of the k-loop (5 LDS-DMA units of ~15 VALU instructions each since the halo-unit
      // table; with the ~50-instruction index arithmetic of the first version, spreading the units over the loop was the faster order)
      if (has_next) {
#pragma unroll
        for (int k = 0; k < NUNITS; ++k) stage_unit(k, nn, ni, nj0, buf ^ 1);
      }
      load_frag(0, ac, bc);
#pragma unroll 4
      for (int ks = 0; ks < WU_PIX / 2; ++ks) {
        if (ks + 1 < WU_PIX / 2) load_frag(ks + 1, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        const bool col0 = ((2 * ks) & (WU_PW - 1)) == 0;
        const unsigned rsign = (2 * ks < WU_PW) ? rsign0 : 0u;
        if (!MODEB) {
          float a[3];
#pragma unroll
          for (int s = B; s < 3; ++s) {
            const unsigned sg = rsign ^ ((col0 && s == 0) ? csign : 0u);
            a[s] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ac[s]) ^ sg);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = B; s < 3; ++s) acc[j * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bc[j], acc[j * 3 + s], 0, 0, 0);
          if (BIAS && do_bias) {
            bsum[0] += bc[0];
            bsum[1] += bc[1];
          }
        } else {
          const unsigned sg = (col0 && sh == 0) ? csign : 0u;                       // (dr >= 1: never the row above the map)
          const float a = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ac[0]) ^ sg);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bc[j], acc[j], 0, 0, 0);
          if (BIAS && do_bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bsum[j] += bc[j];
          }
        }
#pragma unroll
        for (int i2 = 0; i2 < NA; ++i2) ac[MODEB ? i2 : (i2 < B ? B : i2)] = an[MODEB ? i2 : (i2 < B ? B : i2)];
#pragma unroll
        for (int i2 = 0; i2 < NB; ++i2) bc[i2] = bn[i2];
      }
    } else if (has_next) {
#pragma unroll
      for (int k = 0; k < NUNITS; ++k) stage_unit(k, nn, ni, nj0, buf ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's LDS-DMA pieces of the next patch have landed
    __syncthreads();
  }

  if (!active) return;
  // partial G tiles of this (class, split): out[split][local tap][Cin][Cout] (the class's block of the workspace)
  float* const o0 = out + (size_t)split * WU_NTAPS[CLS] * Cin * Cout;
  if (!MODEB) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nn = co0 + (njp * 2 + j) * 32 + l31;
#pragma unroll
      for (int s = B; s < 3; ++s) {
        float* o = o0 + (size_t)(dr * NS + (s - B)) * Cin * Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = ci0 + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (nn < Cout) o[(size_t)ci * Cout + nn] = acc[j * 3 + s][e];
        }
      }
      if (BIAS && do_bias) {
        const float v = bsum[j] + __shfl_down(bsum[j], 32, 64);   // odd + even pixels of the k-step pairs
        if (lh == 0 && nn < Cout) bias_part[nn] = v;
      }
    }
  } else {
    float* o = o0 + (size_t)((dr - 1) * NS + (sh - B)) * Cin * Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nn = co0 + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (nn < Cout) o[(size_t)ci * Cout + nn] = acc[j][e];
      }
      if (BIAS && do_bias) {
        const float v = bsum[j] + __shfl_down(bsum[j], 32, 64);
        if (lh == 0 && nn < Cout) bias_part[nn] = v;
      }
    }
  }
}

template <int WU_PW, bool BIAS>
__global__ __launch_bounds__(WU_THREADS, 3) void wgrad3x3_up2_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                         float* __restrict__ out, float* __restrict__ bias_part,
                                                                         const int N, const int H, const int W, const int Cin, const int Cout,
                                                                         const WgradUp2Plan p, const int xs) {
  // ONE LDS object (hipcc serialises LDS-DMA against ds_reads when several __shared__ objects exist): [buf][halo | dY]
  __shared__ __attribute__((aligned(16))) float lds[WuGeo<WU_PW>::LDS_FLOATS];   // + the lanes' packed halo units
  // workgroup -> (slab pair, class, split of the class): the classes of a pair are laid out one after another
  const int pair = blockIdx.x / p.total, w = blockIdx.x - pair * p.total;
  const int cls = w >= p.s0[3] ? 3 : (w >= p.s0[2] ? 2 : (w >= p.s0[1] ? 1 : 0));
  const int split = w - p.s0[cls];
  const int ci0 = (pair / p.tiles_co) * WU_CI, co0 = (pair % p.tiles_co) * WU_CO;
  // class block of the workspace: [class][split][tap][Cin][Cout]; bias partials [workgroup of a co tile with ci0 == 0][Cout]
  size_t off = 0;
  for (int c = 0; c < cls; ++c) off += (size_t)p.ns[c] * WU_NTAPS[c];
  float* o = out + off * Cin * Cout;
  float* bp = bias_part != nullptr ? bias_part + (size_t)w * Cout : nullptr;
  switch (cls) {
    case 0: wgrad_up2_body<0, WU_PW, BIAS>(lds, x, dy, o, bp, N, H, W, Cin, Cout, ci0, co0, split, p.ns[0], xs); break;
    case 1: wgrad_up2_body<1, WU_PW, BIAS>(lds, x, dy, o, bp, N, H, W, Cin, Cout, ci0, co0, split, p.ns[1], xs); break;
    case 2: wgrad_up2_body<2, WU_PW, BIAS>(lds, x, dy, o, bp, N, H, W, Cin, Cout, ci0, co0, split, p.ns[2], xs); break;
    default: wgrad_up2_body<3, WU_PW, BIAS>(lds, x, dy, o, bp, N, H, W, Cin, Cout, ci0, co0, split, p.ns[3], xs); break;
  }
}

// edge operands: the 1-D upsampled last row / column of x (float4 units) and the matching lines of dy (the corner belongs to the row)
__global__ __launch_bounds__(256) void wgrad_up2_edge_operands_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      float* __restrict__ u_row, float* __restrict__ v_col,
                                                                      float* __restrict__ dy_row, float* __restrict__ dy_col, int N, int H, int W,
                                                                      int Cin, int Cout, int xs) {
  const int CV = Cin >> 2, DV = Cout >> 2;
  const long n_row = (long)N * 2 * W * CV, n_col = (long)N * 2 * H * CV, d_row = (long)N * 2 * W * DV, d_col = (long)N * 2 * H * DV;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n_row + n_col + d_row + d_col; t += (long)gridDim.x * 256) {
    if (t < n_row + n_col) {
      const bool row = t < n_row;
      const long j = row ? t : t - n_row;
      const int L = row ? W : H;
      const int cv = (int)(j % CV), q = (int)((j / CV) % (2 * L)), n = (int)(j / ((long)CV * 2 * L));
      const int lo = q >> 1, hi = min(lo + 1, L - 1);
      const float4* base = reinterpret_cast<const float4*>(row ? x + (((long)n * (H * xs) + (H - 1) * xs) * (W * xs)) * Cin
                                                               : x + (((long)n * (H * xs)) * (W * xs) + (W - 1) * xs) * Cin) + cv;
      const long stride = row ? (long)xs * CV : (long)xs * (W * xs) * CV;
      const float4 xl = base[lo * stride];
      float4 v = xl;
      if (q & 1) {                                                // the arithmetic of the resize kernel (lerp, weight 1/2)
        const float4 xh = base[hi * stride];
        v = make_float4(xl.x + (xh.x - xl.x) * 0.5f, xl.y + (xh.y - xl.y) * 0.5f, xl.z + (xh.z - xl.z) * 0.5f, xl.w + (xh.w - xl.w) * 0.5f);
      }
      reinterpret_cast<float4*>(row ? u_row : v_col)[j] = v;
    } else {
      const long j0 = t - n_row - n_col;
      const bool row = j0 < d_row;
      const long j = row ? j0 : j0 - d_row;
      const int L2 = row ? 2 * W : 2 * H;
      const int cv = (int)(j % DV), q = (int)((j / DV) % L2), n = (int)(j / ((long)DV * L2));
      const long pix = row ? ((long)n * 2 * H + 2 * H - 1) * 2 * W + q : ((long)n * 2 * H + q) * 2 * W + 2 * W - 1;
      float4 v = reinterpret_cast<const float4*>(dy)[pix * DV + cv];
      if (!row && q == 2 * H - 1) v = make_float4(0.f, 0.f, 0.f, 0.f);     // the corner is the row's
      reinterpret_cast<float4*>(row ? dy_row : dy_col)[j] = v;
    }
  }
}

// dw[r][s] = sum_{a,b,dr,dc} A_a[dr][r] A_b[dc][s] G_ab[dr][dc] + [r < 2] E_row[s] + [s < 2] E_col[r];  G [25][Cin][Cout] (classes 9 | 6 | 6 | 4,
// taps in (dr, dc) order), E_row [3][Cin][Cout], E_col [3][Cin][Cout]; one thread per (ci, co quad)
__global__ __launch_bounds__(256) void wgrad_up2_combine_kernel(const float* __restrict__ G, const float* __restrict__ e_row,
                                                                const float* __restrict__ e_col, float* __restrict__ dw, size_t kn4) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= kn4) return;
  const float A0[3][3] = {{0.5f, 0.f, 0.f}, {0.5f, 1.f, 0.5f}, {0.f, 0.f, 0.5f}}, A1[3][3] = {{0.f, 0.f, 0.f}, {1.f, 0.5f, 0.f}, {0.f, 0.5f, 1.f}};
  float4 o[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int combo = 0;
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int a = cls >> 1, b = cls & 1;
#pragma unroll
    for (int dr = 0; dr < 3; ++dr)
#pragma unroll
      for (int dc = 0; dc < 3; ++dc) {
        if ((a == 1 && dr == 0) || (b == 1 && dc == 0)) continue;
        const float4 g = reinterpret_cast<const float4*>(G)[(size_t)combo * kn4 + t];
        ++combo;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float ar = a ? A1[dr][r] : A0[dr][r];
          if (ar == 0.f) continue;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const float c = ar * (b ? A1[dc][s] : A0[dc][s]);
            if (c == 0.f) continue;
            float4& q = o[r * 3 + s];
            q.x += c * g.x; q.y += c * g.y; q.z += c * g.z; q.w += c * g.w;
          }
        }
      }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float4 q = o[r * 3 + s];
      if (r < 2) { const float4 e = reinterpret_cast<const float4*>(e_row)[(size_t)s * kn4 + t]; q.x += e.x; q.y += e.y; q.z += e.z; q.w += e.w; }
      if (s < 2) { const float4 e = reinterpret_cast<const float4*>(e_col)[(size_t)r * kn4 + t]; q.x += e.x; q.y += e.y; q.z += e.z; q.w += e.w; }
      reinterpret_cast<float4*>(dw)[(size_t)(r * 3 + s) * kn4 + t] = q;
    }
}

__global__ void wgrad_up2_bias_kernel(const float* __restrict__ db_main, const float* __restrict__ db_row, const float* __restrict__ db_col,
                                      float* __restrict__ db, int Cout) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < Cout) db[c] = db_main[c] + db_row[c] + db_col[c];
}

// ---- exact border lines of the fused low-resolution backward-data (ladder_conv3x3_up2_bwd_borders) -------------------------------------
// Per axis the exact transpose of the legacy factor-2 resize is (E v)[p] = v[2p] + v[2p-1] / 2 [p >= 1] + v[2p+1] / 2 [p <= L-2] + v[2L-1] [p = L-1],
// while the 5x5 / stride-2 correlation of the main launch applies (M v)[p] = v[2p] + v[2p-1] / 2 + v[2p+1] / 2 to the backward-data d_up of
// the ZERO-PADDED dy on the unbounded grid: M = E + D with (D v)[p] = v[-1] / 2 [p = 0] - v[2L-1] / 2 [p = L-1].  Hence
//     exact = (M_r - D_r) (x) (M_c - D_c) d_up = main - D_r (x) M_c - M_r (x) D_c + D_r (x) D_c:
// the correction needs ONE line of d_up per border -- row -1 (a 1x3 correlation of dy row 0 with the taps w[0][.]), row 2H-1 (2x3 taps
// w[1..2][.] over dy rows 2H-2, 2H-1), column -1 (3x1, w[.][0]) and column 2W-1 (3x2, w[.][1..2]), each on the range [-1, 2L-1] of the other
// axis -- instead of the 2-3 full lines with all 9 taps the strip path recomputes (45 -> 9 line-taps per axis).
// operands: the four small banks in forward-convolution layout [KH][KW][C][Cout] (K[a][b] = w[2-a][2-b] restricted to the rows / columns
// listed) and contiguous copies of the dy lines they read
__global__ __launch_bounds__(256) void up2_border_operands_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ k_top,
                                                                  float* __restrict__ k_bot, float* __restrict__ k_left, float* __restrict__ k_right,
                                                                  float* __restrict__ in_top, float* __restrict__ in_bot, float* __restrict__ in_left,
                                                                  float* __restrict__ in_right, int N, int H, int W, int C, int Cout) {
  const long kn = (long)C * Cout, DV = C >> 2;
  const long n_k = 18 * kn;                                                       // 3 + 6 + 3 + 6 tap matrices
  const long n_top = (long)N * 2 * W * DV, n_bot = 2 * n_top, n_left = (long)N * 2 * H * DV, n_right = 2 * n_left;
  const float4* dy4 = reinterpret_cast<const float4*>(dy);
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n_k + n_top + n_bot + n_left + n_right; t += (long)gridDim.x * 256) {
    if (t < n_k) {
      const int tap = (int)(t / kn);
      const long e = t - (long)tap * kn;
      const int co = (int)(e / Cout), ci = (int)(e - (long)co * Cout);             // bank element [tap][co (dy channel)][ci (dx channel)]
      int r, sx;
      float* dst;
      if (tap < 3) { r = 0; sx = 2 - tap; dst = k_top + (long)tap * kn; }                                          // [1][3]: b = tap
      else if (tap < 9) { const int a = (tap - 3) / 3, b = (tap - 3) % 3; r = 2 - a; sx = 2 - b; dst = k_bot + (long)(tap - 3) * kn; }      // [2][3]
      else if (tap < 12) { const int a = tap - 9; r = 2 - a; sx = 0; dst = k_left + (long)a * kn; }                // [3][1]
      else { const int a = (tap - 12) / 2, b = (tap - 12) % 2; r = 2 - a; sx = 2 - b; dst = k_right + (long)(tap - 12) * kn; }             // [3][2]
      dst[e] = w[(((long)r * 3 + sx) * Cout + ci) * C + co];                       // w = the layer's HWIO bank [3][3][Cout (its input)][C (its output)]
    } else {
      long j = t - n_k;
      if (j < n_top) {                                                             // dy row 0: [N][1][2W][C]
        const long cv = j % DV, x_ = (j / DV) % (2 * W), n = j / (DV * 2 * W);
        reinterpret_cast<float4*>(in_top)[j] = dy4[(((n * 2 * H + 0) * 2 * W) + x_) * DV + cv];
      } else if ((j -= n_top) < n_bot) {                                           // dy rows 2H-2, 2H-1: [N][2][2W][C]
        const long cv = j % DV, x_ = (j / DV) % (2 * W), a = (j / (DV * 2 * W)) % 2, n = j / (DV * 2 * W * 2);
        reinterpret_cast<float4*>(in_bot)[j] = dy4[(((n * 2 * H + 2 * H - 2 + a) * 2 * W) + x_) * DV + cv];
      } else if ((j -= n_bot) < n_left) {                                          // dy column 0: [N][2H][1][C]
        const long cv = j % DV, y_ = (j / DV) % (2 * H), n = j / (DV * 2 * H);
        reinterpret_cast<float4*>(in_left)[j] = dy4[(((n * 2 * H + y_) * 2 * W) + 0) * DV + cv];
      } else {                                                                     // dy columns 2W-2, 2W-1: [N][2H][2][C]
        j -= n_left;
        const long cv = j % DV, b = (j / DV) % 2, y_ = (j / (DV * 2)) % (2 * H), n = j / (DV * 2 * 2 * H);
        reinterpret_cast<float4*>(in_right)[j] = dy4[(((n * 2 * H + y_) * 2 * W) + 2 * W - 2 + b) * DV + cv];
      }
    }
  }
}

// dx[border] += - D_r (x) M_c - M_r (x) D_c + D_r (x) D_c from the four d_up lines: rt / rb [N][2W+1][Cout] (index u = t + 1, t = -1 .. 2W-1),
// cl / cr [N][2H+1][Cout] (index q + 1).  Row threads own the corners; column threads cover i = 1 .. H-2.
__global__ __launch_bounds__(256) void up2_border_fixup_kernel(const float* __restrict__ rt, const float* __restrict__ rb, const float* __restrict__ cl,
                                                               const float* __restrict__ cr, float* __restrict__ dx, int N, int H, int W, int Cout,
                                                               const float* __restrict__ gate, int gate_act) {
  const int CV = Cout >> 2;
  const long n_rows = (long)N * 2 * W * CV, n_cols = (long)N * 2 * (H - 2) * CV;
  const float4* rt4 = reinterpret_cast<const float4*>(rt);
  const float4* rb4 = reinterpret_cast<const float4*>(rb);
  const float4* cl4 = reinterpret_cast<const float4*>(cl);
  const float4* cr4 = reinterpret_cast<const float4*>(cr);
  auto M = [&](const float4* v, long base, int p) -> float4 {                      // (M v)[p] = v[2p] + v[2p-1] / 2 + v[2p+1] / 2 with index + 1
    const float4 a = v[base + (long)(2 * p) * CV], b = v[base + (long)(2 * p + 1) * CV], c = v[base + (long)(2 * p + 2) * CV];
    return make_float4(b.x + 0.5f * (a.x + c.x), b.y + 0.5f * (a.y + c.y), b.z + 0.5f * (a.z + c.z), b.w + 0.5f * (a.w + c.w));
  };
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n_rows + n_cols; t += (long)gridDim.x * 256) {
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    long o;
    if (t < n_rows) {
      const int cv = (int)(t % CV), j = (int)((t / CV) % W), bot = (int)((t / ((long)CV * W)) % 2), n = (int)(t / ((long)CV * W * 2));
      const long lb = (long)n * (2 * W + 1) * CV + cv, cb = (long)n * (2 * H + 1) * CV + cv;
      const float4 m = M(bot ? rb4 : rt4, lb, j);
      const float sg = bot ? 0.5f : -0.5f;                                         // - D_r (x) M_c: -1/2 M_c(row -1) at p = 0, +1/2 M_c(row 2H-1) at p = H-1
      d = make_float4(sg * m.x, sg * m.y, sg * m.z, sg * m.w);
      const int i = bot ? H - 1 : 0;
      if (j == 0 || j == W - 1) {                                                  // corner: also - M_r (x) D_c and + D_r (x) D_c
        const float4 mc = M(j == 0 ? cl4 : cr4, cb, i);
        const float sc = j == 0 ? -0.5f : 0.5f;
        const float4 q = (bot ? rb4 : rt4)[lb + (long)(j == 0 ? 0 : 2 * W) * CV];
        const float s3 = ((j == 0) == (bot == 0)) ? 0.25f : -0.25f;               // +1/4 at (0,0) and (H-1,W-1), -1/4 at the other two
        d = make_float4(d.x + sc * mc.x + s3 * q.x, d.y + sc * mc.y + s3 * q.y, d.z + sc * mc.z + s3 * q.z, d.w + sc * mc.w + s3 * q.w);
      }
      o = (((long)n * H + i) * W + j) * CV + cv;
    } else {
      const long u = t - n_rows;
      const int cv = (int)(u % CV), i = 1 + (int)((u / CV) % (H - 2)), right = (int)((u / ((long)CV * (H - 2))) % 2), n = (int)(u / ((long)CV * (H - 2) * 2));
      const long cb = (long)n * (2 * H + 1) * CV + cv;
      const float4 m = M(right ? cr4 : cl4, cb, i);
      const float sg = right ? 0.5f : -0.5f;
      d = make_float4(sg * m.x, sg * m.y, sg * m.z, sg * m.w);
      o = (((long)n * H + i) * W + (right ? W - 1 : 0)) * CV + cv;
    }
    if (gate != nullptr) {                                          // the main launch applied act'(gate) to its value: the correction carries the same factor
      const float4 g = reinterpret_cast<const float4*>(gate)[o];
      d.x *= ladder_act_grad_from_out(g.x, gate_act); d.y *= ladder_act_grad_from_out(g.y, gate_act);
      d.z *= ladder_act_grad_from_out(g.z, gate_act); d.w *= ladder_act_grad_from_out(g.w, gate_act);
    }
    float4 v = reinterpret_cast<float4*>(dx)[o];
    v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
    reinterpret_cast<float4*>(dx)[o] = v;
  }
}

WgradUp2Plan plan_wgrad_up2(int N, int H, int W, int Cin, int Cout) {
  WgradUp2Plan p{};
  p.pw = (W % 32) == 0 ? 32 : ((W % 16) == 0 ? 16 : 8);                                // patch width: 2 x 32, 4 x 16 or 8 x 8 pixels
  const int ph = WU_PIX / p.pw;
  p.ok = N > 0 && H >= 2 && (W % p.pw) == 0 && (Cin % WU_CI) == 0 && (Cout % 4) == 0 && Cout >= 64 &&
         (long)N * ((H + ph - 1) / ph) * (W / p.pw) >= (p.pw == 32 ? 1024 : 128);
  if (!p.ok) return p;
  p.tiles_ci = Cin / WU_CI;
  p.tiles_co = (Cout + WU_CO - 1) / WU_CO;
  const long pairs = (long)p.tiles_ci * p.tiles_co;
  // two rounds of the chip (one workgroup per CU: 84 KB of LDS), shared between the classes in proportion to their cost per patch (MFMA
  // slots per SIMD and k-step: 18 / 12 / 12 / 8 -- the busiest SIMD of class 3 still issues 2 wavefronts x 4)
  static const int rounds_env = getenv("LADDER_WU_WGS") ? atoi(getenv("LADDER_WU_WGS")) : 0;      // (experiment switch: workgroups per launch)
  long per_pair = (rounds_env > 0 ? rounds_env : 512) / pairs;
  if (per_pair < 8) per_pair = 8;
  // measured per-patch time of the classes (2 x 32-pixel patches, one class at a time on the chip): 21.0 / 14.6 / 15.6 / 10.9 us -- the MFMA
  // time (16.0 / 10.7 / 10.7 / 7.1 us) + ~4-5 us per patch that the direct kernel pays as well (its matrix pipe is 76 % busy); 512 or
  // 2048 workgroups per launch measure the same
  const int cost[4] = {210, 146, 156, 109};
  const long g0 = (H + ph - 1) / ph, g1 = (H - 1 + ph - 1) / ph;                      // row groups of the classes a = 0 / a = 1
  const long q[4] = {(long)N * g0 * (W / p.pw), (long)N * g0 * (W / p.pw), (long)N * g1 * (W / p.pw), (long)N * g1 * (W / p.pw)};
  double tot = 0;
  for (int c = 0; c < 4; ++c) tot += (double)cost[c] * q[c];
  int used = 0;
  for (int c = 0; c < 4; ++c) {
    long s = (long)(per_pair * ((double)cost[c] * q[c] / tot) + 0.5);
    if (s < 1) s = 1;
    if (s > q[c]) s = q[c];
    p.ns[c] = (int)s;
    p.s0[c] = used;
    used += (int)s;
  }
  p.total = used;
  return p;
}

static size_t wu_align(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int ladder_conv3x3_up2_wgrad_eligible(int N, int H, int W, int Cin, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_UP2") != nullptr;
  return (!off && plan_wgrad_up2(N, H, W, Cin, Cout).ok) ? 1 : 0;
}

size_t ladder_conv3x3_up2_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  const WgradUp2Plan p = plan_wgrad_up2(N, H, W, Cin, Cout);
  if (!p.ok) return 0;
  const size_t kn = (size_t)Cin * Cout;
  size_t tiles = 0;
  for (int c = 0; c < 4; ++c) tiles += (size_t)p.ns[c] * WU_NTAPS[c];
  const size_t mr = (size_t)N * 2 * W, mc = (size_t)N * 2 * H;
  const size_t g1 = ladder_conv2d_bwd_filter_workspace_bytes(N, 1, 2 * W, Cin, 1, 2 * W, Cout, 1, 3);
  const size_t g2 = ladder_conv2d_bwd_filter_workspace_bytes(N, 2 * H, 1, Cin, 2 * H, 1, Cout, 3, 1);
  return wu_align(tiles * kn * 4) + wu_align((size_t)(p.total + 1) * Cout * 4) + wu_align(25 * kn * 4) + 2 * wu_align(3 * kn * 4) + 2 * wu_align((size_t)Cout * 4) +
         wu_align(mr * Cin * 4) + wu_align(mc * Cin * 4) + wu_align(mr * Cout * 4) + wu_align(mc * Cout * 4) + wu_align(g1 > g2 ? g1 : g2) + 512;
}

// dw [3][3][Cin][Cout] (and db [Cout], may be NULL) of y = conv3x3_same(resize2x(x), w) from the LOW-resolution x [N, H, W, Cin]
// (x_upsampled != 0: x points at the materialised upsample [N, 2H, 2W, Cin], read at its even rows / columns) and dy [N, 2H, 2W, Cout].
int ladder_conv3x3_up2_wgrad(const float* x, int x_upsampled, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                             void* ws, size_t ws_bytes, ladder_stream_t stream) {
  const WgradUp2Plan p = plan_wgrad_up2(N, H, W, Cin, Cout);
  if (!p.ok) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(dy) || !ladder_aligned16(dw)) return LADDER_E_ALIGN;
  if (ws == nullptr || ws_bytes < ladder_conv3x3_up2_wgrad_workspace_bytes(N, H, W, Cin, Cout)) return LADDER_E_WORKSPACE;
  const size_t kn = (size_t)Cin * Cout;
  size_t tiles = 0;
  for (int c = 0; c < 4; ++c) tiles += (size_t)p.ns[c] * WU_NTAPS[c];
  const size_t mr = (size_t)N * 2 * W, mc = (size_t)N * 2 * H;
  char* q = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* part = (float*)q; q += wu_align(tiles * kn * 4);
  const int nbias = p.total;                              // bias partials: one row [Cout] per class split (written by its ci0 == 0 workgroups, one per co tile)
  float* bias_part = (float*)q; q += wu_align((size_t)(nbias + 1) * Cout * 4);
  float* db_main = bias_part + (size_t)nbias * Cout;
  float* G = (float*)q; q += wu_align(25 * kn * 4);
  float* e_row = (float*)q; q += wu_align(3 * kn * 4);
  float* e_col = (float*)q; q += wu_align(3 * kn * 4);
  float* db_row = (float*)q; q += wu_align((size_t)Cout * 4);
  float* db_col = (float*)q; q += wu_align((size_t)Cout * 4);
  float* u_row = (float*)q; q += wu_align(mr * Cin * 4);
  float* v_col = (float*)q; q += wu_align(mc * Cin * 4);
  float* dy_row = (float*)q; q += wu_align(mr * Cout * 4);
  float* dy_col = (float*)q; q += wu_align(mc * Cout * 4);
  const size_t g = ws_bytes - (size_t)(q - (char*)ws);
  const int xs = x_upsampled ? 2 : 1;
  if (db != nullptr && hipMemsetAsync(bias_part, 0, (size_t)nbias * Cout * 4, stream) != hipSuccess) return LADDER_E_LAUNCH;
  // bias partial rows: the kernel indexes them by the workgroup's position inside its pair (w) -- pairs with ci0 == 0 are the first tiles_co
#define LADDER_WU_LAUNCH(PW_) \
  do { \
    if (db != nullptr) hipLaunchKernelGGL((wgrad3x3_up2_f32_kernel<PW_, true>), dim3(p.tiles_ci * p.tiles_co * p.total), dim3(WU_THREADS), 0, stream, x, dy, part, bias_part, N, H, W, Cin, Cout, p, xs); \
    else hipLaunchKernelGGL((wgrad3x3_up2_f32_kernel<PW_, false>), dim3(p.tiles_ci * p.tiles_co * p.total), dim3(WU_THREADS), 0, stream, x, dy, part, (float*)nullptr, N, H, W, Cin, Cout, p, xs); \
  } while (0)
  if (p.pw == 32) LADDER_WU_LAUNCH(32);
  else if (p.pw == 16) LADDER_WU_LAUNCH(16);
  else LADDER_WU_LAUNCH(8);
#undef LADDER_WU_LAUNCH
  size_t off = 0;
  for (int c = 0; c < 4; ++c) {                            // G[class taps] = sum over the class's splits (fixed order)
    const int rc = ladder_reduce_splits(part + off * kn, G + (size_t)WU_TAP0[c] * kn, p.ns[c], (size_t)WU_NTAPS[c] * kn, stream);
    if (rc != LADDER_OK) return rc;
    off += (size_t)p.ns[c] * WU_NTAPS[c];
  }
  hipLaunchKernelGGL(wgrad_up2_edge_operands_kernel, dim3(1024), dim3(256), 0, stream, x, dy, u_row, v_col, dy_row, dy_col, N, H, W, Cin, Cout, xs);
  int rc = ladder_conv2d_bwd_filter(u_row, dy_row, e_row, db_row, N, 1, 2 * W, Cin, 1, 2 * W, Cout, 1, 3, 1, 0, 1, q, g, stream);
  if (rc != LADDER_OK) return rc;
  rc = ladder_conv2d_bwd_filter(v_col, dy_col, e_col, db_col, N, 2 * H, 1, Cin, 2 * H, 1, Cout, 3, 1, 1, 1, 0, q, g, stream);
  if (rc != LADDER_OK) return rc;
  hipLaunchKernelGGL(wgrad_up2_combine_kernel, dim3((unsigned)((kn / 4 + 255) / 256)), dim3(256), 0, stream, (const float*)G, (const float*)e_row,
                     (const float*)e_col, dw, kn / 4);
  if (db != nullptr) {
    rc = ladder_reduce_splits(bias_part, db_main, nbias, (size_t)Cout, stream);
    if (rc != LADDER_OK) return rc;
    hipLaunchKernelGGL(wgrad_up2_bias_kernel, dim3((Cout + 255) / 256), dim3(256), 0, stream, (const float*)db_main, (const float*)db_row,
                       (const float*)db_col, db, Cout);
  }
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}


// ---- exact border lines of ladder_conv3x3_up2_bwd_data_split's result (strict fp32) ----------------------------------------------------------
static size_t up2b_part(size_t n) { return wu_align(n * 4); }

size_t ladder_conv3x3_up2_bwd_borders_workspace_bytes(int N, int H, int W, int C, int Cout) {
  if (N <= 0 || H < 3 || W < 3 || C <= 0 || Cout <= 0 || (C % 4) != 0 || (Cout % 4) != 0) return 0;
  const size_t kn = (size_t)C * Cout, lw = (size_t)N * 2 * W, lh = (size_t)N * 2 * H;
  size_t g = 0;
  const size_t gs[4] = {ladder_igemm_fwd_workspace_bytes((long)N * (2 * W + 1), 3 * C, Cout), ladder_igemm_fwd_workspace_bytes((long)N * (2 * W + 1), 6 * C, Cout),
                        ladder_igemm_fwd_workspace_bytes((long)N * (2 * H + 1), 3 * C, Cout), ladder_igemm_fwd_workspace_bytes((long)N * (2 * H + 1), 6 * C, Cout)};
  for (size_t v : gs) g = v > g ? v : g;
  return up2b_part(18 * kn) + up2b_part(lw * C) + up2b_part(2 * lw * C) + up2b_part(lh * C) + up2b_part(2 * lh * C) +
         2 * up2b_part((size_t)N * (2 * W + 1) * Cout) + 2 * up2b_part((size_t)N * (2 * H + 1) * Cout) + wu_align(g) + 512;
}

// dx [N, H, W, Cout] = the result of ladder_conv3x3_up2_bwd_data_split(dy [N, 2H, 2W, C], ...): its four border lines are made exact IN PLACE
// (w = the layer's HWIO bank [3][3][Cout][C]).  Must follow that launch on the same stream.
static int up2_bwd_borders_impl(const float* dy, const float* w, float* dx, const float* gate, int gate_act, int N, int H, int W, int C, int Cout, void* ws,
                                size_t ws_bytes, ladder_stream_t stream) {
  const size_t need = ladder_conv3x3_up2_bwd_borders_workspace_bytes(N, H, W, C, Cout);
  if (need == 0) return LADDER_E_SHAPE;
  if (!ladder_aligned16(dy) || !ladder_aligned16(w) || !ladder_aligned16(dx)) return LADDER_E_ALIGN;
  if (ws == nullptr || ws_bytes < need) return LADDER_E_WORKSPACE;
  const size_t kn = (size_t)C * Cout, lw = (size_t)N * 2 * W, lh = (size_t)N * 2 * H;
  char* q = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* k_top = (float*)q;
  float* k_bot = k_top + 3 * kn;
  float* k_left = k_bot + 6 * kn;
  float* k_right = k_left + 3 * kn;
  q += up2b_part(18 * kn);
  float* in_top = (float*)q; q += up2b_part(lw * C);
  float* in_bot = (float*)q; q += up2b_part(2 * lw * C);
  float* in_left = (float*)q; q += up2b_part(lh * C);
  float* in_right = (float*)q; q += up2b_part(2 * lh * C);
  float* rt = (float*)q; q += up2b_part((size_t)N * (2 * W + 1) * Cout);
  float* rb = (float*)q; q += up2b_part((size_t)N * (2 * W + 1) * Cout);
  float* cl = (float*)q; q += up2b_part((size_t)N * (2 * H + 1) * Cout);
  float* cr = (float*)q; q += up2b_part((size_t)N * (2 * H + 1) * Cout);
  const size_t g = ws_bytes - (size_t)(q - (char*)ws);
  hipLaunchKernelGGL(up2_border_operands_kernel, dim3(1024), dim3(256), 0, stream, dy, w, k_top, k_bot, k_left, k_right, in_top, in_bot, in_left, in_right,
                     N, H, W, C, Cout);
  // the four d_up lines as small forward convolutions on the fp32 matrix cores (zero padding supplies the out-of-range taps)
  int rc = ladder_conv2d_fwd(in_top, k_top, nullptr, rt, N, 1, 2 * W, C, 1, 2 * W + 1, Cout, 1, 3, 1, 0, 2, LADDER_ACT_NONE, q, g, stream);
  if (rc != LADDER_OK) return rc;
  rc = ladder_conv2d_fwd(in_bot, k_bot, nullptr, rb, N, 2, 2 * W, C, 1, 2 * W + 1, Cout, 2, 3, 1, 0, 2, LADDER_ACT_NONE, q, g, stream);
  if (rc != LADDER_OK) return rc;
  rc = ladder_conv2d_fwd(in_left, k_left, nullptr, cl, N, 2 * H, 1, C, 2 * H + 1, 1, Cout, 3, 1, 1, 2, 0, LADDER_ACT_NONE, q, g, stream);
  if (rc != LADDER_OK) return rc;
  rc = ladder_conv2d_fwd(in_right, k_right, nullptr, cr, N, 2 * H, 2, C, 2 * H + 1, 1, Cout, 3, 2, 1, 2, 0, LADDER_ACT_NONE, q, g, stream);
  if (rc != LADDER_OK) return rc;
  hipLaunchKernelGGL(up2_border_fixup_kernel, dim3(512), dim3(256), 0, stream, (const float*)rt, (const float*)rb, (const float*)cl, (const float*)cr, dx,
                     N, H, W, Cout, gate, gate_act);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

int ladder_conv3x3_up2_bwd_borders(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int Cout, void* ws, size_t ws_bytes,
                                   ladder_stream_t stream) {
  return up2_bwd_borders_impl(dy, w, dx, nullptr, 0, N, H, W, C, Cout, ws, ws_bytes, stream);
}

// ---- the gated forms (round 5): dx = act'(gate) * d loss / d x_lo, gate [N, H, W, Cout] = the ACTIVATED tensor the low-resolution input is -- the
// activation backward of the layer below an un-normalised resize -> conv pair (decoder conv2d_5 -> conv2d_6, codes/models.py:556-564) rides on the
// epilogue of the main launch and on the border fix-up; strict fp32, the 8x32-pixel tiling
int ladder_conv3x3_up2_bwd_data_gated_f32_eligible(int N, int H, int W, int C, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_UP2") != nullptr;
  return (!off && (C % 16) == 0 && conv3x3_f32_big_ok(N, H, W, 4 * C, Cout, 4)) ? 1 : 0;
}

int ladder_conv3x3_up2_bwd_data_gated_f32(const float* dy, const void* bank_up2t, float* dx, const float* gate, int gate_act, int N, int H, int W, int C,
                                          int Cout, ladder_stream_t stream) {
  if (!ladder_conv3x3_up2_bwd_data_gated_f32_eligible(N, H, W, C, Cout) || gate == nullptr || !ladder_aligned16(gate)) return LADDER_E_SHAPE;
  return conv3x3_f32_launch(dy, (const float*)bank_up2t, nullptr, dx, gate, nullptr, nullptr, gate_act, N, H, W, 4 * C, Cout, LADDER_ACT_NONE, stream,
                            filter_bank_tap_masks(4), 4);
}

int ladder_conv3x3_up2_bwd_borders_gated(const float* dy, const float* w, float* dx, const float* gate, int gate_act, int N, int H, int W, int C, int Cout,
                                         void* ws, size_t ws_bytes, ladder_stream_t stream) {
  if (gate == nullptr || !ladder_aligned16(gate)) return LADDER_E_SHAPE;
  return up2_bwd_borders_impl(dy, w, dx, gate, gate_act, N, H, W, C, Cout, ws, ws_bytes, stream);
}

}  // extern "C"
