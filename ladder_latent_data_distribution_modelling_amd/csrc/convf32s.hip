// Strict-fp32 3x3 halo convolutions on SMALL maps (gfx950, MI355X; round 5): the tap-masked LDS-halo kernel of convf32.hip re-tiled for
// the 16- and 8-pixel-wide maps of the decoder's inner layers and the encoder's deep layers, where the 8x32-pixel x 128-channel tile of
// conv3x3_halo_f32_kernel either does not divide the map or leaves most of the 256 CUs without a workgroup:
//
//   * resize x2 -> 3x3 conv as ONE convolution over the low-resolution map for decoder conv2d_5 (16x16 -> 32x32, 256 -> 256 channels) and
//     conv2d_4 (8x8 -> 16x16, 512 -> 256) (reference codes/models.py:544-560): forward (orientation 3 of filterbank.h, class-interleaved
//     epilogue for classes of ANY width that is a multiple of the channel tile) and backward-data (orientation 4);
//   * the plain 3x3 / SAME convolution and its backward-data on 16x16 / 8x8 maps (decoder conv2d_3, codes/models.py:539-543);
//   * a 3x3 / stride-2 / SAME convolution on an even map as a stride-1 correlation over the four pixel-parity classes of its input
//     (orientation 5; encoder conv2d_1 ... conv2d_3, codes/models.py:409-439) and its backward-data as one class-structured launch
//     (orientation 2) for class widths other than 128.
//
// Tiling: a workgroup of 8 wavefronts (4 along pixels x 2 along channels) computes MI x 4 pixel tiles of 32 pixels by NI x 2 channel tiles
// of 32 channels; a 32-pixel MFMA tile is 2 rows x 16 or 4 rows x 8 pixels of a SUB-PATCH (16x16, 8x16 or 8x8 pixels with its own 1-pixel
// halo), and a workgroup carries 1, 2 or 4 sub-patches (consecutive images of an 8x8 map).  Per FKS-channel input slab the halos are staged
// in LDS once for all taps (pixel stride FKS + 1 floats), the [FKS x 64 NI] filter slab of the next issued tap is double-buffered, and a tile
// walks only the taps of its class (scalar tap mask).  Class launches give every workgroup the same work: a workgroup computes the class
// PAIR (0, 3) or (1, 2) of its tile one after the other (9 + 4 / 6 + 6 taps for the upsample-fused forward, 4 + 1 / 2 + 2 for a stride-2
// backward-data).
#include <cstdlib>
#include "convf32.h"
#include "filterbank.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FS_THREADS = 512;

__device__ __attribute__((aligned(16))) float f32s_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// sub-patch geometries: TW x SH pixels, NIMG sub-patches per workgroup; MI = NIMG * SH * TW / 128 pixel tiles per wavefront
template <int GEO> struct SGeo;
template <> struct SGeo<0> { static constexpr int TW = 16, SH = 16, NIMG = 1; };   // 256 pixels: one 16x16 sub-patch
template <> struct SGeo<1> { static constexpr int TW = 16, SH = 8, NIMG = 1; };    // 128 pixels: 8 rows x 16
template <> struct SGeo<2> { static constexpr int TW = 8, SH = 8, NIMG = 4; };     // 256 pixels: four 8x8 sub-patches
template <> struct SGeo<3> { static constexpr int TW = 8, SH = 8, NIMG = 2; };     // 128 pixels: two 8x8 sub-patches

// UPM 0: zero halo; 1: upsample-fused forward (signed clamped halo); 2: the input is [N, 2H, 2W, Cin / 4] and the input slabs are grouped
// by pixel-parity class (backward-data of the upsample-fused pair, s2_out 4; stride-2 forward, s2_out 5).
// s2_out & 0xff: 0 plain output; 1 / 2 / 3 class-interleaved output [N, 2H, 2W, creal] (3: x = the even sub-grid of an upsampled tensor);
// 4 / 5 plain output.  Bit 9: class PAIRS per workgroup (class launches).
template <int UPM, int GEO, int NI, int FKS, int TPS>
__global__ __launch_bounds__(FS_THREADS, 4) void conv3x3_halo_f32s_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                                          const int N, const int H, const int W, const int Cin,
                                                                          const int Cout, const int act, const int tiles_n,
                                                                          const unsigned long long tap_masks, const int s2_out,
                                                                          const int creal) {
  using G = SGeo<GEO>;
  constexpr int TW = G::TW, SH = G::SH, NIMG = G::NIMG, PW = TW + 2, PH = SH + 2, SPIX = PH * PW, NPIX = NIMG * SPIX, LDA = FKS + 1;
  constexpr int MI = NIMG * SH * TW / 128, MPS = SH * TW / 32, RPM = 32 / TW;        // pixel tiles per wavefront / per sub-patch, rows per tile
  constexpr int BN = 64 * NI;
  constexpr int KQ = FKS / 4, UNITS = NPIX * KQ, AU = (UNITS + FS_THREADS - 1) / FS_THREADS, PPI = FS_THREADS / KQ;
  constexpr int BU = FKS * BN / 4, BUN = (BU + FS_THREADS - 1) / FS_THREADS, BQ = BN / 4;
  static_assert(MI == 1 || MI == 2, "pixel tiles per wavefront");
  __shared__ float Ah[2][NPIX * LDA];
  __shared__ __attribute__((aligned(16))) float Bh[2][TPS * FKS * BN];       // TPS taps per barrier step

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int s2_mode = s2_out & 0xff;
  const bool cls_mode = s2_mode >= 1 && s2_mode <= 3;
  const bool pair = (s2_out & 0x200) != 0;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  // workgroup -> (patch mt, channel tile): class launches: tiles_n = channel tiles per class (subs) x 2 pairs (pair) or x 4 classes
  const int mt = tile / tiles_n, tn = tile - mt * tiles_n;
  const int th_n = H / SH, tw_n = W / TW, sp_img = th_n * tw_n;
  const long sp_total = (long)N * sp_img;
  const int nslabs = Cin / FKS;
  const int spc = nslabs >> 2;                                               // UPM 2: slabs per parity class of the input
  const int sm = (UPM == 2 || (UPM == 1 && s2_mode == 3)) ? 2 : 1;
  const int cpp = (UPM == 2) ? (Cin >> 2) : Cin;                             // channels per pixel of the tensor behind x

  // halo load units (fixed per workgroup): source offset in floats at channel 0 of the slab (-1: zero); the LDS offset of unit i is
  // hdst0 + i * PPI * LDA
  int hsrc[AU];
  unsigned hneg = 0u;
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = tid + i * FS_THREADS;
    const int pix = u / KQ, kq = u - pix * KQ;
    const int s = pix / SPIX, rem = pix - s * SPIX;
    const int hr = rem / PW, hc = rem - hr * PW;
    const long spg = (long)mt * NIMG + s;
    const int img = (int)(spg / sp_img), r2 = (int)(spg - (long)img * sp_img);
    const int h0 = (r2 / tw_n) * SH, w0 = (r2 - (r2 / tw_n) * tw_n) * TW;
    const int hi = h0 - 1 + hr, wi = w0 - 1 + hc;
    const bool inside = hi >= 0 && hi < H && wi >= 0 && wi < W, up2 = UPM == 1;
    const bool ok = (u < UNITS) && spg < sp_total && (inside || up2);
    const int hs = up2 ? min(max(hi, 0), H - 1) : hi, ws_ = up2 ? min(max(wi, 0), W - 1) : wi;
    hsrc[i] = ok ? (int)((((long)img * (H * sm) + hs * sm) * (W * sm) + ws_ * sm) * cpp + kq * 4) : -1;
    if (up2 && ((hi < 0) != (wi < 0))) hneg |= 1u << i;
  }
  const int hdst0 = (tid / KQ) * LDA + (tid % KQ) * 4;

  // wavefront -> pixel tiles: m = wm * MI + mi; sub-patch s = m / MPS, first row (m % MPS) * RPM; lane -> pixel (l31 / TW, l31 % TW)
  const int m0 = wm * MI;
  // Rows of a pixel tile: consecutive (first row (m % MPS) * RPM), or -- 16x16 sub-patches, ROWI -- INTERLEAVED: tile m holds rows m % MPS and
  // m % MPS + 8.  The fragment read ds_read_b32 serves 32 lanes per cycle from 32 banks at pixel stride LDA (odd): two pixels collide iff their halo
  // indices are congruent mod 32.  Rows r, r + 1 of an 18-pixel-wide halo put lanes 30 / 31 on the banks of lanes 0 / 1 (indices 32, 33); rows r, r + 8
  // are 144 = 4 x 32 + 16 apart: conflict-free (measured: 869 -> 867 us on conv2d_5's fused forward -- the LDS is not what bounds these kernels).
  constexpr bool ROWI = (GEO == 0);
  constexpr int RSTEP = ROWI ? MPS : 1, RMUL = ROWI ? 1 : RPM;                 // row step inside a tile, first-row multiplier of the tile index
  const int a_lane = (((m0 / MPS) * SPIX + (m0 % MPS) * RMUL * PW + (l31 / TW) * RSTEP * PW + (l31 % TW)) * LDA) + lh;
  constexpr int A_MI = RMUL * PW * LDA;                                      // second pixel tile of the wavefront (same sub-patch)

  const int subs = cls_mode ? creal / BN : 0;
  const int ncc = (cls_mode && pair) ? 2 : 1;
  for (int cc = 0; cc < ncc; ++cc) {
  // class and first bank column of this pass
  int cls = 0, n0;
  if (cls_mode) {
    const int sub = tn % subs, q = tn / subs;
    cls = pair ? (cc == 0 ? q : 3 - q) : ((q + mt) & 3);                     // pairs (0, 3) / (1, 2); single classes rotate with the patch (class_tile, common.h)
    n0 = cls * creal + sub * BN;
  } else {
    n0 = tn * BN;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ha[AU], rb[TPS][BUN];
  auto load_halo = [&](int slab) {
    int soff = slab * FKS;
    if (UPM == 2) {                                                          // slab -> (parity class, slab of that class)
      const int c = slab / spc;
      soff = ((c >> 1) * 2 * W + (c & 1)) * (Cin >> 2) + (slab - c * spc) * FKS;
    }
#pragma unroll
    for (int i = 0; i < AU; ++i) ha[i] = *reinterpret_cast<const float4*>(hsrc[i] >= 0 ? x + (size_t)(unsigned)hsrc[i] + soff : f32s_zero16);
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AU; ++i)
      if (tid + i * FS_THREADS < UNITS) {
        float4 v = ha[i];
        if (UPM == 1 && ((hneg >> i) & 1u)) v = make_float4(-v.x, -v.y, -v.z, -v.w);
        float* p = &Ah[buf][hdst0 + i * PPI * LDA];
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
      }
  };
  auto load_b = [&](int slab, int tap, int t) {
#pragma unroll
    for (int j = 0; j < BUN; ++j) {
      const int u = tid + j * FS_THREADS, kr = u / BQ, nq = u - kr * BQ;
      const bool ok = u < BU && (n0 + nq * 4) < Cout;
      rb[t][j] = *reinterpret_cast<const float4*>(ok ? w + ((size_t)tap * Cin + slab * FKS + kr) * Cout + n0 + nq * 4 : f32s_zero16);
    }
  };
  auto store_b = [&](int buf, int t) {
#pragma unroll
    for (int j = 0; j < BUN; ++j) {
      const int u = tid + j * FS_THREADS;
      if (u < BU) *reinterpret_cast<float4*>(&Bh[buf][t * FKS * BN + u * 4]) = rb[t][j];
    }
  };

  const unsigned tmask0 = (unsigned)(tap_masks >> (9 * cls)) & 0x1ffu;
  auto mask_of = [&](int slab) -> unsigned { return (UPM == 2) ? ((unsigned)(tap_masks >> (9 * (slab / spc))) & 0x1ffu) : tmask0; };
  // a barrier step carries up to TPS taps of one slab (the last step of a slab may carry fewer): taps of the step, lowest first; -1 = none
  auto pop_taps = [&](unsigned& m, int* taps) {
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
      taps[t] = m != 0u ? __builtin_ctz(m) : -1;
      m &= m - 1u;                                           // (0 stays 0)
    }
  };

  int cur[TPS], nxt[TPS];
  unsigned rem = mask_of(0);
  pop_taps(rem, cur);
  load_halo(0);
#pragma unroll
  for (int t = 0; t < TPS; ++t)
    if (cur[t] >= 0) load_b(0, cur[t], t);
  store_halo(0);
#pragma unroll
  for (int t = 0; t < TPS; ++t)
    if (cur[t] >= 0) store_b(0, t);
  __syncthreads();
  int bbuf = 0;
  for (int slab = 0; slab < nslabs; ++slab) {
    const int hb = slab & 1;
    const int nsteps = (__popc(mask_of(slab)) + TPS - 1) / TPS, mid = nsteps >> 1;
    const bool more = slab + 1 < nslabs;
#pragma unroll 1
    for (int i = 0; i < nsteps; ++i) {
      // the next step: the rest of this slab, else the first taps of the next slab
      int nslab = slab;
      bool has_next = true;
      if (rem == 0u) {
        nslab = slab + 1;
        has_next = more;
        if (more) rem = mask_of(slab + 1);
      }
      if (has_next) {
        pop_taps(rem, nxt);
#pragma unroll
        for (int t = 0; t < TPS; ++t)
          if (nxt[t] >= 0) load_b(nslab, nxt[t], t);
      }
      if (i == 0 && more) load_halo(slab + 1);
#pragma unroll
      for (int t = 0; t < TPS; ++t) {
        const int tap = cur[t];
        if (tap < 0) continue;
        const int r = tap / 3, sft = tap - 3 * r;
        const float* Ab = &Ah[hb][a_lane + (r * PW + sft) * LDA];
        const float* Bb = &Bh[bbuf][t * FKS * BN + lh * BN + wn * (32 * NI) + l31];
        // (requesting the fragments of a whole group of k-steps ahead of its MFMAs -- in bulk, or software-pipelined one group ahead with counted
        // lgkmcnt waits -- measured 0 ... 5 % SLOWER on every geometry: the LDS latency of one wavefront is already covered by the other three
        // of its SIMD, and the longer-lived fragments cost registers: profiles/r05_small_maps_sweep.txt)
#pragma unroll
        for (int ks = 0; ks < FKS / 2; ++ks) {
          float a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = Ab[mi * A_MI + 2 * ks];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b[ni] = Bb[2 * ks * BN + ni * 32];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
      }
      if (i == mid && more) store_halo(hb ^ 1);
      if (has_next) {
#pragma unroll
        for (int t = 0; t < TPS; ++t)
          if (nxt[t] >= 0) store_b(bbuf ^ 1, t);
#pragma unroll
        for (int t = 0; t < TPS; ++t) cur[t] = nxt[t];
      }
      __syncthreads();
      bbuf ^= 1;
    }
  }

  // epilogue: lane = channel, register e -> pixel (e & 3) + 8 (e >> 2) + 4 lh of the 32-pixel tile = (row ce / TW, column ce % TW + 4 lh)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (32 * NI) + ni * 32 + l31;
    const int nl = cls_mode ? n - cls * creal : n;                           // channel of the output tensor
    const bool n_ok = n < Cout;
    const float bv = (bias != nullptr && n_ok) ? bias[nl] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + mi, s = m / MPS;
      const long spg = (long)mt * NIMG + s;
      if (spg >= sp_total) continue;
      const int img = (int)(spg / sp_img), r2 = (int)(spg - (long)img * sp_img);
      const int hb0 = (r2 / tw_n) * SH + (m % MPS) * RMUL, wb0 = (r2 - (r2 / tw_n) * tw_n) * TW + 4 * lh;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ce = (e & 3) + 8 * (e >> 2);
        const int hh = hb0 + (ce / TW) * RSTEP, ww = wb0 + ce % TW;
        float* yp = cls_mode ? y + (((long)img * 2 * H + 2 * hh + (cls >> 1)) * 2 * W + 2 * ww + (cls & 1)) * creal + nl
                             : y + (((long)img * H + hh) * W + ww) * Cout + n;
        if (n_ok) *yp = ladder_act_fn(acc[mi][ni][e] + bv, act);
      }
    }
  }
  if (cc + 1 < ncc) __syncthreads();                       // (the next class stages into the buffers the slowest wavefront may still read)
  }  // class passes
}

struct F32sPlan { bool ok; int geo, ni, fks, tps, pair, tiles_m, tiles_n, grid; };

// Geometry / tile choice: the first variant (largest tile first) that gives the chip two rounds of workgroup slots (>= 512), else the one with
// the most workgroups.  Class launches (s2_out 1 ... 3) run class pairs.
F32sPlan plan_f32s(int N, int H, int W, int Cin, int Cout, int s2_out) {
  F32sPlan best{};
  best.ok = false;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return best;
  const bool cls_mode = s2_out >= 1 && s2_out <= 3, upm2 = s2_out == 4 || s2_out == 5;
  if ((Cout % 4) != 0 || (cls_mode && (Cout % 4) != 0)) return best;
  const int creal = cls_mode ? Cout / 4 : Cout;
  static const int fks_env = getenv("LADDER_F32S_FKS") ? atoi(getenv("LADDER_F32S_FKS")) : 0;
  static const int geo_env = getenv("LADDER_F32S_GEO") ? atoi(getenv("LADDER_F32S_GEO")) : -1;
  static const int ni_env = getenv("LADDER_F32S_NI") ? atoi(getenv("LADDER_F32S_NI")) : 0;
  const int geos[4][4] = {{16, 16, 1, 0}, {16, 8, 1, 1}, {8, 8, 4, 2}, {8, 8, 2, 3}};   // TW, SH, NIMG, id -- in order of preference
  long best_grid = -1;
  for (int gi = 0; gi < 4; ++gi) {
    const int TW = geos[gi][0], SH = geos[gi][1], NIMG = geos[gi][2], geo = geos[gi][3];
    if (geo_env >= 0 && geo != geo_env) continue;
    if ((W % TW) != 0 || (H % SH) != 0) continue;
    if (TW == 8 && (W % 16) == 0 && (H % 8) == 0 && geo_env < 0) continue;             // (16-wide maps take the 16-wide sub-patches)
    const long sp = (long)N * (H / SH) * (W / TW);
    const long tiles_m = (sp + NIMG - 1) / NIMG;
    for (int ni = 2; ni >= 1; --ni) {
      if (ni_env && ni != ni_env) continue;
      const int BN = 64 * ni;
      if (cls_mode && (creal % BN) != 0) continue;
      if (!cls_mode && creal < BN && ni == 2) continue;
      if ((geo == 1 || geo == 3) && ni == 2) continue;                                  // (not instantiated: the small pixel tiles exist to multiply workgroups)
      if (geo == 2 && ni == 2) continue;
      // work per barrier step: 32-channel slabs on the 128-pixel tiles (their double-buffered halos fit LDS for two workgroups per CU):
      // conv2d_4's fused forward 524 -> 473 us, the 8x8 plain conv 359 -> 323.  Two TAPS per step (TPS = 2) measured 0 ... 7 % slower on every
      // geometry (profiles/r05_small_maps_sweep.txt) and is not instantiated.
      int fks = (geo == 1 || geo == 3) ? 32 : 16;
      const int tps = 1;
      if ((Cin % fks) != 0 || (upm2 && (Cin % (4 * fks)) != 0)) fks = 16;
      if (fks_env == 16) fks = 16;
      if ((Cin % fks) != 0 || (upm2 && (Cin % (4 * fks)) != 0)) continue;
      const long tiles_n = cls_mode ? (long)(creal / BN) * 2 : (creal + BN - 1) / BN;
      const long grid = tiles_m * tiles_n;
      if (grid >= (1L << 30)) continue;
      const bool enough = grid >= 512;
      if (best_grid < 0 || (best_grid < 512 && grid > best_grid)) {
        best = F32sPlan{true, geo, ni, fks, tps, cls_mode ? 1 : 0, (int)tiles_m, (int)tiles_n, (int)grid};
        best_grid = grid;
      }
      if (enough && best_grid >= 512) return best;
    }
  }
  return best;
}


// ---- the last output row / column of the upsample-fused forward in ONE launch (round 5) -----------------------------------------------------
// ladder_conv3x3_up2_edges (convsplit.hip) recomputes row 2H-1 and column 2W-1 of y = act(conv3x3_same(resize2x(x), w) + bias): a 1x3 convolution
// of the 1-D upsampled last row u of x with the taps w[0][s] + w[1][s], and a 3x1 convolution of the upsampled last column with w[r][0] + w[r][1]
// (row 2H-1 of the upsampled map equals row 2H-2, and the line below is the zero padding).  Round 4 ran that as an operand kernel (u, v and the two
// summed banks materialised), two launches of the gather kernel (32 ... 128 tiles each: split-K, 25-35 us apiece) and a scatter: 78-101 us per call,
// eight calls per iteration.  Here both lines are ONE launch: a workgroup of 4 wavefronts computes 64 line pixels x 64 channels; per 16-channel
// slab the three shifted copies of its line pixels (lerp of the two low-resolution neighbours on the fly, zero outside the line) and the three
// summed filter slabs are staged in LDS (double-buffered registers -> LDS), 24 MFMAs per wavefront and slab.  Output: the line buffers e_row
// [N, 2W, Cout], e_col [N, 2H, Cout] (bias and activation applied) that up2_edge_scatter_kernel places (and projects).
constexpr int EL_THREADS = 256, EL_M = 64, EL_N = 64, EL_K = 16, EL_LDA = EL_K + 1;

__global__ __launch_bounds__(EL_THREADS) void up2_edge_lines_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         const float* __restrict__ bias, float* __restrict__ e_row,
                                                                         float* __restrict__ e_col, const int N, const int H, const int W,
                                                                         const int Cin, const int Cout, const int act, const int sm,
                                                                         const int tiles_row, const int tiles_n) {
  __shared__ float Al[2][3 * EL_M * EL_LDA];
  __shared__ __attribute__((aligned(16))) float Bl[2][3 * EL_K * EL_N];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5, wm = wid >> 1, wn = wid & 1;
  const int mt_all = blockIdx.x / tiles_n, tn = blockIdx.x - mt_all * tiles_n;
  const bool row = mt_all < tiles_row;
  const int mt = row ? mt_all : mt_all - tiles_row;
  const int L = row ? W : H;                                                   // low-resolution length of the line; 2L pixels per image
  const long m_total = (long)N * 2 * L;
  const int n0 = tn * EL_N;
  // A units: (tap t, pixel p, channel quad kq) -> 3 x 64 x 4 = 768 float4 units, 3 per thread: unit u = tid + i * 256 -> t = i (256 units per tap)
  const int ap = tid >> 2, akq = tid & 3;
  const long am = (long)mt * EL_M + ap;                                        // line pixel of this thread's units
  const int an = (int)(am / (2 * L)), aq = (int)(am - (long)an * (2 * L));
  // source: pixel `lo` / `hi` of the last row (row) or last column (column) of image an; stride between line pixels in floats
  const long lstride = row ? (long)sm * Cin : (long)sm * (W * sm) * Cin;
  const float* xline = row ? x + (((long)an * (H * sm) + (long)(H - 1) * sm) * (W * sm)) * Cin : x + (((long)an * (H * sm)) * (W * sm) + (long)(W - 1) * sm) * Cin;
  // B units: (tap t, k row, column quad) -> 3 x 16 x 16 = 768 units, 3 per thread: t = i
  const int bk = tid >> 4, bq = tid & 15;
  const bool b_ok = (n0 + bq * 4) < Cout;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float4 ra[3], rb[3];
  auto load_slab = [&](int slab) {
    const int c0 = slab * EL_K;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int q = aq + t - 1;                                                // position on the upsampled line
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (am < m_total && q >= 0 && q < 2 * L) {
        const int lo = q >> 1, hi = min(lo + 1, L - 1);
        const float4 xl = *reinterpret_cast<const float4*>(xline + lo * lstride + c0 + akq * 4);
        v = xl;
        if (q & 1) {                                                           // the arithmetic of the resize kernel (lerp, weight 1/2)
          const float4 xh = *reinterpret_cast<const float4*>(xline + hi * lstride + c0 + akq * 4);
          v = make_float4(xl.x + (xh.x - xl.x) * 0.5f, xl.y + (xh.y - xl.y) * 0.5f, xl.z + (xh.z - xl.z) * 0.5f, xl.w + (xh.w - xl.w) * 0.5f);
        }
      }
      ra[t] = v;
      // summed taps: row line (r = 0, s = t) + (r = 1, s = t); column line (r = t, s = 0) + (r = t, s = 1)
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_ok) {
        const int t0 = row ? t : 3 * t, t1 = row ? 3 + t : 3 * t + 1;
        const float4 b0 = *reinterpret_cast<const float4*>(w + ((size_t)t0 * Cin + c0 + bk) * Cout + n0 + bq * 4);
        const float4 b1 = *reinterpret_cast<const float4*>(w + ((size_t)t1 * Cin + c0 + bk) * Cout + n0 + bq * 4);
        b = make_float4(b0.x + b1.x, b0.y + b1.y, b0.z + b1.z, b0.w + b1.w);
      }
      rb[t] = b;
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float* pa = &Al[buf][(t * EL_M + ap) * EL_LDA + akq * 4];
      pa[0] = ra[t].x; pa[1] = ra[t].y; pa[2] = ra[t].z; pa[3] = ra[t].w;
      *reinterpret_cast<float4*>(&Bl[buf][(t * EL_K + bk) * EL_N + bq * 4]) = rb[t];
    }
  };
  const int nslabs = Cin / EL_K;
  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int slab = 0; slab < nslabs; ++slab) {
    const int buf = slab & 1;
    if (slab + 1 < nslabs) load_slab(slab + 1);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float* Ab = &Al[buf][(t * EL_M + wm * 32 + l31) * EL_LDA + lh];
      const float* Bb = &Bl[buf][(t * EL_K + lh) * EL_N + wn * 32 + l31];
#pragma unroll
      for (int ks = 0; ks < EL_K / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[2 * ks], Bb[2 * ks * EL_N], acc, 0, 0, 0);
    }
    if (slab + 1 < nslabs) store_slab(buf ^ 1);
    __syncthreads();
  }
  const int n = n0 + wn * 32 + l31;
  if (n < Cout) {
    const float bv = bias != nullptr ? bias[n] : 0.f;
    float* out = row ? e_row : e_col;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long m = (long)mt * EL_M + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (m < m_total) out[m * Cout + n] = ladder_act_fn(acc[e] + bv, act);
    }
  }
}

}  // namespace

// e_row [N, 2W, Cout], e_col [N, 2H, Cout] of ladder_conv3x3_up2_edges in one launch (strict fp32; any precision's edge path is fp32)
bool up2_edge_lines_f32_ok(int N, int H, int W, int Cin, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_EDGE_LINES") != nullptr;
  return !off && N > 0 && H > 0 && W > 0 && (Cin % EL_K) == 0 && (Cout % 4) == 0 && (long)N * 4 * H * W * Cin < (1L << 40);
}

int up2_edge_lines_f32(const float* x, const float* w, const float* bias, float* e_row, float* e_col, int N, int H, int W, int Cin, int Cout,
                       int act, int x_upsampled, hipStream_t stream) {
  if (!up2_edge_lines_f32_ok(N, H, W, Cin, Cout)) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(w)) return LADDER_E_ALIGN;
  const long tiles_row = ((long)N * 2 * W + EL_M - 1) / EL_M, tiles_col = ((long)N * 2 * H + EL_M - 1) / EL_M;
  const int tiles_n = (Cout + EL_N - 1) / EL_N;
  if ((tiles_row + tiles_col) * tiles_n >= (1L << 30)) return LADDER_E_SHAPE;
  hipLaunchKernelGGL(up2_edge_lines_f32_kernel, dim3((unsigned)((tiles_row + tiles_col) * tiles_n)), dim3(EL_THREADS), 0, stream, x, w, bias, e_row, e_col,
                     N, H, W, Cin, Cout, act, x_upsampled ? 2 : 1, (int)tiles_row, tiles_n);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

bool conv3x3_f32s_ok(int N, int H, int W, int Cin, int Cout, int s2_out) {
  static const bool off = getenv("LADDER_DISABLE_F32S") != nullptr;          // (test-only switch: the small-map layers back on the gather kernels)
  if (off) return false;
  const F32sPlan p = plan_f32s(N, H, W, Cin, Cout, s2_out);
  // worth a launch of its own: at least one workgroup per CU (below that the split-K gather kernel is the better tool)
  // (32-bit element offsets into the tensor behind x: [N, H, W, Cin], 4x that for the even sub-grid of an upsampled tensor)
  return p.ok && p.grid >= 256 && (long)N * H * W * Cin * (s2_out == 3 ? 4 : 1) < (1L << 31);
}

int conv3x3_f32s_launch(const float* x, const float* bank, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int act,
                        hipStream_t stream, unsigned long long tap_masks, int s2_out) {
  if (!conv3x3_f32s_ok(N, H, W, Cin, Cout, s2_out)) return LADDER_E_SHAPE;
  if (y == nullptr || x == nullptr || bank == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(bank) || !ladder_aligned16(y)) return LADDER_E_ALIGN;
  const F32sPlan p = plan_f32s(N, H, W, Cin, Cout, s2_out);
  const bool cls_mode = s2_out >= 1 && s2_out <= 3;
  const int creal = cls_mode ? Cout / 4 : Cout;
  const int s2x = s2_out | (p.pair ? 0x200 : 0);
  const dim3 grid(p.grid), block(FS_THREADS);
  const int upm = (s2_out == 2 || s2_out == 3) ? 1 : ((s2_out == 4 || s2_out == 5) ? 2 : 0);
#define LADDER_F32S_LAUNCH(UPM_, GEO_, NI_, FKS_, TPS_) \
  hipLaunchKernelGGL((conv3x3_halo_f32s_kernel<UPM_, GEO_, NI_, FKS_, TPS_>), grid, block, 0, stream, x, bank, bias, y, N, H, W, Cin, Cout, act, p.tiles_n, tap_masks, s2x, creal)
#define LADDER_F32S_TPS(UPM_, GEO_, NI_, FKS_) LADDER_F32S_LAUNCH(UPM_, GEO_, NI_, FKS_, 1)
#define LADDER_F32S_GEO(UPM_) \
  do { \
    if (p.geo == 0 && p.ni == 2) LADDER_F32S_TPS(UPM_, 0, 2, 16); \
    else if (p.geo == 0) LADDER_F32S_TPS(UPM_, 0, 1, 16); \
    else if (p.geo == 2) LADDER_F32S_TPS(UPM_, 2, 1, 16); \
    else if (p.geo == 1 && p.fks == 32) LADDER_F32S_TPS(UPM_, 1, 1, 32); \
    else if (p.geo == 1) LADDER_F32S_TPS(UPM_, 1, 1, 16); \
    else if (p.fks == 32) LADDER_F32S_TPS(UPM_, 3, 1, 32); \
    else LADDER_F32S_TPS(UPM_, 3, 1, 16); \
  } while (0)
  if (upm == 0) LADDER_F32S_GEO(0);
  else if (upm == 1) LADDER_F32S_GEO(1);
  else LADDER_F32S_GEO(2);
#undef LADDER_F32S_GEO
#undef LADDER_F32S_TPS
#undef LADDER_F32S_LAUNCH
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

extern "C" {

int ladder_conv3x3_f32_eligible(int N, int H, int W, int Cin, int Cout) { return conv3x3_f32_any_ok(N, H, W, Cin, Cout, 0) ? 1 : 0; }

int ladder_conv3x3_s2_bwd_data_f32_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_S2HALO") != nullptr;
  return (!off && H == 2 * Ho && W == 2 * Wo && Cin > 0 && conv3x3_f32_any_ok(N, Ho, Wo, Cout, 4 * Cin, 1)) ? 1 : 0;
}

int ladder_conv3x3_s2_fwd_f32_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout) {
  static const bool off = getenv("LADDER_DISABLE_S2HALO") != nullptr;
  return (!off && H == 2 * Ho && W == 2 * Wo && Cin > 0 && (Cin % 16) == 0 && conv3x3_f32_any_ok(N, Ho, Wo, 4 * Cin, Cout, 5)) ? 1 : 0;
}

int ladder_conv3x3_s2_fwd_f32(const float* x, const void* bank_s2f, const float* bias, float* y, int N, int H, int W, int Cin, int Ho, int Wo,
                              int Cout, int act, ladder_stream_t stream) {
  if (!ladder_conv3x3_s2_fwd_f32_eligible(N, H, W, Cin, Ho, Wo, Cout)) return LADDER_E_SHAPE;
  return conv3x3_f32_launch(x, (const float*)bank_s2f, bias, y, nullptr, nullptr, nullptr, 0, N, Ho, Wo, 4 * Cin, Cout, act, stream,
                            filter_bank_tap_masks(5), 5);
}

}  // extern "C"
