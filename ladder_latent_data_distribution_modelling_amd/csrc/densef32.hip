// Large strict-fp32 GEMMs of the "project, then upsample" decoder pairs (csrc/upproj.hip; round 5):  C [M][N] = A [M][K] . B [K][N], all row-major,
// M = low-resolution pixels (8 192 ... 524 288), K / N = Cin and 9 Cout (128 ... 2 304), on v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chains).
//
// Why not igemm_fwd_kernel (csrc/igemm.hip), which these calls ran on first: it is one workgroup per 128x128 tile with a 16-deep K chunk per barrier.
// At K = 128 (conv2d_7's projection) a workgroup lives for 8 chunks = 256 MFMAs per wave, and its address set-up, the first global round trip and a
// 64-store epilogue are not covered by anything: the MFMA pipe was busy 60 % of the time (profiles/r05_gemm_pmc.txt; 76 % at K = 1 152).  Here:
//   * PERSISTENT workgroups (2 per CU): each walks a contiguous range of output tiles, N tiles innermost (the A panel stays in L2 / L1 for the 9 ... 18
//     tiles that share it), and the operand pipeline runs straight across tile boundaries -- the first chunk of the next tile is in flight while the
//     current tile's epilogue stores;
//   * 32-deep chunks: one barrier per 64 MFMAs per wave, double-buffered LDS (68.9 KB), global -> registers issued before the chunk's MFMA block,
//     registers -> LDS after it;
//   * A fragments as ONE ds_read_b128 per 4 k-steps: lane (row, half) reads A[row][8 u + 4 half .. + 3] and uses element j in k-step (u, j); the B
//     fragment of that step is B[8 u + 4 half + j][col] -- the K order inside a chunk is a permutation both operands agree on (a sum of products:
//     the order of the fp32 FMA chain changes with it, nothing else).  A rows padded to 36 floats (16-byte aligned, conflict-free 128-bit reads).
// Epilogue: bias, activation, optional gate (dx *= act'(gate), the fused activation backward of ladder_dense_bwd_data), 128-byte row segments per half-wave.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GB_M = 128, GB_N = 128, GB_K = 32, GB_LDA = GB_K + 4, GB_THREADS = 256;

__device__ __forceinline__ float4 g_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// EPI: bias / activation / gate in the epilogue (the plain projections carry none: a straight store keeps the kernel small -- the general epilogue is
// 64 x (activation switch + gate) of straight-line code per wave)
template <bool EPI, bool NT>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                                 float* __restrict__ C, const float* __restrict__ gate, const int M, const int N,
                                                                 const int K, const int act, const int gate_act, const int tiles_n,
                                                                 const int tiles_total) {
  __shared__ __attribute__((aligned(16))) float As[2][GB_M * GB_LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][GB_K * GB_N];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int cpt = K / GB_K;                                                           // chunks per tile
  // Tile order.  Workgroup b runs on XCD b % 8 (round-robin dispatch), and an A panel (128 rows x K) serves the tiles_n column tiles of its row tile.
  // Row tile mt belongs to XCD mt % 8; the (row tile, column tile) pairs of an XCD are numbered r = local row tile * tiles_n + column tile and the
  // XCD's workgroups take r = i, i + W, i + 2W, ... (i = index inside the XCD, W = workgroups per XCD): at any moment the ~64 workgroups of an XCD
  // are inside 7-8 row tiles, so a panel is fetched into that XCD's L2 once and hit by the others.  (A contiguous range of tiles per workgroup --
  // the first version -- re-read every panel tiles_n times from beyond L2: 1.00 GB read per launch against 0.36 GB algorithmic,
  // profiles/r05_f32_pmc_traffic.json of that build.)
  const int tiles_m = tiles_total / tiles_n;
  const bool by_xcd = (gridDim.x & 7) == 0;
  const int xcd = by_xcd ? (int)(blockIdx.x & 7) : 0, wi = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, wpx = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xs = by_xcd ? 8 : 1;
  const int my_rows = (tiles_m - xcd + xs - 1) / xs;                                   // row tiles of this XCD
  const int nr = my_rows * tiles_n;                                                    // tiles of this XCD
  const int ntiles = wi < nr ? (nr - wi + wpx - 1) / wpx : 0;
  const int nchunks = ntiles * cpt;
  if (nchunks <= 0) return;

  // loader coordinates: A unit = (row tid / 8 + 32 i, k quad tid % 8), B unit = (k row tid / 32 + 8 i, n quad tid % 32)
  const int a_r = tid >> 3, a_q = tid & 7, b_r = tid >> 5, b_q = tid & 31;
  // TWO chunks of global loads in flight (register sets 0 / 1): the chunk written to LDS at the end of an iteration was requested one whole iteration
  // (64 MFMAs = ~2 us) earlier.  With one set the loads of chunk c + 1 were issued at the top of iteration c and awaited at its bottom -- under load an
  // HBM round trip is longer than the chunk's MFMA block, and both workgroups of a CU stalled on it together (MFMA pipe 0.69-0.82 busy).
  float4 ra0[4], rb0[4], ra1[4], rb1[4];
  int lt = wi, lc = 0;                                                                 // tile (index r inside the XCD) / chunk the NEXT load fetches
  auto load = [&](float4 (&ra)[4], float4 (&rb)[4]) {
    const int mrow = lt / tiles_n, nt = lt - mrow * tiles_n, mt = mrow * xs + xcd;
    const float* ap = A + ((long)mt * GB_M + a_r) * K + lc * GB_K + a_q * 4;
    const float* bp = B + ((long)lc * GB_K + b_r) * N + nt * GB_N + b_q * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = g_ld4(ap + (long)32 * i * K);
      rb[i] = g_ld4(bp + (long)8 * i * N);
    }
    if (++lc == cpt) { lc = 0; lt += wpx; }
  };
  auto store = [&](int buf, const float4 (&ra)[4], const float4 (&rb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&As[buf][(a_r + 32 * i) * GB_LDA + a_q * 4]) = ra[i];
      *reinterpret_cast<float4*>(&Bs[buf][(b_r + 8 * i) * GB_N + b_q * 4]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  load(ra0, rb0);
  store(0, ra0, rb0);
  if (nchunks > 1) load(ra1, rb1);
  __syncthreads();
  int ct = wi, cc = 0;                                                                 // tile / chunk being multiplied
  const int a_off = (wm * 64 + l31) * GB_LDA + 4 * lh, b_off = (4 * lh) * GB_N + wn * 64 + l31;
  // one iteration: request chunk g + 2 into the set chunk g came from, multiply chunk g, write chunk g + 1 (requested an iteration ago) to the other buffer
  auto iteration = [&](const int g, const int buf, float4 (&ra_ld)[4], float4 (&rb_ld)[4], const float4 (&ra_st)[4], const float4 (&rb_st)[4]) {
    if (g + 2 < nchunks) load(ra_ld, rb_ld);
    {
      // software pipeline over the chunk's four 8-deep groups: the fragments of group u + 1 are requested before the 16 MFMAs of group u are issued
      // (scheduling fences: left alone, the compiler sinks every LDS read to just in front of its first use and the MFMA pipe waits out each round trip)
      const float* Ab = &As[buf][a_off];
      const float* Bb = &Bs[buf][b_off];
      float4 af[2][2];
      float bf[2][4][2];
      auto frags = [&](int u, int s) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[s][mi] = *reinterpret_cast<const float4*>(Ab + mi * 32 * GB_LDA + 8 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[s][j][ni] = Bb[(8 * u + j) * GB_N + ni * 32];
      };
      __builtin_amdgcn_sched_barrier(0);
      frags(0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = u & 1;
        if (u + 1 < 4) frags(u + 1, s ^ 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const float a = j == 0 ? af[s][mi].x : (j == 1 ? af[s][mi].y : (j == 2 ? af[s][mi].z : af[s][mi].w));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s][j][ni], acc[mi][ni], 0, 0, 0);
          }
        }
      }
      // issue order of the block: group 0's six LDS reads, then ONE read of group u + 1 per two MFMAs of group u (a burst of reads in front of each
      // group left the waves queueing on the LDS issue port: SQ_WAIT_INST_LDS 71 M quad-cycles per launch against 4 M in the library GEMM)
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++cc == cpt) {                                                                 // the tile is complete: write it, start the next one from zero
      const int mrow = ct / tiles_n, nt = ct - mrow * tiles_n, mt = mrow * xs + xcd;
      const long voff = (long)4 * lh * N + l31;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = nt * GB_N + wn * 64 + ni * 32;
        const float bv = (EPI && bias != nullptr) ? bias[n + l31] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const long rbase = ((long)mt * GB_M + wm * 64 + mi * 32) * N + n;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const long o = rbase + (long)((e & 3) + 8 * (e >> 2)) * N + voff;
            float v = acc[mi][ni][e];
            if (EPI) {
              v = ladder_act_fn(v + bv, act);
              if (gate != nullptr) v *= ladder_act_grad_from_out(gate[o], gate_act);
            }
            if (NT) __builtin_nontemporal_store(v, &C[o]);      // (a multi-GB product written once: keep it from displacing the operand panels in L2)
            else C[o] = v;
            acc[mi][ni][e] = 0.f;
          }
        }
      }
      cc = 0;
      ct += wpx;
    }
    if (g + 1 < nchunks) store(buf ^ 1, ra_st, rb_st);
    __syncthreads();
  };
  for (int g = 0; g < nchunks; g += 2) {
    iteration(g, 0, ra0, rb0, ra1, rb1);
    if (g + 1 < nchunks) iteration(g + 1, 1, ra1, rb1, ra0, rb0);
  }
}

// ---- the same product on v_mfma_f32_16x16x4_f32, both operands K-contiguous ("NT"): C [M][N] = A [M][K] . Bt [N][K]^T ---------------------------------
// Why a second kernel: un-profiled, gemm_f32_kernel sits at ~0.98 of the MFMA rate the chip sustains AT THE CLOCK IT RUNS (1.92-1.98 GHz under this kernel,
// profiles/r05_gemm_pmc.txt) -- the limit is power, and the library GEMM runs the same shapes at 2.05-2.19 GHz.  Per flop the 16x16x4 instruction moves half
// the accumulator registers of 32x32x2 (4 of them per 2 048 flop against 16 per 4 096), and with Bt given K-contiguous (the projected pairs hold both
// wcat and wcatT) BOTH fragments are one 128-bit LDS read per four k-steps: 16 reads per 32-deep chunk instead of 24.  The roles are swapped in the
// instruction (a = Bt rows, b = A rows) so that a lane ends up with four CONSECUTIVE output columns: the epilogue is 16 float4 stores per wave tile.
template <bool EPI, bool NT>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_nt16_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bt, const float* __restrict__ bias,
                                                                      float* __restrict__ C, const float* __restrict__ gate, const int M, const int N,
                                                                      const int K, const int act, const int gate_act, const int tiles_n,
                                                                      const int tiles_total) {
  // Rows of exactly GB_K = 32 floats, the eight 16-byte quads of row r stored at position q ^ ((r >> 1) & 7).  ds_read_b128 is serviced in the lane groups
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) = eight rows of one k quad and the OTHER eight rows of the next: with the round-5 layout (rows padded
  // to 36 floats) 7 of every 8 positions were 2-way conflicted (SQ_LDS_BANK_CONFLICT = 1/3 of SQ_LDS_IDX_ACTIVE, profiles/r05_gemm_pmc.txt); with the XOR the
  // sixteen lanes of a group hit sixteen different 16-byte bank groups, and the tile takes 64 KB instead of 72.
  __shared__ __attribute__((aligned(16))) float As[2][GB_M * GB_K];
  __shared__ __attribute__((aligned(16))) float Bs[2][GB_N * GB_K];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int cpt = K / GB_K;
  const int tiles_m = tiles_total / tiles_n;
  const bool by_xcd = (gridDim.x & 7) == 0;                                            // (tile order: see gemm_f32_kernel)
  const int xcd = by_xcd ? (int)(blockIdx.x & 7) : 0, wi = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, wpx = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int xs = by_xcd ? 8 : 1;
  const int my_rows = (tiles_m - xcd + xs - 1) / xs;
  const int nr = my_rows * tiles_n;
  const int ntiles = wi < nr ? (nr - wi + wpx - 1) / wpx : 0;
  const int nchunks = ntiles * cpt;
  if (nchunks <= 0) return;

  const int a_r = tid >> 3, a_q = tid & 7;                                             // loader unit i of either operand: row a_r + 32 i, k quad a_q
  float4 ra0[4], rb0[4], ra1[4], rb1[4];
  int lt = wi, lc = 0;
  auto load = [&](float4 (&ra)[4], float4 (&rb)[4]) {
    const int mrow = lt / tiles_n, nt = lt - mrow * tiles_n, mt = mrow * xs + xcd;
    const float* ap = A + ((long)mt * GB_M + a_r) * K + lc * GB_K + a_q * 4;
    const float* bp = Bt + ((long)nt * GB_N + a_r) * K + lc * GB_K + a_q * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = g_ld4(ap + (long)32 * i * K);
      rb[i] = g_ld4(bp + (long)32 * i * K);
    }
    if (++lc == cpt) { lc = 0; lt += wpx; }
  };
  const int st_off = a_r * GB_K + 4 * (a_q ^ ((a_r >> 1) & 7));                        // (rows a_r + 32 i share the swizzle: (32 i) >> 1 is a multiple of 8)
  auto store = [&](int buf, const float4 (&ra)[4], const float4 (&rb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&As[buf][st_off + 32 * i * GB_K]) = ra[i];
      *reinterpret_cast<float4*>(&Bs[buf][st_off + 32 * i * GB_K]) = rb[i];
    }
  };
  f32x4 acc[4][4];                                                                     // [ni][mi]: rows of the instruction = 16 output columns, its columns = 16 output rows
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ni][mi][e] = 0.f;

  load(ra0, rb0);
  store(0, ra0, rb0);
  if (nchunks > 1) load(ra1, rb1);
  __syncthreads();
  int ct = wi, cc = 0;
  // fragment of 16-deep group u: quad 4 u + kq of row (tile row + r16) -> position (4 u + kq) ^ ((r16 >> 1) & 7)
  const int sw0 = 4 * (kq ^ ((r16 >> 1) & 7)), sw1 = sw0 ^ 16;
  const int a_off = (wm * 64 + r16) * GB_K, b_off = (wn * 64 + r16) * GB_K;
  auto iteration = [&](const int g, const int buf, float4 (&ra_ld)[4], float4 (&rb_ld)[4], const float4 (&ra_st)[4], const float4 (&rb_st)[4]) {
    if (g + 2 < nchunks) load(ra_ld, rb_ld);
    {
      const float* Ab = &As[buf][a_off];
      const float* Bb = &Bs[buf][b_off];
      float4 fa[2][4], fb[2][4];
      auto frags = [&](int u, int s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[s][t] = *reinterpret_cast<const float4*>(Ab + t * 16 * GB_K + (u ? sw1 : sw0));
          fb[s][t] = *reinterpret_cast<const float4*>(Bb + t * 16 * GB_K + (u ? sw1 : sw0));
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      frags(0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 0) frags(1, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float b = j == 0 ? fb[u][ni].x : (j == 1 ? fb[u][ni].y : (j == 2 ? fb[u][ni].z : fb[u][ni].w));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              const float a = j == 0 ? fa[u][mi].x : (j == 1 ? fa[u][mi].y : (j == 2 ? fa[u][mi].z : fa[u][mi].w));
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[ni][mi], 0, 0, 0);
            }
          }
      }
      // issue order: the first group's eight reads, then one read of the second group per eight MFMAs of the first, then the second group's MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 64, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++cc == cpt) {
      const int mrow = ct / tiles_n, nt = ct - mrow * tiles_n, mt = mrow * xs + xcd;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nt * GB_N + wn * 64 + ni * 16 + 4 * kq;
        const float4 bv = (EPI && bias != nullptr) ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const long o = ((long)mt * GB_M + wm * 64 + mi * 16 + r16) * N + n;
          float4 v = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
          if (EPI) {
            v = make_float4(ladder_act_fn(v.x + bv.x, act), ladder_act_fn(v.y + bv.y, act), ladder_act_fn(v.z + bv.z, act), ladder_act_fn(v.w + bv.w, act));
            if (gate != nullptr) {
              const float4 gv = *reinterpret_cast<const float4*>(gate + o);
              v.x *= ladder_act_grad_from_out(gv.x, gate_act); v.y *= ladder_act_grad_from_out(gv.y, gate_act);
              v.z *= ladder_act_grad_from_out(gv.z, gate_act); v.w *= ladder_act_grad_from_out(gv.w, gate_act);
            }
          }
          if (NT) {
            f32x4 v4;
            v4[0] = v.x; v4[1] = v.y; v4[2] = v.z; v4[3] = v.w;
            __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(C + o));
          } else {
            *reinterpret_cast<float4*>(C + o) = v;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ni][mi][e] = 0.f;
        }
      }
      cc = 0;
      ct += wpx;
    }
    if (g + 1 < nchunks) store(buf ^ 1, ra_st, rb_st);
    __syncthreads();
  };
  for (int g = 0; g < nchunks; g += 2) {
    iteration(g, 0, ra0, rb0, ra1, rb1);
    if (g + 1 < nchunks) iteration(g + 1, 1, ra1, rb1, ra0, rb0);
  }
}

// Filter gradient of the projected pairs: dWcat [Kc][N] = X^T D, X [M][Kc] (the low-resolution layer input), D [M][N] (the nine gradient planes); the
// reduction runs over the M pixels, so both operands sit in LDS pixel-major exactly as they lie in memory ([32 pixels][128 channels], 512-byte rows:
// conflict-free 32-bit fragment reads, lane = channel) and a (128 x 128 tile, pixel range) pair is one workgroup.  Partial tiles [split][Kc][N] are
// summed in a fixed order by the caller (launch_reduce_splits, csrc/igemm.hip): bit-reproducible.  The bias gradient (column sums of D) rides along
// in the workgroups of tile row 0, which stream every D row anyway.  Same software pipeline as gemm_f32_kernel.
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_tn_f32_kernel(const float* __restrict__ X, const float* __restrict__ D, float* __restrict__ part,
                                                                    float* __restrict__ bias_part, const int M, const int Kc, const int N,
                                                                    const int tiles_n, const int m_per_split, const int splits, const int by_xcd) {
  __shared__ __attribute__((aligned(16))) float At[2][GB_K * GB_M];
  __shared__ __attribute__((aligned(16))) float Bt[2][GB_K * GB_N];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8; the tiles of one pixel range (which all read the same rows of X and D) go to ONE XCD --
  // split s belongs to XCD s % 8 and the (split, tile) pairs of an XCD are handed out in order, so a range's tiles are dispatched together and
  // share the rows through that XCD's L2 (tile-major over all XCDs re-read them from beyond L2: 2.06 GB read per launch against 0.9 GB algorithmic)
  // (by_xcd = 0: tile-major over the whole chip -- tile counts that do not fill an XCD's 64 workgroup slots evenly)
  const int tiles = (Kc / GB_M) * tiles_n;
  const int xcd = (int)(blockIdx.x & 7), wi = (int)(blockIdx.x >> 3);
  const int split = by_xcd ? (wi / tiles) * 8 + xcd : (int)(blockIdx.x / tiles), tile = by_xcd ? wi % tiles : (int)(blockIdx.x % tiles);
  if (split >= splits) return;
  const int kt = tile / tiles_n, nt = tile - kt * tiles_n;
  const int p0 = split * m_per_split, p1 = min(M, p0 + m_per_split);
  const int nchunks = (p1 - p0 + GB_K - 1) / GB_K;
  const int l_r = tid >> 5, l_q = tid & 31;                                            // loader unit i: pixel row l_r + 8 i of the chunk, channel quad l_q
  const float* xp = X + (long)kt * GB_M + l_q * 4;
  const float* dp = D + (long)nt * GB_N + l_q * 4;
  const bool do_bias = bias_part != nullptr && kt == 0;
  float4 ra0[4], rb0[4], ra1[4], rb1[4], bsum[4];                                       // (two chunks of loads in flight, as in gemm_f32_kernel)
#pragma unroll
  for (int i = 0; i < 4; ++i) bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load = [&](int c, float4 (&ra)[4], float4 (&rb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = p0 + c * GB_K + l_r + 8 * i;
      const bool in = p < p1;
      ra[i] = in ? g_ld4(xp + (long)p * Kc) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = in ? g_ld4(dp + (long)p * N) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](int buf, const float4 (&ra)[4], const float4 (&rb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&At[buf][(l_r + 8 * i) * GB_M + l_q * 4]) = ra[i];
      *reinterpret_cast<float4*>(&Bt[buf][(l_r + 8 * i) * GB_N + l_q * 4]) = rb[i];
      if (do_bias) { bsum[i].x += rb[i].x; bsum[i].y += rb[i].y; bsum[i].z += rb[i].z; bsum[i].w += rb[i].w; }
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
  if (nchunks > 0) {
    load(0, ra0, rb0);
    store(0, ra0, rb0);
  }
  if (nchunks > 1) load(1, ra1, rb1);
  __syncthreads();
  // v_mfma_f32_16x16x4_f32, fragment = ONE 128-bit read per operand and k-step (4 pixels): lane (r16, kq) reads pixel 4 s + kq, channels 4 r16 .. 4 r16 + 3 of
  // its wave's 64 -- channel 4 r16 + t belongs to MFMA tile t (which 16 of the 64 rows a tile covers is free as long as the store below agrees)
  const int r16 = lane & 15, kq = lane >> 4;
  const int a_off = kq * GB_M + wm * 64 + 4 * r16, b_off = kq * GB_N + wn * 64 + 4 * r16;
  auto iteration = [&](const int c, const int buf, float4 (&ra_ld)[4], float4 (&rb_ld)[4], const float4 (&ra_st)[4], const float4 (&rb_st)[4]) {
    if (c + 2 < nchunks) load(c + 2, ra_ld, rb_ld);
    {
      const float* Ab = &At[buf][a_off];
      const float* Bb = &Bt[buf][b_off];
      float4 fa[2], fb[2];
      __builtin_amdgcn_sched_barrier(0);
      fa[0] = *reinterpret_cast<const float4*>(Ab);
      fb[0] = *reinterpret_cast<const float4*>(Bb);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int s = ks & 1;
        if (ks + 1 < 8) {
          fa[s ^ 1] = *reinterpret_cast<const float4*>(Ab + 4 * (ks + 1) * GB_M);
          fb[s ^ 1] = *reinterpret_cast<const float4*>(Bb + 4 * (ks + 1) * GB_N);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const float a = mi == 0 ? fa[s].x : (mi == 1 ? fa[s].y : (mi == 2 ? fa[s].z : fa[s].w));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float b = ni == 0 ? fb[s].x : (ni == 1 ? fb[s].y : (ni == 2 ? fb[s].z : fb[s].w));
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[mi][ni], 0, 0, 0);
          }
        }
      }
      // issue order: the first step's two reads; then the next step's two reads ahead of each step's 16 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nchunks) store(buf ^ 1, ra_st, rb_st);
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 2) {
    iteration(c, 0, ra0, rb0, ra1, rb1);
    if (c + 1 < nchunks) iteration(c + 1, 1, ra1, rb1, ra0, rb0);
  }
  // tile (mi, ni), register e of lane (c16 = r16, kq): tile row 4 kq + e = channel 4 (4 kq + e) + mi, tile column c16 = output column 4 c16 + ni
  float* o = part + (size_t)split * Kc * N + ((long)kt * GB_M + wm * 64) * N + nt * GB_N + wn * 64 + 4 * r16;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<float4*>(o + (long)(4 * (4 * kq + e) + mi) * N) = make_float4(acc[mi][0][e], acc[mi][1][e], acc[mi][2][e], acc[mi][3][e]);
  if (do_bias) {                                                                       // fixed-order sum of the loaders' column sums through LDS
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Bt[0][(l_r + 8 * i) * GB_N + l_q * 4]) = bsum[i];
    __syncthreads();
    if (tid < GB_N) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < GB_K; ++r) sacc += Bt[0][r * GB_N + tid];
      bias_part[(size_t)split * N + nt * GB_N + tid] = sacc;
    }
  }
}

}  // namespace

// C-ABI-internal entry (declared in convf32.h): returns false when the shape is not this kernel's
bool dense_f32_big_ok(long M, int K, int N) {
  static const bool off = getenv("LADDER_DISABLE_GEMM_F32") != nullptr;
  return !off && M >= 8192 && (M % GB_M) == 0 && (N % GB_N) == 0 && (K % GB_K) == 0 && K >= GB_K && M < (1L << 31) && (M / GB_M) * (N / GB_N) < (1L << 30);
}

int dense_f32_big_launch(const float* A, const float* B, const float* bias, float* C, const float* gate, int gate_act, long M, int K, int N, int act,
                         hipStream_t stream) {
  if (!dense_f32_big_ok(M, K, N) || A == nullptr || C == nullptr) return LADDER_E_SHAPE;
  if (B == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(A) || !ladder_aligned16(B) || !ladder_aligned16(C)) return LADDER_E_ALIGN;
  const int tiles_n = N / GB_N, tiles_total = (int)(M / GB_M) * tiles_n;
  static int slots = 0;
  if (slots == 0) {
    const char* e = getenv("LADDER_GEMM_F32_WGS");
    slots = e != nullptr ? atoi(e) : 512;                                              // 256 CUs x 2 resident workgroups
    if (slots < 1) slots = 512;
  }
  // whole tiles per workgroup, as even as the count allows: tiles_total / ceil(tiles_total / slots) workgroups, a multiple of the 8 XCDs
  const int per = (tiles_total + slots - 1) / slots;
  int grid = (tiles_total + per - 1) / per;
  if (grid >= 8) grid = (grid + 7) / 8 * 8;
  if (grid > slots && slots >= 8) grid = slots / 8 * 8;
  static const int nt_env = getenv("LADDER_GEMM_F32_NT") != nullptr ? atoi(getenv("LADDER_GEMM_F32_NT")) : -1;
  const bool nt = nt_env >= 0 ? nt_env != 0 : (size_t)M * N * sizeof(float) >= ((size_t)512 << 20);       // products well beyond the 256 MB of L2 + MALL stream out (conv2d_6: 714 -> 686 us)
  if (bias != nullptr || gate != nullptr || act != LADDER_ACT_NONE)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), dim3(grid), dim3(GB_THREADS), 0, stream, A, B, bias, C, gate, (int)M, N, K, act, gate_act, tiles_n, tiles_total);
  else if (nt)
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), dim3(grid), dim3(GB_THREADS), 0, stream, A, B, bias, C, gate, (int)M, N, K, act, gate_act, tiles_n, tiles_total);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3(grid), dim3(GB_THREADS), 0, stream, A, B, bias, C, gate, (int)M, N, K, act, gate_act, tiles_n, tiles_total);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// NT form: Bt [N][K]
int dense_f32_nt_launch(const float* A, const float* Bt, const float* bias, float* C, const float* gate, int gate_act, long M, int K, int N, int act,
                        hipStream_t stream) {
  if (!dense_f32_big_ok(M, K, N) || A == nullptr || C == nullptr) return LADDER_E_SHAPE;
  if (Bt == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(A) || !ladder_aligned16(Bt) || !ladder_aligned16(C) || (bias != nullptr && !ladder_aligned16(bias)) ||
      (gate != nullptr && !ladder_aligned16(gate)))
    return LADDER_E_ALIGN;
  const int tiles_n = N / GB_N, tiles_total = (int)(M / GB_M) * tiles_n;
  const int slots = 512;
  const int per = (tiles_total + slots - 1) / slots;
  int grid = (tiles_total + per - 1) / per;
  if (grid >= 8) grid = (grid + 7) / 8 * 8;
  if (grid > slots) grid = slots;
  static const int nt_env = getenv("LADDER_GEMM_F32_NT") != nullptr ? atoi(getenv("LADDER_GEMM_F32_NT")) : -1;
  const bool nt = nt_env >= 0 ? nt_env != 0 : (size_t)M * N * sizeof(float) >= ((size_t)512 << 20);
  if (bias != nullptr || gate != nullptr || act != LADDER_ACT_NONE)
    hipLaunchKernelGGL((gemm_nt16_f32_kernel<true, false>), dim3(grid), dim3(GB_THREADS), 0, stream, A, Bt, bias, C, gate, (int)M, N, K, act, gate_act, tiles_n, tiles_total);
  else if (nt)
    hipLaunchKernelGGL((gemm_nt16_f32_kernel<false, true>), dim3(grid), dim3(GB_THREADS), 0, stream, A, Bt, bias, C, gate, (int)M, N, K, act, gate_act, tiles_n, tiles_total);
  else
    hipLaunchKernelGGL((gemm_nt16_f32_kernel<false, false>), dim3(grid), dim3(GB_THREADS), 0, stream, A, Bt, bias, C, gate, (int)M, N, K, act, gate_act, tiles_n, tiles_total);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}

// ---- filter gradient dW [K][N] = x^T dy over M rows: plan (splits over the rows so that tiles x splits fills 2 workgroups per CU once)
bool dense_wgrad_f32_ok(long M, int K, int N) {
  static const bool off = getenv("LADDER_DISABLE_GEMM_F32") != nullptr;
  return !off && M >= 8192 && (K % GB_M) == 0 && (N % GB_N) == 0 && M < (1L << 31) && (long)(K / GB_M) * (N / GB_N) <= 512;
}

// XCD-aware split count: splits = 8 x (tile groups per XCD), one or two full rounds of an XCD's 64 workgroup slots; 0 = no even fit
static int wgrad_xcd_groups(long tiles) {
  static const bool off = getenv("LADDER_GEMM_TN_NO_XCD") != nullptr;
  if (off) return 0;
  if ((64 / tiles) * tiles >= 58) return (int)(64 / tiles);
  if ((128 / tiles) * tiles >= 116) return (int)(128 / tiles);
  return 0;
}

void dense_wgrad_f32_plan(long M, int K, int N, int* splits, int* m_per_split) {
  const long tiles = (long)(K / GB_M) * (N / GB_N);
  const int g = wgrad_xcd_groups(tiles);
  long s = g > 0 ? 8L * g : 512 / tiles;
  const long max_s = (M + 1023) / 1024;                                                // at least 1 024 rows per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  long mps = (M + s - 1) / s;
  mps = (mps + GB_K - 1) / GB_K * GB_K;
  *m_per_split = (int)mps;
  *splits = (int)((M + mps - 1) / mps);
}

size_t dense_wgrad_f32_ws_bytes(long M, int K, int N) {
  int splits, mps;
  dense_wgrad_f32_plan(M, K, N, &splits, &mps);
  return ((size_t)splits * K * N + (size_t)splits * N) * sizeof(float);
}

// writes the partial tiles [splits][K][N] to `part` and (bias_part != NULL) the partial column sums [splits][N]; the caller reduces them
int dense_wgrad_f32_launch(const float* x, const float* dy, float* part, float* bias_part, long M, int K, int N, int splits, int m_per_split,
                           hipStream_t stream) {
  if (!dense_wgrad_f32_ok(M, K, N) || x == nullptr || dy == nullptr || part == nullptr) return LADDER_E_SHAPE;
  if (!ladder_aligned16(x) || !ladder_aligned16(dy) || !ladder_aligned16(part)) return LADDER_E_ALIGN;
  const int tiles_n = N / GB_N;
  const int tiles = (K / GB_M) * tiles_n;
  const int by_xcd = wgrad_xcd_groups(tiles) > 0 ? 1 : 0;
  const unsigned grid = by_xcd ? 8u * ((splits + 7) / 8) * tiles : (unsigned)splits * tiles;
  hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(grid), dim3(GB_THREADS), 0, stream, x, dy, part, bias_part, (int)M, K, N, tiles_n, m_per_split, splits, by_xcd);
  LADDER_CHECK_LAUNCH();
  return LADDER_OK;
}
