// Internal interface of convf32.hip: the strict-fp32 (v_mfma_f32_32x32x2_f32) forms of the fused 3x3 halo convolutions.  The exported entry
// points are the ladder_conv3x3_*_split* / ladder_filter_pack_split* functions of include/ladder_hip.h called with prec = LADDER_PREC_F32:
// convsplit.hip forwards those calls here.
#pragma once
#include "common.h"

// geometry the fp32 halo kernel takes (8x32-pixel x 128-channel tiles, >= 512 workgroups): the same predicate as the split kernels'
bool conv3x3_f32_halo_ok(int N, int H, int W, int Cin, int Cout);

// ... or one of the small-map tilings of convf32s.hip does (16- / 8-pixel-wide maps, class widths other than 128).  s2_out as below.
bool conv3x3_f32s_ok(int N, int H, int W, int Cin, int Cout, int s2_out);
bool conv3x3_f32_any_ok(int N, int H, int W, int Cin, int Cout, int s2_out);
int conv3x3_f32s_launch(const float* x, const float* bank, const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int act,
                        hipStream_t stream, unsigned long long tap_masks, int s2_out);

// the two edge lines of ladder_conv3x3_up2_edges (e_row [N, 2W, Cout], e_col [N, 2H, Cout], bias + activation applied) in one launch
bool up2_edge_lines_f32_ok(int N, int H, int W, int Cin, int Cout);
int up2_edge_lines_f32(const float* x, const float* w, const float* bias, float* e_row, float* e_col, int N, int H, int W, int Cin, int Cout,
                       int act, int x_upsampled, hipStream_t stream);

// fp32 bank [ntaps][Cin][Cout] of the LOGICAL filter of an orientation (filterbank.h): bytes, one bank, all banks of a job table
size_t filter_pack_f32_bytes(int ntaps, int Cin, int Cout);
int filter_pack_f32(const float* w, float* bank, int ntaps, int Cin, int Cout, int transpose_flip, hipStream_t stream);
int filter_pack_f32_multi(const ladder_pack_job_t* jobs_dev, int njobs, int total_blocks, hipStream_t stream);

// y = act(conv3x3_same(x, bank) + bias) [+ fused 1x1 projection]; tap_masks / s2_out as in conv3x3_split_launch (convsplit.hip):
//   s2_out 0 plain, 1 class-interleaved output of a stride-2 backward-data, 2 / 3 upsample-fused forward (x low-resolution / the even
//   sub-grid of an upsampled tensor), 4 backward-data of the upsample-fused pair (x = dy, classes = input-channel groups), 5 forward of a
//   3x3 / stride-2 conv (x = the layer input [N, 2H, 2W, Cin / 4], classes = input-channel groups; orientation 5 of filterbank.h)
int conv3x3_f32_launch(const float* x, const float* bank, const float* bias, float* y, const float* pw, const float* pb, float* pout,
                       int pco, int N, int H, int W, int Cin, int Cout, int act, hipStream_t stream, unsigned long long tap_masks,
                       int s2_out);

// csrc/densef32.hip: the persistent 128x128x32 strict-fp32 GEMM of the projected decoder pairs (C [M][N] = A [M][K] . B [K][N] + bias, activation, gate)
bool dense_f32_big_ok(long M, int K, int N);
int dense_f32_big_launch(const float* A, const float* B, const float* bias, float* C, const float* gate, int gate_act, long M, int K, int N, int act,
                         hipStream_t stream);
// the same product with the second operand K-contiguous, Bt [N][K] (v_mfma_f32_16x16x4_f32 kernel)
int dense_f32_nt_launch(const float* A, const float* Bt, const float* bias, float* C, const float* gate, int gate_act, long M, int K, int N, int act,
                        hipStream_t stream);
// ... and the filter-gradient form dW [K][N] = x^T dy over M rows: partial tiles [splits][K][N] (+ partial column sums of dy [splits][N]) for a
// fixed-order second stage
bool dense_wgrad_f32_ok(long M, int K, int N);
void dense_wgrad_f32_plan(long M, int K, int N, int* splits, int* m_per_split);
size_t dense_wgrad_f32_ws_bytes(long M, int K, int N);
int dense_wgrad_f32_launch(const float* x, const float* dy, float* part, float* bias_part, long M, int K, int N, int splits, int m_per_split,
                           hipStream_t stream);
