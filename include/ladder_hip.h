/* libladder_hip.so -- C ABI of the MI355X (gfx950) LaDDer training-path kernels.
 *
 * The reference (lin-shuyu/ladder-latent-data-distribution-modelling) has NO FFI / plugin
 * registry: its hot path is a TF-1.15 graph evaluated by `sess.run(fetches, feed_dict)`
 * (codes/base.py:587-594, 603-605, 615-622, 639).  Each export below replaces one family of
 * TF/TFP op call sites of that graph; the call site it replaces is cited per function
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all exports):
 *   - extern "C", returns int: 0 = LADDER_OK, <0 = LADDER_E_*.  Never throws, never
 *     allocates, never synchronises the device, keeps no global mutable state.
 *     (Test-only exception: eight environment variables, read ONCE per process on first use, switch a specialised kernel family off so that
 *     tests can compare it with the generic path in situ -- LADDER_DISABLE_HALO, LADDER_DISABLE_HALO16, LADDER_DISABLE_S2HALO, LADDER_DISABLE_UP2, LADDER_DISABLE_GEMM16,
 *     LADDER_DISABLE_SMALLCIN, LADDER_DISABLE_COUT1, LADDER_DISABLE_SMALLCOUT.  They select between kernels with identical
 *     semantics, are never written by the library, and nothing in the product path sets them.)
 *   - every pointer is CALLER-OWNED DEVICE memory, fp32 unless stated, dense row-major,
 *     activations NHWC, conv filters HWIO, dense weights [in,out] (the reference's
 *     checkpoint layouts).  16-byte alignment of tensor base pointers is required.
 *   - `stream` is the HIP stream the work is enqueued on (pass the caller's current stream).
 *   - scratch memory is passed in explicitly (`ws`, `ws_bytes`); the matching
 *     `*_workspace_bytes` query returns the requirement.  Reductions use a fixed order
 *     (no float atomics) so results are bit-reproducible for a given shape.
 */
#ifndef LADDER_HIP_H
#define LADDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ladder_stream_t; /* == hipStream_t */

enum { LADDER_OK = 0, LADDER_E_SHAPE = -1, LADDER_E_ALIGN = -2, LADDER_E_WORKSPACE = -3, LADDER_E_LAUNCH = -4 };
enum { LADDER_ACT_NONE = 0, LADDER_ACT_LEAKY = 1 /* alpha 0.2 */, LADDER_ACT_RELU = 2, LADDER_ACT_TANH = 3 };

/* Build/ABI identification.  LADDER_ABI_VERSION is bumped whenever the layout of a buffer behind an EXISTING entry point changes:
 *   2 (round 5): the batch-norm statistics record of ladder_bn_fwd_stats / ladder_bn_stats_from_partials / the *_bnstats convolutions /
 *                ladder_bn_fwd_apply* is 2C DOUBLES (sum | sum of squares; + optionally 2C floats min | max), was 2C / 4C floats in version 1.
 * A caller built against another version must refuse to run (the Python binding does: _lib.load()). */
#define LADDER_ABI_VERSION 2
int ladder_abi_version(void);
/* BM*1000+BN of the implicit-GEMM instantiation a forward-type call (conv fwd / bwd_data / dense fwd / bwd_data) with
 * GEMM extents M x (.) x Cout and gathered channel count Cin dispatches to; negative = non-vectorised variant. */
int ladder_igemm_fwd_tile(long M, int Cin, int Cout);
/* Kernel a conv forward-type call dispatches to: 256128 = conv3x3_halo_kernel (3x3, stride 1, SAME, W%32==0, H%8==0: an
 * 8x32-pixel x 128-channel tile whose input halo is staged once in LDS for all 9 taps); 9003 = conv_smallcin_kernel (1x1
 * from 3 channels to a power-of-two 16..256 channels, i.e. the output conv's backward-data: direct, HBM-bound); else as
 * ladder_igemm_fwd_tile.
 * For bwd_data pass the dy geometry (N,Ho,Wo,Cout as input; H,W,Cin as output), stride 1, ups = stride, flipped pads. */
int ladder_conv2d_fwd_kernel_id(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int ups,
                                int pad_t, int pad_l);
/* Kernel the dense calls below dispatch to (strict fp32): 1 = the persistent 128x128x32 kernels of csrc/densef32.hip -- gemm_f32_kernel for
 * ladder_dense_fwd (M x K x N) and ladder_dense_bwd_data (pass its M, N, K: the contraction runs over N), gemm_tn_f32_kernel + the fixed-order
 * split sum for ladder_dense_bwd_weight; M >= 8192, M and the output width multiples of 128, the contraction a multiple of 32 -- 0 = the
 * implicit-GEMM kernels of csrc/igemm.hip. */
/* The two forward-type dense calls with the weight operand K-CONTIGUOUS (strict fp32, shapes for which ladder_dense_fwd_is_persistent is 1; LADDER_E_SHAPE
 * otherwise): y [M,N] = act(x [M,K] . wT^T + bias) with wT [N][K], and dx [M,K] = (dy [M,N] . w^T ...) -- i.e. dx = dy . (w [K][N])^T -- times act'(gate_y).
 * gemm_nt16_f32_kernel (csrc/densef32.hip): v_mfma_f32_16x16x4_f32, both fragments 128-bit LDS reads.  The projected decoder pairs hold both wcat and
 * wcatT, so they call these: forward with wcatT (orientation 7), backward-data with wcat (orientation 6). */
int ladder_dense_fwd_nt(const float* x, const float* wT, const float* bias, float* y, int M, int K, int N, int act, ladder_stream_t stream);
int ladder_dense_bwd_data_nt(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act, ladder_stream_t stream);
int ladder_dense_fwd_is_persistent(long M, int K, int N);
int ladder_dense_bwd_weight_is_persistent(long M, int K, int N);

/* ---------------------------------------------------------------- N1: tf.layers.conv2d
 * codes/models.py:51-71,115-148,203-229,273-315,398-460,514-585.
 * y[n,ho,wo,co] = act(b[co] + sum_{r,s,ci} x[n, ho*stride+r-pad_t, wo*stride+s-pad_l, ci] * w[r,s,ci,co])
 * (out-of-range taps read 0).  Explicit top/left padding expresses TF SAME (asymmetric) and VALID. */
int ladder_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                      int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                      int KH, int KW, int stride, int pad_t, int pad_l, int act,
                      void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Optional scratch of the forward-type calls (conv fwd / bwd_data, dense fwd / bwd_data): when the output tiling cannot
 * fill the 256 CUs (small M*Cout, long K) the contraction is split over K into `ws` and summed in a fixed order by a
 * second kernel (bias/activation applied there).  ws == NULL or too small => single-pass kernel.  M, K, Cout = GEMM extents. */
size_t ladder_igemm_fwd_workspace_bytes(long M, int K, int Cout);
/* Split-K factor of a DENSE forward-type call of this geometry (1 = single pass; the workspace query above also covers the parity classes
 * of a stride-2 backward-data, which split on their own). */
int ladder_igemm_fwd_splits(long M, int K, int Cout);
/* wT[KH-1-r][KW-1-s][co][ci] = w[r][s][ci][co]: the filter bank bwd_data consumes. */
int ladder_filter_flip_transpose(const float* w, float* wT, int KH, int KW, int Cin, int Cout, ladder_stream_t stream);
/* dx[n,hi,wi,ci] = sum dy[n,ho,wo,co] * w[r,s,ci,co] over {hi = ho*stride + r - pad_t, ...}; wT from above.
 * (N,H,W,Cin) describe dx, (Ho,Wo,Cout) describe dy.  Optional gate (gate_y != NULL, same shape as dx): the result is
 * multiplied by act'(gate_y) with gate_act in LADDER_ACT_*, i.e. the activation backward of the layer that PRODUCED this
 * conv's input is fused into the epilogue (gate_y = that layer's activation output = this conv's forward input). */
int ladder_conv2d_bwd_data(const float* dy, const float* wT, float* dx,
                           int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                           int KH, int KW, int stride, int pad_t, int pad_l,
                           const float* gate_y, int gate_act,
                           void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Kernel a filter-gradient call dispatches to: 9128 = wgrad3x3_halo_kernel (3x3, stride 1, SAME, Cin % 64 == 0, W % 32 == 0,
 * >= 4096 row patches), 0 = any other path (bench.py's per-kernel profile). */
int ladder_conv2d_bwd_filter_kernel_id(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                       int pad_l);
size_t ladder_conv2d_bwd_filter_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW);
/* dw[r,s,ci,co] = sum_{n,ho,wo} x[...] * dy[n,ho,wo,co];  db[co] = sum dy, accumulated inside the same kernel
 * (db may be NULL: e.g. a conv feeding batch-/instance-norm, whose bias gradient is identically zero). */
int ladder_conv2d_bwd_filter(const float* x, const float* dy, float* dw, float* db,
                             int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                             int KH, int KW, int stride, int pad_t, int pad_l,
                             void* ws, size_t ws_bytes, ladder_stream_t stream);

/* ---- fused backward of a 1x1 convolution to <= 4 channels over a wide map (the CelebA output conv, codes/models.py:580-586):
 * ONE pass over x [M, Cin] (the producing layer's OUTPUT) yields dx = (dy . W^T) * act'(x) (gate_act = that layer's activation,
 * 0 = none; dx may be NULL), dw [Cin, Cout] and db [Cout] (may be NULL).  Eligible: Cout <= 4, Cin/4 a power of two in 4..64,
 * M >= 65536 (ladder_conv1x1_smallcout_eligible).  ladder_conv2d_fwd uses the matching forward kernel for the same shapes. */
int ladder_conv1x1_smallcout_eligible(long M, int Cin, int Cout);
size_t ladder_conv1x1_smallcout_bwd_workspace_bytes(long M, int Cin, int Cout);
int ladder_conv1x1_smallcout_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long M, int Cin,
                                 int Cout, int gate_act, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* ... additionally produces the absolute-maximum record of dx (LADDER_ABSMAX_FLOATS floats, see ladder_absmax; dx must not be NULL):
 * per sample (mode 1) when rows_per_sample = H*W > 0 divides M (the workgroups of a sample then sweep its pixels), else one bound for the tensor. */
int ladder_conv1x1_smallcout_bwd_absmax(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, long M, int Cin,
                                        int Cout, int gate_act, void* ws, size_t ws_bytes, float* dx_absmax, long rows_per_sample,
                                        ladder_stream_t stream);

/* ---------------------------------------------------------------- N1s: the same convolutions on the 16-bit matrix cores
 * (operand splitting; csrc/convsplit.hip).  gfx950 has no TF32-class MFMA and its f32-input MFMA runs at the f32 vector rate;
 * v_mfma_f32_32x32x16_{f16,bf16} are 16x faster.  Each fp32 operand is split into 16-bit planes whose cross products are
 * accumulated in fp32 (`prec`):
 *   LADDER_PREC_F16X3   2 fp16 planes = 22 bits, 3 products, dropped term <= 2^-22 relative (fp32 class); operands are scaled by a
 *                       power of two derived from their absolute maximum -- per SAMPLE where the record carries per-sample bounds,
 *                       else per tensor (x_absmax: see the record layouts below; the filter's is taken by the pack call).
 *   LADDER_PREC_BF16X6  3 bf16 planes = 24 bits, 6 products, dropped terms <= 2^-23 relative (fp32 class); no scaling.
 *   LADDER_PREC_BF16X3  2 bf16 planes = 16 bits, 3 products, <= 3 * 2^-17 relative.
 *   LADDER_PREC_F32     (round 4) NO splitting: the same fused kernels on v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chains
 *                       (csrc/convf32.hip).  `packed` is then the fp32 bank [ntaps][Cin][Cout] of the orientation
 *                       (ladder_filter_pack_split with this prec; orientation 0 = the HWIO bank itself, which may be passed directly),
 *                       every *_absmax argument is ignored and may be NULL.  Accepted by ladder_filter_pack_split(_bytes / _multi),
 *                       ladder_conv3x3_split(_proj), ladder_conv3x3_up2_split(_proj), ladder_conv3x3_up2_bwd_data_split and
 *                       ladder_conv3x3_s2_bwd_data_split; the gather / filter-gradient *_split entry points are 16-bit only (their
 *                       strict-fp32 counterparts are ladder_conv2d_fwd / _bwd_data / _bwd_filter).
 * Replaces the same call sites as ladder_conv2d_fwd / ladder_conv2d_bwd_data for 3x3 / stride 1 / SAME layers with W % 32 == 0,
 * H % 8 == 0, Cin % 16 == 0, Cout % 4 == 0, Cout >= 64 and >= 512 workgroups (codes/models.py:538-578: the decoder's
 * 32x32 ... 128x128 maps). */
enum { LADDER_PREC_F32 = 0, LADDER_PREC_BF16X3 = 2, LADDER_PREC_BF16X6 = 3, LADDER_PREC_F16X3 = 4 };
/* Absolute-maximum RECORD of a tensor: LADDER_ABSMAX_FLOATS floats = 16 lines of 128 bytes, exact and order-independent (producers fold
 * block maxima into it with one atomic max per workgroup), in one of two layouts (csrc/common.h):
 *   mode 0 (float 1 == 0): ONE bound for the tensor = the maximum of float 0 of each line.  ladder_absmax; batch-norm producers.
 *   mode 1 (float 1 != 0): one bound PER SAMPLE of an [N, ...] tensor: sample n in float 2 + (n/16) % 30 of line n % 16 (480 distinct
 *                          samples; larger batches share slots).  ladder_absmax_samples; the split conv epilogue, instance-norm
 *                          forward / backward, the fused instance-norm + resize.  Round 3: the f16x3 kernels whose accumulations stay
 *                          inside one sample (3x3 halo conv forward / backward-data, gather conv forward / backward-data) scale every
 *                          sample by ITS OWN maximum (capped at 2^24 above the tensor-wide scale), so the format's fp32-like relative
 *                          precision holds per sample; reductions over all samples (filter gradients) use the tensor-wide bound or
 *                          re-scale their accumulators at sample boundaries.
 * Every entry is an UPPER bound (a looser one only raises the representation floor 2^-38 * bound).  ladder_absmax* fill `out` from
 * scratch; kernels with a `*_absmax` OUTPUT argument zero it and fill it for the tensor they write (NULL = skip). */
#define LADDER_ABSMAX_FLOATS 512
int ladder_absmax(const float* x, size_t n, float* out, ladder_stream_t stream);
/* per-sample record of x [n_samples, per_sample] (per_sample % 4 == 0). */
int ladder_absmax_samples(const float* x, int n_samples, size_t per_sample, float* out, ladder_stream_t stream);
/* ladder_filter_pack_split: once per weight update, HWIO fp32 bank of ntaps = KH*KW taps -> split planes in the kernels' LDS layout
 * (Cin % 16 == 0; output channels zero-padded to a multiple of 128).
 *   transpose_flip = 0: `w` = [KH][KW][Cin][Cout] (forward).  transpose_flip = 1: `w` = the layer's bank [KH][KW][Cout][Cin] read as
 *   the flipped, transposed filter of its backward-data pass (Cin = dy channels, Cout = dx channels).
 *   transpose_flip = 2: the four output-parity classes of a stride-2 backward-data as four 128-channel output tiles (ladder_conv3x3_s2_bwd_data_split).
 *   transpose_flip = 3: the effective taps of the upsample-fused forward, four output-parity classes (Cout = 4 x 128; ladder_conv3x3_up2_split).
 *   transpose_flip = 4: the effective taps of its backward-data, the four pixel-parity classes of dy as input groups (Cin = 4 x C;
 *   ladder_conv3x3_up2_bwd_data_split).  For 3 and 4 the bank's absmax record holds 4 x max|w| (a bound of the effective taps).
 *   transpose_flip = 5 (strict fp32 only): forward of a 3x3 / stride-2 conv, the four pixel-parity classes of x as input groups (Cin = 4 x C;
 *   ladder_conv3x3_s2_fwd_f32).
 *   transpose_flip = 6 / 7 (strict fp32 only, ntaps = 1): the nine taps side by side as one [Cin][9 C] matrix (Cout = 9 C) / its transpose
 *   [9 C][Cin_layer] (Cin = 9 C) -- the operands of the "project, then upsample" form below (ladder_up2proj_*). */
size_t ladder_filter_pack_split_bytes(int ntaps, int Cin, int Cout, int prec);
int ladder_filter_pack_split(const float* w, void* packed, int ntaps, int Cin, int Cout, int transpose_flip, int prec,
                             ladder_stream_t stream);
/* All filter banks of a model in TWO launches (after an optimiser step: forward orientation and the flipped / transposed one of every
 * convolution; the per-bank call above is memset + absmax + pack = three launches each).  `jobs_dev` = job table in DEVICE memory, sorted
 * by block_begin (block_begin = running sum of ladder_filter_pack_job_blocks over the preceding jobs, total_blocks = the sum over all);
 * every job's `packed` buffer has ladder_filter_pack_split_bytes bytes.  Scratch: ladder_filter_pack_split_multi_scratch_bytes(njobs). */
typedef struct ladder_pack_job_t {
  const float* w; /* HWIO fp32 bank */
  void* packed;   /* destination image (+ absmax record behind the payload) */
  int ntaps, Cin, Cout, transpose_flip;
  int block_begin, reserved;
} ladder_pack_job_t;
int ladder_filter_pack_job_blocks(int ntaps, int Cin, int Cout);
size_t ladder_filter_pack_split_multi_scratch_bytes(int njobs);
int ladder_filter_pack_split_multi(const ladder_pack_job_t* jobs_dev, int njobs, int total_blocks, int prec, void* scratch,
                                   size_t scratch_bytes, ladder_stream_t stream);
int ladder_conv3x3_split_eligible(int N, int H, int W, int Cin, int Cout);
/* Strict fp32 (prec = LADDER_PREC_F32) additionally takes the 16- and 8-pixel-wide maps (round 5, csrc/convf32s.hip: 16x16 / 8x16 / 8x8-pixel
 * sub-patches, 64- or 128-channel tiles, >= 256 workgroups; H % 8 == 0, W % 8 == 0, Cin % 16 == 0, Cout % 4 == 0) -- decoder conv2d_3
 * and the backward-data of the encoder's deep layers (codes/models.py:420-460, 539-543).  1 when ladder_conv3x3_split(prec F32) takes the call. */
int ladder_conv3x3_f32_eligible(int N, int H, int W, int Cin, int Cout);
/* y = act(conv3x3_same(x, F) + bias) with F as packed above (bias may be NULL; x_absmax is read only for LADDER_PREC_F16X3 and may
 * hold any upper bound of max |x|: a looser bound only raises the absolute representation floor 2^-38 * bound).  y_absmax (may be
 * NULL): record of the output, produced in the epilogue. */
int ladder_conv3x3_split(const float* x, const float* x_absmax, const void* packed, const float* bias, float* y, float* y_absmax, int N,
                         int H, int W, int Cin, int Cout, int act, int prec, ladder_stream_t stream);

/* The same convolution with a FUSED 1x1 projection of its activated output to proj_cout <= 4 channels (Cout <= 128 so that a workgroup
 * holds all channels of its pixels): proj_out[n,h,w,o] = proj_b[o] + sum_c y[n,h,w,c] * proj_w[c][o] -- the CelebA decoder's last 3x3
 * conv + its 1x1 output conv (codes/models.py:573-586) in one launch.  y may be NULL: a forward-only evaluation (RUN#2, val_step)
 * then never writes nor re-reads the 128-channel full-resolution activation. */
int ladder_conv3x3_split_proj(const float* x, const float* x_absmax, const void* packed, const float* bias, float* y, const float* proj_w,
                              const float* proj_b, float* proj_out, int proj_cout, int N, int H, int W, int Cin, int Cout, int act,
                              int prec, ladder_stream_t stream);

/* Backward-data of a 3x3 / stride-2 / SAME convolution over even maps (codes/models.py:398-460: the encoder layers; pad_t = pad_l = 0,
 * Cin = 128, dy map eligible for the halo kernel: Wo % 32 == 0, Ho % 8 == 0) as ONE launch of the 3x3 halo kernel over dy: the four
 * output-parity classes are the four 128-channel output tiles, each issues only its 4 / 2 / 2 / 1 taps and writes the interleaved pixels of
 * dx [N, H, W, 128].  packed_s2 = ladder_filter_pack_split(w, ., 9, Cout, 4 * Cin, transpose_flip = 2, prec) from the layer's HWIO bank
 * [3][3][Cin][Cout].  dy_absmax / dx_absmax as for ladder_conv3x3_split (per-sample records). */
int ladder_conv3x3_s2_bwd_data_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                              int pad_l);
int ladder_conv3x3_s2_bwd_data_split(const float* dy, const float* dy_absmax, const void* packed_s2, float* dx, float* dx_absmax, int N, int H,
                                     int W, int Cin, int Ho, int Wo, int Cout, int prec, ladder_stream_t stream);
/* Strict fp32 (round 5): the call above with prec = LADDER_PREC_F32 takes ANY class width Cin (% 64 == 0 on the small-map tilings of
 * csrc/convf32s.hip, 128 on the 8x32-pixel tiling) and 16- / 8-pixel-wide dy maps -- encoder conv2d_2 ... conv2d_3 (codes/models.py:420-439). */
int ladder_conv3x3_s2_bwd_data_f32_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout);
/* FORWARD of a 3x3 / stride-2 / SAME convolution over an even map (TF padding: pad_t = pad_l = 0; codes/models.py:409-439, encoder conv2d_1 ...
 * conv2d_3) as ONE launch of the fp32 halo kernels: a stride-1 correlation over the four pixel-parity classes of x [N, H, W, Cin] taken as four
 * groups of input slabs, each group walking only its 4 / 2 / 2 / 1 taps -- x is staged once per slab for all taps instead of gathered per tap.
 * bank_s2f = ladder_filter_pack_split(w, ., 9, 4 * Cin, Cout, transpose_flip = 5, LADDER_PREC_F32) from the layer's HWIO bank [3][3][Cin][Cout].
 * y [N, H/2, W/2, Cout] = act(conv + bias).  Strict fp32 only. */
int ladder_conv3x3_s2_fwd_f32_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout);
int ladder_conv3x3_s2_fwd_f32(const float* x, const void* bank_s2f, const float* bias, float* y, int N, int H, int W, int Cin, int Ho, int Wo,
                              int Cout, int act, ladder_stream_t stream);

/* "Project, then upsample" (round 5, strict fp32; csrc/upproj.hip): tf.image.resize_images(x, [2H, 2W]) (TF1 legacy bilinear) followed by
 * tf.layers.conv2d(3x3, SAME) -- decoder conv2d_4 ... conv2d_7, codes/models.py:544-578 -- with 9 / 36 of the direct form's products.  The resize is
 * linear and per channel, so it commutes with the channel contraction: conv3x3(up(x))[p, q] = b + sum_{r,s} up(Z_rs)[p + r - 1, q + s - 1] with
 * Z_rs = x . w[r][s], nine 1x1 convolutions at LOW resolution (up(.) = 0 outside the map: the convolution's zero padding).  Exact on every pixel:
 * no edge / border helper launches.  A layer's three passes, M = N H W low-resolution pixels:
 *   forward        ladder_dense_fwd(x, wcat, NULL, z, M, Cin, 9 Cout)  then  ladder_up2proj_fwd_combine(z, bias, y, ...)       y [N, 2H, 2W, Cout]
 *   backward-data  ladder_up2proj_bwd_combine(dy, d, ...)              then  ladder_dense_fwd(d, wcatT, NULL, dx, M, 9 Cout, Cin)
 *   filter grad.   ladder_dense_bwd_weight(x, d, dwcat, db9, M, Cin, 9 Cout)  then  ladder_up2proj_wgrad_unpack(dwcat, db9, dw, db, Cin, Cout)
 * wcat [Cin][9 Cout] (wcat[ci][t Cout + co] = w[t][ci][co], t = 3 r + s) = ladder_filter_pack_split(w, ., 1, Cin, 9 Cout, transpose_flip = 6, LADDER_PREC_F32)
 * and wcatT [9 Cout][Cin], its transpose, = ladder_filter_pack_split(w, ., 1, 9 Cout, Cin, transpose_flip = 7, LADDER_PREC_F32) -- once per weight update,
 * batched with every other bank by ladder_filter_pack_split_multi.  z / d [M][9 Cout]: plane t of pixel m at [m][t Cout ...].
 * fwd_combine: y = act(bias + combination) and / or, for Cout == 128, proj_out [N, 2H, 2W, proj_cout <= 4] = y . proj_w [128][proj_cout] + proj_b
 * (the decoder's 1x1 conv2d_8, codes/models.py:580-586, fused; y may then be NULL).  bwd_combine: d = (shift o up)^T dy, dy [N, 2H, 2W, C] already
 * multiplied by the activation's derivative.  wgrad_unpack: dw in the layer's HWIO layout, db (may be NULL) = the centre plane's column sums. */
int ladder_up2proj_eligible(int N, int H, int W, int Cin, int Cout);
int ladder_up2proj_fwd_combine(const float* z, const float* bias, float* y, const float* proj_w, const float* proj_b, float* proj_out, int proj_cout,
                               int N, int H, int W, int C, int act, ladder_stream_t stream);
int ladder_up2proj_bwd_combine(const float* dy, float* d, int N, int H, int W, int C, ladder_stream_t stream);
int ladder_up2proj_wgrad_unpack(const float* dwcat, const float* db9, float* dw, float* db, int Cin, int Cout, ladder_stream_t stream);
/* The two combinations for a resize factor of 2 or 4 (decoder conv2d_3 sits behind the 2x2 -> 8x8 resize, codes/models.py:536-542: up(Z)[F i + k] =
 * (1 - k/F) Z[i] + (k/F) Z[min(i+1, L-1)] per axis): y [N, F H, F W, C], dy likewise; z / d [N H W][9 C] as above.  The GEMM-shaped calls do not change. */
int ladder_upfproj_eligible(int factor, int N, int H, int W, int Cin, int Cout);
int ladder_upfproj_fwd_combine(const float* z, const float* bias, float* y, int factor, int N, int H, int W, int C, int act, ladder_stream_t stream);
int ladder_upfproj_bwd_combine(const float* dy, float* d, int factor, int N, int H, int W, int C, ladder_stream_t stream);
/* Round 6: the backward combination of the LAST factor-2 pair straight from the gradient of the 1x1 convolution behind it (reference codes/models.py:572-586:
 * conv2d_7 -> leaky ReLU -> conv2d_8): d = (shift o up)^T dy with dy = act'(y) * (dyp . pw^T) formed where it is consumed -- y [N, 2H, 2W, C] the pair's
 * activated output, dyp [N, 2H, 2W, pco] the gradient of the 1x1 convolution's (pre-activation) output, pw [C][pco] its filter -- and the 1x1 convolution's
 * own filter / bias gradient dpw [C][pco] / dpb [pco] (dpb may be NULL; both overwritten) from the same read of y.  Replaces ladder_conv1x1_smallcout_bwd
 * followed by ladder_up2proj_bwd_combine: dy (1.07 GB at batch 128) is neither written nor read.  Eligible: C / 4 a power of two in [5 pco, 64], pco <= 4, H >= 4, act in {none, leaky, relu};
 * `ws` >= ladder_up2proj_bwd_combine_proj_workspace_bytes (per-workgroup partial sums, added in a fixed order). */
/* ladder_up2proj_bwd_combine as a walk down the rows (a thread keeps the column-folded rows of its 5-row window in registers: 10 instead of 25 loads per
 * low-resolution pixel); rows_per_thread 0 = chosen by the library.  H >= 4. */
int ladder_up2proj_bwd_combine_walk(const float* dy, float* d, int N, int H, int W, int C, int rows_per_thread, ladder_stream_t stream);
int ladder_up2proj_bwd_combine_proj_eligible(int N, int H, int W, int C, int pco);
size_t ladder_up2proj_bwd_combine_proj_workspace_bytes(int N, int H, int W, int C, int pco);
int ladder_up2proj_bwd_combine_proj(const float* y, const float* dyp, const float* pw, float* d, float* dpw, float* dpb, int pco, int N, int H, int W, int C,
                                    int act, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Round 6: the FORWARD of a factor-2 pair in ONE launch (reference codes/models.py:544-586: tf.image.resize_images -> tf.layers.conv2d, the last pair
 * followed by the 1x1 conv2d_8) -- y = act(bias + sum_rs shift_rs(up(x . w_rs))) straight from x [N, H, W, Cin] and wcatT [9 Cout][Cin] (transpose_flip 7):
 * a workgroup owns 64 / W images x 16 output channels, streams down the rows of the low-resolution map and keeps the nine planes of the last three rows
 * in an LDS ring, so Z [N H W][9 Cout] never exists in HBM (2 x 2.4 GB per forward of conv2d_7 at batch 128).  Same products and sums as the two-call
 * form above (9 of the direct form's 36 per 2x2 output block), exact on every pixel.  Eligible: W in {8, 16, 32, 64}, N a multiple of 64 / W, H >= 2,
 * Cin a multiple of 32 (>= 64), Cout a multiple of 16.  proj_out != NULL: also proj_out [N, 2H, 2W, proj_cout] = y . proj_w + proj_b (proj_w
 * [Cout][proj_cout], proj_cout <= 4) through per-slab partial sums in `ws` (ladder_up2proj_fused_workspace_bytes), added in a fixed order; y may then be
 * NULL (forward-only runs never write the activation). */
int ladder_up2proj_fused_eligible(int N, int H, int W, int Cin, int Cout);
/* 1 where the fused form also measured FASTER than the two-call form as an ISOLATED launch (Cin <= 128: the weight slab stays in LDS; 8-pixel-wide maps; the shapes of
 * ladder_up2proj_fused_wide_tile):
 * the engine's `fused_projected_forward: 1`.  Its default is 2 = every eligible pair (fastest over the whole iteration); 0 = never. */
int ladder_up2proj_fused_preferred(int N, int H, int W, int Cin, int Cout);
/* 1 when the call (without projection) runs the 128-pixel row step variant: an MFMA wave owns two 16-pixel tiles, so a weight fragment feeds 8 MFMAs -- used
 * where the weight slab cannot stay in LDS (Cin > 128), W is 16 or 32, N a multiple of 128 / W and the grid (N W / 128) x (Cout / 16) >= 256 workgroups. */
int ladder_up2proj_fused_wide_tile(int N, int H, int W, int Cin, int Cout);
size_t ladder_up2proj_fused_workspace_bytes(int N, int H, int W, int Cout, int proj_cout);
int ladder_up2proj_fused_fwd(const float* x, const float* wcatT, const float* bias, float* y, const float* proj_w, const float* proj_b, float* proj_out,
                             int proj_cout, int N, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes, ladder_stream_t stream);

/* (strict fp32, round 5: Cout may be any multiple of 64 and the low-resolution map 16 or 8 pixels wide -- decoder conv2d_5 / conv2d_4,
 * codes/models.py:544-560; the fused projection form stays at Cout = 128)
 * A 3x3 / SAME convolution of the factor-2 legacy-bilinear upsample of x [N, H, W, Cin] -> y [N, 2H, 2W, 128], without the upsampled tensor
 * (replaces tf.image.resize_images(x, [2H, 2W]) + tf.layers.conv2d of the decoder, codes/models.py:554-578, in one launch; see csrc/convsplit.hip).
 * packed_up2 = ladder_filter_pack_split(w, ., 9, Cin, 4 * 128, transpose_flip = 3, prec) from the layer's HWIO bank; bias [128]; the absmax
 * record of x as for ladder_conv3x3_split.  The last output row and column are NOT final after this call: ladder_conv3x3_up2_edges. */
int ladder_conv3x3_up2_split_eligible(int N, int H, int W, int Cin, int Cout, int prec);
int ladder_conv3x3_up2_split(const float* x, const float* x_absmax, const void* packed_up2, const float* bias, float* y, float* y_absmax,
                             int N, int H, int W, int Cin, int Cout, int act, int prec, int x_upsampled, ladder_stream_t stream);
/* x_upsampled != 0 (here and in ladder_conv3x3_up2_edges): `x` points at an already upsampled tensor [N, 2H, 2W, Cin] whose even rows / columns
 * are the low-resolution map (up[2i][2j] = x[i][j]) -- a training forward keeps that tensor for the backward pass of the layer. */
int ladder_conv3x3_up2_split_proj(const float* x, const float* x_absmax, const void* packed_up2, const float* bias, float* y, const float* proj_w,
                                  const float* proj_b, float* proj_out, int proj_cout, int N, int H, int W, int Cin, int Cout, int act, int prec,
                                  int x_upsampled, ladder_stream_t stream);
/* Backward-data of the pair resize x2 -> 3x3 conv in one launch: dy [N, 2H, 2W, C] -> dx [N, H, W, Cout] (the gradient with respect to the
 * LOW-resolution input; the [N, 2H, 2W, Cout] intermediate of conv backward-data + resize transpose is never written).  packed_up2t =
 * ladder_filter_pack_split(w, ., 9, 4 * C, Cout, transpose_flip = 4, prec) from the layer's HWIO bank [3][3][Cout][C].  Exact everywhere but on
 * the four border lines of dx (rows 0, H-1, columns 0, W-1), which the caller recomputes from strips of dy (engine.Conv2D.backward_up2). */
int ladder_conv3x3_up2_bwd_data_split_eligible(int N, int H, int W, int C, int Cout, int prec);
int ladder_conv3x3_up2_bwd_data_split(const float* dy, const float* dy_absmax, const void* packed_up2t, float* dx, float* dx_absmax, int N, int H,
                                      int W, int C, int Cout, int prec, ladder_stream_t stream);
/* One border line of dx (axis 1: row 0 / H-1, axis 2: column 0 / W-1; first != 0: the 0 side) from d_up = the plain backward-data on the adjoining
 * strip of dy (rows strip [N, n_up, 2W, C] / columns strip [N, 2H, n_up, C], n_up = 2 on the 0 side, 3 on the other): the resize transpose across
 * the strip and along the line in one pass; dx_absmax (the record of the main launch) is raised where needed. */
int ladder_conv3x3_up2_bwd_border(const float* d_up, float* dx, float* dx_absmax, int N, int H, int W, int C, int axis, int first,
                                  ladder_stream_t stream);
/* Strict fp32 (round 4): the four border lines of ladder_conv3x3_up2_bwd_data_split's dx [N, H, W, Cout] made exact IN PLACE, from ONE d_up line per
 * border -- exact = main - D_r (x) M_c - M_r (x) D_c + D_r (x) D_c (csrc/convf32.hip): row -1 / 2H-1 and column -1 / 2W-1 of the backward-data of the
 * zero-padded dy as 1x3 / 2x3 / 3x1 / 3x2-tap convolutions over the first / last lines of dy [N, 2H, 2W, C]; w = the layer's HWIO bank
 * [3][3][Cout][C].  Replaces the four strip launches + ladder_conv3x3_up2_bwd_border (45 -> 9 line-taps per axis).  Same stream, after the
 * main launch. */
size_t ladder_conv3x3_up2_bwd_borders_workspace_bytes(int N, int H, int W, int C, int Cout);
int ladder_conv3x3_up2_bwd_borders(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int Cout, void* ws, size_t ws_bytes,
                                   ladder_stream_t stream);
/* The gated forms (round 5, strict fp32, the 8x32-pixel tiling): dx = act'(gate) x (d loss / d x_lo) with gate [N, H, W, Cout] = the ACTIVATED tensor
 * that x_lo is -- the activation backward of the layer below an un-normalised resize -> conv pair (decoder conv2d_5 -> conv2d_6,
 * codes/models.py:556-564: tf.gradients through tf.nn.leaky_relu) rides on the epilogue of the main launch and on the border fix-up; the separate
 * ladder_act_bwd pass over that tensor disappears.  bank_up2t as for ladder_conv3x3_up2_bwd_data_split with prec = LADDER_PREC_F32. */
int ladder_conv3x3_up2_bwd_data_gated_f32_eligible(int N, int H, int W, int C, int Cout);
int ladder_conv3x3_up2_bwd_data_gated_f32(const float* dy, const void* bank_up2t, float* dx, const float* gate, int gate_act, int N, int H, int W, int C,
                                          int Cout, ladder_stream_t stream);
int ladder_conv3x3_up2_bwd_borders_gated(const float* dy, const float* w, float* dx, const float* gate, int gate_act, int N, int H, int W, int C, int Cout,
                                         void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Filter gradient of the same pair over the LOW-resolution map (round 4, strict fp32; csrc/convf32.hip): dw [3][3][Cin][Cout] (and db [Cout],
 * may be NULL) of y = conv3x3_same(resize2x(x), w) from x [N, H, W, Cin] (x_upsampled != 0: x points at the materialised upsample
 * [N, 2H, 2W, Cin] a training forward keeps, read at its even rows / columns) and dy [N, 2H, 2W, Cout]: per output-parity class the 9 / 6 / 6 / 4
 * tap tiles G_ab[dr][dc] = sum x~[i+dr-1, j+dc-1] (x) dy[2i+a, 2j+b] (25 instead of the 36 the direct filter gradient on the upsampled map
 * accumulates), recombined with the tables of filterbank.h, plus the 1x3 / 3x1 filter gradients of the last output row / column.
 * Replaces ladder_conv2d_bwd_filter on the upsampled tensor (tf.gradients of codes/models.py:554-578 w.r.t. conv2d_6 / conv2d_7 kernels).
 * Cin % 64 == 0, Cout % 4 == 0; patches of 64 low-resolution pixels: 2 x 32 (W % 32 == 0, >= 1024 patches), 4 x 16 or 8 x 8 (round 5: W % 16 == 0 /
 * W % 8 == 0, >= 128 patches -- decoder conv2d_5 / conv2d_4, codes/models.py:544-560). */
int ladder_conv3x3_up2_wgrad_eligible(int N, int H, int W, int Cin, int Cout);
size_t ladder_conv3x3_up2_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int ladder_conv3x3_up2_wgrad(const float* x, int x_upsampled, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                             void* ws, size_t ws_bytes, ladder_stream_t stream);
/* The last output row and column of the call above, recomputed in fp32 from the last row / column of x and the layer's HWIO bank w
 * [3][3][Cin][Cout] (row 2H-1 sees x[H-1] twice -- the resize clamps -- and the zero padding below; two [N*2W, 3 Cin] x [3 Cin, Cout] GEMMs):
 * written to y [N, 2H, 2W, Cout] and / or, through the fused 1x1 projection pw [Cout][pco] + pb, to pout [N, 2H, 2W, pco]; y_absmax (the
 * record of the main launch) is raised where needed.  Call after ladder_conv3x3_up2_split(_proj) on the same stream. */
size_t ladder_conv3x3_up2_edges_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int ladder_conv3x3_up2_edges(const float* x, const float* w, const float* bias, float* y, float* y_absmax, const float* proj_w, const float* proj_b,
                             float* proj_out, int proj_cout, int N, int H, int W, int Cin, int Cout, int act, int x_upsampled, void* ws,
                             size_t ws_bytes, ladder_stream_t stream);

/* planes[p][i] = 16-bit plane p of x[i] (scaled by a power of two derived from x_absmax for LADDER_PREC_F16X3), plane-major,
 * n % 8 == 0, followed by 16 zero bytes (the source of out-of-image taps) and a 16-byte header; ladder_presplit_bytes = planes * n * 2 + 32.
 * n_samples > 0 (x is [n_samples, ...], (n / n_samples) % 8 == 0) AND a per-sample record (ladder_absmax_samples, or a producer that
 * emits one): every sample is scaled by ITS OWN maximum and header word 0 is set to 1 -- the gather kernels below read the header and
 * un-scale per output row (forward / backward-data) or re-scale their accumulators at sample boundaries (filter gradient; the caller
 * must then request per-sample planes only where a sample's output pixels are a multiple of 32).  n_samples = 0: one scale. */
size_t ladder_presplit_bytes(size_t n, int prec);
int ladder_presplit(const float* x, const float* x_absmax, void* planes, size_t n, int n_samples, int prec, ladder_stream_t stream);
/* The gather kernel on split operands: every other large convolution (128x128 output tiles; gathered channels % 32 == 0; tap table
 * <= 28 taps), i.e. the strided encoder layers and the 8x8 / 16x16 decoder maps (codes/models.py:398-460, 522-547) and their
 * backward-data passes (stride 2: the four output-parity classes).  Same semantics as ladder_conv2d_fwd / ladder_conv2d_bwd_data with
 * `packed` from ladder_filter_pack_split (transpose_flip = 1 for backward-data), the gathered tensor's absolute maximum and the gathered
 * tensor given as its PRE-SPLIT planes (ladder_presplit: these kernels re-read every input element once per tap and per output-channel
 * tile, so the fp32 -> 16-bit split is done once per tensor, not 18-36 times).
 * *_eligible: 1 when EVERY launch of the call runs on the split kernel (call the fp32 entry point otherwise). */
int ladder_conv2d_fwd_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                     int pad_l);
size_t ladder_conv2d_fwd_split_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                               int pad_t, int pad_l);
int ladder_conv2d_fwd_split(const void* x_planes, const float* x_absmax, const void* packed, const float* bias, float* y, int N, int H, int W,
                            int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int act, int prec,
                            void* ws, size_t ws_bytes, ladder_stream_t stream);
/* The forward call + the batch-norm statistics of its output from the epilogue (sums4 = the statistics record, minmax form: 2 Cout doubles sum | sum of squares, then min | max as 2 Cout floats, per
 * channel, as ladder_bn_fwd_stats_minmax); only for calls that run without split-K: the workspace query returns 0 otherwise. */
size_t ladder_conv2d_fwd_split_bnstats_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                                       int pad_t, int pad_l);
int ladder_conv2d_fwd_split_bnstats(const void* x_planes, const float* x_absmax, const void* packed, const float* bias, float* y, int N,
                                    int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l,
                                    int act, int prec, float* sums4, void* stats_ws, size_t stats_ws_bytes, ladder_stream_t stream);
/* The strict-fp32 form (round 4): ladder_conv2d_fwd (gather kernel, 128x128 tiles, no split-K) whose epilogue also leaves the batch-norm
 * statistics of y in sums4 (the statistics record in its minmax form: 6 Cout floats) per channel (reference: tf.layers.batch_normalization behind
 * tf.layers.conv2d, codes/models.py:398-460) -- the separate statistics pass over y disappears.  workspace_bytes == 0: the geometry does
 * not run on that kernel configuration (the caller falls back to ladder_conv2d_fwd + ladder_bn_fwd_stats). */
size_t ladder_conv2d_fwd_bnstats_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                                 int pad_l);
int ladder_conv2d_fwd_bnstats(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                              int KH, int KW, int stride, int pad_t, int pad_l, int act, float* sums4, void* stats_ws, size_t stats_ws_bytes,
                              ladder_stream_t stream);
int ladder_conv2d_bwd_data_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                          int pad_l, int gated);
size_t ladder_conv2d_bwd_data_split_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                                                    int pad_t, int pad_l);
int ladder_conv2d_bwd_data_split(const void* dy_planes, const float* dy_absmax, const void* packed_T, float* dx, int N, int H, int W, int Cin,
                                 int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, const float* gate_y,
                                 int gate_act, int prec, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Filter gradient of the same layers on split operands (Cin % 64 == 0, W % 32 == 0, >= 4096 row patches): dw[3][3][Cin][Cout] =
 * sum x (*) dy, db[co] = sum dy (db may be NULL).  The reduction index of the matrix instruction is the pixel: fragments are read
 * with the transposing LDS load (ds_read_b64_tr_b16) from channel-contiguous images.  Replaces ladder_conv2d_bwd_filter's call sites
 * for the decoder's 32x32 ... 128x128 maps. */
int ladder_conv3x3_wgrad_split_eligible(int N, int H, int W, int Cin, int Cout, int prec);
size_t ladder_conv3x3_wgrad_split_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int ladder_conv3x3_wgrad_split(const float* x, const float* x_absmax, const float* dy, const float* dy_absmax, float* dw, float* db,
                               int N, int H, int W, int Cin, int Cout, int prec, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Every other large filter gradient on split operands (Cin % 128 == 0, Cout > 64, >= 4096 output pixels; transposing LDS reads as
 * above; 128 x 128 tiles of dW, pixel range split over the grid, fixed-order second stage).  Same semantics as
 * ladder_conv2d_bwd_filter for the strided encoder layers and the 8x8 / 16x16 decoder maps. */
int ladder_conv2d_bwd_filter_split_eligible(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t,
                                            int pad_l);
size_t ladder_conv2d_bwd_filter_split_workspace_bytes(int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW);
int ladder_conv2d_bwd_filter_split(const void* x_planes, const float* x_absmax, const void* dy_planes, const float* dy_absmax, float* dw, float* db,
                                   int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad_t, int pad_l,
                                   int prec, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* out[i] = sum_s ws[s*n + i] in the fixed order s = 0..splits-1 (second stage of every split reduction). */
int ladder_reduce_splits(const float* ws, float* out, int splits, size_t n, ladder_stream_t stream);

/* ---------------------------------------------------------------- N2: tf.layers.dense
 * codes/models.py:73-95,109,231-253,267,478-488,501-510; codes/modules.py:8; codes/base.py:145-186.
 * y[M,N] = act(x[M,K] @ w[K,N] + b).  MFMA-f32 (v_mfma_f32_32x32x2_f32). */
int ladder_dense_fwd(const float* x, const float* w, const float* bias, float* y,
                     int M, int K, int N, int act, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* dx[M,K] = dy[M,N] @ w^T, wT = ladder_filter_flip_transpose(w, 1,1,K,N) i.e. [N,K]. */
int ladder_dense_bwd_data(const float* dy, const float* wT, float* dx, int M, int K, int N,
                          const float* gate_y, int gate_act, void* ws, size_t ws_bytes, ladder_stream_t stream);
size_t ladder_dense_bwd_weight_workspace_bytes(int M, int K, int N);
int ladder_dense_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                            int M, int K, int N, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* The same three calls for BATCH-SIZED layers (M <= 512 rows, M*N <= 2^20: the mapping MLP, style projections, inner VAE and encoder
 * heads, codes/models.py:478-510, codes/base.py:145-186) on the bf16 matrix cores in the fp32-class bf16x6 format (csrc/densesplit.hip):
 * ONE launch per call -- a 32x32 output tile per workgroup, operand fragments loaded straight from global memory in MFMA layout and
 * split in registers, no split-K pass, no workspace, no transposed weight copy (bwd_data takes w itself, [K,N]). */
int ladder_dense_small_eligible(int M, int K, int N);
int ladder_dense_fwd_small(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, ladder_stream_t stream);
int ladder_dense_bwd_data_small(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act,
                                ladder_stream_t stream);
int ladder_dense_bwd_weight_small(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, ladder_stream_t stream);
/* Both gradient calls of a layer in ONE launch (a launch costs as much as the arithmetic of one of them): dx and dw must not be NULL. */
int ladder_dense_bwd_small(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int M, int K, int N,
                           const float* gate_y, int gate_act, ladder_stream_t stream);
/* The same four calls in STRICT fp32 (round 4; the engine uses them when matmul_precision is "f32", the reference's arithmetic:
 * tf.layers.dense on fp32 tensors, codes/models.py:478-510, codes/base.py:145-186): every product on v_mfma_f32_32x32x2_f32 /
 * v_mfma_f32_16x16x4_f32 -- bit-exact fp32 FMA chains; same one-launch structure, same eligibility, deterministic fixed-order reduction
 * over the workgroup's wavefronts.  Replaces ladder_dense_fwd / _bwd_data / _bwd_weight (+ ladder_filter_flip_transpose + the split-K
 * second pass) for batch-sized layers. */
int ladder_dense_fwd_small_f32(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, ladder_stream_t stream);
int ladder_dense_bwd_data_small_f32(const float* dy, const float* w, float* dx, int M, int K, int N, const float* gate_y, int gate_act,
                                    ladder_stream_t stream);
int ladder_dense_bwd_weight_small_f32(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, ladder_stream_t stream);
int ladder_dense_bwd_small_f32(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int M, int K, int N,
                               const float* gate_y, int gate_act, ladder_stream_t stream);


/* The image-side convolution of the CelebA encoder (codes/models.py:398-405: 3x3, stride 2, SAME, 3 -> Cout channels over even-sized RGB
 * maps; pad_t == pad_l == 0, i.e. the SAME padding is the bottom row / right column) on the fp16 matrix cores in the fp32-class f16x3
 * format with per-workgroup scales (csrc/convrgb.hip): forward writes y = act(conv(x, w) + bias); the filter gradient reduces over pixel
 * runs into `ws` and sums them in a fixed order (db = column sums of dy, may be NULL).  (H/2) % 8 == 0, (W/2) % 32 == 0, Cout % 4 == 0. */
int ladder_conv_rgb_s2_eligible(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l);
int ladder_conv_rgb_s2_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                           ladder_stream_t stream);
/* The forward call + the batch-norm statistics of its output in the same launch (per-patch column statistics from the epilogue, reduced in
 * a fixed order): sums4 as ladder_bn_fwd_stats_minmax would produce them from a second pass over y. */
size_t ladder_conv_rgb_s2_fwd_bnstats_workspace_bytes(int N, int H, int W, int Cout);
int ladder_conv_rgb_s2_fwd_bnstats(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                                   float* sums4 /*[4 Cout]: sum | sum of squares | min | max*/, void* ws, size_t ws_bytes,
                                   ladder_stream_t stream);
/* dw [3,3,3,Cout], db [Cout] (may be NULL); x_absmax / dy_absmax = the tensors' absolute-maximum records (ladder_absmax or a producer's). */
/* Strict-fp32 instantiations of the two forward calls above (round 4): the im2col matrix and the filter stay fp32 in LDS, 14 K-steps of
 * v_mfma_f32_32x32x2_f32, no operand scaling; same arguments, same statistics workspace. */
int ladder_conv_rgb_s2_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                               ladder_stream_t stream);
int ladder_conv_rgb_s2_fwd_bnstats_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act,
                                       float* sums, void* ws, size_t ws_bytes, ladder_stream_t stream);
/* Strict-fp32 filter (db != NULL: and bias) gradient of the same layer (reference: tf.gradients of the loss with respect to the first encoder
 * conv2d's kernel / bias, models.py:398-405): fp32 im2col image in LDS, dY fragments straight from global memory, fp32 MFMA; no absmax records;
 * workspace = ladder_conv_rgb_s2_bwd_filter_workspace_bytes. */
int ladder_conv_rgb_s2_bwd_filter_f32(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cout, void* ws,
                                      size_t ws_bytes, ladder_stream_t stream);
size_t ladder_conv_rgb_s2_bwd_filter_workspace_bytes(int N, int H, int W, int Cout);
int ladder_conv_rgb_s2_bwd_filter(const float* x, const float* x_absmax, const float* dy, const float* dy_absmax, float* dw, float* db,
                                  int N, int H, int W, int Cout, void* ws, size_t ws_bytes, ladder_stream_t stream);

/* ---------------------------------------------------------------- N12: activations (backward)
 * dx = dy * act'(y) evaluated from the activation OUTPUT y (leaky/relu/tanh).  In-place (dx==dy) allowed. */
int ladder_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, ladder_stream_t stream);

/* ---------------------------------------------------------------- N3: tf.layers.batch_normalization(training=True)
 * codes/models.py:398-460 (is_training is the constant True, models.py:471).  Two-phase so that the host can
 * all-reduce the 2C statistics between the phases (data-parallel: statistics of the GLOBAL batch).
 * x is [rows, C] with rows = N*H*W.  eps = 1e-3 in the reference (TF default). */
size_t ladder_bn_workspace_bytes(size_t rows, int C);
/* THE STATISTICS RECORD (round 5; `sums` / `sums4` of every entry point below, 8-byte aligned): 2C DOUBLES = sum_rows x | sum_rows x^2 -- the
 * first 4C floats of the buffer -- and, in the "minmax" form, min x | max x as 2C floats behind them (6C floats in all).  TF's fused batch
 * norm forms the variance about the mean; E[x^2] - mean^2 is that exact only in fp64 (relative error eps x (1 + mean^2 / var): rounds 1-4 kept
 * the sums in fp32 and lost 1.5e-4 of the variance of a channel 50 standard deviations off zero).  ladder_bn_fwd_stats(_minmax) accumulate
 * every element in fp64 (fixed order); the data-parallel exchange (C2) all-reduces the 2C doubles. */
int ladder_bn_fwd_stats(const float* x, float* sums /*record: 4C floats*/, size_t rows, int C,
                        void* ws, size_t ws_bytes, ladder_stream_t stream);
/* mean = sum/count ; var = sum of squares/count - mean^2 (biased, fp64) ; y = act(gamma*(x-mean)*rsqrt(var+eps)+beta).
 * Writes mean_rstd[0:C]=mean, [C:2C]=rstd.  `count` = GLOBAL row count (after the all-reduce). */
/* Second stage alone: the record from per-block fp32 partials [nblk][2][C] (what a convolution epilogue emits), fixed order, fp64. */
int ladder_bn_stats_from_partials(const float* partials, int nblk, float* sums /*record: 4C floats*/, int C, ladder_stream_t stream);
/* Statistics with the per-channel extremes: sums4 = the record in its minmax form (6C floats; only the 2C doubles are summed across ranks).
 * Workspace: 2 x ladder_bn_workspace_bytes.  C % 4 == 0. */
int ladder_bn_fwd_stats_minmax(const float* x, float* sums4, size_t rows, int C, void* ws, size_t ws_bytes, ladder_stream_t stream);
int ladder_bn_stats_minmax_from_partials(const float* partials /*[nblk][4][C]*/, int nblk, float* sums4, int C, ladder_stream_t stream);
/* ladder_bn_fwd_apply that writes y as the two fp16 PLANES the split gather kernels read (ladder_presplit layout for LADDER_PREC_F16X3) and
 * the record of max|y| it scaled them with -- known in advance because y is monotone in x per channel, so max|y| sits at a channel's min or
 * max.  y (fp32) is optional: NULL = never written (the consumer convolutions read only the planes).  rows*C % 8 == 0, C % 4 == 0. */
int ladder_bn_fwd_apply_planes(const float* x, const float* sums4, double count, const float* gamma, const float* beta, float* y,
                               void* y_planes, float* mean_rstd, size_t rows, int C, float eps, int act, float* y_absmax,
                               ladder_stream_t stream);
int ladder_bn_fwd_apply(const float* x, const float* sums, double count,
                        const float* gamma, const float* beta, float* y, float* mean_rstd,
                        size_t rows, int C, float eps, int act, ladder_stream_t stream);
/* dp = dy*act'(pre), pre = gamma*xhat+beta ;  dsums[0:C] = sum dp, dsums[C:2C] = sum dp*xhat. */
int ladder_bn_bwd_stats(const float* dy, const float* x, const float* mean_rstd, const float* gamma,
                        const float* beta, float* dsums /*[2C]*/, size_t rows, int C, int act,
                        void* ws, size_t ws_bytes, ladder_stream_t stream);
/* dx = gamma*rstd*(dp - dsums[c]/count - xhat*dsums[C+c]/count); dgamma = dsums[C+c]; dbeta = dsums[c].
 * dx may be NULL (first layer), dgamma/dbeta may be NULL. */
int ladder_bn_bwd_apply(const float* dy, const float* x, const float* mean_rstd, const float* gamma,
                        const float* beta, const float* dsums, double count, float* dx,
                        float* dgamma, float* dbeta, size_t rows, int C, int act, ladder_stream_t stream);
/* The two apply calls, additionally producing the absolute-maximum record (ladder_absmax, section N1s) of the tensor they write --
 * y resp. dx -- for the split-precision convolution that consumes it (C % 4 == 0, 16-byte aligned tensors; dx not NULL). */
int ladder_bn_fwd_apply_absmax(const float* x, const float* sums, double count, const float* gamma, const float* beta, float* y,
                               float* mean_rstd, size_t rows, int C, float eps, int act, float* y_absmax, ladder_stream_t stream);
int ladder_bn_bwd_apply_absmax(const float* dy, const float* x, const float* mean_rstd, const float* gamma, const float* beta,
                               const float* dsums, double count, float* dx, float* dgamma, float* dbeta, size_t rows, int C, int act,
                               float* dx_absmax, ladder_stream_t stream);

/* ---------------------------------------------------------------- N4+N5+N12: instance_norm + style_mod + leaky
 * codes/models.py:522-528,531-537,547-554,564-571; codes/modules.py:6-10.
 * x [N,HW,C]; style [N,2C] (raw output of the StyleMod dense: [:,0:C] scale-1, [:,C:2C] shift).
 * y = act(((x-mean)*rstd) * (style0+1) + style1), moments over HW per (n,c), biased var, eps.
 * With a workspace (ladder_in_style_workspace_bytes) and C%4==0 the H*W axis is split over several workgroups per
 * (sample, 64 channels) and streamed with 16-byte loads (fixed-order second stage); ws == NULL selects the
 * one-workgroup-per-slab kernels. */
size_t ladder_in_style_workspace_bytes(int N, int HW, int C);
int ladder_in_style_fwd(const float* x, const float* style, float* y, float* mean_rstd /*[N,2C]*/,
                        int N, int HW, int C, float eps, int act, void* ws, size_t ws_bytes, ladder_stream_t stream);
int ladder_in_style_bwd(const float* dy, const float* x, const float* style, const float* mean_rstd,
                        float* dx, float* dstyle /*[N,2C]*/, int N, int HW, int C, int act,
                        void* ws, size_t ws_bytes, ladder_stream_t stream);
/* The same two calls, additionally producing the absolute-maximum record (ladder_absmax) of the tensor they write -- y resp. dx --
 * for the split-precision convolution that consumes it (C % 4 == 0, workspace supplied). */
int ladder_in_style_fwd_absmax(const float* x, const float* style, float* y, float* mean_rstd, int N, int HW, int C, float eps, int act,
                               void* ws, size_t ws_bytes, float* y_absmax, ladder_stream_t stream);
int ladder_in_style_bwd_absmax(const float* dy, const float* x, const float* style, const float* mean_rstd, float* dx, float* dstyle,
                               int N, int HW, int C, int act, void* ws, size_t ws_bytes, float* dx_absmax, ladder_stream_t stream);
/* Forward fused with the factor-2 tf.image.resize_images that follows it in the CelebA decoder (models.py:528-538, 554-561, 571-578):
 * up [N,2H,2W,C] = resize(act(style_mod(instance_norm(x)))), bit-identical to ladder_in_style_fwd + ladder_resize_bilinear_fwd; the
 * normalised tensor is never written.  C % 4 == 0, workspace required; up_absmax (may be NULL) receives the record of max|up|. */
int ladder_in_style_fwd_resize2x(const float* x, const float* style, float* up, float* mean_rstd, int N, int H, int W, int C, float eps,
                                 int act, void* ws, size_t ws_bytes, float* up_absmax, ladder_stream_t stream);
/* the same, also writing the normalised / styled / activated tensor y [N, H, W, C] itself (= the even rows and columns of `up`; may be NULL) */
int ladder_in_style_fwd_resize2x_keep(const float* x, const float* style, float* up, float* y, float* mean_rstd, int N, int H, int W, int C,
                                      float eps, int act, void* ws, size_t ws_bytes, float* up_absmax, ladder_stream_t stream);

/* ---------------------------------------------------------------- N6: tf.image.resize_images (TF1 legacy bilinear)
 * codes/models.py:519,538,544,555,561,572,578.  align_corners=False, half_pixel_centers=False;
 * OH/H and OW/W must be integers (1->2, 2->8, 8->16, ... in the reference). */
int ladder_resize_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, ladder_stream_t stream);
int ladder_resize_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, ladder_stream_t stream);
/* The factor-2 transpose with the activation backward of the producing layer fused in: dx *= act'(gate_y), gate_y = that layer's output
 * [N,H,W,C] (codes/models.py:538-545, 561-568: leaky conv -> resize). */
int ladder_resize_bilinear_bwd_gated(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, const float* gate_y,
                                     int gate_act, ladder_stream_t stream);

/* ---------------------------------------------------------------- N7: depth_to_space (DCR) / tf.pad SYMMETRIC
 * codes/models.py:48-50,113,122,131,140,200-202,271,...,307.  inverse!=0 gives space_to_depth (the backward). */
int ladder_depth_to_space(const float* x, float* y, int N, int H, int W, int C, int r, int inverse, ladder_stream_t stream);
int ladder_pad_symmetric(const float* x, float* y, int N, int H, int W, int C, int p, ladder_stream_t stream);
/* its transpose (gradient w.r.t. the unpadded input; only the VampPrior pseudo-inputs need it). */
int ladder_pad_symmetric_bwd(const float* dy, float* dx, int N, int H, int W, int C, int p, ladder_stream_t stream);

/* ---------------------------------------------------------------- N8: reparameterised sampling
 * codes/models.py:97-103,255-262,490-497; codes/base.py:164-167.  Philox4x32-10 + Box-Muller normals. */
int ladder_randn(float* out, size_t n, uint64_t seed, uint64_t offset, ladder_stream_t stream);

/* ---------------------------------------------------------------- N9: tfd.Mixture(...).log_prob + gradient
 * codes/base.py:109-124, 308-313.
 * prepare: per component Cholesky L_k of cov_k (fp32), packed[k] = { logw_k - sum log diag L_k - R/2 log 2pi,
 *          mean_k[R], Linv_k (row-major lower-triangular, R*(R+1)/2) } ; stride = ladder_gmm_packed_stride(R). */
int ladder_gmm_packed_stride(int R);
int ladder_gmm_prepare(const float* weights, const float* means, const float* covs, int K, int R,
                       float* packed, ladder_stream_t stream);
/* R in 1..8.  t[l,b,:] = mu[b,:] + sd[b,:]*eps[l,b,:];  lp = logsumexp_k(...).  Outputs: sum_logp[0] = sum_{l,b} lp;
 * dmu[b,:] = sum_l dlp/dt ; dsd[b,:] = sum_l dlp/dt * eps   (un-normalised; host scales by 1/(L*B_global)).
 * lane = mixture component; K <= 64: every lane keeps its component's parameters in registers, the L samples of a batch row are
 * spread over ~2048 / B wavefronts, a sample costs two wave-shuffle reductions (max, sum of exponentials), the gradient terms
 * accumulate per lane and are reduced once per wavefront; partials in `ws` are summed in a fixed order.  K > 64: one workgroup per
 * batch row, components in chunks of 64. */
size_t ladder_gmm_workspace_bytes(int L, int B);
int ladder_gmm_logprob_fwd_bwd(const float* mu, const float* sd, const float* eps, const float* packed,
                               int L, int B, int R, int K, float* sum_logp, float* dmu, float* dsd,
                               void* ws, size_t ws_bytes, ladder_stream_t stream);

/* log p(t_i) of n separate points t [n,R] under the prepared mixture (R in 1..8), one wavefront per point.  Replaces
 * `sess.run(prior.prob(pos))` / `prior.log_prob` on the tfd.Mixture the reference's demo builds (demo/demo_tools.py:88-99, 265). */
int ladder_gmm_logprob_rows(const float* t, const float* packed, int n, int R, int K, float* logp, ladder_stream_t stream);

/* ---------------------------------------------------------------- N10: ELBO reductions + scalar algebra
 * codes/base.py:262-305,374-402; codes/models.py:152-159,319-326,591-597.
 *
 * Partial sums live in ONE device vector `partials` (floats) so the host can all-reduce it in one call:
 *   [LADDER_P_*] fixed slots, then Z floats (sum_b sd_z[b,:]) and R floats (sum_b sd_t[b,:]).
 * Fetched scalars + backward coefficients live in the device vector `scalars` ([LADDER_S_*]). */
enum { LADDER_P_PIX_ABS = 0, LADDER_P_PIX_SQ = 1, LADDER_P_LOG_SDZ = 2, LADDER_P_MU2SD2_Z = 3, LADDER_P_CODE_ERR = 4,
       LADDER_P_CODE_SQRT = 5, LADDER_P_CODE_ABS = 6, LADDER_P_LOG_SDT = 7, LADDER_P_MU2SD2_T = 8, LADDER_P_LOGP = 9,
       LADDER_P_FIXED = 16 };
enum { LADDER_S_SIGMA = 0, LADDER_S_MPE = 1, LADDER_S_ENTROPY_Z = 2, LADDER_S_XENT_SG = 3, LADDER_S_XENT_PRIOR = 4,
       LADDER_S_L1 = 5, LADDER_S_L2 = 6, LADDER_S_RECON_LL = 7, LADDER_S_SIGMA_REG = 8, LADDER_S_ELBO = 9,
       LADDER_S_LOSS_AE = 10, LADDER_S_INNER_SIGMA = 11, LADDER_S_MEAN_CODE_ERROR = 12, LADDER_S_CODE_LL = 13,
       LADDER_S_CODE_L1 = 14, LADDER_S_REP_REG = 15, LADDER_S_ENTROPY_T = 16, LADDER_S_XENT_T = 17,
       LADDER_S_ELBO_PRIOR = 18, LADDER_S_LOSS_PRIOR = 19,
       /* backward coefficients, already divided by the GLOBAL batch */
       LADDER_S_G_PIX = 20,             /* d loss_ae/d xhat = G_PIX * sign(xhat-x) (incl. the sigma=max(.,mpe) path) */
       LADDER_S_G_SIGMA_VAR = 21,       /* d loss_ae / d sigma/Variable                                             */
       LADDER_S_G_CODE = 22,            /* d loss / d err[b,j] = 1/(2 inner_sigma^2 B)                               */
       LADDER_S_G_INNER_SIGMA_VAR = 23, /* d loss_prior / d inner_sigma/Variable                                     */
       LADDER_S_INV_B = 24, LADDER_S_INV_LB = 25, LADDER_S_COUNT = 32 };
typedef struct {
  int B_global, D, Z, R, L;
  int sigma_uses_mpe;    /* celeba: 1 ; mnist: TRAIN_sigma (models.py:158,325,597) */
  int has_inner;         /* prior in {"ours", "hierarchical"} */
  int use_sg;            /* use_standard_gaussian_prior feed (base.py:318-320) */
  int clamp_inner_sigma; /* TRAIN_inner_sigma (base.py:210-212) */
  float inner_sigma_lb, inner_sigma_ub;
  int hierarchical;      /* prior == "hierarchical" (base.py:331-359): crossEntropy_representation in closed form against N(0,I)
                          * from P_MU2SD2_T (no mixture term), entropy_t with the reference's hard-coded dimension 2 */
  int prior_gmm;         /* prior == "GMM" (base.py:322-329): crossEntropy_prior = P_LOGP / (L*B), the MC mean of the mixture
                          * log-prob of z samples; no standard-Gaussian switch */
} LadderElboCfg;

/* out[0] = sum |x-xhat|, out[1] = sum (x-xhat)^2 over n elements (fp64 accumulation across workgroups). */
size_t ladder_pixel_partials_workspace_bytes(size_t n);
int ladder_pixel_partials(const float* x, const float* xhat, size_t n, float* out /*[2]*/,
                          void* ws, size_t ws_bytes, ladder_stream_t stream);
/* dxhat = coef[0] * sign(xhat - x); coef is a DEVICE scalar (&scalars[LADDER_S_G_PIX]). */
int ladder_pixel_grad(const float* x, const float* xhat, const float* coef, float* dxhat, size_t n, ladder_stream_t stream);
/* Latent block (models.py:95-103, base.py:162-167, 269-280, 302-305): sd = sd_raw + lvp ; z = mu + sd*eps (z may be NULL).
 * p_log[0] = sum_bj log sd ; p_mu2sd2[0] = sum_bj (mu^2 + sd^2) ; p_sdsum[j] = sum_b sd[b,j] (may be NULL). */
int ladder_latent_fwd(const float* mu, const float* sd_raw, const float* eps, float lvp, float* z, float* sd,
                      float* p_log, float* p_mu2sd2, float* p_sdsum, int B, int Z, ladder_stream_t stream);
/* base.py:286-297: err = (z-zhat)^2 (0 where use_mask && sd_z>1). out[0]=sum err, out[1]=sum sqrt(err), out[2]=sum|z-zhat|. */
int ladder_code_partials(const float* z, const float* zhat, const float* sd_z, int use_mask,
                         float* out /*[3]*/, int B, int Z, ladder_stream_t stream);
/* The scalar algebra of define_loss (base.py:257-413) + the sigma / inner_sigma blocks, on device. */
int ladder_elbo_finalize(const float* partials, const float* sigma_var, const float* inner_sigma_var,
                         LadderElboCfg cfg, float* scalars, ladder_stream_t stream);
/* dz_accum += 2*G_CODE*(z-zhat)*mask (dz_accum may be NULL) ; dzhat = -2*G_CODE*(z-zhat)*mask. */
int ladder_code_grad(const float* z, const float* zhat, const float* sd_z, int use_mask, const float* scalars,
                     float* dz_accum, float* dzhat, int B, int Z, ladder_stream_t stream);
/* Gradient of a reparameterised latent block w.r.t. its heads:
 *   dmu = g_sample [+ mu*INV_B if mode&2] + extra_sign*INV_LB*extra_mu
 *   dsd = g_sample*eps [- INV_B/sd if mode&1] [+ sd*INV_B if mode&2] + extra_sign*INV_LB*extra_sd
 *   dsd_raw = dsd * (sd_raw > 0)     (relu of the std head)
 * mode bit0 = entropy term, bit1 = standard-Gaussian cross-entropy term; g_sample / extra_* may be NULL. */
int ladder_latent_bwd(const float* g_sample, const float* mu, const float* sd, const float* sd_raw, const float* eps,
                      const float* extra_mu, const float* extra_sd, float extra_sign, const float* scalars, int mode,
                      float* dmu, float* dsd_raw, int B, int Z, ladder_stream_t stream);

/* ---------------------------------------------------------------- N11: tf.train.AdamOptimizer + clip_by_value
 * codes/base.py:457-517.  g <- clip(g,-clip,clip); m,v update; theta -= lr_t * m/(sqrt(v)+eps) with
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller (TF form).  Flat buffers: one launch per optimiser.
 * `g` may point into the device `scalars` vector for the two 1-element sigma optimisers. */
int ladder_adam_clip(float* theta, const float* g, float* m, float* v, size_t n,
                     float lr_t, float beta1, float beta2, float eps, float clip, ladder_stream_t stream);

/* hipGraph-friendly variants: every per-step scalar lives in DEVICE memory, so a captured run can be replayed verbatim.
 *   state = {float lr, float lr_t, int step}: the call first bumps `step` and refreshes lr_t = lr*sqrt(1-b2^t)/(1-b1^t) on
 *   the device, then applies clip+Adam with it.  The host only rewrites state[0] when the learning rate changes. */
int ladder_adam_clip_dev(float* theta, const float* g, float* m, float* v, size_t n, float* state,
                         float beta1, float beta2, float eps, float clip, ladder_stream_t stream);
/* Philox stream position = *offset_base + offset_add (offset_base: device counter advanced by ladder_u64_add). */
int ladder_randn_dev(float* out, size_t n, uint64_t seed, const uint64_t* offset_base, uint64_t offset_add, ladder_stream_t stream);
int ladder_u64_add(uint64_t* p, uint64_t inc, ladder_stream_t stream);

/* ---- the same mixture on a WIDE latent (prior "GMM": the mixture sits on z, R = code_size; codes/base.py:101-106, 322-329).
 * 8 < R <= 64, R % 4 == 0.  The whitening of all components is one GEMM on the dense MFMA kernel (see csrc/elbo.hip);
 * params = ladder_gmm_dense_param_floats(K,R) floats filled by ladder_gmm_prepare_dense (float64 Cholesky per component).
 * Same outputs / conventions as ladder_gmm_logprob_fwd_bwd; dmu = dsd = NULL evaluates the log-prob only. */
size_t ladder_gmm_dense_param_floats(int K, int R);
int ladder_gmm_prepare_dense(const float* weights, const float* means, const float* covs, int K, int R, float* params,
                             ladder_stream_t stream);
size_t ladder_gmm_dense_workspace_bytes(int L, int B, int R, int K);
int ladder_gmm_dense_logprob_fwd_bwd(const float* mu, const float* sd, const float* eps, const float* params, int L, int B, int R,
                                     int K, float* sum_logp, float* dmu, float* dsd, void* ws, size_t ws_bytes,
                                     ladder_stream_t stream);

/* per-point log p(t_i), t [n,R], on the wide-latent path (same workspace size as the training call with L = 1, B = n). */
int ladder_gmm_dense_logprob_rows(const float* t, const float* params, int n, int R, int K, float* logp, void* ws, size_t ws_bytes,
                                  ladder_stream_t stream);

/* ---- VampPrior (codes/base.py:216-254, 361-370): equally weighted mixture of K DIAGONAL Gaussians on z (Z <= 64) whose
 * components (comp_mean, comp_sd [K,Z]) are the encoder's outputs on the trainable pseudo-inputs.  sum_logp = sum over the L*B MC
 * samples of log p(z); dmu/dsd as in ladder_gmm_logprob_fwd_bwd; dcomp_mean/dcomp_sd [K,Z] = sum over samples of
 * dlog p/d comp_mean, dlog p/d comp_sd (fixed-order reduction) -- the gradient entering the pseudo-input encoder pass. */
size_t ladder_diag_mixture_workspace_bytes(int B, int Z, int K);
int ladder_diag_mixture_fwd_bwd(const float* mu, const float* sd, const float* eps, const float* comp_mean, const float* comp_sd,
                                int L, int B, int Z, int K, float* sum_logp, float* dmu, float* dsd, float* dcomp_mean,
                                float* dcomp_sd, void* ws, size_t ws_bytes, ladder_stream_t stream);

/* ---------------------------------------------------------------- N14: minibatch assembly (the input pipeline)
 * models.py:354-371 (CelebA: uint8 HWC pixels * 1/255), data_loader.py:19-33 (MNIST floats), shuffle + batch of models.py:33-40:
 * out[b, :] = scale * float(src[idx[b], :]) for a data set resident in device memory; src is uint8 (src_is_u8) or float32 rows of
 * D elements, idx = int64 row indices of the minibatch (a slice of the epoch's permutation). */
int ladder_gather_rows(const void* src, int src_is_u8, const int64_t* idx, float* out, int B, int64_t D, float scale,
                       ladder_stream_t stream);

/* ---------------------------------------------------------------- N13: the mixture fit that produces the hyper-prior feed
 * sklearn.mixture.BayesianGaussianMixture(K, 'full', weight_concentration_prior_type=..., weight_concentration_prior=0.1,
 * warm_start=True).fit(samples) of codes/base.py:93-99 (per-epoch "fast" fit, 681-721) and 723-789 ("accurate" fit), as ONE
 * persistent-workgroup launch running the whole variational loop in float64 (csrc/vbgmm.hip).
 *   X [N,R] fp32 samples (R <= 8, K <= 64, N >= K); labels [N] int32 = hard initial assignment (k-means labels) or NULL to
 *   warm-start from `state` (ladder_vbgmm_state_doubles(K,R) doubles, caller-owned, persists between fits; its last four
 *   entries are lower_bound_, n_iter_, converged_ (-1 = ill-defined covariance, sklearn raises ValueError there) and the `done` flag of the
 *   sharded fit below).
 *   prior_type 0 = dirichlet_distribution, 1 = dirichlet_process.  mean/covariance priors are taken from X as sklearn does.
 *   Outputs weights [K], means [K,R], covs [K,R,R] fp32 = weights_, means_, covariances_ (float64 copies live in `state`). */
size_t ladder_vbgmm_state_doubles(int K, int R);
size_t ladder_vbgmm_workspace_bytes(int N, int K);
int ladder_vbgmm_fit(const float* X, int N, int K, int R, const int* labels, double* state, int prior_type, double wc_prior,
                     double mean_prec_prior, double reg_covar, double tol, int max_iter, float* weights, float* means,
                     float* covs, void* ws, size_t ws_bytes, ladder_stream_t stream);

/* The same fit SHARDED over data-parallel ranks (exchange step C5: "mixture sufficient statistics all-reduce"): every rank holds only its
 * samples X [N_local, R].  Per variational iteration `it` (0 = the M-step on the hard initial labels, then 1 .. max_iter):
 *     ladder_vbgmm_shard_estep(...)  ->  stats [ladder_vbgmm_shard_stats_doubles] = local sufficient statistics
 *     all-reduce(stats, SUM)             (the caller: torch.distributed over RCCL)
 *     ladder_vbgmm_shard_mstep(...)  ->  M-step from the global statistics, lower bound, convergence test into `state`
 * with the data-derived priors from ladder_vbgmm_shard_moments (all-reduced once).  state[-1] != 0 marks the end of the fit (converged,
 * max_iter reached or state[-2] = -1: ill-defined covariance); both step kernels are no-ops from then on, so the caller may enqueue
 * iterations ahead and read the flag every few iterations.  The caller clears state[-2] and state[-1] before a (warm-started) fit.
 * labels: hard assignment of the LOCAL samples for it = 0, NULL afterwards.
 * The E-step walks the local samples in slices of 256, one workgroup each, and reduces the slice statistics in a fixed order (the same
 * sum the ranks then form): `ws` = ladder_vbgmm_shard_workspace_bytes(N_local, K, R) = responsibilities + per-slice statistics.  With an
 * identity "all-reduce" this is also the single-GPU fit for LARGE sample counts (the reference's accurate fit, codes/base.py:723-789:
 * 20 096 samples, up to 2 000 iterations), where the one-workgroup persistent kernel above spends 3.8 ms per iteration. */
size_t ladder_vbgmm_shard_stats_doubles(int K, int R);
size_t ladder_vbgmm_shard_workspace_bytes(int N, int K, int R);
size_t ladder_vbgmm_shard_moments_doubles(int R);
int ladder_vbgmm_shard_moments(const float* X, int N, int R, double* moments, ladder_stream_t stream);
int ladder_vbgmm_shard_estep(const float* X, int N, int K, int R, const int* labels, const double* state, int prior_type, double* stats,
                             void* ws, size_t ws_bytes, ladder_stream_t stream);
int ladder_vbgmm_shard_mstep(const double* stats, const double* moments, int K, int R, double* state, int prior_type, double wc_prior,
                             double mean_prec_prior, double reg_covar, double tol, int max_iter, int it, float* weights, float* means,
                             float* covs, ladder_stream_t stream);

/* ---------------------------------------------------------------- helpers */
/* HOST function (no device work): CRC-32C (Castagnoli) of host memory, crc = 0 to start, chainable.  Used by the
 * tf.train.Saver checkpoint-v2 reader/writer (codes/base.py:37-85: saver_ae / saver_prior) for block and tensor checksums. */
uint32_t ladder_crc32c_extend(uint32_t crc, const void* data, size_t n);
/* accumulate 0: out[i] = scale * in[i] (in == out: in-place scaling); 1: out[i] += scale * in[i]; 2: out[i] = scale (fill; `in` is not read). */
int ladder_axpy(const float* in, float* out, size_t n, float scale, int accumulate, ladder_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LADDER_HIP_H */
