"""HIP-event timing of the contraction launches, grouped by the kernel they dispatch to (bench.py's roofline leg)."""
import torch


class KernelProfiler:
    """HIP-event timing of the forward-type contraction launches, grouped by the kernel they dispatch to (bench.py's
    roofline leg).  Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    NAMES = {256128: "conv3x3_halo_kernel (8x32 px x 128 ch LDS-halo tile, 8 waves, fp32 MFMA 32x32x2)",
             256120: "conv3x3_halo_f32_kernel<PROJ, UPM> (8x32 px x 128 ch LDS-halo tile, 8 waves, fp32 MFMA 32x32x2; instantiations: plain, "
                     "fused 1x1 projection, upsample-fused forward with / without projection, upsample-fused backward-data, stride-2 backward-data)",
             256123: "conv3x3_halo_split_kernel<bf16x6> (8x32 px x 128 ch LDS-halo tile, 8 waves, fp32 operands as 3 bf16 planes, 6 x MFMA 32x32x16 bf16)",
             256122: "conv3x3_halo_split16_kernel / conv3x3_halo_split_kernel<bf16x3> (16x32 px x 128 ch LDS-halo tile, 16 waves -- 8x32, 8 waves below 512 tiles; fp32 operands as 2 bf16 planes, 3 x MFMA 32x32x16 bf16)",
             256124: "conv3x3_halo_split16_kernel / conv3x3_halo_split_kernel<f16x3> (16x32 px x 128 ch LDS-halo tile, 16 waves -- 8x32, 8 waves below 512 tiles; fp32 operands as 2 scaled fp16 planes, 3 x MFMA 32x32x16 f16)",
             128128: "igemm_fwd_kernel<128,128,2,2,true,true> (gather implicit GEMM, fp32 MFMA 32x32x2)",
             64064: "igemm_fwd_kernel<64,64,2,2,true,true> (implicit GEMM, 64x64 tile, fp32 MFMA 32x32x2; here: the 128- / 512-row GEMMs of the projected pairs conv2d_1 / conv2d_3)",
             128124: "igemm_fwd_split_kernel<f16x3> (gather implicit GEMM, 128x128 tile, 3 x MFMA 32x32x16 f16)",
             128122: "igemm_fwd_split_kernel<bf16x3>", 128123: "igemm_fwd_split_kernel<bf16x6>",
             9003: "conv_smallcin_kernel (direct 1x1 from 3 channels, HBM-bound)",
             9124: "wgrad3x3_split_kernel<f16x3> (64ci x 128co slab x 9 taps, 8 waves x 9 tiles, transposing LDS reads, 3 x MFMA 32x32x16 f16; incl. its split reduction)",
             9122: "wgrad3x3_split_kernel<bf16x3>", 9123: "wgrad3x3_split_kernel<bf16x6>",
             9128: "wgrad3x3_halo_kernel (64ci x 128co slab x 9 taps, 12 waves, LDS-DMA staged 1x32-pixel patches, fp32 MFMA 32x32x2)",
             256064: "conv3x3_halo_f32s_kernel<UPM, GEO, NI, FKS> (small maps: 16x16 / 8x16 / 8x8-px sub-patches x 64 / 128 ch LDS-halo tile, 8 waves, fp32 MFMA "
                     "32x32x2; instantiations: plain, upsample-fused forward (class pairs), parity-class input (upsample-fused backward-data, stride-2 forward), "
                     "stride-2 backward-data classes)",
             9120: "wgrad3x3_up2_f32_kernel + reductions / edge lines (filter gradient of resize x2 -> 3x3 conv over the low-resolution map: 25 of 36 tap tiles, "
                   "one parity class per workgroup, 12 waves, LDS-DMA staged 1x32-pixel patches, fp32 MFMA 32x32x2)",
             9130: "igemm_wgrad_kernel<128,128> + its fixed-order split reduction (dWcat [Cin][9 Cout] = x^T D of the project-then-upsample pairs, fp32 MFMA 32x32x2)",
             9132: "gemm_tn_f32_kernel + its fixed-order split reduction (dWcat [Cin][9 Cout] = x^T D of the project-then-upsample pairs: 128x128 tile x pixel range "
                   "per workgroup, 32-pixel chunks, software-pipelined LDS fragments, fp32 MFMA 32x32x2)",
             128132: "gemm_f32_kernel (projection GEMMs of the project-then-upsample pairs, Z = x . wcat: persistent workgroups over 128x128 "
                     "tiles, XCD-aware tile order, 32-deep chunks, 128-bit A fragments, software-pipelined LDS reads, fp32 MFMA 32x32x2)",
             128136: "up2proj_fused_fwd_kernel / up2proj_fused2_fwd_kernel (forward of a project-then-upsample pair in ONE launch: per (few images, 16 output channels) "
                     "workgroup the projection GEMM Z = x . wcat row by row on fp32 MFMA 16x16x4 by four MFMA-only waves -- weight slab resident in LDS at Cin 128, "
                     "32-pixel wave tiles with rolled fragments above --, one row of the nine planes in LDS, the row halo in registers, combination + activation "
                     "(+ 1x1 output projection) by four other waves; Z never written to HBM)",
             128134: "gemm_nt16_f32_kernel (backward-data GEMMs of the project-then-upsample pairs, dx = D . wcat^T with both operands K-contiguous: persistent "
                     "workgroups over 128x128 tiles, 128-bit fragments of both operands, fp32 MFMA 16x16x4)",
             7700: "gmm_logprob_kernel<R> + gmm_sum_kernel (mixture log-prob / responsibilities, lane = component, wave-shuffle logsumexp)"}
    LATENCY_BOUND = (7700,)          # not contraction kernels: reported beside the roofline, never as the dominant MFMA kernel

    def __init__(self):
        self.records = {}

    def add(self, kid, s, e, flops, executed=None):
        """`flops` = the reference's operation count of the launch (algorithmic); `executed` = what the kernel issues when that is less
        (the upsample-fused convolutions: 25 of the 36 low-resolution tap products per 2x2 output block)."""
        self.records.setdefault(kid, []).append((s, e, flops, flops if executed is None else executed))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kid, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            fl = sum(f for _, _, f, _ in recs)
            fx = sum(x for _, _, _, x in recs)
            n = len(recs)
            out[kid] = dict(kernel=self.NAMES.get(kid, "igemm tile %d" % kid), launches=n, total_ms=ms, avg_ms=ms / n,
                            flops_per_launch=fl / n, tflops=(fl / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                            executed_flops_per_launch=fx / n, executed_tflops=(fx / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                            bound="latency" if kid in self.LATENCY_BOUND else "mfma")
        return out
