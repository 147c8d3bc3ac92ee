"""(round 5: + the small-map halo kernel conv3x3_halo_f32s_kernel at its four CelebA uses: conv2d_5 fused forward / backward-data, conv2d_4 fused
forward, conv2d_3 plain.)
Three launches each of the strict-fp32 hot kernels at their CelebA batch-128 geometry, for rocprofv3 --pmc passes (counters in separate
runs with --kernel-trace only, as the MI355X guide prescribes):
   conv3x3_halo_f32_kernel<true,1>   upsample-fused conv2d_7 forward + fused RGB projection (64x64 -> 128x128, 128 -> 128 channels)
   conv3x3_halo_f32_kernel<false,1>  upsample-fused conv2d_6 forward (32x32 -> 64x64, 256 -> 128)
   conv3x3_halo_f32_kernel<false,2>  backward-data of the conv2d_7 pair (5x5 / stride-2 correlation over dy)
   conv3x3_halo_f32_kernel<false,0>  plain 3x3 convolution conv2d_5 (32x32, 256 -> 256)
   wgrad3x3_up2_f32_kernel           filter gradient of the conv2d_7 pair over the low-resolution map
   wgrad3x3_halo_kernel<1,32,1>      direct filter gradient of conv2d_5
   igemm_fwd_kernel<128,128>         gather convolution dec.conv2d_4 (16x16, 512 -> 256)
usage (from the repository root; one counter set per run -- profiles/make_profiles.sh does all of it):
   rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d out1 -- python3 profiles/tools/r4_pmc_probe.py
   python3 profiles/tools/r4_pmc_probe.py --show out1/*/*.db out2/*/*.db ..."""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

KERNELS = ("conv3x3_halo_f32_kernel", "conv3x3_halo_f32s_kernel", "wgrad3x3_up2_f32_kernel", "wgrad3x3_halo_kernel", "igemm_fwd_kernel<128, 128")
# executed MFMA FLOPs per launch are counted (SQ_INSTS_MFMA x 4096: v_mfma_f32_32x32x2_f32 = 32 x 32 x 2 x 2 flop)
PEAK_TF = 157.3


def show(paths):
    """Per kernel instantiation: mean of every counter, and the derived figures DESIGN 4b quotes.  Units as the MI355X guide states them:
    SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over the
    1 024 SIMDs, GRBM_GUI_ACTIVE is cycles summed over the 8 XCDs."""
    res, dur, meta = {}, {}, {}
    for f in paths:
        db = sqlite3.connect(f)
        for k, g, c, v, d, vg, lds, scr in db.execute("select kernel_name, grid_size, counter_name, value, duration, vgpr_count, "
                                                      "lds_block_size, scratch_size from counters_collection"):
            if any(t in k for t in KERNELS):
                key = (k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], g)
                res.setdefault((key, c), []).append(v)
                dur.setdefault(key, []).append(d)
                meta[key] = (vg, lds, scr)
    for key in sorted(dur, key=lambda k: -sum(dur[k]) / len(dur[k])):
        m = {c: sum(v) / len(v) for (kk, c), v in res.items() if kk == key}
        us = sum(dur[key]) / len(dur[key]) / 1e3
        print("%s  grid %d threads\n  arch VGPRs %d, LDS %d B, scratch %d B; mean duration over the counter passes %.1f us"
              % ((key[0], key[1]) + meta[key] + (us,)))
        for c in sorted(m):
            print("    %-28s %.5g" % (c, m[c]))
        if "GRBM_GUI_ACTIVE" in m and us > 100:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            print("  -> effective clock %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)" % (cyc / us / 1e3))
            if m.get("SQ_INSTS_MFMA"):
                tf = m["SQ_INSTS_MFMA"] * (2048 if ("nt16" in key[0] or "gemm_tn" in key[0] or "MI16x16" in key[0]) else 4096) / us / 1e6
                print("  -> MFMA pipe busy %.3f of the elapsed SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x %.4g cycles)); %.0f busy cycles per MFMA"
                      % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"]))
                print("  -> executed fp32 MFMA rate %.1f TFLOP/s (SQ_INSTS_MFMA x 4096 flop -- 2048 for the 16x16x4 kernels -- / duration) = %.3f of %.1f" % (tf, tf / PEAK_TF, PEAK_TF))
        if "SQ_WAVE_CYCLES" in m and "SQ_INSTS_VALU" in m:
            waves = key[1] / 64
            print("  -> per wave-cycle: issuing %.3f, waiting on anything %.3f, waiting to issue %.3f; VALU / LDS / MFMA instructions per wave = "
                  "%.0f / %.0f / %.0f" % (m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                                          m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"] / waves, m["SQ_INSTS_LDS"] / waves,
                                          m.get("SQ_INSTS_MFMA", 0) / waves))
        print()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--show":
        return show(sys.argv[2:])
    import torch
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    L.load()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()

    def bank(w, cin, cout, flip):
        b = torch.empty(9, cin, cout, device="cuda")
        L.call("ladder_filter_pack_split", p(w), p(b), 9, cin, cout, flip, 0, st)
        return b

    N = 128
    # conv2d_7 pair: forward + projection, backward-data, filter gradient
    H = W = 64
    Cin = Cout = 128
    xlo = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    bias = torch.randn(Cout, device="cuda")
    y = torch.empty(N, 2 * H, 2 * W, Cout, device="cuda")
    out = torch.empty(N, 2 * H, 2 * W, 3, device="cuda")
    pw, pb = torch.randn(Cout, 3, device="cuda"), torch.randn(3, device="cuda")
    b3, b4 = bank(w, Cin, 4 * Cout, 3), bank(w, 4 * Cout, Cin, 4)
    dy = torch.randn(N, 2 * H, 2 * W, Cout, device="cuda")
    dx = torch.empty(N, H, W, Cin, device="cuda")
    dw, db = torch.empty(3, 3, Cin, Cout, device="cuda"), torch.empty(Cout, device="cuda")
    ws = torch.empty(L.query("ladder_conv3x3_up2_wgrad_workspace_bytes", N, H, W, Cin, Cout), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        L.call("ladder_conv3x3_up2_split_proj", p(xlo), None, p(b3), p(bias), p(y), p(pw), p(pb), p(out), 3, N, H, W, Cin, Cout, 1, 0, 0, st)
        L.call("ladder_conv3x3_up2_bwd_data_split", p(dy), None, p(b4), p(dx), None, N, H, W, Cout, Cin, 0, st)
        L.call("ladder_conv3x3_up2_wgrad", p(xlo), 0, p(dy), p(dw), p(db), N, H, W, Cin, Cout, p(ws), ws.numel(), st)
    torch.cuda.synchronize()
    del xlo, y, out, dy, dx, ws
    # conv2d_6 forward
    H = W = 32
    Cin, Cout = 256, 128
    xlo = torch.randn(N, H, W, Cin, device="cuda")
    w6 = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    y = torch.empty(N, 2 * H, 2 * W, Cout, device="cuda")
    b36 = bank(w6, Cin, 4 * Cout, 3)
    for _ in range(3):
        L.call("ladder_conv3x3_up2_split", p(xlo), None, p(b36), p(bias), p(y), None, N, H, W, Cin, Cout, 1, 0, 0, st)
    torch.cuda.synchronize()
    del xlo, y
    # conv2d_5: plain halo convolution and its direct filter gradient
    Cin = Cout = 256
    x5 = torch.randn(N, H, W, Cin, device="cuda")
    w5 = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    b5 = torch.randn(Cout, device="cuda")
    y5 = torch.empty(N, H, W, Cout, device="cuda")
    dw5, db5 = torch.empty_like(w5), torch.empty(Cout, device="cuda")
    ws5 = torch.empty(L.query("ladder_conv2d_bwd_filter_workspace_bytes", N, H, W, Cin, H, W, Cout, 3, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        L.call("ladder_conv3x3_split", p(x5), None, p(w5), p(b5), p(y5), None, N, H, W, Cin, Cout, 1, 0, st)
        L.call("ladder_conv2d_bwd_filter", p(x5), p(y5), p(dw5), p(db5), N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, p(ws5), ws5.numel(), st)
    torch.cuda.synchronize()
    del x5, y5
    # round 5: the small-map halo kernel -- conv2d_5 fused forward (16x16 -> 32x32, 256 -> 256) and backward-data, conv2d_4 fused forward (8x8 -> 16x16,
    # 512 -> 256), conv2d_3 plain (8x8, 512 -> 512)
    for (Hs, Ci, Co) in ((16, 256, 256), (8, 512, 256)):
        xs = torch.randn(N, Hs, Hs, Ci, device="cuda")
        wsm = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
        bs = torch.randn(Co, device="cuda")
        ys = torch.empty(N, 2 * Hs, 2 * Hs, Co, device="cuda")
        b3s = bank(wsm, Ci, 4 * Co, 3)
        for _ in range(3):
            L.call("ladder_conv3x3_up2_split", p(xs), None, p(b3s), p(bs), p(ys), None, N, Hs, Hs, Ci, Co, 1, 0, 0, st)
        if Hs == 16:
            b4s = bank(wsm, 4 * Co, Ci, 4)
            dxs = torch.empty(N, Hs, Hs, Ci, device="cuda")
            for _ in range(3):
                L.call("ladder_conv3x3_up2_bwd_data_split", p(ys), None, p(b4s), p(dxs), None, N, Hs, Hs, Co, Ci, 0, st)
        torch.cuda.synchronize()
        del xs, ys
    x3 = torch.randn(N, 8, 8, 512, device="cuda")
    w3 = torch.randn(3, 3, 512, 512, device="cuda") * 0.05
    y3 = torch.empty(N, 8, 8, 512, device="cuda")
    for _ in range(3):
        L.call("ladder_conv3x3_split", p(x3), None, p(w3), None, p(y3), None, N, 8, 8, 512, 512, 1, 0, st)
    torch.cuda.synchronize()
    # dec.conv2d_4 on the gather kernel (two split-K passes + second pass: the plan of the iteration)
    H = W = 16
    Cin, Cout = 512, 256
    x4 = torch.randn(N, H, W, Cin, device="cuda")
    w4 = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    b4_ = torch.randn(Cout, device="cuda")
    y4 = torch.empty(N, H, W, Cout, device="cuda")
    ws4 = torch.empty(max(L.query("ladder_igemm_fwd_workspace_bytes", N * H * W, 9 * Cin, Cout), 16), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        L.call("ladder_conv2d_fwd", p(x4), p(w4), p(b4_), p(y4), N, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, 0, p(ws4), ws4.numel(), st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
