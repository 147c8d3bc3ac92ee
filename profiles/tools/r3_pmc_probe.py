"""One launch each of the round-3 hot kernels for rocprofv3 --pmc passes (counters in separate runs, as the MI355X guide prescribes):
   the 16-wave f16x3 halo convolution at dec.conv7 (batch 128, per-sample record), the one-launch stride-2 backward-data of enc.conv1 and
   the register-resident mixture kernel at R = 8 / K = 50 (L * B * K = 640 000).  usage (from the repository root, one counter set per run):
     rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d out1 -- python3 profiles/tools/r3_pmc_probe.py
   then: python3 profiles/tools/r3_pmc_probe.py --show out1/*/*.db out2/*/*.db ..."""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def show(paths):
    """Per (kernel, grid) means of every counter in the given rocprofv3 databases, plus the derived figures DESIGN 4a quotes.
    Units as the MI355X guide states them: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
    SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over the 1 024 SIMDs, GRBM_GUI_ACTIVE is cycles summed over the 8 XCDs,
    FETCH_SIZE / WRITE_SIZE are KiB."""
    names = {4194304: "conv3x3_halo_split16_kernel<4,false>  dec.conv7 fwd  N=128 128x128x128->128 (4096 blocks x 1024)",
             1048576: "conv3x3_halo_split16_kernel<4,false>  enc.conv1 stride-2 bwd-data, one launch (1024 blocks x 1024)",
             131072: "gmm_logprob_reg_kernel<8>  L*B*K = 100*128*50 (512 blocks x 256)"}
    res, dur, meta = {}, {}, {}
    for f in paths:
        db = sqlite3.connect(f)
        for k, g, c, v, d, vg, lds, scr in db.execute("select kernel_name, grid_size, counter_name, value, duration, vgpr_count, "
                                                      "lds_block_size, scratch_size from counters_collection"):
            if any(t in k for t in ("halo_split16", "gmm_logprob_reg")):
                res.setdefault((g, c), []).append(v)
                dur.setdefault(g, []).append(d)
                meta[g] = (vg, lds, scr)
    for g in sorted(dur, reverse=True):
        m = {c: sum(v) / len(v) for (gg, c), v in res.items() if gg == g}
        us = sum(dur[g]) / len(dur[g]) / 1e3
        print("%s\n  arch VGPRs %d, LDS %d B, scratch %d B; mean duration over the counter passes %.1f us" % ((names.get(g, str(g)),) + meta[g] + (us,)))
        for c in sorted(m):
            print("    %-28s %.5g" % (c, m[c]))
        if "GRBM_GUI_ACTIVE" in m and us > 100:             # (GUI_ACTIVE includes the dispatch ramp: meaningless for a 13 us launch)
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            print("  -> effective clock %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)" % (cyc / us / 1e3))
            if m.get("SQ_INSTS_MFMA"):
                print("  -> MFMA pipe busy %.3f of the elapsed SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x %.4g cycles)); "
                      "%.0f busy cycles per MFMA" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"]))
                print("  -> executed 16-bit MFMA rate %.0f TFLOP/s (SQ_INSTS_MFMA x 32768 flop / duration); algorithmic fp32 rate = 1/3 of it"
                      % (m["SQ_INSTS_MFMA"] * 32768 / us / 1e6))
        if "SQ_WAVE_CYCLES" in m:
            print("  -> per wave-cycle: issuing %.3f, waiting on anything %.3f, waiting to issue %.3f; VALU/LDS/MFMA instructions per wave = "
                  "%.0f / %.0f / %.0f" % (m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                                          m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"] / (g / 64), m["SQ_INSTS_LDS"] / (g / 64),
                                          m.get("SQ_INSTS_MFMA", 0) / (g / 64)))
        if "FETCH_SIZE" in m:
            print("  -> HBM-side traffic per launch: fetch %.1f MiB (2 x FETCH_SIZE, the guide's gfx950 correction for 16-byte coalesced reads), "
                  "write %.1f MiB (%.0f GB/s over the launch)"
                  % (2 * m["FETCH_SIZE"] / 1024, m["WRITE_SIZE"] / 1024, (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / us / 1e3))
        print()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--show":
        return show(sys.argv[2:])
    import numpy as np
    import torch
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    L.load()
    st = torch.cuda.current_stream().cuda_stream
    N, H, W, Ci, Co = 128, 128, 128, 128, 128
    x = torch.randn(N, H, W, Ci, device="cuda")
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
    y = torch.empty(N, H, W, Co, device="cuda")
    pk = torch.empty(L.query("ladder_filter_pack_split_bytes", 9, Ci, Co, 4), dtype=torch.uint8, device="cuda")
    L.call("ladder_filter_pack_split", w.data_ptr(), pk.data_ptr(), 9, Ci, Co, 0, 4, st)
    rec, yrec = torch.empty(512, device="cuda"), torch.empty(512, device="cuda")
    L.call("ladder_absmax_samples", x.data_ptr(), N, H * W * Ci, rec.data_ptr(), st)
    for _ in range(3):
        L.call("ladder_conv3x3_split", x.data_ptr(), rec.data_ptr(), pk.data_ptr(), None, y.data_ptr(), yrec.data_ptr(), N, H, W, Ci, Co, 1, 4, st)
    # enc.conv1 backward-data: dy [128, 32, 32, 128] -> dx [128, 64, 64, 128]
    dy = torch.randn(128, 32, 32, 128, device="cuda")
    pk2 = torch.empty(L.query("ladder_filter_pack_split_bytes", 9, 128, 512, 4), dtype=torch.uint8, device="cuda")
    L.call("ladder_filter_pack_split", w.data_ptr(), pk2.data_ptr(), 9, 128, 512, 2, 4, st)
    drec = torch.empty(512, device="cuda")
    L.call("ladder_absmax_samples", dy.data_ptr(), 128, 32 * 32 * 128, drec.data_ptr(), st)
    dx = torch.empty(128, 64, 64, 128, device="cuda")
    for _ in range(3):
        L.call("ladder_conv3x3_s2_bwd_data_split", dy.data_ptr(), drec.data_ptr(), pk2.data_ptr(), dx.data_ptr(), yrec.data_ptr(), 128, 64, 64, 128, 32, 32, 128, 4, st)
    # mixture kernel, configs[4] shape
    Lmc, B, R, K = 100, 128, 8, 50
    rng = np.random.default_rng(3)
    A = rng.normal(0, 0.3, (K, R, R))
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    wg, m, c = f(rng.dirichlet(np.ones(K))), f(rng.normal(0, 1.5, (K, R))), f(A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))
    packed = torch.empty(K * L.query("ladder_gmm_packed_stride", R), device="cuda")
    L.call("ladder_gmm_prepare", wg.data_ptr(), m.data_ptr(), c.data_ptr(), K, R, packed.data_ptr(), st)
    mu, sd, eps = torch.randn(B, R, device="cuda"), torch.rand(B, R, device="cuda") + 0.1, torch.randn(Lmc, B, R, device="cuda")
    out, dmu, dsd = torch.empty(1, device="cuda"), torch.empty(B, R, device="cuda"), torch.empty(B, R, device="cuda")
    ws = torch.empty(L.query("ladder_gmm_workspace_bytes", Lmc, B), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        L.call("ladder_gmm_logprob_fwd_bwd", mu.data_ptr(), sd.data_ptr(), eps.data_ptr(), packed.data_ptr(), Lmc, B, R, K, out.data_ptr(),
               dmu.data_ptr(), dsd.data_ptr(), ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
