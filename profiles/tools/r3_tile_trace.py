"""Per-workgroup timeline of the split halo kernels; needs the library built with profiles/tools/r3_tile_trace.patch applied
(the patch is against csrc/convsplit.hip of commit b1e9961: check that file out, git apply profiles/tools/r3_tile_trace.patch, python -m ladder_latent_data_distribution_modelling_amd.csrc.build; revert afterwards).
usage: [LADDER_DISABLE_HALO16=1] [LADDER_HALO_STAGGER=n] python3 profiles/tools/r3_tile_trace.py"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ladder_latent_data_distribution_modelling_amd import _lib as L
L.load()
st = torch.cuda.current_stream().cuda_stream
for name, (N, H, W, Ci, Co) in dict(conv7=(128, 128, 128, 128, 128), conv6=(128, 64, 64, 256, 128)).items():
    x = torch.randn(N, H, W, Ci, device="cuda")
    w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.03
    y = torch.empty(N, H, W, Co, device="cuda")
    pk = torch.empty(L.query("ladder_filter_pack_split_bytes", 9, Ci, Co, 4), dtype=torch.uint8, device="cuda")
    L.call("ladder_filter_pack_split", w.data_ptr(), pk.data_ptr(), 9, Ci, Co, 0, 4, st)
    rec, yrec = torch.empty(512, device="cuda"), torch.empty(512, device="cuda")
    L.call("ladder_absmax_samples", x.data_ptr(), N, H * W * Ci, rec.data_ptr(), st)
    f = lambda: L.call("ladder_conv3x3_split", x.data_ptr(), rec.data_ptr(), pk.data_ptr(), None, y.data_ptr(), yrec.data_ptr(), N, H, W, Ci, Co, 1, 4, st)
    for _ in range(3): f()
    TH = 8 if os.environ.get('LADDER_DISABLE_HALO16') else 16
    nb = N * (H // TH) * (W // 32) * ((Co + 127) // 128)
    tr = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
    os.environ["LADDER_TRACE_PTR"] = str(tr.data_ptr())
    torch.cuda.synchronize()
    f()
    torch.cuda.synchronize()
    del os.environ["LADDER_TRACE_PTR"]
    t = tr.cpu().numpy()
    t0 = t[:, 0].min()
    T = (t[:, :5] - t0) / 100.0          # us (100 MHz)
    hw, xcc = t[:, 5], t[:, 6] & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    print(name, "blocks", nb, "distinct CUs", len(set(key.tolist())), "kernel span %.1f us" % T[:, 4].max())
    pro, main, epi, drain = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3]
    print("  per block (us): prologue %.1f  main loop %.1f  epilogue-issue %.1f  store drain %.1f  total %.1f" % (pro.mean(), main.mean(), epi.mean(), drain.mean(), (T[:, 4] - T[:, 0]).mean()))
    print("  main loop percentiles", np.percentile(main, [5, 50, 95]).round(1), " epilogue+drain pct", np.percentile(epi + drain, [5, 50, 95]).round(1))
    gaps, ovl = [], []
    both = 0.0; one = 0.0; none = 0.0
    for k in set(key.tolist()):
        idx = np.where(key == k)[0]
        o = idx[np.argsort(T[idx, 0])]
        for a, b in zip(o[:-1], o[1:]):
            gaps.append(T[b, 0] - T[a, 4])
        # time with 0 / 1 / 2 blocks of this CU inside their main loop
        ev = sorted([(T[i, 1], 1) for i in idx] + [(T[i, 2], -1) for i in idx])
        cur, last = 0, T[idx, 0].min()
        for tt, d in ev:
            dt = tt - last
            if cur == 0: none += dt
            elif cur == 1: one += dt
            else: both += dt
            cur += d; last = tt
    tot = none + one + both
    print("  per CU: time with 0 / 1 / >=2 workgroups in the main loop: %.1f%% / %.1f%% / %.1f%%" % (100 * none / tot, 100 * one / tot, 100 * both / tot))
    first = np.argsort(T[:, 0])[:512]
    print("  wave slots (HW_ID & 15) of blockIdx 0..255:", np.bincount((hw[:256] & 15).astype(int), minlength=8)[:8], " of 256..511:", np.bincount((hw[256:512] & 15).astype(int), minlength=8)[:8] if nb >= 512 else "")
    same = sum(1 for i in range(256, min(512, nb)) if key[i] in set(key[:256].tolist()))
    print("  blocks 256..511 whose CU also hosts one of blocks 0..255:", same, " distinct CUs among blocks 0..255:", len(set(key[:256].tolist())))
    gaps = np.array(gaps)
    print("  gap between consecutive blocks on a CU (next start - previous end, us): mean %.2f  pct" % gaps.mean(), np.percentile(gaps, [5, 50, 95]).round(2))
    # phase spread: at mid-kernel, fraction of CUs in main loop
    for frac in (0.25, 0.5, 0.75):
        tm = T[:, 4].max() * frac
        inmain = ((T[:, 1] <= tm) & (T[:, 2] > tm)).sum()
        alive = ((T[:, 0] <= tm) & (T[:, 4] > tm)).sum()
        print("  at %.0f%% of the kernel: %d blocks alive, %d in their main loop" % (100 * frac, alive, inmain))
