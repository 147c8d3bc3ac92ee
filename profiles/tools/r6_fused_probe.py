"""Round 6: the fused forward of a projected pair (ladder_up2proj_fused_fwd: Z in an LDS ring) beside the two-call form it replaces (projection GEMM +
combination), at the four decoder pairs of BASELINE configs[2] (batch 128); with / without the y write and the 1x1 output projection (conv2d_7)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ladder_latent_data_distribution_modelling_amd import _lib as L
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
ws = lambda n: torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")

def timeit(fn, reps=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

B = int(os.environ.get("PROBE_BATCH", "128"))
only = os.environ.get("PROBE_LAYERS", "")
for (name, H, Cin, Cout) in (("conv2d_7", 64, 128, 128), ("conv2d_6", 32, 256, 128), ("conv2d_5", 16, 256, 256), ("conv2d_4", 8, 512, 256)):
    if only and name not in only:
        continue
    M, N9 = B * H * H, 9 * Cout
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    wcat = torch.empty(Cin, N9, device="cuda"); wcatT = torch.empty(N9, Cin, device="cuda")
    L.call("ladder_filter_pack_split", p(w), p(wcat), 1, Cin, N9, 6, 0, st); L.call("ladder_filter_pack_split", p(w), p(wcatT), 1, N9, Cin, 7, 0, st)
    y = torch.empty(B, 2 * H, 2 * H, Cout, device="cuda"); bias = torch.randn(Cout, device="cuda")
    fl = 2.0 * M * Cin * N9
    assert L.query("ladder_up2proj_fused_eligible", B, H, H, Cin, Cout) == 1
    tf = timeit(lambda: L.call("ladder_up2proj_fused_fwd", p(x), p(wcatT), p(bias), p(y), None, None, None, 0, B, H, H, Cin, Cout, 1, None, 0, st))
    z = torch.empty(M, N9, device="cuda"); y2 = torch.empty_like(y)
    w1 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, Cin, N9))
    tg = timeit(lambda: L.call("ladder_dense_fwd", p(x), p(wcat), None, p(z), M, Cin, N9, 0, p(w1), w1.numel(), st))
    tc = timeit(lambda: L.call("ladder_up2proj_fwd_combine", p(z), p(bias), p(y2), None, None, None, 0, B, H, H, Cout, 1, st))
    err = float((y - y2).abs().max() / y2.abs().max())
    print("%s fused %8.1f us = %6.1f TF issued | two-call form: GEMM %8.1f + combine %7.1f = %8.1f us | fused / two-call %.3f | max rel diff %.1e" % (
        name, tf, fl / tf * 1e-6, tg, tc, tg + tc, tf / (tg + tc), err))
    if name == "conv2d_7":
        pw = torch.randn(128, 3, device="cuda"); pb = torch.randn(3, device="cuda"); out = torch.empty(B, 2 * H, 2 * H, 3, device="cuda"); out2 = torch.empty_like(out)
        wp = ws(L.query("ladder_up2proj_fused_workspace_bytes", B, H, H, Cout, 3))
        t1 = timeit(lambda: L.call("ladder_up2proj_fused_fwd", p(x), p(wcatT), p(bias), p(y), p(pw), p(pb), p(out), 3, B, H, H, Cin, Cout, 1, p(wp), wp.numel(), st))
        t2 = timeit(lambda: L.call("ladder_up2proj_fused_fwd", p(x), p(wcatT), p(bias), None, p(pw), p(pb), p(out), 3, B, H, H, Cin, Cout, 1, p(wp), wp.numel(), st))
        c1 = timeit(lambda: L.call("ladder_up2proj_fwd_combine", p(z), p(bias), p(y2), p(pw), p(pb), p(out2), 3, B, H, H, Cout, 1, st))
        c2 = timeit(lambda: L.call("ladder_up2proj_fwd_combine", p(z), p(bias), None, p(pw), p(pb), p(out2), 3, B, H, H, Cout, 1, st))
        print("%s + 1x1 projection: fused %8.1f us (no y: %8.1f) | two-call %8.1f (no y: %8.1f) | projection max rel diff %.1e" % (
            name, t1, t2, tg + c1, tg + c2, float((out - out2).abs().max() / out2.abs().max())))
    del x, w, z, y, y2, wcat, wcatT
    torch.cuda.empty_cache()
