"""Round 5: per-launch times of the 'project, then upsample' passes (csrc/upproj.hip + the dense kernels) at the four decoder pairs of
BASELINE configs[2] (batch 128), beside the algorithmic HBM bytes of the elementwise passes."""
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
for (name, H, Cin, Cout) in (("conv2d_7", 64, 128, 128), ("conv2d_6", 32, 256, 128), ("conv2d_5", 16, 256, 256), ("conv2d_4", 8, 512, 256)):
    M, N9 = B * H * H, 9 * Cout
    x = torch.randn(M, Cin, device="cuda"); w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.05
    wcat = torch.empty(Cin, N9, device="cuda"); wcatT = torch.empty(N9, Cin, device="cuda")
    z = torch.empty(M, N9, device="cuda"); y = torch.empty(B, 2 * H, 2 * H, Cout, device="cuda"); bias = torch.randn(Cout, device="cuda")
    fl = 2.0 * M * Cin * N9
    tot = {}
    t = timeit(lambda: (L.call("ladder_filter_pack_split", p(w), p(wcat), 1, Cin, N9, 6, 0, st), L.call("ladder_filter_pack_split", p(w), p(wcatT), 1, N9, Cin, 7, 0, st)))
    print("%s pack (both operands) %.1f us" % (name, t))
    w1 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, Cin, N9))
    t = timeit(lambda: L.call("ladder_dense_fwd", p(x), p(wcat), None, p(z), M, Cin, N9, 0, p(w1), w1.numel(), st)); tot["fwd"] = t
    print("%s proj GEMM        %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    t = timeit(lambda: L.call("ladder_up2proj_fwd_combine", p(z), p(bias), p(y), None, None, None, 0, B, H, H, Cout, 1, st)); tot["fwd"] += t
    by = (z.numel() + y.numel()) * 4
    print("%s fwd combine      %8.1f us %6.2f TB/s (%d MB)" % (name, t, by / t * 1e-6, by >> 20))
    if Cout == 128:
        pw = torch.randn(128, 3, device="cuda"); pb = torch.randn(3, device="cuda"); out = torch.empty(B, 2 * H, 2 * H, 3, device="cuda")
        t = timeit(lambda: L.call("ladder_up2proj_fwd_combine", p(z), p(bias), p(y), p(pw), p(pb), p(out), 3, B, H, H, Cout, 1, st))
        print("%s fwd combine+proj %8.1f us" % (name, t))
        t = timeit(lambda: L.call("ladder_up2proj_fwd_combine", p(z), p(bias), None, p(pw), p(pb), p(out), 3, B, H, H, Cout, 1, st))
        print("%s fwd combine+proj, no y %8.1f us %6.2f TB/s" % (name, t, z.numel() * 4 / t * 1e-6))
    t = timeit(lambda: L.call("ladder_up2proj_bwd_combine", p(y), p(z), B, H, H, Cout, st)); tot["bwd"] = t
    print("%s bwd combine      %8.1f us %6.2f TB/s" % (name, t, by / t * 1e-6))
    dx = torch.empty(M, Cin, device="cuda")
    w2 = ws(L.query("ladder_igemm_fwd_workspace_bytes", M, N9, Cin))
    t = timeit(lambda: L.call("ladder_dense_bwd_data_nt", p(z), p(wcat), p(dx), M, Cin, N9, None, 0, st)); tot["bwd"] += t      # (as the engine issues it: K-contiguous weights, 16x16x4 kernel)
    print("%s bwd GEMM         %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    dw = torch.empty(Cin, N9, device="cuda"); db9 = torch.empty(N9, device="cuda"); dwo = torch.empty(3, 3, Cin, Cout, device="cuda"); db = torch.empty(Cout, device="cuda")
    w3 = ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, Cin, N9))
    t = timeit(lambda: L.call("ladder_dense_bwd_weight", p(x), p(z), p(dw), p(db9), M, Cin, N9, p(w3), w3.numel(), st)); tot["wgrad"] = t
    print("%s wgrad GEMM       %8.1f us %6.1f TF" % (name, t, fl / t * 1e-6))
    t = timeit(lambda: L.call("ladder_up2proj_wgrad_unpack", p(dw), p(db9), p(dwo), p(db), Cin, Cout, st)); tot["wgrad"] += t
    print("%s wgrad unpack     %8.1f us" % (name, t))
    print("%s TOTAL fwd %.0f  bwd-data %.0f  wgrad %.0f us" % (name, tot["fwd"], tot["bwd"], tot["wgrad"]))
    del x, w, z, y, dx, dw, wcat, wcatT
    torch.cuda.empty_cache()
