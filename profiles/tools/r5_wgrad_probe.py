"""Round 5: the upsample-fused filter-gradient kernel (wgrad3x3_up2_f32_kernel) and the direct one at the CelebA batch-128 shapes.
   python profiles/tools/r5_wgrad_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ladder_latent_data_distribution_modelling_amd import _lib as L  # noqa: E402

st = torch.cuda.current_stream().cuda_stream
N = 128


def p(t):
    return None if t is None else t.data_ptr()


def ws(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (name, H, Cin, Cout, bias) in (("conv2d_7", 64, 128, 128, True), ("conv2d_6", 32, 256, 128, False), ("conv2d_5", 16, 256, 256, True)):
    W = H
    x = torch.randn(N, H, W, Cin, device="cuda")
    dy = torch.randn(N, 2 * H, 2 * W, Cout, device="cuda")
    dw, db = torch.empty(3, 3, Cin, Cout, device="cuda"), torch.empty(Cout, device="cuda")
    fl = 2.0 * N * 4 * H * W * 9 * Cin * Cout * 25 / 36
    wu = ws(L.query("ladder_conv3x3_up2_wgrad_workspace_bytes", N, H, W, Cin, Cout))
    t = timeit(lambda: L.call("ladder_conv3x3_up2_wgrad", p(x), 0, p(dy), p(dw), p(db) if bias else None, N, H, W, Cin, Cout, p(wu), wu.numel(), st))
    print("%s up2 wgrad %8.1f us  %6.1f TF issued" % (name, t, fl / t * 1e-6))
