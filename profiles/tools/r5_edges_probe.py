"""Round 5: the helper launches of the upsample-fused layers (edge lines, border lines, line filter gradients) at the four CelebA decoder
shapes, for a rocprofv3 --kernel-trace --stats run:  rocprofv3 --kernel-trace --stats -d out -- python3 profiles/tools/r5_edges_probe.py"""
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


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (name, H, Cin, Cout) in (("conv2d_7", 64, 128, 128), ("conv2d_6", 32, 256, 128), ("conv2d_5", 16, 256, 256), ("conv2d_4", 8, 512, 256)):
    W = H
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(3, 3, Cin, Cout, device="cuda") / (9 * Cin) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty(N, 2 * H, 2 * W, Cout, device="cuda")
    dy = torch.randn(N, 2 * H, 2 * W, Cout, device="cuda")
    dx = torch.zeros(N, H, W, Cin, device="cuda")
    we = ws(L.query("ladder_conv3x3_up2_edges_workspace_bytes", N, H, W, Cin, Cout))
    t_e = timeit(lambda: L.call("ladder_conv3x3_up2_edges", p(x), p(w), p(b), p(y), None, None, None, None, 0, N, H, W, Cin, Cout, 1, 0, p(we), we.numel(), st))
    wb = ws(L.query("ladder_conv3x3_up2_bwd_borders_workspace_bytes", N, H, W, Cout, Cin))
    t_b = timeit(lambda: L.call("ladder_conv3x3_up2_bwd_borders", p(dy), p(w), p(dx), N, H, W, Cout, Cin, p(wb), wb.numel(), st))
    print("%s  edges %.1f us   borders %.1f us" % (name, t_e, t_b))
