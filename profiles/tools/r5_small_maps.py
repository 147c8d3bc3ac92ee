"""Round 5: times the small-map fp32 halo kernels (csrc/convf32s.hip) at the CelebA batch-128 shapes against the kernels they replace.
   python profiles/tools/r5_small_maps.py  [> gpurun_out/r5_small_maps.txt]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ladder_latent_data_distribution_modelling_amd import _lib as L  # noqa: E402

F32 = 0
st = torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def bank(w, cin, cout, flip):
    b = torch.empty(9, cin, cout, device="cuda")
    L.call("ladder_filter_pack_split", p(w), p(b), 9, cin, cout, flip, F32, st)
    return b


def timeit(fn, reps=30):
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


def ws(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")


def row(name, us, flops):
    print("%-58s %8.1f us  %6.1f TF" % (name, us, flops / us * 1e-6))


N = 128
for (lname, H, Cin, Cout) in (("dec.conv2d_5 (16x16 -> 32x32, 256 -> 256)", 16, 256, 256), ("dec.conv2d_4 (8x8 -> 16x16, 512 -> 256)", 8, 512, 256)):
    W = H
    print("==", lname)
    x = torch.randn(N, H, W, Cin, device="cuda")
    up = torch.empty(N, 2 * H, 2 * W, Cin, device="cuda")
    L.call("ladder_resize_bilinear_fwd", p(x), p(up), N, H, W, Cin, 2 * H, 2 * W, st)
    w = torch.randn(3, 3, Cin, Cout, device="cuda") / (9 * Cin) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty(N, 2 * H, 2 * W, Cout, device="cuda")
    dy = torch.randn(N, 2 * H, 2 * W, Cout, device="cuda")
    fl = 2.0 * N * 4 * H * W * 9 * Cin * Cout
    # direct forward on the upsampled tensor: what the engine runs today
    if L.query("ladder_conv3x3_split_eligible", N, 2 * H, 2 * W, Cin, Cout):
        row("direct fwd (8x32 halo kernel)", timeit(lambda: L.call("ladder_conv3x3_split", p(up), None, p(w), p(b), p(y), None, N, 2 * H, 2 * W, Cin, Cout, 1, F32, st)), fl)
    w1 = ws(L.query("ladder_igemm_fwd_workspace_bytes", N * 4 * H * W, 9 * Cin, Cout))
    row("direct fwd (ladder_conv2d_fwd)", timeit(lambda: L.call("ladder_conv2d_fwd", p(up), p(w), p(b), p(y), N, 2 * H, 2 * W, Cin, 2 * H, 2 * W, Cout, 3, 3, 1, 1, 1, 1, p(w1), w1.numel(), st)), fl)
    if L.query("ladder_conv3x3_f32_eligible", N, 2 * H, 2 * W, Cin, Cout) and not L.query("ladder_conv3x3_split_eligible", N, 2 * H, 2 * W, Cin, Cout):
        row("direct fwd (small-map halo kernel)", timeit(lambda: L.call("ladder_conv3x3_split", p(up), None, p(w), p(b), p(y), None, N, 2 * H, 2 * W, Cin, Cout, 1, F32, st)), fl)
    b3 = bank(w, Cin, 4 * Cout, 3)
    assert L.query("ladder_conv3x3_up2_split_eligible", N, H, W, Cin, Cout, F32)
    row("fused fwd (25/36 issued; rate on issued)", timeit(lambda: L.call("ladder_conv3x3_up2_split", p(x), None, p(b3), p(b), p(y), None, N, H, W, Cin, Cout, 1, F32, 0, st)), fl * 25 / 36)
    we = ws(L.query("ladder_conv3x3_up2_edges_workspace_bytes", N, H, W, Cin, Cout))
    row("  + edges", timeit(lambda: L.call("ladder_conv3x3_up2_edges", p(x), p(w), p(b), p(y), None, None, None, None, 0, N, H, W, Cin, Cout, 1, 0, p(we), we.numel(), st)), 1.0)
    # backward-data
    dxu = torch.empty(N, 2 * H, 2 * W, Cin, device="cuda")
    bT = bank(w, Cout, Cin, 1)
    if L.query("ladder_conv3x3_f32_eligible", N, 2 * H, 2 * W, Cout, Cin):
        row("direct bwd-data (halo kernel, up map)", timeit(lambda: L.call("ladder_conv3x3_split", p(dy), None, p(bT), None, p(dxu), None, N, 2 * H, 2 * W, Cout, Cin, 0, F32, st)), fl)
    b4 = bank(w, 4 * Cout, Cin, 4)
    dx = torch.empty(N, H, W, Cin, device="cuda")
    assert L.query("ladder_conv3x3_up2_bwd_data_split_eligible", N, H, W, Cout, Cin, F32)
    row("fused bwd-data (25/36)", timeit(lambda: L.call("ladder_conv3x3_up2_bwd_data_split", p(dy), None, p(b4), p(dx), None, N, H, W, Cout, Cin, F32, st)), fl * 25 / 36)
    wb = ws(L.query("ladder_conv3x3_up2_bwd_borders_workspace_bytes", N, H, W, Cout, Cin))
    row("  + borders", timeit(lambda: L.call("ladder_conv3x3_up2_bwd_borders", p(dy), p(w), p(dx), N, H, W, Cout, Cin, p(wb), wb.numel(), st)), 1.0)
    # filter gradient
    dw, db = torch.empty_like(w), torch.empty(Cout, device="cuda")
    wg = ws(L.query("ladder_conv2d_bwd_filter_workspace_bytes", N, 2 * H, 2 * W, Cin, 2 * H, 2 * W, Cout, 3, 3))
    row("direct filter gradient (up map)", timeit(lambda: L.call("ladder_conv2d_bwd_filter", p(up), p(dy), p(dw), p(db), N, 2 * H, 2 * W, Cin, 2 * H, 2 * W, Cout, 3, 3, 1, 1, 1, p(wg), wg.numel(), st)), fl)
    if L.query("ladder_conv3x3_up2_wgrad_eligible", N, H, W, Cin, Cout):
        wu = ws(L.query("ladder_conv3x3_up2_wgrad_workspace_bytes", N, H, W, Cin, Cout))
        row("fused filter gradient (25/36)", timeit(lambda: L.call("ladder_conv3x3_up2_wgrad", p(x), 0, p(dy), p(dw), p(db), N, H, W, Cin, Cout, p(wu), wu.numel(), st)), fl * 25 / 36)

print("== plain small maps / stride 2")
for (lname, H, Cin, Cout, stride) in (("dec.conv2d_3 8x8 512->512", 8, 512, 512, 1), ("enc.conv2d_1 64->32 128->128 s2", 64, 128, 128, 2),
                                      ("enc.conv2d_2 32->16 128->256 s2", 32, 128, 256, 2), ("enc.conv2d_3 16->8 256->256 s2", 16, 256, 256, 2)):
    W = H
    Ho = H // stride
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(3, 3, Cin, Cout, device="cuda") / (9 * Cin) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty(N, Ho, Ho, Cout, device="cuda")
    fl = 2.0 * N * Ho * Ho * 9 * Cin * Cout
    pad = 1 if stride == 1 else 0
    w1 = ws(L.query("ladder_igemm_fwd_workspace_bytes", N * Ho * Ho, 9 * Cin, Cout))
    row(lname + " gather fwd", timeit(lambda: L.call("ladder_conv2d_fwd", p(x), p(w), p(b), p(y), N, H, W, Cin, Ho, Ho, Cout, 3, 3, stride, pad, pad, 1, p(w1), w1.numel(), st)), fl)
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda")
    dx = torch.empty(N, H, W, Cin, device="cuda")
    wT = torch.empty(9 * Cin * Cout, device="cuda")
    L.call("ladder_filter_flip_transpose", p(w), p(wT), 3, 3, Cin, Cout, st)
    w2 = ws(1 << 26)
    row(lname + " gather bwd-data", timeit(lambda: L.call("ladder_conv2d_bwd_data", p(dy), p(wT), p(dx), N, H, W, Cin, Ho, Ho, Cout, 3, 3, stride, pad, pad, None, 0, p(w2), w2.numel(), st)), fl)
    if stride == 1:
        if L.query("ladder_conv3x3_f32_eligible", N, H, W, Cin, Cout):
            row(lname + " halo fwd", timeit(lambda: L.call("ladder_conv3x3_split", p(x), None, p(w), p(b), p(y), None, N, H, W, Cin, Cout, 1, F32, st)), fl)
            bT = bank(w, Cout, Cin, 1)
            row(lname + " halo bwd-data", timeit(lambda: L.call("ladder_conv3x3_split", p(dy), None, p(bT), None, p(dx), None, N, H, W, Cout, Cin, 0, F32, st)), fl)
    else:
        if L.query("ladder_conv3x3_s2_fwd_f32_eligible", N, H, W, Cin, Ho, Ho, Cout):
            b5 = bank(w, 4 * Cin, Cout, 5)
            row(lname + " halo s2 fwd", timeit(lambda: L.call("ladder_conv3x3_s2_fwd_f32", p(x), p(b5), p(b), p(y), N, H, W, Cin, Ho, Ho, Cout, 1, st)), fl)
        if L.query("ladder_conv3x3_s2_bwd_data_f32_eligible", N, H, W, Cin, Ho, Ho, Cout):
            b2 = bank(w, Cout, 4 * Cin, 2)
            row(lname + " halo s2 bwd-data", timeit(lambda: L.call("ladder_conv3x3_s2_bwd_data_split", p(dy), None, p(b2), p(dx), None, N, H, W, Cin, Ho, Ho, Cout, F32, st)), fl)
