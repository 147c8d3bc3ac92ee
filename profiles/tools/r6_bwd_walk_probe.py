"""Round 6: the backward combination of the pairs -- neighbourhood kernel against the row walk (segment sweep), and for the last pair
ladder_conv1x1_smallcout_bwd + ladder_up2proj_bwd_combine against ladder_up2proj_bwd_combine_proj (dy never materialised).  Batch 128."""
import os
import sys

import torch

os.environ["LADDER_UP2BWD_WALK"] = "0"          # ladder_up2proj_bwd_combine = the neighbourhood kernel here; the walk through its own entry point

sys.path.insert(0, ".")
from ladder_latent_data_distribution_modelling_amd import _lib as L  # noqa: E402

L.load()
p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


N = 128
for name, H, W, C in (("conv2d_7", 64, 64, 128), ("conv2d_6", 32, 32, 128), ("conv2d_5", 16, 16, 256), ("conv2d_4", 8, 8, 256)):
    dy = torch.randn(N, 2 * H, 2 * W, C, device="cuda")
    d0 = torch.empty(N * H * W, 9 * C, device="cuda")
    d1 = torch.empty_like(d0)
    t0 = timed(lambda: L.call("ladder_up2proj_bwd_combine", p(dy), p(d0), N, H, W, C, st))
    line = "%s  %dx%d C %d: neighbourhood %7.1f us | walk" % (name, H, W, C, t0)
    for seg in (0, 4, 8, 16, 32, 64):
        if seg > H:
            continue
        t1 = timed(lambda: L.call("ladder_up2proj_bwd_combine_walk", p(dy), p(d1), N, H, W, C, seg, st))
        line += "  rows %d: %7.1f" % (seg, t1)
    err = float((d1 - d0).abs().max() / d0.abs().max())
    print(line + "  | max diff / scale %.1e" % err, flush=True)
    del dy, d0, d1

# the last pair under the 1x1 projection
H = W = 64
C, PCO = 128, 3
y = torch.randn(N, 2 * H, 2 * W, C, device="cuda")
g = torch.randn(N, 2 * H, 2 * W, PCO, device="cuda")
pw = torch.randn(C, PCO, device="cuda") / C ** 0.5
M = N * 4 * H * W
dx = torch.empty_like(y)
dw0, db0 = torch.empty(C, PCO, device="cuda"), torch.empty(PCO, device="cuda")
dw1, db1 = torch.empty(C, PCO, device="cuda"), torch.empty(PCO, device="cuda")
d0 = torch.empty(N * H * W, 9 * C, device="cuda")
d1 = torch.empty_like(d0)
ws0 = torch.empty(max(16, L.query("ladder_conv1x1_smallcout_bwd_workspace_bytes", M, C, PCO)), dtype=torch.uint8, device="cuda")
ws1 = torch.empty(max(16, L.query("ladder_up2proj_bwd_combine_proj_workspace_bytes", N, H, W, C, PCO)), dtype=torch.uint8, device="cuda")


def two_calls():
    L.call("ladder_conv1x1_smallcout_bwd_absmax", p(y), p(g), p(pw), p(dx), p(dw0), p(db0), M, C, PCO, 1, p(ws0), ws0.numel(), None, 4 * H * W, st)
    L.call("ladder_up2proj_bwd_combine", p(dx), p(d0), N, H, W, C, st)


def one_call():
    L.call("ladder_up2proj_bwd_combine_proj", p(y), p(g), p(pw), p(d1), p(dw1), p(db1), PCO, N, H, W, C, 1, p(ws1), ws1.numel(), st)


ta = timed(lambda: L.call("ladder_conv1x1_smallcout_bwd_absmax", p(y), p(g), p(pw), p(dx), p(dw0), p(db0), M, C, PCO, 1, p(ws0), ws0.numel(), None, 4 * H * W, st))
tb = timed(lambda: L.call("ladder_up2proj_bwd_combine", p(dx), p(d0), N, H, W, C, st))
t2 = timed(two_calls)
t1 = timed(one_call)
print("conv2d_8 backward + conv2d_7 combination: 1x1 backward %7.1f us + combination %7.1f us = %7.1f us (back to back %7.1f) | one launch %7.1f us" % (ta, tb, ta + tb, t2, t1))
print("  D max diff / scale %.1e   dpw %.1e   dpb %.1e" % (float((d1 - d0).abs().max() / d0.abs().max()), float((dw1 - dw0).abs().max() / dw0.abs().max()),
                                                       float((db1 - db0).abs().max() / db0.abs().max())))
