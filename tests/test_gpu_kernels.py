"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle (float64).

Tolerances: fp32 kernels vs float64 oracle -- relative 2e-5 of the output scale for contractions
(K up to ~4.6k fp32 FMAs), 1e-5 for element-wise / normalisation passes.
"""
import numpy as np
import pytest
import torch

from oracle import ladder_oracle as O

pytestmark = pytest.mark.gpu


def _lib():
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    return L


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()


def p(t):
    return None if t is None else t.data_ptr()


def close(got, ref, rtol, what=""):
    got = got.detach().cpu().numpy().astype(np.float64) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().numpy().astype(np.float64) if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert np.isfinite(got).all() and err < rtol, "%s: rel err %.3e (tol %.1e)" % (what, err, rtol)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, padding, act
    (2, 16, 16, 16, 32, 3, 1, "same", "leaky_relu"),
    (3, 16, 16, 32, 160, 3, 1, "same", None),          # N tile edge (160 = 128 + 32)
    (2, 32, 32, 3, 128, 3, 2, "same", None),           # Cin=3 scalar gather, TF asymmetric SAME pad
    (5, 32, 32, 1, 16, 3, 2, "same", "leaky_relu"),    # MNIST-digit encoder conv (Cin=1)
    (3, 32, 32, 1, 64, 3, 2, "same", "leaky_relu"),    # MNIST-fashion encoder conv
    (24, 128, 128, 128, 3, 1, 1, "same", None),        # CelebA output conv at full resolution: bwd_data = conv_smallcin_kernel
                                                       # (1x1 from 3 channels), more pixels than one grid pass, gated variant too
    (2, 9, 11, 24, 3, 1, 1, "same", None),             # same shape class, 24 output channels of bwd_data: generic gather kernel
    (2, 16, 16, 128, 128, 3, 2, "same", None),
    (4, 4, 4, 64, 64, 3, 1, "valid", "leaky_relu"),
    (5, 1, 1, 48, 48, 1, 1, "same", None),
    (2, 32, 32, 4, 1, 5, 1, "valid", "relu"),          # MNIST output conv
    (2, 8, 8, 128, 3, 1, 1, "same", None),             # CelebA output conv (Cout=3)
    (1, 7, 9, 20, 36, 3, 1, "same", "relu"),           # ragged everything
    (2, 9, 9, 16, 16, 3, 2, "same", None),             # odd size stride 2 (pad 1/1)
    (2, 10, 14, 16, 32, 5, 2, "same", None),           # 5x5 stride 2: parity-class bwd_data with 9/6/6/4 taps
    (3, 7, 9, 32, 16, 3, 2, "valid", "leaky_relu"),    # stride 2 VALID, odd sizes
    (2, 8, 8, 16, 48, 1, 2, "same", None),             # 1x1 stride 2: three parity classes have no tap (zeros)
    (64, 2, 2, 64, 64, 3, 1, "same", None),            # tiny-spatial decoder block
    (4, 2, 2, 16, 64, 1, 1, "same", "leaky_relu"),     # fashion decoder 1x1 (M=16)
    (4, 4, 4, 16, 64, 3, 1, "same", "leaky_relu"),
    (2, 1, 1, 32, 32, 1, 1, "same", None),             # M=2
    (4, 16, 16, 16, 64, 3, 1, "same", "leaky_relu"),
    (4, 32, 32, 16, 1, 5, 1, "valid", "relu"),         # MNIST-digit output conv: conv_cout1_kernel, 4 rows per wavefront
    (9, 32, 32, 64, 1, 5, 1, "valid", "relu"),         # MNIST-fashion output conv: conv_cout1_kernel (lane = input channel)
    (300, 32, 32, 32, 1, 5, 1, "valid", None),         # more rows than one grid pass (grid-stride), 2 rows per wavefront
    (3, 12, 9, 64, 1, 5, 1, "valid", None),            # non-square map, Wo = 5 (one window group)
    (4, 8, 8, 16, 64, 3, 1, "same", "leaky_relu"),
    (4, 4, 4, 32, 32, 3, 1, "valid", "leaky_relu"),    # fashion encoder valid conv
    (32, 2, 2, 256, 256, 3, 1, "same", None),          # split-K regime (few tiles, long K)
    (16, 4, 4, 512, 128, 3, 1, "valid", None),         # split-K regime, encoder conv6 style
    (16, 64, 64, 32, 256, 3, 1, "same", "leaky_relu"), # LDS-halo kernel (>=512 workgroups), fwd + bwd_data (Cin'=256)
    (64, 32, 32, 16, 160, 3, 1, "same", None),         # LDS-halo kernel, Cout tile edge (160 = 128 + 32)
    (32, 64, 64, 64, 128, 3, 1, "same", None),         # halo filter-gradient kernel (>= 4096 patches, Cin % 64 == 0)
    (8, 128, 128, 64, 160, 3, 1, "same", "leaky_relu"),  # halo filter-gradient kernel: Cout edge (second 128-block has 32 channels)
    (32, 64, 64, 128, 64, 3, 1, "same", None),         # halo filter-gradient kernel: 2 ci slabs, Cout = 64 (half a channel pair idle)
    (32, 64, 64, 32, 128, 3, 1, "same", None),         # Cin = 32: halo forward / backward-data, generic filter gradient
    (32, 16, 16, 512, 160, 3, 1, "same", None),        # halo filter gradient on 2 x 16-pixel patches (16-wide maps), Cout edge
    (64, 8, 8, 512, 512, 3, 1, "same", "leaky_relu"),  # halo filter gradient on 4 x 8-pixel patches (8-wide maps)
    (64, 64, 64, 128, 128, 3, 2, "same", None),        # stride-2 halo filter gradient, 1 x 32-pixel patches of the 32 x 32 output
    (64, 32, 32, 256, 160, 3, 2, "same", "leaky_relu"),  # stride-2 halo filter gradient, 2 x 16 patches, Cout edge
    (128, 16, 16, 256, 512, 3, 2, "same", None),       # stride-2 halo filter gradient, 4 x 8 patches
    (128, 8, 8, 256, 512, 3, 2, "same", "leaky_relu"),  # stride-2 backward-data: 128-tile parity classes split over K (strided second pass)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_fwd_bwd(gpu_ctx, case):
    L = _lib()
    from ladder_latent_data_distribution_modelling_amd import arch
    N, H, W, Cin, Cout, k, s, pad, act = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    pre = O.conv2d_tf(xt, wt, bt, s, pad)
    yr = O.act(pre, act)
    dy = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    yr.backward(torch.tensor(dy, dtype=torch.float64))

    pt, Ho = arch.conv_out(H, k, s, pad)
    pl, Wo = arch.conv_out(W, k, s, pad)
    assert (Ho, Wo) == tuple(yr.shape[1:3])
    st = gpu_ctx.stream
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty(N, Ho, Wo, Cout, device="cuda")
    M, Kd = N * Ho * Wo, k * k * Cin
    fws, fwn = gpu_ctx.ws(max(L.query("ladder_igemm_fwd_workspace_bytes", M, Kd, Cout), 16))
    L.call("ladder_conv2d_fwd", p(xd), p(wd), p(bd), p(y), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, L.ACT[act], fws, fwn, st)
    close(y, yr, 2e-5, "fwd (split-K allowed)")
    L.call("ladder_conv2d_fwd", p(xd), p(wd), p(bd), p(y), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, L.ACT[act], None, 0, st)
    close(y, yr, 2e-5, "fwd")
    dyd = dev(dy)
    if act is not None:
        # the activation mask is taken from the ORACLE's output: with millions of outputs a few pre-activations sit within
        # fp32 rounding of 0 and their leaky/relu mask legitimately differs between fp32 and float64 (each flip moves dw by O(1))
        yref = dev(yr.detach().numpy())
        L.call("ladder_act_bwd", p(dyd), p(yref), p(dyd), dyd.numel(), L.ACT[act], st)
    nb = L.query("ladder_conv2d_bwd_filter_workspace_bytes", N, H, W, Cin, Ho, Wo, Cout, k, k)
    wsp, wsn = gpu_ctx.ws(nb)
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    L.call("ladder_conv2d_bwd_filter", p(xd), p(dyd), p(dw), p(db), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, wsp, wsn, st)
    close(dw, wt.grad, 3e-5, "dw")
    close(db, bt.grad, 3e-5, "db")
    wT = torch.empty(wd.numel(), device="cuda")
    L.call("ladder_filter_flip_transpose", p(wd), p(wT), k, k, Cin, Cout, st)
    dx = torch.empty_like(xd)
    dws = torch.empty(max(L.query("ladder_igemm_fwd_workspace_bytes", N * H * W, k * k * Cout, Cin), 16), dtype=torch.uint8, device="cuda")
    L.call("ladder_conv2d_bwd_data", p(dyd), p(wT), p(dx), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, None, 0, p(dws), dws.numel(), st)
    close(dx, xt.grad, 3e-5, "dx")
    # gated epilogue: dx * leaky'(gate) with the conv input as gate (what the engine fuses for the producer layer)
    dxg = torch.empty_like(xd)
    L.call("ladder_conv2d_bwd_data", p(dyd), p(wT), p(dxg), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, p(xd), 1, p(dws), dws.numel(), st)
    assert torch.equal(dxg, dx * torch.where(xd > 0, 1.0, 0.2))
    # db == NULL is allowed (conv feeding a norm layer): dw must be unaffected
    dw2 = torch.empty_like(wd)
    L.call("ladder_conv2d_bwd_filter", p(xd), p(dyd), p(dw2), None, N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, wsp, wsn, st)
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize("M,K,N,act", [(128, 2048, 64, None), (128, 4096, 64, "relu"), (128, 512, 512, "leaky_relu"), (7, 2, 32, "relu"),
                                       (256, 64, 4096, "leaky_relu"), (100, 33, 2, None), (4, 16, 1024, "tanh")])
def test_dense_fwd_bwd(gpu_ctx, M, K, N, act):
    L = _lib()
    rng = np.random.default_rng(M * 7 + K * 3 + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    xt, wt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, w, b))
    yr = O.act(O.dense(xt, wt, bt), act)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    yr.backward(torch.tensor(dy, dtype=torch.float64))
    st = gpu_ctx.stream
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    y = torch.empty(M, N, device="cuda")
    fws = torch.empty(max(L.query("ladder_igemm_fwd_workspace_bytes", M, K, N), 16), dtype=torch.uint8, device="cuda")
    L.call("ladder_dense_fwd", p(xd), p(wd), p(bd), p(y), M, K, N, L.ACT[act], p(fws), fws.numel(), st)
    close(y, yr, 2e-5, "fwd")
    if act is not None:
        L.call("ladder_act_bwd", p(dyd), p(y), p(dyd), dyd.numel(), L.ACT[act], st)
    wsp, wsn = gpu_ctx.ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, K, N))
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    L.call("ladder_dense_bwd_weight", p(xd), p(dyd), p(dw), p(db), M, K, N, wsp, wsn, st)
    close(dw, wt.grad, 3e-5, "dw")
    close(db, bt.grad, 3e-5, "db")
    wT = torch.empty(wd.numel(), device="cuda")
    L.call("ladder_filter_flip_transpose", p(wd), p(wT), 1, 1, K, N, st)
    dx = torch.empty_like(xd)
    bws = torch.empty(max(L.query("ladder_igemm_fwd_workspace_bytes", M, N, K), 16), dtype=torch.uint8, device="cuda")
    L.call("ladder_dense_bwd_data", p(dyd), p(wT), p(dx), M, K, N, None, 0, p(bws), bws.numel(), st)
    dxg = torch.empty_like(xd)
    L.call("ladder_dense_bwd_data", p(dyd), p(wT), p(dxg), M, K, N, p(xd), 2, p(bws), bws.numel(), st)      # relu gate
    assert torch.equal(dxg, dx * (xd > 0))
    close(dx, xt.grad, 3e-5, "dx")


@pytest.mark.parametrize("shape,act", [((4, 8, 8, 128), "leaky_relu"), ((3, 5, 7, 20), None), ((2, 2, 2, 512), "leaky_relu"),
                                       ((16, 32, 32, 64), "leaky_relu")])
def test_batch_norm(gpu_ctx, shape, act):
    L = _lib()
    rng = np.random.default_rng(sum(shape))
    C = shape[-1]
    rows = int(np.prod(shape[:-1]))
    x = (rng.standard_normal(shape) * 1.7 + 0.6).astype(np.float32)
    g = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    be = (0.2 * rng.standard_normal(C)).astype(np.float32)
    xt, gt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, g, be))
    yr = O.act(O.batch_norm_train(xt, gt, bt), act)
    dy = rng.standard_normal(shape).astype(np.float32)
    yr.backward(torch.tensor(dy, dtype=torch.float64))
    st = gpu_ctx.stream
    xd, gd, bd, dyd = dev(x), dev(g), dev(be), dev(dy)
    wsp, wsn = gpu_ctx.ws(L.query("ladder_bn_workspace_bytes", rows, C))
    sums, mr = torch.empty(4 * C, device="cuda"), torch.empty(2 * C, device="cuda")     # (the record: 2C doubles)
    y = torch.empty_like(xd)
    L.call("ladder_bn_fwd_stats", p(xd), p(sums), rows, C, wsp, wsn, st)
    L.call("ladder_bn_fwd_apply", p(xd), p(sums), float(rows), p(gd), p(bd), p(y), p(mr), rows, C, 1e-3, L.ACT[act], st)
    close(y, yr, 1e-5, "y")
    ds = torch.empty(2 * C, device="cuda")
    dx, dg, db = torch.empty_like(xd), torch.empty_like(gd), torch.empty_like(bd)
    L.call("ladder_bn_bwd_stats", p(dyd), p(xd), p(mr), p(gd), p(bd), p(ds), rows, C, L.ACT[act], wsp, wsn, st)
    L.call("ladder_bn_bwd_apply", p(dyd), p(xd), p(mr), p(gd), p(bd), p(ds), float(rows), p(dx), p(dg), p(db), rows, C, L.ACT[act], st)
    close(dx, xt.grad, 2e-5, "dx")
    close(dg, gt.grad, 2e-5, "dgamma")
    close(db, bt.grad, 2e-5, "dbeta")


@pytest.mark.parametrize("shape,minmax", [((128, 4, 4, 64), False), ((16, 32, 32, 128), True), ((8, 8, 8, 100), False), ((64, 64, 64, 32), True)])
def test_batch_norm_far_off_centre_channels(gpu_ctx, shape, minmax):
    """per-channel means of +-50 standard deviations.  TF's fused batch norm (reference codes/models.py:398-460) is two-pass; the
    single-pass E[x^2] - mean^2 from fp32-stored sums of rounds 1-4 lost eps_fp32 x (1 + mean^2 / var) = 1.5e-4 of the variance here.  With the
    fp64 statistics record the normalised output matches the float64 oracle to 1e-5 of its scale, the variance to 1e-6 -- for the separate pass
    (with and without the extremes) on big and small maps."""
    L = _lib()
    rng = np.random.default_rng(sum(shape))
    C = shape[-1]
    rows = int(np.prod(shape[:-1]))
    sd = (0.5 + rng.random(C)).astype(np.float32)
    mean = (50.0 * sd * np.where(rng.random(C) < 0.5, -1.0, 1.0)).astype(np.float32)
    x = (rng.standard_normal(shape) * sd + mean).astype(np.float32)
    g = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    be = (0.2 * rng.standard_normal(C)).astype(np.float32)
    yr = O.batch_norm_train(torch.tensor(x, dtype=torch.float64), torch.tensor(g, dtype=torch.float64), torch.tensor(be, dtype=torch.float64))
    st = gpu_ctx.stream
    xd, gd, bd = dev(x), dev(g), dev(be)
    wsp, wsn = gpu_ctx.ws(2 * L.query("ladder_bn_workspace_bytes", rows, C))
    sums, mr = torch.empty(6 * C if minmax else 4 * C, device="cuda"), torch.empty(2 * C, device="cuda")
    y = torch.empty_like(xd)
    L.call("ladder_bn_fwd_stats_minmax" if minmax else "ladder_bn_fwd_stats", p(xd), p(sums), rows, C, wsp, wsn, st)
    L.call("ladder_bn_fwd_apply", p(xd), p(sums), float(rows), p(gd), p(bd), p(y), p(mr), rows, C, 1e-3, 0, st)
    close(y, yr, 1e-5, "y")
    x64 = x.astype(np.float64).reshape(-1, C)
    var = x64.var(0)
    got_var = 1.0 / mr[C:].double().cpu().numpy() ** 2 - 1e-3
    assert np.abs(got_var - var).max() / var.max() < 1e-6, np.abs(got_var - var).max() / var.max()
    if minmax:
        assert np.array_equal(sums[4 * C:5 * C].cpu().numpy(), x.reshape(-1, C).min(0)) and np.array_equal(sums[5 * C:].cpu().numpy(), x.reshape(-1, C).max(0))


def test_batch_norm_far_off_centre_two_ranks():
    """... and across two data-parallel ranks (C2 all-reduces the fp64 record): the engine's BatchNormAct on two virtual ranks (engine.VirtualComm)
    against the float64 oracle's batch norm over the concatenated batch."""
    from ladder_latent_data_distribution_modelling_amd.engine import BatchNormAct, Ctx, run_virtual_ranks
    rng = np.random.default_rng(77)
    N, H, W, C = 16, 16, 16, 64
    sd = (0.5 + rng.random(C)).astype(np.float32)
    mean = (50.0 * sd * np.where(rng.random(C) < 0.5, -1.0, 1.0)).astype(np.float32)
    x = (rng.standard_normal((N, H, W, C)) * sd + mean).astype(np.float32)
    x[: N // 2] += (3.0 * sd).astype(np.float32)                    # the two shards have different means: the combination matters
    g = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    be = (0.2 * rng.standard_normal(C)).astype(np.float32)
    yr = O.act(O.batch_norm_train(torch.tensor(x, dtype=torch.float64), torch.tensor(g, dtype=torch.float64), torch.tensor(be, dtype=torch.float64)), "leaky_relu")

    class _PS:
        w = {"bn/gamma": torch.as_tensor(g).cuda(), "bn/beta": torch.as_tensor(be).cuda()}

    def job(rank, comm):
        ctx = Ctx("cuda:0", comm)
        bn = BatchNormAct(ctx, _PS, "bn", C, "leaky_relu")
        return bn.forward(torch.as_tensor(x[rank * (N // 2):(rank + 1) * (N // 2)]).cuda()).cpu().numpy()

    out = np.concatenate(run_virtual_ranks(2, job))
    assert np.abs(out - yr.numpy()).max() / np.abs(yr.numpy()).max() < 1e-5


@pytest.mark.parametrize("shape", [(3, 2, 2, 64), (2, 16, 16, 32), (2, 8, 8, 100), (4, 64, 64, 128)])
def test_instance_norm_style(gpu_ctx, shape):
    L = _lib()
    rng = np.random.default_rng(sum(shape))
    N, H, W, C = shape
    x = (rng.standard_normal(shape) * 2 + 0.5).astype(np.float32)
    sty = (0.5 * rng.standard_normal((N, 2 * C))).astype(np.float32)
    xt, stt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, sty))
    yr = O.leaky_relu(O.style_mod(O.instance_norm(xt), stt))
    dy = rng.standard_normal(shape).astype(np.float32)
    yr.backward(torch.tensor(dy, dtype=torch.float64))
    st = gpu_ctx.stream
    xd, sd, dyd = dev(x), dev(sty), dev(dy)
    y, mr = torch.empty_like(xd), torch.empty(N, 2 * C, device="cuda")
    wsp, wsn = gpu_ctx.ws(max(L.query("ladder_in_style_workspace_bytes", N, H * W, C), 16))
    for use_ws in (False, True):       # one-workgroup-per-slab kernels, then the split / vectorised ones
        y.zero_()
        L.call("ladder_in_style_fwd", p(xd), p(sd), p(y), p(mr), N, H * W, C, 1e-6, 1, wsp if use_ws else None, wsn if use_ws else 0, st)
        close(y, yr, 1e-5, "y ws=%s" % use_ws)
    dx, dst = torch.empty_like(xd), torch.empty_like(sd)
    L.call("ladder_in_style_bwd", p(dyd), p(xd), p(sd), p(mr), p(dx), p(dst), N, H * W, C, 1, None, 0, st)
    close(dst, stt.grad, 2e-5, "dstyle (no ws)")
    L.call("ladder_in_style_bwd", p(dyd), p(xd), p(sd), p(mr), p(dx), p(dst), N, H * W, C, 1, wsp, wsn, st)
    close(dst, stt.grad, 2e-5, "dstyle")
    # instance-norm backward at 2x2 with eps=1e-6 is ill-conditioned (rstd up to 1e3): compare at 1e-3 of scale there
    close(dx, xt.grad, 1e-3 if H * W <= 4 else 5e-5, "dx")


@pytest.mark.parametrize("shape,out", [((2, 1, 1, 8), 2), ((2, 2, 2, 16), 8), ((1, 8, 8, 4), 16), ((2, 5, 5, 3), 10), ((2, 4, 4, 4), 4),
                                       ((3, 16, 16, 32), 32), ((2, 3, 3, 8), 12)])
def test_resize_legacy_bilinear(gpu_ctx, shape, out):
    L = _lib()
    rng = np.random.default_rng(out)
    N, H, W, C = shape
    x = rng.standard_normal(shape).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr = O.resize_bilinear_legacy(xt, out, out)
    dy = rng.standard_normal((N, out, out, C)).astype(np.float32)
    (yr * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    st = gpu_ctx.stream
    xd, dyd = dev(x), dev(dy)
    y, dx = torch.empty(N, out, out, C, device="cuda"), torch.empty_like(xd)
    L.call("ladder_resize_bilinear_fwd", p(xd), p(y), N, H, W, C, out, out, st)
    L.call("ladder_resize_bilinear_bwd", p(dyd), p(dx), N, H, W, C, out, out, st)
    close(y, yr, 1e-6, "y")
    close(dx, xt.grad, 1e-6, "dx")
    if out == 2 * H:                       # the gated form: dx * leaky'(gate) in the same pass, bit-identical to the two separate kernels
        gate = dev(rng.standard_normal(shape).astype(np.float32))
        dxg = torch.empty_like(xd)
        L.call("ladder_resize_bilinear_bwd_gated", p(dyd), p(dxg), N, H, W, C, out, out, p(gate), 1, st)
        assert torch.equal(dxg, dx * torch.where(gate > 0, 1.0, 0.2))
    else:
        assert L.query("ladder_resize_bilinear_bwd_gated", p(dyd), p(dx), N, H, W, C, out, out, p(xd), 1, st) != 0


def test_depth_to_space_and_pad(gpu_ctx):
    L = _lib()
    rng = np.random.default_rng(5)
    st = gpu_ctx.stream
    for (N, H, W, C, r) in [(2, 1, 1, 64, 4), (3, 4, 4, 16, 2), (1, 3, 5, 36, 3)]:
        x = rng.standard_normal((N, H, W, C)).astype(np.float32)
        yr = O.depth_to_space(torch.tensor(x), r)
        xd = dev(x)
        y = torch.empty(N, H * r, W * r, C // (r * r), device="cuda")
        L.call("ladder_depth_to_space", p(xd), p(y), N, H, W, C, r, 0, st)
        assert np.array_equal(y.cpu().numpy(), yr.numpy())
        back = torch.empty_like(xd)
        L.call("ladder_depth_to_space", p(y), p(back), N, H, W, C, r, 1, st)
        assert np.array_equal(back.cpu().numpy(), x)
    x = rng.standard_normal((2, 28, 28, 1)).astype(np.float32)
    y = torch.empty(2, 32, 32, 1, device="cuda")
    xd = dev(x)
    L.call("ladder_pad_symmetric", p(xd), p(y), 2, 28, 28, 1, 2, st)
    assert np.array_equal(y.cpu().numpy(), np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="symmetric"))


@pytest.mark.parametrize("K,R,Lmc,B", [(30, 2, 100, 16), (50, 8, 20, 8), (5, 1, 7, 3), (70, 3, 5, 4), (27, 2, 50, 128)])
def test_gmm_logprob(gpu_ctx, golden_dir, K, R, Lmc, B):
    """Mixture log-prob + gradient vs the oracle (which is itself pinned to scipy on the reference's fixture)."""
    import os
    L = _lib()
    rng = np.random.default_rng(K + R)
    fix = np.load(os.path.join(golden_dir, "GM_prior_info.npz"))
    cfg = dict(n_mixtures=K, representation_size=R)
    gm = {k: v.astype(np.float32) for k, v in O.synthetic_gm(cfg, rng, fix if K <= 50 else None).items()}
    if K == 27:
        gm = dict(weights=fix["w_active"].astype(np.float32), means=fix["m_active"].astype(np.float32), covs=fix["K_active"].astype(np.float32))
    mu = rng.standard_normal((B, R)).astype(np.float32) * 2
    sd = (0.05 + rng.random((B, R))).astype(np.float32)
    eps = rng.standard_normal((Lmc, B, R)).astype(np.float32)
    mut, sdt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (mu, sd))
    t = mut.unsqueeze(0) + sdt.unsqueeze(0) * torch.tensor(eps, dtype=torch.float64)
    lp = O.gmm_log_prob(t, *(torch.tensor(gm[k], dtype=torch.float64) for k in ("weights", "means", "covs")))
    lp.sum().backward()
    st = gpu_ctx.stream
    packed = torch.empty(K * L.query("ladder_gmm_packed_stride", R), device="cuda")
    wd, md, cd = dev(gm["weights"]), dev(gm["means"]), dev(gm["covs"])      # keep alive until the kernels ran
    L.call("ladder_gmm_prepare", p(wd), p(md), p(cd), K, R, p(packed), st)
    out, dmu, dsd = torch.empty(1, device="cuda"), torch.empty(B, R, device="cuda"), torch.empty(B, R, device="cuda")
    wsp, wsn = gpu_ctx.ws(L.query("ladder_gmm_workspace_bytes", Lmc, B))
    mud, sdd, epsd = dev(mu), dev(sd), dev(eps)
    L.call("ladder_gmm_logprob_fwd_bwd", p(mud), p(sdd), p(epsd), p(packed), Lmc, B, R, K, p(out), p(dmu), p(dsd), wsp, wsn, st)
    assert abs(out.item() - lp.sum().item()) < 2e-5 * abs(lp.sum().item()) + 1e-3
    close(dmu, mut.grad, 5e-5, "dmu")
    close(dsd, sdt.grad, 5e-5, "dsd")


@pytest.mark.parametrize("K,R,Lmc,B", [(30, 64, 100, 128), (20, 16, 12, 10), (7, 12, 3, 5), (30, 64, 5, 2)])
def test_gmm_dense_logprob(gpu_ctx, K, R, Lmc, B):
    """prior "GMM": the mixture on a wide latent (R = code_size 16 / 64) through the dense MFMA path, vs the oracle; the first
    case is the full BASELINE 'optional extra' shape (R = Z = 64, K = 30, L = 100, B = 128: 12 800 samples x 1 920 whitened
    coordinates).  Forward-only mode (dmu = dsd = NULL) gives the same sum."""
    L = _lib()
    rng = np.random.default_rng(K * 7 + R)
    gm = {k: v.astype(np.float32) for k, v in O.synthetic_gm(dict(n_mixtures=K, representation_size=R), rng).items()}
    mu = (rng.standard_normal((B, R)) * 1.5).astype(np.float32)
    sd = (0.05 + rng.random((B, R))).astype(np.float32)
    eps = rng.standard_normal((Lmc, B, R)).astype(np.float32)
    mut, sdt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (mu, sd))
    t = mut.unsqueeze(0) + sdt.unsqueeze(0) * torch.tensor(eps, dtype=torch.float64)
    lp = O.gmm_log_prob(t, *(torch.tensor(gm[k], dtype=torch.float64) for k in ("weights", "means", "covs")))
    lp.sum().backward()
    st = gpu_ctx.stream
    params = torch.empty(L.query("ladder_gmm_dense_param_floats", K, R), device="cuda")
    wd, md, cd = dev(gm["weights"]), dev(gm["means"]), dev(gm["covs"])
    L.call("ladder_gmm_prepare_dense", p(wd), p(md), p(cd), K, R, p(params), st)
    out, out2, dmu, dsd = torch.empty(1, device="cuda"), torch.empty(1, device="cuda"), torch.empty(B, R, device="cuda"), torch.empty(B, R, device="cuda")
    ws = torch.empty(L.query("ladder_gmm_dense_workspace_bytes", Lmc, B, R, K), dtype=torch.uint8, device="cuda")
    mud, sdd, epsd = dev(mu), dev(sd), dev(eps)
    L.call("ladder_gmm_dense_logprob_fwd_bwd", p(mud), p(sdd), p(epsd), p(params), Lmc, B, R, K, p(out), p(dmu), p(dsd), p(ws), ws.numel(), st)
    L.call("ladder_gmm_dense_logprob_fwd_bwd", p(mud), p(sdd), p(epsd), p(params), Lmc, B, R, K, p(out2), None, None, p(ws), ws.numel(), st)
    ref = lp.sum().item()
    assert abs(out.item() - ref) < 3e-5 * abs(ref) + 1e-3 and out2.item() == out.item()
    close(dmu, mut.grad, 1e-4, "dmu")
    close(dsd, sdt.grad, 1e-4, "dsd")
    assert L.query("ladder_gmm_prepare_dense", p(wd), p(md), p(cd), K, 72, p(params), st) != 0       # R > 64: LADDER_E_SHAPE


@pytest.mark.parametrize("K,Z,Lmc,B", [(30, 64, 100, 16), (10, 8, 7, 5), (3, 16, 2, 3), (70, 5, 9, 4)])
def test_diag_mixture_vamp(gpu_ctx, K, Z, Lmc, B):
    """VampPrior term: log of an equally weighted diagonal mixture over L MC samples, with the gradients to the posterior heads
    AND to the K components (the pseudo-input path), vs float64 autograd of the formula in codes/base.py:244-254, 361-370."""
    L = _lib()
    rng = np.random.default_rng(K + Z)
    mu = (rng.standard_normal((B, Z)) * 1.2).astype(np.float32)
    sd = (0.1 + rng.random((B, Z))).astype(np.float32)
    eps = rng.standard_normal((Lmc, B, Z)).astype(np.float32)
    cm = rng.standard_normal((K, Z)).astype(np.float32)
    cs = (0.3 + rng.random((K, Z))).astype(np.float32)
    mut, sdt, cmt, cst = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (mu, sd, cm, cs))
    z = (mut.unsqueeze(0) + sdt.unsqueeze(0) * torch.tensor(eps, dtype=torch.float64)).unsqueeze(2)
    u = (z - cmt) / cst
    lp = -0.5 * (u ** 2).sum(-1) - torch.log(cst).sum(-1) - 0.5 * Z * np.log(2 * np.pi) - np.log(K)
    tot = torch.logsumexp(lp, dim=-1).sum()
    tot.backward()
    st = gpu_ctx.stream
    out = torch.empty(1, device="cuda")
    dmu, dsd, dcm, dcs = (torch.empty(B, Z, device="cuda"), torch.empty(B, Z, device="cuda"), torch.empty(K, Z, device="cuda"),
                          torch.empty(K, Z, device="cuda"))
    ws = torch.empty(L.query("ladder_diag_mixture_workspace_bytes", B, Z, K), dtype=torch.uint8, device="cuda")
    a = [dev(v) for v in (mu, sd, eps, cm, cs)]
    L.call("ladder_diag_mixture_fwd_bwd", *(p(t) for t in a), Lmc, B, Z, K, p(out), p(dmu), p(dsd), p(dcm), p(dcs), p(ws), ws.numel(), st)
    assert abs(out.item() - tot.item()) < 3e-5 * abs(tot.item()) + 1e-3
    close(dmu, mut.grad, 1e-4, "dmu")
    close(dsd, sdt.grad, 1e-4, "dsd")
    close(dcm, cmt.grad, 1e-4, "dcomp_mean")
    close(dcs, cst.grad, 1e-4, "dcomp_sd")


def test_pad_symmetric_bwd(gpu_ctx):
    L = _lib()
    rng = np.random.default_rng(2)
    for (N, H, W, C, pp) in ((3, 28, 28, 1, 2), (2, 5, 7, 3, 3), (1, 4, 4, 2, 4)):
        x = torch.tensor(rng.standard_normal((N, H, W, C)), dtype=torch.float64, requires_grad=True)
        y = O.pad_symmetric(x, pp)
        dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(torch.tensor(dy, dtype=torch.float64))
        dyd, dx = dev(dy), torch.empty(N, H, W, C, device="cuda")
        L.call("ladder_pad_symmetric_bwd", p(dyd), p(dx), N, H, W, C, pp, gpu_ctx.stream)
        close(dx, x.grad, 1e-6, "pad_sym_bwd")


@pytest.mark.parametrize("N,H,Cin,Cout,gate", [(8, 128, 128, 3, "leaky_relu"), (5, 128, 64, 1, None), (70, 32, 16, 4, "relu")])
def test_conv1x1_smallcout_fused_backward(gpu_ctx, N, H, Cin, Cout, gate):
    """The CelebA output conv (1x1, 128 -> 3) at full resolution: dedicated forward kernel (via ladder_conv2d_fwd) and the fused
    backward -- dx with the producer's activation-derivative gate, dW and db from one pass over x -- vs float64 autograd."""
    L = _lib()
    rng = np.random.default_rng(N + Cin)
    M = N * H * H
    assert L.query("ladder_conv1x1_smallcout_eligible", M, Cin, Cout) == 1
    pre = rng.standard_normal((N, H, H, Cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, H, H, Cout)).astype(np.float32)
    pt = torch.tensor(pre, dtype=torch.float64, requires_grad=True)
    wt, bt = torch.tensor(w, dtype=torch.float64, requires_grad=True), torch.tensor(b, dtype=torch.float64, requires_grad=True)
    xt = O.act(pt, gate)                                   # x = the producing layer's OUTPUT
    yt = O.conv2d_tf(xt, wt, bt, 1, "same")
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    st = gpu_ctx.stream
    xd, wd, bd, dyd = dev(xt.detach().numpy().astype(np.float32)), dev(w), dev(b), dev(dy)
    y = torch.empty(N, H, H, Cout, device="cuda")
    L.call("ladder_conv2d_fwd", p(xd), p(wd), p(bd), p(y), N, H, H, Cin, H, H, Cout, 1, 1, 1, 0, 0, 0, None, 0, st)
    close(y, yt, 2e-5, "fwd")
    dx, dw, db = torch.empty_like(xd), torch.empty_like(wd), torch.empty_like(bd)
    ws = torch.empty(L.query("ladder_conv1x1_smallcout_bwd_workspace_bytes", M, Cin, Cout), dtype=torch.uint8, device="cuda")
    L.call("ladder_conv1x1_smallcout_bwd", p(xd), p(dyd), p(wd), p(dx), p(dw), p(db), M, Cin, Cout, L.ACT[gate], p(ws), ws.numel(), st)
    close(dx, pt.grad, 3e-5, "dx (gated)")
    close(dw, wt.grad, 3e-5, "dw")
    close(db, bt.grad, 3e-5, "db")
    assert L.query("ladder_conv1x1_smallcout_eligible", 1000, Cin, Cout) == 0 and L.query("ladder_conv1x1_smallcout_eligible", M, Cin, 5) == 0


def test_adam_clip_matches_tf_form(gpu_ctx):
    L = _lib()
    rng = np.random.default_rng(0)
    n = 1000
    th = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    thd, md, vd = dev(th), dev(m), dev(v)
    th64, m64, v64 = th.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for t in range(1, 4):
        g = (rng.standard_normal(n) * 2).astype(np.float32)     # many |g|>1 -> exercises the clip
        O.adam_tf(th64, g.astype(np.float64), m64, v64, t, 3e-4)
        lr_t = 3e-4 * np.sqrt(1 - 0.95 ** t) / (1 - 0.9 ** t)
        gd = dev(g)
        L.call("ladder_adam_clip", p(thd), p(gd), p(md), p(vd), n, float(lr_t), 0.9, 0.95, 1e-8, 1.0, gpu_ctx.stream)
    close(thd, th64, 1e-6, "theta")
    close(md, m64, 1e-6, "m")
    close(vd, v64, 1e-6, "v")


def test_randn_moments_and_determinism(gpu_ctx):
    L = _lib()
    a, b, c = (torch.empty(1 << 20, device="cuda") for _ in range(3))
    L.call("ladder_randn", p(a), a.numel(), 42, 0, gpu_ctx.stream)
    L.call("ladder_randn", p(b), b.numel(), 42, 0, gpu_ctx.stream)
    L.call("ladder_randn", p(c), c.numel(), 42, 1, gpu_ctx.stream)
    assert torch.equal(a, b) and not torch.equal(a, c)
    x = a.double()
    assert abs(x.mean().item()) < 5e-3 and abs(x.var().item() - 1) < 1e-2
    assert abs((x ** 3).mean().item()) < 2e-2 and abs((x ** 4).mean().item() - 3) < 5e-2
    assert abs((a * c).double().mean().item()) < 5e-3


def test_pixel_partials_full_size(gpu_ctx):
    """BASELINE-size pixel reduction (128 x 128x128x3): size-independent identities."""
    L = _lib()
    n = 128 * 49152
    x = torch.rand(n, device="cuda")
    xh = torch.rand(n, device="cuda")
    out = torch.empty(2, device="cuda")
    wsp, wsn = gpu_ctx.ws(L.query("ladder_pixel_partials_workspace_bytes", n))
    L.call("ladder_pixel_partials", p(x), p(xh), n, p(out), wsp, wsn, gpu_ctx.stream)
    d = (x.double() - xh.double())
    assert abs(out[0].item() - d.abs().sum().item()) < 1e-6 * d.abs().sum().item()
    assert abs(out[1].item() - (d * d).sum().item()) < 1e-6 * (d * d).sum().item()
    L.call("ladder_pixel_partials", p(x), p(x), n, p(out), wsp, wsn, gpu_ctx.stream)
    assert out[0].item() == 0.0 and out[1].item() == 0.0
