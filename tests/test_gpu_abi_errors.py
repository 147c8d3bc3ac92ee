"""Error behaviour of the C ABI (include/ladder_hip.h: "returns 0 or a negative error code, never throws, never allocates"):
degenerate shapes, misaligned pointers and short workspaces are rejected with the documented codes and leave the outputs alone;
the smallest legal problems (one pixel, one channel, one sample, one component) and ragged sizes compute correctly."""
import numpy as np
import pytest
import torch

from oracle import ladder_oracle as O

pytestmark = pytest.mark.gpu
E_SHAPE, E_ALIGN, E_WORKSPACE = -1, -2, -3


def _L():
    from ladder_latent_data_distribution_modelling_amd import _lib
    _lib.load()
    return _lib


def test_rejections_return_codes(gpu_ctx):
    L = _L()
    st = gpu_ctx.stream
    x = torch.zeros(2, 8, 8, 16, device="cuda"); w = torch.zeros(3, 3, 16, 32, device="cuda"); b = torch.zeros(32, device="cuda")
    y = torch.full((2, 8, 8, 32), 7.0, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    q = L.query
    conv = lambda N, xp=x.data_ptr(): q("ladder_conv2d_fwd", xp, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, 8, 8, 16, 8, 8, 32, 3, 3, 1, 1, 1, 0,
                                        ws.data_ptr(), ws.numel(), st)
    assert conv(0) == E_SHAPE and conv(-3) == E_SHAPE                       # empty / negative batch
    assert conv(2, x.data_ptr() + 4) == E_ALIGN                             # 4-byte-offset input: float4 loads need 16-byte bases
    assert torch.all(y == 7.0)                                              # a rejected call wrote nothing
    assert conv(2) == 0
    dw, db, dy = torch.empty_like(w), torch.empty_like(b), torch.zeros_like(y)
    assert q("ladder_conv2d_bwd_filter", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), 2, 8, 8, 16, 8, 8, 32, 3, 3, 1, 1, 1,
             None, 0, st) == E_WORKSPACE                                    # split partials need a workspace
    assert q("ladder_dense_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, 16, 32, 0, None, 0, st) == E_SHAPE
    assert q("ladder_resize_bilinear_fwd", x.data_ptr(), y.data_ptr(), 2, 8, 8, 16, 12, 16, st) == E_SHAPE       # non-integer factor
    assert q("ladder_depth_to_space", x.data_ptr(), y.data_ptr(), 2, 8, 8, 16, 3, 0, st) == E_SHAPE              # 16 % 9 != 0
    assert q("ladder_gmm_prepare", x.data_ptr(), x.data_ptr(), x.data_ptr(), 5, 9, y.data_ptr(), st) == E_SHAPE  # R > 8: dense path only
    assert q("ladder_gmm_prepare_dense", x.data_ptr(), x.data_ptr(), x.data_ptr(), 5, 10, y.data_ptr(), st) == E_SHAPE   # R % 4 != 0
    assert q("ladder_gmm_logprob_fwd_bwd", x.data_ptr(), x.data_ptr(), x.data_ptr(), y.data_ptr(), 4, 2, 2, 3, y.data_ptr(), y.data_ptr(),
             y.data_ptr(), ws.data_ptr(), 8, st) == E_WORKSPACE
    st8 = torch.zeros(64, dtype=torch.float64, device="cuda")
    assert q("ladder_vbgmm_fit", x.data_ptr(), 5, 10, 2, None, st8.data_ptr(), 0, 0.1, 1.0, 1e-6, 1e-3, 10, y.data_ptr(), y.data_ptr(),
             y.data_ptr(), ws.data_ptr(), ws.numel(), st) == E_SHAPE        # fewer samples than components
    assert q("ladder_diag_mixture_fwd_bwd", *([x.data_ptr()] * 5), 3, 2, 65, 4, *([y.data_ptr()] * 5), ws.data_ptr(), ws.numel(), st) == E_SHAPE
    assert q("ladder_elbo_finalize", x.data_ptr(), x.data_ptr(), None, L.LadderElboCfg(0, 1, 1, 1, 1, 0, 0, 0, 0, 0.0, 0.0, 0, 0), y.data_ptr(),
             st) == E_SHAPE
    assert q("ladder_adam_clip", x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, 0.1, 0.9, 0.95, 1e-8, 1.0, st) == 0   # n = 0: no-op
    # round 5, projected resize -> conv pairs (csrc/upproj.hip, csrc/densef32.hip)
    z = torch.zeros(2 * 8 * 8, 9 * 32, device="cuda"); y2 = torch.full((2, 16, 16, 32), 7.0, device="cuda"); po = torch.full((2, 16, 16, 3), 7.0, device="cuda")
    assert q("ladder_up2proj_fwd_combine", z.data_ptr(), None, None, None, None, None, 0, 2, 8, 8, 32, 0, st) == E_SHAPE          # nothing to write
    assert q("ladder_up2proj_fwd_combine", z.data_ptr(), None, y2.data_ptr(), w.data_ptr(), None, po.data_ptr(), 3, 2, 8, 8, 32, 0, st) == E_SHAPE   # fused projection needs C == 128
    assert q("ladder_up2proj_fwd_combine", z.data_ptr() + 4, None, y2.data_ptr(), None, None, None, 0, 2, 8, 8, 32, 0, st) == E_ALIGN
    assert q("ladder_up2proj_fwd_combine", z.data_ptr(), None, y2.data_ptr(), None, None, None, 0, 2, 8, 8, 30, 0, st) == E_SHAPE   # C % 4 != 0
    assert q("ladder_upfproj_fwd_combine", z.data_ptr(), None, y2.data_ptr(), 3, 2, 8, 8, 32, 0, st) == E_SHAPE                     # factor 3: only 2 and 4
    assert q("ladder_upfproj_bwd_combine", y2.data_ptr(), z.data_ptr(), 8, 2, 8, 8, 32, st) == E_SHAPE
    assert q("ladder_up2proj_wgrad_unpack", None, None, w.data_ptr(), None, 16, 32, st) == E_SHAPE
    assert torch.all(y2 == 7.0) and torch.all(po == 7.0)
    assert q("ladder_upfproj_eligible", 2, 2, 8, 8, 24, 32) == 0 and q("ladder_upfproj_eligible", 4, 2, 8, 8, 32, 32) == 1      # Cin % 16
    assert q("ladder_dense_fwd_is_persistent", 4096, 128, 1152) == 0 and q("ladder_dense_fwd_is_persistent", 8192, 128, 1152) == 1
    assert q("ladder_dense_fwd_is_persistent", 8192, 128, 1100) == 0 and q("ladder_dense_bwd_weight_is_persistent", 8192, 96, 1152) == 0
    assert q("ladder_dense_fwd_nt", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 128, 16, 32, 0, st) == E_SHAPE               # below the persistent kernel's sizes
    assert q("ladder_dense_fwd_nt", x.data_ptr(), w.data_ptr(), None, None, 8192, 128, 1152, 0, st) == E_SHAPE                      # no output
    assert q("ladder_dense_fwd", x.data_ptr(), w.data_ptr(), None, None, 8192, 128, 1152, 0, None, 0, st) == E_SHAPE
    assert q("ladder_filter_pack_split", w.data_ptr(), z.data_ptr(), 9, 16, 9 * 32, 6, 0, st) == E_SHAPE                            # orientation 6 is ONE [Cin][9 C] matrix: ntaps = 1
    assert q("ladder_filter_pack_split", w.data_ptr(), z.data_ptr(), 1, 16, 9 * 32 + 16, 6, 0, st) == E_SHAPE                       # ... with 9 | Cout


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,pad", [(1, 1, 1, 1, 1, 1, 1, "same"), (1, 3, 3, 1, 1, 3, 1, "valid"), (1, 2, 5, 3, 2, 3, 2, "same"),
                                                    (7, 1, 9, 5, 6, 3, 1, "same")])
def test_smallest_and_ragged_convs(gpu_ctx, N, H, W, Cin, Cout, k, s, pad):
    """One pixel / one channel / odd channel counts (no float4 path) / stride 2 on a 2-row map: forward, backward-data and filter
    gradient vs float64 autograd."""
    from ladder_latent_data_distribution_modelling_amd import arch
    L = _L()
    st = gpu_ctx.stream
    rng = np.random.default_rng(N * 10 + W)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = rng.standard_normal((k, k, Cin, Cout)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xt, wt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, w, b))
    yt = O.conv2d_tf(xt, wt, bt, s, pad)
    dy = rng.standard_normal(tuple(yt.shape)).astype(np.float32)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    pt, Ho = arch.conv_out(H, k, s, pad)
    pl, Wo = arch.conv_out(W, k, s, pad)
    d = lambda a: torch.as_tensor(a).cuda()
    xd, wd, bd, dyd = d(x), d(w), d(b), d(dy)
    y = torch.empty(N, Ho, Wo, Cout, device="cuda")
    ws = torch.empty(1 << 22, dtype=torch.uint8, device="cuda")
    L.call("ladder_conv2d_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0,
           ws.data_ptr(), ws.numel(), st)
    assert np.abs(y.cpu().numpy() - yt.detach().numpy()).max() < 1e-5 * max(1.0, float(yt.detach().abs().max()))
    dw, db, dx, wT = torch.empty_like(wd), torch.empty_like(bd), torch.empty_like(xd), torch.empty_like(wd)
    L.call("ladder_conv2d_bwd_filter", xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl,
           ws.data_ptr(), ws.numel(), st)
    L.call("ladder_filter_flip_transpose", wd.data_ptr(), wT.data_ptr(), k, k, Cin, Cout, st)
    L.call("ladder_conv2d_bwd_data", dyd.data_ptr(), wT.data_ptr(), dx.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, None, 0,
           ws.data_ptr(), ws.numel(), st)
    for got, ref in ((dw, wt.grad), (db, bt.grad), (dx, xt.grad)):
        assert np.abs(got.cpu().numpy() - ref.numpy()).max() < 1e-5 * max(1.0, float(ref.abs().max()))


def test_single_sample_single_component_mixture(gpu_ctx):
    L = _L()
    st = gpu_ctx.stream
    rng = np.random.default_rng(0)
    gm = dict(weights=np.ones(1, np.float32), means=rng.standard_normal((1, 3)).astype(np.float32), covs=(np.eye(3) * 0.7).astype(np.float32)[None])
    mu, sd, eps = (rng.standard_normal((1, 3)).astype(np.float32), np.full((1, 3), 0.5, np.float32), rng.standard_normal((1, 1, 3)).astype(np.float32))
    t = torch.tensor(mu + sd * eps[0], dtype=torch.float64)
    ref = O.gmm_log_prob(t.unsqueeze(0), *(torch.tensor(gm[k], dtype=torch.float64) for k in ("weights", "means", "covs"))).sum().item()
    d = lambda a: torch.as_tensor(a).cuda()
    w_, m_, c_, mud, sdd, epsd = d(gm["weights"]), d(gm["means"]), d(gm["covs"]), d(mu), d(sd), d(eps)
    packed = torch.empty(L.query("ladder_gmm_packed_stride", 3), device="cuda")
    L.call("ladder_gmm_prepare", w_.data_ptr(), m_.data_ptr(), c_.data_ptr(), 1, 3, packed.data_ptr(), st)
    out, dmu, dsd = torch.empty(1, device="cuda"), torch.empty(1, 3, device="cuda"), torch.empty(1, 3, device="cuda")
    ws = torch.empty(max(L.query("ladder_gmm_workspace_bytes", 1, 1), 16), dtype=torch.uint8, device="cuda")
    L.call("ladder_gmm_logprob_fwd_bwd", mud.data_ptr(), sdd.data_ptr(), epsd.data_ptr(), packed.data_ptr(), 1, 1, 3, 1, out.data_ptr(),
           dmu.data_ptr(), dsd.data_ptr(), ws.data_ptr(), ws.numel(), st)
    assert abs(out.item() - ref) < 1e-5 * abs(ref) + 1e-5
