"""'Project, then upsample' (csrc/upproj.hip, round 5, strict fp32): resize x2 (TF1 legacy bilinear) -> 3x3 / SAME convolution -- decoder conv2d_4 ...
conv2d_7, reference codes/models.py:544-578 -- as nine 1x1 convolutions at LOW resolution + an exact elementwise combination, through the C ABI
against the float64 oracle (oracle/ladder_oracle.py: resize_bilinear_legacy + conv2d_tf, and float64 autograd through them):

  * the two operand layouts (orientations 6 / 7 of ladder_filter_pack_split) against numpy (bit-exact: a permutation);
  * forward (with bias, activation, and the fused 1x1 projection of the last layer), every pixel including all four borders;
  * backward-data and the filter / bias gradient, borders and corners included;
  * ragged shapes: odd channel multiples of 16, non-square and 1-pixel-wide maps, batch sizes that are not a multiple of anything.

Tolerance (stated): fp32 accumulation of K = Cin <= 512 products followed by <= 16 additions of fp32 values -- 3e-6 of the output scale for the
forward and backward-data maps (measured ~3e-7), 3e-6 of the gradient scale for the filter gradient (M up to 2^17 terms, pairwise over splits)."""
import numpy as np
import pytest
import torch

from oracle import ladder_oracle as O
from test_gpu_split import _lib, close, dev, p

pytestmark = pytest.mark.gpu
TOL32 = 3e-6


def _ws(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")


def _pack(L, w, st):
    cin, cout = w.shape[2], w.shape[3]
    wd = dev(w)
    wcat = torch.full((cin, 9 * cout), float("nan"), device="cuda")
    wcatT = torch.full((9 * cout, cin), float("nan"), device="cuda")
    assert L.query("ladder_filter_pack_split_bytes", 1, cin, 9 * cout, 0) == wcat.numel() * 4
    L.call("ladder_filter_pack_split", p(wd), p(wcat), 1, cin, 9 * cout, 6, 0, st)
    L.call("ladder_filter_pack_split", p(wd), p(wcatT), 1, 9 * cout, cin, 7, 0, st)
    return wcat, wcatT


def _forward(L, xd, wcat, bd, N, H, W, cin, cout, act, st, y=True, proj=None):
    M = N * H * W
    z = torch.full((M, 9 * cout), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_igemm_fwd_workspace_bytes", M, cin, 9 * cout))
    L.call("ladder_dense_fwd", p(xd), p(wcat), None, p(z), M, cin, 9 * cout, 0, p(ws), ws.numel(), st)
    yd = torch.full((N, 2 * H, 2 * W, cout), float("nan"), device="cuda") if y else None
    out = None
    if proj is not None:
        pwd, pbd = proj
        out = torch.full((N, 2 * H, 2 * W, pwd.shape[1]), float("nan"), device="cuda")
        L.call("ladder_up2proj_fwd_combine", p(z), p(bd), p(yd), p(pwd), p(pbd), p(out), pwd.shape[1], N, H, W, cout, act, st)
    else:
        L.call("ladder_up2proj_fwd_combine", p(z), p(bd), p(yd), None, None, None, 0, N, H, W, cout, act, st)
    return yd, out


def _ref_fwd(x, w, b, act):
    N, H, W, _ = x.shape
    up = O.resize_bilinear_legacy(torch.as_tensor(x, dtype=torch.float64), 2 * H, 2 * W)
    y = O.conv2d_tf(up, torch.as_tensor(w, dtype=torch.float64), None if b is None else torch.as_tensor(b, dtype=torch.float64), 1, "same")
    return (O.leaky_relu(y) if act == "leaky_relu" else y).numpy()


def test_upproj_operand_layouts_are_permutations_of_the_bank(gpu_ctx):
    L = _lib()
    rng = np.random.default_rng(0)
    w = rng.standard_normal((3, 3, 48, 32)).astype(np.float32)
    wcat, wcatT = _pack(L, w, gpu_ctx.stream)
    ref = w.reshape(9, 48, 32).transpose(1, 0, 2).reshape(48, 9 * 32)
    assert np.array_equal(wcat.cpu().numpy(), ref)
    assert np.array_equal(wcatT.cpu().numpy(), ref.T)
    assert L.query("ladder_up2proj_eligible", 128, 64, 64, 128, 128) == 1
    assert L.query("ladder_up2proj_eligible", 128, 64, 64, 120, 128) == 0


CASES = [(16, 64, 64, 32, 128, "leaky_relu"), (128, 8, 8, 64, 256, "leaky_relu"), (3, 5, 7, 16, 48, None), (5, 1, 9, 48, 16, "leaky_relu"), (7, 6, 1, 16, 16, None),
         (2, 1, 1, 32, 32, None), (64, 16, 16, 256, 64, "leaky_relu")]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_c%d_co%d_%s" % c)
def test_upproj_forward_vs_oracle(gpu_ctx, case):
    L = _lib()
    N, H, W, cin, cout, act = case
    st = gpu_ctx.stream
    rng = np.random.default_rng(H * 100 + cin + cout)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    wcat, _ = _pack(L, w, st)
    y, _ = _forward(L, dev(x), wcat, dev(b), N, H, W, cin, cout, 1 if act else 0, st)
    close(y, _ref_fwd(x, w, b, act), TOL32, "y (every pixel)")


@pytest.mark.parametrize("case", [(16, 32, 32, 64, True), (4, 64, 64, 128, False), (3, 3, 5, 16, True)], ids=lambda c: "n%d_%dx%d_c%d_y%d" % c)
def test_upproj_forward_with_fused_projection_vs_oracle(gpu_ctx, case):
    L = _lib()
    N, H, W, cin, keep_y = case
    st = gpu_ctx.stream
    rng = np.random.default_rng(H + cin)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, 128)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32) * 0.1
    pw_ = (rng.standard_normal((128, 3)) / np.sqrt(128)).astype(np.float32)
    pb_ = rng.standard_normal(3).astype(np.float32) * 0.1
    wcat, _ = _pack(L, w, st)
    y, out = _forward(L, dev(x), wcat, dev(b), N, H, W, cin, 128, 1, st, y=keep_y, proj=(dev(pw_), dev(pb_)))
    ref = _ref_fwd(x, w, b, "leaky_relu")
    close(out, ref @ pw_.astype(np.float64) + pb_.astype(np.float64), TOL32, "projection")
    if keep_y:
        close(y, ref, TOL32, "map")


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_c%d_co%d_%s" % c)
def test_upproj_backward_vs_autograd(gpu_ctx, case):
    """dx, dw and db of resize x2 -> 3x3 conv from dy, against float64 autograd through the oracle's resize + convolution."""
    L = _lib()
    N, H, W, cin, cout, _ = case
    st = gpu_ctx.stream
    M = N * H * W
    rng = np.random.default_rng(cin + cout + H)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    dy = rng.standard_normal((N, 2 * H, 2 * W, cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    O.conv2d_tf(O.resize_bilinear_legacy(xt, 2 * H, 2 * W), wt, bt, 1, "same").backward(torch.as_tensor(dy, dtype=torch.float64))
    _, wcatT = _pack(L, w, st)
    xd, dyd = dev(x), dev(dy)
    d = torch.full((M, 9 * cout), float("nan"), device="cuda")
    L.call("ladder_up2proj_bwd_combine", p(dyd), p(d), N, H, W, cout, st)
    dx = torch.full((N, H, W, cin), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_igemm_fwd_workspace_bytes", M, 9 * cout, cin))
    L.call("ladder_dense_fwd", p(d), p(wcatT), None, p(dx), M, 9 * cout, cin, 0, p(ws), ws.numel(), st)
    close(dx, xt.grad, TOL32, "dx (every pixel)")
    dwcat = torch.full((cin, 9 * cout), float("nan"), device="cuda")
    db9 = torch.full((9 * cout,), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, cin, 9 * cout))
    L.call("ladder_dense_bwd_weight", p(xd), p(d), p(dwcat), p(db9), M, cin, 9 * cout, p(ws), ws.numel(), st)
    dw = torch.full((3, 3, cin, cout), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    L.call("ladder_up2proj_wgrad_unpack", p(dwcat), p(db9), p(dw), p(db), cin, cout, st)
    close(dw, wt.grad, TOL32, "dw")
    close(db, bt.grad, TOL32, "db")


def test_upproj_combine_is_the_transpose_of_itself(gpu_ctx):
    """<combine(z), dy> == <z, combine^T(dy)> at the full conv2d_7 map size (size-independent adjointness property)."""
    L = _lib()
    st = gpu_ctx.stream
    N, H, W, C = 8, 64, 64, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    z = torch.randn(N * H * W, 9 * C, device="cuda", generator=g)
    dy = torch.randn(N, 2 * H, 2 * W, C, device="cuda", generator=g)
    y = torch.empty(N, 2 * H, 2 * W, C, device="cuda")
    d = torch.empty_like(z)
    L.call("ladder_up2proj_fwd_combine", p(z), None, p(y), None, None, None, 0, N, H, W, C, 0, st)
    L.call("ladder_up2proj_bwd_combine", p(dy), p(d), N, H, W, C, st)
    torch.cuda.synchronize()
    a = float((y.double() * dy.double()).sum())
    b = float((z.double() * d.double()).sum())
    assert abs(a - b) < 1e-6 * (abs(a) + float(y.double().norm() * dy.double().norm())), (a, b)


def _ref_bwd_combine(dy):
    """D [N H W][9 C] = (shift o up)^T dy in float64: autograd through y = sum_rs shift_rs(up(Z_rs)), the oracle's legacy-bilinear resize and zero-padded shifts."""
    N, H2, W2, C = dy.shape
    H, W = H2 // 2, W2 // 2
    z = torch.zeros(N, H, W, 9, C, dtype=torch.float64, requires_grad=True)
    y = torch.zeros(N, H2, W2, C, dtype=torch.float64)
    for r in range(3):
        for s_ in range(3):
            up = torch.nn.functional.pad(O.resize_bilinear_legacy(z[:, :, :, 3 * r + s_, :], H2, W2), (0, 0, 1, 1, 1, 1))
            y = y + up[:, r:r + H2, s_:s_ + W2, :]                      # y[p, q] += up[p + r - 1, q + s - 1]
    y.backward(torch.as_tensor(dy, dtype=torch.float64))
    return z.grad.reshape(N * H * W, 9 * C).numpy()


WALK_CASES = [(2, 64, 64, 128, 0), (3, 5, 7, 48, 0), (2, 4, 9, 16, 3), (5, 16, 16, 256, 16), (1, 33, 2, 32, 5), (2, 8, 8, 64, 1)]


@pytest.mark.parametrize("case", WALK_CASES, ids=lambda c: "n%d_%dx%d_c%d_rows%d" % c)
def test_upproj_bwd_combine_walk_equals_the_neighbourhood_form(gpu_ctx, case):
    """The row-walking backward combination (a thread folds each high-resolution row along the columns once and keeps the folded rows of its window in
    registers; what ladder_up2proj_bwd_combine runs for H >= 4) against float64 autograd through the oracle's resize: 2e-6 of the scale; segment lengths that
    do and do not divide the map height, rows above / below the map, one-column maps."""
    L = _lib()
    N, H, W, C, rows = case
    st = gpu_ctx.stream
    g = torch.Generator(device="cuda").manual_seed(H + W + C)
    dy = torch.randn(N, 2 * H, 2 * W, C, device="cuda", generator=g)
    d0 = torch.full((N * H * W, 9 * C), float("nan"), device="cuda")
    d1 = torch.full((N * H * W, 9 * C), float("nan"), device="cuda")
    L.call("ladder_up2proj_bwd_combine", p(dy), p(d0), N, H, W, C, st)
    L.call("ladder_up2proj_bwd_combine_walk", p(dy), p(d1), N, H, W, C, rows, st)
    torch.cuda.synchronize()
    ref = _ref_bwd_combine(dy.cpu().numpy())
    close(d1, ref, 2e-6, "D (walk, %d rows a thread)" % rows)
    close(d0, ref, 2e-6, "D (library's choice)")
    from ladder_latent_data_distribution_modelling_amd._lib import LadderHipError
    with pytest.raises(LadderHipError, match="LADDER_E_SHAPE"):
        L.call("ladder_up2proj_bwd_combine_walk", p(dy), p(d1), N, 3, W, C, 0, st)               # fewer than four rows: the neighbourhood form


PROJ_BWD_CASES = [(4, 16, 16, 128, 3, "leaky_relu"), (3, 5, 7, 32, 1, None), (2, 4, 6, 128, 4, "leaky_relu"), (1, 9, 3, 64, 2, "leaky_relu"), (2, 64, 64, 128, 3, "leaky_relu"),
                  (5, 6, 1, 256, 3, "leaky_relu")]


@pytest.mark.parametrize("case", PROJ_BWD_CASES, ids=lambda c: "n%d_%dx%d_c%d_p%d_%s" % c)
def test_upproj_bwd_combine_from_projection_gradient_vs_oracle(gpu_ctx, case):
    """ladder_up2proj_bwd_combine_proj (reference codes/models.py:572-586, conv2d_7 -> leaky ReLU -> 1x1 conv2d_8): D from the pair's activated output y and the
    gradient dyp of the 1x1 convolution, and that convolution's filter / bias gradient, against float64: dy = act'(y) * (dyp . pw^T) through autograd of the
    oracle's resize + shifts, dpw = y^T dyp, dpb = column sums of dyp.  3e-6 of the scale."""
    L = _lib()
    N, H, W, C, pco, act = case
    st = gpu_ctx.stream
    rng = np.random.default_rng(H * 7 + C + pco)
    y = rng.standard_normal((N, 2 * H, 2 * W, C)).astype(np.float32)
    dyp = rng.standard_normal((N, 2 * H, 2 * W, pco)).astype(np.float32)
    pw_ = (rng.standard_normal((C, pco)) / np.sqrt(C)).astype(np.float32)
    assert L.query("ladder_up2proj_bwd_combine_proj_eligible", N, H, W, C, pco) == 1
    y64, g64 = y.astype(np.float64), dyp.astype(np.float64)
    dy64 = (g64 @ pw_.astype(np.float64).T) * (np.where(y64 > 0, 1.0, 0.2) if act else 1.0)
    dref = _ref_bwd_combine(dy64)
    yd, gd, pwd = dev(y), dev(dyp), dev(pw_)
    d = torch.full((N * H * W, 9 * C), float("nan"), device="cuda")
    dpw = torch.full((C, pco), float("nan"), device="cuda")
    dpb = torch.full((pco,), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_up2proj_bwd_combine_proj_workspace_bytes", N, H, W, C, pco))
    L.call("ladder_up2proj_bwd_combine_proj", p(yd), p(gd), p(pwd), p(d), p(dpw), p(dpb), pco, N, H, W, C, 1 if act else 0, p(ws), ws.numel(), st)
    torch.cuda.synchronize()
    close(d, dref, TOL32, "D")
    close(dpw, y64.reshape(-1, C).T @ g64.reshape(-1, pco), TOL32, "projection filter gradient")
    close(dpb, g64.reshape(-1, pco).sum(0), TOL32, "projection bias gradient")
    # no bias gradient wanted; workspace too small; not eligible
    L.call("ladder_up2proj_bwd_combine_proj", p(yd), p(gd), p(pwd), p(d), p(dpw), None, pco, N, H, W, C, 1 if act else 0, p(ws), ws.numel(), st)
    from ladder_latent_data_distribution_modelling_amd._lib import LadderHipError
    with pytest.raises(LadderHipError, match="LADDER_E_WORKSPACE"):
        L.call("ladder_up2proj_bwd_combine_proj", p(yd), p(gd), p(pwd), p(d), p(dpw), None, pco, N, H, W, C, 0, p(ws), 8, st)
    q = lambda *a: L.query("ladder_up2proj_bwd_combine_proj_eligible", *a)  # noqa: E731
    assert q(N, H, W, 24, pco) == 0 and q(N, H, W, C, 5) == 0 and q(N, H, W, 16, 1) == 0 and q(N, 3, W, C, pco) == 0 and q(N, H, W, 512, pco) == 0


F4_CASES = [(128, 2, 2, 64, 64), (3, 1, 1, 16, 32), (5, 3, 2, 32, 16), (2, 4, 5, 16, 16)]


@pytest.mark.parametrize("case", F4_CASES, ids=lambda c: "n%d_%dx%d_c%d_co%d" % c)
def test_upproj_factor4_forward_and_backward_vs_oracle(gpu_ctx, case):
    """The 2x2 -> 8x8 resize in front of decoder conv2d_3 (codes/models.py:536-542) in the projected form: forward against the oracle's resize + conv,
    dx / dw / db against float64 autograd through them."""
    L = _lib()
    N, H, W, cin, cout = case
    st = gpu_ctx.stream
    M, F = N * H * W, 4
    assert L.query("ladder_upfproj_eligible", F, N, H, W, cin, cout) == 1
    assert L.query("ladder_upfproj_eligible", 3, N, H, W, cin, cout) == 0
    rng = np.random.default_rng(cin + cout + H)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    dy = rng.standard_normal((N, F * H, F * W, cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    ref = O.leaky_relu(O.conv2d_tf(O.resize_bilinear_legacy(xt, F * H, F * W), wt, bt, 1, "same"))
    wcat, wcatT = _pack(L, w, st)
    xd, dyd = dev(x), dev(dy)
    z = torch.full((M, 9 * cout), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_igemm_fwd_workspace_bytes", M, cin, 9 * cout))
    L.call("ladder_dense_fwd", p(xd), p(wcat), None, p(z), M, cin, 9 * cout, 0, p(ws), ws.numel(), st)
    y = torch.full((N, F * H, F * W, cout), float("nan"), device="cuda")
    L.call("ladder_upfproj_fwd_combine", p(z), p(dev(b)), p(y), F, N, H, W, cout, 1, st)
    close(y, ref.detach(), TOL32, "y")
    # backward of the un-activated pair (the engine applies the activation derivative to dy first)
    O.conv2d_tf(O.resize_bilinear_legacy(xt, F * H, F * W), wt, bt, 1, "same").backward(torch.as_tensor(dy, dtype=torch.float64))
    d = torch.full((M, 9 * cout), float("nan"), device="cuda")
    L.call("ladder_upfproj_bwd_combine", p(dyd), p(d), F, N, H, W, cout, st)
    dx = torch.full((N, H, W, cin), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_igemm_fwd_workspace_bytes", M, 9 * cout, cin))
    L.call("ladder_dense_fwd", p(d), p(wcatT), None, p(dx), M, 9 * cout, cin, 0, p(ws), ws.numel(), st)
    close(dx, xt.grad, TOL32, "dx")
    dwcat = torch.full((cin, 9 * cout), float("nan"), device="cuda")
    db9 = torch.full((9 * cout,), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, cin, 9 * cout))
    L.call("ladder_dense_bwd_weight", p(xd), p(d), p(dwcat), p(db9), M, cin, 9 * cout, p(ws), ws.numel(), st)
    dw = torch.full((3, 3, cin, cout), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    L.call("ladder_up2proj_wgrad_unpack", p(dwcat), p(db9), p(dw), p(db), cin, cout, st)
    close(dw, wt.grad, TOL32, "dw")
    close(db, bt.grad, TOL32, "db")


@pytest.mark.parametrize("shape", [(8192, 128, 1152), (16384, 1152, 128), (8192, 512, 2304), (24576, 32, 256), (8320, 2304, 256)], ids=lambda s: "x".join(map(str, s)))
def test_persistent_dense_kernel_vs_float64(gpu_ctx, shape):
    """csrc/densef32.hip (the GEMM-shaped calls of the projected pairs: ladder_dense_fwd / ladder_dense_bwd_data with M >= 8192, M and N multiples of
    128, K of 32): bias + activation forward, gated backward-data; against float64 numpy.  Tile counts that do not divide over the persistent
    workgroups (8320 rows = 65 row tiles) included.  fp32 accumulation of K <= 2304 products: 3e-6 of the output scale."""
    L = _lib()
    M, K, N = shape
    st = gpu_ctx.stream
    rng = np.random.default_rng(M + K + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) * 0.1
    ad, bd = dev(a), dev(b)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    y = torch.full((M, N), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_igemm_fwd_workspace_bytes", M, K, N))
    L.call("ladder_dense_fwd", p(ad), p(bd), p(dev(bias)), p(y), M, K, N, 1, p(ws), ws.numel(), st)
    r1 = ref + bias.astype(np.float64)
    close(y, np.where(r1 > 0, r1, 0.2 * r1), TOL32, "forward (bias, leaky ReLU)")
    # backward-data form: dx [M, N] = dy [M, K] . wT [K, N], times act'(gate)
    gate = rng.standard_normal((M, N)).astype(np.float32)
    dx = torch.full((M, N), float("nan"), device="cuda")
    L.call("ladder_dense_bwd_data", p(ad), p(bd), p(dx), M, N, K, p(dev(gate)), 1, p(ws), ws.numel(), st)
    close(dx, ref * np.where(gate > 0, 1.0, 0.2), TOL32, "backward-data (gated)")
    # the K-contiguous forms (gemm_nt16_f32_kernel, v_mfma_f32_16x16x4_f32): the weight operand transposed, same results to rounding
    bt = dev(np.ascontiguousarray(b.T))
    y2 = torch.full((M, N), float("nan"), device="cuda")
    L.call("ladder_dense_fwd_nt", p(ad), p(bt), p(dev(bias)), p(y2), M, K, N, 1, st)
    close(y2, np.where(r1 > 0, r1, 0.2 * r1), TOL32, "forward, K-contiguous weights")
    dx2 = torch.full((M, N), float("nan"), device="cuda")
    L.call("ladder_dense_bwd_data_nt", p(ad), p(bt), p(dx2), M, N, K, p(dev(gate)), 1, st)
    close(dx2, ref * np.where(gate > 0, 1.0, 0.2), TOL32, "backward-data, K-contiguous weights (gated)")
    y3 = torch.full((M, N), float("nan"), device="cuda")
    L.call("ladder_dense_fwd_nt", p(ad), p(bt), None, p(y3), M, K, N, 0, st)
    close(y3, ref, TOL32, "plain product, K-contiguous weights")


@pytest.mark.parametrize("shape", [(16480, 128, 256), (8192, 256, 1152), (65536, 128, 1152), (8200, 512, 128)], ids=lambda s: "x".join(map(str, s)))
def test_persistent_dense_filter_gradient_vs_float64(gpu_ctx, shape):
    """gemm_tn_f32_kernel (csrc/densef32.hip) behind ladder_dense_bwd_weight: dW = x^T dy and db = column sums of dy over M rows, row counts that are
    not a multiple of the 32-row chunk or of the split included; against float64 numpy.  Two runs must agree bit for bit (fixed-order split sums)."""
    L = _lib()
    M, K, N = shape
    st = gpu_ctx.stream
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    xd, dyd = dev(x), dev(dy)
    ws = _ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, K, N))
    outs = []
    for _ in range(2):
        dw = torch.full((K, N), float("nan"), device="cuda")
        db = torch.full((N,), float("nan"), device="cuda")
        L.call("ladder_dense_bwd_weight", p(xd), p(dyd), p(dw), p(db), M, K, N, p(ws), ws.numel(), st)
        outs.append((dw.cpu().numpy(), db.cpu().numpy()))
    close(outs[0][0], x.astype(np.float64).T @ dy.astype(np.float64), TOL32, "dw")
    close(outs[0][1], dy.astype(np.float64).sum(0), TOL32, "db")
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


# ---- round 6: the fused forward (Z in an LDS ring, csrc/upproj.hip: up2proj_fused_fwd_kernel) ------------------------------------------------------------
# (N, H, W, Cin, Cout, act): every eligible width (8 / 16 / 32 / 64 -> 8 / 4 / 2 / 1 images per row step), H = 2 (one ring turn) ... 64, non-square maps,
# several slabs, Cin = 64 (two chunks: the minimum) ... 512
FUSED_CASES = [(2, 64, 64, 128, 128, "leaky_relu"), (4, 32, 32, 256, 128, "leaky_relu"), (8, 16, 16, 256, 256, None), (16, 8, 8, 512, 256, "leaky_relu"),
               (8, 2, 8, 64, 16, None), (2, 3, 32, 96, 48, "leaky_relu"), (1, 5, 64, 64, 32, None), (12, 7, 16, 64, 16, "leaky_relu")]


@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: "n%d_%dx%d_c%d_co%d_%s" % c)
def test_upproj_fused_forward_vs_oracle(gpu_ctx, case):
    """ladder_up2proj_fused_fwd against the float64 oracle's resize + convolution on every pixel (all four borders, image boundaries inside a row step),
    and against the two-call form (GEMM + combination) it replaces: same products, another summation order -> the same 3e-6 bar."""
    L = _lib()
    N, H, W, cin, cout, act = case
    st = gpu_ctx.stream
    assert L.query("ladder_up2proj_fused_eligible", N, H, W, cin, cout) == 1
    rng = np.random.default_rng(H * 100 + cin + cout + N)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    wcat, wcatT = _pack(L, w, st)
    xd, bd = dev(x), dev(b)
    y = torch.full((N, 2 * H, 2 * W, cout), float("nan"), device="cuda")
    L.call("ladder_up2proj_fused_fwd", p(xd), p(wcatT), p(bd), p(y), None, None, None, 0, N, H, W, cin, cout, 1 if act else 0, None, 0, st)
    ref = _ref_fwd(x, w, b, act)
    close(y, ref, TOL32, "y (every pixel)")
    y2, _ = _forward(L, xd, wcat, bd, N, H, W, cin, cout, 1 if act else 0, st)
    close(y, y2.cpu().numpy().astype(np.float64), TOL32, "fused vs two-call form")
    # bit-reproducible: a second launch gives the identical tensor
    y3 = torch.full_like(y, float("nan"))
    L.call("ladder_up2proj_fused_fwd", p(xd), p(wcatT), p(bd), p(y3), None, None, None, 0, N, H, W, cin, cout, 1 if act else 0, None, 0, st)
    assert torch.equal(y, y3)


# shapes that take the 128-pixel row step (32-pixel wave tile): Cin > 128, W 16 / 32, >= 256 workgroups -- two rows (one hand-over of the plane row), odd row counts,
# the real conv2d_5 / conv2d_6 shapes
WIDE_CASES = [(128, 2, 32, 160, 128, "leaky_relu"), (128, 5, 32, 256, 128, None), (64, 3, 16, 192, 512, "leaky_relu"), (128, 16, 16, 256, 256, "leaky_relu"),
              (128, 32, 32, 256, 128, "leaky_relu")]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "n%d_%dx%d_c%d_co%d_%s" % c)
def test_upproj_fused_forward_wide_tile_vs_oracle(gpu_ctx, case):
    """up2proj_fused2_fwd_kernel (two pixel tiles per MFMA wave, rolled weight fragments, two stages, two combination items per thread) against the float64
    oracle on every pixel and against the 64-pixel kernel's result (LADDER_UP2FUSE_NO_PT2 cannot be toggled in-process: the two-call form stands in)."""
    L = _lib()
    N, H, W, cin, cout, act = case
    st = gpu_ctx.stream
    assert L.query("ladder_up2proj_fused_wide_tile", N, H, W, cin, cout) == 1
    rng = np.random.default_rng(H * 10 + cin + N)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    wcat, wcatT = _pack(L, w, st)
    xd, bd = dev(x), dev(b)
    y = torch.full((N, 2 * H, 2 * W, cout), float("nan"), device="cuda")
    L.call("ladder_up2proj_fused_fwd", p(xd), p(wcatT), p(bd), p(y), None, None, None, 0, N, H, W, cin, cout, 1 if act else 0, None, 0, st)
    y2, _ = _forward(L, xd, wcat, bd, N, H, W, cin, cout, 1 if act else 0, st)
    close(y, y2.cpu().numpy().astype(np.float64), TOL32, "wide tile vs two-call form")
    if N * H * W * cout <= (1 << 24):                                        # (the float64 convolution of the two largest cases takes minutes on the host)
        close(y, _ref_fwd(x, w, b, act), TOL32, "y (every pixel)")
    y3 = torch.full_like(y, float("nan"))
    L.call("ladder_up2proj_fused_fwd", p(xd), p(wcatT), p(bd), p(y3), None, None, None, 0, N, H, W, cin, cout, 1 if act else 0, None, 0, st)
    assert torch.equal(y, y3)


@pytest.mark.parametrize("case", [(2, 64, 64, 128, 128, 3, True), (4, 32, 32, 64, 128, 3, False), (8, 4, 16, 64, 64, 4, True), (8, 6, 8, 96, 32, 1, False)],
                         ids=lambda c: "n%d_%dx%d_c%d_co%d_p%d_y%d" % c)
def test_upproj_fused_forward_with_projection_vs_oracle(gpu_ctx, case):
    """... with the 1x1 output conv through per-slab partial sums (any Cout that is a multiple of 16, proj_cout <= 4), with and without the y write."""
    L = _lib()
    N, H, W, cin, cout, pco, keep_y = case
    st = gpu_ctx.stream
    rng = np.random.default_rng(H + cin + pco)
    x = rng.standard_normal((N, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    pw_ = (rng.standard_normal((cout, pco)) / np.sqrt(cout)).astype(np.float32)
    pb_ = rng.standard_normal(pco).astype(np.float32) * 0.1
    _, wcatT = _pack(L, w, st)
    xd, bd, pwd, pbd = dev(x), dev(b), dev(pw_), dev(pb_)
    y = torch.full((N, 2 * H, 2 * W, cout), float("nan"), device="cuda") if keep_y else None
    out = torch.full((N, 2 * H, 2 * W, pco), float("nan"), device="cuda")
    ws = _ws(L.query("ladder_up2proj_fused_workspace_bytes", N, H, W, cout, pco))
    L.call("ladder_up2proj_fused_fwd", p(xd), p(wcatT), p(bd), p(y), p(pwd), p(pbd), p(out), pco, N, H, W, cin, cout, 1, p(ws), ws.numel(), st)
    ref = _ref_fwd(x, w, b, "leaky_relu")
    close(out, ref @ pw_.astype(np.float64) + pb_.astype(np.float64), TOL32, "projection")
    if keep_y:
        close(y, ref, TOL32, "map")


def test_upproj_fused_eligibility_and_errors(gpu_ctx):
    L = _lib()
    q = lambda *a: L.query("ladder_up2proj_fused_eligible", *a)
    assert q(128, 64, 64, 128, 128) == 1 and q(128, 32, 32, 256, 128) == 1 and q(128, 16, 16, 256, 256) == 1 and q(128, 8, 8, 512, 256) == 1
    assert q(128, 1, 8, 512, 512) == 0          # one row: no ring turn
    assert q(3, 8, 8, 512, 256) == 0            # 8 images per row step
    assert q(8, 8, 8, 32, 16) == 0              # one K chunk (the ring hand-over needs two barriers per row)
    assert q(8, 8, 12, 64, 16) == 0 and q(8, 8, 8, 64, 24) == 0 and q(8, 8, 8, 72, 16) == 0
    from ladder_latent_data_distribution_modelling_amd._lib import LadderHipError
    x = torch.zeros(8, 8, 8, 64, device="cuda")
    wT = torch.zeros(9 * 16, 64, device="cuda")
    y = torch.zeros(8, 16, 16, 16, device="cuda")
    with pytest.raises(LadderHipError, match="LADDER_E_SHAPE"):
        L.call("ladder_up2proj_fused_fwd", p(x), p(wT), None, None, None, None, None, 0, 8, 8, 8, 64, 16, 0, None, 0, gpu_ctx.stream)      # nothing to write
    with pytest.raises(LadderHipError, match="LADDER_E_WORKSPACE"):
        L.call("ladder_up2proj_fused_fwd", p(x), p(wT), None, p(y), p(wT), None, p(y), 3, 8, 8, 8, 64, 16, 0, None, 0, gpu_ctx.stream)      # projection without workspace
