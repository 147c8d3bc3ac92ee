"""The algebra behind the upsample-fused decoder convolutions (csrc/convsplit.hip: filter-pack modes 3 / 4, ladder_conv3x3_up2_*), checked in
float64 against the oracle's resize_bilinear_legacy + conv2d_tf (reference codes/models.py:554-578) -- no GPU, no product code: this pins the
tables and the border rules the kernels implement.

  forward   conv3x3_same(up2(x))[2i+a, 2j+b] = sum_{dr,dc} W_eff[a,b][dr,dc] . x~[i+dr-1, j+dc-1],   W_eff = A_a (x) A_b applied to w,
            x~ = x with the halo -x[edge] above / left, +x[edge] below / right; exact but for the last output row / column, which are
            1x3 / 3x1 convolutions of the 1-D upsampled last row / column of x with the summed taps w[0] + w[1];
  backward  d/dx_lo = a zero-padded 5-tap / stride-2 correlation per axis (G_k = mirrored tables), exact but on the four border lines."""
import numpy as np
import torch

from oracle import ladder_oracle as O

A = np.zeros((2, 3, 3))
A[0] = [[0.5, 0, 0], [0.5, 1, 0.5], [0, 0, 0.5]]
A[1] = [[0, 0, 0], [1, 0.5, 0], [0, 0.5, 1]]


def _ref(x, w):
    N, H, W, _ = x.shape
    up = O.resize_bilinear_legacy(torch.as_tensor(x), 2 * H, 2 * W)
    return up.numpy(), O.conv2d_tf(up, torch.as_tensor(w), None, 1, "same").numpy()


def test_forward_classes_signed_halo_and_edge_lines():
    rng = np.random.default_rng(0)
    N, H, W, Ci, Co = 2, 6, 8, 5, 4
    x, w = rng.normal(size=(N, H, W, Ci)), rng.normal(size=(3, 3, Ci, Co))
    up, ref = _ref(x, w)
    xp = np.zeros((N, H + 2, W + 2, Ci))
    xp[:, 1:-1, 1:-1] = x
    xp[:, 0, 1:-1], xp[:, -1, 1:-1], xp[:, 1:-1, 0], xp[:, 1:-1, -1] = -x[:, 0], x[:, -1], -x[:, :, 0], x[:, :, -1]
    xp[:, 0, 0], xp[:, 0, -1], xp[:, -1, 0], xp[:, -1, -1] = x[:, 0, 0], -x[:, 0, -1], -x[:, -1, 0], x[:, -1, -1]
    out = np.zeros_like(ref)
    taps = 0
    for a in range(2):
        for b in range(2):
            we = np.einsum("dr,es,rsio->deio", A[a], A[b], w)
            o = np.zeros((N, H, W, Co))
            for dr in range(3):
                for dc in range(3):
                    if (a == 0 or dr >= 1) and (b == 0 or dc >= 1):          # the tap masks of ladder_conv3x3_up2_split
                        o += xp[:, dr:dr + H, dc:dc + W] @ we[dr, dc]
                        taps += 1
                    else:
                        assert not np.any(we[dr, dc])
            out[:, a::2, b::2] = o
    assert taps == 25
    np.testing.assert_allclose(out[:, :-1, :-1], ref[:, :-1, :-1], atol=1e-12)
    assert np.abs(out[:, -1] - ref[:, -1]).max() > 1e-3                      # the last row / column are NOT covered by the halo trick
    # ... they are line convolutions of the 1-D upsampled last row / column with the summed taps (ladder_conv3x3_up2_edges)
    uL = np.zeros((N, 2 * W + 2, Ci))
    uL[:, 1:-1] = up[:, -1]
    row = sum(uL[:, s:s + 2 * W] @ (w[0, s] + w[1, s]) for s in range(3))
    vL = np.zeros((N, 2 * H + 2, Ci))
    vL[:, 1:-1] = up[:, :, -1]
    col = sum(vL[:, r:r + 2 * H] @ (w[r, 0] + w[r, 1]) for r in range(3))
    np.testing.assert_allclose(row, ref[:, -1], atol=1e-12)
    np.testing.assert_allclose(col, ref[:, :, -1], atol=1e-12)


def test_backward_is_a_5x5_stride2_correlation_off_the_border_lines():
    rng = np.random.default_rng(1)
    N, H, W, Ci, Co = 2, 6, 7, 3, 4
    w, dy = rng.normal(size=(3, 3, Ci, Co)), rng.normal(size=(N, 2 * H, 2 * W, Co))
    x = torch.zeros(N, H, W, Ci, dtype=torch.float64, requires_grad=True)
    O.conv2d_tf(O.resize_bilinear_legacy(x, 2 * H, 2 * W), torch.as_tensor(w), None, 1, "same").backward(torch.as_tensor(dy))
    ref = x.grad.numpy()
    # per axis: dx[p] = sum_k G_k^T dy[2p + k], k = 2 (dr - 1) + a, coefficient of w[r] in tap dr of class a = A_a[2 - dr][r] (pack mode 4)
    dyp = np.zeros((N, 2 * H + 4, 2 * W + 4, Co))
    dyp[:, 2:-2, 2:-2] = dy
    got = np.zeros((N, H, W, Ci))
    taps = 0
    for a in range(2):
        for b in range(2):
            for dr in range(3):
                for dc in range(3):
                    if not ((a == 0 or dr <= 1) and (b == 0 or dc <= 1)):     # the tap masks of ladder_conv3x3_up2_bwd_data_split
                        continue
                    g = np.einsum("r,s,rsio->io", A[a][2 - dr], A[b][2 - dc], w)          # [Ci, Co]
                    ky, kx = 2 * (dr - 1) + a, 2 * (dc - 1) + b
                    got += dyp[:, 2 + ky:2 + ky + 2 * H:2, 2 + kx:2 + kx + 2 * W:2] @ g.T
                    taps += 1
    assert taps == 25
    np.testing.assert_allclose(got[:, 1:-1, 1:-1], ref[:, 1:-1, 1:-1], atol=1e-12)
    for line in (got[:, 0] - ref[:, 0], got[:, -1] - ref[:, -1], got[:, :, 0] - ref[:, :, 0], got[:, :, -1] - ref[:, :, -1]):
        assert np.abs(line).max() > 1e-3                                         # the four border lines need the strips (engine.Conv2D._dx_lowres)
    # what the strips compute: row 0 = R(d_up[0] + d_up[1] / 2), row H-1 = R(d_up[2H-3] / 2 + d_up[2H-2] + d_up[2H-1]), R = 1-D resize transpose
    dup = torch.zeros(N, 2 * H, 2 * W, Ci, dtype=torch.float64, requires_grad=True)
    O.conv2d_tf(dup, torch.as_tensor(w), None, 1, "same").backward(torch.as_tensor(dy))
    d = dup.grad.numpy()

    def R(t):                                                                  # [N, 2L, C] -> [N, L, C]
        L_ = t.shape[1] // 2
        lo = t[:, 0::2].copy()
        lo[:, 1:] += 0.5 * t[:, 1:-1:2]
        lo[:, :-1] += 0.5 * t[:, 1:-1:2]
        lo[:, -1] += t[:, -1]
        return lo[:, :L_]

    np.testing.assert_allclose(R(d[:, 0] + 0.5 * d[:, 1]), ref[:, 0], atol=1e-12)
    np.testing.assert_allclose(R(0.5 * d[:, -3] + d[:, -2] + d[:, -1]), ref[:, -1], atol=1e-12)
    np.testing.assert_allclose(R(d[:, :, 0] + 0.5 * d[:, :, 1]), ref[:, :, 0], atol=1e-12)
    np.testing.assert_allclose(R(0.5 * d[:, :, -3] + d[:, :, -2] + d[:, :, -1]), ref[:, :, -1], atol=1e-12)
