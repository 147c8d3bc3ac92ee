"""Pins the CPU oracle (oracle/ladder_oracle.py) -- the checker every GPU parity test relies on -- against
(a) the reference's own artefacts (fitted mixture, checkpoint indexes) and (b) independent restatements
(scipy, pure-numpy loops, worked examples of SURVEY Appendix C).  No GPU needed."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ladder_oracle as O


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def test_gmm_log_prob_vs_scipy_on_reference_mixture(golden_dir):
    """figures/mnist_digit/result/GM_prior_info.npz (written by codes/base.py:769-777): 50-component 2-D mixture."""
    from scipy.special import logsumexp
    from scipy.stats import multivariate_normal
    fix = np.load(os.path.join(golden_dir, "GM_prior_info.npz"))
    for w, m, c in ((fix["w_full"], fix["m_full"], fix["K_full"]), (fix["w_active"], fix["m_active"], fix["K_active"])):
        assert np.all(np.linalg.eigvalsh(c) > 0)
        rng = np.random.default_rng(0)
        t = rng.normal(0, 3, size=(7, 11, 2))
        comp = np.stack([multivariate_normal(m[k], c[k]).logpdf(t) for k in range(len(w))], -1)
        ref = logsumexp(comp + np.log(w / w.sum()), axis=-1)
        got = O.gmm_log_prob(t64(t), t64(w), t64(m), t64(c)).numpy()
        assert np.abs(got - ref).max() < 1e-10
    assert len(fix["w_active"]) == 27 and abs(fix["w_active"].sum() - 1) < 1e-12


def test_variable_inventory_matches_reference_checkpoints(golden_dir):
    """Variable names + shapes of all three architectures vs pretrained_models/*/*.index."""
    inv = json.load(open(os.path.join(golden_dir, "ckpt_inventory.json")))
    trained = {"celeba": dict(num_hidden_units=512, code_size=256, representation_size=32, dim_input_channel=3),
               "mnist_digit": dict(num_hidden_units=256, code_size=16, representation_size=2, dim_input_channel=1),
               "mnist_fashion": dict(num_hidden_units=512, code_size=32, representation_size=2, dim_input_channel=1)}
    totals = {"celeba": (18861572, 2413889), "mnist_digit": (1120502, 2121749), "mnist_fashion": (3448898, 2138149)}
    for exp, over in trained.items():
        cfg = dict(exp_name=exp, prior="ours", kernel_size=3, num_hidden_units_inner_VAE=512, n_layers_inner_VAE=5, **over)
        specs = O.param_specs(cfg)
        vae = {k: list(v) for k, v in specs.items() if O.group_of(k) in ("ae", "sigma")}
        pri = {k: list(v) for k, v in specs.items() if O.group_of(k) in ("prior", "inner_sigma")}
        assert vae == inv[exp]["vae-model"], exp
        assert pri == inv[exp]["prior-model"], exp
        n = lambda d: sum(int(np.prod(s)) if s else 1 for s in d.values())
        assert (n(vae), n(pri)) == totals[exp]


def _conv_loops(x, w, b, stride, pt, pl, Ho, Wo):
    N, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    y = np.zeros((N, Ho, Wo, Co))
    for n in range(N):
        for ho in range(Ho):
            for wo in range(Wo):
                for r in range(kh):
                    for s in range(kw):
                        hi, wi = ho * stride + r - pt, wo * stride + s - pl
                        if 0 <= hi < H and 0 <= wi < W:
                            y[n, ho, wo] += x[n, hi, wi] @ w[r, s]
    return y + b


@pytest.mark.parametrize("H,k,s,pad", [(8, 3, 2, "same"), (7, 3, 2, "same"), (6, 3, 1, "same"), (6, 3, 1, "valid"), (9, 5, 1, "valid"), (4, 1, 1, "same")])
def test_conv2d_tf_semantics_vs_loops(H, k, s, pad):
    """TF SAME: out=ceil(in/s), pad_total=max((out-1)s+k-in,0), before=total//2 (0 before / 1 after for k=3,s=2, even in)."""
    rng = np.random.default_rng(H * 10 + k)
    x, w, b = rng.standard_normal((2, H, H, 3)), rng.standard_normal((k, k, 3, 4)), rng.standard_normal(4)
    got = O.conv2d_tf(t64(x), t64(w), t64(b), s, pad).numpy()
    if pad == "same":
        pt, _, Ho = O.same_pad(H, k, s)
    else:
        pt, Ho = 0, (H - k) // s + 1
    assert np.abs(got - _conv_loops(x, w, b, s, pt, pt, Ho, Ho)).max() < 1e-12
    if (H, k, s, pad) == (8, 3, 2, "same"):
        assert O.same_pad(8, 3, 2) == (0, 1, 4)


def test_resize_legacy_bilinear_worked_examples():
    """2->8 of [a,b] = [a,.75a+.25b,.5a+.5b,.25a+.75b,b,b,b,b]; 2x: out[2i]=in[i], out[2i+1]=mean; 1->2 broadcast."""
    a, b = 3.0, -5.0
    x = t64([[a, b]]).reshape(1, 1, 2, 1)
    got = O.resize_bilinear_legacy(x, 1, 8).numpy().ravel()
    assert np.allclose(got, [a, .75 * a + .25 * b, .5 * a + .5 * b, .25 * a + .75 * b, b, b, b, b])
    rng = np.random.default_rng(0)
    v = rng.standard_normal(8)
    got = O.resize_bilinear_legacy(t64(v).reshape(1, 8, 1, 1), 16, 1).numpy().ravel()
    exp = np.empty(16)
    exp[0::2] = v
    exp[1::2] = 0.5 * (v + np.append(v[1:], v[-1]))
    assert np.allclose(got, exp)
    one = t64(rng.standard_normal((2, 1, 1, 5)))
    assert np.allclose(O.resize_bilinear_legacy(one, 2, 2).numpy(), np.broadcast_to(one.numpy(), (2, 2, 2, 5)))
    same = t64(rng.standard_normal((1, 4, 4, 2)))
    assert O.resize_bilinear_legacy(same, 4, 4) is same


def test_depth_to_space_dcr_and_symmetric_pad():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 4, 8))
    r, Cp = 2, 2
    y = O.depth_to_space(t64(x), r).numpy()
    for h in range(3):
        for w in range(4):
            for i in range(r):
                for j in range(r):
                    assert np.array_equal(y[:, h * r + i, w * r + j, :], x[:, h, w, (i * r + j) * Cp:(i * r + j + 1) * Cp])
    img = rng.standard_normal((1, 5, 6, 2))
    assert np.array_equal(O.pad_symmetric(t64(img), 2).numpy(), np.pad(img, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="symmetric"))


def test_norms_and_activations():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 4, 5, 6)) * 2 + 1
    g, b = rng.standard_normal(6), rng.standard_normal(6)
    y = O.batch_norm_train(t64(x), t64(g), t64(b)).numpy()
    flat = x.reshape(-1, 6)
    assert np.allclose(y, (x - flat.mean(0)) / np.sqrt(flat.var(0) + 1e-3) * g + b)
    z = O.instance_norm(t64(x)).numpy()
    assert np.allclose(z, (x - x.mean((1, 2), keepdims=True)) / np.sqrt(x.var((1, 2), keepdims=True) + 1e-6))
    assert np.allclose(O.leaky_relu(t64([-2.0, 0.0, 3.0])).numpy(), [-0.4, 0.0, 3.0])
    st = rng.standard_normal((3, 12))
    s = O.style_mod(t64(x), t64(st)).numpy()
    assert np.allclose(s, x * (st[:, None, None, :6] + 1) + st[:, None, None, 6:])


def test_tf_max_min_gradient_routing():
    """tf.maximum/minimum send the gradient to the FIRST argument on ties (x>=y / x<=y)."""
    x = torch.tensor(0.1, dtype=torch.float64, requires_grad=True)
    y = torch.tensor(0.1, dtype=torch.float64, requires_grad=True)
    O.tf_minimum(O.tf_maximum(x, torch.tensor(0.05, dtype=torch.float64)), y).backward()
    assert x.grad.item() == 1.0 and y.grad is None or y.grad.item() == 0.0


def test_adam_tf_form_first_steps():
    th, m, v = np.array([1.0, -2.0]), np.zeros(2), np.zeros(2)
    g = np.array([0.5, -3.0])            # second element clips to -1
    O.adam_tf(th, g, m, v, 1, 1e-3)
    gc = np.array([0.5, -1.0])
    lr_t = 1e-3 * np.sqrt(1 - 0.95) / (1 - 0.9)
    assert np.allclose(m, 0.1 * gc) and np.allclose(v, 0.05 * gc ** 2)
    assert np.allclose(th, np.array([1.0, -2.0]) - lr_t * (0.1 * gc) / (np.sqrt(0.05 * gc ** 2) + 1e-8))


def test_sg_feed_is_standard_normal_and_closed_form():
    """K identical N(0,I) components with uniform weights == standard normal log-prob (codes/base.py:870-876)."""
    cfg = dict(n_mixtures=7, representation_size=3)
    gm = O.sg_feed(cfg)
    t = t64(np.random.default_rng(0).standard_normal((5, 4, 3)))
    lp = O.gmm_log_prob(t, gm["weights"], gm["means"], gm["covs"]).numpy()
    assert np.allclose(lp, -1.5 * np.log(2 * np.pi) - 0.5 * (t.numpy() ** 2).sum(-1))


def test_elbo_gradient_finite_differences(golden_dir):
    """loss_ae / loss_prior gradients of the restated graph vs central finite differences (float64)."""
    d = np.load(os.path.join(golden_dir, "oracle_mnist_digit.npz"))
    cfg = json.loads(str(d["config"]))
    cfg.update(num_hidden_units=64, n_MC_samples=3)
    rng = np.random.default_rng(3)
    x = rng.random((2, 28, 28, 1))
    P = O.init_params(cfg, seed=4)
    gm = dict(weights=d["gm_w"], means=d["gm_m"], covs=d["gm_c"])
    noise = O.make_noise(cfg, 2, rng)
    st = O.OracleState(cfg, P, np.float64)
    for group, key, names in (("ae", "loss_ae", ["decoder/conv2d_2/kernel", "encoder/code_std_dev/bias"]),
                              ("prior", "loss_prior", ["prior/dense_3/kernel"])):
        res = O.run(st, x, noise, gm, False, False, train=group, lr=0.0)
        for name in names:
            g = res["_grads"][name]
            idx = np.unravel_index(np.argmax(np.abs(g)), g.shape) if g.ndim else ()
            h = 1e-6
            vals = []
            for sgn in (+1, -1):
                st.P[name][idx] += sgn * h
                vals.append(float(O.run(st, x, noise, gm, False, False)[key]))
                st.P[name][idx] -= sgn * h
            fd = (vals[0] - vals[1]) / (2 * h)
            assert abs(fd - g[idx]) < 1e-5 * max(1.0, abs(fd)), (name, fd, g[idx])


def test_golden_vectors_reproduce(golden_dir):
    """The committed fixtures are what the oracle produces today (first iteration, MNIST-digit)."""
    d = np.load(os.path.join(golden_dir, "oracle_mnist_digit.npz"))
    cfg = json.loads(str(d["config"]))
    st = O.OracleState(cfg, O.init_params(cfg, seed=1), np.float64)
    gm = dict(weights=d["gm_w"], means=d["gm_m"], covs=d["gm_c"])
    nz = [{k: d["it0_run%d_%s" % (r, k)] for k in ("eps_z", "eps_t", "eps_mc")} for r in (1, 2, 3, 4)]
    f = O.train_iteration(st, d["x"], nz, gm, cur_epoch=2, lr_ae=cfg["learning_rate_ae"])
    for k in ("elbo", "elbo_prior", "entropy_z", "l1_reconstruction_error"):
        assert abs(float(f["run1"][k]) - float(d["it0_run1_" + k])) < 1e-9 * max(1, abs(float(d["it0_run1_" + k])))
    assert abs(float(f["run3"]["elbo_prior"]) - float(d["it0_run3_elbo_prior"])) < 1e-9 * abs(float(d["it0_run3_elbo_prior"]))
