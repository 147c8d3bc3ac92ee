"""CPU oracle for the LaDDer training path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch-CPU tensors + autograd, float64 by default)
of the arithmetic the reference builds as a TF-1.15 graph.  It exists to CHECK the
HIP path; it is never shipped, measured as the product, or imported by the
package `ladder_latent_data_distribution_modelling_amd`.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

PARITY UNPINNED: the reference cannot be executed here (tensorflow-gpu==1.15 /
tensorflow_probability==0.8.0, reference requirements.txt:13-14, are absent and
un-installable) and the reference ships no tests or golden vectors for this path.
What IS pinned (tests/test_oracle_*.py):
  * mixture log-prob vs scipy.stats.multivariate_normal on the reference's own fitted
    mixture figures/mnist_digit/result/GM_prior_info.npz (tests/golden/GM_prior_info.npz);
  * variable names + shapes of all three architectures vs the reference's own
    checkpoint indexes pretrained_models/*/*.index (tests/golden/ckpt_inventory.json);
  * every TF default restated here (SAME padding, legacy bilinear, DCR depth-to-space,
    SYMMETRIC pad, BN/IN eps, TF-form Adam) against an independent pure-numpy loop
    restatement and worked examples.

Each function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)
LEAKY_ALPHA = 0.2          # tf.nn.leaky_relu default alpha
BN_EPS = 1e-3              # tf.layers.batch_normalization default epsilon
IN_EPS = 1e-6              # tf.contrib.layers.instance_norm default epsilon
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.95, 1e-8   # codes/base.py:459-461 (+TF default eps)


# --------------------------------------------------------------------------------------
# TF-1.15 primitive semantics
# --------------------------------------------------------------------------------------
def same_pad(n_in: int, k: int, s: int):
    """TF 'SAME' padding amounts (before, after) and output size."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    before = total // 2
    return before, total - before, out


def conv2d_tf(x, w, b=None, stride=1, padding="same"):
    """tf.layers.conv2d: NHWC input, HWIO filter, cross-correlation, bias add.

    codes/models.py:51-71,115-148,203-229,273-315,398-460,514-585.
    """
    kh, kw = int(w.shape[0]), int(w.shape[1])
    if padding.lower() == "same":
        pt, pb, _ = same_pad(int(x.shape[1]), kh, stride)
        pl, pr, _ = same_pad(int(x.shape[2]), kw, stride)
    else:
        pt = pb = pl = pr = 0
    xn = x.permute(0, 3, 1, 2)
    if pt or pb or pl or pr:
        xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn.contiguous(), w.permute(3, 2, 0, 1).contiguous(), bias=b, stride=stride)
    return y.permute(0, 2, 3, 1)


def dense(x, w, b=None):
    """tf.layers.dense: x @ W + b with W [in, out]."""
    y = x @ w
    return y if b is None else y + b


def leaky_relu(x):
    return torch.where(x > 0, x, LEAKY_ALPHA * x)


def relu(x):
    return torch.clamp_min(x, 0.0)


def tf_maximum(x, y):
    """tf.maximum with TF's gradient routing: to the first argument where x >= y (ties included)."""
    return torch.where(x >= y, x, y)


def tf_minimum(x, y):
    """tf.minimum with TF's gradient routing: to the first argument where x <= y (ties included)."""
    return torch.where(x <= y, x, y)


def f32c(v, dt):
    """A Python-float graph constant as TF sees it: rounded to float32 (then held in the working dtype)."""
    return torch.tensor(float(np.float32(v)), dtype=dt)


def act(x, name):
    if name is None or name == "none":
        return x
    if name == "leaky_relu":
        return leaky_relu(x)
    if name == "relu":
        return relu(x)
    if name == "tanh":
        return torch.tanh(x)
    raise ValueError(name)


def batch_norm_train(x, gamma, beta, eps=BN_EPS, allreduce=None):
    """tf.layers.batch_normalization(training=True): batch mean / biased variance over N,H,W.

    codes/models.py:398-460 (is_training is the constant True, models.py:471).
    `allreduce(t)` (optional) sums a tensor over data-parallel ranks; counts are
    summed the same way so that the statistics are those of the GLOBAL batch.
    """
    C = x.shape[-1]
    xf = x.reshape(-1, C)
    n = torch.tensor(float(xf.shape[0]), dtype=x.dtype)
    s1 = xf.sum(0)
    if allreduce is not None:
        s1 = allreduce(s1)
        n = allreduce(n)
    mean = s1 / n
    s2 = ((xf - mean) ** 2).sum(0)
    if allreduce is not None:
        s2 = allreduce(s2)
    var = s2 / n
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def instance_norm(x, eps=IN_EPS):
    """tf.contrib.layers.instance_norm(center=False, scale=False): moments over H,W."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps)


def style_mod(x, style):
    """codes/modules.py:6-10 with `style` = dense(dlatent, 2C) already evaluated."""
    C = x.shape[-1]
    s = style.reshape(-1, 2, 1, 1, C)
    return x * (s[:, 0] + 1.0) + s[:, 1]


def resize_bilinear_legacy(x, oh, ow):
    """tf.image.resize_images v1: bilinear, align_corners=False, half_pixel_centers=False.

    src = dst * (in/out); lo = floor(src); hi = min(lo+1, in-1); lerp.
    codes/models.py:519,538,544,555,561,572,578.
    """
    N, H, W, C = x.shape
    if (H, W) == (oh, ow):
        return x

    def axis(n_in, n_out):
        src = torch.arange(n_out, dtype=torch.float64) * (n_in / n_out)
        lo = torch.floor(src).to(torch.long)
        hi = torch.clamp(lo + 1, max=n_in - 1)
        frac = (src - lo.to(torch.float64)).to(x.dtype)
        return lo, hi, frac

    ylo, yhi, yf = axis(H, oh)
    xlo, xhi, xf = axis(W, ow)
    top = x[:, ylo]
    bot = x[:, yhi]
    yf = yf.view(1, oh, 1, 1)
    rows = top + (bot - top) * yf
    left = rows[:, :, xlo]
    right = rows[:, :, xhi]
    xf = xf.view(1, 1, ow, 1)
    return left + (right - left) * xf


def depth_to_space(x, r):
    """tf.nn.depth_to_space NHWC (DCR): out[b,h*r+i,w*r+j,c] = in[b,h,w,(i*r+j)*C'+c]."""
    N, H, W, C = x.shape
    Cp = C // (r * r)
    y = x.reshape(N, H, W, r, r, Cp).permute(0, 1, 3, 2, 4, 5)
    return y.reshape(N, H * r, W * r, Cp)


def pad_symmetric(x, p):
    """tf.pad(..., 'SYMMETRIC') on H and W by p (mirror INCLUDING the edge pixel)."""
    if p == 0:
        return x
    H, W = x.shape[1], x.shape[2]
    hi = torch.tensor(list(range(p - 1, -1, -1)) + list(range(H)) + list(range(H - 1, H - 1 - p, -1)))
    wi = torch.tensor(list(range(p - 1, -1, -1)) + list(range(W)) + list(range(W - 1, W - 1 - p, -1)))
    return x[:, hi][:, :, wi]


def gmm_log_prob(t, weights, means, covs):
    """tfd.Mixture(Categorical(probs=w), [MVNFullCovariance(m_k, S_k)]).log_prob(t).

    codes/base.py:109-124.  log-prob = logsumexp_k(log(w_k/sum w) + logN(t; m_k, S_k)),
    logN via scale_tril = cholesky(S_k).  t: [..., R] -> [...].
    """
    R = t.shape[-1]
    Lc = torch.linalg.cholesky(covs)                       # [K,R,R]
    logw = torch.log(weights) - torch.log(weights.sum())
    diff = t.unsqueeze(-2) - means                         # [...,K,R]
    shp = diff.shape
    d2 = diff.reshape(-1, shp[-2], R).permute(1, 2, 0)     # [K,R,M]
    y = torch.linalg.solve_triangular(Lc, d2, upper=False)  # [K,R,M]
    maha = (y ** 2).sum(1).permute(1, 0).reshape(shp[:-1])  # [...,K]
    logdet = torch.log(torch.diagonal(Lc, dim1=-2, dim2=-1)).sum(-1)   # [K]
    comp = -0.5 * maha - logdet - 0.5 * R * LOG_2PI
    return torch.logsumexp(comp + logw, dim=-1)


# --------------------------------------------------------------------------------------
# Variable inventory (TF names, shapes) and initialisation
# --------------------------------------------------------------------------------------
def _tfname(base, idx):
    return base if idx == 0 else "%s_%d" % (base, idx)


class _Namer:
    """Reproduces TF's per-scope unique layer naming (conv2d, conv2d_1, dense, dense_1, ...)."""

    def __init__(self, scope):
        self.scope = scope
        self.count = {}

    def __call__(self, kind):
        i = self.count.get(kind, 0)
        self.count[kind] = i + 1
        return "%s/%s" % (self.scope, _tfname(kind, i))


def param_specs(config):
    """Ordered {tf_variable_name: shape} for config['exp_name']."""
    exp = config["exp_name"]
    nh = int(config["num_hidden_units"])
    Z = int(config["code_size"])
    ks = int(config.get("kernel_size", 3))
    specs = OrderedDict()

    def conv(namer, kh, cin, cout):
        n = namer("conv2d")
        specs[n + "/kernel"] = (kh, kh, cin, cout)
        specs[n + "/bias"] = (cout,)

    def dens(namer, cin, cout, name=None):
        n = namer("dense") if name is None else "%s/%s" % (namer.scope, name)
        specs[n + "/kernel"] = (cin, cout)
        specs[n + "/bias"] = (cout,)

    def bn(namer, c):
        n = namer("batch_normalization")
        specs[n + "/gamma"] = (c,)
        specs[n + "/beta"] = (c,)

    enc, dec = _Namer("encoder"), _Namer("decoder")
    if exp == "mnist_digit":            # codes/models.py:46-148
        conv(enc, ks, 1, nh // 16)
        conv(enc, ks, nh // 16, nh // 4)
        conv(enc, ks, nh // 4, nh)
        dens(enc, 16 * nh, nh // 4)
        dens(enc, nh // 4, Z, "code_mean")
        dens(enc, nh // 4, Z, "code_std_dev")
        dens(dec, Z, 16 * nh)
        conv(dec, 3, nh, nh)
        conv(dec, 3, nh // 4, nh // 4)
        conv(dec, 3, nh // 16, nh // 16)
        conv(dec, 5, nh // 64, 1)
    elif exp == "mnist_fashion":        # codes/models.py:199-315
        conv(enc, 3, 1, nh // 4)
        conv(enc, 3, nh // 4, nh // 4)
        conv(enc, 3, nh // 4, nh // 2)
        conv(enc, 3, nh // 2, nh // 2)
        dens(enc, 4 * (nh // 2), nh)
        dens(enc, nh, Z, "code_mean")
        dens(enc, nh, Z, "code_std_dev")
        dens(dec, Z, nh)
        conv(dec, 1, nh // 4, nh)
        conv(dec, 3, nh // 4, nh)
        conv(dec, 3, nh // 4, nh)
        conv(dec, 3, nh // 4, nh)
        conv(dec, 5, nh // 4, 1)
    elif exp == "celeba":               # codes/models.py:392-587
        cin = int(config["dim_input_channel"])
        for cout in (nh // 4, nh // 4, nh // 2, nh // 2, nh, nh):
            conv(enc, ks, cin, cout)
            bn(enc, cout)
            cin = cout
        dens(enc, 4 * nh, Z, "code_mean")
        dens(enc, 4 * nh, Z, "code_std_dev")
        dens(dec, Z, nh)
        for _ in range(8):
            dens(dec, nh, nh)
        conv(dec, 1, nh, nh)
        style_c = []
        for (k, ci, co, styled) in ((3, nh, nh, True), (3, nh, nh, True), (3, nh, nh, False),
                                    (3, nh, nh // 2, True), (3, nh // 2, nh // 2, False),
                                    (3, nh // 2, nh // 4, True), (3, nh // 4, nh // 4, False)):
            conv(dec, k, ci, co)
            if styled:
                i = len(style_c)
                style_c.append(co)
                specs["decoder/StyleMod_%d/dense/kernel" % i] = (nh, 2 * co)
                specs["decoder/StyleMod_%d/dense/bias" % i] = (2 * co,)
        conv(dec, 1, nh // 4, int(config["dim_input_channel"]))
    else:
        raise ValueError("unknown exp_name %r" % exp)
    specs["sigma/Variable"] = ()

    if config["prior"] in ("ours", "hierarchical"):     # codes/base.py:127-213
        H = int(config["num_hidden_units_inner_VAE"])
        R = int(config["representation_size"])
        nl = int(config["n_layers_inner_VAE"])
        pr = _Namer("prior")
        dens(pr, Z, H)
        for _ in range(nl - 1):
            dens(pr, H, H)
        dens(pr, H, R)
        dens(pr, H, R)
        dens(pr, R, H)
        for _ in range(nl - 1):
            dens(pr, H, H)
        dens(pr, H, Z)
        specs["inner_sigma/Variable"] = ()
    elif config["prior"] == "vampPrior":                 # codes/base.py:216-226: trainable pseudo-inputs, scope "prior"
        specs["prior/Variable"] = (int(config["n_mixtures"]), int(config["dim_input_x"]), int(config["dim_input_y"]),
                                   int(config["dim_input_channel"]))
    return specs


def init_params(config, seed=1):
    """Glorot-uniform kernels, zero biases, BN gamma=1/beta=0, sigma/inner_sigma from config.

    Draw order: lexicographic variable name, one numpy default_rng(seed).
    """
    specs = param_specs(config)
    rng = np.random.default_rng(seed)
    P = {}
    for name in sorted(specs):
        shp = specs[name]
        if name.endswith("/kernel"):
            if len(shp) == 4:
                fan_in, fan_out = shp[0] * shp[1] * shp[2], shp[0] * shp[1] * shp[3]
            else:
                fan_in, fan_out = shp
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            P[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
        elif name.endswith("/gamma"):
            P[name] = np.ones(shp, np.float32)
        elif name == "sigma/Variable":
            P[name] = np.float32(config["sigma"]).reshape(())
        elif name == "inner_sigma/Variable":
            P[name] = np.float32(config["inner_sigma"]).reshape(())
        elif name == "prior/Variable":                      # tf.random.normal (base.py:224)
            P[name] = rng.standard_normal(shp).astype(np.float32)
        else:
            P[name] = np.zeros(shp, np.float32)
    return P


def group_of(name):
    """Optimiser group of a variable (codes/base.py:415-430)."""
    if name.startswith("encoder/") or name.startswith("decoder/"):
        return "ae"
    if name.startswith("sigma/"):
        return "sigma"
    if name.startswith("prior/"):
        return "prior"
    if name.startswith("inner_sigma/"):
        return "inner_sigma"
    raise ValueError(name)


# --------------------------------------------------------------------------------------
# Networks
# --------------------------------------------------------------------------------------
def _cv(P, n, x, stride=1, padding="same", a=None):
    return act(conv2d_tf(x, P[n + "/kernel"], P[n + "/bias"], stride, padding), a)


def _dn(P, n, x, a=None):
    return act(dense(x, P[n + "/kernel"], P[n + "/bias"]), a)


def encoder(config, P, x, allreduce=None):
    """x [B,H,W,C] -> code_mean, code_std_dev [B,Z]."""
    exp = config["exp_name"]
    lvp = f32c(config["latent_variance_precision"], x.dtype)
    B = x.shape[0]
    if exp == "mnist_digit":            # codes/models.py:46-95
        h = pad_symmetric(x, 2)
        h = _cv(P, "encoder/conv2d", h, 2, "same", "leaky_relu")
        h = _cv(P, "encoder/conv2d_1", h, 2, "same", "leaky_relu")
        h = _cv(P, "encoder/conv2d_2", h, 2, "same", "leaky_relu")
        h = _dn(P, "encoder/dense", h.reshape(B, -1), "leaky_relu")
    elif exp == "mnist_fashion":        # codes/models.py:199-253
        h = pad_symmetric(x, 2)
        h = _cv(P, "encoder/conv2d", h, 2, "same", "leaky_relu")
        h = _cv(P, "encoder/conv2d_1", h, 2, "same", "leaky_relu")
        h = _cv(P, "encoder/conv2d_2", h, 2, "same", "leaky_relu")
        h = _cv(P, "encoder/conv2d_3", h, 1, "valid", "leaky_relu")
        h = _dn(P, "encoder/dense", h.reshape(B, -1), "leaky_relu")
    else:                               # codes/models.py:392-488
        h = x
        for i in range(6):
            n = _tfname("encoder/conv2d", i)
            bnn = _tfname("encoder/batch_normalization", i)
            h = _cv(P, n, h, 2 if i < 5 else 1, "same" if i < 5 else "valid", None)
            h = leaky_relu(batch_norm_train(h, P[bnn + "/gamma"], P[bnn + "/beta"], BN_EPS, allreduce))
        h = h.reshape(B, -1)
    mean = _dn(P, "encoder/code_mean", h, None)
    std = _dn(P, "encoder/code_std_dev", h, "relu") + lvp
    return mean, std


def decoder(config, P, z):
    """z [B,Z] -> decoded [B,H,W,C]."""
    exp = config["exp_name"]
    nh = int(config["num_hidden_units"])
    B = z.shape[0]
    if exp == "mnist_digit":            # codes/models.py:106-148
        h = _dn(P, "decoder/dense", z, "leaky_relu").reshape(B, 1, 1, 16 * nh)
        h = depth_to_space(h, 4)
        h = depth_to_space(_cv(P, "decoder/conv2d", h, 1, "same", "leaky_relu"), 2)
        h = depth_to_space(_cv(P, "decoder/conv2d_1", h, 1, "same", "leaky_relu"), 2)
        h = depth_to_space(_cv(P, "decoder/conv2d_2", h, 1, "same", "leaky_relu"), 2)
        return _cv(P, "decoder/conv2d_3", h, 1, "valid", "relu")
    if exp == "mnist_fashion":          # codes/models.py:264-315
        h = _dn(P, "decoder/dense", z, "leaky_relu").reshape(B, 1, 1, nh)
        h = depth_to_space(h, 2)
        h = depth_to_space(_cv(P, "decoder/conv2d", h, 1, "same", "leaky_relu"), 2)
        h = depth_to_space(_cv(P, "decoder/conv2d_1", h, 1, "same", "leaky_relu"), 2)
        h = depth_to_space(_cv(P, "decoder/conv2d_2", h, 1, "same", "leaky_relu"), 2)
        h = depth_to_space(_cv(P, "decoder/conv2d_3", h, 1, "same", "leaky_relu"), 2)
        return _cv(P, "decoder/conv2d_4", h, 1, "valid", "relu")
    # celeba: codes/models.py:499-587
    encoded = _dn(P, "decoder/dense", z, "leaky_relu")
    d = encoded
    for i in range(1, 9):
        d = _dn(P, "decoder/dense_%d" % i, d, "leaky_relu")
    dlatent = d

    def styled(h, conv, si):
        h = instance_norm(_cv(P, conv, h, 1, "same", None))
        st = _dn(P, "decoder/StyleMod_%d/dense" % si, dlatent, None)
        return leaky_relu(style_mod(h, st))

    h = _cv(P, "decoder/conv2d", encoded.reshape(B, 1, 1, nh), 1, "same", None)
    h = resize_bilinear_legacy(h, 2, 2)
    h = styled(h, "decoder/conv2d_1", 0)
    h = styled(h, "decoder/conv2d_2", 1)
    h = resize_bilinear_legacy(h, 8, 8)
    h = _cv(P, "decoder/conv2d_3", h, 1, "same", "leaky_relu")
    h = resize_bilinear_legacy(h, 16, 16)
    h = styled(h, "decoder/conv2d_4", 2)
    h = resize_bilinear_legacy(h, 32, 32)
    h = _cv(P, "decoder/conv2d_5", h, 1, "same", "leaky_relu")
    h = resize_bilinear_legacy(h, 64, 64)
    h = styled(h, "decoder/conv2d_6", 3)
    h = resize_bilinear_legacy(h, 128, 128)
    h = _cv(P, "decoder/conv2d_7", h, 1, "same", "leaky_relu")
    h = resize_bilinear_legacy(h, 128, 128)
    return _cv(P, "decoder/conv2d_8", h, 1, "same", None)


def inner_encoder(config, P, z):
    """codes/base.py:141-162: z -> representation_mean, representation_std_dev."""
    a = config["inner_activation"]
    nl = int(config["n_layers_inner_VAE"])
    h = z
    for i in range(nl):
        h = _dn(P, _tfname("prior/dense", i), h, a)
    mean = _dn(P, _tfname("prior/dense", nl), h, None)
    std = _dn(P, _tfname("prior/dense", nl + 1), h, "relu") + f32c(config["latent_variance_precision"], z.dtype)
    return mean, std


def inner_decoder(config, P, t):
    """codes/base.py:171-186: t -> decoded_code."""
    a = config["inner_activation"]
    nl = int(config["n_layers_inner_VAE"])
    h = t
    for i in range(nl):
        h = _dn(P, _tfname("prior/dense", nl + 2 + i), h, a)
    return _dn(P, _tfname("prior/dense", 2 * nl + 2), h, None)


# --------------------------------------------------------------------------------------
# Forward graph + ELBO  (codes/base.py:257-413, codes/models.py:152-159,319-326,591-597)
# --------------------------------------------------------------------------------------
def sg_feed(config, dtype=torch.float64):
    """Dummy mixture fed during SG pre-training (codes/base.py:870-876)."""
    K, R = int(config["n_mixtures"]), int(config["representation_size"])
    return dict(weights=torch.full((K,), 1.0 / K, dtype=dtype),
                means=torch.zeros(K, R, dtype=dtype),
                covs=torch.eye(R, dtype=dtype).repeat(K, 1, 1))


def forward(config, P, x, eps_z, eps_t=None, eps_mc=None, gm=None,
            use_sg=True, use_mask=False, code_input=None, allreduce=None, global_batch=None, stat_allreduce=None):
    """Evaluate every tensor the step functions fetch.  All batch means are over
    `global_batch` samples (defaults to the local batch); with `allreduce` given,
    partial sums are summed over ranks first (data-parallel restatement).
    Two kinds of exchange: `allreduce` sums the scalar partials (C3) -- what follows is replicated scalar
    algebra, so its backward is the identity; `stat_allreduce` sums batch-norm statistics (C2) -- their
    consumers are the rank's own samples, so its backward must all-reduce the gradient as well."""
    dt = x.dtype
    B = x.shape[0]
    Bg = float(global_batch if global_batch is not None else B)
    ar = (lambda v: v) if allreduce is None else allreduce
    Z = int(config["code_size"])
    D = int(config["dim_input_x"]) * int(config["dim_input_y"]) * int(config["dim_input_channel"])
    out = {}
    mu_z, sd_z = encoder(config, P, x, stat_allreduce if stat_allreduce is not None else allreduce)
    z = mu_z + sd_z * eps_z                                     # models.py:97-103
    out.update(code_mean=mu_z, code_std_dev=sd_z, code_sample=z)
    dec_in = z if code_input is None else code_input           # models.py:107,265,500
    xhat = decoder(config, P, dec_in)
    out["decoded"] = xhat

    # sigma block
    sig_v = P["sigma/Variable"]
    sigma = torch.sqrt(sig_v * sig_v)
    abs_sum = ar((xhat - x).abs().sum())
    mpe = abs_sum / (Bg * D)
    if config["exp_name"] == "celeba" or int(config["TRAIN_sigma"]) == 1:
        sigma = tf_maximum(sigma, mpe)
    out.update(sigma=sigma, mean_pixel_error=mpe)

    out["std_dev_code"] = ar(sd_z.sum(0)) / Bg
    out["entropy_z"] = ar((-0.5 * Z * LOG_2PI - 0.5 * Z - 0.5 * (2.0 * torch.log(sd_z)).sum(1)).sum()) / Bg
    out["crossEntropy_prior_sg"] = ar((-0.5 * Z * LOG_2PI
                                       - 0.5 * ((mu_z ** 2).sum(1) + (sd_z ** 2).sum(1))).sum()) / Bg

    prior = config["prior"]
    if prior == "standard_gaussian":
        out["crossEntropy_prior"] = out["crossEntropy_prior_sg"]
    elif prior == "vampPrior":                                   # base.py:216-254, 361-370
        # the encoder (shared weights, batch statistics of the K pseudo-inputs themselves) maps the trainable pseudo-inputs to the
        # components of an equally weighted diagonal-Gaussian mixture on z; pseudo-inputs are replicated under data parallelism
        mu_p, sd_p = encoder(config, P, P["prior/Variable"], None)
        K = mu_p.shape[0]
        L = eps_mc.shape[0]
        z_mc = (mu_z.unsqueeze(0) + sd_z.unsqueeze(0) * eps_mc).unsqueeze(2)          # [L,B,1,Z]
        u = (z_mc - mu_p) / sd_p                                                       # [L,B,K,Z]
        lp = -0.5 * (u ** 2).sum(-1) - torch.log(sd_p).sum(-1) - 0.5 * Z * LOG_2PI - math.log(K)
        vamp = ar(torch.logsumexp(lp, dim=-1).sum()) / (L * Bg)
        out["crossEntropy_prior_vamp"] = vamp
        out["crossEntropy_prior"] = out["crossEntropy_prior_sg"] if use_sg else vamp
    elif prior == "GMM":                                         # base.py:322-329: MC mean of the mixture log-prob of z samples
        L = eps_mc.shape[0]
        z_mc = mu_z.unsqueeze(0) + sd_z.unsqueeze(0) * eps_mc
        lp = gmm_log_prob(z_mc, gm["weights"], gm["means"], gm["covs"])
        out["crossEntropy_prior"] = ar(lp.sum()) / (L * Bg)
    elif prior in ("ours", "hierarchical"):
        hier = prior == "hierarchical"
        R = int(config["representation_size"])
        mu_t, sd_t = inner_encoder(config, P, z)
        t = mu_t + sd_t * eps_t                                  # base.py:164-167
        zhat = inner_decoder(config, P, t)
        iv = P["inner_sigma/Variable"]
        inner_sigma = torch.sqrt(iv * iv)
        if int(config["TRAIN_inner_sigma"]) == 1:                # base.py:210-212
            inner_sigma = tf_minimum(tf_maximum(inner_sigma, f32c(config["inner_sigma_lb"], dt)),
                                     f32c(config["inner_sigma_ub"], dt))
        out.update(representation_mean=mu_t, representation_std_dev=sd_t, representation_sample=t,
                   decoded_code=zhat, inner_sigma=inner_sigma)
        out["mean_code_error"] = ar((zhat - z).abs().sum()) / (Bg * Z)
        out["std_dev_representation"] = ar(sd_t.sum(0)) / Bg
        err = (z - zhat) ** 2                                    # base.py:286-297
        if use_mask and not hier:                                # the hierarchical branch never masks (base.py:334)
            err = torch.where(sd_z > 1.0, torch.zeros_like(err), err)
        out["code_reconstruction_likelihood"] = -ar((err / (2.0 * inner_sigma ** 2)).sum()) / Bg
        out["code_l1_reconstruction_error"] = ar(torch.sqrt(err).sum()) / Bg
        out["representation_regularisor"] = -Z * torch.log(inner_sigma) - 0.5 * Z * LOG_2PI
        Re = 2 if hier else R                                    # base.py:346-347 hard-codes "2" in the hierarchical entropy
        out["entropy_t"] = ar((-0.5 * Re * LOG_2PI - 0.5 * Re - 0.5 * (2.0 * torch.log(sd_t)).sum(1)).sum()) / Bg
        if hier:                                                 # base.py:350-353: closed form against N(0, I)
            out["crossEntropy_representation"] = ar((-0.5 * R * LOG_2PI
                                                     - 0.5 * ((mu_t ** 2).sum(1) + (sd_t ** 2).sum(1))).sum()) / Bg
        else:
            L = eps_mc.shape[0]
            t_mc = mu_t.unsqueeze(0) + sd_t.unsqueeze(0) * eps_mc    # base.py:308-311
            lp = gmm_log_prob(t_mc, gm["weights"], gm["means"], gm["covs"])
            out["crossEntropy_representation"] = ar(lp.sum()) / (L * Bg)
        out["elbo_prior"] = (out["code_reconstruction_likelihood"] + out["representation_regularisor"]
                             - out["entropy_t"] + out["crossEntropy_representation"])
        out["crossEntropy_prior"] = out["crossEntropy_prior_sg"] if use_sg else out["elbo_prior"]
        out["loss_prior"] = -out["elbo_prior"]
    else:
        raise NotImplementedError("unknown prior %r" % prior)

    diff = x - xhat                                             # base.py:374-396
    out["l2_reconstruction_error"] = ar((diff ** 2).sum()) / Bg
    out["l1_reconstruction_error"] = ar(torch.sqrt(diff ** 2).sum()) / Bg
    out["reconstruction_likelihood"] = -(ar(diff.abs().sum()) / Bg) / sigma
    out["sigma_regularisor"] = -D * torch.log(2.0 * sigma)
    out["elbo"] = (out["reconstruction_likelihood"] + out["sigma_regularisor"]
                   - out["entropy_z"] + out["crossEntropy_prior"])
    out["loss_ae"] = -out["elbo"]
    if prior == "vampPrior":
        out["loss_prior"] = -out["elbo"]                         # base.py:408-409
    return out


# --------------------------------------------------------------------------------------
# Optimiser + the four sess.run's of one iteration (codes/base.py:457-517, 583-641)
# --------------------------------------------------------------------------------------
def adam_tf(theta, g, m, v, t, lr):
    """tf.train.AdamOptimizer(beta1=.9, beta2=.95) update with element-wise clip to [-1,1]."""
    g = np.clip(g, -1.0, 1.0)
    m[...] = ADAM_B1 * m + (1 - ADAM_B1) * g
    v[...] = ADAM_B2 * v + (1 - ADAM_B2) * g * g
    lr_t = lr * math.sqrt(1 - ADAM_B2 ** t) / (1 - ADAM_B1 ** t)
    theta[...] = theta - lr_t * m / (np.sqrt(v) + ADAM_EPS)


class OracleState:
    """Parameters + the four optimisers' slots (numpy, dtype float64 or float32)."""

    def __init__(self, config, params, dtype=np.float64):
        self.config = config
        self.dtype = dtype
        self.P = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.t = dict(ae=0, sigma=0, prior=0, inner_sigma=0)

    def torch_params(self, grad_groups=()):
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        P = {}
        for k, v in self.P.items():
            tt = torch.tensor(v, dtype=tdt)
            if group_of(k) in grad_groups:
                tt.requires_grad_(True)
            P[k] = tt
        return P

    def apply(self, group, grads, lr):
        self.t[group] += 1
        for k, g in grads.items():
            adam_tf(self.P[k], g.astype(self.dtype), self.m[k], self.v[k], self.t[group], lr)


_TO = lambda a, dt: None if a is None else torch.as_tensor(np.asarray(a), dtype=dt)


def _gm_t(gm, dt):
    return None if gm is None else {k: _TO(v, dt) for k, v in gm.items()}


def run(state, x, noise, gm, use_sg, use_mask, train=None, lr=0.0, allreduce=None,
        global_batch=None, grad_allreduce=None, stat_allreduce=None):
    """One `sess.run`: forward with `noise` = dict(eps_z, eps_t, eps_mc); if `train` names an
    optimiser group, differentiate its loss w.r.t. that group, (all-reduce,) clip, Adam.
    Returns {fetch: numpy}."""
    cfg = state.config
    tdt = torch.float64 if state.dtype == np.float64 else torch.float32
    P = state.torch_params((train,) if train else ())
    out = forward(cfg, P, _TO(x, tdt), _TO(noise["eps_z"], tdt), _TO(noise.get("eps_t"), tdt),
                  _TO(noise.get("eps_mc"), tdt), _gm_t(gm, tdt), use_sg, use_mask,
                  allreduce=allreduce, global_batch=global_batch, stat_allreduce=stat_allreduce)
    grads = None
    if train:
        loss = out["loss_ae"] if train in ("ae", "sigma") else out["loss_prior"]
        names = [k for k in P if group_of(k) == train]
        gs = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
        grads = {}
        for k, g in zip(names, gs):
            g = torch.zeros_like(P[k]) if g is None else g
            if grad_allreduce is not None:
                g = grad_allreduce(g)
            grads[k] = g.detach().numpy()
        state.apply(train, grads, lr)
    res = {k: v.detach().numpy() for k, v in out.items()}
    if grads is not None:
        res["_grads"] = grads
    return res


def train_iteration(state, x, noises, gm, cur_epoch, lr_ae, allreduce=None, global_batch=None,
                    grad_allreduce=None):
    """The reference's per-minibatch 4-run structure (codes/trainers.py:33-40,148-155;
    codes/base.py:583-641).  `noises` = list of 4 noise dicts (one per sess.run)."""
    cfg = state.config
    use_sg = cur_epoch <= int(cfg["sg_pretraining"])           # base.py:868
    use_mask = cur_epoch >= int(cfg["use_mask_start"])         # base.py:896
    kw = dict(allreduce=allreduce, global_batch=global_batch, grad_allreduce=grad_allreduce)
    fetch = {}
    if int(cfg["TRAIN_VAE"]) == 1:
        fetch["run1"] = run(state, x, noises[0], gm, use_sg, use_mask, "ae", lr_ae, **kw)
        if int(cfg["TRAIN_sigma"]) == 1:
            lr_s = float(cfg["learning_rate_sigma"]) * (0.99 ** (cur_epoch - 1))
            fetch["run2"] = run(state, x, noises[1], gm, use_sg, use_mask, "sigma", lr_s, **kw)
    if (cur_epoch > int(cfg["sg_pretraining"]) - 1 and cfg["prior"] in ("ours", "hierarchical")
            and int(cfg["TRAIN_prior"]) == 1):
        lr_p = float(cfg["learning_rate_prior"]) * (1.01 ** (cur_epoch - 1))
        fetch["run3"] = run(state, x, noises[2], gm, use_sg, use_mask, "prior", lr_p, **kw)
        if int(cfg["TRAIN_inner_sigma"]) == 1:
            lr_i = float(cfg["learning_rate_inner_sigma"]) * (1.01 ** (cur_epoch - 1))
            fetch["run4"] = run(state, x, noises[3], gm, use_sg, use_mask, "inner_sigma", lr_i, **kw)
    return fetch


def make_noise(config, B, rng, dtype=np.float64):
    Z, R, L = int(config["code_size"]), int(config.get("representation_size", 1)), int(config["n_MC_samples"])
    Rmc = Z if config.get("prior") in ("GMM", "vampPrior") else R   # these priors sample z, not t (base.py:324-327, 363-366)
    return dict(eps_z=rng.standard_normal((B, Z)).astype(dtype),
                eps_t=rng.standard_normal((B, R)).astype(dtype),
                eps_mc=rng.standard_normal((L, B, Rmc)).astype(dtype))


def synthetic_gm(config, rng=None, fixture=None):
    """Mixture used by benches/tests: R=2 -> first K rows of the reference's fitted
    mixture (renormalised); otherwise a seeded SPD recipe."""
    K, R = int(config["n_mixtures"]), int(config["representation_size"])
    if R == 2 and fixture is not None and K <= fixture["w_full"].shape[0]:
        w = np.asarray(fixture["w_full"][:K], np.float64)
        return dict(weights=w / w.sum(), means=np.asarray(fixture["m_full"][:K], np.float64),
                    covs=np.asarray(fixture["K_full"][:K], np.float64))
    rng = rng or np.random.default_rng(3)
    m = rng.normal(0.0, 1.5, size=(K, R))
    A = rng.normal(0.0, 0.3, size=(K, R, R))
    cov = A @ np.transpose(A, (0, 2, 1)) / R + 0.05 * np.eye(R)
    w = rng.dirichlet(np.ones(K))
    return dict(weights=w, means=m, covs=cov)
