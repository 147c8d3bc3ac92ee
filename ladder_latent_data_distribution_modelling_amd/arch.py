"""Architecture tables of the three reference models (variable names, shapes, layer wiring).

Follows codes/models.py:46-148 (MNIST digit), 199-315 (MNIST fashion), 392-587 (CelebA) and
codes/base.py:127-213 (inner VAE) of the reference; names reproduce TF's per-scope unique layer
naming so checkpoints keep the reference's variable names.
"""
from collections import OrderedDict
import math

import numpy as np


def same_pad(n_in, k, s):
    """TF 'SAME': out = ceil(in/s); pad_total = max((out-1)s + k - in, 0); before = total//2."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, out


def conv_out(n_in, k, s, padding):
    if padding == "same":
        return same_pad(n_in, k, s)
    return 0, (n_in - k) // s + 1


_RESIZE_GAIN = {}


def resize_transpose_gain(n_in, n_out):
    """Largest column sum of the 1-D TF1-legacy bilinear interpolation matrix n_in -> n_out (tf.image.resize_images without
    half-pixel centres: src = y * n_in / n_out, neighbours floor(src) and min(floor(src) + 1, n_in - 1)).  The transpose of the resize
    multiplies max|dy| by at most this factor per axis; the LAST source row also collects the clamped outputs, so for an integer
    factor f the maximum is 1.5 f - 0.5 (2.5 for f = 2), not f."""
    key = (int(n_in), int(n_out))
    g = _RESIZE_GAIN.get(key)
    if g is None:
        col = np.zeros(key[0])
        src = np.arange(key[1], dtype=np.float64) * (key[0] / key[1])
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, key[0] - 1)
        fr = src - i0
        np.add.at(col, i0, 1.0 - fr)
        np.add.at(col, i1, fr)
        g = _RESIZE_GAIN[key] = float(col.max())
    return g


def tfname(base, i):
    return base if i == 0 else "%s_%d" % (base, i)


def encoder_convs(cfg):
    """[(cin, cout, k, stride, padding, act, bn)] of the encoder conv stack."""
    exp, nh = cfg["exp_name"], int(cfg["num_hidden_units"])
    ks = int(cfg.get("kernel_size", 3))
    if exp == "mnist_digit":
        return [(1, nh // 16, ks, 2, "same", "leaky_relu", False), (nh // 16, nh // 4, ks, 2, "same", "leaky_relu", False),
                (nh // 4, nh, ks, 2, "same", "leaky_relu", False)]
    if exp == "mnist_fashion":
        return [(1, nh // 4, 3, 2, "same", "leaky_relu", False), (nh // 4, nh // 4, 3, 2, "same", "leaky_relu", False),
                (nh // 4, nh // 2, 3, 2, "same", "leaky_relu", False), (nh // 2, nh // 2, 3, 1, "valid", "leaky_relu", False)]
    if exp == "celeba":
        c0 = int(cfg["dim_input_channel"])
        chans = [c0, nh // 4, nh // 4, nh // 2, nh // 2, nh, nh]
        return [(chans[i], chans[i + 1], ks, 2 if i < 5 else 1, "same" if i < 5 else "valid", None, True) for i in range(6)]
    raise ValueError("unknown exp_name %r" % exp)


def encoder_flat_dim(cfg):
    exp, nh = cfg["exp_name"], int(cfg["num_hidden_units"])
    return {"mnist_digit": 16 * nh, "mnist_fashion": 2 * nh, "celeba": 4 * nh}[exp]


def encoder_hidden(cfg):
    """Width of the encoder's dense layer before the heads (None: heads read the flattened conv output)."""
    exp, nh = cfg["exp_name"], int(cfg["num_hidden_units"])
    return {"mnist_digit": nh // 4, "mnist_fashion": nh, "celeba": None}[exp]


def celeba_decoder_convs(cfg):
    """[(k, cin, cout, styled, act, resize_to)] for decoder/conv2d_1..7 (conv2d and conv2d_8 are the 1x1s)."""
    nh = int(cfg["num_hidden_units"])
    return [(3, nh, nh, True, None, None), (3, nh, nh, True, None, 8), (3, nh, nh, False, "leaky_relu", 16),
            (3, nh, nh // 2, True, None, 32), (3, nh // 2, nh // 2, False, "leaky_relu", 64),
            (3, nh // 2, nh // 4, True, None, 128), (3, nh // 4, nh // 4, False, "leaky_relu", 128)]


def mnist_decoder_convs(cfg):
    """[(k, cin, cout, padding, act, d2s_after)]; preceded by dense + reshape + depth_to_space(first_r)."""
    exp, nh = cfg["exp_name"], int(cfg["num_hidden_units"])
    if exp == "mnist_digit":
        return 16 * nh, 4, [(3, nh, nh, "same", "leaky_relu", 2), (3, nh // 4, nh // 4, "same", "leaky_relu", 2),
                            (3, nh // 16, nh // 16, "same", "leaky_relu", 2), (5, nh // 64, 1, "valid", "relu", 0)]
    return nh, 2, [(1, nh // 4, nh, "same", "leaky_relu", 2), (3, nh // 4, nh, "same", "leaky_relu", 2),
                   (3, nh // 4, nh, "same", "leaky_relu", 2), (3, nh // 4, nh, "same", "leaky_relu", 2),
                   (5, nh // 4, 1, "valid", "relu", 0)]


def param_specs(cfg):
    """Ordered {tf_variable_name: shape}."""
    exp, nh, Z = cfg["exp_name"], int(cfg["num_hidden_units"]), int(cfg["code_size"])
    specs = OrderedDict()

    def add_conv(scope, i, k, cin, cout):
        n = "%s/%s" % (scope, tfname("conv2d", i))
        specs[n + "/kernel"] = (k, k, cin, cout)
        specs[n + "/bias"] = (cout,)

    def add_dense(name, cin, cout):
        specs[name + "/kernel"] = (cin, cout)
        specs[name + "/bias"] = (cout,)

    for i, (cin, cout, k, _s, _p, _a, bn) in enumerate(encoder_convs(cfg)):
        add_conv("encoder", i, k, cin, cout)
        if bn:
            b = "encoder/" + tfname("batch_normalization", i)
            specs[b + "/gamma"] = (cout,)
            specs[b + "/beta"] = (cout,)
    hid = encoder_hidden(cfg)
    feat = encoder_flat_dim(cfg)
    if hid is not None:
        add_dense("encoder/dense", feat, hid)
        feat = hid
    add_dense("encoder/code_mean", feat, Z)
    add_dense("encoder/code_std_dev", feat, Z)

    if exp == "celeba":
        add_dense("decoder/dense", Z, nh)
        for i in range(1, 9):
            add_dense("decoder/dense_%d" % i, nh, nh)
        add_conv("decoder", 0, 1, nh, nh)
        si = 0
        for i, (k, ci, co, styled, _a, _r) in enumerate(celeba_decoder_convs(cfg)):
            add_conv("decoder", i + 1, k, ci, co)
            if styled:
                add_dense("decoder/StyleMod_%d/dense" % si, nh, 2 * co)
                si += 1
        add_conv("decoder", 8, 1, nh // 4, int(cfg["dim_input_channel"]))
    else:
        width, _r0, convs = mnist_decoder_convs(cfg)
        add_dense("decoder/dense", Z, width)
        for i, (k, ci, co, _p, _a, _d) in enumerate(convs):
            add_conv("decoder", i, k, ci, co)
    specs["sigma/Variable"] = ()
    if cfg["prior"] in ("ours", "hierarchical"):
        H, R, nl = int(cfg["num_hidden_units_inner_VAE"]), int(cfg["representation_size"]), int(cfg["n_layers_inner_VAE"])
        dims = [(Z, H)] + [(H, H)] * (nl - 1) + [(H, R), (H, R), (R, H)] + [(H, H)] * (nl - 1) + [(H, Z)]
        for i, (a, b) in enumerate(dims):
            add_dense("prior/" + tfname("dense", i), a, b)
        specs["inner_sigma/Variable"] = ()
    elif cfg["prior"] == "vampPrior":                     # trainable pseudo-inputs (codes/base.py:216-226), scope "prior"
        specs["prior/Variable"] = (int(cfg["n_mixtures"]), int(cfg["dim_input_x"]), int(cfg["dim_input_y"]), int(cfg["dim_input_channel"]))
    return specs


def group_of(name):
    """Optimiser group (codes/base.py:415-430)."""
    for prefix, g in (("encoder/", "ae"), ("decoder/", "ae"), ("sigma/", "sigma"), ("prior/", "prior"),
                      ("inner_sigma/", "inner_sigma")):
        if name.startswith(prefix):
            return g
    raise ValueError(name)


def init_values(cfg, seed=1):
    """Glorot-uniform kernels (tf xavier_initializer / tf.layers default), zero biases, gamma=1, beta=0,
    sigma / inner_sigma from the config.  Draw order: lexicographic variable name from one
    numpy default_rng(seed) (the synthetic-input recipe of SURVEY 8d)."""
    specs = param_specs(cfg)
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(specs):
        shp = specs[name]
        if name.endswith("/kernel"):
            rf = int(np.prod(shp[:-2])) if len(shp) == 4 else 1
            lim = math.sqrt(6.0 / (rf * shp[-2] + rf * shp[-1]))
            out[name] = rng.uniform(-lim, lim, size=shp).astype(np.float32)
        elif name.endswith("/gamma"):
            out[name] = np.ones(shp, np.float32)
        elif name == "sigma/Variable":
            out[name] = np.asarray(cfg["sigma"], np.float32)
        elif name == "inner_sigma/Variable":
            out[name] = np.asarray(cfg["inner_sigma"], np.float32)
        elif name == "prior/Variable":                      # tf.random.normal (base.py:224)
            out[name] = rng.standard_normal(shp).astype(np.float32)
        else:
            out[name] = np.zeros(shp, np.float32)
    return out
