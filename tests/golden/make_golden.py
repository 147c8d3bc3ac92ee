"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only
(it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs (all data, no reference source text):
  GM_prior_info.npz      copy of the reference's fitted 50-component mixture
                         (figures/mnist_digit/result/GM_prior_info.npz, written by codes/base.py:769-777)
  ckpt_inventory.json    {model: {ckpt: {variable_name: shape}}} parsed from the reference's
                         pretrained_models/*/*.index (TF bundle index = leveldb-format table of
                         BundleEntryProto), no TensorFlow needed
  ref_ckpt_index/*.index copies of the reference's six TensorFlow-written checkpoint index DATA files (1-3 KB each; variable names,
                         shapes, offsets and CRCs, no tensor data): tests/test_host_cpu.py re-serialises the parsed entries with
                         codes/tf_bundle.py and requires the identical bytes
  oracle_*.npz           inputs + float64 oracle outputs on tiny shapes (see make_oracle_vectors)
"""
import json
import os
import shutil
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------- TF bundle .index reader
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block_entries(buf, off, size):
    blk = buf[off:off + size]
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        yield key, blk[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("wire type %d" % wt)
        yield fn, wt, v


def read_bundle_index(path):
    """-> {name: dict(dtype, shape, offset, size)} from a TF checkpoint-v2 .index file."""
    buf = open(path, "rb").read()
    footer = buf[-48:]
    assert footer[-8:] == struct.pack("<Q", 0xdb4775248b80fb57), "not a leveldb-format table"
    pos = 0
    _, pos = _varint(footer, pos)
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    out = {}
    for _, handle in _block_entries(buf, idx_off, idx_size):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _block_entries(buf, boff, bsize):
            if key == b"":
                continue                      # BundleHeaderProto
            ent = dict(dtype=0, shape=[], offset=0, size=0)
            for fn, wt, v in _proto_fields(val):
                if fn == 1:
                    ent["dtype"] = v
                elif fn == 2:                 # TensorShapeProto
                    for f2, _, v2 in _proto_fields(v):
                        if f2 == 2:           # Dim
                            sz = 0
                            for f3, _, v3 in _proto_fields(v2):
                                if f3 == 1:
                                    sz = v3
                            ent["shape"].append(sz)
                elif fn == 4:
                    ent["offset"] = v
                elif fn == 5:
                    ent["size"] = v
            out[key.decode()] = ent
    return out


def make_ckpt_inventory():
    inv = {}
    for model in ("celeba", "mnist_digit", "mnist_fashion"):
        inv[model] = {}
        for ck in ("vae-model", "prior-model"):
            ents = read_bundle_index(os.path.join(REF, "pretrained_models", model, ck + ".index"))
            inv[model][ck] = {k: v["shape"] for k, v in sorted(ents.items())}
            tot = sum(int(np.prod(s)) if s else 1 for s in inv[model][ck].values())
            print(model, ck, len(ents), "tensors", tot, "elements")
    with open(os.path.join(HERE, "ckpt_inventory.json"), "w") as f:
        json.dump(inv, f, indent=1, sort_keys=True)


# ---------------------------------------------------------------- oracle vectors
def tiny_config(exp):
    cfg = dict(exp_name=exp, prior="ours", inner_activation="leaky_relu", n_mixtures=5, n_MC_samples=7,
               sg_pretraining=1, use_mask_start=100, kernel_size=3, learning_rate_ae=3e-4,
               learning_rate_sigma=5e-4, learning_rate_prior=3e-4, learning_rate_inner_sigma=2e-4,
               batch_size=4, code_size=8, representation_size=2, TRAIN_VAE=1, TRAIN_sigma=1, TRAIN_prior=1,
               TRAIN_inner_sigma=1, TRAIN_decoded_z_std=0, sigma=0.5, inner_sigma=0.1, inner_sigma_ub=0.1,
               inner_sigma_lb=0.05, latent_variance_precision=1e-3, num_hidden_units=64,
               num_hidden_units_inner_VAE=32, n_layers_inner_VAE=2, dim_input_x=28, dim_input_y=28,
               dim_input_channel=1)
    if exp == "celeba":
        cfg.update(dim_input_x=128, dim_input_y=128, dim_input_channel=3, num_hidden_units=32, batch_size=2)
    return cfg


def make_oracle_vectors():
    """Tiny end-to-end vectors: inputs, params (seeded), noise, mixture -> float64 oracle fetches for
    two consecutive iterations (epoch 2 > sg_pretraining=1, i.e. the 4-run regime) plus the
    post-iteration parameters.  tests/test_oracle_cpu.py re-derives them; tests/test_gpu_*.py
    compare the HIP path against them."""
    import torch
    from oracle import ladder_oracle as O
    torch.set_num_threads(8)
    fix = np.load(os.path.join(HERE, "GM_prior_info.npz"))
    for exp in ("mnist_digit", "mnist_fashion", "celeba"):
        cfg = tiny_config(exp)
        B = cfg["batch_size"]
        rng = np.random.default_rng(0)
        x = rng.random((B, cfg["dim_input_x"], cfg["dim_input_y"], cfg["dim_input_channel"])).astype(np.float32)
        P = O.init_params(cfg, seed=1)
        gm = {k: v.astype(np.float32) for k, v in O.synthetic_gm(cfg, fixture=fix).items()}   # fed as fp32 (base.py:110-112)
        st = O.OracleState(cfg, P, np.float64)
        nrng = np.random.default_rng(2)
        save = dict(x=x.astype(np.float32), config=json.dumps(cfg), gm_w=gm["weights"], gm_m=gm["means"], gm_c=gm["covs"])
        for it in range(2):
            noises = [O.make_noise(cfg, B, nrng, np.float32) for _ in range(4)]
            f = O.train_iteration(st, x, noises, gm, cur_epoch=2, lr_ae=cfg["learning_rate_ae"])
            for r, nz in enumerate(noises):
                for k, v in nz.items():
                    save["it%d_run%d_%s" % (it, r + 1, k)] = v.astype(np.float32)
            for rn, fe in f.items():
                for k, v in fe.items():
                    if k == "_grads":
                        continue
                    if np.ndim(v) == 0:
                        save["it%d_%s_%s" % (it, rn, k)] = np.asarray(v, np.float64)
                    elif k in ("code_sample", "representation_sample") or (k == "decoded" and it == 0 and rn == "run1"):
                        save["it%d_%s_%s" % (it, rn, k)] = np.asarray(v, np.float32)
        for k, v in st.P.items():
            save["final/" + k] = v.astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "oracle_%s.npz" % exp), **save)
        print(exp, "elbo it0:", save["it0_run1_elbo"], "it1:", save["it1_run1_elbo"])


def copy_index_fixtures():
    os.makedirs(os.path.join(HERE, "ref_ckpt_index"), exist_ok=True)
    for model in ("celeba", "mnist_digit", "mnist_fashion"):
        for ck in ("vae-model", "prior-model"):
            shutil.copyfile(os.path.join(REF, "pretrained_models", model, ck + ".index"),
                            os.path.join(HERE, "ref_ckpt_index", "%s_%s.index" % (model, ck)))


if __name__ == "__main__":
    shutil.copyfile(os.path.join(REF, "figures/mnist_digit/result/GM_prior_info.npz"),
                    os.path.join(HERE, "GM_prior_info.npz"))
    copy_index_fixtures()
    make_ckpt_inventory()
    make_oracle_vectors()
