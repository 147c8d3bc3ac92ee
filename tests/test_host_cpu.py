"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared symbol, architecture tables
agree with the oracle and the reference checkpoints, config / data-loader behaviour, and the data-parallel exchange
scheme (C1-C3) verified with world_size-2 gloo processes on the oracle."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from ladder_latent_data_distribution_modelling_amd.csrc import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(lib_path):
    header = open(os.path.join(ROOT, "include", "ladder_hip.h")).read()
    declared = set(re.findall(r"\b(ladder_[a-z0-9_]+)\s*\(", header))
    from ladder_latent_data_distribution_modelling_amd import _lib
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name
    # no-compute calls are safe without a GPU
    lib.ladder_abi_version.restype = ctypes.c_int
    assert lib.ladder_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define LADDER_ABI_VERSION (\d+)", header).group(1))
    assert lib.ladder_gmm_packed_stride(2) == 6 and lib.ladder_gmm_packed_stride(8) == 45
    lib.ladder_conv2d_bwd_filter_workspace_bytes.restype = ctypes.c_size_t
    assert lib.ladder_conv2d_bwd_filter_workspace_bytes(128, 128, 128, 128, 128, 128, 128, 3, 3) > 0


def test_product_path_fails_loudly_without_gpu():
    import torch
    from ladder_latent_data_distribution_modelling_amd import _lib
    from ladder_latent_data_distribution_modelling_amd.engine import Ctx
    with pytest.raises(_lib.LadderHipError):
        Ctx("cpu")
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            Ctx("cuda:0").empty(4)
    with pytest.raises(_lib.LadderHipError):
        saved, _lib._lib = _lib._lib, None
        try:
            _lib.load("/nonexistent/libladder_hip.so")
        finally:
            _lib._lib = saved


def test_arch_tables_match_oracle_and_checkpoints(golden_dir):
    from ladder_latent_data_distribution_modelling_amd import arch
    from oracle import ladder_oracle as O
    inv = json.load(open(os.path.join(golden_dir, "ckpt_inventory.json")))
    for exp in ("mnist_digit", "mnist_fashion", "celeba"):
        cfg = json.load(open(os.path.join(ROOT, "codes", "%s_config.json" % exp)))
        assert list(arch.param_specs(cfg).items()) == list(O.param_specs(cfg).items())
        a, b = arch.init_values(cfg, 3), O.init_params(cfg, 3)
        assert all(np.array_equal(a[k], b[k]) for k in a)
    cfg = dict(exp_name="celeba", prior="ours", kernel_size=3, num_hidden_units=512, code_size=256, representation_size=32,
               dim_input_channel=3, num_hidden_units_inner_VAE=512, n_layers_inner_VAE=5)
    got = {k: list(v) for k, v in arch.param_specs(cfg).items()}
    ref = dict(inv["celeba"]["vae-model"], **inv["celeba"]["prior-model"])
    assert got == ref
    # BASELINE config 3 parameter counts
    cfg = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
    n = lambda pre: sum(int(np.prod(s)) if s else 1 for k, s in arch.param_specs(cfg).items() if k.startswith(pre))
    assert n("encoder/") + n("decoder/") == 17976451 and n("prior/") == 2170948
    assert arch.same_pad(128, 3, 2) == (0, 64) and arch.conv_out(4, 3, 1, "valid") == (0, 2)


def test_process_config_and_dirs(tmp_path, monkeypatch):
    from ladder_latent_data_distribution_modelling_amd.codes import utils
    monkeypatch.chdir(tmp_path)
    cfg = utils.process_config(os.path.join(ROOT, "codes", "mnist_digit_config.json"))
    assert cfg["result_dir"] == "./experiments/mnist_digit/batch-128/prior-ours-256-8-2-leaky_relu-5-mixture-10/result/"
    assert cfg["checkpoint_dir"].endswith("/checkpoint/") and cfg["summary_dir"].endswith("/summary/")
    assert utils.create_dirs([cfg["result_dir"], cfg["checkpoint_dir"]]) == 0
    utils.save_config(cfg)
    assert any(f.startswith("training_config_") for f in os.listdir(cfg["checkpoint_dir"]))
    j = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
    assert (j["code_size"], j["n_mixtures"], j["batch_size"], j["num_hidden_units"]) == (64, 30, 128, 512)   # BASELINE configs[2]


def test_data_loader_mnist_synthetic_and_tfrecords(tmp_path):
    from ladder_latent_data_distribution_modelling_amd.codes import data_loader as dl
    cfg = dict(exp_name="mnist_digit", batch_size=64, data_path=str(tmp_path))
    d = dl.DataGenerator(cfg, None)
    assert d.synthetic and d.test_set["image"].shape == (64, 28, 28, 1)
    counts = np.bincount(d.test_set["attrib"], minlength=10)
    assert tuple(counts) == (7, 7, 7, 7, 6, 6, 6, 6, 6, 6)          # class-balanced test batch, data_loader.py:37-44
    with pytest.raises(ValueError):
        dl.DataGenerator(dict(exp_name="mnist_digit", batch_size=100, data_path=str(tmp_path)), None)
    it = dl.BatchIterator(d.train_set["image"], 64, seed=1)
    assert it.next().shape == (64, 28, 28, 1) and it.next().dtype == np.float32
    with pytest.raises(ValueError, match="smaller than one minibatch"):
        dl.BatchIterator(d.train_set["image"][:10], 64)
    # CelebA TFRecord round trip (tf.Example, bytes feature 'X' = raw uint8 HWC, models.py:354-371)
    imgs = np.random.default_rng(0).integers(0, 256, (5, 8, 8, 3), dtype=np.uint8)
    dl.write_tfrecord(str(tmp_path / "celebA_train.tfrecords"), imgs)
    c = dl.DataGenerator(dict(exp_name="celeba", batch_size=2, data_path=str(tmp_path) + "/", dim_input_x=8, dim_input_y=8, dim_input_channel=3), None)
    arr = c.celeba_images("train")
    assert arr.shape == (5, 8, 8, 3) and np.array_equal(arr, imgs.astype(np.float32) * np.float32(1 / 255))
    assert (c.n_train, c.n_val) == (180000, 20000)


def test_draw_ellipse_geometry():
    """BaseTrain.draw_ellipse (reference codes/base.py:825-841): 2-sigma principal axes and angle of a full covariance, axis-aligned
    ellipse for a diagonal one, line width 10 x weight -- host-only helper of the notebook's prior plots."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from ladder_latent_data_distribution_modelling_amd.codes.base import BaseTrain
    t = BaseTrain.__new__(BaseTrain)
    _, ax = plt.subplots()
    th = np.radians(30.0)
    Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    cov = Rm @ np.diag([4.0, 0.25]) @ Rm.T
    e = t.draw_ellipse((1.0, -2.0), cov, 0.3, ax=ax, color="b")
    assert e in ax.patches and tuple(e.center) == (1.0, -2.0)
    assert abs(e.width - 2 * 2 * 2.0) < 1e-9 and abs(e.height - 2 * 2 * 0.5) < 1e-9       # nsig x 2 sqrt(eigenvalue)
    assert abs(((e.angle - 30.0) + 90.0) % 180.0 - 90.0) < 1e-6 and abs(e.get_linewidth() - 3.0) < 1e-12 and not e.get_fill()
    d = t.draw_ellipse((0.0, 0.0), np.array([9.0, 1.0]), 0.1, ax=ax)
    assert (d.width, d.height, d.angle) == (12.0, 4.0, 0.0)
    plt.close("all")


def test_trainer_schedules():
    from ladder_latent_data_distribution_modelling_amd.codes.trainers import CelebATrainer_joint_training as T
    t = T.__new__(T)
    t.config = dict(learning_rate_ae=1.0)
    for e, want in ((1, 1.0), (25, 0.99 ** 24), (26, 0.5 * 0.99), (51, 0.2 * 0.99), (76, 0.1 * 0.99)):
        t.cur_epoch = e
        t.compute_cur_lr()
        assert abs(t.cur_lr - want) < 1e-12
    # mid-epoch check points (reference trainers.py:139,156-158): the hook fires behind exactly those iterations
    import numpy as np
    calls = []
    t.config, t.n_train_iter = dict(num_iter_to_plot=2), 1406
    t.idx_check_point = np.arange(0, t.n_train_iter - 1, t.n_train_iter // 2)
    t.test_batch = "tb"
    t.test_step = lambda batch_data, print_result: calls.append((batch_data, print_result))
    for i in range(t.n_train_iter):
        t._mid_epoch(i)
    assert list(t.idx_check_point) == [0, 703] and calls == [("tb", False)] * 2
    t.config = dict(num_iter_to_plot=1)
    t._mid_epoch(0)
    assert len(calls) == 2


DP_WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from oracle import ladder_oracle as O
WORLD = %(world)d
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=WORLD)
rank = dist.get_rank()
d = np.load(os.path.join(%(root)r, "tests", "golden", "oracle_celeba.npz"))
cfg = json.loads(str(d["config"]))
cfg["batch_size"] = 4
rng = np.random.default_rng(0)
x = rng.random((4, 128, 128, 3)).astype(np.float32)
noise = O.make_noise(cfg, 4, rng, np.float32)
gm = dict(weights=d["gm_w"], means=d["gm_m"], covs=d["gm_c"])
P = O.init_params(cfg, seed=1)
per = 4 // WORLD
sl = slice(per * rank, per * rank + per)
nz = dict(eps_z=noise["eps_z"][sl], eps_t=noise["eps_t"][sl], eps_mc=noise["eps_mc"][:, sl])
def summed(t):
    t = t.detach().clone()
    dist.all_reduce(t)
    return t
def ar_scalar(t):            # C3: replicated scalar algebra follows -> identity backward
    return t + (summed(t) - t.detach())
class _StatAR(torch.autograd.Function):     # C2: per-rank consumers follow -> the gradient is all-reduced too
    @staticmethod
    def forward(ctx, t):
        return summed(t)
    @staticmethod
    def backward(ctx, g):
        return summed(g)
st = O.OracleState(cfg, P, np.float64)
# shard: BN statistics + scalar partials all-reduced (C2, C3), gradients summed (C1); every mean over the GLOBAL batch
res = O.run(st, x[sl], nz, gm, False, False, train="ae", lr=1e-3, allreduce=ar_scalar, global_batch=4,
            grad_allreduce=summed, stat_allreduce=_StatAR.apply)
if rank == 0:
    st1 = O.OracleState(cfg, P, np.float64)
    ref = O.run(st1, x, noise, gm, False, False, train="ae", lr=1e-3)
    for k in ("elbo", "elbo_prior", "sigma", "entropy_z", "l1_reconstruction_error", "crossEntropy_representation"):
        assert abs(float(res[k]) - float(ref[k])) < 1e-9 * max(1.0, abs(float(ref[k]))), k
    for k, g in ref["_grads"].items():
        assert np.abs(res["_grads"][k] - g).max() < 1e-9 * np.abs(g).max() + 1e-9, k
    for k in st1.P:
        if k in ref["_grads"] and np.abs(ref["_grads"][k]).max() < 1e-6:
            continue       # bias absorbed by a norm layer: exactly-zero true gradient, Adam's sign-like step amplifies 1e-17 noise
        assert np.abs(st.P[k] - st1.P[k]).max() < 1e-6, k   # lr=1e-3; Adam amplifies 1e-12 gradient noise on ~0 gradients to ~1e-8
    print("DP_OK")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_scheme_world2_gloo(tmp_path, world):
    """2 (4) gloo ranks, each with half (a quarter: ONE image -- per-rank batch statistics would be degenerate) of the batch: all-reduced BN
    statistics (C2), scalar partials (C3) and summed gradients of the global-mean loss (C1, clip AFTER the reduction) reproduce the
    single-process global-batch step."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % dict(root=ROOT, port=29500 + (os.getpid() + 7 * world) % 2000, world=world))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP_OK" in outs[0]


def test_comm_wrapper_world2_gloo(tmp_path):
    """The product's Comm class (engine.Comm) on a gloo group: sum all-reduce and broadcast semantics."""
    code = r'''
import sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=int(sys.argv[1]), world_size=2)
from ladder_latent_data_distribution_modelling_amd.engine import Comm
c = Comm()
assert c.on and c.world == 2 and c.rank == int(sys.argv[1])
t = torch.full((5,), float(c.rank + 1))
c.allreduce_(t)
assert torch.equal(t, torch.full((5,), 3.0))
b = torch.full((3,), float(c.rank))
c.broadcast_(b, 0)
assert torch.equal(b, torch.zeros(3))
print("COMM_OK")
dist.destroy_process_group()
''' % (ROOT, 31500 + os.getpid() % 2000)
    script = tmp_path / "comm_worker.py"
    script.write_text(code)
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("COMM_OK" in o for o in outs)


# ------------------------------------------------------------------------------------------------ TF checkpoint-v2 bundles
REF_INDEX = [(m, ck) for m in ("celeba", "mnist_digit", "mnist_fashion") for ck in ("vae-model", "prior-model")]


@pytest.mark.parametrize("model,ck", REF_INDEX)
def test_tf_bundle_index_reproduces_reference_bytes(golden_dir, tmp_path, model, ck):
    """tests/golden/ref_ckpt_index/*.index are the reference's own TensorFlow-written `pretrained_models/<model>/<ck>.index`
    data files.  Reading one (every block's masked CRC-32C is verified, which pins crc32c + masking against TF-produced
    bytes) and re-serialising the parsed entries must give the identical file: table blocks, prefix compression, restart
    arrays, index separators, footer and BundleEntryProto encoding all as TensorFlow writes them."""
    from ladder_latent_data_distribution_modelling_amd.codes import tf_bundle as T
    src = os.path.join(golden_dir, "ref_ckpt_index", "%s_%s.index" % (model, ck))
    ents = T.read_index(src, verify=True)
    inv = json.load(open(os.path.join(golden_dir, "ckpt_inventory.json")))[model][ck]
    assert {k: v["shape"] for k, v in ents.items()} == inv
    assert all(e["dtype"] == 1 and e["size"] == 4 * int(np.prod(e["shape"] or [1])) for e in ents.values())
    offs = sorted((e["offset"], e["size"]) for e in ents.values())       # tensors are packed back to back in key order
    assert offs[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(offs, offs[1:]))
    out = str(tmp_path / "rewritten.index")
    T.write_index(out, ents)
    assert open(out, "rb").read() == open(src, "rb").read()


def test_tf_bundle_crc32c_known_answers():
    from ladder_latent_data_distribution_modelling_amd.codes import tf_bundle as T
    assert T.crc32c(b"123456789") == 0xE3069283                       # the standard CRC-32C check value
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA and T.crc32c(b"\xff" * 32) == 0x62A8AB43   # RFC 3720 B.4
    assert T.unmask_crc(T.mask_crc(0x12345678)) == 0x12345678
    a = np.random.default_rng(0).integers(0, 255, 100003, dtype=np.uint8)
    assert T.crc32c(a) == T._crc32c_py(a.tobytes())                    # C ABI slicing-by-8 == bytewise table
    assert T.crc32c(a[3:]) == T._crc32c_py(a[3:].tobytes())            # unaligned start
    half = T.crc32c(a[:50000])
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    assert L.query("ladder_crc32c_extend", half, a[50000:].ctypes.data, a.size - 50000) == T.crc32c(a)   # chainable


def test_tf_bundle_checkpoint_roundtrip_and_corruption(tmp_path, monkeypatch):
    from ladder_latent_data_distribution_modelling_amd.codes import tf_bundle as T
    rng = np.random.default_rng(4)
    tens = {"encoder/conv2d/kernel": rng.normal(size=(3, 3, 1, 16)).astype(np.float32),
            "encoder/conv2d/bias": np.zeros(16, np.float32), "sigma/Variable": np.float32(0.5),
            "decoder/dense/kernel": rng.normal(size=(300, 40)).astype(np.float32)}
    tens.update({"prior/dense_%d/kernel" % i: rng.normal(size=(7, i + 1)).astype(np.float32) for i in range(40)})
    prefix = str(tmp_path / "vae-model")
    T.save_checkpoint(prefix, tens)
    back = T.load_checkpoint(prefix)
    assert set(back) == set(tens) and back["sigma/Variable"].shape == ()
    assert all(np.array_equal(back[k], tens[k]) for k in tens)
    assert open(str(tmp_path / "checkpoint")).read().startswith('model_checkpoint_path: "vae-model"')
    only = T.load_checkpoint(prefix, names={"sigma/Variable"})
    assert list(only) == ["sigma/Variable"]
    # several data blocks + a multi-entry index block (TensorFlow's 256 KiB block size never splits these models' indices)
    monkeypatch.setattr(T, "BLOCK_SIZE", 256)
    T.save_checkpoint(str(tmp_path / "small-blocks"), tens)
    back = T.load_checkpoint(str(tmp_path / "small-blocks"))
    assert all(np.array_equal(back[k], tens[k]) for k in tens)
    # a flipped payload byte, a flipped index byte and a missing shard are all detected
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[1000] ^= 1
    open(data, "wb").write(bytes(raw))
    with pytest.raises(T.BundleError, match="checksum"):
        T.load_checkpoint(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[20] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(T.BundleError, match="checksum"):
        T.read_index(prefix + ".index")
    os.remove(str(tmp_path / "small-blocks") + ".data-00000-of-00001")
    with pytest.raises(T.BundleError, match="missing"):
        T.load_checkpoint(str(tmp_path / "small-blocks"))
    with pytest.raises(T.BundleError, match="magic"):
        open(str(tmp_path / "junk.index"), "wb").write(b"x" * 100)
        T.read_index(str(tmp_path / "junk.index"))


@pytest.mark.parametrize("h,oh", [(1, 2), (2, 8), (4, 8), (8, 16), (64, 128), (3, 7), (16, 16)])
def test_resize_transpose_absmax_bound(h, oh):
    """engine.Resize.backward registers max|dx| <= max|dy| * gain_h * gain_w for the f16x3 scale records (the last source
    row / column of the TF1-legacy bilinear resize also collects the clamped outputs -- 2.5 per axis for factor 2, not 2).  The bound
    is checked against the transpose of the oracle's resize on an all-ones dy (the worst case: all weights are non-negative)."""
    import torch
    from oracle import ladder_oracle as O
    from ladder_latent_data_distribution_modelling_amd import arch
    x = torch.zeros(1, h, h, 1, dtype=torch.float64, requires_grad=True)
    O.resize_bilinear_legacy(x, oh, oh).sum().backward()
    g = arch.resize_transpose_gain(h, oh)
    assert abs(float(x.grad.max()) - g * g) < 1e-9, (float(x.grad.max()), g)
    if oh == 2 * h and h > 1:
        assert g == 2.5


def test_bench_refuses_a_gpu_count_it_cannot_honour():
    """`python bench.py --gpus N` must never print an N-GPU line from fewer ranks.  Without a launcher it spawns the
    ranks itself -- and refuses (exit 2, no JSON) when fewer than N GPUs are visible; under a launcher whose WORLD_SIZE disagrees with
    --gpus it refuses as well.  (No GPU needed: both checks happen before any device is touched.)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LADDER_BENCH_SINGLE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "refusing" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "refusing" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_workload_labels_and_default_graph_registry():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cel = dict(exp_name="celeba", representation_size=2, n_mixtures=30)
    assert bench.workload_index(cel, 1) == "2" and bench.workload_index(cel, 8) == "3" and "per-GPU leg" in bench.workload_index(cel, 2)
    assert bench.workload_index(dict(cel, representation_size=8, n_mixtures=50), 8) == "4"
    assert bench.workload_index(dict(exp_name="mnist_fashion"), 1) == "1"
    assert abs(bench.FLOP_PER_IMG["celeba"]["executed"] - 40.172e9) < 1e6 and bench.FLOP_PER_IMG["celeba"]["algorithmic"] == 41.4e9
    # codes.utils.count_trainable_variables(scope_name): the reference's single-argument form on the default "graph"
    from codes.utils import count_trainable_variables, register_trainable_scope
    register_trainable_scope("interpolation", 10)
    assert count_trainable_variables("interpolation") == 10
    with pytest.raises(TypeError):
        count_trainable_variables(object(), "encoder")


def test_virtual_ranks_fail_instead_of_hanging_on_unequal_collective_counts():
    """a virtual rank that returns early (or issues fewer collectives) must make the job FAIL, not wait for ever."""
    import time
    import torch
    from ladder_latent_data_distribution_modelling_amd.engine import run_virtual_ranks

    def fn(rank, comm):
        t = torch.full((4,), float(rank + 1))
        comm.allreduce_(t)
        if rank == 1:
            time.sleep(0.2)
            comm.allreduce_(t)            # rank 0 never joins this one
        return t

    t0 = time.time()
    with pytest.raises(RuntimeError, match="virtual rank 1 failed"):
        run_virtual_ranks(2, fn)
    assert time.time() - t0 < 30

    def ok(rank, comm):
        t = torch.full((4,), float(rank + 1))
        return comm.allreduce_(comm.allreduce_(t))

    a, b = run_virtual_ranks(2, ok)
    assert a.tolist() == b.tolist() == [6.0] * 4
