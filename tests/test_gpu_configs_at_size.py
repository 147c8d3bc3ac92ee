"""BASELINE.json configs[3] / configs[4] exercised AT SIZE on one GPU (round 3, VERDICT r2 items 1 + 9):

  * configs[4] -- CelebA nh=512, z=64, 2-rung ladder with representation_size=8 and K=50 full-covariance components
    (codes/celeba_r8k50_config.json; mixture of /root/reference codes/base.py:88-124, MC term base.py:308-313): the full-resolution
    network against the float64 oracle run live (batch 8), and the complete batch-128 step in situ (`gmm_logprob_kernel<8>` at
    L*B*K = 640 000 component evaluations per launch) in f16x3 against the native fp32 build;
  * the data-parallel scheme C1-C4 at full size: 2 ranks x 64 images sharing cuda:0 over gloo against 1 rank x 128 images, f16x3,
    so that the split / halo kernels, the batch-norm statistics emitted by conv epilogues, plane-only activations and the
    asynchronous decoder gradient bucket all run together with the exchange steps (the tiny golden config engages none of them);
  * hipGraph replay followed by an EAGER evaluation must see the weights the replays produced (packed split-filter images are
    keyed on the optimiser-group version, which a replay has to bump).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ladder_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALARS_RUN1 = ["loss_ae", "elbo", "l1_reconstruction_error", "l2_reconstruction_error", "entropy_z", "crossEntropy_prior",
                "sigma_regularisor", "reconstruction_likelihood", "sigma", "mean_pixel_error", "elbo_prior",
                "crossEntropy_representation", "entropy_t", "code_reconstruction_likelihood", "code_l1_reconstruction_error",
                "representation_regularisor", "inner_sigma", "mean_code_error", "crossEntropy_prior_sg"]
SCALARS_RUN3 = ["elbo_prior", "code_l1_reconstruction_error", "code_reconstruction_likelihood", "entropy_t",
                "crossEntropy_representation", "inner_sigma", "loss_prior"]


def _ok(a, b, rtol, atol=1e-4):
    return abs(a - b) <= rtol * abs(b) + atol


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-6)


def _cfg(name):
    return json.load(open(os.path.join(ROOT, "codes", name)))


def _gm(cfg):
    """R = 2 -> the reference's own fitted mixture (first K components, renormalised); R = 8 -> m ~ N(0, 1.5^2),
    Sigma = A A^T / R + 0.05 I, w ~ Dirichlet(1), default_rng(3) -- exactly what bench.py feeds."""
    K, R = int(cfg["n_mixtures"]), int(cfg["representation_size"])
    if R == 2:
        fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
        return dict(weights=fix["w_full"][:K] / fix["w_full"][:K].sum(), means=fix["m_full"][:K], covs=fix["K_full"][:K])
    rng = np.random.default_rng(3)
    A = rng.normal(0, 0.3, (K, R, R))
    return dict(weights=rng.dirichlet(np.ones(K)), means=rng.normal(0, 1.5, (K, R)), covs=A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))


def test_celeba_r8k50_fullres_vs_live_oracle():
    """configs[4]'s network (codes/celeba_r8k50_config.json) at batch 8 against the float64 oracle on the same inputs and noise:
    RUN#1 fetches to 2e-5 (bar 1e-3 on the ELBO), RUN#3 fetches to 5e-5, every gradient tensor of both groups to
    max(1.5e-3 of its scale, 5x the deviation of the oracle evaluated in fp32)."""
    from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
    cfg = _cfg("celeba_r8k50_config.json")
    assert cfg["representation_size"] == 8 and cfg["n_mixtures"] == 50 and cfg["num_hidden_units"] == 512 and cfg["code_size"] == 64
    cfg["batch_size"] = B = 8
    rng = np.random.default_rng(23)
    x = rng.random((B, 128, 128, 3)).astype(np.float32)
    P = O.init_params(cfg, seed=7)
    gm = {k: np.asarray(v, np.float32) for k, v in _gm(cfg).items()}
    noise = O.make_noise(cfg, B, rng, np.float32)
    engs = {}
    for prec in ("f16x3", "f32"):                      # (both precisions against ONE evaluation of the oracle)
        engs[prec] = LadderEngine(dict(cfg, matmul_precision=prec), "cuda:0", values=P, seed=1)
        engs[prec].set_mixture(gm["weights"], gm["means"], gm["covs"])
    for group, names, tol in (("ae", SCALARS_RUN1, 2e-5), ("prior", SCALARS_RUN3, 5e-5)):
        ref = O.run(O.OracleState(cfg, P, np.float64), x, noise, gm, False, False, train=group, lr=0.0)
        ref32 = O.run(O.OracleState(cfg, P, np.float32), x, noise, gm, False, False, train=group, lr=0.0)
        for prec, eng in engs.items():
            (eng.run_ae if group == "ae" else eng.run_prior)(x, 0.0, noise, False, False)
            f = eng.fetch()
            for k in names:
                assert _ok(f[k], float(ref[k]), tol), (prec, group, k, f[k], float(ref[k]))
            worst = 0.0
            for name, g in ref["_grads"].items():
                got = eng.ps.g[name].cpu().numpy().reshape(g.shape).astype(np.float64)
                scale = np.abs(g).max()
                if scale < 1e-9:
                    continue
                cond = np.abs(ref32["_grads"][name].astype(np.float64) - g).max()
                err = np.abs(got - g).max()
                worst = max(worst, err / max(1.5e-3 * scale, 5 * cond))
                assert err < max(1.5e-3 * scale, 5 * cond), (prec, group, name, err, scale, cond)
            print("r8k50 %s %s: worst gradient error / bound = %.3f" % (prec, group, worst))


STEP_WORKER = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
cfg = json.load(open(os.path.join(%(root)r, "codes", os.environ["LADDER_TEST_CONFIG"])))
cfg["matmul_precision"] = os.environ.get("LADDER_TEST_PRECISION", "f32")
B = 128
x = torch.rand(B, 128, 128, 3, generator=torch.Generator().manual_seed(5)).numpy()
eng = LadderEngine(cfg, "cuda:0", seed=1, noise_seed=99)
K, R = int(cfg["n_mixtures"]), int(cfg["representation_size"])
rng = np.random.default_rng(3)
A = rng.normal(0, 0.3, (K, R, R))
eng.set_mixture(rng.dirichlet(np.ones(K)), rng.normal(0, 1.5, (K, R)), A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))
out = {}
eng.run_ae(x, 2.5e-4, None, False, False); out["ae"] = eng.fetch()
out["grad_norm"] = float(eng.ps.grad["ae"].double().norm())
mu_t, sd_t = eng.lat_t[0].double().cpu().numpy(), eng.lat_t[1].double().cpu().numpy()
eng.run_sigma(x, 2.5e-4, None, False, False); out["sigma"] = eng.fetch(["sigma"])
eng.run_prior(x, 1e-4, None, False, False, reuse_encoder=True); out["prior"] = eng.fetch()
out["prior_grad_norm"] = float(eng.ps.grad["prior"].double().norm())
eng.run_inner_sigma(x, 2e-4, None, False, False, reuse_encoder=True)
np.savez(sys.argv[1], fetch=json.dumps(out), mu_t=mu_t, sd_t=sd_t)
'''


def test_celeba_r8k50_full_size_in_situ(tmp_path):
    """configs[4]'s per-GPU step at full size (batch 128, R=8, K=50, L=100: `gmm_logprob_kernel<8>` at 640 000 component evaluations
    per launch inside the real training graph): the complete 4-run iteration in f16x3 must reproduce the native-fp32 build with the
    same seeds and device noise -- RUN#1 fetches (incl. the mixture cross-entropy and elbo_prior) to 2e-6, the gradient norm to 1e-4,
    RUN#3 (evaluated after the first Adam step) to 2e-3."""
    script = tmp_path / "step_worker.py"
    script.write_text(STEP_WORKER % dict(root=ROOT))
    res = {}
    for tag in ("f32", "f16x3"):
        outp = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, LADDER_TEST_PRECISION=tag, LADDER_TEST_CONFIG="celeba_r8k50_config.json")
        p = subprocess.run([sys.executable, str(script), outp], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:]
        res[tag] = np.load(outp)
    fa, fb = json.loads(str(res["f16x3"]["fetch"])), json.loads(str(res["f32"]["fetch"]))
    for k in SCALARS_RUN1:
        assert np.isfinite(fa["ae"][k]) and _ok(fa["ae"][k], fb["ae"][k], 2e-6, 1e-5), (k, fa["ae"][k], fb["ae"][k])
    # (round 5: 1.2e-4 measured -- the strict-fp32 build now contracts conv2d_5 / conv2d_4 against EFFECTIVE taps (sums of up to nine fp32 filter
    # values, one more rounding per tap) where the f16x3 build keeps the direct form; round 4, both direct there: 4.7e-5)
    assert _rel(fa["grad_norm"], fb["grad_norm"]) < 3e-4
    # sigma AFTER the first Adam step (lr 2.5e-4): that step moves every weight by ~lr along the SIGN of its gradient, so the entries whose gradient is
    # rounding noise land +-lr apart in two builds.  Measured (profiles/tools/r5_sigma_probe.py): the strict-fp32 build at `upsample_fused_convs` 0 / 3 / 4
    # gives 0.6909418 / 0.6909575 / 0.6909476 (2e-5 among themselves), f16x3 0.6907746 (1.8e-4 from all three; its gradient norm differs by 1.2e-4)
    assert _ok(fa["sigma"]["sigma"], fb["sigma"]["sigma"], 1e-5, 3e-4)
    for k in SCALARS_RUN3:
        assert _ok(fa["prior"][k], fb["prior"][k], 2e-3, 1e-5), (k, fa["prior"][k], fb["prior"][k])
    # (the prior group's gradient norm after that Adam step: 483.23 / 483.21 / 484.13 for the strict-fp32 build at levels 0 / 3 / 4, 479.19 in f16x3 -- the
    # same sign-of-noise amplification; 1.0 % measured between the two default builds)
    assert _rel(fa["prior_grad_norm"], fb["prior_grad_norm"]) < 2e-2
    # the 8-dimensional posterior the mixture kernel integrates over is the same in both builds
    assert np.abs(res["f16x3"]["mu_t"] - res["f32"]["mu_t"]).max() < 1e-4 * max(1.0, np.abs(res["f32"]["mu_t"]).max())


DP_WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world = int(sys.argv[1]), int(sys.argv[2])
virtual = len(sys.argv) > 4 and sys.argv[4] == "virtual"        # `world` virtual ranks in THIS process (engine.run_virtual_ranks)
everything = len(sys.argv) > 4                                   # save every gradient tensor and every parameter
if world > 1 and not virtual:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine, run_virtual_ranks
cfg = json.load(open(os.path.join(%(root)r, "codes", %(config)r)))
cfg["matmul_precision"] = %(prec)r
if len(sys.argv) > 4 and sys.argv[4] == "deterministic":
    cfg["deterministic_allreduce"] = 1
Bg = 128
Bl = Bg // world
Z, R, Lmc, K = cfg["code_size"], cfg["representation_size"], cfg["n_MC_samples"], cfg["n_mixtures"]
g = torch.Generator().manual_seed(11)
x = torch.rand(Bg, 128, 128, 3, generator=g).numpy()
rng = np.random.default_rng(12)
noise = [dict(eps_z=rng.standard_normal((Bg, Z)).astype(np.float32), eps_t=rng.standard_normal((Bg, R)).astype(np.float32),
              eps_mc=rng.standard_normal((Lmc, Bg, R)).astype(np.float32)) for _ in range(4)]
from ladder_latent_data_distribution_modelling_amd import _lib as L
grng = np.random.default_rng(3)
if R == 2:
    fix = np.load(os.path.join(%(root)r, "tests", "golden", "GM_prior_info.npz"))
    gm = (fix["w_full"][:K] / fix["w_full"][:K].sum(), fix["m_full"][:K], fix["K_full"][:K])
else:
    A = grng.normal(0, 0.3, (K, R, R))
    gm = (grng.dirichlet(np.ones(K)), grng.normal(0, 1.5, (K, R)), A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))


def job(rank, comm=None):
    sl = slice(Bl * rank, Bl * (rank + 1))
    shard = [dict(eps_z=n["eps_z"][sl], eps_t=n["eps_t"][sl], eps_mc=np.ascontiguousarray(n["eps_mc"][:, sl])) for n in noise]
    eng = LadderEngine(dict(cfg, batch_size=Bl) if everything else cfg, "cuda:0", seed=1, comm=comm)     # comm None: Comm() picks up the initialised group
    assert eng.ctx.comm.world == world
    eng.set_mixture(*gm)
    # which kernels engage at this per-rank batch (the point of the test): the split halo conv on the big decoder maps
    assert Bl < 64 or L.query("ladder_conv3x3_split_eligible", Bl, 128, 128, 128, 128) == 1
    out = {}
    eng.run_ae(x[sl], 2.5e-4, shard[0], False, False); out["ae"] = eng.fetch()
    names = list(eng.ps.g) if everything else ["decoder/conv2d_7/kernel", "decoder/conv2d_5/kernel", "encoder/conv2d_1/kernel",
                                               "encoder/batch_normalization/gamma", "decoder/dense/kernel"]
    gsel = {n: eng.ps.g[n].detach().cpu().numpy().copy() for n in names}
    out["grad_norm"] = float(eng.ps.grad["ae"].double().norm())
    eng.run_sigma(x[sl], 2.5e-4, shard[1], False, False); out["sigma"] = eng.fetch(["sigma"])
    eng.run_prior(x[sl], 1e-4, shard[2], False, False); out["prior"] = eng.fetch()
    if everything:
        gsel.update({n: eng.ps.g[n].detach().cpu().numpy().copy() for n in eng.ps.g if n.startswith("prior/")})
    eng.run_inner_sigma(x[sl], 2e-4, shard[3], False, False)
    pn = list(eng.ps.w) if everything else ["decoder/conv2d_7/kernel", "encoder/conv2d_1/kernel", "prior/dense/kernel", "sigma/Variable", "inner_sigma/Variable"]
    psel = {n: eng.ps.w[n].detach().cpu().numpy().copy() for n in pn}
    return out, gsel, psel


if virtual:
    out, gsel, psel = run_virtual_ranks(world, job)[0]
else:
    out, gsel, psel = job(rank)
if rank == 0:
    np.savez(sys.argv[3], fetch=json.dumps(out), **{"g/" + k: v for k, v in gsel.items()}, **{"p/" + k: v for k, v in psel.items()})
if world > 1 and not virtual:
    dist.barrier()
    dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("config", ["celeba_config.json", "celeba_r8k50_config.json"])
def test_data_parallel_equals_virtual_ranks_bit_for_bit(tmp_path, config, world):
    """VERDICT r4 #4 / SURVEY 8(b) ("deterministic reductions (fixed split order) so DP parity tests are bit-stable"): the `world`-PROCESS
    data-parallel job (gloo, every rank on cuda:0; global batch 128 = 2 x 64 / 4 x 32, strict fp32) against the SAME job as `world` virtual
    ranks in one process (engine.VirtualComm: per-rank batch-norm / ELBO partials and filter-gradient sums, combined in the all-reduce's
    order) -- all four runs of an iteration: every fetch, EVERY gradient tensor of the AE and prior groups and EVERY updated parameter must
    be IDENTICAL, bit for bit.  Two ranks: the backend's own all-reduce (a + b in any schedule); four ranks: the rank-ordered all-reduce
    (`deterministic_allreduce`), whose order the virtual ranks reproduce.  Global-batch batch norm (reference codes/models.py:398-460) and
    clip-after-mean (codes/base.py:462-464) are inside both jobs; their values are held against the float64 oracle in
    test_data_parallel_full_resolution_vs_live_float64_oracle."""
    port = 33000 + (os.getpid() + 7 * world) % 2000
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % dict(root=ROOT, port=port, config=config, prec="f32"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outp, outv = str(tmp_path / "dp.npz"), str(tmp_path / "virtual.npz")
    mode = "native" if world == 2 else "deterministic"
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), outp, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-2500:] for o in outs]
    pv = subprocess.run([sys.executable, str(script), "0", str(world), outv, "virtual"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert pv.returncode == 0, pv.stdout[-2500:]
    a, b = np.load(outp), np.load(outv)
    fa, fb = json.loads(str(a["fetch"])), json.loads(str(b["fetch"]))
    assert fa == fb, [(k, q, fa[k].get(q) if isinstance(fa[k], dict) else fa[k], fb[k].get(q) if isinstance(fb[k], dict) else fb[k])
                      for k in fa for q in (fa[k] if isinstance(fa[k], dict) else [None]) if (fa[k][q] if q else fa[k]) != (fb[k][q] if q else fb[k])][:5]
    assert sorted(a.files) == sorted(b.files) and sum(k.startswith("g/") for k in a.files) > 60 and sum(k.startswith("p/") for k in a.files) > 60
    for k in a.files:
        if k != "fetch":
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (k, float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()))



@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("config", ["celeba_config.json", "celeba_r8k50_config.json"])
def test_data_parallel_full_size_two_ranks_one_gpu(tmp_path, config, prec):
    """configs[3] / configs[4] per-GPU legs as a 2-rank data-parallel job at FULL size on one GPU (2 x 64 images over gloo, both
    ranks on cuda:0; RCCL refuses two ranks on one device) against the single-process step on the 128 images, strict fp32 (the
    default) and f16x3, explicit noise sharded by sample: the global-batch all-reduced step must equal the single-rank step -- RUN#1
    fetches to 1e-5, the all-reduced gradient (norm 1e-4; selected tensors incl. a batch-norm gamma, whose gradient is global BEFORE
    C1) to bars derived from the measured rounding sensitivity of each tensor (see below), and the updated parameters to Adam-noise level."""
    port = 31000 + os.getpid() % 2000
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % dict(root=ROOT, port=port, config=config, prec=prec))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out2, out1 = str(tmp_path / "dp2.npz"), str(tmp_path / "dp1.npz")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", out2], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-2500:] for o in outs]
    p1 = subprocess.run([sys.executable, str(script), "0", "1", out1], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p1.returncode == 0, p1.stdout[-2500:]
    a, b = np.load(out2), np.load(out1)
    fa, fb = json.loads(str(a["fetch"])), json.loads(str(b["fetch"]))
    for k in SCALARS_RUN1:
        assert _ok(fa["ae"][k], fb["ae"][k], 1e-5, 1e-5), (k, fa["ae"][k], fb["ae"][k])
    assert _rel(fa["grad_norm"], fb["grad_norm"]) < 1e-4
    # sigma AFTER the first Adam step (lr 2.5e-4): that step moves every weight by ~lr along the SIGN of its gradient, so the entries whose gradient is
    # rounding noise land +-lr apart in two builds.  Measured (profiles/tools/r5_sigma_probe.py): the strict-fp32 build at `upsample_fused_convs` 0 / 3 / 4
    # gives 0.6909418 / 0.6909575 / 0.6909476 (2e-5 among themselves), f16x3 0.6907746 (1.8e-4 from all three; its gradient norm differs by 1.2e-4)
    assert _ok(fa["sigma"]["sigma"], fb["sigma"]["sigma"], 1e-5, 3e-4)
    for k in SCALARS_RUN3:          # evaluated after the first (sign-like) Adam step of RUN#1: see test_celeba_full_size_halo_kernels_in_situ
        assert _ok(fa["prior"][k], fb["prior"][k], 2e-3, 1e-5), (k, fa["prior"][k], fb["prior"][k])
    for k in a.files:
        if k.startswith("g/"):
            ga, gb = a[k].astype(np.float64), b[k].astype(np.float64)
            # Per SAMPLE the two jobs run the same kernels on the same tiles (a halo / gather tile never straddles images; the f16x3 scales
            # are per sample since round 3), so activations differ only through the ROUNDING of the batch statistics (per-rank partial sums +
            # all-reduce against one pass) and the gradients through the order of the filter-gradient reductions.  How far that alone moves
            # these tensors was measured in round 4 on SINGLE-process pairs (profiles/r04_dp_sensitivity.txt): computing the batch-norm sums
            # in the conv epilogue instead of a separate pass -- nothing else changed -- moves encoder/conv2d_1 by 1.5e-3, the batch-norm gamma
            # by 1.7e-3 and decoder/dense by 2.2e-3 of their scale (this randomly initialised network amplifies a rounding-level change of
            # the normalisation statistics ~1e4-fold on its deepest tensors).  2 x 64 against 1 x 128 measures 1.5e-5 / 2.4e-4 on the decoder
            # convolutions and 4e-3 ... 7e-3 on those three (f16x3: 1.2e-3 ... 1.4e-3).  Bars: 4x the measured values (round 3: 5e-2 for
            # everything); a wrong C2 backward term or a mis-scaled C1 bucket is an O(1) error on every tensor behind it, and the DP path is
            # held against the float64 oracle with conditioning-based bounds in test_data_parallel_full_resolution_vs_live_float64_oracle.
            tol = 2e-4 if k.endswith("conv2d_7/kernel") else (2e-3 if k.endswith("conv2d_5/kernel") else 2.5e-2)
            err = np.abs(ga - gb).max() / np.abs(gb).max()
            print("%s %s: all-reduced gradient of 2 x 64 images vs 1 x 128: max error %.1e of the tensor scale" % (prec, k, err))
            assert np.isfinite(ga).all() and err < tol, (k, err)
        elif k.startswith("p/"):
            diff = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            assert np.median(diff) < 2e-6 and diff.max() <= 2 * 2.5e-4 + 1e-7, (k, np.median(diff), diff.max())


def test_graph_replay_then_eager_evaluation_sees_current_weights():
    """with hipGraph replay on, only the host-side Adam call bumped the optimiser-group version that keys the packed
    split-filter images; a replay updates the weights on the device, so an EAGER forward after replays (val_step, fit_GMM_VI, decode,
    Session) reused the images the last graph had packed from the PRE-update weights.  Six AE-only iterations (SG regime: no later
    graph re-packs), then evaluate(): must equal a fresh engine loaded with the same parameters bit for bit."""
    from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
    cfg = _cfg("celeba_config.json")
    cfg["batch_size"] = B = 8
    cfg["matmul_precision"] = "f16x3"
    x = torch.rand(B, 128, 128, 3, generator=torch.Generator().manual_seed(3)).cuda()
    eng = LadderEngine(cfg, "cuda:0", seed=1, noise_seed=5)
    eng.use_graphs = True
    eng.set_sg_mixture()
    for _ in range(6):
        eng.run_ae(x, 2e-3, None, True, False)                   # (a large step: stale filters would differ visibly)
    assert len(eng._graphs) >= 1, "the AE run was never replayed as a graph"
    rng = np.random.default_rng(1)
    noise = O.make_noise(cfg, B, rng, np.float32)
    eng.evaluate(x, noise, True, False)
    got, xhat = eng.fetch(), eng.xhat.clone()
    fresh = LadderEngine(cfg, "cuda:0", values=eng.ps.to_dict(), seed=1)
    fresh.set_sg_mixture()
    fresh.evaluate(x, noise, True, False)
    ref = fresh.fetch()
    assert torch.equal(xhat, fresh.xhat), float((xhat - fresh.xhat).abs().max())
    for k in ("elbo", "l1_reconstruction_error", "entropy_z", "crossEntropy_prior", "sigma_regularisor", "loss_ae"):
        assert got[k] == ref[k], (k, got[k], ref[k])


DP_ORACLE_WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world = int(sys.argv[1]), int(sys.argv[4])
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
from oracle import ladder_oracle as O
from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
from ladder_latent_data_distribution_modelling_amd import _lib as L
cfg = json.load(open(os.path.join(%(root)r, "codes", "celeba_config.json")))
cfg["matmul_precision"] = %(prec)r
inp = np.load(sys.argv[2])
Bg = inp["x"].shape[0]
Bl = Bg // world
cfg["batch_size"] = Bl
sl = slice(Bl * rank, Bl * (rank + 1))
noise = dict(eps_z=inp["eps_z"][sl], eps_t=inp["eps_t"][sl], eps_mc=np.ascontiguousarray(inp["eps_mc"][:, sl]))
eng = LadderEngine(cfg, "cuda:0", values=O.init_params(cfg, seed=6), seed=1)
assert eng.ctx.comm.world == world
eng.set_mixture(inp["gm_w"], inp["gm_m"], inp["gm_c"])
calls, real = [], L.call
L.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
eng.run_ae(inp["x"][sl], 0.0, noise, False, False)
L.call = real
f = eng.fetch()
if rank == 0:
    np.savez(sys.argv[3], fetch=json.dumps(f), calls=json.dumps(sorted(set(calls))), **{"g/" + k: v.detach().cpu().numpy() for k, v in eng.ps.g.items()})
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_data_parallel_full_resolution_vs_live_float64_oracle(tmp_path, prec):
    """the PRODUCT data-parallel path at full resolution against the float64 ORACLE, not against another GPU run: 2 ranks x
    8 images (the per-rank batch at which the fused halo kernels of conv2d_7 engage) over gloo on one GPU -- C2 (global-batch batch-norm
    statistics, forward and backward, reference codes/models.py:398-460 on one device), C3 (ELBO partials) and C1 (gradient sum, both
    buckets; clip after the mean, codes/base.py:462-464) -- must reproduce the oracle's single-process step on the 16 images: fetches to
    2e-5, EVERY gradient tensor in relative L2 and in max norm (bars and their derivation below).  (That the oracle's own data-parallel
    restatement equals its single-process step is pinned on the CPU by tests/test_host_cpu.py.)"""
    cfg = _cfg("celeba_config.json")
    Bg = 16
    cfg["batch_size"] = Bg
    rng = np.random.default_rng(23)
    x = rng.random((Bg, 128, 128, 3)).astype(np.float32)
    P = O.init_params(cfg, seed=6)
    K = int(cfg["n_mixtures"])
    fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
    gm = dict(weights=fix["w_full"][:K] / fix["w_full"][:K].sum(), means=fix["m_full"][:K], covs=fix["K_full"][:K])
    noise = O.make_noise(cfg, Bg, rng, np.float32)
    inp, outp = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(inp, x=x, gm_w=gm["weights"], gm_m=gm["means"], gm_c=gm["covs"], **noise)
    script = tmp_path / "dp_oracle_worker.py"
    script.write_text(DP_ORACLE_WORKER % dict(root=ROOT, port=33000 + os.getpid() % 2000, prec=prec))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), inp, outp, "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    # the oracle runs on the host while the two ranks run on the GPU
    ref = O.run(O.OracleState(cfg, P, np.float64), x, noise, gm, False, False, train="ae", lr=0.0)
    ref32 = O.run(O.OracleState(cfg, P, np.float32), x, noise, gm, False, False, train="ae", lr=0.0)
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-2500:] for o in outs]
    got = np.load(outp)
    f, calls = json.loads(str(got["fetch"])), json.loads(str(got["calls"]))
    # the fused forward pair (resize + conv2d_7 + RGB projection) engages at 8 images per rank; the fused backward-data needs 32 (512
    # workgroups of 4x fewer pixels) -- that one is held against the direct path at batch 128 in tests/test_gpu_up2.py
    # (round 5, strict fp32: the PROJECTED form instead -- every resize -> conv pair from the low-resolution tensor at any batch size, no resize launch left)
    if prec == "f32":
        # (round 6: the forward of a pair is the one-launch ladder_up2proj_fused_fwd)
        assert "ladder_up2proj_fused_fwd" in calls and "ladder_upfproj_bwd_combine" in calls and not any("resize" in c for c in calls), calls
    else:
        assert "ladder_conv3x3_up2_split_proj" in calls and "ladder_in_style_fwd_resize2x_keep" in calls, calls
    for k in SCALARS_RUN1:
        assert _ok(f[k], float(ref[k]), 2e-5), (k, f[k], float(ref[k]))
    # Per tensor: relative L2 error <= max(5e-3, 5 x the L2 deviation of the oracle evaluated in fp32 on the CPU) AND max error <= 5e-2 of the
    # tensor's scale.  Why not the max-norm bar of the batch-8 test: at 16 images some leaky-ReLU pre-activation of the small decoder /
    # encoder maps lands within fp32 rounding of zero in one build or another (measured in round 4, profiles/r04_dp_sensitivity.txt: the SINGLE
    # process fp32 run flips one in decoder/conv2d_3's 8x8 map, the single-process f16x3 run one in the encoder's 4x4 map, the 2-rank
    # runs other ones or none): the float64 oracle takes slope 1 where the GPU takes 0.2, which moves THAT output channel's gradient by
    # 1-3 % of the tensor's largest element and everything behind it in the backward chain by ~1.5e-3 in L2 -- a property of the
    # comparison against float64, not of the kernels (every convolution of these layers measures <= 2e-6 against float64 at batch 8 / 16 /
    # 32).  A wrong C2 backward term or a mis-scaled C1 bucket is an O(1e-1 ... 1) L2 error on every tensor behind it.
    worst, wname, wmax = 0.0, None, 0.0
    for name, g in ref["_grads"].items():
        a = got["g/" + name].reshape(g.shape).astype(np.float64)
        n2, scale = np.linalg.norm(g), np.abs(g).max()
        if scale < 1e-9:
            continue
        c2 = np.linalg.norm(ref32["_grads"][name].astype(np.float64) - g) / n2
        e2, emax = np.linalg.norm(a - g) / n2, np.abs(a - g).max() / scale
        bound = max(5e-3, 5 * c2)
        if e2 / bound > worst:
            worst, wname = e2 / bound, name
        wmax = max(wmax, emax)
        assert e2 < bound and emax < 5e-2, (prec, name, e2, c2, emax)
    print("data parallel 2 x 8 images, %s: worst L2 gradient error / bound = %.3f (%s), worst max-norm error %.1e" % (prec, worst, wname, wmax))


def test_celeba_configs2_full_batch_128_vs_float64_oracle():
    """VERDICT r5 "weak" #1: BASELINE configs[2] at its FULL batch (codes/celeba_config.json: 128 images of 128x128, nh 512, z 64, K 30) against the
    float64 oracle evaluated live on the same 128 images, parameters and noise (~75 GB and a few minutes of host time: the GPU hosts have 3 TB) --
    not against another build of the same library.  RUN#1 (reference codes/base.py:587-594): every fetched scalar to 2e-5 (ELBO bar of
    BASELINE.json: 1e-3), the reconstruction to 2e-5 of its range, and the filter gradients of the layers that carry the step -- conv2d_7,
    conv2d_6, conv2d_5 (projected pairs at their largest maps), the 1x1 output conv, the two image-side encoder convs and the first dense layer -- in
    relative L2 and in max norm against per-tensor bars (a few times the measured values, which are printed)."""
    import time
    from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
    cfg = _cfg("celeba_config.json")
    B = int(cfg["batch_size"])
    assert B == 128 and cfg["num_hidden_units"] == 512 and cfg["code_size"] == 64 and cfg["n_mixtures"] == 30 and cfg["matmul_precision"] == "f32"
    rng = np.random.default_rng(31)
    x = rng.random((B, 128, 128, 3)).astype(np.float32)
    P = O.init_params(cfg, seed=11)
    gm = {k: np.asarray(v, np.float32) for k, v in _gm(cfg).items()}
    noise = O.make_noise(cfg, B, rng, np.float32)
    eng = LadderEngine(cfg, "cuda:0", values=P, seed=1)
    eng.set_mixture(gm["weights"], gm["means"], gm["covs"])
    eng.run_ae(x, 0.0, noise, False, False)
    f = eng.fetch()
    xhat = eng.xhat.cpu().numpy().astype(np.float64)
    # name -> (bar on the relative L2 error, bar on max error / tensor scale); measured on MI355X (round 6): conv2d_7 9.9e-6 / 1.3e-5, conv2d_6 7.0e-5 / 9.7e-5,
    # conv2d_5 1.2e-4 / 2.5e-4, conv2d_8 4.5e-6 / 6.9e-6, encoder/conv2d 1.3e-3 / 1.3e-3, encoder/conv2d_1 1.3e-3 / 6.9e-3, decoder/dense 6.4e-4 / 7.2e-4 (the
    # deep tensors collect every leaky-ReLU pre-activation that float64 and fp32 put on different sides of zero: single filter taps move, the L2 error stays)
    bars = {"decoder/conv2d_7/kernel": (1e-4, 1e-4), "decoder/conv2d_6/kernel": (5e-4, 5e-4), "decoder/conv2d_5/kernel": (2e-3, 2e-3),
            "decoder/conv2d_8/kernel": (5e-5, 5e-5), "decoder/conv2d_8/bias": (5e-5, 5e-5), "encoder/conv2d/kernel": (5e-3, 5e-3),
            "encoder/conv2d_1/kernel": (5e-3, 2.5e-2), "decoder/dense/kernel": (3e-3, 3e-3)}
    names = list(bars)
    got = {n: eng.ps.g[n].cpu().numpy().astype(np.float64) for n in names}
    del eng
    torch.cuda.empty_cache()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    t0 = time.time()
    ref = O.run(O.OracleState(cfg, P, np.float64), x, noise, gm, False, False, train="ae", lr=0.0)
    print("float64 oracle, batch %d: %.0f s on %d threads" % (B, time.time() - t0, torch.get_num_threads()))
    for k in SCALARS_RUN1:
        assert np.isfinite(f[k]) and _ok(f[k], float(ref[k]), 2e-5), (k, f[k], float(ref[k]))
    dec = np.asarray(ref["decoded"], np.float64)
    err = np.abs(xhat - dec).max() / np.abs(dec).max()
    print("reconstruction: max error / range = %.2e" % err)
    assert err < 5e-5                                      # (measured 1.6e-5)
    for n in names:
        g = np.asarray(ref["_grads"][n], np.float64).reshape(got[n].shape)
        l2 = np.linalg.norm(got[n] - g) / np.linalg.norm(g)
        mx = np.abs(got[n] - g).max() / np.abs(g).max()
        print("%-28s rel L2 %.2e   max / scale %.2e" % (n, l2, mx))
        assert l2 < bars[n][0] and mx < bars[n][1], (n, l2, mx)
