"""Long-horizon numerics of the default f16x3 contraction mode (round 3, VERDICT r2 items 2c / "what's missing" 5):
no test before this one ran more than 2 iterations in f16x3.

  * five complete 4-run iterations of the full-resolution CelebA network (batch 8) against the float64 oracle with explicit noise, in
    f16x3 AND in native fp32: every iteration's ELBO / elbo_prior within the BASELINE bar, and f16x3 no further from the oracle than
    the fp32 build is (up to a small factor);
  * a 200-iteration full-size trajectory (batch 128, device noise, four alternating minibatches) in f16x3, bf16x6 and native fp32 from the
    same seeds.  Training dynamics amplify ANY rounding difference (Adam's first steps are sign-like; leaky-ReLU masks flip): two fp32-class
    builds decorrelate after ~20 iterations -- measured on MI355X (profiles/r03_traj_*.json): |ELBO_f16x3 - ELBO_f32| / |ELBO_f32| up to
    1.4e-3 over iterations 0-4, 6e-3 over 5-19, then 2-10 % like any pair of differently rounded runs (fp32 halo kernels vs fp32 generic
    kernels: 5 %).  So the test pins (i) the early window, where deviations are still rounding-sized, against the bf16x6 build as the
    yardstick of "another fp32-class rounding", and (ii) the statistics of the late window (mean ELBO of the last 50 iterations, that
    training made the same progress, nothing non-finite)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ladder_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_five_iterations_fullres_vs_oracle_f16x3_and_f32():
    from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
    cfg0 = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
    cfg0["batch_size"] = B = 8
    rng = np.random.default_rng(31)
    x = rng.random((B, 128, 128, 3)).astype(np.float32)
    P = O.init_params(cfg0, seed=9)
    K = int(cfg0["n_mixtures"])
    fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
    gm = dict(weights=(fix["w_full"][:K] / fix["w_full"][:K].sum()).astype(np.float32), means=fix["m_full"][:K].astype(np.float32),
              covs=fix["K_full"][:K].astype(np.float32))
    n_it = 5
    noises = [[O.make_noise(cfg0, B, rng, np.float32) for _ in range(4)] for _ in range(n_it)]
    epoch = int(cfg0["sg_pretraining"]) + 1
    lr_ae = float(cfg0["learning_rate_ae"])
    lr_s = float(cfg0["learning_rate_sigma"]) * 0.99 ** (epoch - 1)
    lr_p = float(cfg0["learning_rate_prior"]) * 1.01 ** (epoch - 1)
    lr_i = float(cfg0["learning_rate_inner_sigma"]) * 1.01 ** (epoch - 1)
    st = O.OracleState(cfg0, P, np.float64)
    ref = [O.train_iteration(st, x, noises[i], gm, cur_epoch=epoch, lr_ae=lr_ae) for i in range(n_it)]
    dev, raw = {}, {}
    # "f32-generic": the strict-fp32 build with every convolution on the round-1 gather kernels and nothing fused (the run-time test switches
    # LADDER_DISABLE_HALO / LADDER_DISABLE_BNSTATS) -- the same arithmetic as the default in other summation orders
    for prec in ("f32", "f16x3", "f32-generic"):
        cfg = dict(cfg0, matmul_precision=prec.split("-")[0])
        for k_ in ("LADDER_DISABLE_HALO", "LADDER_DISABLE_BNSTATS"):
            if prec == "f32-generic":
                os.environ[k_] = "1"
            else:
                os.environ.pop(k_, None)
        eng = LadderEngine(cfg, "cuda:0", values=P, seed=1)
        eng.set_mixture(gm["weights"], gm["means"], gm["covs"])
        d = []
        for i in range(n_it):
            eng.run_ae(x, lr_ae, noises[i][0], False, False)
            f1 = eng.fetch()
            eng.run_sigma(x, lr_s, noises[i][1], False, False)
            sg = eng.fetch(["sigma"])["sigma"]
            eng.run_prior(x, lr_p, noises[i][2], False, False)
            f3 = eng.fetch()
            eng.run_inner_sigma(x, lr_i, noises[i][3], False, False)
            r = ref[i]
            d.append(dict(elbo=abs(f1["elbo"] - float(r["run1"]["elbo"])) / abs(float(r["run1"]["elbo"])),
                          l1=abs(f1["l1_reconstruction_error"] - float(r["run1"]["l1_reconstruction_error"])) / abs(float(r["run1"]["l1_reconstruction_error"])),
                          sigma=abs(sg - float(r["run2"]["sigma"])) / abs(float(r["run2"]["sigma"])),
                          elbo_prior=abs(f3["elbo_prior"] - float(r["run3"]["elbo_prior"])) / max(abs(float(r["run3"]["elbo_prior"])), 1.0)))
            raw.setdefault(prec, []).append((f1["elbo"], f3["elbo_prior"]))
        dev[prec] = d
        print(prec, [{k: "%.1e" % v for k, v in e.items()} for e in d])
    for k_ in ("LADDER_DISABLE_HALO", "LADDER_DISABLE_BNSTATS"):
        os.environ.pop(k_, None)
    # Measured on MI355X (deviation from the float64 oracle, iterations 0..4):
    #   f32    elbo 5.7e-08 9.9e-05 5.2e-04 1.7e-03 7.7e-04   l1 3.9e-08 1.5e-04 1.5e-04 6.6e-03 8.6e-03   elbo_prior 6.2e-05 .. 5.9e-02
    #   f16x3  elbo 5.7e-08 3.3e-05 6.3e-04 8.5e-04 8.3e-04   l1 3.9e-08 4.9e-05 9.0e-04 4.0e-03 1.8e-03   elbo_prior 4.4e-05 .. 1.7e-02
    # Iteration 0 sees identical parameters: kernel accuracy alone.  From iteration 1 on the parameters went through Adam's sign-like
    # first steps (an element whose gradient is within rounding of zero moves by +-lr either way), which ANY fp32 build resolves
    # differently from float64: by iteration 3 the round-3 NATIVE fp32 build was 1.7e-3 / 6.6e-3 away.  Bars: iteration 0 at kernel
    # accuracy, iteration 1 at BASELINE's 1e-3, later iterations inside the CONDITIONING of this 5-iteration problem itself, measured in
    # round 4 on the float64 ORACLE alone (no GPU involved; profiles/r04_traj_variants.txt): the same oracle started from parameters
    # perturbed by 1e-7 relative (one fp32 rounding) leaves its own trajectory by 3.2e-5 / 2.1e-4 / 3.1e-3 / 7.6e-3 in ELBO at iterations
    # 1 / 2 / 3 / 4 (a second perturbation: 4.6e-5 / 6.3e-4 / 9.6e-4 / 4.8e-3), i.e. "parameters -> ELBO after k steps" amplifies a
    # rounding-sized difference ~10x per iteration.  The round-4 fp32 build (fused halo kernels, fp32 small-dense kernels: per-output error
    # 1e-8 mean / 1e-7 max against float64, as the round-1 kernels) measures 6e-8 / 1.0e-4 / 3.3e-3 / 1.3e-3 / 2.9e-2; five differently
    # rounded fp32-class builds spread by up to 2.7e-2 at iteration 4 among THEMSELVES.  A bar below that spread tests the dice, not the kernels.
    for i in range(n_it):
        for prec in ("f32", "f16x3"):
            d = dev[prec][i]
            if i == 0:
                assert d["elbo"] < 2e-5 and d["l1"] < 2e-5 and d["elbo_prior"] < 2e-4, (prec, i, d)
            elif i == 1:
                assert d["elbo"] < 1e-3 and d["l1"] < 1e-3 and d["elbo_prior"] < 2e-3, (prec, i, d)
            else:
                assert d["elbo"] < 8e-2 and d["l1"] < 8e-2 and d["elbo_prior"] < 0.2, (prec, i, d)
            assert d["sigma"] < 4e-3, (prec, i, d)      # (fetched by RUN#2, i.e. after RUN#1's Adam step, in every iteration; measured <= 2.1e-3)
        if i <= 1:
            for k in ("elbo", "l1"):
                assert dev["f16x3"][i][k] <= 3 * dev["f32"][i][k] + 1e-4, (i, k, dev["f16x3"][i][k], dev["f32"][i][k])
    # ADVICE r4 (medium) / VERDICT r5: the late-iteration oracle bars above (8e-2) cannot see a 1e-2 regression of the fp32 kernels that are now
    # the headline.  Two strict-fp32 BUILDS of the same arithmetic stay much closer to each other than either stays to float64: the default
    # (projected pairs, halo kernels, fused statistics) against the generic gather-kernel build, iteration by iteration.
    pair = [(abs(a[0] - b[0]) / abs(b[0]), abs(a[1] - b[1]) / max(abs(b[1]), 1.0)) for a, b in zip(raw["f32"], raw["f32-generic"])]
    print("f32 default vs f32 generic (elbo, elbo_prior):", [("%.1e" % a, "%.1e" % b) for a, b in pair])
    # measured on MI355X (round 6): elbo 0 / 1.2e-5 / 2.8e-4 / 2.0e-3 / 8.1e-3, elbo_prior 7.1e-5 / 2.0e-3 / 4.1e-3 / 3.9e-2 / 8.6e-2 -- the ~10x per iteration
    # amplification of a rounding-sized difference described above, starting from IDENTICAL values at iteration 0.  Bars = 4-5x the measurement:
    # a kernel regression of 1e-3 shows at iterations 0-2, where the oracle bars above are blind to it.
    for i, (de, dp) in enumerate(pair):
        assert de < (2e-6, 1e-4, 1.5e-3, 1e-2, 4e-2)[i], (i, de)
        assert dp < (5e-4, 1e-2, 2e-2, 0.15, 0.35)[i], (i, dp)


TRAJ_WORKER = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from ladder_latent_data_distribution_modelling_amd import engine as E
prec, steps, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = json.load(open(os.path.join(%(root)r, "codes", "celeba_config.json")))
cfg["matmul_precision"] = prec
B = cfg["batch_size"]
eng = E.LadderEngine(cfg, "cuda:0", seed=1, noise_seed=99)
fix = np.load(os.path.join(%(root)r, "tests", "golden", "GM_prior_info.npz"))
K = cfg["n_mixtures"]
eng.set_mixture(fix["w_full"][:K] / fix["w_full"][:K].sum(), fix["m_full"][:K], fix["K_full"][:K])
xs = [torch.rand(B, 128, 128, 3, generator=torch.Generator().manual_seed(5 + i)).cuda() for i in range(4)]
rec = dict(elbo=[], elbo_prior=[], grad_norm=[], sigma=[])
for it in range(steps):
    x = xs[it %% 4]
    eng.run_ae(x, 2.5e-4, None, False, False)
    rec["elbo"].append(eng.fetch()["elbo"])
    rec["grad_norm"].append(float(eng.ps.grad["ae"].double().norm()))
    eng.run_sigma(x, 2.5e-4, None, False, False, reuse_encoder=False)
    rec["sigma"].append(eng.fetch(["sigma"])["sigma"])
    eng.run_prior(x, 1.25e-4, None, False, False, reuse_encoder=True)
    rec["elbo_prior"].append(eng.fetch()["elbo_prior"])
    eng.run_inner_sigma(x, 2e-4, None, False, False, reuse_encoder=True)
json.dump(rec, open(outp, "w"))
'''


def test_200_iteration_trajectory_f16x3_vs_f32_full_size(tmp_path):
    steps = 200
    script = tmp_path / "traj_worker.py"
    script.write_text(TRAJ_WORKER % dict(root=ROOT))
    import time
    T = {}
    for prec in ("f32", "f16x3", "bf16x6"):
        outp = str(tmp_path / (prec + ".json"))
        t0 = time.time()
        n = steps if prec != "bf16x6" else 24           # (the bf16x6 leg is the yardstick of the EARLY windows only)
        p = subprocess.run([sys.executable, str(script), prec, str(n), outp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
        assert p.returncode == 0, p.stdout[-2000:]
        print("trajectory %s: %.1f s" % (prec, time.time() - t0))
        T[prec] = {k: np.asarray(v, np.float64) for k, v in json.load(open(outp)).items()}
    for prec, t in T.items():
        for k, v in t.items():
            assert np.isfinite(v).all() and len(v) == (steps if prec != "bf16x6" else 24), (prec, k)
    rel = lambda a, b: np.abs(a - b) / np.abs(b)
    e32 = T["f32"]["elbo"]
    d16, dbf = rel(T["f16x3"]["elbo"], e32), rel(T["bf16x6"]["elbo"], e32[:24])
    g16, gbf = rel(T["f16x3"]["grad_norm"], T["f32"]["grad_norm"]), rel(T["bf16x6"]["grad_norm"], T["f32"]["grad_norm"][:24])
    print("ELBO deviation from f32, max over iterations 0-4 / 5-19 / 20-199: f16x3 %.1e %.1e %.1e | bf16x6 %.1e %.1e (24 iterations)" % (
        d16[:5].max(), d16[5:20].max(), d16[20:].max(), dbf[:5].max(), dbf[5:20].max()))
    print("gradient-norm deviation, iteration 0 / max 0-4: f16x3 %.1e %.1e | bf16x6 %.1e %.1e" % (g16[0], g16[:5].max(), gbf[0], gbf[:5].max()))
    # iteration 0: identical parameters -- kernel rounding only
    assert d16[0] < 2e-6 and g16[0] < 3e-4        # (round 5: 1.2e-4 -- the fp32 build's conv2d_5 / conv2d_4 run on effective taps, the f16x3 build's do not)
    # early window: still rounding-sized, and f16x3 behaves like the other fp32-class split format (same planes-and-products scheme
    # with 24-bit operands): within 4x of its deviation (+ a floor), and small in absolute terms
    # measured: ELBO deviation from f32 over iterations 0-4 / 5-19 / 20-199: f16x3 1.4e-3 / 4.1e-3 / 1.0e-1, bf16x6 2.7e-4 / 5.3e-3 / 4.8e-2;
    # gradient norm at iteration 0: f16x3 4.7e-5, bf16x6 7.6e-6 (two 11-bit planes and 3 products leave ~5x the rounding noise of three
    # 8-bit planes and 6 products -- DESIGN 4a states it; both are far inside the 1e-3 ELBO bar per iteration)
    assert d16[:5].max() < 5e-3 and d16[:5].max() <= 8 * dbf[:5].max() + 5e-4, (d16[:5].max(), dbf[:5].max())
    assert d16[5:20].max() < 3e-2 and d16[5:20].max() <= 4 * dbf[5:20].max() + 3e-3, (d16[5:20].max(), dbf[5:20].max())
    # late window: decorrelated trajectories of the same training run -- the statistics agree
    m32, m16 = e32[-50:].mean(), T["f16x3"]["elbo"][-50:].mean()
    print("mean ELBO of the last 50 iterations: f32 %.1f f16x3 %.1f" % (m32, m16))
    assert abs(m16 - m32) / abs(m32) < 0.05, (m32, m16)
    for prec in ("f32", "f16x3"):
        e = T[prec]["elbo"]
        assert e[-50:].mean() > e[1:11].mean(), prec                       # the ELBO of the 4 minibatches improved in every build
        assert e[-50:].mean() > 0.5 * e[0], prec                           # ... by a lot (it starts at -7.4e4 and reaches ~-2.3e4)
    s32, s16 = T["f32"]["sigma"], T["f16x3"]["sigma"]
    assert rel(s16, s32)[:20].max() < 2e-2
