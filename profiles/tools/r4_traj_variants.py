import json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import ladder_oracle as O
from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
cfg0 = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
cfg0["batch_size"] = B = 8
rng = np.random.default_rng(31)
x = rng.random((B, 128, 128, 3)).astype(np.float32)
P = O.init_params(cfg0, seed=9)
K = int(cfg0["n_mixtures"])
fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
gm = dict(weights=(fix["w_full"][:K] / fix["w_full"][:K].sum()).astype(np.float32), means=fix["m_full"][:K].astype(np.float32), covs=fix["K_full"][:K].astype(np.float32))
n_it = 7
noises = [[O.make_noise(cfg0, B, rng, np.float32) for _ in range(4)] for _ in range(n_it)]
epoch = int(cfg0["sg_pretraining"]) + 1
lr_ae = float(cfg0["learning_rate_ae"]); lr_s = float(cfg0["learning_rate_sigma"]) * 0.99 ** (epoch - 1)
lr_p = float(cfg0["learning_rate_prior"]) * 1.01 ** (epoch - 1); lr_i = float(cfg0["learning_rate_inner_sigma"]) * 1.01 ** (epoch - 1)
res = {}
from ladder_latent_data_distribution_modelling_amd import engine as E
_small, _asd = E.Dense._small, E.Conv2D._as_dense
for name, over, env in (("f32 generic", dict(matmul_precision="f32", upsample_fused_convs=0), {"LADDER_DISABLE_HALO": "1"}), ("f32 fused", dict(matmul_precision="f32"), {}),
                        ("f32 fused nodense", dict(matmul_precision="f32"), {}), ("f32 fused nobank", dict(matmul_precision="f32"), {}),
                        ("bf16x6", dict(matmul_precision="bf16x6"), {})):
    if name.endswith("nodense"):
        E.Dense._small = lambda self, M: False
        E.Conv2D._as_dense = lambda self, M: False
    else:
        E.Dense._small, E.Conv2D._as_dense = _small, _asd
    os.environ.pop("LADDER_DISABLE_HALO", None); os.environ.update(env)
    eng = LadderEngine(dict(cfg0, **over), "cuda:0", values=P, seed=1)
    eng.set_mixture(gm["weights"], gm["means"], gm["covs"])
    tr = []
    for i in range(n_it):
        eng.run_ae(x, lr_ae, noises[i][0], False, False); f1 = eng.fetch()
        eng.run_sigma(x, lr_s, noises[i][1], False, False)
        eng.run_prior(x, lr_p, noises[i][2], False, False)
        eng.run_inner_sigma(x, lr_i, noises[i][3], False, False)
        tr.append((f1["elbo"], f1["l1_reconstruction_error"]))
    res[name] = tr
    print(name, " ".join("%.6g" % e for e, _ in tr), flush=True)
base = res["f32 generic"]
for name, tr in res.items():
    print("%-12s rel dev from f32 generic: %s" % (name, " ".join("%.1e" % (abs(a[0] - b[0]) / abs(b[0])) for a, b in zip(tr, base))))
