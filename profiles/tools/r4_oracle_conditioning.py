import json, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import ladder_oracle as O
cfg0 = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
cfg0["batch_size"] = B = 8
rng = np.random.default_rng(31)
x = rng.random((B, 128, 128, 3)).astype(np.float32)
P = O.init_params(cfg0, seed=9)
K = int(cfg0["n_mixtures"])
fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
gm = dict(weights=(fix["w_full"][:K] / fix["w_full"][:K].sum()).astype(np.float32), means=fix["m_full"][:K].astype(np.float32), covs=fix["K_full"][:K].astype(np.float32))
n_it = 5
noises = [[O.make_noise(cfg0, B, rng, np.float32) for _ in range(4)] for _ in range(n_it)]
epoch = int(cfg0["sg_pretraining"]) + 1
lr_ae = float(cfg0["learning_rate_ae"])
out = {}
for tag, eps in (("exact", 0.0), ("pert 1e-7 a", 1e-7), ("pert 1e-7 b", 1e-7)):
    prng = np.random.default_rng(hash(tag) % 1000)
    Pp = {k: (np.asarray(v, np.float64) * (1 + eps * prng.standard_normal(np.shape(v)))) for k, v in P.items()}
    st = O.OracleState(cfg0, Pp, np.float64)
    t0 = time.time()
    ref = [O.train_iteration(st, x, noises[i], gm, cur_epoch=epoch, lr_ae=lr_ae) for i in range(n_it)]
    out[tag] = [float(r["run1"]["elbo"]) for r in ref]
    print(tag, out[tag], "%.0f s" % (time.time() - t0), flush=True)
for tag in out:
    print(tag, ["%.1e" % (abs(a - b) / abs(b)) for a, b in zip(out[tag], out["exact"])])
