"""Round 3: N-step full-size trajectory (B=128, device noise, fixed batch) in one precision mode -> gpurun_out/r03_traj_<tag>.json.
   python scratch/r3_traj.py <tag> <precision> [steps]   (env LADDER_DISABLE_HALO=1 for the generic fp32 kernels)"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ladder_latent_data_distribution_modelling_amd import engine as E
tag, prec = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = json.load(open(os.path.join(ROOT, "codes", os.environ.get("CFG", "celeba_config.json"))))
cfg["matmul_precision"] = prec
B = int(os.environ.get("B", cfg["batch_size"]))
eng = E.LadderEngine(cfg, "cuda:0", seed=1, noise_seed=99)
fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
K = cfg["n_mixtures"]
eng.set_mixture(fix["w_full"][:K] / fix["w_full"][:K].sum(), fix["m_full"][:K], fix["K_full"][:K])
nb = int(os.environ.get("NBATCH", 4))
xs = [torch.rand(B, 128, 128, 3, generator=torch.Generator().manual_seed(5 + i)).cuda() for i in range(nb)]
rec = dict(elbo=[], elbo_prior=[], grad_norm=[], sigma=[], l1=[])
t0 = time.time()
for it in range(steps):
    x = xs[it % nb]
    eng.run_ae(x, 2.5e-4, None, False, False)
    f = eng.fetch()
    rec["elbo"].append(f["elbo"]); rec["l1"].append(f["l1_reconstruction_error"])
    rec["grad_norm"].append(float(eng.ps.grad["ae"].double().norm()))
    eng.run_sigma(x, 2.5e-4, None, False, False, reuse_encoder=False)
    rec["sigma"].append(eng.fetch(["sigma"])["sigma"])
    eng.run_prior(x, 1.25e-4, None, False, False, reuse_encoder=True)
    rec["elbo_prior"].append(eng.fetch()["elbo_prior"])
    eng.run_inner_sigma(x, 2e-4, None, False, False, reuse_encoder=True)
rec["seconds"] = time.time() - t0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "r03_traj_%s.json" % tag), "w"))
print(tag, prec, "elbo first/last", rec["elbo"][0], rec["elbo"][-1], "gn", rec["grad_norm"][0], rec["grad_norm"][-1], "%.1fs" % rec["seconds"])
