import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ladder_latent_data_distribution_modelling_amd import engine as E
cfg = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
B = cfg["batch_size"]
for graphs in (False, True):
    eng = E.LadderEngine(cfg, "cuda:0", seed=1, noise_seed=99)
    eng.use_graphs = graphs
    fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz")); K = cfg["n_mixtures"]
    eng.set_mixture(fix["w_full"][:K] / fix["w_full"][:K].sum(), fix["m_full"][:K], fix["K_full"][:K])
    x = torch.rand(B, 128, 128, 3, generator=torch.Generator().manual_seed(5)).cuda()
    def step(fetch):
        eng.run_ae(x, 2.5e-4, None, False, False)
        if fetch: eng.fetch()
        eng.run_sigma(x, 2.5e-4, None, False, False, reuse_encoder=False)
        if fetch: eng.fetch(["sigma"])
        eng.run_prior(x, 1.25e-4, None, False, False, reuse_encoder=True)
        if fetch: eng.fetch()
        eng.run_inner_sigma(x, 2e-4, None, False, False, reuse_encoder=True)
    for fetch in (True, False, True, False):
        for _ in range(10): step(fetch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): step(fetch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        print("graphs", graphs, "fetch", fetch, "%.3f ms/step  %.0f img/s" % (dt * 1e3, B / dt), flush=True)
    del eng
