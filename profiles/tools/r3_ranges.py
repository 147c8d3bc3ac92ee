"""Round 3: measured dynamic ranges of every tensor the f16x3 split kernels consume during one full-size CelebA training step
(B=128, codes/celeba_config.json) -> gpurun_out/r03_f16x3_ranges.json / .md.   python scratch/r3_ranges.py [config]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ladder_latent_data_distribution_modelling_amd import engine as E

cfgp = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "codes", "celeba_config.json")
cfg = json.load(open(cfgp))
cfg["matmul_precision"] = "f16x3"
B = int(os.environ.get("B", cfg["batch_size"]))
eng = E.LadderEngine(cfg, "cuda:0", seed=1, noise_seed=99)
fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
K, R = cfg["n_mixtures"], cfg["representation_size"]
if R == 2:
    eng.set_mixture(fix["w_full"][:K] / fix["w_full"][:K].sum(), fix["m_full"][:K], fix["K_full"][:K])
else:
    rng = np.random.default_rng(3); A = rng.normal(0, 0.3, (K, R, R))
    eng.set_mixture(rng.dirichlet(np.ones(K)), rng.normal(0, 1.5, (K, R)), A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))
x = torch.rand(B, 128, 128, 3, generator=torch.Generator().manual_seed(5)).cuda()
for _ in range(int(os.environ.get("PRE", 3))):                      # a few steps first: gradients after the sign-like first Adam steps
    eng.run_ae(x, 2.5e-4, None, False, False)

rows = []
orig = E.Ctx.absmax


def stats(t, who, role):
    if isinstance(t, E.PlanesOnly) or t.dim() != 4:
        return
    a = t.detach().abs().double()
    mx = float(a.max())
    if mx == 0:
        return
    N, H, W, C = a.shape
    l2 = torch.floor(torch.log2(torch.clamp(a / mx, min=2.0 ** -60)))
    hist = torch.bincount((-l2).clamp(0, 60).long().flatten(), minlength=61).cpu().numpy()
    nz = float((a == 0).double().mean())
    tot = float(a.sum())
    mass_below = {k: float(a[a < mx * 2.0 ** -k].sum() / tot) for k in (12, 16, 20, 24)}
    frac_below = {k: float((a < mx * 2.0 ** -k).double().mean()) for k in (12, 16, 20, 24)}
    smax = a.amax(dim=(1, 2, 3))
    per_sample = float(torch.log2(smax.min() / mx))
    th, tw = (16, 32) if (H >= 16 and W >= 32) else (H, W)
    tmax = a.reshape(N, H // th, th, W // tw, tw, C).amax(dim=(2, 4, 5)).flatten()
    tl = torch.log2(torch.clamp(tmax / mx, min=2.0 ** -60))
    pmax = a.amax(dim=3).flatten()                         # per pixel (over channels): what one output's K-sum sees at least
    pl = torch.log2(torch.clamp(pmax / mx, min=2.0 ** -60))
    cmax = a.amax(dim=(0, 1, 2))
    cl = torch.log2(torch.clamp(cmax / mx, min=2.0 ** -60))
    rows.append(dict(layer=who, role=role, shape=[N, H, W, C], absmax=mx, zero_frac=nz, frac_below=frac_below, mass_below=mass_below,
                     log2_min_sample_max=per_sample, log2_tile_max_min=float(tl.min()), log2_tile_max_p1=float(tl.quantile(0.01)),
                     log2_pixel_max_min=float(pl.min()), log2_pixel_max_p01=float(pl.quantile(0.001)) if pl.numel() < 16e6 else float(pl[::8].quantile(0.001)),
                     log2_channel_max_min=float(cl.min()), hist_log2_below_max=hist.tolist()))


def patched(self, t):
    import sys as _s
    fr = _s._getframe(1)
    me = fr.f_locals.get("self")
    who = getattr(me, "name", type(me).__name__)
    code = fr.f_code.co_name
    role = "?"
    for nm in ("x", "dy"):
        if fr.f_locals.get(nm) is t:
            role = nm
    stats(t, "%s.%s" % (who, code), role)
    return orig(self, t)


E.Ctx.absmax = patched
eng.run_ae(x, 2.5e-4, None, False, False)
torch.cuda.synchronize()
E.Ctx.absmax = orig
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = os.path.basename(cfgp).replace("_config.json", "")
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r03_f16x3_ranges_%s.json" % tag), "w"))
with open(os.path.join(ROOT, "gpurun_out", "r03_f16x3_ranges_%s.md" % tag), "w") as fh:
    fh.write("# Dynamic range of the operands of the f16x3 split kernels, one RUN#1 of %s at B=%d (after %s warm-up steps)\n\n" % (tag, B, os.environ.get("PRE", 3)))
    fh.write("log2 columns are relative to the tensor's absolute maximum (the per-tensor f16x3 scale keeps fp32-like relative precision down to -16).\n\n")
    fh.write("| consumer | operand | shape | max | zeros | elems < 2^-16 max | L1 mass < 2^-16 max | L1 mass < 2^-12 | min sample max | min 16x32-tile max | p1 tile max | min pixel max | p0.1 pixel max | min channel max |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write("| %s | %s | %s | %.3g | %.3f | %.2e | %.2e | %.2e | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |\n" % (
            r["layer"], r["role"], "x".join(map(str, r["shape"])), r["absmax"], r["zero_frac"], r["frac_below"][16], r["mass_below"][16],
            r["mass_below"][12], r["log2_min_sample_max"], r["log2_tile_max_min"], r["log2_tile_max_p1"], r["log2_pixel_max_min"],
            r["log2_pixel_max_p01"], r["log2_channel_max_min"]))
print(open(os.path.join(ROOT, "gpurun_out", "r03_f16x3_ranges_%s.md" % tag)).read())
