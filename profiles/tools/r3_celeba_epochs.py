"""Multi-epoch CelebA training at BASELINE size through the reference-shaped objects (DataGenerator -> CelebAModel_densenet ->
CelebATrainer_joint_training.train_epoch, reference train.py:41-70 / trainers.py:141-198) on a synthetic split resident in HBM:
epoch 1 is standard-Gaussian pre-training, from its end on the mixture is fitted once per epoch (device VB-GMM) and the prior runs train.
Prints wall time per epoch (training + validation + fit), images/s over the training iterations, ELBO first/last, peak memory.
usage: python3 profiles/tools/r3_celeba_epochs.py [n_train=12800] [epochs=4] > gpurun_out/celeba_epochs.txt"""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ladder_latent_data_distribution_modelling_amd.codes.data_loader import DataGenerator
from ladder_latent_data_distribution_modelling_amd.codes.models import CelebAModel_densenet
from ladder_latent_data_distribution_modelling_amd.codes.trainers import CelebATrainer_joint_training

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = json.load(open(os.path.join(ROOT, "codes", "celeba_config.json")))
out = "/tmp/celeba_epochs/"
os.makedirs(out, exist_ok=True)
cfg.update(num_epochs=epochs, sg_pretraining=1, synthetic_n_train=n_train, data_path="/nonexistent/", result_dir=out, checkpoint_dir=out,
           summary_dir=out)
data = DataGenerator(cfg, None)
model = CelebAModel_densenet(cfg)
tr = CelebATrainer_joint_training(None, model, data, cfg)
B = int(cfg["batch_size"])
fit_s = [0.0]
_fit = tr.fit_GM


def timed_fit(*a, **k):                                     # (the per-epoch mixture fit, codes/base.py:988-999: "fast" every epoch, + "accurate" every accurate_fit-th)
    torch.cuda.synchronize()
    t = time.time()
    r = _fit(*a, **k)
    torch.cuda.synchronize()
    fit_s[0] += time.time() - t
    return r


tr.fit_GM = timed_fit
print("config: codes/celeba_config.json, batch %d, matmul_precision %s, synthetic split of %d training images (%d iterations / epoch, %d validation "
      "iterations), sg_pretraining = 1" % (B, tr.engine.precision if hasattr(tr.engine, "precision") else cfg.get("matmul_precision"), n_train,
                                            tr.n_train_iter, tr.n_val_iter), flush=True)
for e in range(epochs):
    torch.cuda.synchronize()
    fit_s[0] = 0.0
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):          # (the trainer prints the reference's per-epoch report)
        tr.train_epoch()
    torch.cuda.synchronize()
    dt = time.time() - t0
    el = np.asarray(tr.elbo_train[-tr.n_train_iter:], dtype=np.float64)
    gm = getattr(tr, "gm_params", None)
    acc = (e + 1) % int(cfg["accurate_fit"]) == 0 or e + 1 == epochs
    print("epoch %d: %.2f s wall, of which mixture fit %.2f s (%s) -> %.0f images/s over training + validation; elbo %.2f -> %.2f; mixture %s; "
          "peak memory %.1f GB" % (e + 1, dt, fit_s[0], "fast + accurate Dirichlet-process fit" if acc else "fast fit",
                                   tr.n_train_iter * B / (dt - fit_s[0]), el[0], el[-1],
                                   "fitted (%d components, max weight %.3f)" % (len(gm[0]), float(torch.as_tensor(gm[0]).max())) if gm is not None else "not yet",
                                   torch.cuda.max_memory_allocated() / 1e9), flush=True)
model.save(None, "joint")
fin = bool(np.isfinite(tr.elbo_train).all() and np.isfinite(tr.code_elbo_train).all())
print("checkpoint files:", sorted(f for f in os.listdir(out) if "model" in f or f == "checkpoint"))
print("all recorded ELBO / code-ELBO values finite:", fin)
sys.exit(0 if fin else 1)
