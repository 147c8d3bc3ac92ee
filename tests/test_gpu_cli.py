"""The reference's command line, unchanged: `python3 train.py --config <json>` (reference train.py:18-74) end to end on the GPU
with a shrunk copy of codes/mnist_digit_config.json, run from a scratch working directory exactly like a user would; a second
invocation restores the checkpoints the first one wrote (model.load before training, train.py:62-66)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_cli_drop_in(tmp_path):
    cfg = json.load(open(os.path.join(ROOT, "codes", "mnist_digit_config.json")))
    cfg.update(num_epochs=2, sg_pretraining=1, num_hidden_units=64, num_hidden_units_inner_VAE=32, n_layers_inner_VAE=2,
               n_mixtures=4, n_MC_samples=8, batch_size=64, accurate_fit=2, GM_fit_restart=1, synthetic_n_train=256,
               synthetic_n_val=640, data_path=str(tmp_path) + "/no-data-here/")
    cpath = str(tmp_path / "mnist_digit_config.json")
    json.dump(cfg, open(cpath, "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--config", cpath]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    exp = os.path.join(str(tmp_path), "experiments", "mnist_digit", "batch-64")
    (run,) = os.listdir(exp)                                  # prior-ours-64-<code>-<rep>-<act>-2-mixture-4 (utils.py:62-73)
    assert run.startswith("prior-ours-64-") and run.endswith("-mixture-4")
    ck, res = os.path.join(exp, run, "checkpoint"), os.path.join(exp, run, "result")
    for f in ("vae-model.index", "vae-model.data-00000-of-00001", "prior-model.index", "checkpoint"):
        assert os.path.isfile(os.path.join(ck, f)), f
    r = np.load(os.path.join(res, "mnist_digit-result.npz"))
    assert len(r["elbo_train"]) == 2 * (256 // 64) and np.isfinite(r["elbo_train"]).all()
    assert os.path.isfile(os.path.join(res, "GM_prior_info.npz"))
    assert "2/2:" in out.stdout and "Outer VAE model saved." in out.stdout
    # second run: the saved models are restored first (the "No ... model found" branch is NOT taken)
    out2 = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out2.returncode == 0, out2.stderr[-3000:]
    assert "Outer VAE model loaded." in out2.stdout and "Prior model loaded." in out2.stdout
    assert "Outer VAE model loaded." not in out.stdout
    # a broken config is reported the reference's way (bare except -> message -> exit 0, train.py:21-27)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "train.py")], cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert bad.returncode == 0 and "missing or invalid arguments" in bad.stdout
