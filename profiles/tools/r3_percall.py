"""Per-call table of one steady-state CelebA iteration: every C-ABI call the engine makes is bracketed by HIP events on the stream it
is enqueued on and listed with its entry point and integer arguments (the shapes), so time can be attributed to LAYERS -- which a kernel
trace, keyed by kernel name, cannot do.  The prior runs stay on the main stream here (overlap off) so that the durations add up.
usage: python3 profiles/tools/r3_percall.py [--config codes/celeba_config.json] [--top 60] > gpurun_out/percall.md"""
import argparse
import collections
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "codes", "celeba_config.json"))
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--precision", default=None, help="override the config's matmul_precision (f32 / f16x3 / bf16x6 / bf16x3)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from ladder_latent_data_distribution_modelling_amd import _lib as L
    from ladder_latent_data_distribution_modelling_amd.codes.models import CelebAModel_densenet
    from ladder_latent_data_distribution_modelling_amd.codes.base import BaseTrain_joint
    cfg = json.load(open(args.config))
    cfg.update(checkpoint_dir="/tmp/ladder_bench/", result_dir="/tmp/ladder_bench/", use_hip_graphs=0, overlap_prior_runs=0)
    if args.precision:
        cfg["matmul_precision"] = args.precision
    with contextlib.redirect_stdout(io.StringIO()):
        model = CelebAModel_densenet(cfg, device="cuda:0", seed=1)
    trainer = BaseTrain_joint(None, model, None, cfg)
    trainer.cur_epoch = int(cfg["sg_pretraining"]) + 1
    K, R = int(cfg["n_mixtures"]), int(cfg["representation_size"])
    rng = np.random.default_rng(3)
    A = rng.normal(0, 0.3, (K, R, R))
    trainer.gm_params = (rng.dirichlet(np.ones(K)), rng.normal(0, 1.5, (K, R)), A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))
    B = int(cfg["batch_size"])
    x = torch.as_tensor(np.random.default_rng(0).random((B, cfg["dim_input_x"], cfg["dim_input_y"], cfg["dim_input_channel"]), dtype=np.float32)).cuda()
    lr = float(cfg["learning_rate_ae"])

    def step():
        trainer.train_step_ae(cur_lr=lr, batch_data=x)
        trainer.train_step_prior(batch_data=x)

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    recs, real = [], L.call

    def traced(name, *a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        real(name, *a)
        e.record()
        # shapes: the small integers of the argument list (pointers, byte counts and the stream handle are > 2^20 or not ints)
        recs.append((name, tuple(v for v in a if isinstance(v, int) and not isinstance(v, bool) and 0 <= v < (1 << 20)), s, e))

    L.call = traced
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.iters):
        step()
    t1.record()
    torch.cuda.synchronize()
    L.call = real
    agg = collections.OrderedDict()
    for name, shp, s, e in recs:
        k = (name, shp)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += s.elapsed_time(e) * 1e3
    n = args.iters
    tot = sum(v[1] for v in agg.values()) / n
    print("%d C-ABI calls per iteration, %.2f ms in calls, %.2f ms wall per iteration (event-bracketed: slower than the bench)\n"
          % (len(recs) // n, tot / 1e3, t0.elapsed_time(t1) / n))
    print("| entry point | integer arguments | calls/iter | us/call | us/iter | % |\n|---|---|---|---|---|---|")
    for (name, shp), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print("| `%s` | %s | %.1f | %.1f | %.1f | %.2f |" % (name.replace("ladder_", ""), " ".join(map(str, shp)), c / n, us / c, us / n, 100 * us / n / tot))
    by = collections.defaultdict(float)
    for (name, _), (c, us) in agg.items():
        by[name] += us / n
    print("\n| entry point | us/iter | % |\n|---|---|---|")
    for name, us in sorted(by.items(), key=lambda kv: -kv[1])[:40]:
        print("| `%s` | %.1f | %.2f |" % (name.replace("ladder_", ""), us, 100 * us / tot))


if __name__ == "__main__":
    main()
