"""bench.py -- the reference's headline workload on MI355X: LaDDer training iterations, CelebA 128x128.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one minibatch = the reference's four sess.run's (codes/base.py:583-641) in
the post-pretraining regime (fitted-GM feed, prior + inner-sigma training on), through the trainer's own step functions.
Workload at N=1: BASELINE.json configs[2] (codes/celeba_config.json: CelebA 128x128, nh=512, z=64, R=2, K=30, B=128);
N>1: the same per-GPU batch on every rank (weak scaling, global batch N*128 = configs[3] at N=8) with RCCL all-reduces C1-C4;
`--config codes/celeba_r8k50_config.json` is the per-GPU leg of configs[4] (2-rung ladder, R=8, K=50).
Inputs are synthetic (x ~ U[0,1), seeded Glorot weights) and resident in HBM before the timed region starts.

Precision: the headline (`value`, `ms_per_step`, `dtype`, `roofline`, `sustained`) is STRICT fp32 -- the arithmetic of the reference
(codes/models.py:348,388; four fp32 Adam optimisers, codes/base.py:457-517): every contraction on v_mfma_f32_32x32x2_f32 /
v_mfma_f32_16x16x4_f32, bit-exact fp32 FMA chains.  The 16-bit split format f16x3 (fp32 operands as 2 scaled fp16 planes = 22 bits:
narrower than fp32) is an explicit opt-in (`"matmul_precision": "f16x3"`) and appears here only as the labelled extra `fast_f16x3`.

Timing: W warm-up steps, then `--repeats` (5) timed regions of EXACTLY K steps each, every region bracketed by
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over ranks, NO profiler events inside them; `value` is the MEDIAN
region (all regions are listed in `repeats_images_per_sec`).  The roofline's per-kernel durations come from ONE further region of K steps
with HIP events around every contraction launch; a `sustained` leg of >= 30 s of back-to-back steps follows (clock / power settle there).
`roofline.achieved` / `frac` count the FLOPs the kernel ISSUES (`effective`: the reference's operation count of the same launches).
`parity` (N = 1): the HIP engine evaluated on the FIRST iteration of the `cpu_baseline` leg -- the same 4 images, seeded parameters, noise draws and mixture -- and
the relative deviation of its `elbo` / `elbo_prior` from the fp32 CPU oracle's (north_star: identical inputs, same run; tolerance 1e-3).  `comm.measured_small_allreduce_us`
(N > 1): back-to-back 8 B / 352 B / 2 KB all-reduces of the job's backend, measured before the timed regions; the ring model's latency term is that 8-byte number.
Prints ONE JSON line on rank 0.  Exits non-zero when the number of ranks actually running differs from --gpus.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY 8(d), CelebA nh=512 z=64: forward 10.04 GFLOP / image (encoder 0.614 + decoder 9.420 + inner VAE 0.0044)
#   algorithmic: RUN#1 30.1 + RUN#2 10.0 + RUN#3 0.63 + RUN#4 0.62 = 41.4 GFLOP / image / iteration (what the reference's four sess.run evaluate)
#   executed:    RUN#3 and RUN#4 reuse RUN#2's encoder output (same minibatch, unchanged encoder weights: bit-identical):
#                2 x 0.614 GFLOP of the 41.4 are never issued -> 40.17
FLOP_PER_IMG = {"celeba": {"algorithmic": 41.4e9, "executed": 41.4e9 - 2 * 0.614e9}}
FP32_PEAK_TFLOPS = 157.3                # MI355X_MICROARCH.md: fp32 MFMA (= vector) dense peak
F16_PEAK_TFLOPS = 2516.6                # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz)
# matrix instructions issued per algorithmic fp32 multiply-add by each matmul_precision (csrc/convsplit.hip)
MFMA_PER_PRODUCT = {"f32": 1, "f16x3": 3, "bf16x3": 3, "bf16x6": 6}
TRAFFIC_FILES = {"f16x3": ("r03_pmc_traffic.json", "r02_pmc_traffic.json"), "f32": ("r06_f32_pmc_traffic.json", "r05_f32_pmc_traffic.json", "r04_f32_pmc_traffic.json")}


def cpu_baseline(cfg, gm, seconds_budget=20.0, threads=None):
    """The oracle (CPU restatement of the TF1 reference path, fp32) timed on a bounded sample of the same workload.
    Thread count: min(host cores, 32) -- torch-CPU convolutions at this batch size get slower, not faster, beyond
    that on the 256-thread GPU hosts (measured: 256 threads ran 100x slower than 8).
    Returns (baseline record, parity sample): the FIRST timed iteration starts from the seeded initial parameters with explicit noise, and its
    inputs + fetches are handed back so that the HIP engine can be run on exactly them in the same process (`parity` in the JSON line)."""
    import numpy as np
    import torch
    from oracle import ladder_oracle as O
    cores = threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    Bc = 4
    rng = np.random.default_rng(0)
    x = rng.random((Bc, cfg["dim_input_x"], cfg["dim_input_y"], cfg["dim_input_channel"])).astype(np.float32)
    params0 = O.init_params(cfg, seed=1)
    nrng = np.random.default_rng(2)
    epoch = int(cfg["sg_pretraining"]) + 1

    def one(st, xb):
        noises = [O.make_noise(cfg, xb.shape[0], nrng, np.float32) for _ in range(4)]
        t0 = time.time()
        fetch = O.train_iteration(st, xb, noises, gm, cur_epoch=epoch, lr_ae=cfg["learning_rate_ae"])
        return time.time() - t0, noises, fetch

    one(O.OracleState(cfg, dict(params0), np.float32), x[:1])   # warm-up (thread pool, allocator) on a throw-away state, not timed
    st = O.OracleState(cfg, dict(params0), np.float32)
    n_it, t_tot, sample = 0, 0.0, None
    while n_it < 1 or (t_tot < seconds_budget and n_it < 8):
        dt, noises, fetch = one(st, x)
        if sample is None:
            sample = dict(x=x, noises=noises, params=params0, epoch=epoch,
                          cpu={"elbo": float(fetch["run1"]["elbo"]), "elbo_prior": float(fetch["run3"]["elbo_prior"]),
                               "l1_reconstruction_error": float(fetch["run1"]["l1_reconstruction_error"])})
        t_tot += dt
        n_it += 1
    rec = dict(value=round(Bc * n_it / t_tot, 3), unit="images/sec", cores=cores, kind="port",
               sample="%d full 4-run iterations at batch %d of the same %s %dx%d nh=%d z=%d R=%d K=%d network "
                      "(oracle/ladder_oracle.py, torch-CPU fp32, %d threads)" % (
                          n_it, Bc, cfg["exp_name"], cfg["dim_input_x"], cfg["dim_input_y"], cfg["num_hidden_units"], cfg["code_size"],
                          cfg["representation_size"], cfg["n_mixtures"], cores))
    return rec, sample


def parity_on_cpu_sample(cfg, gm, sample, device):
    """north_star: "outputs match the reference CPU path on identical inputs ... in the same run".  The HIP engine evaluates the FIRST iteration of
    the CPU-baseline leg -- the same 4 images, the same seeded initial parameters, the same four noise draws, the same mixture -- and the JSON line
    carries the relative deviation of `elbo` (RUN#1) and `elbo_prior` (RUN#3) from the fp32 CPU oracle's values (tolerance 1e-3, BASELINE.json)."""
    import contextlib
    import io
    from ladder_latent_data_distribution_modelling_amd.engine import LadderEngine
    c4 = dict(cfg, batch_size=int(sample["x"].shape[0]))
    with contextlib.redirect_stdout(io.StringIO()):
        eng = LadderEngine(c4, device, values=sample["params"], seed=1)
    eng.set_mixture(gm["weights"], gm["means"], gm["covs"])
    e, x, nz = sample["epoch"], sample["x"], sample["noises"]
    use_sg, use_mask = e <= int(cfg["sg_pretraining"]), e >= int(cfg["use_mask_start"])
    eng.run_ae(x, float(cfg["learning_rate_ae"]), nz[0], use_sg, use_mask)
    f1 = eng.fetch()
    eng.run_sigma(x, float(cfg["learning_rate_sigma"]) * (0.99 ** (e - 1)), nz[1], use_sg, use_mask)
    eng.run_prior(x, float(cfg["learning_rate_prior"]) * (1.01 ** (e - 1)), nz[2], use_sg, use_mask)
    f3 = eng.fetch()
    gpu = {"elbo": f1["elbo"], "elbo_prior": f3["elbo_prior"], "l1_reconstruction_error": f1["l1_reconstruction_error"]}
    rel = {k: abs(gpu[k] - v) / max(abs(v), 1e-12) for k, v in sample["cpu"].items()}
    tol = 1e-3
    return {"elbo_rel_err": float("%.3e" % rel["elbo"]), "elbo_prior_rel_err": float("%.3e" % rel["elbo_prior"]),
            "l1_reconstruction_error_rel_err": float("%.3e" % rel["l1_reconstruction_error"]), "tolerance": tol,
            "ok": bool(rel["elbo"] <= tol and rel["elbo_prior"] <= tol), "gpu": gpu, "cpu": sample["cpu"],
            "sample": "iteration 0 of the cpu_baseline leg: the same %d images, seeded initial parameters, four explicit noise draws and mixture; "
                      "HIP engine (strict fp32) against oracle/ladder_oracle.py (torch-CPU fp32): RUN#1 elbo, RUN#3 elbo_prior" % x.shape[0]}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as fresh child processes through
    torch.distributed.run.  This parent has not touched the GPU (torch.cuda.device_count() does not initialise it on this image), and it
    never re-executes itself: it waits for the children and exits with their status."""
    import socket
    import torch
    one_dev = os.environ.get("LADDER_BENCH_SINGLE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not one_dev:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible: refusing to print a %d-GPU number under an "
                         "%d-GPU label\n" % (args.gpus, have, have, args.gpus))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def workload_index(cfg, world):
    if cfg["exp_name"] == "mnist_digit":
        return "0"
    if cfg["exp_name"] == "mnist_fashion":
        return "1"
    if int(cfg["representation_size"]) == 8 and int(cfg["n_mixtures"]) == 50:
        return "4" if world == 8 else "4, per-GPU leg at %d GPU(s)" % world
    return "2" if world == 1 else ("3" if world == 8 else "3, per-GPU leg at %d GPUs" % world)


def roofline_of(prof, precision, step_seconds, traffic_for=None):
    """The dominant contraction kernel (largest share of GPU time) of a profiled leg against its MFMA peak."""
    rows = [r for r in prof.values() if r.get("bound", "mfma") == "mfma"]
    if not rows:                                               # (the MNIST nets: no contraction launch large enough to be attributed)
        return None
    dom = max(rows, key=lambda r: r["total_ms"])
    split = "split" in dom["kernel"]
    nm = MFMA_PER_PRODUCT[precision] if split else 1
    # peak for the ALGORITHMIC (fp32) flops of the kernel: the dense MFMA peak of the instruction it issues divided by the number of
    # matrix instructions it needs per fp32 product (f16x3: 2516.6 / 3); native fp32 kernels: the fp32 MFMA peak
    peak = (F16_PEAK_TFLOPS / nm) if split else FP32_PEAK_TFLOPS
    traffic, tsrc = None, None
    if traffic_for:
        for fn in TRAFFIC_FILES.get(precision, ()):
            tpath = os.path.join(ROOT, "profiles", fn)
            if os.path.isfile(tpath):
                tj = json.load(open(tpath))
                if tj.get("kernel", "") in dom["kernel"]:
                    traffic = tj["hbm_bytes_per_launch"]
                    tsrc = ("static: profiles/%s -- two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; gfx950 corrections per "
                            "MI355X_MICROARCH.md) over this same command and launch mix; not re-measured in this run" % fn)
                break
    ex_tf = dom.get("executed_tflops", dom["tflops"])
    return {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(ex_tf, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(ex_tf / peak, 4),
            "achieved_basis": "FLOPs the kernel ISSUES per launch / its HIP-event duration (for the upsample-fused launches 25 / 36 of the "
                              "reference's count: `effective` below carries the reference's operation count of the same launches)",
            "effective": round(dom["tflops"], 2), "effective_frac": round(dom["tflops"] / peak, 4),
            "traffic": traffic, "traffic_source": tsrc,
            "peak_basis": ("dense %s MFMA peak %.1f TFLOP/s / %d matrix instructions per fp32 product" % (
                "fp16" if precision == "f16x3" else "bf16", F16_PEAK_TFLOPS, nm)) if split else "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
            "mfma_issued_tflops": round(ex_tf * nm, 1),
            "launches": dom["launches"], "avg_launch_ms": round(dom["avg_ms"], 4),
            "flop_per_launch": dom.get("executed_flops_per_launch", dom["flops_per_launch"]), "effective_flop_per_launch": dom["flops_per_launch"],
            "share_of_step_time": round(dom["total_ms"] / (1e3 * step_seconds), 3),
            # (every row on ISSUED FLOPs, like the dominant kernel; `effective` = the reference's operation count of the same launches, which
            # exceeds the peak where a fused launch issues 25 of 36 products -- VERDICT r4 weak #7)
            "other_kernels": [{"kernel": r["kernel"], "achieved": round(r.get("executed_tflops", r["tflops"]), 2), "frac": round(r.get("executed_tflops", r["tflops"]) / peak, 4),
                               "effective": round(r["tflops"], 2), "launches": r["launches"], "avg_launch_ms": round(r["avg_ms"], 4),
                               "share_of_step_time": round(r["total_ms"] / (1e3 * step_seconds), 3)}
                              for r in prof.values() if r is not dom and r.get("bound", "mfma") == "mfma"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; `value` is their median")
    ap.add_argument("--sustained-seconds", type=float, default=30.0, help="length of the sustained-throughput leg (0: skip)")
    ap.add_argument("--config", default=os.path.join(ROOT, "codes", "celeba_config.json"))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (default: config batch_size)")
    ap.add_argument("--precision", default="", help="matmul_precision override: f32 | f16x3 | bf16x6 | bf16x3 (default: the config's; the shipped configs and the engine default are f32)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override a config key (JSON value), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-compare", action="store_true", help="skip the labelled f16x3 extra leg after the headline")
    ap.add_argument("--graphs", type=int, default=-1,
                    help="0: eager launches, the dominant kernel is timed with HIP events INSIDE the timed region (default for "
                         "CelebA, where replay changes nothing); 1: replay each run as a captured hipGraph (default for the MNIST "
                         "configs, whose ~350 short launches per iteration are host-bound in eager mode: 2.87 -> 2.49 ms on digit); "
                         "the per-kernel profile then comes from a second, eager pass after the timed region")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                       # never returns

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d rank(s) are running: refusing to report under the wrong label\n"
                             % (args.gpus, world))
        sys.exit(2)
    # test hook (tests/test_gpu_model.py): all ranks on cuda:0 over gloo, to exercise the multi-rank path of this script on a
    # 1-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    one_dev = os.environ.get("LADDER_BENCH_SINGLE_DEVICE") == "1"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ones = torch.ones(1, device="cuda")
    if world > 1:
        dist.all_reduce(ones)
    ranks_seen = int(ones.item())
    if ranks_seen != args.gpus:
        sys.stderr.write("bench.py: all-reduce of ones saw %d ranks, --gpus %d\n" % (ranks_seen, args.gpus))
        sys.exit(2)

    # measured small-message all-reduce latency of THIS job's backend, before anything is timed (the ring model below used an
    # ASSUMED 5 us per hop): 8 B = the pure latency of a collective launch, 2 KB = a C2 batch-norm statistics record, 352 B ~ the C3 partials
    small_allreduce_us = None
    if world > 1:
        small_allreduce_us = {}
        for nbytes in (8, 352, 2048):
            t = torch.zeros(nbytes // 4, device="cuda")
            for _ in range(20):
                dist.all_reduce(t)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                dist.all_reduce(t)
            torch.cuda.synchronize()
            small_allreduce_us[str(nbytes)] = round((time.perf_counter() - t0) / 200 * 1e6, 2)

    from ladder_latent_data_distribution_modelling_amd import engine as E
    from ladder_latent_data_distribution_modelling_amd.codes.models import CelebAModel_densenet, MNISTModel_digit, MNISTModel_fashion
    from ladder_latent_data_distribution_modelling_amd.codes.base import BaseTrain_joint

    cfg = json.load(open(args.config))
    if args.batch:
        cfg["batch_size"] = args.batch
    if args.precision:
        cfg["matmul_precision"] = args.precision
    for kv in args.set:
        k, _, v = kv.partition("=")
        try:
            cfg[k] = json.loads(v)
        except ValueError:
            cfg[k] = v
    precision = str(cfg.get("matmul_precision", E.DEFAULT_PRECISION))
    cfg.setdefault("checkpoint_dir", "/tmp/ladder_bench/")
    cfg.setdefault("result_dir", "/tmp/ladder_bench/")
    B = int(cfg["batch_size"])
    Model = {"celeba": CelebAModel_densenet, "mnist_digit": MNISTModel_digit, "mnist_fashion": MNISTModel_fashion}[cfg["exp_name"]]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = Model(cfg, device="cuda:%d" % local, seed=1)
    cfg["use_hip_graphs"] = args.graphs if args.graphs >= 0 else (1 if cfg["exp_name"].startswith("mnist") else 0)
    trainer = BaseTrain_joint(None, model, None, cfg)
    trainer.cur_epoch = int(cfg["sg_pretraining"]) + 1          # post-pretraining regime: all four runs active
    fix = np.load(os.path.join(ROOT, "tests", "golden", "GM_prior_info.npz"))
    K, R = int(cfg["n_mixtures"]), int(cfg["representation_size"])
    if R == 2 and K <= 50:
        w = fix["w_full"][:K] / fix["w_full"][:K].sum()
        gm = dict(weights=w, means=fix["m_full"][:K], covs=fix["K_full"][:K])
    else:                                                       # m ~ N(0, 1.5^2), Sigma = A A^T / R + 0.05 I, w ~ Dirichlet(1)
        rng = np.random.default_rng(3)
        A = rng.normal(0, 0.3, (K, R, R))
        gm = dict(weights=rng.dirichlet(np.ones(K)), means=rng.normal(0, 1.5, (K, R)),
                  covs=A @ A.transpose(0, 2, 1) / R + 0.05 * np.eye(R))
    trainer.gm_params = (gm["weights"], gm["means"], gm["covs"])
    x = torch.as_tensor(np.random.default_rng(rank).random((B, cfg["dim_input_x"], cfg["dim_input_y"], cfg["dim_input_channel"]),
                                                          dtype=np.float32)).cuda()
    lr = float(cfg["learning_rate_ae"])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        """EXACTLY n calls bracketed by barrier + synchronize on both sides; seconds, MAX over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(trainer, steps, warmup, repeats, sustained_seconds, profile):
        """One precision leg under the full protocol: W warm-up steps, `repeats` timed regions of EXACTLY `steps` steps (NO profiler
        events inside them), then -- `profile` -- one more region of the same length with HIP events around the contraction launches
        (the roofline's per-kernel durations: live, same process, same state, same command), then the sustained leg."""
        graphs = trainer.engine.use_graphs and world == 1
        step = lambda: (trainer.train_step_ae(cur_lr=lr, batch_data=x), trainer.train_step_prior(batch_data=x))
        if graphs:                                              # set-up, not warm-up: capture the four run graphs first
            for _ in range(8):
                step()
            assert len(trainer.engine._graphs) >= 4, "hipGraph capture did not settle"
        for _ in range(warmup):
            step()
        dts = [timed(step, steps) for _ in range(max(1, repeats))]
        prof, prof_seconds = None, None
        comm = trainer.engine.ctx.comm
        if profile:
            if comm.on:
                comm.enable_trace(True)
            if graphs:                                          # per-kernel HIP events need individual launches: eager pass
                trainer.engine.use_graphs = False
            kprof = E.KernelProfiler()
            E.set_profiler(kprof)
            prof_seconds = timed(step, steps)
            prof = kprof.summary()
            E.set_profiler(None)
            comm_table = comm.trace_summary(steps) if comm.on else None
            comm.enable_trace(False)
            if graphs:
                trainer.engine.use_graphs = True
        dt = sorted(dts)[len(dts) // 2]                         # the median region
        res = dict(value=B * world * steps / dt, ms_per_step=1e3 * dt / steps, dts=dts, prof=prof, prof_seconds=prof_seconds, graphs=graphs,
                   comm=comm_table if profile else None,
                   profiled_region_images_per_sec=(B * world * steps / prof_seconds) if prof_seconds else None)
        if sustained_seconds > 0:
            n_s = max(steps, int(sustained_seconds / (dt / steps)) + 1)
            t_s = timed(step, n_s)
            res["sustained"] = {"seconds": round(t_s, 2), "steps": n_s, "images_per_sec": round(B * world * n_s / t_s, 2),
                                "ms_per_step": round(1e3 * t_s / n_s, 3)}
        return res

    def executed_flops(eng, fl):
        """FLOP model of a leg: the algorithmic 41.4 GF minus what is never issued (encoder reuse in RUN#3 / #4; 27 / 36 resp. 11 / 36 of the launches that fold the factor-2 resize)."""
        fl = dict(fl)
        note = "RUN#3/#4 reuse RUN#2's encoder output (bit-identical): 1.23 GFLOP of the algorithmic 41.4 are not executed"
        used = getattr(eng.ctx, "up2_used", {})
        if used:      # conv2d_7 (4.832 GFLOP / image forward), conv2d_6 (2.416), conv2d_5 (1.208), conv2d_4 (0.604) in RUN#1 (":train") and RUN#2, ":bwd", ":wgrad": 11 / 36 not issued
            per = {}
            for lname, gf in (("decoder/conv2d_7", 4.832e9), ("decoder/conv2d_6", 2.416e9), ("decoder/conv2d_5", 1.208e9), ("decoder/conv2d_4", 0.604e9), ("decoder/conv2d_3", 0.302e9)):
                for sfx in ("", ":train", ":bwd", ":wgrad"):      # forward-only run, training forward, backward-data, filter gradient
                    per[lname + sfx] = gf
            skipped = getattr(eng.ctx, "up2_skipped", {})
            fl["executed"] -= sum(v * skipped.get(k, 11.0 / 36.0) for k, v in per.items() if k in used)
            note += ("; resize x2 -> 3x3 conv pairs computed from the low-resolution tensor (%s; forward, ':bwd' backward-data, ':wgrad' filter gradient): "
                     "projected form (nine 1x1 convolutions at low resolution + an elementwise combination) issues 9 of every 36 products of the "
                     "reference's count (1 of 16 behind the factor-4 resize), the tap-folded form 25 of 36" % ", ".join("%s %.4g/36" % (k, 36 * (1 - skipped.get(k, 11.0 / 36.0))) for k in sorted(used)))
        return dict(fl, note=note)

    head = measure(trainer, args.steps, args.warmup, args.repeats, args.sustained_seconds, not args.no_profile)
    value, dts, prof = head["value"], head["dts"], head["prof"]
    graphs = head["graphs"]
    f = trainer.last_fetch_ae
    exp_label = {"celeba": "CelebA", "mnist_digit": "MNIST-digit", "mnist_fashion": "MNIST-fashion"}[cfg["exp_name"]]
    out = {
        "metric": "training images/sec (full 4-run LaDDer iteration, %s %dx%d)" % (exp_label, cfg["dim_input_x"], cfg["dim_input_y"]),
        "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if precision == "f32" else "f32 (%s split MFMA)" % precision,
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[%s]: %s %dx%dx%d nh=%d z=%d R=%d K=%d L=%d per-GPU batch=%d, 4 runs/iteration "
                               "(AE step, sigma step, prior step, inner-sigma step), fitted-GM regime" % (
                                   workload_index(cfg, world), cfg["exp_name"], cfg["dim_input_x"], cfg["dim_input_y"], cfg["dim_input_channel"],
                                   cfg["num_hidden_units"], cfg["code_size"], R, K, cfg["n_MC_samples"], B),
                   "global_batch": B * world, "parallelism": "dp%d" % world},
        "timing": "median of %d timed regions of exactly %d steps (each: barrier + synchronize on both sides, MAX over ranks; no profiler "
                  "events inside them)" % (len(dts), args.steps),
        "repeats_images_per_sec": [round(B * world * args.steps / t, 2) for t in dts],
        "launch": "hipGraph replay (4 graphs/iteration)" if graphs else "eager", "matmul_precision": precision,
        "elbo": f["elbo"], "elbo_prior": trainer.last_fetch_prior["elbo_prior"],
    }
    if prof:
        rl = roofline_of(prof, precision, head["prof_seconds"], traffic_for=(cfg["exp_name"] == "celeba" and B == 128))
        if rl is not None:
            rl["measured_in"] = ("one further region of %d steps right after the timed regions, HIP events (hipEventRecord on the launch stream) around "
                                 "every contraction launch: %.1f images/s with the events on" % (args.steps, head["profiled_region_images_per_sec"]))
            out["roofline"] = rl
        mix = [r for r in prof.values() if r.get("bound") == "latency"]
        if mix:                                                 # the hyper-prior ELBO kernel (configs[4] stresses it): lane = component
            r = mix[0]
            out["mixture_kernel"] = {"kernel": r["kernel"], "launches": r["launches"], "avg_launch_us": round(1e3 * r["avg_ms"], 2),
                                     "component_evals_per_launch": int(cfg["n_MC_samples"]) * B * K,
                                     "component_gevals_per_sec": round(int(cfg["n_MC_samples"]) * B * K / (r["avg_ms"] * 1e-3) / 1e9, 2),
                                     "share_of_step_time": round(r["total_ms"] / (1e3 * head["prof_seconds"]), 5)}
    if head.get("comm"):
        # per exchange step (rank 0's view, HIP events on the compute stream in the profiled region): C1 gradient buckets, C2 batch-norm
        # statistics per layer, C3 ELBO partials, C4 prior gradients -- wall (issue -> compute stream may continue) and exposed time
        tot_w = sum(v["wall_us_per_step"] for v in head["comm"].values())
        tot_x = sum(v["exposed_us_per_step"] for v in head["comm"].values())
        out["comm"] = {"collectives": head["comm"], "wall_ms_per_step": round(tot_w / 1e3, 3), "exposed_ms_per_step": round(tot_x / 1e3, 3),
                       "backend": dist.get_backend(), "calls_per_step": round(sum(v["calls_per_step"] for v in head["comm"].values()), 1),
                       "note": "under data parallelism the engine keeps every run on one stream (no RUN#3 / RUN#4 overlap, eager launches): "
                               "`single_stream_cost` of the 1-GPU line says what that costs per rank"}
        # ... and what a ring all-reduce over xGMI SHOULD cost for the same calls (the first real multi-GPU run is read against
        # a model, not against nothing): per call 2 (N-1)/N x bytes over one 153 GB/s link direction (point-to-point xGMI, ring = per-link
        # bound: MI355X_MICROARCH / brief) + a fixed latency of 2 (N-1) hops x 5 us (stated assumption: small-message RCCL ring step on xGMI;
        # to be replaced by the first measured 8-byte all-reduce)
        LINK_GBS, HOP_US = 153.0, 5.0
        pred = {}
        lat_meas = small_allreduce_us.get("8") if small_allreduce_us else None
        out["comm"]["measured_small_allreduce_us"] = small_allreduce_us       # back-to-back 8 B / 352 B / 2 KB all-reduces of this backend, measured above
        for label, v in head["comm"].items():
            bw_us = 2.0 * (world - 1) / world * v["bytes_per_call"] / (LINK_GBS * 1e3)
            lat_us = lat_meas if lat_meas is not None else 2.0 * (world - 1) * HOP_US       # the MEASURED 8-byte latency replaces the assumed hop cost
            pred[label] = {"predicted_us_per_call": round(bw_us + lat_us, 1), "bandwidth_term_us": round(bw_us, 1), "latency_term_us": round(lat_us, 1),
                           "predicted_us_per_step": round((bw_us + lat_us) * v["calls_per_step"], 1)}
        out["comm"]["predicted"] = {"model": "ring all-reduce: 2 (N-1)/N x bytes / %.0f GB/s per xGMI link direction + %s per call" % (
                                        LINK_GBS, ("the measured 8-byte all-reduce latency (%.1f us)" % lat_meas) if lat_meas is not None else "2 (N-1) x %.0f us (assumed)" % HOP_US),
                                    "n_ranks": world, "collectives": pred,
                                    "wall_ms_per_step": round(sum(q["predicted_us_per_step"] for q in pred.values()) / 1e3, 3),
                                    "note": "the asynchronous C1 decoder bucket overlaps the encoder's backward pass: its predicted time is exposed only "
                                            "beyond that pass (~1.4 ms at batch 128)"}
    if "sustained" in head:
        out["sustained"] = head["sustained"]
    # also the AE-step-only (RUN#1) and forward-only (val_step, VAE fetches) rates; untimed extras after the metric
    n_x = max(3, args.steps // 2)
    use_sg, use_mask = trainer.compute_feeddict(x, "VAE")
    ae_only = lambda: (trainer.engine.run_ae(x, lr, None, use_sg, use_mask), trainer.engine.fetch())
    ae_only()
    out["ae_step_only_images_per_sec"] = round(B * world * n_x / timed(ae_only, n_x), 2)
    fwd_only = lambda: trainer.val_step("VAE", x)
    fwd_only()
    out["forward_only_images_per_sec"] = round(B * world * n_x / timed(fwd_only, n_x), 2)
    if world == 1 and trainer.engine._aux_on and not graphs:
        # what a data-parallel rank gives up (engine.enable_prior_overlap is off under DP: collectives stay on one stream)
        trainer.flush()
        trainer.engine.enable_prior_overlap(False)
        one = lambda: (trainer.train_step_ae(cur_lr=lr, batch_data=x), trainer.train_step_prior(batch_data=x))
        one()
        t1 = timed(one, n_x)
        trainer.engine.enable_prior_overlap(True)
        out["single_stream_cost"] = {"ms_per_step_without_prior_overlap": round(1e3 * t1 / n_x, 3), "ms_per_step": out["ms_per_step"],
                                     "note": "the schedule every rank of a data-parallel job runs (RUN#3 / RUN#4 behind RUN#2 on one stream)"}
    fl = FLOP_PER_IMG.get(cfg["exp_name"])
    full_size = bool(fl and int(cfg["num_hidden_units"]) == 512 and int(cfg["code_size"]) == 64)
    if full_size:
        fle = executed_flops(trainer.engine, fl)
        out["flop_per_image"] = fle
        out["whole_step_tflops_per_gpu"] = round(fle["executed"] * value / world / 1e12, 2)
        if precision == "f32":
            out["whole_step_frac_of_fp32_peak"] = round(out["whole_step_tflops_per_gpu"] / FP32_PEAK_TFLOPS, 4)
    if precision == "f32" and cfg["exp_name"] == "celeba" and not args.no_compare:
        # labelled EXTRA, never the headline: the same iteration with the contractions on the 16-bit matrix cores in the f16x3 split format
        # (fp32 operands cut to 2 scaled fp16 planes = 22 significand bits, 3 fp16 MFMAs per product: NARROWER than the reference's fp32;
        # opt-in through `"matmul_precision": "f16x3"`) -- thinner protocol (3 warm-up, 2 regions of steps/2)
        cfgx = dict(cfg, matmul_precision="f16x3")
        with contextlib.redirect_stdout(io.StringIO()):
            modelx = Model(cfgx, device="cuda:%d" % local, seed=1)
        trx = BaseTrain_joint(None, modelx, None, cfgx)
        trx.cur_epoch, trx.gm_params = trainer.cur_epoch, trainer.gm_params
        hx = measure(trx, n_x, 3, 2, 0, not args.no_profile)
        ex = {"images_per_sec": round(hx["value"], 2), "ms_per_step": round(hx["ms_per_step"], 3), "steps": n_x, "regions": 2,
              "matmul_precision": "f16x3", "dtype": "fp32 operands as 2 scaled fp16 planes (22 significand bits), 3 fp16 MFMAs per product, fp32 accumulate",
              "note": "opt-in fast path; narrower arithmetic than the reference's fp32 -- NOT the headline", "elbo": trx.last_fetch_ae["elbo"]}
        rlx = roofline_of(hx["prof"], "f16x3", hx["prof_seconds"], traffic_for=(B == 128)) if hx["prof"] else None
        if rlx is not None:
            ex["roofline"] = rlx
        if full_size:
            ex["whole_step_tflops_per_gpu"] = round(executed_flops(trx.engine, fl)["executed"] * hx["value"] / world / 1e12, 2)
        out["fast_f16x3"] = ex
        del trx, modelx
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], sample = cpu_baseline(cfg, gm)
        if cfg["prior"] == "ours":
            out["parity"] = parity_on_cpu_sample(cfg, gm, sample, "cuda:%d" % local)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
