"""Trainer base classes with the reference's step-function surface (codes/base.py:520-1010).

Each step function issues the same sequence of "runs" the reference issues as `sess.run` calls; a run
is evaluated by the HIP engine.  Plotting (matplotlib figures) is out of scope.
"""
import time

import numpy as np
import torch

from .models import BaseModel  # noqa: F401  (re-export, reference exposes BaseModel from codes.base)


class BaseTrain:
    def __init__(self, sess, model, data, config):
        self.model, self.config, self.sess, self.data = model, config, sess, data
        self.engine = model.engine
        # replay each run as one captured hipGraph after two eager warm-ups (config key `use_hip_graphs`, default on)
        # (default: on for the MNIST nets, whose ~350 short launches per iteration are host-bound in eager mode; off for CelebA, where
        # replay measured SLOWER than eager launches -- 21.6 against 21.0 ms per iteration -- and would rule out the stream overlap below)
        self.engine.use_graphs = bool(int(config.get("use_hip_graphs", 0 if config.get("exp_name") == "celeba" else 1)))
        # config key `overlap_prior_runs` (default 1): RUN#3 / RUN#4 on a second HIP stream beside RUN#2's decoder forward
        # (engine.enable_prior_overlap; bit-identical results; eager launches, one rank)
        self.engine.enable_prior_overlap(bool(int(config.get("overlap_prior_runs", 1))) and not self.engine.use_graphs)
        self.cur_epoch = 0
        # same record lists as codes/base.py:531-570
        for name in ("train_loss train_loss_prior val_loss val_loss_prior train_loss_ave_epoch val_loss_ave_epoch elbo_train "
                     "elbo_val recons_error_train recons_error_val entropy_z_train entropy_z_val crossEntropy_prior_train "
                     "crossEntropy_prior_val vampPrior_crossEntropy_prior_val vampPrior_crossEntropy_prior_train "
                     "sigma_reguarisor_train sigma_reguarisor_val code_elbo_train code_elbo_val entropy_t_train entropy_t_val "
                     "crossEntropy_t_train crossEntropy_t_val code_recons_error_train code_recons_error_val "
                     "code_recons_likelihood_train code_inner_sigma_train iter_epochs_list test_batch_code_mean "
                     "test_batch_code_std_dev test_sigma sigma_train classifier_accuracy gmm_mean gmm_cov gmm_weight").split():
            setattr(self, name, [])
        self.n_train_iter, self.n_val_iter = 0, 0
        self.gm_params = None          # (weights, means, covs) currently fed as the GM hyper-prior
        self.GM_prior_final = None
        # config key `async_fetch` (default 1): the train steps read the fetched scalars of a run through pinned-memory copies AFTER the
        # next run has been enqueued, so the GPU never idles while the host reads (engine.fetch_async).  The record lists of RUN#2 / RUN#3
        # (sigma_train, code_elbo_train, ...) are then completed one call later; flush() -- called by every evaluation / fit / save /
        # epoch-end path and by the last_fetch_* properties -- completes them at once.  0 restores a blocking fetch after every run.
        self.async_fetch = bool(int(config.get("async_fetch", 1)))
        self._pending = []             # [(fetch handle, callback)] in enqueue order
        self._last_fetch_sigma, self._last_fetch_prior = None, None

    def flush(self, wait=True):
        """Complete the record lists from every fetch still in flight (in order).  `wait=False`: only the leading fetches whose copy has
        already landed -- the host never blocks (the order of the record lists is kept: a fetch still in flight holds back the ones behind it)."""
        while self._pending:
            h, cb = self._pending[0]
            if not wait and not h.ready():
                break
            self._pending.pop(0)
            cb(h.get())

    @property
    def last_fetch_prior(self):
        self.flush()
        return self._last_fetch_prior

    @property
    def last_fetch_sigma(self):
        self.flush()
        return self._last_fetch_sigma

    def compute_execution_time(self, cur_epoch, total_epoch):
        self.current_time = time.time()
        elapsed = (self.current_time - self.start_time) / 60
        print("Already trained for {} min.".format(elapsed))
        remain = (self.current_time - self.start_time) / (cur_epoch + 1) * total_epoch / 60 - elapsed
        print("Remaining {} min.\n".format(remain))

    def draw_ellipse(self, position, covariance, weight, ax=None, color="r"):
        """The 2-sigma outline of one mixture component on a matplotlib axes (reference codes/base.py:825-841, used by the notebook's
        prior plots): a full 2x2 covariance gives the principal axes (SVD) and their angle, a length-2 diagonal an axis-aligned
        ellipse; line width = 10 x the component weight.  Returns the patch."""
        import matplotlib.pyplot as plt
        from matplotlib.patches import Ellipse
        ax = ax or plt.gca()
        covariance = np.asarray(covariance, dtype=np.float64)
        if covariance.shape == (2, 2):
            U, s, _ = np.linalg.svd(covariance)
            angle = float(np.degrees(np.arctan2(U[1, 0], U[0, 0])))
            width, height = 2.0 * np.sqrt(s)
        else:
            angle = 0.0
            width, height = 2.0 * np.sqrt(covariance)
        nsig = 2
        return ax.add_patch(Ellipse(tuple(np.asarray(position, dtype=np.float64)), nsig * float(width), nsig * float(height), angle=angle,
                                    color=color, fill=False, lw=float(weight) * 10))

    # ------------------------------------------------------------------ regime / feed (base.py:862-899)
    def compute_feeddict(self, batch_data=None, model_to_train=None):
        """Selects the feed regime: SG pre-training (dummy N(0,I) mixture, use_standard_gaussian_prior=True)
        vs the fitted mixture; use_mask from use_mask_start.  Returns (use_sg, use_mask)."""
        cfg, eng = self.config, self.engine
        if cfg["prior"] == "standard_gaussian":
            return True, False
        if cfg["prior"] in ("hierarchical", "vampPrior"):   # codes/base.py:901-911, 934-941: no mixture feed, no mask
            return self.cur_epoch <= int(cfg["sg_pretraining"]), False
        if cfg["prior"] == "GMM":                     # codes/base.py:912-933: dummy N(0,I) mixture in epoch 1, then the fit + 0.01 I
            if self.cur_epoch == 1 or self.gm_params is None:
                if getattr(self, "_fed", None) != "sg":
                    eng.set_sg_mixture()
                    self._fed = "sg"
            elif getattr(self, "_fed", None) is not self.gm_params:
                w, m, c = self.gm_params
                eng.set_mixture(w, m, np.asarray(c) + 0.01 * np.eye(int(cfg["code_size"])))
                self._fed = self.gm_params
            return False, False
        use_sg = self.cur_epoch <= int(cfg["sg_pretraining"])
        if use_sg or self.gm_params is None:
            if getattr(self, "_fed", None) != "sg":
                eng.set_sg_mixture()
                self._fed = "sg"
        elif getattr(self, "_fed", None) is not self.gm_params:
            eng.set_mixture(*self.gm_params)
            self._fed = self.gm_params
        return use_sg, self.cur_epoch >= int(cfg["use_mask_start"])

    # ------------------------------------------------------------------ step functions
    def train_step_ae(self, cur_lr, batch_data, noise=None):
        """RUN#1 (+ RUN#2 if TRAIN_sigma): codes/base.py:583-608.  `noise` (optional) = [noise_run1, noise_run2]."""
        eng, cfg = self.engine, self.config
        use_sg, use_mask = self.compute_feeddict(batch_data, "VAE")
        eng.run_ae(batch_data, cur_lr, noise[0] if noise else None, use_sg, use_mask)
        h1 = eng.fetch_async()
        if int(cfg["TRAIN_sigma"]) == 1:                # enqueued BEFORE RUN#1's values are read: the GPU works on it meanwhile
            lr_s = float(cfg["learning_rate_sigma"]) * (0.99 ** (self.cur_epoch - 1))
            eng.run_sigma(batch_data, lr_s, noise[1] if noise else None, use_sg, use_mask)
            self._enc_batch = batch_data      # RUN#2 evaluated the encoder with the post-RUN#1 weights on this batch
            self._pending.append((eng.fetch_async(["sigma"]), self._record_sigma))
        f = h1.get()
        self.recons_error_train.append(f["l1_reconstruction_error"])
        self.entropy_z_train.append(f["entropy_z"])
        self.crossEntropy_prior_train.append(f["crossEntropy_prior"])
        self.sigma_reguarisor_train.append(f["sigma_regularisor"])
        self.elbo_train.append(f["elbo"])
        self.last_fetch_ae = f
        if not self.async_fetch:
            self.flush()
        return f["loss_ae"]

    def _record_sigma(self, f):
        self._last_fetch_sigma = f
        self.sigma_train.append(f["sigma"])

    def _record_prior(self, f):
        self._last_fetch_prior = f
        if self.config["prior"] == "vampPrior":               # base.py:629-634: loss_prior = -elbo, crossEntropy_prior
            self.train_loss_prior.append(f["loss_ae"])
            self.vampPrior_crossEntropy_prior_train.append(f["crossEntropy_prior"])
            return
        self.code_recons_error_train.append(f["code_l1_reconstruction_error"])
        self.code_recons_likelihood_train.append(f["code_reconstruction_likelihood"])
        self.entropy_t_train.append(f["entropy_t"])
        self.crossEntropy_t_train.append(f["crossEntropy_representation"])
        self.code_elbo_train.append(f["elbo_prior"])
        self.code_inner_sigma_train.append(f["inner_sigma"])

    def train_step_prior(self, batch_data, noise=None):
        """RUN#3 (+ RUN#4 if TRAIN_inner_sigma): codes/base.py:610-641."""
        eng, cfg = self.engine, self.config
        use_sg, use_mask = self.compute_feeddict(batch_data, "prior")
        lr_p = float(cfg["learning_rate_prior"]) * (1.01 ** (self.cur_epoch - 1))
        # RUN#3/#4 see the same encoder weights as RUN#2 (only sigma / prior variables changed since): reuse its output
        reuse = getattr(self, "_enc_batch", None) is batch_data
        eng.run_prior(batch_data, lr_p, noise[0] if noise else None, use_sg, use_mask, reuse_encoder=reuse)
        self._enc_batch = batch_data
        h3 = eng.fetch_async()
        if cfg["prior"] != "vampPrior" and int(cfg["TRAIN_inner_sigma"]) == 1:
            lr_i = float(cfg["learning_rate_inner_sigma"]) * (1.01 ** (self.cur_epoch - 1))
            eng.run_inner_sigma(batch_data, lr_i, noise[1] if noise else None, use_sg, use_mask, reuse_encoder=True)
        # RUN#2's sigma and earlier iterations' values -- whatever has landed.  Never a blocking wait here: with the prior runs on the aux
        # stream RUN#2 is still executing beside them, and a host stall now would delay the enqueue of the next RUN#1
        self.flush(wait=False)
        self._pending.append((h3, self._record_prior))
        if not self.async_fetch:
            self.flush()

    def val_step(self, model_to_train, batch_data, noise=None):
        """codes/base.py:643-679."""
        self.flush()
        self._enc_batch = None          # this call overwrites the engine's encoder cache with another batch
        use_sg, use_mask = self.compute_feeddict(batch_data, model_to_train)
        self.engine.evaluate(batch_data, noise, use_sg, use_mask)
        f = self.engine.fetch()
        if model_to_train == "VAE":
            self.val_loss.append(f["loss_ae"])
            self.recons_error_val.append(f["l1_reconstruction_error"])
            self.entropy_z_val.append(f["entropy_z"])
            self.elbo_val.append(f["elbo"])
            self.crossEntropy_prior_val.append(f["crossEntropy_prior"])
            return f["loss_ae"]
        if self.config["prior"] == "vampPrior":       # base.py:674-677
            self.val_loss_prior.append(f["loss_ae"])
            self.vampPrior_crossEntropy_prior_val.append(f["crossEntropy_prior"])
            return f["loss_ae"]
        self.val_loss_prior.append(f["loss_prior"])
        self.code_recons_error_val.append(f["code_l1_reconstruction_error"])
        self.entropy_t_val.append(f["entropy_t"])
        self.code_elbo_val.append(f["elbo_prior"])
        self.crossEntropy_t_val.append(f["crossEntropy_representation"])
        return f["loss_prior"]

    # ------------------------------------------------------------------ GM fit (base.py:681-789)
    def _sharded_fit(self, gm):
        """C5 as an all-reduce of the mixture's sufficient statistics (default under data parallelism with the device fit; config key
        `gm_fit_mode`: "allreduce_stats" | "replicated" = all-gather the samples and run the same deterministic fit on every rank)."""
        from .vbgmm import DeviceBayesianGaussianMixture
        return (self.engine.ctx.comm.on and isinstance(gm, DeviceBayesianGaussianMixture)
                and self.config.get("gm_fit_mode", "allreduce_stats") == "allreduce_stats")

    def _draw_t_samples(self, iterator, n_batch, space="t", gather=True):
        """t- (or z-) samples of n_batch minibatches as ONE device tensor [n, R]: every rank's, rank-major (rank 0's samples first), from
        ONE all-gather under data parallelism -- or, gather=False, this rank's only: the sharded fit exchanges statistics, not samples."""
        eng = self.engine
        chunks = []
        for _ in range(n_batch):
            t = eng.sample_representation(iterator.next()) if space == "t" else eng.sample_code(iterator.next())
            chunks.append(t.clone())
        t = torch.cat(chunks, 0)
        if eng.ctx.comm.on and gather:   # C5 ("replicated" mode): gather every rank's samples
            parts = [torch.empty_like(t) for _ in range(eng.ctx.comm.world)]
            eng.ctx.comm.dist.all_gather(parts, t.contiguous(), group=eng.ctx.comm.group)
            t = torch.cat(parts, 0)
        return t

    def _share_gm(self, gm):
        """sklearn backend: rank 0 fits on the host, every rank receives (weights, means, covs)."""
        eng = self.engine
        K, R = int(self.config["n_mixtures"]), int(self.config["representation_size"])
        buf = torch.zeros(K + K * R + K * R * R, dtype=torch.float64, device=eng.ctx.device)
        if eng.ctx.comm.rank == 0:
            flat = np.concatenate([gm.weights_.ravel(), gm.means_.ravel(), gm.covariances_.ravel()])
            buf.copy_(torch.as_tensor(flat))
        eng.ctx.comm.broadcast_(buf, 0)
        a = buf.cpu().numpy()
        return a[:K].copy(), a[K:K + K * R].reshape(K, R).copy(), a[K + K * R:].reshape(K, R, R).copy()

    def _fit(self, gm, samples):
        """-> (weights, means, covs) to feed.  Device backend: every rank runs the same deterministic fit on the same gathered
        samples (only a cold start's k-means labels come from rank 0), the parameters never leave the GPU.  sklearn backend:
        rank 0 fits on a host copy and broadcasts."""
        from .vbgmm import DeviceBayesianGaussianMixture
        if isinstance(gm, DeviceBayesianGaussianMixture):
            if self._sharded_fit(gm):
                gm.fit_sharded(samples, self.engine.ctx.comm)      # `samples` = this rank's: statistics are all-reduced per iteration
            else:
                gm.fit(samples)
            return gm.weights_dev, gm.means_dev, gm.covariances_dev
        if self.engine.ctx.comm.rank == 0:
            gm.fit(samples.cpu().numpy().astype(np.float64))
        return self._share_gm(gm)

    def fit_GMM_VI(self, iterator, mode="fast", space="t"):
        self.flush()
        self._enc_batch = None          # this call overwrites the engine's encoder cache with another batch
        if space == "z":
            return self._fit_GMM_z(iterator, mode)
        comm = self.engine.ctx.comm
        bs_global = int(self.config["batch_size"]) * comm.world
        rank0 = comm.rank == 0
        if mode == "fast":
            samples = self._draw_t_samples(iterator, 2000 // bs_global + 1, gather=not self._sharded_fit(self.model.GM_prior_training))
            self.gm_params = self._fit(self.model.GM_prior_training, samples)
            w = self.gm_params[0]
        else:
            kw = dict(n_components=int(self.config["n_mixtures"]), covariance_type="full", max_iter=2000,
                      n_init=int(self.config["GM_fit_restart"]), weight_concentration_prior_type="dirichlet_process",
                      weight_concentration_prior=0.1, warm_start=False, random_state=self.config.get("gm_random_state"))
            if self.config.get("gm_fit_backend", "hip") == "hip":
                from .vbgmm import DeviceBayesianGaussianMixture
                self.GM_prior_final = DeviceBayesianGaussianMixture(
                    device=self.engine.ctx.device, label_broadcast=(lambda t: comm.broadcast_(t, 0)) if comm.on else None, **kw)
            else:
                from sklearn.mixture import BayesianGaussianMixture
                self.GM_prior_final = BayesianGaussianMixture(**kw)
            samples = self._draw_t_samples(iterator, 20000 // bs_global + 1, gather=not self._sharded_fit(self.GM_prior_final))
            self.gm_final_params = self._fit(self.GM_prior_final, samples)
            gmf = self.GM_prior_final
            if rank0 or hasattr(gmf, "weights_dev"):
                w, m, K = np.asarray(gmf.weights_), np.asarray(gmf.means_), np.asarray(gmf.covariances_)   # float64, like sklearn's
            else:
                w, m, K = self.gm_final_params
            idx = np.flatnonzero(w >= 1e-2)
            if rank0:
                aw = w[idx] / np.sum(w[idx]) if len(idx) else w[idx]
                np.savez("{}GM_prior_info.npz".format(self.config["result_dir"]), w_active=aw, m_active=m[idx],
                         K_active=K[idx], w_full=w, m_full=m, K_full=K)          # same keys as base.py:772-777
                print("Final fitted prior saved.")
        w = w.cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
        print("There are {} active mixtures.".format(int(np.sum(w >= 1e-2))))
        if comm.on and self._sharded_fit(self.model.GM_prior_training if mode == "fast" else self.GM_prior_final):
            # the sharded fit kept every rank's samples local; the reference returns the set the mixture was fitted on (base.py:747,789):
            # ONE all-gather at the end (rank-major), off the fit's critical path
            parts = [torch.empty_like(samples) for _ in range(comm.world)]
            comm.dist.all_gather(parts, samples.contiguous(), group=comm.group)
            samples = torch.cat(parts, 0)
        return samples.cpu().numpy().astype(np.float64)

    def _fit_GMM_z(self, iterator, mode):
        """prior "GMM" (base.py:699-710, 749-789): sklearn EM mixture on z samples, host fit on rank 0 + broadcast."""
        from sklearn.mixture import GaussianMixture
        comm = self.engine.ctx.comm
        bs_global = int(self.config["batch_size"]) * comm.world
        samples = self._draw_t_samples(iterator, (2000 if mode == "fast" else 20000) // bs_global + 1, space="z")
        if mode == "fast":
            gm = self.model.GM_prior_training
        else:
            gm = self.GM_prior_final = GaussianMixture(n_components=int(self.config["n_mixtures"]), covariance_type="full",
                                                       max_iter=2000, n_init=1, warm_start=False)
        K, R = int(self.config["n_mixtures"]), int(self.config["code_size"])
        if comm.rank == 0:
            gm.fit(samples.cpu().numpy().astype(np.float64))
        buf = torch.zeros(K + K * R + K * R * R, dtype=torch.float64, device=self.engine.ctx.device)
        if comm.rank == 0:
            buf.copy_(torch.as_tensor(np.concatenate([gm.weights_.ravel(), gm.means_.ravel(), gm.covariances_.ravel()])))
        comm.broadcast_(buf, 0)
        a = buf.cpu().numpy()
        w, m, c = a[:K].copy(), a[K:K + K * R].reshape(K, R).copy(), a[K + K * R:].reshape(K, R, R).copy()
        if mode == "fast":
            self.gm_params = (w, m, c)
        else:
            self.gm_final_params = (w, m, c)
            idx = np.flatnonzero(w >= 1e-2)
            if comm.rank == 0:
                np.savez("{}GM_prior_info.npz".format(self.config["result_dir"]), w_active=w[idx] / np.sum(w[idx]) if len(idx) else w[idx],
                         m_active=m[idx], K_active=c[idx], w_full=w, m_full=m, K_full=c)
                print("Final fitted prior saved.")
        print("There are {} active mixtures.".format(int(np.sum(w >= 1e-2))))
        return samples.cpu().numpy().astype(np.float64)

    def save_variables_VAE(self):
        """<result_dir>/<exp_name>-result.npz with the reference's keys (codes/base.py:791-823)."""
        self.flush()
        if self.engine.ctx.comm.rank != 0:
            return
        file_name = "{}{}-result.npz".format(self.config["result_dir"], self.config["exp_name"])
        np.savez(file_name, iter_list_val=self.iter_epochs_list, n_train_iter=self.n_train_iter, n_val_iter=self.n_val_iter,
                 train_loss=self.train_loss, elbo_train=self.elbo_train, val_loss=self.val_loss, elbo_val=self.elbo_val,
                 train_loss_prior=self.train_loss_prior, val_loss_prior=self.val_loss_prior,
                 code_elbo_train=self.code_elbo_train, code_elbo_val=self.code_elbo_val,
                 recons_loss_train=self.recons_error_train, recons_loss_val=self.recons_error_val,
                 recons_loss_prior_train=self.code_recons_error_train, recons_loss_prior_val=self.code_recons_error_val,
                 entropy_z_train=self.entropy_z_train, entropy_z_val=self.entropy_z_val,
                 entropy_t_train=self.entropy_t_train, entropy_t_val=self.entropy_t_val,
                 crossentropy_z_train=self.crossEntropy_prior_train, crossentropy_z_val=self.crossEntropy_prior_val,
                 crossentropy_t_train=self.crossEntropy_t_train, crossentropy_t_val=self.crossEntropy_t_val,
                 vampPrior_crossEntropy_z_train_prior=self.vampPrior_crossEntropy_prior_train,
                 vampPrior_crossEntropy_z_val_prior=self.vampPrior_crossEntropy_prior_val,
                 sigma_regularisor_train=self.sigma_reguarisor_train, sigma_regularisor_val=self.sigma_reguarisor_val,
                 num_para_VAE=self.model.num_para_list, sigma=self.test_sigma)


class BaseTrain_joint(BaseTrain):
    def train(self):
        """codes/base.py:848-860."""
        self.start_time = time.time()
        for _ in range(int(self.config["num_epochs"])):
            self.train_epoch()
            self.model.save(self.sess, model="joint" if self.config["prior"] in ("ours", "hierarchical", "vampPrior") else "VAE")
            self.compute_execution_time(self.cur_epoch - 1, self.config["num_epochs"])

    def test_step(self, batch_data, print_result=False, noise=None):
        """codes/base.py:944-986."""
        self.flush()
        self._enc_batch = None          # this call overwrites the engine's encoder cache with another batch
        use_sg, use_mask = self.compute_feeddict(batch_data)
        eng = self.engine
        eng.evaluate(batch_data, noise, use_sg, use_mask)
        f = eng.fetch()
        self.output_test = np.squeeze(eng.xhat.cpu().numpy())
        if print_result:
            print("test loss: elbo: {:.4f}, recons_loss_l1: {:.4f}, entropy z: {:.4f}, cross entropy z: {:.4f}, "
                  "sigma_regularisor: {:.4f}".format(f["elbo"], f["l1_reconstruction_error"], f["entropy_z"],
                                                     f["crossEntropy_prior"], f["sigma_regularisor"]))
        self.test_sigma.append(f["sigma"])
        print("current sigma: mean: {:.7f}; pixel mean error: {:.7f}".format(f["sigma"], f["mean_pixel_error"]))
        if print_result:
            print("current z std: {}".format(eng.std_dev_code()))
            if eng.has_inner:
                print("current t std: {}".format(eng.std_dev_representation()))
                print("current inner VAE sigma: {}".format(f["inner_sigma"]))
                print("current code prediction error per channel: {}".format(f["mean_code_error"]))
        return f

    def fit_GM(self, iterator):
        """codes/base.py:988-999 (prior 'ours')."""
        if self.config["prior"] == "ours":
            self.fit_GMM_VI(iterator=iterator, mode="fast", space="t")
            if self.cur_epoch % int(self.config["accurate_fit"]) == 0 or self.cur_epoch == int(self.config["num_epochs"]):
                self.fit_GMM_VI(iterator=iterator, mode="accurate", space="t")
        elif self.config["prior"] == "GMM":               # base.py:1000-1010
            self.fit_GMM_VI(iterator=iterator, mode="fast" if self.cur_epoch < int(self.config["num_epochs"]) else "accurate",
                            space="z")

    def generate_samples_from_prior(self, n_sample=10):
        """Latent samples -> decoded images (codes/base.py:1064-1168, figure writing omitted)."""
        eng, cfg = self.engine, self.config
        n = n_sample ** 2
        rng = np.random.default_rng(self.cur_epoch)
        if cfg["prior"] == "ours" and self.cur_epoch > int(cfg["sg_pretraining"]) and self.gm_params is not None:
            accurate = (self.cur_epoch % int(cfg["accurate_fit"]) == 0 or self.cur_epoch == int(cfg["num_epochs"]))
            w, m, K = self.gm_final_params if (accurate and self.GM_prior_final is not None) else self.gm_params
            w, m, K = (np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a, np.float64) for a in (w, m, K))
            w = np.clip(w, 0, None)
            comp = rng.choice(len(w), size=n, p=w / w.sum())
            Lc = np.linalg.cholesky(K)
            t = m[comp] + np.einsum("nij,nj->ni", Lc[comp], rng.standard_normal((n, m.shape[1])))
            code = eng.decode_representation(t)
        elif cfg["prior"] == "GMM" and self.gm_params is not None:     # ancestral sample of the mixture on z
            w, m, K = (np.asarray(a, np.float64) for a in self.gm_params)
            comp = rng.choice(len(w), size=n, p=np.clip(w, 0, None) / np.clip(w, 0, None).sum())
            Lc = np.linalg.cholesky(K + 0.01 * np.eye(K.shape[-1]))
            code = m[comp] + np.einsum("nij,nj->ni", Lc[comp], rng.standard_normal((n, m.shape[1])))
        else:
            code = rng.standard_normal((n, int(cfg["code_size"])))
        self.generated_samples = eng.decode(code).cpu().numpy()
        return self.generated_samples
