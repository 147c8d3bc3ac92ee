"""Model classes with the reference's names and constructor signature (codes/models.py:10,163,330;
codes/base.py:32-124).  A model owns a LadderEngine (HIP kernels + parameters + optimiser slots) and the
sklearn mixture object that produces the hyper-prior feed (base.py:93-106).
"""
import os

import numpy as np

from ..engine import LadderEngine
from .. import arch


class BaseModel:
    exp_name = None

    def __init__(self, config, device=None, values=None, seed=1, comm=None):
        if self.exp_name is not None and config["exp_name"] != self.exp_name:
            raise ValueError("%s expects exp_name=%r, got %r" % (type(self).__name__, self.exp_name, config["exp_name"]))
        self.config = config
        if device is None:
            device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
        self.engine = LadderEngine(config, device, values, seed, comm)
        self.define_GM_prior()
        # codes/base.py:437-451: per-scope counts through the single-argument helper (this model becomes the default "graph")
        from .utils import count_trainable_variables, set_default_model
        set_default_model(self)
        self.num_encoder = count_trainable_variables("encoder")
        self.num_decoder = count_trainable_variables("decoder")
        self.num_sigma = count_trainable_variables("sigma")
        self.num_prior_ae = count_trainable_variables("prior") if config["prior"] in ("ours", "hierarchical", "vampPrior") else 0
        self.num_prior_sigma = count_trainable_variables("inner_sigma") if config["prior"] in ("ours", "hierarchical") else 0
        self.num_para_list = [self.num_encoder, self.num_decoder, self.num_sigma, self.num_prior_ae, self.num_prior_sigma]
        print("Total number of trainable parameters in VAE network is:\n{}k\n".format(np.around(sum(self.num_para_list) / 1000, 2)))
        self.init_saver()
        from .session import attach_handles
        attach_handles(self)          # graph-tensor attribute surface for `sess.run(fetches, feed_dict)` (codes/session.py)

    # codes/base.py:88-106 -- the producer of (prior_weight, prior_mean, prior_cov).  config["gm_fit_backend"]: "hip" (default)
    # runs the variational fit on the device (codes/vbgmm.py -> csrc/vbgmm.hip, sklearn-parity tested); "sklearn" keeps the
    # reference's host object.
    def define_GM_prior(self):
        self.GM_prior_training = None
        if self.config["prior"] == "ours":
            # (config key `gm_random_state`, optional: seeds the k-means initialisation of a cold fit; None = sklearn's default, as the reference)
            kw = dict(n_components=int(self.config["n_mixtures"]), covariance_type="full", max_iter=1000, n_init=1,
                      weight_concentration_prior_type="dirichlet_distribution", weight_concentration_prior=0.1, warm_start=True,
                      random_state=self.config.get("gm_random_state"))
            if self.config.get("gm_fit_backend", "hip") == "hip":
                from .vbgmm import DeviceBayesianGaussianMixture
                comm = self.engine.ctx.comm
                self.GM_prior_training = DeviceBayesianGaussianMixture(
                    device=self.engine.ctx.device, label_broadcast=(lambda t: comm.broadcast_(t, 0)) if comm.on else None, **kw)
            else:
                from sklearn.mixture import BayesianGaussianMixture
                self.GM_prior_training = BayesianGaussianMixture(**kw)
        elif self.config["prior"] == "GMM":          # base.py:101-106: plain EM mixture on z (R = code_size), host fit
            from sklearn.mixture import GaussianMixture
            self.GM_prior_training = GaussianMixture(n_components=int(self.config["n_mixtures"]), covariance_type="full",
                                                     max_iter=1000, n_init=1, warm_start=True)

    # codes/base.py:37-85 -- two savers: vae-model (encoder+decoder+sigma), prior-model (prior/* + inner sigma).
    # Adam slots / epoch counter are not saved by the reference either.  Format: the reference's own -- a TensorFlow
    # checkpoint-v2 bundle (<prefix>.index/.data-00000-of-00001/.meta + `checkpoint`) keyed by the TF variable names, written
    # and read by codes/tf_bundle.py without TensorFlow, so checkpoints are interchangeable with the reference in both
    # directions.  config["checkpoint_format"] = "npz" selects a plain numpy archive instead.
    def init_saver(self):
        self.saver_path_ae = os.path.join(self.config.get("checkpoint_dir", "."), "vae-model")
        self.saver_path_prior = os.path.join(self.config.get("checkpoint_dir", "."), "prior-model")

    def _save(self, path, groups):
        if self.engine.ctx.comm.rank != 0:
            return
        tensors = self.engine.ps.to_dict(groups)
        if self.config.get("checkpoint_format", "tf_bundle") == "npz":
            np.savez(path + ".npz", **tensors)
        else:
            from . import tf_bundle
            tf_bundle.save_checkpoint(path, tensors)

    def save(self, sess, model):
        print("Saving model...")
        if model == "VAE" or (model == "joint" and int(self.config["TRAIN_VAE"]) == 1):
            self._save(self.saver_path_ae, ("ae", "sigma"))
            print("Outer VAE model saved.")
        if self.config["prior"] in ("ours", "hierarchical", "vampPrior") and (
                model == "prior" or (model == "joint" and int(self.config["TRAIN_prior"]) == 1)):
            self._save(self.saver_path_prior, ("prior", "inner_sigma"))
            print("Prior model saved.")

    def load(self, sess, model):
        """Restores like saver.restore: every variable of the saver's list must be in the checkpoint with its shape
        (a mismatch raises, as TF's restore does); a missing checkpoint only prints (base.py:66-85)."""
        print("\ncheckpoint_dir to be loaded:\n{}\n".format(self.config.get("checkpoint_dir")))
        path = self.saver_path_ae if model == "VAE" else self.saver_path_prior
        groups = ("ae", "sigma") if model == "VAE" else ("prior", "inner_sigma")
        label = "Outer VAE" if model == "VAE" else "Prior"
        if os.path.isfile(path + ".index"):          # (the reference tests for the .meta file; the .index is what restore reads)
            from . import tf_bundle
            values = tf_bundle.load_checkpoint(path)
        elif os.path.isfile(path + ".npz"):
            values = dict(np.load(path + ".npz"))
        else:
            print("No %s model found. No %s model loaded." % (label.lower(), "VAE" if model == "VAE" else "prior"))
            return
        ps = self.engine.ps
        want = {n: tuple(ps.specs[n]) for n in ps.specs if arch.group_of(n) in groups}
        for n, shp in want.items():
            if n not in values:
                raise KeyError("checkpoint %s has no variable %s" % (path, n))
            if tuple(values[n].shape) != shp and values[n].size != int(np.prod(shp, dtype=np.int64)):
                raise ValueError("checkpoint %s: %s has shape %s, the model needs %s" % (path, n, values[n].shape, shp))
        ps.load_dict({n: values[n] for n in want}, strict=False)
        print("%s model loaded." % label)

    @staticmethod
    def ClipIfNotNone(grad):
        """codes/base.py:514-517 (element-wise clip to [-1,1]); numpy/torch arrays."""
        return None if grad is None else grad.clip(-1, 1)

    @property
    def variable_shapes(self):
        return dict(arch.param_specs(self.config))


class MNISTModel_digit(BaseModel):
    exp_name = "mnist_digit"


class MNISTModel_fashion(BaseModel):
    exp_name = "mnist_fashion"


class CelebAModel_densenet(BaseModel):
    exp_name = "celeba"
