"""Demo helpers with the reference's names and call shapes (demo/demo_tools.py:41-120 of the reference), as adapters over the
`Session` facade (codes/session.py) and the HIP mixture kernel -- no TensorFlow / TFP.

    embedding = get_embeddings_from_val_set(idx, config, exp_name, sess, data, model, trainer)      # reference :41-79
    prior     = define_prior_distribution(config, sess, model, gmm_info=GM)                         # reference :83-115
    samples   = generate_prior_embeddings(prior, sess, n_embeddings)                                # reference :118-120

`define_prior_distribution` returns an object with the three members the reference's callers use on the TFP distribution:
`prior.sample(n)`, `prior.prob(pos)` and `prior.log_prob(pos)`.  Like TFP in graph mode they return DEFERRED values which
`sess.run(...)` evaluates (`Session.run` accepts them next to model handles), so reference-style code such as
`sess.run(prior.sample(n_embeddings))` or `sess.run(prior.prob(pos)) + 1e-8` (reference :120, :265) runs unchanged.
Densities of full-covariance mixtures are evaluated by `ladder_gmm_logprob_fwd_bwd` (csrc/elbo.hip, lane = component) for
R <= 8 and by the dense-GEMM mixture path (`ladder_gmm_dense_logprob_fwd_bwd`) for wider latents; diagonal priors in closed form
from a handful of numbers.  Sampling is ancestral (component draw + Cholesky), as tfd.Mixture.sample does.

Plotting: the reference's figure helpers are out of scope; `plot_images_and_its_reconstruction` is kept as a thin
matplotlib call because `get_embeddings_from_val_set` invokes it, and is skipped when matplotlib is missing or `show_plot=False`.
"""
import math

import numpy as np
import torch

from .. import _lib as L
from ..codes.session import Deferred


def plot_images_and_its_reconstruction(x, x_decoded, config, x_from_t=None, save_plot=False, idx=0):
    """Original | decoded from z | decoded from t (reference :12-38); writes <result_dir>original_image_<idx>.pdf when asked."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # noqa: BLE001  (headless GPU box without matplotlib)
        return None
    three = config["prior"] in ("ours", "hierarchical")
    imgs = [(x, "original"), (x_decoded, "decoded from z")] + ([(x_from_t, "decoded from t")] if three else [])
    fig, axs = plt.subplots(1, len(imgs), figsize=(2 * len(imgs), 2), edgecolor="k")
    fig.subplots_adjust(hspace=.2, wspace=.4)
    for ax, (im, title) in zip(np.ravel(axs), imgs):
        ax.imshow(np.squeeze(im))
        ax.set_title(title)
        ax.grid(False)
        ax.set_xticks([])
        ax.set_yticks([])
    if save_plot:
        fig.savefig(config["result_dir"] + "original_image_{}.pdf".format(idx))
    plt.close(fig)
    return fig


def get_embeddings_from_val_set(idx, config, exp_name, sess, data, model, trainer, save_plot=False, show_plot=True):
    """Embedding of validation image `idx` (representation_mean for the ladder priors, code_mean otherwise) through the same three
    `sess.run` calls as the reference (:41-79): x -> (t-mean, decoded); t-mean -> decoded_code; decoded_code -> decoded."""
    x = data.val_set["image"] if exp_name == "mnist_digit" else trainer.test_batch
    Z = int(config["code_size"])
    if config["prior"] in ("ours", "hierarchical"):
        R = int(config["representation_size"])
        feed_dict = {model.original_signal: x, model.is_code_input: False, model.code_input: np.zeros((1, Z)),
                     model.is_outer_VAE_input: True, model.customised_inner_VAE_input: np.zeros((1, Z)),
                     model.is_representation_input: False, model.representation_input: np.zeros((1, R))}
        embedding, x_decoded = sess.run([model.representation_mean, model.decoded], feed_dict=feed_dict)
        feed_dict[model.is_representation_input] = True
        feed_dict[model.representation_input] = embedding
        z_decoded = sess.run(model.decoded_code, feed_dict=feed_dict)
        x_decoded = np.clip(x_decoded, 0., 1.)
        feed_dict = {model.original_signal: x, model.is_code_input: True, model.code_input: z_decoded}
        x_from_t = np.clip(sess.run(model.decoded, feed_dict=feed_dict), 0., 1.)
        if show_plot:
            plot_images_and_its_reconstruction(x[idx], x_decoded[idx], config, x_from_t[idx], save_plot=save_plot, idx=idx)
    else:
        feed_dict = {model.original_signal: x, model.is_code_input: False, model.code_input: np.zeros((1, Z))}
        embedding, x_decoded = sess.run([model.code_mean, model.decoded], feed_dict=feed_dict)
        x_decoded = np.clip(x_decoded, 0., 1.)
        if show_plot:
            plot_images_and_its_reconstruction(x[idx], x_decoded[idx], config, save_plot=save_plot, idx=idx)
    return np.squeeze(embedding[idx])


# ------------------------------------------------------------------------------------------------ prior objects
class _Prior:
    """Common surface of the objects define_prior_distribution returns (the subset of tfd.Distribution the reference touches)."""

    def sample(self, n, seed=None):
        return Deferred(lambda: self._sample(int(n), seed))

    def log_prob(self, value):
        return Deferred(lambda: self._log_prob(np.asarray(value, np.float32)))

    def prob(self, value):
        return Deferred(lambda: np.exp(self._log_prob(np.asarray(value, np.float32))))


class DiagonalGaussianPrior(_Prior):
    """tfd.MultivariateNormalDiag(loc, scale_diag): the standard-Gaussian / hierarchical priors (reference :84-87, :100-103)."""

    def __init__(self, loc, scale_diag):
        self.loc, self.scale = np.asarray(loc, np.float32), np.asarray(scale_diag, np.float32)
        self._rng = np.random.default_rng(0)

    def _sample(self, n, seed):
        rng = np.random.default_rng(seed) if seed is not None else self._rng
        return (self.loc + self.scale * rng.standard_normal((n, self.loc.size))).astype(np.float32)

    def _log_prob(self, v):
        z = (v - self.loc) / self.scale
        return (-0.5 * (z * z).sum(-1) - np.log(self.scale).sum() - 0.5 * self.loc.size * math.log(2 * math.pi)).astype(np.float32)


class MixturePrior(_Prior):
    """tfd.Mixture(Categorical(probs=w), [MultivariateNormalFullCovariance(m_k, K_k)]) (reference :88-99) -- or, with
    `scale_diag`, the VampPrior's mixture of diagonal Gaussians (:104-115).  log_prob runs on the device the model lives on."""

    def __init__(self, engine, weights, means, covs):
        self.eng = engine
        dev = engine.ctx.device
        self.w = np.asarray(weights, np.float64) / np.sum(weights)
        self.m, self.c = np.asarray(means, np.float64), np.asarray(covs, np.float64)
        self.K, self.R = self.m.shape
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        w, m, c = f(self.w), f(self.m), f(self.c)
        self.dense = self.R > 8
        n = L.query("ladder_gmm_dense_param_floats", self.K, self.R) if self.dense else self.K * L.query("ladder_gmm_packed_stride", self.R)
        self.packed = torch.empty(n, device=dev)
        L.call("ladder_gmm_prepare_dense" if self.dense else "ladder_gmm_prepare", w.data_ptr(), m.data_ptr(), c.data_ptr(), self.K, self.R,
               self.packed.data_ptr(), engine.ctx.stream)
        torch.cuda.current_stream(dev).synchronize()
        self._chol = np.linalg.cholesky(self.c)
        self._rng = np.random.default_rng(0)

    def _sample(self, n, seed):
        rng = np.random.default_rng(seed) if seed is not None else self._rng
        comp = rng.choice(self.K, size=n, p=self.w)
        return (self.m[comp] + np.einsum("nij,nj->ni", self._chol[comp], rng.standard_normal((n, self.R)))).astype(np.float32)

    def _log_prob(self, v):
        """log p of every point of `v` [..., R] in one launch (ladder_gmm_logprob_rows: one wavefront per point, lane = component;
        wide latents: whitening GEMM + per-row logsumexp, ladder_gmm_dense_logprob_rows)."""
        shape = v.shape[:-1]
        pts = np.ascontiguousarray(v.reshape(-1, self.R), dtype=np.float32)
        n = pts.shape[0]
        dev, st = self.eng.ctx.device, self.eng.ctx.stream
        t, out = torch.as_tensor(pts).to(dev), torch.empty(n, device=dev)
        if self.dense:
            ws = torch.empty(L.query("ladder_gmm_dense_workspace_bytes", 1, n, self.R, self.K), dtype=torch.uint8, device=dev)
            L.call("ladder_gmm_dense_logprob_rows", t.data_ptr(), self.packed.data_ptr(), n, self.R, self.K, out.data_ptr(), ws.data_ptr(),
                   ws.numel(), st)
        else:
            L.call("ladder_gmm_logprob_rows", t.data_ptr(), self.packed.data_ptr(), n, self.R, self.K, out.data_ptr(), st)
        return out.cpu().numpy().reshape(shape)


def define_prior_distribution(config, sess, model, gmm_info=None):
    """reference :83-115.  `gmm_info`: {'w', 'm', 'K'} of the fitted mixture (priors "ours" / "GMM")."""
    prior = config["prior"]
    if prior == "standard_gaussian":
        Z = int(config["code_size"])
        return DiagonalGaussianPrior(np.zeros(Z), np.ones(Z))
    if prior in ("GMM", "ours"):
        return MixturePrior(model.engine, gmm_info["w"], gmm_info["m"], gmm_info["K"])
    if prior == "hierarchical":
        R = int(config["representation_size"])
        return DiagonalGaussianPrior(np.zeros(R), np.ones(R))
    if prior == "vampPrior":
        # the encoder's posterior at the K trainable pseudo-inputs defines K equally weighted diagonal components (reference :104-115)
        feed_dict = {model.original_signal: sess.run(model.psedeu_input), model.code_input: np.zeros((1, int(config["code_size"]))),
                     model.is_code_input: False}
        mean, std = sess.run([model.code_mean, model.code_std_dev], feed_dict=feed_dict)
        K = int(config["n_mixtures"])
        covs = np.stack([np.diag(np.asarray(s, np.float64) ** 2) for s in std])
        return MixturePrior(model.engine, np.full(K, 1.0 / K), mean, covs)
    raise ValueError("unknown prior %r" % prior)


def generate_prior_embeddings(prior, sess, n_embeddings):
    """reference :118-120."""
    return sess.run(prior.sample(n_embeddings))
