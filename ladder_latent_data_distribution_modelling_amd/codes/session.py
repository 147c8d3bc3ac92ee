"""`sess.run(fetches, feed_dict)` facade over the HIP engine.

The reference's notebook and demo/demo_tools.py:41-120 drive the model through the TF contract
`sess.run([model.representation_mean, model.decoded], feed_dict={model.original_signal: x, model.is_code_input: False, ...})`
on graph-tensor attributes of the model (codes/models.py:27-44,153-159,348-390; codes/base.py:110-213).  Here every such
attribute is a named `Handle`; `Session.run` evaluates the requested handles with the HIP kernels, honouring the three routing
switches of the graph:

    is_code_input            decoded        <- decoder(code_input)                 instead of decoder(code_sample)   (models.py:107,265,500)
    is_outer_VAE_input       inner encoder  <- customised_inner_VAE_input          instead of code_sample            (base.py:142-144)
    is_representation_input  decoded_code   <- inner decoder(representation_input) instead of representation_sample  (base.py:171-173)

Train ops (`train_step_ae`, `train_step_sigma`, `train_step_prior`, `train_step_inner_sigma`) in the fetch list execute the
corresponding run of codes/base.py:583-641 with the fed learning rate.  Like `tf.Session.run`, every call draws fresh noise,
unfed placeholders that a fetch needs raise, and the result mirrors the structure of `fetches` (single handle, list or tuple).
"""
import weakref

import numpy as np
import torch

PLACEHOLDERS = ("original_signal", "seed", "data_file", "code_input", "is_code_input", "representation_input",
                "is_representation_input", "is_outer_VAE_input", "customised_inner_VAE_input", "prior_mean", "prior_cov",
                "prior_weight", "use_standard_gaussian_prior", "use_mask", "lr_ae", "lr_sigma", "lr_prior", "lr_inner_sigma")
TENSORS = ("input_image", "code_mean", "code_std_dev", "code_sample", "decoded", "representation_mean",
           "representation_std_dev", "representation_sample", "decoded_code", "std_dev_code", "std_dev_representation",
           "psedeu_input")        # (sic: the reference's spelling of the VampPrior pseudo-inputs, base.py:219)
SCALARS = ("sigma", "mean_pixel_error", "inner_sigma", "mean_code_error", "entropy_z", "crossEntropy_prior",
           "crossEntropy_prior_sg", "code_reconstruction_likelihood", "code_l1_reconstruction_error",
           "representation_regularisor", "entropy_t", "crossEntropy_representation", "elbo_prior", "l1_reconstruction_error",
           "l2_reconstruction_error", "reconstruction_likelihood", "sigma_regularisor", "elbo", "negative_elbo", "loss_ae",
           "loss_prior")
TRAIN_OPS = {"train_step_ae": ("ae", "lr_ae"), "train_step_sigma": ("sigma", "lr_sigma"),
             "train_step_prior": ("prior", "lr_prior"), "train_step_inner_sigma": ("inner_sigma", "lr_inner_sigma")}


class Handle:
    """Stand-in for a tf.Tensor / tf.placeholder / tf.Operation attribute of the model: hashable, usable as a feed_dict key."""
    __slots__ = ("name", "kind", "owner")

    def __init__(self, name, kind, owner=None):
        self.name, self.kind = name, kind
        self.owner = weakref.ref(owner) if owner is not None else None

    def __repr__(self):
        return "<ladder.%s %r>" % (self.kind, self.name)


class Deferred:
    """A value that exists once `Session.run` evaluates it -- what a TFP distribution method returns in graph mode
    (`prior.sample(n)`, `prior.prob(pos)`: demo/demo_tools.py:120, 265 of the reference)."""
    __slots__ = ("thunk",)

    def __init__(self, thunk):
        self.thunk = thunk


def attach_handles(model):
    """Give `model` the attribute surface of SURVEY Appendix E2."""
    for names, kind in ((PLACEHOLDERS, "placeholder"), (TENSORS, "tensor"), (SCALARS, "scalar"), (TRAIN_OPS, "op")):
        for n in names:
            setattr(model, n, Handle(n, kind, model))


class Session:
    """`Session(model)`, or `Session()` as in the reference's `sess = tf.Session(...)` (train.py:41-47): an unbound session
    evaluates on the model that owns the fetched handles.  Trainers accept it as their `sess` argument and ignore it."""

    def __init__(self, model=None):
        self.model = model

    def bind(self, model):
        self.model = model
        return self

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # ------------------------------------------------------------------------------------------------------------
    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, Deferred):
            return fetches.thunk()
        single = isinstance(fetches, Handle)
        flist = [fetches] if single else list(fetches)
        if flist and all(isinstance(h, Deferred) for h in flist):
            out = [h.thunk() for h in flist]
            return tuple(out) if isinstance(fetches, tuple) else out
        for h in flist:
            if not isinstance(h, Handle):
                raise TypeError("fetch %r is not a model attribute handle" % (h,))
        model = self.model or (flist[0].owner() if flist and flist[0].owner is not None else None)
        if model is None:
            raise RuntimeError("Session is not bound to a model: use Session(model) or sess.bind(model)")
        for h in flist + list(feed_dict or {}):
            if isinstance(h, Handle) and h.owner is not None and h.owner() is not model:
                raise ValueError("%r belongs to a different model (TF: 'Tensor is not an element of this graph')" % (h,))
        self._model = model
        feeds = {}
        for k, v in (feed_dict or {}).items():
            if not isinstance(k, Handle) or k.kind != "placeholder":
                raise TypeError("feed_dict key %r is not a placeholder of the model" % (k,))
            feeds[k.name] = v
        vals = self._evaluate({h.name for h in flist}, feeds)
        out = [vals[h.name] for h in flist]
        return out[0] if single else (tuple(out) if isinstance(fetches, tuple) else out)

    # ------------------------------------------------------------------------------------------------------------
    def _evaluate(self, want, feeds):
        eng = self._model.engine
        flag = lambda k, d: bool(np.asarray(feeds[k]).item()) if k in feeds else d

        def need(k):
            if k not in feeds:
                raise ValueError("You must feed a value for placeholder %r" % k)      # TF: InvalidArgumentError
            return feeds[k]

        uses_mixture = eng.has_inner or eng.gmm_z                       # priors whose graph holds the GM placeholders (base.py:88-124)
        fed_mixture = all(k in feeds for k in ("prior_weight", "prior_mean", "prior_cov"))
        if fed_mixture and uses_mixture:
            eng.set_mixture(feeds["prior_weight"], feeds["prior_mean"], feeds["prior_cov"])
        use_sg = flag("use_standard_gaussian_prior", True)
        use_mask = flag("use_mask", False)
        code_in = flag("is_code_input", False)
        outer_in = flag("is_outer_VAE_input", True)
        rep_in = flag("is_representation_input", False)
        vals = {}
        if "psedeu_input" in want:
            if not eng.vamp:
                raise ValueError("prior %r has no pseudo-inputs" % eng.cfg["prior"])
            vals["psedeu_input"] = eng.ps.w["prior/Variable"].detach().cpu().numpy()
            want = want - {"psedeu_input"}
            if not want:
                return vals

        ops = [n for n in want if n in TRAIN_OPS]
        if ops:
            if code_in or not outer_in or rep_in:
                raise ValueError("train ops run on the default routing (encoder -> decoder, outer VAE -> inner VAE)")
            x = need("original_signal")
            for n in ("train_step_ae", "train_step_sigma", "train_step_prior", "train_step_inner_sigma"):
                if n in ops:
                    kind, lrk = TRAIN_OPS[n]
                    getattr(eng, "run_" + kind)(x, float(np.asarray(need(lrk)).item()), None, use_sg, use_mask)
                    vals[n] = None
            # like TF, other fetches of the same run see the pre-update forward pass of the (last) executed run
            self._collect(vals, want, eng, x)
            return vals

        default_route = not code_in and outer_in and not rep_in
        wants_scalar = any(n in SCALARS for n in want) or "std_dev_code" in want or "std_dev_representation" in want
        if wants_scalar and not default_route:
            raise ValueError("loss / ELBO fetches are defined on the default routing (is_code_input=False, "
                             "is_outer_VAE_input=True, is_representation_input=False)")
        if default_route and (wants_scalar or want & {"decoded", "decoded_code", "representation_mean", "representation_std_dev",
                                                      "representation_sample"}):
            x = need("original_signal")
            parts = ["dec"] if (wants_scalar or "decoded" in want) else []
            if eng.has_inner:
                parts.append("inner")
            if wants_scalar:
                # the loss graph reads use_standard_gaussian_prior and, for the mixture priors, the three GM placeholders: TF raises
                # InvalidArgumentError for an unfed placeholder, it never falls back to a default
                if eng.cfg["prior"] in ("ours", "hierarchical", "vampPrior"):   # the priors whose loss goes through the tf.cond
                    need("use_standard_gaussian_prior")                         # (base.py:318-320, 357-359, 368-370; "GMM" has none)
                if eng.gmm_z or (eng.has_inner and not eng.hier):
                    if not fed_mixture and eng._gm_packed is None:
                        need("prior_weight"), need("prior_mean"), need("prior_cov")
                    parts.append("gmm")
                elif eng.vamp:
                    parts.append("gmm")                                 # encoder pass over the pseudo-inputs (base.py:216-254)
            eng.forward(x, None, use_sg, use_mask, tuple(parts))
            self._collect(vals, want, eng, x)
            return vals

        # ---- piecewise evaluation for the generation / embedding routes of the demo
        lat = None
        if want & {"code_mean", "code_std_dev", "code_sample", "input_image"} or (not code_in and "decoded" in want) or (
                outer_in and want & {"representation_mean", "representation_std_dev", "representation_sample"}) or (
                outer_in and not rep_in and "decoded_code" in want):
            x = need("original_signal")
            eng.forward(x, None, use_sg, use_mask, ())
            lat = eng.lat_z                                         # (mu, sd, sd_raw, eps, z)
            vals.update(input_image=eng.x, code_mean=lat[0], code_std_dev=lat[1], code_sample=lat[4])
        if "decoded" in want:
            vals["decoded"] = eng.decode(need("code_input")) if code_in else eng.decoder.forward(lat[4])
        if want & {"representation_mean", "representation_std_dev", "representation_sample"} or (
                "decoded_code" in want and not rep_in):
            if not eng.has_inner:
                raise ValueError("prior %r has no inner VAE" % eng.cfg["prior"])
            zin = lat[4] if outer_in else eng._dev(need("customised_inner_VAE_input"))
            mu_t, sdraw_t = eng.inner.encode(zin)
            from .. import _lib as L
            B, R, P = mu_t.shape[0], eng.R, eng.partials
            eng._run_calls = 0
            eps_t = eng._randn(B, R)
            L.call("ladder_u64_add", eng.rng_counter.data_ptr(), 1, eng.ctx.stream)
            t, sd_t = eng.ctx.empty(B, R), eng.ctx.empty(B, R)      # sd = relu(raw) + precision, t = mu + sd*eps (base.py:158-167)
            L.call("ladder_latent_fwd", mu_t.data_ptr(), sdraw_t.data_ptr(), eps_t.data_ptr(), eng.lvp, t.data_ptr(), sd_t.data_ptr(),
                   P[L.P_LOG_SDT:].data_ptr(), P[L.P_MU2SD2_T:].data_ptr(), P[L.P_FIXED + eng.Z:].data_ptr(), B, R, eng.ctx.stream)
            vals.update(representation_mean=mu_t, representation_std_dev=sd_t, representation_sample=t)
        if "decoded_code" in want:
            t = eng._dev(need("representation_input")) if rep_in else vals["representation_sample"]
            vals["decoded_code"] = eng.decode_representation(t)
        return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in vals.items() if k in want or k == "psedeu_input"}

    @staticmethod
    def _collect(vals, want, eng, x):
        f = None
        for n in want:
            if n in vals:
                continue
            if n in SCALARS:
                if f is None:
                    f = eng.fetch()
                vals[n] = np.float32(-f["elbo"] if n == "negative_elbo" else f[n])
            elif n == "std_dev_code":
                vals[n] = eng.std_dev_code()
            elif n == "std_dev_representation":
                vals[n] = eng.std_dev_representation()
            elif n == "input_image":
                vals[n] = eng.x.cpu().numpy()
            elif n in ("code_mean", "code_std_dev", "code_sample"):
                vals[n] = eng.lat_z[{"code_mean": 0, "code_std_dev": 1, "code_sample": 4}[n]].cpu().numpy()
            elif n in ("representation_mean", "representation_std_dev", "representation_sample"):
                vals[n] = eng.lat_t[{"representation_mean": 0, "representation_std_dev": 1, "representation_sample": 4}[n]].cpu().numpy()
            elif n == "decoded":
                if eng.xhat is None:
                    raise ValueError("decoded was not evaluated by this run")
                vals[n] = eng.xhat.cpu().numpy()
            elif n == "decoded_code":
                vals[n] = eng.zhat.cpu().numpy()
