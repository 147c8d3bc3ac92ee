"""Shortest-likely-path (SLP) interpolation in the latent space of the fitted mixture prior -- the optimisation of the reference's
notebook `latent-space-interpolation-mnist.ipynb` (cells 18-23), without TensorFlow.

    objective(pts) = w_path * sum_i |p_{i+1} - p_i|  +  w_equal * std_i |p_{i+1} - p_i|  -  sum_i log p_GM(p_i)

over `n_step` intermediate points between two fixed embeddings, minimised with clip-[-1,1] + Adam(beta1=.9, beta2=.95) exactly like
the notebook's `opt_interpolation` (cell 19).  The mixture term and its gradient come from the HIP mixture kernel
(`ladder_gmm_logprob_fwd_bwd` with one "MC sample" and eps = 0, so t = mean = the path points); the remaining algebra is a handful
of numbers and stays on the host.  `decode_path` maps the optimised path to images (t -> inner decoder -> z -> decoder), as
demo/demo_tools.py:163-186 does through `sess.run`.
"""
import math

import numpy as np
import torch

from .. import _lib as L


def path_terms(pts, start, end):
    """-> (entire_path_length, equal_length_constraint, d/dpts of each) for the intermediate points `pts` [n, R]."""
    full = np.concatenate([start[None], pts, end[None]], 0)
    d = full[1:] - full[:-1]                                  # segments, [n+1, R]
    ln = np.sqrt((d ** 2).sum(1))
    unit = d / np.maximum(ln, 1e-30)[:, None]
    g_len = unit[:-1] - unit[1:]                              # d sum(len) / d p_i : +from the segment ending at p_i, -from the one leaving it
    mean = ln.mean()
    std = math.sqrt(((ln - mean) ** 2).mean())                # tf.math.reduce_std: population standard deviation
    c = (ln - mean) / (len(ln) * max(std, 1e-30))             # d std / d len_j
    g_std = c[:-1, None] * unit[:-1] - c[1:, None] * unit[1:]
    return ln.sum(), std, g_len, g_std


class SLPInterpolator:
    def __init__(self, engine, weights, means, covs):
        """`engine`: a LadderEngine (for the device / stream / decoders); (weights, means, covs): the fitted mixture (R <= 8)."""
        self.eng = engine
        dev = engine.ctx.device
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        w, m, c = f(weights), f(means), f(covs)
        self.K, self.R = int(m.shape[0]), int(m.shape[1])
        self.packed = torch.empty(self.K * L.query("ladder_gmm_packed_stride", self.R), device=dev)
        L.call("ladder_gmm_prepare", w.data_ptr(), m.data_ptr(), c.data_ptr(), self.K, self.R, self.packed.data_ptr(), engine.ctx.stream)
        torch.cuda.current_stream(dev).synchronize()

    def neg_log_likelihood(self, pts):
        """-> (-sum_i log p(p_i), gradient [n, R]) from the HIP mixture kernel."""
        dev, st = self.eng.ctx.device, self.eng.ctx.stream
        n = pts.shape[0]
        mu = torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32)).to(dev)
        sd, eps = torch.ones_like(mu), torch.zeros(1, n, self.R, device=dev)
        out, dmu, dsd = torch.empty(1, device=dev), torch.empty_like(mu), torch.empty_like(mu)
        ws = torch.empty(max(L.query("ladder_gmm_workspace_bytes", 1, n), 16), dtype=torch.uint8, device=dev)
        L.call("ladder_gmm_logprob_fwd_bwd", mu.data_ptr(), sd.data_ptr(), eps.data_ptr(), self.packed.data_ptr(), 1, n, self.R, self.K,
               out.data_ptr(), dmu.data_ptr(), dsd.data_ptr(), ws.data_ptr(), ws.numel(), st)
        return -float(out.item()), -dmu.cpu().numpy().astype(np.float64)

    def optimise(self, start, end, n_step=5, n_iter=500, lr=1e-2, w_equal_length=100.0, w_path_dist=10.0, init=None):
        """Notebook cells 18-21.  Returns (pts [n_step, R], record dict of the per-iteration loss terms)."""
        start, end = np.asarray(start, np.float64), np.asarray(end, np.float64)
        pts = np.asarray(init, np.float64).copy() if init is not None else np.linspace(start, end, n_step + 1, endpoint=False)[1:]
        from .utils import register_trainable_scope
        register_trainable_scope("interpolation", pts.size)      # notebook cell 19: count_trainable_variables('interpolation')
        m, v = np.zeros_like(pts), np.zeros_like(pts)
        rec = dict(loss=[], path_length=[], step_var=[], neg_ll=[])
        for t in range(1, n_iter + 1):
            plen, std, g_len, g_std = path_terms(pts, start, end)
            nll, g_nll = self.neg_log_likelihood(pts)
            rec["loss"].append(w_path_dist * plen + w_equal_length * std + nll)
            rec["path_length"].append(plen); rec["step_var"].append(std); rec["neg_ll"].append(nll)
            g = np.clip(w_path_dist * g_len + w_equal_length * g_std + g_nll, -1.0, 1.0)      # model.ClipIfNotNone
            m = 0.9 * m + 0.1 * g
            v = 0.95 * v + 0.05 * g * g
            pts = pts - lr * math.sqrt(1 - 0.95 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        return pts, rec

    def decode_path(self, start, pts, end):
        """Images along [start, pts..., end] (clipped to [0,1] as the demo does)."""
        t = np.concatenate([np.asarray(start)[None], pts, np.asarray(end)[None]], 0)
        code = self.eng.decode_representation(t) if self.eng.has_inner else self.eng._dev(t)
        return np.clip(self.eng.decode(code).cpu().numpy(), 0.0, 1.0)
