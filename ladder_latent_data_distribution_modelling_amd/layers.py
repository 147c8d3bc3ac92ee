"""Device context, parameter store and the layers with explicit backward passes of the HIP LaDDer path.  PyTorch supplies device memory, streams and
torch.distributed only; every arithmetic op below is a call into libladder_hip.so (see _lib.py) -- there is no CPU path."""
import os

import numpy as np
import torch

from . import _lib as L
from . import arch
from .comm import Comm, _NoComm

BN_EPS = 1e-3      # tf.layers.batch_normalization default
IN_EPS = 1e-6      # tf.contrib.layers.instance_norm default
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.95, 1e-8   # codes/base.py:459-461


def _p(t):
    return None if t is None else t.data_ptr()


# config key `matmul_precision`: how the large 3x3 convolutions are contracted (csrc/convsplit.hip; values = LADDER_PREC_*).
#   "f32"     v_mfma_f32_32x32x2_f32: bit-exact fp32 FMA chains at the fp32 vector rate
#   "f16x3"   fp32 operands scaled by a power of two (from the absolute maximum of their sample, or tensor) and split into 2 fp16 planes (22 bits),
#             3 plane products on the fp16 matrix cores, fp32 accumulate: fp32-class error (measured: below the f32 FMA chain's)
#   "bf16x6"  3 bf16 planes (24 bits), 6 plane products, no scaling: fp32-class error
#   "bf16x3"  2 bf16 planes (16 bits), 3 products (error ~1e-5 relative per product: between TF32 and fp32)
PRECISIONS = {"f32": 0, "bf16x3": 2, "bf16x6": 3, "f16x3": 4}
DEFAULT_PRECISION = "f32"       # the reference computes in fp32 end to end (codes/models.py:348,388): split formats are an explicit opt-in

PRECISION_NOTES = {
    "f32": "native fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-exact fp32 FMA chains",
    "f16x3": "fp32-class EMULATION: fp32 operands as 2 scaled fp16 planes (22 significand bits, power-of-two scales per sample / "
             "per tensor, see DESIGN 4a), 3 fp16 MFMAs per product, fp32 accumulation; set \"matmul_precision\": \"f32\" for strict fp32",
    "bf16x6": "fp32-class EMULATION: 3 bf16 planes (24 bits), 6 bf16 MFMAs per product, fp32 accumulation",
    "bf16x3": "REDUCED precision: 2 bf16 planes (16 bits), 3 bf16 MFMAs per product (between TF32 and fp32)"}

PROF = None   # set to a KernelProfiler by bench.py
_WS_NEED, _KID = {}, {}


def _igemm(ctx, name, M, Cin, Cout, Kdim, *args, conv=None):
    """Forward-type implicit-GEMM call (conv fwd / bwd_data / dense fwd / bwd_data): appends the split-K workspace and the
    stream; when a profiler is installed the launch is bracketed by HIP events and attributed to its kernel."""
    key = (M, Kdim, Cout)
    nb = _WS_NEED.get(key)
    if nb is None:
        nb = _WS_NEED[key] = L.query("ladder_igemm_fwd_workspace_bytes", M, Kdim, Cout)
    wsp, wsn = ctx.ws(nb) if nb else (None, 0)
    args = args + (wsp, wsn, ctx.stream)
    if PROF is None or conv == "skip":
        L.call(name, *args)
        return
    kkey = (M, Cin, Cout, conv)
    kid = _KID.get(kkey)
    if kid is None:
        kid = L.query("ladder_conv2d_fwd_kernel_id", *conv) if conv else L.query("ladder_igemm_fwd_tile", M, Cin, Cout)
        if kid != 256128 and nb and L.query("ladder_igemm_fwd_splits", M, Kdim, Cout) > 1:
            kid = 0         # split-K launch: two kernels, not attributed
        _KID[kkey] = kid
    if kid in (256128, 128128):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        L.call(name, *args)
        e.record()
        PROF.add(kid, s, e, 2.0 * M * Kdim * Cout)
    else:
        L.call(name, *args)


def _timed(kid, flops, name, args, executed=None):
    """One launch, bracketed by HIP events on the launch stream when a profiler is installed (bench.py's roofline leg)."""
    if PROF is None:
        L.call(name, *args)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    L.call(name, *args)
    e.record()
    PROF.add(kid, s, e, flops, executed)


class PlanesOnly:
    """Stand-in for an activation that exists ONLY as its fp16 plane images (registered with Ctx.set_planes / set_amax): the batch-norm
    apply of an encoder layer whose consumer is a split gather convolution never writes the fp32 tensor.  Any fp32 use raises."""

    def __init__(self, shape):
        self.shape = torch.Size(shape)

    def numel(self):
        n = 1
        for d in self.shape:
            n *= int(d)
        return n

    def data_ptr(self):
        raise RuntimeError("this activation exists only as fp16 planes (PlanesOnly): an fp32 kernel was routed to it")


class Ctx:
    """Device context shared by all layers: stream handle, grow-only workspace, communicator."""

    def __init__(self, device, comm=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.LadderHipError("the LaDDer HIP path needs a GPU device (got %s); there is no CPU fallback" % device)
        L.load()
        self.comm = comm or Comm()
        self._ws = torch.empty(1 << 20, dtype=torch.uint8, device=self.device)
        self._ws_retired, self.ws_generation = [], 0
        self._amax = {}      # id(tensor) -> (weakref, absolute-maximum record)
        self._planes = {}    # id(tensor) -> (weakref, pre-split planes)
        self.pack_banks, self._pack_table = [], None              # split filter images known so far; their device job table
        # second HIP stream for the filter gradients (MFMA-bound, needed only by the optimiser step): they run beside the backward-data /
        # resize / norm backward kernels of the layers below, which are HBM-bound and fit on the same CUs (Conv2D.backward, join_side)
        self.side, self._side_active, self._side_refs = None, False, []
        self._ws_side = torch.empty(1 << 20, dtype=torch.uint8, device=self.device)
        self._ws_side_retired = []
        self.aux = None      # third HIP stream: RUN#3 / RUN#4 beside RUN#2's decoder forward (LadderEngine.enable_prior_overlap)
        self._ws_aux = None
        self.keep_activations = True   # False inside forward-only runs: fused kernels may skip writing tensors only a backward pass reads
        self.ns = 0          # LADDER_PREC_* of the split-precision contraction kernels (0 = native f32 MFMA); set by the engine
        self.up2_used = {}   # layer name -> number of upsample-fused launches so far (bench.py's executed-FLOP model)
        self.up2_skipped = {}  # ... and the fraction of the reference's products such a launch never issues (11 / 36 tap-folded, 27 / 36 projected)
        self.fuse_bwd_proj = 1   # last pair: backward combination straight from the 1x1 output conv's gradient (config `fused_projection_backward`)
        self.fuse_fwd = 2    # projected pairs: forward GEMM + combination in one launch (config `fused_projected_forward`: 2 wherever eligible (default), 1 only where the isolated launch measured faster, 0 off)
        self.up2 = True      # resize -> 3x3 conv pairs of the decoder as ONE upsample-fused convolution in forward-only runs (config `upsample_fused_convs`)

    @property
    def sfx(self):
        """Suffix of the batch-sized dense entry points for the configured precision ("_f32": strict fp32 MFMA; "": bf16x6)."""
        return "" if self.ns else "_f32"

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def fork_side(self, *keep):
        """Context manager: what is launched inside runs on the side stream, ordered after everything enqueued on the main stream so
        far.  `keep`: tensors the side stream reads -- referenced until join_side() so that the caching allocator (which only knows the
        main stream) does not hand their memory out again."""
        ev = torch.cuda.Event()
        ev.record()
        self.side.wait_event(ev)
        self._side_refs.extend(k for k in keep if k is not None)
        self._side_active = True
        return torch.cuda.stream(self.side)

    def side_or_main(self, *keep):
        """fork_side() when the side stream is enabled (eager mode, not inside a hipGraph capture), else a no-op context."""
        if self.side is None or torch.cuda.is_current_stream_capturing():
            import contextlib
            return contextlib.nullcontext()
        return self.fork_side(*keep)

    def join_side(self):
        """The main stream waits for the side stream (before anything reads the filter gradients)."""
        if self._side_active:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self._side_active = False
            self._side_refs.clear()
            self._ws_side_retired.clear()

    def ws(self, nbytes):
        if self.aux is not None and torch.cuda.current_stream(self.device) == self.aux:        # ... and so has the prior-run stream
            if self._ws_aux is None or self._ws_aux.numel() < nbytes:
                # (stream-ordered: the old buffer belongs to this stream's allocator pool, kernels already enqueued on it run first)
                self._ws_aux = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
            return self._ws_aux.data_ptr(), self._ws_aux.numel()
        if self.side is not None and torch.cuda.current_stream(self.device) == self.side:      # the side stream has its own scratch
            if self._ws_side.numel() < nbytes:
                self._ws_side_retired.append(self._ws_side)
                self._ws_side = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
            return self._ws_side.data_ptr(), self._ws_side.numel()
        if self._ws.numel() < nbytes:
            # a captured hipGraph has the pointer of the workspace it was recorded with baked in: superseded buffers stay alive (the
            # caching allocator must never hand their memory to a live tensor) and every graph recorded so far is dropped, so the
            # next call of a run re-captures against the new buffer
            self._ws_retired.append(self._ws)
            self._ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
            self.ws_generation += 1
        return self._ws.data_ptr(), self._ws.numel()

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    # -- absolute-maximum records of the f16x3 split kernels (include/ladder_hip.h: ladder_absmax) --------------------------------
    # Producers that can compute max|y| while they write y (split conv epilogue, instance-norm apply, ...) register the record
    # for the tensor object; consumers look it up and fall back to a standalone pass over the tensor.  Entries are keyed by the
    # tensor OBJECT (dropped when it is collected), never by address: a view / reshape is a different object and simply misses.
    def new_amax(self):
        return torch.empty(L.ABSMAX_FLOATS, dtype=torch.float32, device=self.device)

    def set_amax(self, t, rec):
        if self.ns == 4 and rec is not None:
            import weakref
            k, reg = id(t), self._amax
            reg[k] = (weakref.ref(t, lambda _r, k=k, reg=reg: reg.pop(k, None)), rec)

    def known_amax(self, t):
        e = self._amax.get(id(t))
        return e[1] if e is not None and e[0]() is t else None

    def drop_amax(self, t):
        self._amax.pop(id(t), None)
        self._planes.pop(id(t), None)

    def absmax(self, t):
        """Absolute-maximum record of `t` (an upper bound is as good: it only moves the 2^-38 representation floor); None when
        the precision mode needs none.  A standalone pass writes a PER-SAMPLE record for [N, ...] tensors (include/ladder_hip.h)."""
        if self.ns != 4:
            return None
        rec = self.known_amax(t)
        if rec is None:
            rec = self.new_amax()
            per = t.numel() // int(t.shape[0]) if t.dim() >= 2 else 0
            if per and per % 4 == 0 and int(t.shape[0]) > 1:
                L.call("ladder_absmax_samples", _p(t), int(t.shape[0]), per, _p(rec), self.stream)
            else:
                L.call("ladder_absmax", _p(t), t.numel(), _p(rec), self.stream)
            self.set_amax(t, rec)
        return rec

    def set_planes(self, t, buf):
        """Registers plane images a producer wrote for `t` itself (ladder_bn_fwd_apply_planes)."""
        import weakref
        k, reg = id(t), self._planes
        reg[k] = (weakref.ref(t, lambda _r, k=k, reg=reg: reg.pop(k, None)), buf)

    def planes(self, t, per_sample=False):
        """Pre-split 16-bit planes of `t` (ladder_presplit) for the gather kernels, cached per tensor object like the absmax records:
        a layer input is split once and serves the forward call and the filter gradient, an output gradient the backward-data call
        and the filter gradient.  `per_sample`: scale every sample by its own maximum (when the record carries per-sample bounds; the
        planes' header tells the consumer) -- the caller asks for it only where the filter-gradient kernel can follow (a sample's output
        pixels a multiple of its 32-pixel chunks)."""
        e = self._planes.get(id(t))
        if e is not None and e[0]() is t:
            return e[1]
        import weakref
        buf = torch.empty(L.query("ladder_presplit_bytes", t.numel(), self.ns), dtype=torch.uint8, device=self.device)
        ns = int(t.shape[0]) if (per_sample and t.dim() == 4 and (t.numel() // int(t.shape[0])) % 8 == 0) else 0
        L.call("ladder_presplit", _p(t), _p(self.absmax(t)), _p(buf), t.numel(), ns, self.ns, self.stream)
        k, reg = id(t), self._planes
        reg[k] = (weakref.ref(t, lambda _r, k=k, reg=reg: reg.pop(k, None)), buf)
        return buf

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def local(self):
        """The same device context without cross-rank exchange (batch statistics of a replicated batch stay local)."""
        import copy
        c = copy.copy(self)
        c.comm = _NoComm()
        return c


class ParamStore:
    """Flat fp32 buffers per optimiser group (theta, grad, m, v) with named views.

    One flat gradient buffer per group = one all-reduce (C1/C4) and one clip+Adam launch (N11)."""

    ALIGN = 64  # elements (256 B): keeps every view 16-byte aligned for float4 loads

    def __init__(self, cfg, ctx, values=None, seed=1):
        self.cfg, self.ctx = cfg, ctx
        self.specs = arch.param_specs(cfg)
        values = values if values is not None else arch.init_values(cfg, seed)
        self.offsets, sizes = {}, {}
        for name, shp in self.specs.items():
            g = arch.group_of(name)
            off = sizes.get(g, 0)
            n = int(np.prod(shp)) if len(shp) else 1
            self.offsets[name] = (g, off, n)
            sizes[g] = off + (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.theta = {g: ctx.zeros(n) for g, n in sizes.items()}
        self.grad = {g: ctx.zeros(n) for g, n in sizes.items()}
        self.m = {g: ctx.zeros(n) for g, n in sizes.items()}
        self.v = {g: ctx.zeros(n) for g, n in sizes.items()}
        self.step = {g: 0 for g in sizes}                       # host mirror of the per-optimiser step counters
        self.version = {g: 0 for g in sizes}                    # bumped whenever a group's values change (packed-filter caches)
        # device-resident optimiser state {lr, lr_t, step}: a captured hipGraph replays the step without host scalars
        self.adam_state = {g: ctx.zeros(4) for g in sizes}
        self._lr_host = {g: None for g in sizes}
        self.w, self.g = {}, {}
        for name, shp in self.specs.items():
            g, off, n = self.offsets[name]
            self.w[name] = self.theta[g][off:off + n].view(*shp) if len(shp) else self.theta[g][off:off + 1]
            self.g[name] = self.grad[g][off:off + n].view(*shp) if len(shp) else self.grad[g][off:off + 1]
        self.load_dict(values)

    def prefix_range(self, group, prefix):
        """[lo, hi) of the flat `group` buffers covered by the variables whose name starts with `prefix`, or None if they are
        not one contiguous run (param_specs orders names, so each scope is contiguous)."""
        runs = [(off, off + (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN, name.startswith(prefix))
                for name, (g, off, n) in self.offsets.items() if g == group]
        runs.sort()
        inside = [r for r in runs if r[2]]
        if not inside:
            return None
        lo, hi = inside[0][0], inside[-1][1]
        if any(lo <= r[0] < hi and not r[2] for r in runs):
            return None
        return lo, hi

    def load_dict(self, values, strict=True):
        getattr(self, "before_read", lambda: None)()
        for name in self.specs:
            if name not in values:
                if strict:
                    raise KeyError("missing variable %s" % name)
                continue
            v = torch.as_tensor(np.asarray(values[name], np.float32).reshape(self.w[name].shape))
            self.w[name].copy_(v.to(self.ctx.device))
            self.version[arch.group_of(name)] += 1

    def to_dict(self, groups=None):
        getattr(self, "before_read", lambda: None)()           # (the engine: wait for runs in flight on the aux stream)
        return {n: self.w[n].detach().cpu().numpy().reshape(self.specs[n]) for n in self.specs
                if groups is None or arch.group_of(n) in groups}

    def num_params(self, prefix):
        return sum(self.offsets[n][2] for n in self.specs if n.startswith(prefix))

    def set_lr(self, group, lr):
        """Write the learning rate into the device state (only when it changed; never inside a graph capture)."""
        if self._lr_host[group] != lr:
            L.call("ladder_axpy", None, _p(self.adam_state[group]), 1, float(lr), 2, self.ctx.stream)       # (fill: no torch elementwise launch on the path)
            self._lr_host[group] = lr

    def adam(self, group, lr, grad=None, n=None):
        """clip to [-1,1] + TF-form Adam on the whole group (codes/base.py:459-517); lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is
        evaluated on the device from the device step counter.  `grad` may be a device pointer into the scalars vector
        (n = 1) for the two scalar optimisers."""
        self.ctx.join_side()                                    # filter gradients computed on the side stream
        self.set_lr(group, lr)
        self.step[group] += 1
        self.version[group] += 1
        g = self.grad[group] if grad is None else grad
        L.call("ladder_adam_clip_dev", _p(self.theta[group]), _p(g), _p(self.m[group]), _p(self.v[group]),
               self.theta[group].numel() if n is None else n, _p(self.adam_state[group]), ADAM_B1, ADAM_B2, ADAM_EPS, 1.0,
               self.ctx.stream)


# ------------------------------------------------------------------------------------------ layers
UP2T_MIN_PIXELS = int(os.environ.get("LADDER_UP2T_MIN_PIXELS", "256"))      # smallest low-resolution map whose backward-data runs upsample-fused
PROJ_MAX_BYTES = int(os.environ.get("LADDER_PROJ_MAX_BYTES", str(32 << 30)))  # largest Z / D temporary of a projected pair (conv2d_7 at batch 128: 2.4 GB)
UP2W_MIN_PIXELS = int(os.environ.get("LADDER_UP2W_MIN_PIXELS", "256"))      # ... and whose filter gradient does (8x8: 758 us fused against 612 direct)


class Conv2D:
    """tf.layers.conv2d (NHWC / HWIO), bias + activation fused in the kernel epilogue."""

    def __init__(self, ctx, ps, name, k, cin, cout, stride=1, padding="same", act=None, bias_grad=True):
        self.ctx, self.ps, self.name = ctx, ps, name
        self.k, self.cin, self.cout, self.stride, self.padding, self.act = k, cin, cout, stride, padding, act
        # a conv feeding batch-/instance-norm has an identically-zero bias gradient (the norm subtracts the mean):
        # it is not computed and stays 0 in the flat gradient buffer
        self.bias_grad = bias_grad
        self._packed = {}      # (transpose_flip, ns) -> [weight version, packed bf16 planes]
        self.group = arch.group_of(name + "/kernel")           # optimiser group whose version stamps the packed images
        self.want_bn_sums, self.bn_sums = False, None          # batch-norm statistics of the output from the conv epilogue (RGB conv)
        self.x_is_up2 = False                                   # set by forward_up2(keep_y): self.x is a factor-2 legacy-bilinear upsample
        self.x_is_lo = False                                    # ... or self.x is the LOW-resolution tensor itself (the upsample was never materialised)
        self.lo_factor = 2                                      # ... by this resize factor
        self.x = self.y = None                                  # operands kept by a training forward for the backward pass

    def _halo_ok(self, N, H, W, cin, cout):
        """The layer runs on the fused 3x3 halo kernels of the configured precision (strict fp32: csrc/convf32.hip; split formats:
        csrc/convsplit.hip) -- same tiling, same eligibility."""
        if self.ctx.ns == 0 and os.environ.get("LADDER_DISABLE_HALO") == "1":      # (test switch: the round-1 generic fp32 gather kernels everywhere)
            return False
        # (strict fp32 also takes the 16- / 8-pixel-wide maps: csrc/convf32s.hip)
        return bool(self.k == 3 and self.stride == 1 and self.padding == "same"
                    and L.query("ladder_conv3x3_f32_eligible" if self.ctx.ns == 0 else "ladder_conv3x3_split_eligible", N, H, W, cin, cout))

    def _halo_kid(self, N, H, W, cin, cout, class_cout=None):
        """Profiler id of the halo-kernel launch over an [N, H, W] map with `cin` gathered and `cout` bank columns: the 8x32-pixel tiling
        (csrc/convf32.hip, csrc/convsplit.hip) or -- strict fp32 only -- a small-map tiling (csrc/convf32s.hip).  `class_cout`: channels per
        class of a class-structured launch (the 8x32 tiling needs 128)."""
        if self.ctx.ns:
            return 256120 + self.ctx.ns
        big = L.query("ladder_conv3x3_split_eligible", N, H, W, cin, cout) and (class_cout is None or class_cout == 128)
        return 256120 if big else 256064

    def _split_ok(self, N, H, W, cin, cout):
        """... and the precision is one of the 16-bit split formats (their filter gradient / planes / absmax machinery)."""
        return bool(self.ctx.ns and self._halo_ok(N, H, W, cin, cout))

    def _as_dense(self, M):
        return bool(self.k == 1 and self.stride == 1 and M <= 512 and self.cin >= 16
                    and L.query("ladder_dense_small_eligible", M, self.cin, self.cout))

    @staticmethod
    def _ps(Ho, Wo):
        """Per-sample f16x3 scales for the planes of this layer's operands: where the split filter-gradient kernel, which shares them,
        can re-scale at sample boundaries (a sample's output pixels = whole 32-pixel chunks)."""
        return (Ho * Wo) % 32 == 0

    def _rgb(self, N, H, W):
        # (the kernels are f16x3 inside -- the filter gradient takes tensor-wide absmax records -- so they belong to that precision mode)
        return bool(self.ctx.ns == 4 and self.cin == 3 and L.query("ladder_conv_rgb_s2_eligible", N, H, W, self.cin, self.cout, self.k, self.k,
                                                               self.stride, self.pt, self.pl))

    def _rgb_fwd32(self, N, H, W):
        # strict fp32: the forward of the same layer on the fp32 instantiation of the kernel (its filter gradient stays on the generic kernel)
        return bool(self.ctx.ns == 0 and self.cin == 3 and os.environ.get("LADDER_DISABLE_HALO") != "1" and
                    L.query("ladder_conv_rgb_s2_eligible", N, H, W, self.cin, self.cout, self.k, self.k, self.stride, self.pt, self.pl))

    def planes_demand(self, in_shape):
        """(wants_planes, needs_fp32) for an input of `in_shape`: whether this layer's forward reads the fp16 plane images of its input
        (split gather kernel) and whether anything of it still needs the fp32 tensor (a filter gradient outside the split kernel)."""
        ctx = self.ctx
        N, H, W, _ = in_shape
        if ctx.ns != 4 or self.cin == 3 or self._split_ok(N, H, W, self.cin, self.cout):
            return False, True
        pt, Ho = arch.conv_out(H, self.k, self.stride, self.padding)
        pl, Wo = arch.conv_out(W, self.k, self.stride, self.padding)
        geo = (N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride, pt, pl)
        if self._as_dense(N * H * W) or not L.query("ladder_conv2d_fwd_split_eligible", *geo):
            return False, True
        return True, bool(ctx.keep_activations and not L.query("ladder_conv2d_bwd_filter_split_eligible", *geo))

    def _packed_filter(self, transpose_flip):
        """Split bf16 planes of the filter bank in the kernel's LDS layout, re-packed when the weights changed (always while a
        hipGraph is being captured, so that a replay re-packs the then-current weights)."""
        ns, ps = self.ctx.ns, self.ps
        if ns == 0 and transpose_flip == 0:           # strict fp32: the HWIO bank IS the forward bank [tap][Cin][Cout]
            return ps.w[self.name + "/kernel"]
        cin, cout = (self.cout, self.cin) if transpose_flip in (1, 2) else (self.cin, self.cout)
        if transpose_flip == 2:                       # the four parity classes of a stride-2 backward-data as output-channel blocks
            cout = 4 * self.cin
        elif transpose_flip == 3:                     # the four output-parity classes of the upsample-fused forward (effective taps)
            cout = 4 * self.cout
        elif transpose_flip == 4:                     # backward-data of the upsample-fused pair: the four pixel-parity classes of dy as input groups
            cin, cout = 4 * self.cout, self.cin
        elif transpose_flip == 5:                     # stride-2 forward: the four pixel-parity classes of x as input groups (strict fp32)
            cin, cout = 4 * self.cin, self.cout
        taps = self.k * self.k
        if transpose_flip == 6:                       # project-then-upsample: the nine taps side by side, ONE [cin][9 cout] matrix (strict fp32) ...
            taps, cin, cout = 1, self.cin, 9 * self.cout
        elif transpose_flip == 7:                     # ... and its transpose [9 cout][cin], the backward-data operand
            taps, cin, cout = 1, 9 * self.cout, self.cin
        ent = self._packed.get((transpose_flip, ns))
        if ent is None:
            nb = L.query("ladder_filter_pack_split_bytes", taps, cin, cout, ns)
            ent = self._packed[(transpose_flip, ns)] = [-1, torch.empty(nb, dtype=torch.uint8, device=self.ctx.device)]
            # known to the batched re-pack after an optimiser step (LadderEngine._repack_filters): (entry, bank, taps, cin, cout, flip, ns)
            self.ctx.pack_banks.append((ent, ps.w[self.name + "/kernel"], taps, cin, cout, transpose_flip, ns, self.group))
        ver = ps.version[self.group]
        if ent[0] != ver or torch.cuda.is_current_stream_capturing():
            L.call("ladder_filter_pack_split", _p(ps.w[self.name + "/kernel"]), _p(ent[1]), taps, cin, cout, transpose_flip, ns,
                   self.ctx.stream)
            ent[0] = ver
        return ent[1]

    def forward_fused_proj(self, x, proj, keep_y):
        """This 3x3 conv + activation followed by the 1x1 conv `proj` (<= 4 output channels, no activation) in ONE launch of the split
        halo kernel (ladder_conv3x3_split_proj); returns proj's output or None when the pair is not eligible.  `keep_y` = the
        activation is needed later (training forward: both layers' backward read it); a forward-only run never writes it."""
        N, H, W, _ = x.shape
        if not (self._halo_ok(N, H, W, self.cin, self.cout) and self.cout <= 128 and proj.k == 1 and proj.stride == 1
                and proj.cout <= 4 and proj.act is None and proj.cin == self.cout):
            return None
        self.pt, _ = arch.conv_out(H, self.k, self.stride, self.padding)
        self.pl, _ = arch.conv_out(W, self.k, self.stride, self.padding)
        proj.pt = proj.pl = 0
        y = self.ctx.empty(N, H, W, self.cout) if keep_y else None
        out = self.ctx.empty(N, H, W, proj.cout)
        self.x_amax = self.ctx.absmax(x)
        args = (_p(x), _p(self.x_amax), _p(self._packed_filter(0)), _p(self.ps.w[self.name + "/bias"]), _p(y),
                _p(self.ps.w[proj.name + "/kernel"]), _p(self.ps.w[proj.name + "/bias"]), _p(out), proj.cout, N, H, W, self.cin, self.cout,
                L.ACT[self.act], self.ctx.ns, self.ctx.stream)
        _timed(256120 + self.ctx.ns, 2.0 * N * H * W * 9 * self.cin * self.cout, "ladder_conv3x3_split_proj", args)
        self.x, self.y = x, y
        self.x_is_up2 = self.x_is_lo = False
        proj.x, proj.y = y, out
        return out

    def _proj_rides(self, proj):
        """The 1x1 conv `proj` behind this layer can ride on the epilogue of its upsample-fused / projected launch."""
        return bool(self.cout == 128 and proj.k == 1 and proj.stride == 1 and proj.cout <= 4 and proj.act is None and proj.cin == self.cout)

    def up2_ok(self, N, H, W):
        """This layer can take the LOW-resolution tensor [N, H, W, cin] that a factor-2 legacy-bilinear resize would have blown up for it
        (ladder_conv3x3_up2_split: four output-parity classes with effective taps, 25 instead of 36 low-resolution tap products and no
        upsampled tensor).  Forward-only runs use it; a training forward keeps the resized tensor for its backward pass."""
        if self.ctx.ns == 0 and os.environ.get("LADDER_DISABLE_HALO") == "1":
            return False
        if self.proj_ok(N, H, W):
            return True
        return bool(self.ctx.up2 and self.ctx.ns in (0, 2, 4) and self.k == 3 and self.stride == 1 and self.padding == "same"
                    and L.query("ladder_conv3x3_up2_split_eligible", N, H, W, self.cin, self.cout, self.ctx.ns))

    def upf_ok(self, N, H, W, f):
        """up2_ok for a resize factor f: 2 (every form) or 4 (projected form only -- decoder conv2d_3 behind the 2x2 -> 8x8 resize)."""
        return self.up2_ok(N, H, W) if f == 2 else self.proj_ok(N, H, W, f)

    def virtual_upf_ok(self, N, H, W, f):
        return self.virtual_up2_ok(N, H, W) if f == 2 else self.proj_ok(N, H, W, f)

    def proj_ok(self, N, H, W, f=2):
        """'Project, then upsample' (csrc/upproj.hip; strict fp32, config `upsample_fused_convs` >= 4): resize x2 -> this conv over a LOW-resolution
        [N, H, W, cin] tensor as nine 1x1 convolutions on it (9 of the direct form's 36 products per 2x2 output block, against 25 for the tap-folded
        form above) + an exact elementwise combination.  Forward, backward-data and the filter gradient all run from the low-resolution tensor.
        `f` = the resize factor, 2 or 4 (1 of 16 products at 4)."""
        # (the nine planes Z / D are a transient [N H W, 9 cout] fp32 tensor: beyond PROJ_MAX_BYTES the layer takes the forms that allocate none)
        return bool(self.ctx.ns == 0 and self.ctx.up2 >= 4 and self.k == 3 and self.stride == 1 and self.padding == "same"
                    and os.environ.get("LADDER_DISABLE_HALO") != "1" and 36 * N * H * W * self.cout <= PROJ_MAX_BYTES
                    and L.query("ladder_upfproj_eligible", f, N, H, W, self.cin, self.cout))

    def virtual_up2_ok(self, N, H, W):
        """A training forward may skip materialising the factor-2 upsample of its [N, H, W, cin] input altogether: strict fp32, and forward,
        backward-data AND filter gradient of this layer all run from the low-resolution tensor (csrc/convf32.hip)."""
        if self.proj_ok(N, H, W):
            return True
        return bool(self.ctx.ns == 0 and self.ctx.up2 >= 3 and self.up2_ok(N, H, W) and self.up2t_ok(N, H, W) and H * W >= UP2W_MIN_PIXELS
                    and L.query("ladder_conv3x3_up2_wgrad_eligible", N, H, W, self.cin, self.cout))

    def _forward_proj(self, x, proj, keep_y, upsampled, f=2):
        """forward_up2 in the project-then-upsample form: Z [M, 9 cout] = x [M, cin] . wcat (dense kernel), then the elementwise combination with
        bias, activation and -- for the last layer -- the 1x1 output conv on the activated value."""
        ctx, st = self.ctx, self.ctx.stream
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
        M, n9 = N * H * W, 9 * self.cout
        self.pt = self.pl = 1
        flops = 2.0 * N * f * f * H * W * 9 * self.cin * self.cout      # the reference's operation count (algorithmic) ...
        executed = 2.0 * M * self.cin * n9                               # ... of which 9 / 36 are issued (factor 2; 1 / 16 at factor 4)
        ukey = self.name + (":train" if keep_y else "")
        ctx.up2_used[ukey] = ctx.up2_used.get(ukey, 0) + 1
        ctx.up2_skipped[ukey] = 1.0 - 1.0 / (f * f)
        bias = self.ps.w[self.name + "/bias"]
        if f == 2 and ctx.fuse_fwd and L.query("ladder_up2proj_fused_eligible" if ctx.fuse_fwd >= 2 else "ladder_up2proj_fused_preferred", N, H, W, self.cin, self.cout):
            # round 6: GEMM + combination in ONE launch, the nine planes in an LDS ring (csrc/upproj.hip: up2proj_fused_fwd_kernel) -- Z never reaches HBM
            y = ctx.empty(N, 2 * H, 2 * W, self.cout) if (keep_y or proj is None) else None
            out, pw, pb, pco = y, None, None, 0
            if proj is not None:
                proj.pt = proj.pl = 0
                out, pco = ctx.empty(N, 2 * H, 2 * W, proj.cout), proj.cout
                pw, pb = self.ps.w[proj.name + "/kernel"], self.ps.w[proj.name + "/bias"]
                proj.x, proj.y = (y, out) if keep_y else (None, None)
            nb = L.query("ladder_up2proj_fused_workspace_bytes", N, H, W, self.cout, pco)
            wsp, wsn = ctx.ws(nb) if nb else (None, 0)
            _timed(128136, flops, "ladder_up2proj_fused_fwd",
                   (_p(x), _p(self._packed_filter(7)), _p(bias), _p(y), _p(pw), _p(pb), _p(out) if proj is not None else None, pco, N, H, W, self.cin, self.cout,
                    L.ACT[self.act], wsp, wsn, st), executed)
            self.x_amax = None
            self.lo_factor = f
            self.x, self.y = ((upsampled if upsampled is not None else x), y) if keep_y else (None, None)
            self.x_is_up2 = bool(keep_y and upsampled is not None)
            self.x_is_lo = bool(keep_y and upsampled is None)
            return out
        z = ctx.empty(M, n9)
        wsp, wsn = ctx.ws(L.query("ladder_igemm_fwd_workspace_bytes", M, self.cin, n9))
        _timed(128132 if L.query("ladder_dense_fwd_is_persistent", M, self.cin, n9) else abs(L.query("ladder_igemm_fwd_tile", M, self.cin, n9)), flops, "ladder_dense_fwd",
               (_p(x), _p(self._packed_filter(6)), None, _p(z), M, self.cin, n9, 0, wsp, wsn, st), executed)
        if proj is not None:
            proj.pt = proj.pl = 0
            y = ctx.empty(N, 2 * H, 2 * W, self.cout) if keep_y else None
            out = ctx.empty(N, 2 * H, 2 * W, proj.cout)
            L.call("ladder_up2proj_fwd_combine", _p(z), _p(bias), _p(y), _p(self.ps.w[proj.name + "/kernel"]), _p(self.ps.w[proj.name + "/bias"]), _p(out),
                   proj.cout, N, H, W, self.cout, L.ACT[self.act], st)
            proj.x, proj.y = (y, out) if keep_y else (None, None)
        else:
            out = y = ctx.empty(N, f * H, f * W, self.cout)
            L.call("ladder_upfproj_fwd_combine", _p(z), _p(bias), _p(y), f, N, H, W, self.cout, L.ACT[self.act], st)
        self.x_amax = None
        self.lo_factor = f
        self.x, self.y = ((upsampled if upsampled is not None else x), y) if keep_y else (None, None)
        self.x_is_up2 = bool(keep_y and upsampled is not None)
        self.x_is_lo = bool(keep_y and upsampled is None)
        return out

    def bwd_proj_ok(self, proj):
        """The backward combination of this (last) pair can be formed straight from the gradient of the 1x1 conv `proj` behind it -- dy of this layer is
        never materialised and proj's filter / bias gradient comes out of the same launch (ladder_up2proj_bwd_combine_proj)."""
        if not (self.ctx.fuse_bwd_proj and self.x_is_lo and self.lo_factor == 2 and self.x is not None and self.y is not None
                and self.act in (None, "leaky_relu", "relu") and proj.k == 1 and proj.stride == 1 and proj.act is None and proj.cin == self.cout):
            return False
        N, H, W, _ = self.x.shape
        return bool(self.proj_ok(N, H, W, 2) and L.query("ladder_up2proj_bwd_combine_proj_eligible", N, H, W, self.cout, proj.cout))

    def _backward_proj(self, dy, need_dx, wgrad, gate, proj_grad=None):
        """Backward of the project-then-upsample form from the low-resolution x: D [M, 9 cout] = (shift o up)^T dy once (elementwise), then
        dWcat = x^T D (+ the bias gradient as the centre plane's column sums) and dx_lo = D . wcatT -- two dense calls, exact on every pixel.
        `proj_grad` = (dyp, proj): dy is not given -- D comes from this layer's activated output and the gradient dyp of the 1x1 conv `proj`
        behind it, together with proj's own filter / bias gradient (bwd_proj_ok)."""
        ctx, st = self.ctx, self.ctx.stream
        x = self.x
        N, H, W, _ = x.shape
        M, n9, f = N * H * W, 9 * self.cout, self.lo_factor
        flops = 2.0 * N * f * f * H * W * 9 * self.cin * self.cout
        executed = 2.0 * M * self.cin * n9
        d = ctx.empty(M, n9)
        if proj_grad is not None:
            dyp, proj = proj_grad
            wsp, wsn = ctx.ws(L.query("ladder_up2proj_bwd_combine_proj_workspace_bytes", N, H, W, self.cout, proj.cout))
            L.call("ladder_up2proj_bwd_combine_proj", _p(self.y), _p(dyp), _p(proj.ps.w[proj.name + "/kernel"]), _p(d), _p(proj.ps.g[proj.name + "/kernel"]),
                   _p(proj.ps.g[proj.name + "/bias"]) if proj.bias_grad else None, proj.cout, N, H, W, self.cout, L.ACT[self.act] if self.act else 0,
                   wsp, wsn, st)
            proj.x = proj.y = None
        else:
            L.call("ladder_upfproj_bwd_combine", _p(dy), _p(d), f, N, H, W, self.cout, st)
        if wgrad:
            ctx.up2_used[self.name + ":wgrad"] = ctx.up2_used.get(self.name + ":wgrad", 0) + 1
            ctx.up2_skipped[self.name + ":wgrad"] = 1.0 - 1.0 / (f * f)
            dwcat, db9 = ctx.empty(self.cin, n9), (ctx.empty(n9) if self.bias_grad else None)   # (a conv in front of a norm has no bias gradient)
            wsp, wsn = ctx.ws(L.query("ladder_dense_bwd_weight_workspace_bytes", M, self.cin, n9))
            _timed(9132 if L.query("ladder_dense_bwd_weight_is_persistent", M, self.cin, n9) else 9130, flops, "ladder_dense_bwd_weight",
                   (_p(x), _p(d), _p(dwcat), _p(db9), M, self.cin, n9, wsp, wsn, st), executed)
            L.call("ladder_up2proj_wgrad_unpack", _p(dwcat), _p(db9), _p(self.ps.g[self.name + "/kernel"]),
                   _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, self.cin, self.cout, st)
        dx = None
        if need_dx:
            ctx.up2_used[self.name + ":bwd"] = ctx.up2_used.get(self.name + ":bwd", 0) + 1
            ctx.up2_skipped[self.name + ":bwd"] = 1.0 - 1.0 / (f * f)
            dx = ctx.empty(N, H, W, self.cin)
            wsp, wsn = ctx.ws(L.query("ladder_igemm_fwd_workspace_bytes", M, n9, self.cin))
            gy, gact = gate if gate is not None else (None, None)
            # (ladder_dense_bwd_data: dx [M, K] = dy [M, N] . wT [N, K], optionally times act'(gate) -- here dy := D, wT := wcatT)
            if L.query("ladder_dense_fwd_is_persistent", M, n9, self.cin):
                # K-contiguous weight operand = wcat itself (orientation 6): gemm_nt16_f32_kernel (v_mfma_f32_16x16x4_f32; 3-5 % ahead of the 32x32x2 kernel
                # on this long-K shape: conv2d_7 1 202 against 1 266 us, profiles/r05_gemm_library_probe.txt)
                _timed(128134, flops, "ladder_dense_bwd_data_nt",
                       (_p(d), _p(self._packed_filter(6)), _p(dx), M, self.cin, n9, _p(gy), L.ACT[gact] if gact else 0, st), executed)
            else:
                _timed(abs(L.query("ladder_igemm_fwd_tile", M, n9, self.cin)), flops, "ladder_dense_bwd_data",
                       (_p(d), _p(self._packed_filter(7)), _p(dx), M, self.cin, n9, _p(gy), L.ACT[gact] if gact else 0, wsp, wsn, st), executed)
        self.x = self.y = None
        return dx

    def forward_up2(self, x, proj=None, keep_y=False, x_for_backward=None, factor=2):
        """conv(resize2x(x)) from the low-resolution x itself; with `proj` the 1x1 output conv rides on the epilogue as in forward_fused_proj.
        Forward-only runs keep nothing.  A training forward (`keep_y`) passes `x_for_backward` = the resized tensor, which it has to keep for
        the backward pass anyway (filter gradient and backward-data are those of the plain convolution on it): x / y are then kept exactly
        as forward_fused_proj keeps them."""
        ctx = self.ctx
        src = x
        N, H, W = x.shape[0], x.shape[1], x.shape[2]
        if proj is not None and not self._proj_rides(proj):
            # the fused 1x1 epilogues hold one pixel's 128 channels in a half-wave and project to <= 4 columns (csrc/upproj.hip, convf32.hip):
            # any other last-layer width (num_hidden_units != 512) runs the pair unfused -- this conv from the low-resolution tensor, then `proj`
            y = self.forward_up2(x, None, keep_y, x_for_backward, factor)
            out = proj.forward(y)
            if not keep_y:
                proj.x = proj.y = None
            return out
        if self.proj_ok(N, H, W, factor):
            return self._forward_proj(x, proj, keep_y, x_for_backward, factor)
        if factor != 2:
            raise RuntimeError("%s: a factor-%d resize folds into the convolution in the projected form only" % (self.name, factor))
        self.lo_factor = 2
        strided = 0
        upsampled = x_for_backward
        self.pt = self.pl = 1
        bias, wk = self.ps.w[self.name + "/bias"], self.ps.w[self.name + "/kernel"]
        x_amax = ctx.absmax(src)                                         # (max |upsampled| = max |x|: the resize is a convex combination)
        flops = 2.0 * N * 4 * H * W * 9 * self.cin * self.cout          # the reference's operation count (algorithmic) ...
        executed = flops * 25.0 / 36.0                                   # ... of which 25 / 36 are issued
        ukey = self.name + (":train" if keep_y else "")                   # (bench.py's executed-FLOP model: forward-only / training forward)
        ctx.up2_used[ukey] = ctx.up2_used.get(ukey, 0) + 1
        ctx.up2_skipped[ukey] = 11.0 / 36.0
        wsp, wsn = ctx.ws(L.query("ladder_conv3x3_up2_edges_workspace_bytes", N, H, W, self.cin, self.cout))
        if proj is not None:
            proj.pt = proj.pl = 0
            y = ctx.empty(N, 2 * H, 2 * W, self.cout) if keep_y else None
            out = ctx.empty(N, 2 * H, 2 * W, proj.cout)
            pw, pb = self.ps.w[proj.name + "/kernel"], self.ps.w[proj.name + "/bias"]
            _timed(self._halo_kid(N, H, W, self.cin, 4 * self.cout, self.cout), flops, "ladder_conv3x3_up2_split_proj",
                   (_p(src), _p(x_amax), _p(self._packed_filter(3)), _p(bias), _p(y), _p(pw), _p(pb), _p(out), proj.cout, N, H, W, self.cin, self.cout,
                    L.ACT[self.act], ctx.ns, strided, ctx.stream), executed)
            L.call("ladder_conv3x3_up2_edges", _p(src), _p(wk), _p(bias), _p(y), None, _p(pw), _p(pb), _p(out), proj.cout, N, H, W, self.cin, self.cout,
                   L.ACT[self.act], strided, wsp, wsn, ctx.stream)
            self.x_amax = x_amax if keep_y else None
            # (keep_y without a resized tensor: the LOW-resolution tensor is what the backward pass gets -- virtual_up2_ok)
            self.x, self.y = ((upsampled if upsampled is not None else src), y) if keep_y else (None, None)
            self.x_is_up2 = bool(keep_y and upsampled is not None)       # x = the factor-2 upsample of a tensor: the filter gradient reads its even sub-grid
            self.x_is_lo = bool(keep_y and upsampled is None)
            proj.x, proj.y = (y, out) if keep_y else (None, None)
            return out
        y = ctx.empty(N, 2 * H, 2 * W, self.cout)
        y_amax = ctx.new_amax() if ctx.ns == 4 else None
        _timed(self._halo_kid(N, H, W, self.cin, 4 * self.cout, self.cout), flops, "ladder_conv3x3_up2_split",
               (_p(src), _p(x_amax), _p(self._packed_filter(3)), _p(bias), _p(y), _p(y_amax), N, H, W, self.cin, self.cout, L.ACT[self.act], ctx.ns,
                strided, ctx.stream), executed)
        L.call("ladder_conv3x3_up2_edges", _p(src), _p(wk), _p(bias), _p(y), _p(y_amax), None, None, None, 0, N, H, W, self.cin, self.cout,
               L.ACT[self.act], strided, wsp, wsn, ctx.stream)
        ctx.set_amax(y, y_amax)
        self.x_amax = x_amax if keep_y else None
        self.x, self.y = ((upsampled if upsampled is not None else src), y) if keep_y else (None, None)
        self.x_is_up2 = bool(keep_y and upsampled is not None)
        self.x_is_lo = bool(keep_y and upsampled is None)
        return y

    def forward(self, x):
        N, H, W, _ = x.shape
        self.pt, Ho = arch.conv_out(H, self.k, self.stride, self.padding)
        self.pl, Wo = arch.conv_out(W, self.k, self.stride, self.padding)
        y = self.ctx.empty(N, Ho, Wo, self.cout)
        self.x_amax = None
        self.x_is_up2 = self.x_is_lo = False
        if self._halo_ok(N, H, W, self.cin, self.cout):
            self.x_amax = self.ctx.absmax(x)
            y_amax = self.ctx.new_amax() if self.ctx.ns == 4 else None
            args = (_p(x), _p(self.x_amax), _p(self._packed_filter(0)), _p(self.ps.w[self.name + "/bias"]), _p(y), _p(y_amax), N, H, W,
                    self.cin, self.cout, L.ACT[self.act], self.ctx.ns, self.ctx.stream)
            self.ctx.set_amax(y, y_amax)
            _timed(self._halo_kid(N, H, W, self.cin, self.cout), 2.0 * N * H * W * 9 * self.cin * self.cout, "ladder_conv3x3_split", args)
            self.x, self.y = x, y
            return y
        if self._as_dense(N * H * W):                   # 1x1 conv over a tiny map (decoder conv0 on the 1x1 map) = a batch-sized dense layer
            L.call("ladder_dense_fwd_small" + self.ctx.sfx, _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y), N * H * W,
                   self.cin, self.cout, L.ACT[self.act], self.ctx.stream)
            self.x, self.y = x, y
            return y
        geo = (N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride, self.pt, self.pl)
        if self._rgb(N, H, W) or self._rgb_fwd32(N, H, W):   # the image-side encoder conv (3 -> Cout channels, stride 2): csrc/convrgb.hip
            sfx = "" if self.ctx.ns else "_f32"
            if self.want_bn_sums and self.act is None:
                wsp, wsn = self.ctx.ws(L.query("ladder_conv_rgb_s2_fwd_bnstats_workspace_bytes", N, H, W, self.cout))
                self.bn_sums = self.ctx.empty(6 * self.cout)       # sum | sum of squares | min | max per channel
                L.call("ladder_conv_rgb_s2_fwd_bnstats" + sfx, _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y),
                       N, H, W, self.cout, 0, _p(self.bn_sums), wsp, wsn, self.ctx.stream)
            else:
                L.call("ladder_conv_rgb_s2_fwd" + sfx, _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y), N, H,
                       W, self.cout, L.ACT[self.act], self.ctx.stream)
            self.x, self.y = x, y
            return y
        if self.ctx.ns and L.query("ladder_conv2d_fwd_split_eligible", *geo):
            self.x_amax = self.ctx.absmax(x)
            nb = L.query("ladder_conv2d_fwd_split_workspace_bytes", *geo)
            wsp, wsn = self.ctx.ws(nb)
            if self.want_bn_sums and self.act is None and not nb:
                snb = L.query("ladder_conv2d_fwd_split_bnstats_workspace_bytes", *geo)
                if snb:                                  # the epilogue also emits the batch-norm statistics of y (no second pass over it)
                    swp, swn = self.ctx.ws(snb)
                    self.bn_sums = self.ctx.empty(6 * self.cout)
                    L.call("ladder_conv2d_fwd_split_bnstats", _p(self.ctx.planes(x, self._ps(Ho, Wo))), _p(self.x_amax), _p(self._packed_filter(0)),
                           _p(self.ps.w[self.name + "/bias"]), _p(y), *geo, 0, self.ctx.ns, _p(self.bn_sums), swp, swn, self.ctx.stream)
                    self.x, self.y = x, y
                    return y
            args = (_p(self.ctx.planes(x, self._ps(Ho, Wo))), _p(self.x_amax), _p(self._packed_filter(0)), _p(self.ps.w[self.name + "/bias"]), _p(y)) + geo + (
                L.ACT[self.act], self.ctx.ns, wsp, wsn, self.ctx.stream)
            if nb:                                       # split-K launch: two kernels, not attributed by the profiler
                L.call("ladder_conv2d_fwd_split", *args)
            else:
                _timed(128120 + self.ctx.ns, 2.0 * N * Ho * Wo * self.k * self.k * self.cin * self.cout, "ladder_conv2d_fwd_split", args)
            self.x, self.y = x, y
            return y
        if self.ctx.ns == 0 and self.want_bn_sums and self.act is None:
            snb = L.query("ladder_conv2d_fwd_bnstats_workspace_bytes", *geo)
            if snb:                                      # strict fp32: the epilogue also emits the batch-norm statistics of y (no second pass over it)
                swp, swn = self.ctx.ws(snb)
                self.bn_sums = self.ctx.empty(6 * self.cout)
                L.call("ladder_conv2d_fwd_bnstats", _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y), *geo, 0,
                       _p(self.bn_sums), swp, swn, self.ctx.stream)
                self.x, self.y = x, y
                return y
        if (self.ctx.ns == 0 and self.k == 3 and self.stride == 2 and self.pt == 0 and self.pl == 0 and os.environ.get("LADDER_DISABLE_HALO") != "1"
                and L.query("ladder_conv3x3_s2_fwd_f32_eligible", N, H, W, self.cin, Ho, Wo, self.cout)):
            # strict fp32, 3x3 / stride 2 over an even map (encoder conv2d_2 / conv2d_3): a stride-1 correlation over the four pixel-parity classes
            # of x on the halo kernels (x staged once per slab for all taps; csrc/convf32s.hip): 185 / 115 us against 202 / 127 on the gather kernel.
            # (Behind the statistics-epilogue branch above: conv2d_1 measures 344 us + a statistics pass here against 356 us with them.)
            args = (_p(x), _p(self._packed_filter(5)), _p(self.ps.w[self.name + "/bias"]), _p(y), N, H, W, self.cin, Ho, Wo, self.cout,
                    L.ACT[self.act], self.ctx.stream)
            _timed(self._halo_kid(N, Ho, Wo, 4 * self.cin, self.cout), 2.0 * N * Ho * Wo * 9 * self.cin * self.cout, "ladder_conv3x3_s2_fwd_f32", args)
            self.x, self.y = x, y
            return y
        _igemm(self.ctx, "ladder_conv2d_fwd", N * Ho * Wo, self.cin, self.cout, self.k * self.k * self.cin,
               _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y),
               N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride, self.pt, self.pl, L.ACT[self.act],
               conv=(N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride, 1, self.pt, self.pl))
        self.x, self.y = x, y
        return y

    def up2t_ok(self, N, H, W):
        """The gradient with respect to the LOW-resolution tensor [N, H, W, cin] behind a factor-2 resize in front of this layer can come from
        ONE launch (ladder_conv3x3_up2_bwd_data_split) + border strips, instead of backward-data on the upsampled map + the resize transpose."""
        if self.ctx.ns == 0 and os.environ.get("LADDER_DISABLE_HALO") == "1":
            return False
        if self.proj_ok(N, H, W) and (self.x is None or self.x_is_lo):
            # (asked before the forward: the geometry decides; asked in backward: only when the forward really kept the LOW-resolution tensor --
            # a materialised upsample goes through the tap-folded eligibility below)
            return True
        if self.ctx.ns == 0 and H * W < UP2T_MIN_PIXELS:
            # policy (measured, profiles/r05_small_maps.txt): on an 8x8 low-resolution map the four exact border lines cost more than the
            # 11 / 36 of the products the fused launch saves (conv2d_4: 461 + 221 us against 595 + 26 for the direct pair)
            return False
        return bool(self.ctx.up2 >= 2 and self.ctx.ns in (0, 2, 4) and self.k == 3 and self.stride == 1 and self.padding == "same"
                    and L.query("ladder_conv3x3_up2_bwd_data_split_eligible", N, H, W, self.cout, self.cin, self.ctx.ns)
                    and (self.ctx.ns == 0 or all(L.query("ladder_conv2d_bwd_data_split_eligible", *g, 0) for g in (
                        (N, 2, 2 * W, self.cin, 3, 2 * W, self.cout, 3, 3, 1, 1, 1), (N, 3, 2 * W, self.cin, 4, 2 * W, self.cout, 3, 3, 1, 2, 1),
                        (N, 2 * H, 2, self.cin, 2 * H, 3, self.cout, 3, 3, 1, 1, 1), (N, 2 * H, 3, self.cin, 2 * H, 4, self.cout, 3, 3, 1, 1, 2)))))

    def lowres_gate_ok(self, N, H, W):
        """_dx_lowres can apply the activation backward of the layer below (its `gate`) in the same launches (strict fp32, 8x32-pixel tiling).
        OFF by default (LADDER_ENABLE_LOWRES_GATE=1 turns it on): measured in round 5, the 64 gate loads per lane in the epilogue of conv2d_6's fused
        backward-data cost 72 us (1 631 -> 1 703 us) -- the 67 us ladder_act_bwd pass they replace (profiles/r05_f32_percall.md was taken with it on)."""
        if self.x_is_lo and self.proj_ok(N, H, W, self.lo_factor):
            # (projected form: the gate would ride on the dense kernel's epilogue -- measured 772 against 655 us on conv2d_6's backward-data GEMM, more than
            # the ~85 us activation pass it replaces: opt-in like the tap-folded form's)
            return os.environ.get("LADDER_ENABLE_LOWRES_GATE") == "1"
        return bool(self.ctx.ns == 0 and os.environ.get("LADDER_ENABLE_LOWRES_GATE") == "1"
                    and L.query("ladder_conv3x3_up2_bwd_data_gated_f32_eligible", N, H, W, self.cout, self.cin))

    def _dx_lowres(self, dy, dy_amax, gate=None):
        """d loss / d x_lo for y = conv(resize2x(x_lo)): the composite transpose is a zero-padded 5x5 / stride-2 correlation over dy (one launch of
        the halo kernel, 25 instead of 36 tap products per low-resolution pixel, the [N, 2H, 2W, cin] intermediate never written), exact on
        every pixel but the four border lines of dx, where the resize's clamp and the convolution's padding change the coefficients: those
        come from 4-pixel-wide strips of dy through the plain backward-data + resize-transpose kernels (exact there by construction)."""
        ctx, st = self.ctx, self.ctx.stream
        N, OH, OW, _ = dy.shape
        H, W = OH // 2, OW // 2
        dx = ctx.empty(N, H, W, self.cin)
        dx_amax = ctx.new_amax() if ctx.ns == 4 else None
        flops = 2.0 * N * OH * OW * 9 * self.cin * self.cout
        ctx.up2_used[self.name + ":bwd"] = ctx.up2_used.get(self.name + ":bwd", 0) + 1
        ctx.up2_skipped[self.name + ":bwd"] = 11.0 / 36.0
        pk4 = self._packed_filter(4)
        if ctx.ns == 0:
            # strict fp32: the main launch, then its four border lines made exact in place from ONE d_up line per border (csrc/convf32.hip:
            # ladder_conv3x3_up2_bwd_borders -- 9 instead of 45 line-taps per axis; the strip path below cost 1.07 ms per iteration)
            wsp, wsn = ctx.ws(L.query("ladder_conv3x3_up2_bwd_borders_workspace_bytes", N, H, W, self.cout, self.cin))
            if gate is not None:        # (y of the layer below, its activation): dx *= act'(y) in the epilogue and in the border fix-up
                gy, gact = gate
                _timed(self._halo_kid(N, H, W, 4 * self.cout, self.cin), flops, "ladder_conv3x3_up2_bwd_data_gated_f32",
                       (_p(dy), _p(pk4), _p(dx), _p(gy), L.ACT[gact], N, H, W, self.cout, self.cin, st), flops * 25.0 / 36.0)
                L.call("ladder_conv3x3_up2_bwd_borders_gated", _p(dy), _p(self.ps.w[self.name + "/kernel"]), _p(dx), _p(gy), L.ACT[gact], N, H, W, self.cout,
                       self.cin, wsp, wsn, st)
                return dx
            _timed(self._halo_kid(N, H, W, 4 * self.cout, self.cin), flops, "ladder_conv3x3_up2_bwd_data_split",
                   (_p(dy), None, _p(pk4), _p(dx), None, N, H, W, self.cout, self.cin, 0, st), flops * 25.0 / 36.0)
            L.call("ladder_conv3x3_up2_bwd_borders", _p(dy), _p(self.ps.w[self.name + "/kernel"]), _p(dx), N, H, W, self.cout, self.cin, wsp, wsn, st)
            return dx
        pk = self._packed_filter(1)
        # border lines: dx row 0 = R(d_up[0] + d_up[1] / 2), row H-1 = R(d_up[2H-3] / 2 + d_up[2H-2] + d_up[2H-1]) with d_up = the plain
        # backward-data (needs dy rows 0..2 resp. 2H-4..2H-1) and R = the resize transpose ALONG the line; columns alike
        strips = []
        for axis, first in ((1, True), (1, False), (2, True), (2, False)):
            n_dy, n_up = (3, 2) if first else (4, 3)                      # strip widths: dy lines read, d_up lines produced
            O_ = OH if axis == 1 else OW
            sl = slice(0, n_dy) if first else slice(O_ - n_dy, O_)
            pad = 1 if first else 2                                        # forward-convolution padding that aligns the strip (see csrc/igemm.hip)
            if axis == 1:
                geo = (N, n_up, OW, self.cin, n_dy, OW, self.cout, 3, 3, 1, pad, 1)
                view, s, dup = dy[:, sl], ctx.empty(N, n_dy, OW, self.cout), ctx.empty(N, n_up, OW, self.cin)
            else:
                geo = (N, OH, n_up, self.cin, OH, n_dy, self.cout, 3, 3, 1, 1, pad)
                view, s, dup = dy[:, :, sl], ctx.empty(N, OH, n_dy, self.cout), ctx.empty(N, OH, n_up, self.cin)
            strips.append((axis, first, geo, view, s, dup))
        # (Round 4 tried the four strips -- 256-384 tiles of the gather kernel each: a quarter of the chip's workgroup slots -- on four side
        # streams beside each other in the fp32 build: 3 125 -> 2 845 img/s.  Every cross-stream dependency drains both queues on this runtime;
        # 18 of them per iteration cost far more than the 0.5 ms the overlap could save.)
        _timed(256120 + ctx.ns, flops, "ladder_conv3x3_up2_bwd_data_split",
               (_p(dy), _p(dy_amax), _p(pk4), _p(dx), _p(dx_amax), N, H, W, self.cout, self.cin, ctx.ns, st), flops * 25.0 / 36.0)
        for axis, first, geo, view, s, dup in strips:
            s.copy_(view)
            ctx.set_amax(s, dy_amax)                                       # (a strip of dy: the per-sample record of dy bounds it, no extra pass)
            s_amax = ctx.absmax(s)
            wsp, wsn = ctx.ws(L.query("ladder_conv2d_bwd_data_split_workspace_bytes", *geo))
            L.call("ladder_conv2d_bwd_data_split", _p(ctx.planes(s, self._ps(geo[1], geo[2]))), _p(s_amax), _p(pk), _p(dup), *geo, None, 0, ctx.ns, wsp, wsn, st)
            L.call("ladder_conv3x3_up2_bwd_border", _p(dup), _p(dx), _p(dx_amax), N, H, W, self.cin, axis, 1 if first else 0, st)
        ctx.set_amax(dx, dx_amax)
        return dx

    def backward(self, dy, need_dx=True, wgrad=True, act_done=False, gate_prev=None, lowres_dx=False, lowres_gate=None, proj_grad=None):
        """`act_done`: dy already carries this layer's activation derivative (fused into the consumer's epilogue).
        `proj_grad` = (dyp, proj) instead of dy: see _backward_proj / bwd_proj_ok.
        `gate_prev`: activation name of the layer that produced this conv's input x: its derivative act'(x) is fused into
        the backward-data epilogue, so that layer must then be called with act_done=True.
        `lowres_dx`: x is the factor-2 upsample of a tensor the caller wants the gradient of: return d / d (that tensor) (see _dx_lowres)."""
        x, y = self.x, self.y
        N, H, W, _ = x.shape
        if self.x_is_lo:                                  # x is the low-resolution tensor: the layer's input is its (never materialised) upsample
            H, W = self.lo_factor * H, self.lo_factor * W
            if not lowres_dx and need_dx:
                raise RuntimeError("%s: only the low-resolution gradient exists for a virtual upsample" % self.name)
        if proj_grad is not None:
            if not (wgrad and self.bwd_proj_ok(proj_grad[1])):
                raise RuntimeError("%s: the projection-gradient form of the backward combination does not apply" % self.name)
            return self._backward_proj(None, need_dx, wgrad, lowres_gate, proj_grad)
        _, Ho, Wo, _ = y.shape
        st = self.ctx.stream
        if self.act is not None and not act_done:
            L.call("ladder_act_bwd", _p(dy), _p(y), _p(dy), dy.numel(), L.ACT[self.act], st)
        if self.x_is_lo and self.proj_ok(N, H // self.lo_factor, W // self.lo_factor, self.lo_factor):
            return self._backward_proj(dy, need_dx, wgrad, lowres_gate)
        if (wgrad and self.k == 1 and self.stride == 1 and L.query("ladder_conv1x1_smallcout_eligible", N * H * W, self.cin, self.cout)):
            # 1x1 to <= 4 channels over a wide map (the CelebA output conv): dx, dW and db from ONE pass over x
            M = N * H * W
            wsp, wsn = self.ctx.ws(L.query("ladder_conv1x1_smallcout_bwd_workspace_bytes", M, self.cin, self.cout))
            dx = self.ctx.empty(N, H, W, self.cin) if need_dx else None
            dx_amax = self.ctx.new_amax() if (need_dx and self.ctx.ns == 4) else None
            L.call("ladder_conv1x1_smallcout_bwd_absmax", _p(x), _p(dy), _p(self.ps.w[self.name + "/kernel"]), _p(dx),
                   _p(self.ps.g[self.name + "/kernel"]), _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, M, self.cin,
                   self.cout, L.ACT[gate_prev] if gate_prev else 0, wsp, wsn, _p(dx_amax), H * W, st)
            if dx is not None:
                self.ctx.set_amax(dx, dx_amax)
            self.x = self.y = None
            return dx
        if self._as_dense(N * H * W):
            M = N * H * W
            if wgrad:
                L.call("ladder_dense_bwd_weight_small" + self.ctx.sfx, _p(x), _p(dy), _p(self.ps.g[self.name + "/kernel"]),
                       _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, M, self.cin, self.cout, st)
            dx = None
            if need_dx:
                dx = self.ctx.empty(N, H, W, self.cin)
                L.call("ladder_dense_bwd_data_small" + self.ctx.sfx, _p(dy), _p(self.ps.w[self.name + "/kernel"]), _p(dx), M, self.cin, self.cout,
                       _p(x) if gate_prev else None, L.ACT[gate_prev] if gate_prev else 0, st)
            self.x = self.y = None
            return dx
        if wgrad and self._rgb(N, H, W):
            xa, da = self.ctx.absmax(x), self.ctx.absmax(dy)
            with self.ctx.side_or_main(x, dy, xa, da):
                wsp, wsn = self.ctx.ws(L.query("ladder_conv_rgb_s2_bwd_filter_workspace_bytes", N, H, W, self.cout))
                L.call("ladder_conv_rgb_s2_bwd_filter", _p(x), _p(xa), _p(dy), _p(da), _p(self.ps.g[self.name + "/kernel"]),
                       _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, N, H, W, self.cout, wsp, wsn, self.ctx.stream)
            wgrad = False
        if wgrad and self._rgb_fwd32(N, H, W):            # strict fp32: the fp32 filter-gradient kernel of the same layer
            with self.ctx.side_or_main(x, dy):
                wsp, wsn = self.ctx.ws(L.query("ladder_conv_rgb_s2_bwd_filter_workspace_bytes", N, H, W, self.cout))
                L.call("ladder_conv_rgb_s2_bwd_filter_f32", _p(x), _p(dy), _p(self.ps.g[self.name + "/kernel"]),
                       _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, N, H, W, self.cout, wsp, wsn, self.ctx.stream)
            wgrad = False
        dy_amax = None
        split_w = bool(wgrad and self._split_ok(N, H, W, self.cin, self.cout)
                       and L.query("ladder_conv3x3_wgrad_split_eligible", N, H, W, self.cin, self.cout, self.ctx.ns))
        if split_w and self.ctx.ns == 4 and getattr(self, "x_amax", None) is None:
            self.x_amax = self.ctx.absmax(x)
        split_d = bool(need_dx and not gate_prev and self._halo_ok(N, Ho, Wo, self.cout, self.cin))
        if split_w or split_d:
            dy_amax = self.ctx.absmax(dy)               # one pass serves the filter gradient and the backward-data call
        if split_w:
            # (on the side stream: the filter gradient is MFMA-bound and only the optimiser step needs it; the backward-data call below
            # and the HBM-bound resize / norm backward kernels of the layers underneath run beside it)
            with self.ctx.side_or_main(x, dy, self.x_amax, dy_amax):
                wsp, wsn = self.ctx.ws(L.query("ladder_conv3x3_wgrad_split_workspace_bytes", N, H, W, self.cin, self.cout))
                args = (_p(x), _p(self.x_amax), _p(dy), _p(dy_amax), _p(self.ps.g[self.name + "/kernel"]),
                        _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, N, H, W, self.cin, self.cout, self.ctx.ns, wsp, wsn,
                        self.ctx.stream)
                _timed(9120 + self.ctx.ns, 2.0 * N * H * W * 9 * self.cin * self.cout, "ladder_conv3x3_wgrad_split", args)
        elif wgrad and self.ctx.ns and L.query("ladder_conv2d_bwd_filter_split_eligible", N, H, W, self.cin, Ho, Wo, self.cout, self.k,
                                               self.k, self.stride, self.pt, self.pl):
            if self.ctx.ns == 4:
                if getattr(self, "x_amax", None) is None:
                    self.x_amax = self.ctx.absmax(x)
                dy_amax = self.ctx.absmax(dy)
            ps_ = self._ps(Ho, Wo)
            xpl, dpl = self.ctx.planes(x, ps_), self.ctx.planes(dy, ps_)          # (split on the main stream: backward-data reads dy's planes too)
            with self.ctx.side_or_main(x, dy, xpl, dpl, self.x_amax, dy_amax):
                wsp, wsn = self.ctx.ws(L.query("ladder_conv2d_bwd_filter_split_workspace_bytes", N, H, W, self.cin, Ho, Wo, self.cout, self.k,
                                               self.k))
                L.call("ladder_conv2d_bwd_filter_split", _p(xpl), _p(self.x_amax), _p(dpl), _p(dy_amax), _p(self.ps.g[self.name + "/kernel"]),
                       _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k,
                       self.stride, self.pt, self.pl, self.ctx.ns, wsp, wsn, self.ctx.stream)
        elif (wgrad and self.ctx.ns == 0 and (self.x_is_up2 or self.x_is_lo) and self.ctx.up2 >= 2 and (H // 2) * (W // 2) >= UP2W_MIN_PIXELS
              and L.query("ladder_conv3x3_up2_wgrad_eligible", N, H // 2, W // 2, self.cin, self.cout)):
            # strict fp32, x = resize2x(x_lo): 25 instead of 36 tap tiles, read from the even sub-grid of the kept upsample (csrc/convf32.hip)
            wsp, wsn = self.ctx.ws(L.query("ladder_conv3x3_up2_wgrad_workspace_bytes", N, H // 2, W // 2, self.cin, self.cout))
            fl = 2.0 * N * H * W * 9 * self.cin * self.cout
            self.ctx.up2_used[self.name + ":wgrad"] = self.ctx.up2_used.get(self.name + ":wgrad", 0) + 1
            self.ctx.up2_skipped[self.name + ":wgrad"] = 11.0 / 36.0
            _timed(9120, fl, "ladder_conv3x3_up2_wgrad",
                   (_p(x), 0 if self.x_is_lo else 1, _p(dy), _p(self.ps.g[self.name + "/kernel"]), _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None, N, H // 2, W // 2,
                    self.cin, self.cout, wsp, wsn, st), fl * 25.0 / 36.0)
        elif wgrad:
            if self.x_is_lo:        # (x is [N, H/2, W/2, cin]: the generic kernel would read 4x past it -- the forward's decision must hold here)
                raise RuntimeError("%s: virtual upsample without the low-resolution filter gradient" % self.name)
            nb = L.query("ladder_conv2d_bwd_filter_workspace_bytes", N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k)
            wsp, wsn = self.ctx.ws(nb)
            wargs = (_p(x), _p(dy), _p(self.ps.g[self.name + "/kernel"]), _p(self.ps.g[self.name + "/bias"]) if self.bias_grad else None,
                     N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride, self.pt, self.pl, wsp, wsn, st)
            if PROF is not None and L.query("ladder_conv2d_bwd_filter_kernel_id", N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k,
                                            self.stride, self.pt, self.pl) == 9128:
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                L.call("ladder_conv2d_bwd_filter", *wargs)         # (+ its fixed-order split reduction: counted with the kernel)
                e_.record()
                PROF.add(9128, s_, e_, 2.0 * N * Ho * Wo * self.k * self.k * self.cin * self.cout)
            else:
                L.call("ladder_conv2d_bwd_filter", *wargs)
        dx = None
        if lowres_dx and not split_d:
            raise RuntimeError("%s: the low-resolution backward-data was requested for a call the split halo kernels do not take" % self.name)
        if split_d and lowres_dx:
            dx = self._dx_lowres(dy, dy_amax, lowres_gate)
        elif split_d:
            dx = self.ctx.empty(N, H, W, self.cin)
            dx_amax = self.ctx.new_amax() if self.ctx.ns == 4 else None
            args = (_p(dy), _p(dy_amax), _p(self._packed_filter(1)), None, _p(dx), _p(dx_amax), N, H, W, self.cout, self.cin, 0,
                    self.ctx.ns, st)
            self.ctx.set_amax(dx, dx_amax)
            _timed(self._halo_kid(N, H, W, self.cout, self.cin), 2.0 * N * H * W * 9 * self.cin * self.cout, "ladder_conv3x3_split", args)
        elif (need_dx and self.ctx.ns in (0, 4) and not gate_prev and self.stride == 2 and not (self.ctx.ns == 0 and os.environ.get("LADDER_DISABLE_HALO") == "1")
              and (L.query("ladder_conv3x3_s2_bwd_data_split_eligible", N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride,
                           self.pt, self.pl) or (self.ctx.ns == 0 and self.k == 3 and self.pt == 0 and self.pl == 0 and
                                                 L.query("ladder_conv3x3_s2_bwd_data_f32_eligible", N, H, W, self.cin, Ho, Wo, self.cout)))):
            # 3x3 / stride 2 over a map whose gradient is halo-kernel sized (enc.conv1): the four output-parity classes in ONE launch
            if dy_amax is None:
                dy_amax = self.ctx.absmax(dy)
            dx = self.ctx.empty(N, H, W, self.cin)
            dx_amax = self.ctx.new_amax() if self.ctx.ns == 4 else None
            args = (_p(dy), _p(dy_amax), _p(self._packed_filter(2)), _p(dx), _p(dx_amax), N, H, W, self.cin, Ho, Wo, self.cout, self.ctx.ns, st)
            self.ctx.set_amax(dx, dx_amax)
            _timed(self._halo_kid(N, Ho, Wo, self.cout, 4 * self.cin, self.cin), 2.0 * N * Ho * Wo * 9 * self.cin * self.cout, "ladder_conv3x3_s2_bwd_data_split", args)
        elif need_dx and self.ctx.ns and L.query("ladder_conv2d_bwd_data_split_eligible", N, H, W, self.cin, Ho, Wo, self.cout, self.k,
                                                 self.k, self.stride, self.pt, self.pl, 1 if gate_prev else 0):
            geo = (N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k, self.stride, self.pt, self.pl)
            if dy_amax is None:
                dy_amax = self.ctx.absmax(dy)
            wsp, wsn = self.ctx.ws(L.query("ladder_conv2d_bwd_data_split_workspace_bytes", *geo))
            dx = self.ctx.empty(N, H, W, self.cin)
            L.call("ladder_conv2d_bwd_data_split", _p(self.ctx.planes(dy, self._ps(Ho, Wo))), _p(dy_amax), _p(self._packed_filter(1)), _p(dx), *geo,
                   _p(x) if gate_prev else None, L.ACT[gate_prev] if gate_prev else 0, self.ctx.ns, wsp, wsn, st)
        elif need_dx:
            if self.ctx.ns == 0 and self.cout % 16 == 0 and self.k == 3:
                wT = self._packed_filter(1)               # flipped / transposed fp32 bank, re-packed with all others in one launch per step
            else:
                w = self.ps.w[self.name + "/kernel"]
                wT = self.ctx.empty(w.numel())
                L.call("ladder_filter_flip_transpose", _p(w), _p(wT), self.k, self.k, self.cin, self.cout, st)
            dx = self.ctx.empty(N, H, W, self.cin)
            _igemm(self.ctx, "ladder_conv2d_bwd_data", N * H * W, self.cout, self.cin, self.k * self.k * self.cout,
                   _p(dy), _p(wT), _p(dx), N, H, W, self.cin, Ho, Wo, self.cout, self.k, self.k,
                   self.stride, self.pt, self.pl, _p(x) if gate_prev else None, L.ACT[gate_prev] if gate_prev else 0,
                   # a strided backward-data call is several parity-class launches: not attributed by the profiler
                   conv=(N, Ho, Wo, self.cout, H, W, self.cin, self.k, self.k, 1, 1, self.k - 1 - self.pt, self.k - 1 - self.pl)
                   if self.stride == 1 else "skip")
        self.x = self.y = None
        return dx


class Dense:
    """tf.layers.dense, W [in,out]; MFMA-f32 GEMM with bias + activation epilogue."""

    def __init__(self, ctx, ps, name, cin, cout, act=None):
        self.ctx, self.ps, self.name, self.cin, self.cout, self.act = ctx, ps, name, cin, cout, act

    def _small(self, M):
        """Batch-sized layer: the one-launch kernels of csrc/densesplit.hip (strict fp32 MFMA when matmul_precision is "f32", else bf16x6)."""
        return bool(L.query("ladder_dense_small_eligible", M, self.cin, self.cout))

    def forward(self, x):
        M = x.shape[0]
        y = self.ctx.empty(M, self.cout)
        if self._small(M):
            L.call("ladder_dense_fwd_small" + self.ctx.sfx, _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y),
                   M, self.cin, self.cout, L.ACT[self.act], self.ctx.stream)
            self.x, self.y = x, y
            return y
        _igemm(self.ctx, "ladder_dense_fwd", M, self.cin, self.cout, self.cin,
               _p(x), _p(self.ps.w[self.name + "/kernel"]), _p(self.ps.w[self.name + "/bias"]), _p(y),
               M, self.cin, self.cout, L.ACT[self.act])
        self.x, self.y = x, y
        return y

    def backward(self, dy, need_dx=True, wgrad=True, act_done=False, gate_prev=None):
        x, y, st = self.x, self.y, self.ctx.stream
        M = x.shape[0]
        if self.act is not None and not act_done:
            L.call("ladder_act_bwd", _p(dy), _p(y), _p(dy), dy.numel(), L.ACT[self.act], st)
        if self._small(M):
            if wgrad and need_dx:                               # both gradient GEMMs in one launch
                dx = self.ctx.empty(M, self.cin)
                L.call("ladder_dense_bwd_small" + self.ctx.sfx, _p(x), _p(dy), _p(self.ps.w[self.name + "/kernel"]), _p(dx),
                       _p(self.ps.g[self.name + "/kernel"]), _p(self.ps.g[self.name + "/bias"]), M, self.cin, self.cout,
                       _p(x) if gate_prev else None, L.ACT[gate_prev] if gate_prev else 0, st)
                self.x = self.y = None
                return dx
            if wgrad:
                L.call("ladder_dense_bwd_weight_small" + self.ctx.sfx, _p(x), _p(dy), _p(self.ps.g[self.name + "/kernel"]),
                       _p(self.ps.g[self.name + "/bias"]), M, self.cin, self.cout, st)
            dx = None
            if need_dx:
                dx = self.ctx.empty(M, self.cin)
                L.call("ladder_dense_bwd_data_small" + self.ctx.sfx, _p(dy), _p(self.ps.w[self.name + "/kernel"]), _p(dx), M, self.cin, self.cout,
                       _p(x) if gate_prev else None, L.ACT[gate_prev] if gate_prev else 0, st)
            self.x = self.y = None
            return dx
        if wgrad:
            nb = L.query("ladder_dense_bwd_weight_workspace_bytes", M, self.cin, self.cout)
            wsp, wsn = self.ctx.ws(nb)
            L.call("ladder_dense_bwd_weight", _p(x), _p(dy), _p(self.ps.g[self.name + "/kernel"]),
                   _p(self.ps.g[self.name + "/bias"]), M, self.cin, self.cout, wsp, wsn, st)
        dx = None
        if need_dx:
            w = self.ps.w[self.name + "/kernel"]
            wT = self.ctx.empty(w.numel())
            L.call("ladder_filter_flip_transpose", _p(w), _p(wT), 1, 1, self.cin, self.cout, st)
            dx = self.ctx.empty(M, self.cin)
            _igemm(self.ctx, "ladder_dense_bwd_data", M, self.cout, self.cin, self.cout, _p(dy), _p(wT), _p(dx), M, self.cin, self.cout,
                   _p(x) if gate_prev else None, L.ACT[gate_prev] if gate_prev else 0)
        self.x = self.y = None
        return dx


class BatchNormAct:
    """tf.layers.batch_normalization(training=True) + activation; statistics of the GLOBAL batch (C2)."""

    def __init__(self, ctx, ps, name, C, act):
        self.ctx, self.ps, self.name, self.C, self.act = ctx, ps, name, C, act

    def forward(self, x, sums=None, planes=(False, True)):
        """`sums`: the statistics record of x when its producer already computed it (conv epilogue: 2C doubles sum | sum of squares, then min | max),
        else a pass over x.  `planes` = (emit the fp16 plane images of y, also keep y in fp32): with the per-channel extremes max|y| is
        known before y is written, so the apply kernel can split it on the fly -- and when nothing needs the fp32 tensor it is never
        written (returns a PlanesOnly stand-in)."""
        C, ctx = self.C, self.ctx
        rows = x.numel() // C
        want_planes, need_fp32 = planes
        want_planes = bool(want_planes and ctx.ns == 4 and C % 4 == 0 and x.numel() % 8 == 0)
        if sums is None:
            nb = L.query("ladder_bn_workspace_bytes", rows, C)
            if want_planes:
                wsp, wsn = ctx.ws(2 * nb)
                sums = ctx.empty(6 * C)
                L.call("ladder_bn_fwd_stats_minmax", _p(x), _p(sums), rows, C, wsp, wsn, ctx.stream)
            else:
                wsp, wsn = ctx.ws(nb)
                sums = ctx.empty(4 * C)
                L.call("ladder_bn_fwd_stats", _p(x), _p(sums), rows, C, wsp, wsn, ctx.stream)
        # the statistics RECORD (csrc/norm.hip): 2C doubles = sum x | sum x^2 (the first 4C floats of the buffer), then optionally min | max as
        # 2C floats.  C2 all-reduces the doubles: E[x^2] - mean^2 in fp64 keeps the variance exact on channels far off zero (TF's fused batch
        # norm centres first, reference codes/models.py:398-460) -- 2 KB per layer instead of 1 KB.
        want_planes = want_planes and sums.numel() == 6 * C
        ctx.comm.allreduce_(sums[:4 * C].view(torch.float64), "C2 fwd " + self.name.split("/")[-1])   # (the extremes stay local: they bound THIS rank's tensor)
        self.count = float(rows) * ctx.comm.world
        self.mean_rstd = ctx.empty(2 * C)
        gam, bet = self.ps.w[self.name + "/gamma"], self.ps.w[self.name + "/beta"]
        if want_planes:
            y = torch.empty_like(x) if need_fp32 else None
            out = y if need_fp32 else PlanesOnly(x.shape)
            buf = torch.empty(L.query("ladder_presplit_bytes", x.numel(), ctx.ns), dtype=torch.uint8, device=ctx.device)
            y_amax = ctx.new_amax()
            L.call("ladder_bn_fwd_apply_planes", _p(x), _p(sums), self.count, _p(gam), _p(bet), _p(y), _p(buf), _p(self.mean_rstd), rows, C,
                   BN_EPS, L.ACT[self.act], _p(y_amax), ctx.stream)
            ctx.set_amax(out, y_amax)
            ctx.set_planes(out, buf)
            self.x = x
            return out
        y = torch.empty_like(x)
        y_amax = ctx.new_amax() if (ctx.ns == 4 and C % 4 == 0) else None
        L.call("ladder_bn_fwd_apply_absmax", _p(x), _p(sums), self.count, _p(gam), _p(bet), _p(y), _p(self.mean_rstd), rows, C, BN_EPS,
               L.ACT[self.act], _p(y_amax), ctx.stream)
        ctx.set_amax(y, y_amax)
        self.x = x
        return y

    def backward(self, dy, need_dx=True, wgrad=True):
        C, ctx, x = self.C, self.ctx, self.x
        rows = x.numel() // C
        gam, bet = self.ps.w[self.name + "/gamma"], self.ps.w[self.name + "/beta"]
        nb = L.query("ladder_bn_workspace_bytes", rows, C)
        wsp, wsn = ctx.ws(nb)
        dsums = ctx.empty(2 * C)
        L.call("ladder_bn_bwd_stats", _p(dy), _p(x), _p(self.mean_rstd), _p(gam), _p(bet), _p(dsums), rows, C,
               L.ACT[self.act], wsp, wsn, ctx.stream)
        ctx.comm.allreduce_(dsums, "C2 bwd " + self.name.split("/")[-1])
        dx = torch.empty_like(x) if need_dx else None
        # dgamma/dbeta are global sums already: written on every rank, the group all-reduce must not re-sum
        # them -> the engine divides BN parameter grads by world size before the flat all-reduce.
        dx_amax = ctx.new_amax() if (need_dx and ctx.ns == 4 and C % 4 == 0) else None
        L.call("ladder_bn_bwd_apply_absmax", _p(dy), _p(x), _p(self.mean_rstd), _p(gam), _p(bet), _p(dsums), self.count, _p(dx),
               _p(self.ps.g[self.name + "/gamma"]) if wgrad else None, _p(self.ps.g[self.name + "/beta"]) if wgrad else None,
               rows, C, L.ACT[self.act], _p(dx_amax), ctx.stream)
        if dx is not None:
            ctx.set_amax(dx, dx_amax)
        if wgrad and ctx.comm.world > 1:
            for t in (self.ps.g[self.name + "/gamma"], self.ps.g[self.name + "/beta"]):
                L.call("ladder_axpy", _p(t), _p(t), t.numel(), 1.0 / ctx.comm.world, 0, ctx.stream)
        self.x = None
        return dx


class InstanceNormStyleAct:
    """instance_norm(center=False, scale=False) -> style_mod -> activation (models.py:522-528 ..., modules.py:6-10)."""

    def __init__(self, ctx, C, act):
        self.ctx, self.C, self.act = ctx, C, act

    def forward(self, x, style):
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        self.mean_rstd = self.ctx.empty(N, 2 * C)
        wsp, wsn = self.ctx.ws(L.query("ladder_in_style_workspace_bytes", N, H * W, C))
        y_amax = self.ctx.new_amax() if (self.ctx.ns == 4 and C % 4 == 0) else None
        L.call("ladder_in_style_fwd_absmax", _p(x), _p(style), _p(y), _p(self.mean_rstd), N, H * W, C, IN_EPS, L.ACT[self.act],
               wsp, wsn, _p(y_amax), self.ctx.stream)
        self.ctx.set_amax(y, y_amax)
        self.x, self.style = x, style
        return y

    def forward_resized(self, x, style, rs, keep_lowres=False):
        """forward() followed by the factor-2 resize `rs` in ONE pass over x (ladder_in_style_fwd_resize2x): the normalised tensor is
        never written -- unless `keep_lowres` (then it is left in self.y_lo: the input of an upsample-fused convolution behind the resize);
        returns None when the pair is not eligible.  The backward passes are those of the two separate layers."""
        N, H, W, C = x.shape
        self.y_lo = None
        if not (C % 4 == 0 and (rs.oh, rs.ow) == (2 * H, 2 * W)):
            return None
        up = self.ctx.empty(N, 2 * H, 2 * W, C)
        self.mean_rstd = self.ctx.empty(N, 2 * C)
        wsp, wsn = self.ctx.ws(L.query("ladder_in_style_workspace_bytes", N, H * W, C))
        up_amax = self.ctx.new_amax() if self.ctx.ns == 4 else None
        if keep_lowres:
            self.y_lo = self.ctx.empty(N, H, W, C)
            L.call("ladder_in_style_fwd_resize2x_keep", _p(x), _p(style), _p(up), _p(self.y_lo), _p(self.mean_rstd), N, H, W, C, IN_EPS,
                   L.ACT[self.act], wsp, wsn, _p(up_amax), self.ctx.stream)
            self.ctx.set_amax(self.y_lo, up_amax)              # max |y| = max |up| (the record receives max |y|)
        else:
            L.call("ladder_in_style_fwd_resize2x", _p(x), _p(style), _p(up), _p(self.mean_rstd), N, H, W, C, IN_EPS, L.ACT[self.act], wsp, wsn,
                   _p(up_amax), self.ctx.stream)
        self.ctx.set_amax(up, up_amax)
        self.x, self.style = x, style
        rs.in_shape = (N, H, W, C)
        return up

    def backward(self, dy):
        x = self.x
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        dstyle = self.ctx.empty(N, 2 * C)
        wsp, wsn = self.ctx.ws(L.query("ladder_in_style_workspace_bytes", N, H * W, C))
        dx_amax = self.ctx.new_amax() if (self.ctx.ns == 4 and C % 4 == 0) else None
        L.call("ladder_in_style_bwd_absmax", _p(dy), _p(x), _p(self.style), _p(self.mean_rstd), _p(dx), _p(dstyle), N, H * W, C,
               L.ACT[self.act], wsp, wsn, _p(dx_amax), self.ctx.stream)
        self.ctx.set_amax(dx, dx_amax)
        self.x = self.style = None
        return dx, dstyle


class Resize:
    """tf.image.resize_images TF1-legacy bilinear."""

    def __init__(self, ctx, oh, ow):
        self.ctx, self.oh, self.ow = ctx, oh, ow

    def forward(self, x):
        N, H, W, C = x.shape
        self.in_shape = (N, H, W, C)
        if (H, W) == (self.oh, self.ow):
            return x
        y = self.ctx.empty(N, self.oh, self.ow, C)
        L.call("ladder_resize_bilinear_fwd", _p(x), _p(y), N, H, W, C, self.oh, self.ow, self.ctx.stream)
        self.ctx.set_amax(y, self.ctx.known_amax(x))        # bilinear interpolation is a convex combination: max|y| <= max|x|
        return y

    def backward(self, dy, gate=None):
        """`gate` = (y, act) of the layer that produced the resized tensor: its activation backward is applied to dx in the same pass
        (factor-2 resizes; returns (dx, True) then, so that the caller skips that layer's own activation backward)."""
        N, H, W, C = self.in_shape
        if (H, W) == (self.oh, self.ow):
            return (dy, False) if gate is not None else dy
        dx = self.ctx.empty(N, H, W, C)
        if gate is not None and (self.oh, self.ow) == (2 * H, 2 * W) and gate[1] is not None:
            L.call("ladder_resize_bilinear_bwd_gated", _p(dy), _p(dx), N, H, W, C, self.oh, self.ow, _p(gate[0]), L.ACT[gate[1]],
                   self.ctx.stream)
            return dx, True                                     # (no absmax record: the gate rescales elements)
        L.call("ladder_resize_bilinear_bwd", _p(dy), _p(dx), N, H, W, C, self.oh, self.ow, self.ctx.stream)
        rec = self.ctx.known_amax(dy)
        if rec is not None:      # the transpose sums interpolation weights: column sums are bounded per axis (arch.resize_transpose_gain)
            self.ctx.set_amax(dx, rec * float(arch.resize_transpose_gain(H, self.oh) * arch.resize_transpose_gain(W, self.ow)))
        return (dx, False) if gate is not None else dx


class DepthToSpace:
    def __init__(self, ctx, r):
        self.ctx, self.r = ctx, r

    def forward(self, x):
        N, H, W, C = x.shape
        r = self.r
        y = self.ctx.empty(N, H * r, W * r, C // (r * r))
        L.call("ladder_depth_to_space", _p(x), _p(y), N, H, W, C, r, 0, self.ctx.stream)
        return y

    def backward(self, dy):
        N, HR, WR, Cp = dy.shape
        r = self.r
        dx = self.ctx.empty(N, HR // r, WR // r, Cp * r * r)
        L.call("ladder_depth_to_space", _p(dy), _p(dx), N, HR // r, WR // r, Cp * r * r, r, 1, self.ctx.stream)
        return dx


def pad_symmetric(ctx, x, p):
    N, H, W, C = x.shape
    y = ctx.empty(N, H + 2 * p, W + 2 * p, C)
    L.call("ladder_pad_symmetric", _p(x), _p(y), N, H, W, C, p, ctx.stream)
    return y


def add_(ctx, out, inp):
    L.call("ladder_axpy", _p(inp), _p(out), out.numel(), 1.0, 1, ctx.stream)
    ctx.drop_amax(out)                       # values changed in place: a registered absolute-maximum record no longer bounds them
    return out
