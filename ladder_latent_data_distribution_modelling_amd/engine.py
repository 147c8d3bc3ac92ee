"""Host-side driver of the HIP LaDDer path: parameter store, layers with explicit backward,
the three reference architectures and the four per-minibatch "runs" of the reference iteration
(codes/base.py:583-641).  PyTorch supplies device memory, streams and torch.distributed only;
every arithmetic op below is a call into libladder_hip.so (see _lib.py) -- there is no CPU path.
"""
import math

import numpy as np
import os

import torch

from . import _lib as L
from . import arch

BN_EPS = 1e-3      # tf.layers.batch_normalization default
IN_EPS = 1e-6      # tf.contrib.layers.instance_norm default
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.95, 1e-8   # codes/base.py:459-461


def _p(t):
    return None if t is None else t.data_ptr()


class KernelProfiler:
    """HIP-event timing of the forward-type contraction launches, grouped by the kernel they dispatch to (bench.py's
    roofline leg).  Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    NAMES = {256128: "conv3x3_halo_kernel (8x32 px x 128 ch LDS-halo tile, 8 waves, fp32 MFMA 32x32x2)",
             256120: "conv3x3_halo_f32_kernel<PROJ, UPM> (8x32 px x 128 ch LDS-halo tile, 8 waves, fp32 MFMA 32x32x2; instantiations: plain, "
                     "fused 1x1 projection, upsample-fused forward with / without projection, upsample-fused backward-data, stride-2 backward-data)",
             256123: "conv3x3_halo_split_kernel<bf16x6> (8x32 px x 128 ch LDS-halo tile, 8 waves, fp32 operands as 3 bf16 planes, 6 x MFMA 32x32x16 bf16)",
             256122: "conv3x3_halo_split16_kernel / conv3x3_halo_split_kernel<bf16x3> (16x32 px x 128 ch LDS-halo tile, 16 waves -- 8x32, 8 waves below 512 tiles; fp32 operands as 2 bf16 planes, 3 x MFMA 32x32x16 bf16)",
             256124: "conv3x3_halo_split16_kernel / conv3x3_halo_split_kernel<f16x3> (16x32 px x 128 ch LDS-halo tile, 16 waves -- 8x32, 8 waves below 512 tiles; fp32 operands as 2 scaled fp16 planes, 3 x MFMA 32x32x16 f16)",
             128128: "igemm_fwd_kernel<128,128,2,2,true,true> (gather implicit GEMM, fp32 MFMA 32x32x2)",
             64064: "igemm_fwd_kernel<64,64,2,2,true,true> (implicit GEMM, 64x64 tile, fp32 MFMA 32x32x2; here: the 128- / 512-row GEMMs of the projected pairs conv2d_1 / conv2d_3)",
             128124: "igemm_fwd_split_kernel<f16x3> (gather implicit GEMM, 128x128 tile, 3 x MFMA 32x32x16 f16)",
             128122: "igemm_fwd_split_kernel<bf16x3>", 128123: "igemm_fwd_split_kernel<bf16x6>",
             9003: "conv_smallcin_kernel (direct 1x1 from 3 channels, HBM-bound)",
             9124: "wgrad3x3_split_kernel<f16x3> (64ci x 128co slab x 9 taps, 8 waves x 9 tiles, transposing LDS reads, 3 x MFMA 32x32x16 f16; incl. its split reduction)",
             9122: "wgrad3x3_split_kernel<bf16x3>", 9123: "wgrad3x3_split_kernel<bf16x6>",
             9128: "wgrad3x3_halo_kernel (64ci x 128co slab x 9 taps, 12 waves, LDS-DMA staged 1x32-pixel patches, fp32 MFMA 32x32x2)",
             256064: "conv3x3_halo_f32s_kernel<UPM, GEO, NI, FKS> (small maps: 16x16 / 8x16 / 8x8-px sub-patches x 64 / 128 ch LDS-halo tile, 8 waves, fp32 MFMA "
                     "32x32x2; instantiations: plain, upsample-fused forward (class pairs), parity-class input (upsample-fused backward-data, stride-2 forward), "
                     "stride-2 backward-data classes)",
             9120: "wgrad3x3_up2_f32_kernel + reductions / edge lines (filter gradient of resize x2 -> 3x3 conv over the low-resolution map: 25 of 36 tap tiles, "
                   "one parity class per workgroup, 12 waves, LDS-DMA staged 1x32-pixel patches, fp32 MFMA 32x32x2)",
             9130: "igemm_wgrad_kernel<128,128> + its fixed-order split reduction (dWcat [Cin][9 Cout] = x^T D of the project-then-upsample pairs, fp32 MFMA 32x32x2)",
             9132: "gemm_tn_f32_kernel + its fixed-order split reduction (dWcat [Cin][9 Cout] = x^T D of the project-then-upsample pairs: 128x128 tile x pixel range "
                   "per workgroup, 32-pixel chunks, software-pipelined LDS fragments, fp32 MFMA 32x32x2)",
             128132: "gemm_f32_kernel (projection GEMMs of the project-then-upsample pairs, Z = x . wcat: persistent workgroups over 128x128 "
                     "tiles, XCD-aware tile order, 32-deep chunks, 128-bit A fragments, software-pipelined LDS reads, fp32 MFMA 32x32x2)",
             128136: "up2proj_fused_fwd_kernel (forward of a project-then-upsample pair in ONE launch: per (64 / W images, 16 output channels) workgroup the "
                     "projection GEMM Z = x . wcat row by row on fp32 MFMA 16x16x4, the nine planes of three rows in an LDS ring, the combination + activation "
                     "(+ 1x1 output projection) from the ring; Z never written to HBM)",
             128134: "gemm_nt16_f32_kernel (backward-data GEMMs of the project-then-upsample pairs, dx = D . wcat^T with both operands K-contiguous: persistent "
                     "workgroups over 128x128 tiles, 128-bit fragments of both operands, fp32 MFMA 16x16x4)",
             7700: "gmm_logprob_kernel<R> + gmm_sum_kernel (mixture log-prob / responsibilities, lane = component, wave-shuffle logsumexp)"}
    LATENCY_BOUND = (7700,)          # not contraction kernels: reported beside the roofline, never as the dominant MFMA kernel

    def __init__(self):
        self.records = {}

    def add(self, kid, s, e, flops, executed=None):
        """`flops` = the reference's operation count of the launch (algorithmic); `executed` = what the kernel issues when that is less
        (the upsample-fused convolutions: 25 of the 36 low-resolution tap products per 2x2 output block)."""
        self.records.setdefault(kid, []).append((s, e, flops, flops if executed is None else executed))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kid, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            fl = sum(f for _, _, f, _ in recs)
            fx = sum(x for _, _, _, x in recs)
            n = len(recs)
            out[kid] = dict(kernel=self.NAMES.get(kid, "igemm tile %d" % kid), launches=n, total_ms=ms, avg_ms=ms / n,
                            flops_per_launch=fl / n, tflops=(fl / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                            executed_flops_per_launch=fx / n, executed_tflops=(fx / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                            bound="latency" if kid in self.LATENCY_BOUND else "mfma")
        return out


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


class Comm:
    """Data-parallel exchange steps C1-C4 of SURVEY 2.3 over torch.distributed (RCCL on ROCm).
    With world_size 1 every method is a no-op."""

    def __init__(self, group=None, deterministic=False):
        import torch.distributed as dist
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.group = group
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.trace = None          # label -> [(start event, end event, bytes, exposed-start event or None)]: see enable_trace()
        # deterministic (config key `deterministic_allreduce`): every all-reduce is an all-gather followed by the RANK-ORDERED sum
        # ((r0 + r1) + r2) + ... on every rank -- a summation order that does not depend on the backend's ring / tree schedule, so an N-rank
        # job is reproducible bit for bit by VirtualComm below ("fixed split order ... bit-stable").  N x the bytes of a ring
        # all-reduce: a parity / debugging mode, not the production setting.  (With two ranks ANY all-reduce is a + b: already bit-stable.)
        self.deterministic = bool(deterministic)

    def _ordered_sum_(self, t):
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t.contiguous(), group=self.group)
        t.copy_(parts[0])
        for q in parts[1:]:
            t.add_(q)
        return t

    def enable_trace(self, on=True):
        """Per-collective timing for `bench.py --gpus N` (the first multi-GPU run must be diagnosable).  Every exchange step
        is bracketed by HIP events on the COMPUTE stream: `wall` = from the point the collective is issued to the point the compute stream
        may continue behind it; for a blocking collective that is also the time it was EXPOSED (nothing else runs on the compute stream
        meanwhile); for the asynchronous C1 bucket `exposed` = from wait() to completion only -- the rest ran under backward kernels."""
        self.trace = {} if on else None

    def _ev(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def trace_summary(self, steps):
        """{label: calls / step, bytes / call, wall and exposed microseconds per call and per step} of the traced collectives."""
        if not self.trace:
            return {}
        torch.cuda.synchronize()
        out = {}
        for label, recs in self.trace.items():
            wall = sum(s.elapsed_time(e) for s, e, _, _ in recs) * 1e3
            expo = sum((x if x is not None else s).elapsed_time(e) for s, e, _, x in recs) * 1e3
            n = len(recs)
            out[label] = {"calls_per_step": round(n / steps, 2), "bytes_per_call": int(sum(b for _, _, b, _ in recs) / n),
                          "wall_us_per_call": round(wall / n, 1), "exposed_us_per_call": round(expo / n, 1),
                          "wall_us_per_step": round(wall / steps, 1), "exposed_us_per_step": round(expo / steps, 1)}
        return out

    def allreduce_(self, t, label="allreduce"):
        if self.on and self.deterministic:
            return self._ordered_sum_(t)
        if self.on:
            if self.trace is not None:
                s = self._ev()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
                self.trace.setdefault(label, []).append((s, self._ev(), t.numel() * t.element_size(), None))
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_async_(self, t, label="allreduce (async)"):
        """Start the all-reduce and return a handle (None with one rank): the collective runs on the process group's own
        stream, so kernels enqueued afterwards on the compute stream overlap it; `wait()` orders the compute stream after it."""
        if not self.on:
            return None
        if self.deterministic:
            self._ordered_sum_(t)
            return _DoneWork()
        if self.trace is None:
            return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        s = self._ev()
        return _TracedWork(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), self, label, s,
                           t.numel() * t.element_size())

    def broadcast_(self, t, src=0):
        if self.on:
            self.dist.broadcast(t, src=src, group=self.group)
        return t


class _TracedWork:
    """Handle of a traced asynchronous collective: wait() records when the compute stream started to wait and when it could continue."""

    def __init__(self, work, comm, label, start, nbytes):
        self.work, self.comm, self.label, self.start, self.nbytes = work, comm, label, start, nbytes

    def wait(self):
        x = self.comm._ev()
        self.work.wait()
        self.comm.trace.setdefault(self.label, []).append((self.start, self.comm._ev(), self.nbytes, x))


class _DoneWork:
    """Handle of a collective that completed inside the call."""

    def wait(self):
        return None


class VirtualGroup:
    """Shared state of `world` VIRTUAL ranks: engines that run in `world` Python threads of one process on one device and ONE HIP stream (the
    default stream of every thread), so host issue order is device execution order."""

    def __init__(self, world):
        import threading
        self.world = int(world)
        self.barrier = threading.Barrier(self.world)
        # a rank that issues fewer collectives than the others (or returns early) must FAIL the job, not hang it: every wait has a time-out
        # (threading.BrokenBarrierError in all ranks) and run_virtual_ranks aborts the barrier as soon as one rank's function has returned
        self.timeout = float(os.environ.get("LADDER_VIRTUAL_BARRIER_TIMEOUT", "300"))
        self.lock, self.finished = threading.Lock(), 0
        self.slots = [None] * self.world

    def wait(self):
        with self.lock:
            if self.finished:                 # a rank has already returned: this collective can never complete
                self.barrier.abort()
        self.barrier.wait(self.timeout)


class VirtualComm:
    """The exchange steps C1-C4 of an N-rank data-parallel job inside ONE process: each virtual rank runs the unchanged engine
    on its shard of the batch -- the same launches, tiles and split plans a real rank of that per-rank batch runs -- and an all-reduce is
    `deposit, barrier, rank-ordered sum ((r0 + r1) + r2) + ..., barrier`: the sum Comm(deterministic=True) forms across processes, and for two
    ranks the a + b of any all-reduce.  An N-process job is therefore reproduced BIT FOR BIT (tests/test_gpu_configs_at_size.py:
    test_data_parallel_equals_virtual_ranks_bit_for_bit); see run_virtual_ranks."""
    deterministic = True

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, int(rank), group.world
        self.on = self.world > 1
        self.trace = None

    def enable_trace(self, on=True):
        self.trace = None

    def trace_summary(self, steps):
        return {}

    def allreduce_(self, t, label=None):
        if not self.on:
            return t
        g = self.g
        # (ONE shared stream: host issue order is device order only on the default stream every thread starts with)
        assert not t.is_cuda or torch.cuda.current_stream(t.device) == torch.cuda.default_stream(t.device), "VirtualComm needs the default stream"
        g.slots[self.rank] = t
        g.wait()             # every rank's tensor is deposited (and its producers are enqueued on the shared stream)
        acc = g.slots[0].clone()
        for r in range(1, self.world):
            acc.add_(g.slots[r])
        g.wait()             # every rank has enqueued its reads of all slots: the in-place results may now be written
        t.copy_(acc)
        return t

    def allreduce_async_(self, t, label=None):
        if not self.on:
            return None
        self.allreduce_(t)
        return _DoneWork()

    def broadcast_(self, t, src=0):
        if not self.on:
            return t
        g = self.g
        g.slots[self.rank] = t
        g.wait()
        v = g.slots[src].clone()
        g.wait()
        t.copy_(v)
        return t


def run_virtual_ranks(world, fn):
    """fn(rank, comm) -> result in `world` threads, one VirtualComm each; returns the list of results.  An exception in one rank breaks the
    barrier, and so does a rank that returns while others still wait for (or later enter) a collective -- the job fails instead of hanging."""
    import threading
    group = VirtualGroup(world)
    out, err = [None] * world, []

    def body(r):
        try:
            out[r] = fn(r, VirtualComm(group, r))
            with group.lock:
                group.finished += 1
                # a rank still inside a collective waits for one this rank will never join (unequal collective counts): fail it now
                if group.finished < world and group.barrier.n_waiting:
                    group.barrier.abort()
        except BaseException as e:          # noqa: BLE001 -- reported below
            err.append((r, e))
            group.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), name="virtual-rank-%d" % r) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if torch.cuda.is_available():             # (the CPU unit test of the barrier logic runs without a device)
        torch.cuda.synchronize()
    if err:
        import threading as _t
        first = [e for e in err if not isinstance(e[1], _t.BrokenBarrierError)] or err
        raise RuntimeError("virtual rank %d failed: %r" % first[0]) from first[0][1]
    return out


class _NoComm:
    """Communicator stand-in for computations that are REPLICATED on every rank (the VampPrior pseudo-input pass)."""
    on, world, rank = False, 1, 0

    def allreduce_(self, t, label=None):
        return t

    def allreduce_async_(self, t, label=None):
        return None

    def broadcast_(self, t, src=0):
        return t


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
        self.fuse_fwd = 1    # projected pairs: forward GEMM + combination in one launch (config `fused_projected_forward`: 1 where measured faster, 2 wherever eligible, 0 off)
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

    def _backward_proj(self, dy, need_dx, wgrad, gate):
        """Backward of the project-then-upsample form from the low-resolution x: D [M, 9 cout] = (shift o up)^T dy once (elementwise), then
        dWcat = x^T D (+ the bias gradient as the centre plane's column sums) and dx_lo = D . wcatT -- two dense calls, exact on every pixel."""
        ctx, st = self.ctx, self.ctx.stream
        x = self.x
        N, H, W, _ = x.shape
        M, n9, f = N * H * W, 9 * self.cout, self.lo_factor
        flops = 2.0 * N * f * f * H * W * 9 * self.cin * self.cout
        executed = 2.0 * M * self.cin * n9
        d = ctx.empty(M, n9)
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

    def backward(self, dy, need_dx=True, wgrad=True, act_done=False, gate_prev=None, lowres_dx=False, lowres_gate=None):
        """`act_done`: dy already carries this layer's activation derivative (fused into the consumer's epilogue).
        `gate_prev`: activation name of the layer that produced this conv's input x: its derivative act'(x) is fused into
        the backward-data epilogue, so that layer must then be called with act_done=True.
        `lowres_dx`: x is the factor-2 upsample of a tensor the caller wants the gradient of: return d / d (that tensor) (see _dx_lowres)."""
        x, y = self.x, self.y
        N, H, W, _ = x.shape
        if self.x_is_lo:                                  # x is the low-resolution tensor: the layer's input is its (never materialised) upsample
            H, W = self.lo_factor * H, self.lo_factor * W
            if not lowres_dx and need_dx:
                raise RuntimeError("%s: only the low-resolution gradient exists for a virtual upsample" % self.name)
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


# ------------------------------------------------------------------------------------------ networks
class Encoder:
    def __init__(self, ctx, ps, cfg):
        self.ctx, self.cfg = ctx, cfg
        self.exp = cfg["exp_name"]
        self.convs, self.bns = [], []
        for i, (cin, cout, k, s, pad, act, bn) in enumerate(arch.encoder_convs(cfg)):
            self.convs.append(Conv2D(ctx, ps, "encoder/" + arch.tfname("conv2d", i), k, cin, cout, s, pad, act, bias_grad=not bn))
            self.bns.append(BatchNormAct(ctx, ps, "encoder/" + arch.tfname("batch_normalization", i), cout, "leaky_relu") if bn else None)
        feat, hid = arch.encoder_flat_dim(cfg), arch.encoder_hidden(cfg)
        self.hidden = Dense(ctx, ps, "encoder/dense", feat, hid, "leaky_relu") if hid is not None else None
        feat = hid if hid is not None else feat
        Z = int(cfg["code_size"])
        self.head_mu = Dense(ctx, ps, "encoder/code_mean", feat, Z, None)
        self.head_sd = Dense(ctx, ps, "encoder/code_std_dev", feat, Z, "relu")

    def forward(self, x):
        h = pad_symmetric(self.ctx, x, 2) if self.exp != "celeba" else x
        for i, (conv, bn) in enumerate(zip(self.convs, self.bns)):
            conv.want_bn_sums = bn is not None                  # a conv that can emit the statistics of its output does (conv.bn_sums)
            h = conv.forward(h)
            if bn is not None:
                nxt = self.convs[i + 1] if i + 1 < len(self.convs) else None
                h = bn.forward(h, sums=conv.bn_sums, planes=nxt.planes_demand(h.shape) if nxt is not None else (False, True))
            conv.bn_sums = None
        self.conv_shape = h.shape
        h = h.reshape(h.shape[0], -1)
        if self.hidden is not None:
            h = self.hidden.forward(h)
        return self.head_mu.forward(h), self.head_sd.forward(h)

    def backward(self, dmu, dsd_raw, wgrad=True, need_input_dx=False):
        """`need_input_dx`: also return d loss / d x (only the VampPrior pseudo-inputs are trainable inputs)."""
        dh = self.head_mu.backward(dmu, wgrad=wgrad)
        add_(self.ctx, dh, self.head_sd.backward(dsd_raw, wgrad=wgrad, act_done=True))
        if self.hidden is not None:
            dh = self.hidden.backward(dh, wgrad=wgrad)
        dh = dh.reshape(self.conv_shape)
        for i in range(len(self.convs) - 1, -1, -1):
            if self.bns[i] is not None:
                dh = self.bns[i].backward(dh, wgrad=wgrad)
            dh = self.convs[i].backward(dh, need_dx=(i > 0 or need_input_dx), wgrad=wgrad)
        if need_input_dx and self.exp != "celeba":                 # undo the SYMMETRIC 28 -> 32 pad
            N, Hp, Wp, C = dh.shape
            dx = self.ctx.empty(N, Hp - 4, Wp - 4, C)
            L.call("ladder_pad_symmetric_bwd", _p(dh), _p(dx), N, Hp - 4, Wp - 4, C, 2, self.ctx.stream)
            dh = dx
        return dh if need_input_dx else None


class MnistDecoder:
    def __init__(self, ctx, ps, cfg):
        self.ctx = ctx
        width, r0, convs = arch.mnist_decoder_convs(cfg)
        self.dense = Dense(ctx, ps, "decoder/dense", int(cfg["code_size"]), width, "leaky_relu")
        self.d2s0 = DepthToSpace(ctx, r0)
        self.convs, self.d2s = [], []
        for i, (k, ci, co, pad, act, d) in enumerate(convs):
            self.convs.append(Conv2D(ctx, ps, "decoder/" + arch.tfname("conv2d", i), k, ci, co, 1, pad, act))
            self.d2s.append(DepthToSpace(ctx, d) if d else None)

    def forward(self, z):
        h = self.dense.forward(z)
        h = self.d2s0.forward(h.view(h.shape[0], 1, 1, -1))
        for conv, d in zip(self.convs, self.d2s):
            h = conv.forward(h)
            if d is not None:
                h = d.forward(h)
        return h

    def backward(self, dxhat, need_dz=True):
        dh = dxhat
        for conv, d in zip(reversed(self.convs), reversed(self.d2s)):
            if d is not None:
                dh = d.backward(dh)
            dh = conv.backward(dh)
        dh = self.d2s0.backward(dh)
        return self.dense.backward(dh.reshape(dh.shape[0], -1), need_dx=need_dz)


class CelebADecoder:
    """codes/models.py:499-587: mapping MLP -> dlatent; 1x1 conv on `encoded`; 7 3x3 convs with
    IN+style+leaky on four of them and legacy-bilinear upsampling in between; 1x1 conv to RGB."""

    def __init__(self, ctx, ps, cfg):
        self.ctx = ctx
        nh, Z = int(cfg["num_hidden_units"]), int(cfg["code_size"])
        self.nh = nh
        self.dense0 = Dense(ctx, ps, "decoder/dense", Z, nh, "leaky_relu")
        self.mapping = [Dense(ctx, ps, "decoder/dense_%d" % i, nh, nh, "leaky_relu") for i in range(1, 9)]
        self.conv0 = Conv2D(ctx, ps, "decoder/conv2d", 1, nh, nh, 1, "same", None)
        self.up0 = Resize(ctx, 2, 2)
        self.blocks = []
        si = 0
        for i, (k, ci, co, styled, act, rs) in enumerate(arch.celeba_decoder_convs(cfg)):
            conv = Conv2D(ctx, ps, "decoder/conv2d_%d" % (i + 1), k, ci, co, 1, "same", act, bias_grad=not styled)
            sty = norm = None
            if styled:
                sty = Dense(ctx, ps, "decoder/StyleMod_%d/dense" % si, nh, 2 * co, None)
                norm = InstanceNormStyleAct(ctx, co, "leaky_relu")
                si += 1
            self.blocks.append((conv, sty, norm, Resize(ctx, rs, rs) if rs else None))
        self.conv_out = Conv2D(ctx, ps, "decoder/conv2d_8", 1, nh // 4, int(cfg["dim_input_channel"]), 1, "same", None)

    def forward(self, z):
        B = z.shape[0]
        encoded = self.dense0.forward(z)
        d = encoded
        for lyr in self.mapping:
            d = lyr.forward(d)
        dlatent = d
        h = self.conv0.forward(encoded.view(B, 1, 1, self.nh))
        lowres = 0                # != 0: h is the LOW-resolution input of a resize by this factor that the next conv applies itself (forward-only runs)
        lowres_copy = None        # training forward: the low-resolution tensor behind h = its upsample (kept for the backward pass) ...
        lo_f = 2                  # ... by this factor

        def _fac(rs_, t):         # integer factor (2 or 4) of resize `rs_` applied to t's map, 0 when it is neither
            for f_ in (2, 4):
                if (rs_.oh, rs_.ow) == (f_ * t.shape[1], f_ * t.shape[2]):
                    return f_
            return 0
        # the 1x1 -> 2x2 resize in front of conv2d_1 folds into it like every other one (projected form: all three passes from the 1x1 map)
        f0 = _fac(self.up0, h)
        self.up0_folded = bool(f0 and self.blocks[0][0].upf_ok(B, h.shape[1], h.shape[2], f0)
                               and (not self.ctx.keep_activations or self.blocks[0][0].virtual_upf_ok(B, h.shape[1], h.shape[2], f0)))
        if self.up0_folded and self.ctx.keep_activations:
            self.up0.in_shape = tuple(h.shape)
            lowres_copy, lo_f, h = h, f0, None
        elif self.up0_folded:
            lowres = f0
        else:
            h = self.up0.forward(h)
        for bi, (conv, sty, norm, rs) in enumerate(self.blocks):
            x_lo, lowres_copy = lowres_copy, None
            conv_done = False
            if lowres:
                f_in, lowres = lowres, 0
                last = (bi == len(self.blocks) - 1 and norm is None and f_in == 2
                        and (rs is None or (rs.oh, rs.ow) == (2 * h.shape[1], 2 * h.shape[2])))
                if last:
                    return conv.forward_up2(h, self.conv_out)
                h = conv.forward_up2(h, factor=f_in)
                conv_done = True
            elif bi == len(self.blocks) - 1 and norm is None and (rs is None or h is None or (rs.oh, rs.ow) == tuple(h.shape[1:3])):
                if x_lo is not None and lo_f == 2:
                    # training forward: h = the resized tensor (kept for the backward pass) -- or None when every consumer of it runs from the
                    # low-resolution tensor (virtual upsample); the convolution reads the low-resolution one
                    if rs is not None:
                        rs.in_shape = (x_lo.shape[0], 2 * x_lo.shape[1], 2 * x_lo.shape[2], conv.cout)
                    return conv.forward_up2(x_lo, self.conv_out, keep_y=True, x_for_backward=h)
                # the last 3x3 conv feeds the 1x1 output conv directly (its resize is the identity): one fused launch
                out = conv.forward_fused_proj(h, self.conv_out, keep_y=self.ctx.keep_activations)
                if out is not None:
                    if rs is not None:
                        rs.in_shape = tuple(h.shape[:3]) + (conv.cout,)
                    return out
            if not conv_done and x_lo is not None:
                # training forward of an inner layer (conv2d_6): h = the resized tensor, kept for this layer's filter gradient / backward-data;
                # the convolution itself reads the low-resolution tensor (25 of 36 tap products)
                h = conv.forward_up2(x_lo, keep_y=True, x_for_backward=h, factor=lo_f)
                conv_done = True
            if not conv_done:
                h = conv.forward(h)
            # the resize behind this block folds into the NEXT conv when that one can take the low-resolution tensor (forward-only runs)
            nxt = self.blocks[bi + 1][0] if bi + 1 < len(self.blocks) else None
            f_rs = _fac(rs, h) if (rs is not None and nxt is not None) else 0
            fold = bool(f_rs and not self.ctx.keep_activations and nxt.upf_ok(h.shape[0], h.shape[1], h.shape[2], f_rs))
            if norm is not None:
                style = sty.forward(dlatent)
                # (training forward: the NEXT conv reads the low-resolution tensor, written beside the resized one it keeps for backward)
                want_lo = bool(self.ctx.up2 >= 2 and f_rs and not fold and self.ctx.keep_activations
                               and nxt.upf_ok(h.shape[0], h.shape[1], h.shape[2], f_rs))
                if want_lo and nxt.virtual_upf_ok(h.shape[0], h.shape[1], h.shape[2], f_rs):
                    # the resized tensor has no reader left (the next layer's forward, backward-data and filter gradient all take the
                    # low-resolution tensor): plain instance norm, no resize, 1/4 of the bytes
                    rs.in_shape = tuple(h.shape)
                    lowres_copy, lo_f = norm.forward(h, style), f_rs
                    h = None
                    continue
                up = norm.forward_resized(h, style, rs, keep_lowres=want_lo and f_rs == 2) if (rs is not None and not fold) else None
                if up is not None:
                    h = up
                    lowres_copy, lo_f = (norm.y_lo if want_lo else None), 2
                    continue
                h = norm.forward(h, style)
            if fold:
                lowres = f_rs
                continue
            if rs is not None:
                # (training forward, un-normalised layer in front of a factor-2 resize -- conv2d_5: its output IS the low-resolution tensor)
                keep_lo = bool(norm is None and self.ctx.up2 >= 2 and self.ctx.keep_activations and f_rs
                               and nxt.upf_ok(h.shape[0], h.shape[1], h.shape[2], f_rs)
                               and (f_rs == 2 or nxt.virtual_upf_ok(h.shape[0], h.shape[1], h.shape[2], f_rs)))
                lo = h
                if keep_lo and nxt.virtual_upf_ok(h.shape[0], h.shape[1], h.shape[2], f_rs):
                    rs.in_shape = tuple(h.shape)                 # (virtual upsample: see above)
                    h = None
                else:
                    h = rs.forward(h)
                lowres_copy, lo_f = (lo if keep_lo else None), (f_rs or 2)
        return self.conv_out.forward(h)

    def backward(self, dxhat, need_dz=True):
        ctx = self.ctx
        # the last 3x3 conv feeds conv_out directly (its resize is the identity at full resolution): its leaky-ReLU
        # backward is fused into conv_out's backward-data epilogue (saves a read+write pass over the largest map)
        last_conv, _, last_norm, last_rs = self.blocks[-1]
        fuse_last = (last_norm is None and last_conv.act is not None
                     and (last_rs is None or tuple(last_rs.in_shape[1:3]) == (last_rs.oh, last_rs.ow)))
        dh = self.conv_out.backward(dxhat, gate_prev=last_conv.act if fuse_last else None)
        ddlat = None
        lowres = False            # dh is already the gradient of the LOW-resolution tensor behind the next resize (fused into the conv's backward-data)
        pre_gated = False         # ... and already carries this block's activation derivative (gated low-resolution backward-data of the block above)
        for bi, (conv, sty, norm, rs) in enumerate(reversed(self.blocks)):
            gated = False
            if lowres:
                lowres = False                                   # (this block's resize transpose is done)
            elif rs is not None:
                if norm is None and conv.act is not None and not (bi == 0 and fuse_last):
                    dh, gated = rs.backward(dh, gate=(conv.y, conv.act))   # leaky conv -> resize: its activation backward rides on the transpose
                else:
                    dh = rs.backward(dh)
            if norm is not None:
                dh, dstyle = norm.backward(dh)
                g = sty.backward(dstyle)
                ddlat = g if ddlat is None else add_(ctx, ddlat, g)
            # the block below ends in a factor-2 resize: this conv's backward-data can return the gradient of the resize's INPUT (conv2d_7, conv2d_6)
            below = self.blocks[len(self.blocks) - 2 - bi] if bi + 1 < len(self.blocks) else None
            lowres = bool(conv.x_is_lo or (
                below is not None and below[3] is not None and conv.x is not None and (bi == 0 or self.ctx.up2 >= 3)
                and (below[3].oh, below[3].ow) == tuple(conv.x.shape[1:3]) and conv.x.shape[1] % 2 == 0
                and tuple(getattr(below[3], "in_shape", (0, 0, 0))[1:3]) == (conv.x.shape[1] // 2, conv.x.shape[2] // 2)
                and conv.up2t_ok(conv.x.shape[0], conv.x.shape[1] // 2, conv.x.shape[2] // 2)))
            # ... and that block's own activation backward rides on it when it is an un-normalised leaky conv (conv2d_5 under conv2d_6)
            lgate = None
            if (lowres and below is not None and below[2] is None and below[0].act is not None and below[0].y is not None and conv.x is not None
                    and conv.lowres_gate_ok(conv.x.shape[0], below[0].y.shape[1], below[0].y.shape[2])):
                lgate = (below[0].y, below[0].act)
            dh = conv.backward(dh, act_done=(bi == 0 and fuse_last) or gated or pre_gated, lowres_dx=lowres, lowres_gate=lgate)
            pre_gated = lgate is not None                      # (the NEXT block's activation backward is done)
        dh = self.conv0.backward(dh if lowres else self.up0.backward(dh))     # (lowres: conv2d_1 returned the gradient of the 1x1 map itself)
        denc = dh.reshape(dh.shape[0], self.nh)
        # mapping MLP: each layer's backward-data epilogue applies the previous layer's leaky-ReLU derivative
        for i in range(len(self.mapping) - 1, -1, -1):
            ddlat = self.mapping[i].backward(ddlat, act_done=(i < len(self.mapping) - 1), gate_prev="leaky_relu" if i > 0 else None)
        add_(ctx, denc, ddlat)
        return self.dense0.backward(denc, need_dx=need_dz)


class InnerVAE:
    """codes/base.py:127-213."""

    def __init__(self, ctx, ps, cfg):
        self.ctx = ctx
        Z, H = int(cfg["code_size"]), int(cfg["num_hidden_units_inner_VAE"])
        R, nl = int(cfg["representation_size"]), int(cfg["n_layers_inner_VAE"])
        a = cfg["inner_activation"]
        dims = [(Z, H)] + [(H, H)] * (nl - 1)
        self.enc = [Dense(ctx, ps, "prior/" + arch.tfname("dense", i), ci, co, a) for i, (ci, co) in enumerate(dims)]
        self.head_mu = Dense(ctx, ps, "prior/" + arch.tfname("dense", nl), H, R, None)
        self.head_sd = Dense(ctx, ps, "prior/" + arch.tfname("dense", nl + 1), H, R, "relu")
        dims = [(R, H)] + [(H, H)] * (nl - 1)
        self.dec = [Dense(ctx, ps, "prior/" + arch.tfname("dense", nl + 2 + i), ci, co, a) for i, (ci, co) in enumerate(dims)]
        self.dec_out = Dense(ctx, ps, "prior/" + arch.tfname("dense", 2 * nl + 2), H, Z, None)

    def encode(self, z):
        h = z
        for lyr in self.enc:
            h = lyr.forward(h)
        return self.head_mu.forward(h), self.head_sd.forward(h)

    def decode(self, t):
        h = t
        for lyr in self.dec:
            h = lyr.forward(h)
        return self.dec_out.forward(h)

    def decode_backward(self, dzhat, wgrad):
        # every backward-data epilogue applies the activation derivative of the layer below it (its input)
        a = self.dec[0].act
        dh = self.dec_out.backward(dzhat, wgrad=wgrad, gate_prev=a)
        for i in range(len(self.dec) - 1, -1, -1):
            dh = self.dec[i].backward(dh, wgrad=wgrad, act_done=True, gate_prev=a if i > 0 else None)
        return dh

    def encode_backward(self, dmu, dsd_raw, wgrad, need_dz):
        dh = self.head_mu.backward(dmu, wgrad=wgrad)
        add_(self.ctx, dh, self.head_sd.backward(dsd_raw, wgrad=wgrad, act_done=True))
        a = self.enc[0].act
        for i in range(len(self.enc) - 1, -1, -1):
            # the head gradients were summed with a separate add, so the top hidden layer applies its own activation
            # derivative; below it each backward-data epilogue carries the derivative of the layer underneath
            dh = self.enc[i].backward(dh, need_dx=(need_dz or i > 0), wgrad=wgrad, act_done=(i < len(self.enc) - 1),
                                      gate_prev=a if i > 0 else None)
        return dh


class _AsyncFetch:
    """Handle of LadderEngine.fetch_async()."""

    def __init__(self, host, event, names, pool, undefined=()):
        self._host, self._event, self._names, self._pool, self._val = host, event, names, pool, None
        self._undefined = frozenset(undefined)

    def ready(self):
        """True when get() would not block."""
        return self._val is not None or self._event.query()

    def get(self):
        if self._val is None:
            self._event.synchronize()
            s = self._host.numpy()
            self._val = {n: (float("nan") if n in self._undefined else float(s[L.S_INDEX[n]])) for n in self._names}
            self._pool.append(self._host)          # the pinned buffer goes back to the engine's pool
            self._host = None
        return self._val


# scalars that exist only when the decoder ran (RUN#3 / RUN#4, sample_code, sample_representation evaluate encoder + inner VAE only)
DEC_SCALARS = frozenset(("sigma", "mean_pixel_error", "l1_reconstruction_error", "l2_reconstruction_error", "reconstruction_likelihood",
                         "sigma_regularisor", "elbo", "loss_ae"))


# ------------------------------------------------------------------------------------------ engine
class LadderEngine:
    """Owns parameters + optimiser state and evaluates the reference's four per-minibatch runs.

    run_ae          = RUN#1  (codes/base.py:587-594)   full fwd+bwd, clip+Adam on encoder+decoder
    run_sigma       = RUN#2  (601-606)                 encoder+decoder fwd, scalar Adam on sigma/Variable
    run_prior       = RUN#3  (615-622)                 encoder fwd, inner-VAE fwd+bwd, Adam on prior/*
    run_inner_sigma = RUN#4  (636-639)                 encoder + inner-VAE fwd, scalar Adam on inner_sigma
    evaluate        = val_step / test_step fetches     (643-679, 944-986)
    """

    def __init__(self, cfg, device="cuda:0", values=None, seed=1, comm=None, noise_seed=1234):
        self.cfg = cfg
        if comm is None and bool(int(cfg.get("deterministic_allreduce", 0))):
            comm = Comm(deterministic=True)
        self.ctx = Ctx(device, comm)
        prec = str(cfg.get("matmul_precision", DEFAULT_PRECISION))
        if prec not in PRECISIONS:
            raise ValueError("matmul_precision %r: expected one of %s" % (prec, sorted(PRECISIONS)))
        self.ctx.ns = PRECISIONS[prec]
        # 0: off, 1: forward-only runs, 2: also the training forward and the backward-data of the last 3x3 conv, 3: also conv2d_6's backward-data (no gain measured)
        # (strict fp32 default 3: with the fp32 MFMA the 11 / 36 of conv2d_6's backward-data outweigh its border strips, +0.5 %; f16x3: no gain, 2)
        self.ctx.up2 = int(cfg.get("upsample_fused_convs", 4 if prec == "f32" else 2))
        self.ctx.fuse_fwd = int(cfg.get("fused_projected_forward", 1))
        self.precision = prec
        if self.ctx.comm.rank == 0:
            print("Contraction precision (config key matmul_precision): {} -- {}".format(prec, PRECISION_NOTES[prec]))
        self.ps = ParamStore(cfg, self.ctx, values, seed)
        self.encoder = Encoder(self.ctx, self.ps, cfg)
        self.decoder = (CelebADecoder if cfg["exp_name"] == "celeba" else MnistDecoder)(self.ctx, self.ps, cfg)
        self.has_inner = cfg["prior"] in ("ours", "hierarchical")
        self.hier = cfg["prior"] == "hierarchical"          # inner VAE against N(0,I): no mixture term, no mask (base.py:331-359)
        self.gmm_z = cfg["prior"] == "GMM"                   # mixture directly on z (R = code_size), no inner VAE (base.py:322-329)
        self.vamp = cfg["prior"] == "vampPrior"              # diagonal mixture on z from trainable pseudo-inputs (base.py:216-254)
        if cfg["prior"] not in ("ours", "hierarchical", "GMM", "vampPrior", "standard_gaussian"):
            raise ValueError("unknown prior %r" % cfg["prior"])
        if self.vamp:
            # second pass of the SAME encoder weights over the K pseudo-inputs (tf.variable_scope('encoder', reuse=True)); the
            # pseudo-inputs are replicated on every rank, so its batch-norm statistics are not exchanged
            self.encoder_p = Encoder(self.ctx.local(), self.ps, cfg)
            self._enc_range = self.ps.prefix_range("ae", "encoder/")
        self.inner = InnerVAE(self.ctx, self.ps, cfg) if self.has_inner else None
        self.Z = int(cfg["code_size"])
        self.R = int(cfg["code_size"]) if cfg["prior"] in ("GMM", "vampPrior") else int(cfg.get("representation_size", 1))   # mixture dimension
        self.K = int(cfg.get("n_mixtures", 1))
        self.Lmc = int(cfg.get("n_MC_samples", 1))
        self.D = int(cfg["dim_input_x"]) * int(cfg["dim_input_y"]) * int(cfg["dim_input_channel"])
        self.lvp = float(cfg["latent_variance_precision"])
        self.partials = self.ctx.zeros(L.P_FIXED + self.Z + self.R)
        self.scalars = self.ctx.zeros(L.S_COUNT)
        self.noise_seed = int(noise_seed) + 7919 * self.ctx.comm.rank
        self.rng_counter = torch.zeros(1, dtype=torch.int64, device=self.ctx.device)   # Philox stream position (device)
        self._run_calls = 0
        self._gm_packed = None
        self.use_graphs = False
        # filter gradients on a second stream beside the backward chain (config key `overlap_filter_gradients` / environment variable
        # LADDER_OVERLAP_FILTER_GRADIENTS): +0.7-1 % per iteration with the f16x3 kernels, nothing with the fp32 ones (round 4: 3 328 / 3 327 img/s --
        # both partners are bound by the fp32 matrix pipe), bit-identical results -- OFF by default because kernels that share the
        # chip stretch each other, so the per-kernel HIP-event durations bench.py reports (roofline.achieved) stop describing the kernels
        if bool(int(cfg.get("overlap_filter_gradients", os.environ.get("LADDER_OVERLAP_FILTER_GRADIENTS", 0)))):
            self.ctx.side = torch.cuda.Stream(device=self.ctx.device)
        self._graphs, self._warm = {}, {}
        self._dec_range = self.ps.prefix_range("ae", "decoder/")   # C1 bucket boundary (data parallel)
        # RUN#3 / RUN#4 on their own HIP stream beside RUN#2's decoder forward (enable_prior_overlap; the trainer turns it on)
        self.ps.before_read = self._join_aux
        self._aux_on = False
        self._enc_event, self._aux_event = None, None              # "z of the last main-stream forward exists" / "the aux runs are done"
        self._main_calls, self._aux_calls = 0, 0                    # noise calls of the last main forward / of the aux runs since
        self._on_aux = False
        self._fetch_src = None                                      # (scalars buffer, stream) of the last run
        self._sigma_one = torch.ones(1, dtype=torch.float32, device=self.ctx.device)   # stand-in for sigma/Variable in decoder-less runs

    def enable_prior_overlap(self, on=True):
        """RUN#3 and RUN#4 (inner VAE forward / backward, mixture term, Adam on prior/* and inner_sigma: ~150 launches of a few
        microseconds, 0.9 ms of latency-bound chain on CelebA) read only the encoder output that RUN#2 computes FIRST and write only the
        prior variables, which RUN#2 never touches: with this on, run_prior / run_inner_sigma called with reuse_encoder=True are
        enqueued on a second HIP stream behind an event recorded right after RUN#2's code sample, and run beside RUN#2's decoder
        forward (big MFMA-bound kernels).  Separate partials / scalars buffers and scratch; the noise positions are those of the
        sequential order (snapshot of the device counter + host-known call counts), so results are bit-identical to the sequential
        schedule.  The next main-stream run waits for the aux stream first.  Off under data parallelism (collectives stay on one
        stream) and with hipGraph replay."""
        on = bool(on) and not self.ctx.comm.on
        self._aux_on = on
        if on and self.ctx.aux is None:
            # (high priority: its few-microsecond kernels should not queue behind the thousands of workgroups of a decoder conv)
            self.ctx.aux = torch.cuda.Stream(device=self.ctx.device, priority=int(os.environ.get("LADDER_AUX_PRIORITY", -1)))
            self.partials_aux, self.scalars_aux = torch.zeros_like(self.partials), torch.zeros_like(self.scalars)
            self.rng_counter_aux = torch.zeros_like(self.rng_counter)

    def _join_aux(self):
        """The main stream waits for the runs enqueued on the aux stream (no-op when none are pending)."""
        if self._aux_event is not None:
            torch.cuda.current_stream(self.ctx.device).wait_event(self._aux_event)
            self._aux_event = None
            self._aux_calls = 0

    # -- inputs ---------------------------------------------------------------------------------
    def _dev(self, a):
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            return a.to(device=self.ctx.device, dtype=torch.float32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.ctx.device)

    @staticmethod
    def _batch_token(x):
        """(address, shape) of a minibatch as the caller holds it: the key of the encoder-output cache."""
        if isinstance(x, torch.Tensor):
            return ("t", x.data_ptr(), tuple(x.shape))
        a = np.asarray(x)
        return ("n", a.__array_interface__["data"][0], tuple(a.shape))

    def _randn(self, *shape):
        t = self.ctx.empty(*shape)
        if self._on_aux:      # sequential position: the counter as RUN#2 saw it + RUN#2's calls + the earlier aux runs' calls + this run's
            L.call("ladder_randn_dev", _p(t), t.numel(), self.noise_seed, _p(self.rng_counter_aux),
                   self._main_calls + self._aux_calls + self._run_calls, self.ctx.stream)
        else:
            L.call("ladder_randn_dev", _p(t), t.numel(), self.noise_seed, _p(self.rng_counter), self._run_calls, self.ctx.stream)
        self._run_calls += 1
        return t

    def _noise(self, noise, key, shape):
        if noise is not None and noise.get(key) is not None:
            t = self._dev(noise[key])
            assert tuple(t.shape) == tuple(shape), (key, t.shape, shape)
            return t
        return self._randn(*shape)

    def set_mixture(self, weights, means, covs):
        """Feed of prior_weight / prior_mean / prior_cov (codes/base.py:110-112, 870-888); fp32 like the placeholders."""
        K, R = self.K, self.R
        w, m, c = self._dev(weights), self._dev(means), self._dev(covs)
        assert tuple(w.shape) == (K,) and tuple(m.shape) == (K, R) and tuple(c.shape) == (K, R, R)
        self._gm_dense = R > 8                               # wide latent (prior "GMM"): whitening as a GEMM on the dense kernel
        if getattr(self, "_gm_buf", None) is None:           # persistent: captured graphs keep pointing at the current mixture
            self._gm_buf = self.ctx.empty(L.query("ladder_gmm_dense_param_floats", K, R) if self._gm_dense
                                          else K * L.query("ladder_gmm_packed_stride", R))
        L.call("ladder_gmm_prepare_dense" if self._gm_dense else "ladder_gmm_prepare", _p(w), _p(m), _p(c), K, R, _p(self._gm_buf),
               self.ctx.stream)
        torch.cuda.current_stream(self.ctx.device).synchronize()   # w, m, c are temporaries: keep them alive until the kernel ran
        self._gm_packed = self._gm_buf

    def set_sg_mixture(self):
        """The dummy N(0,I) mixture of the SG-pretraining feed (codes/base.py:870-876)."""
        K, R = self.K, self.R
        self.set_mixture(np.full(K, 1.0 / K), np.zeros((K, R)), np.tile(np.eye(R), (K, 1, 1)))

    # -- forward --------------------------------------------------------------------------------
    def forward(self, x, noise=None, use_sg=True, use_mask=False, parts=("dec", "inner", "gmm"), reuse_encoder=False, keep_acts=True):
        """`reuse_encoder`: the caller asserts that this run evaluates the SAME minibatch as the previous run and that no
        encoder variable changed in between (RUN#2/#3/#4 after RUN#1: only sigma / prior variables are updated,
        codes/base.py:601-639).  The encoder output (code_mean, code_std_dev) of the previous run is then bit-identical to
        what a re-evaluation would give (deterministic kernels, batch statistics of the same batch) and is reused; the
        fresh noise of the run still produces a new code_sample.  Guarded by the AE optimiser step counter."""
        ctx, st = self.ctx, self.ctx.stream
        if not self._on_aux:
            self._join_aux()                       # (prior variables / noise position: the aux runs of the last iteration come first)
            # whoever calls forward() on the main stream -- a run, evaluate(), sample_*, the Session facade -- fetches THIS forward's scalars
            # (evaluate() after an overlapped RUN#3 / RUN#4 used to read the aux run's stale buffer)
            self._fetch_src = (self.scalars, torch.cuda.current_stream(ctx.device))
        ctx.keep_activations = bool(keep_acts)     # forward-only runs (RUN#2, val_step): fused kernels skip backward-only tensors
        Z, R = self.Z, self.R
        P = self.partials
        L.call("ladder_axpy", None, _p(P), P.numel(), 0.0, 2, st)
        self._run_calls = 0
        use_mask = bool(use_mask) and not self.hier
        # identity of the minibatch the caller handed in (set by _run; direct callers of forward() are identified by their tensor)
        tok = self.__dict__.pop("_x_token", None) or self._batch_token(x)
        cache = getattr(self, "_enc_cache", None)
        if (reuse_encoder and cache is not None and cache[0] == self.ps.step["ae"] and cache[4] == tok
                and not self._enc_needs_grad(parts)):
            # the cached codes belong to ONE minibatch: a caller handing in another tensor (or shape) gets a fresh encoder pass, never
            # stale codes (an in-place refill of the same buffer must go through a run without reuse_encoder first)
            _, x, mu, sd_raw, _ = cache
        else:
            x = self._dev(x)
            mu, sd_raw = self.encoder.forward(x)
            self._enc_cache = (self.ps.step["ae"], x, mu, sd_raw, tok)
        B = x.shape[0]
        self.x, self.B = x, B
        self.Bg = B * ctx.comm.world
        eps_z = self._noise(noise, "eps_z", (B, Z))
        z, sd = ctx.empty(B, Z), ctx.empty(B, Z)
        L.call("ladder_latent_fwd", _p(mu), _p(sd_raw), _p(eps_z), self.lvp, _p(z), _p(sd), _p(P[L.P_LOG_SDZ:]),
               _p(P[L.P_MU2SD2_Z:]), _p(P[L.P_FIXED:]), B, Z, st)
        self.lat_z = (mu, sd, sd_raw, eps_z, z)
        # everything a following aux run reads from this forward exists now (the encoder output).  If this forward draws no further noise
        # (RUN#2: decoder only) the aux stream may start here, beside the decoder; else only after the last draw (end of the forward)
        early_aux = "inner" not in parts and "gmm" not in parts
        if early_aux:
            self._mark_for_aux()
        self.xhat = None
        if "dec" in parts:
            xhat = self.decoder.forward(z)
            nb = L.query("ladder_pixel_partials_workspace_bytes", x.numel())
            wsp, wsn = ctx.ws(nb)
            L.call("ladder_pixel_partials", _p(x), _p(xhat), x.numel(), _p(P[L.P_PIX_ABS:]), wsp, wsn, st)
            self.xhat = xhat
        if self.gmm_z and "gmm" in parts:
            self.gmm_grads = self._mixture_term(mu, sd, noise, B, need_grad="gmm_grad" in parts)
        vamp_on = self.vamp and "gmm" in parts and not use_sg
        if vamp_on:
            self.gmm_grads = self._vamp_term(mu, sd, noise, B)
        inner_on = self.has_inner and "inner" in parts
        if inner_on:
            mu_t, sdraw_t = self.inner.encode(z)
            eps_t = self._noise(noise, "eps_t", (B, R))
            t, sd_t = ctx.empty(B, R), ctx.empty(B, R)
            L.call("ladder_latent_fwd", _p(mu_t), _p(sdraw_t), _p(eps_t), self.lvp, _p(t), _p(sd_t), _p(P[L.P_LOG_SDT:]),
                   _p(P[L.P_MU2SD2_T:]), _p(P[L.P_FIXED + Z:]), B, R, st)
            zhat = self.inner.decode(t)
            L.call("ladder_code_partials", _p(z), _p(zhat), _p(sd), int(use_mask), _p(P[L.P_CODE_ERR:]), B, Z, st)
            self.lat_t = (mu_t, sd_t, sdraw_t, eps_t, t)
            self.zhat = zhat
            self.gmm_grads = None
            if "gmm" in parts and not self.hier:
                if self._gm_packed is None:
                    raise L.LadderHipError("set_mixture()/set_sg_mixture() must be called before a run that evaluates the GM prior")
                self.gmm_grads = self._mixture_term(mu_t, sd_t, noise, B)
        ctx.comm.allreduce_(P, "C3 partials")                    # C3: scalar partials of the GLOBAL batch
        ecfg = L.LadderElboCfg(self.Bg, self.D, Z, R, self.Lmc,
                               1 if (self.cfg["exp_name"] == "celeba" or int(self.cfg["TRAIN_sigma"]) == 1) else 0,
                               1 if inner_on else 0, 1 if use_sg else 0,
                               1 if (self.has_inner and int(self.cfg["TRAIN_inner_sigma"]) == 1) else 0,
                               float(self.cfg.get("inner_sigma_lb", 0.0)), float(self.cfg.get("inner_sigma_ub", 0.0)),
                               1 if self.hier else 0, 1 if (self.gmm_z or vamp_on) else 0)
        # A run without the decoder (RUN#3 / RUN#4, the t-sample passes of the mixture fit) has no pixel term: its sigma-dependent scalars
        # (sigma, reconstruction likelihood, elbo, loss_ae) are not part of what the reference fetches from it (base.py:615-639).  They are
        # evaluated against a CONSTANT 1 instead of sigma/Variable, which RUN#2's optimiser step may be writing on the main stream while
        # these runs execute on the aux stream (cross-stream race; the values were timing-dependent).  inner_sigma/Variable is
        # only read by runs that evaluate the inner VAE, all of which are ordered with its writer (RUN#4, same stream / joined).
        sig = self.ps.w["sigma/Variable"] if "dec" in parts else self._sigma_one
        # (the decoder-dependent scalars of such a run are computed against that stand-in and mean nothing: fetch() reports them as NaN)
        self._undefined = () if "dec" in parts else DEC_SCALARS
        L.call("ladder_elbo_finalize", _p(P), _p(sig),
               _p(self.ps.w["inner_sigma/Variable"]) if self.has_inner else None, ecfg, _p(self.scalars), st)
        if not early_aux:
            self._mark_for_aux()
        if self._run_calls:
            L.call("ladder_u64_add", _p(self.rng_counter), self._run_calls, st)      # advance the device noise stream (atomic add)
        if self._on_aux:
            self._aux_calls += self._run_calls
        else:
            self._main_calls = self._run_calls
        self.use_sg, self.use_mask = use_sg, use_mask

    def _mark_for_aux(self):
        """Main-stream forward: snapshot of the noise position (before this run's advance) + the event an aux run waits for."""
        if self._aux_on and not self._on_aux and not torch.cuda.is_current_stream_capturing():
            self.rng_counter_aux.copy_(self.rng_counter, non_blocking=True)
            self._enc_event = torch.cuda.Event()
            self._enc_event.record(torch.cuda.current_stream(self.ctx.device))

    def _mixture_term(self, mu, sd, noise, B, need_grad=True):
        """MC estimate of E_q[log p_GM] over L samples of N(mu, sd^2) (base.py:308-313 on t, 322-329 on z): writes the sum of
        the log-probs into the partials and returns (sum_l dlogp/dt, sum_l dlogp/dt * eps) for the latent backward."""
        ctx, st, R, P = self.ctx, self.ctx.stream, self.R, self.partials
        if self._gm_packed is None:
            raise L.LadderHipError("set_mixture()/set_sg_mixture() must be called before a run that evaluates the GM prior")
        eps_mc = self._noise(noise, "eps_mc", (self.Lmc, B, R))
        dmu, dsd = (ctx.empty(B, R), ctx.empty(B, R)) if need_grad else (None, None)
        if self._gm_dense:
            wsp, wsn = ctx.ws(L.query("ladder_gmm_dense_workspace_bytes", self.Lmc, B, R, self.K))
            L.call("ladder_gmm_dense_logprob_fwd_bwd", _p(mu), _p(sd), _p(eps_mc), _p(self._gm_packed), self.Lmc, B, R, self.K,
                   _p(P[L.P_LOGP:]), _p(dmu), _p(dsd), wsp, wsn, st)
        else:
            if dmu is None:
                dmu, dsd = ctx.empty(B, R), ctx.empty(B, R)
            wsp, wsn = ctx.ws(L.query("ladder_gmm_workspace_bytes", self.Lmc, B))
            # (flop count of the profiler entry: two Mahalanobis passes + the gradient accumulation per component evaluation)
            _timed(7700, float(self.Lmc) * B * self.K * (3.0 * R * (R + 1) + 4.0 * R + 8.0), "ladder_gmm_logprob_fwd_bwd",
                   (_p(mu), _p(sd), _p(eps_mc), _p(self._gm_packed), self.Lmc, B, R, self.K, _p(P[L.P_LOGP:]), _p(dmu), _p(dsd), wsp, wsn, st))
        return dmu, dsd

    def _vamp_term(self, mu, sd, noise, B):
        """crossEntropy_prior of the VampPrior (base.py:361-370): encoder pass over the pseudo-inputs -> K diagonal components,
        then the MC mixture term and its gradients towards both the posterior heads and the components."""
        ctx, st, Z, K, P = self.ctx, self.ctx.stream, self.Z, self.K, self.partials
        mu_p, sdraw_p = self.encoder_p.forward(self.ps.w["prior/Variable"])
        eps0, sd_p, scratch = ctx.zeros(K, Z), ctx.empty(K, Z), ctx.zeros(4)
        L.call("ladder_latent_fwd", _p(mu_p), _p(sdraw_p), _p(eps0), self.lvp, None, _p(sd_p), _p(scratch), _p(scratch[1:]), None, K, Z, st)
        eps_mc = self._noise(noise, "eps_mc", (self.Lmc, B, Z))
        dmu, dsd, dcm, dcs = ctx.empty(B, Z), ctx.empty(B, Z), ctx.empty(K, Z), ctx.empty(K, Z)
        wsp, wsn = ctx.ws(L.query("ladder_diag_mixture_workspace_bytes", B, Z, K))
        L.call("ladder_diag_mixture_fwd_bwd", _p(mu), _p(sd), _p(eps_mc), _p(mu_p), _p(sd_p), self.Lmc, B, Z, K, _p(P[L.P_LOGP:]),
               _p(dmu), _p(dsd), _p(dcm), _p(dcs), wsp, wsn, st)
        self.vamp_state = (mu_p, sd_p, sdraw_p, eps0, dcm, dcs)
        return dmu, dsd

    def _vamp_backward(self, wgrad, need_input_dx):
        """Backward of the pseudo-input pass: d loss/d components = -(1/LB) * (dcomp_mean, dcomp_sd) through the heads."""
        ctx, st, Z, K = self.ctx, self.ctx.stream, self.Z, self.K
        mu_p, sd_p, sdraw_p, eps0, dcm, dcs = self.vamp_state
        dmu_h, dsdraw_h = ctx.empty(K, Z), ctx.empty(K, Z)
        L.call("ladder_latent_bwd", None, _p(mu_p), _p(sd_p), _p(sdraw_p), _p(eps0), _p(dcm), _p(dcs), -1.0, _p(self.scalars), 0,
               _p(dmu_h), _p(dsdraw_h), K, Z, st)
        return self.encoder_p.backward(dmu_h, dsdraw_h, wgrad=wgrad, need_input_dx=need_input_dx)

    @staticmethod
    def _enc_needs_grad(parts):
        return False          # only run_ae back-propagates into the encoder, and it never asks for reuse

    def fetch(self, names=None):
        """Host copy of the fetched scalars (ONE device->host sync)."""
        self._join_aux()
        src = self._fetch_src[0] if self._fetch_src is not None else self.scalars
        s = src.detach().cpu().numpy()
        und = getattr(self, "_undefined", ())
        # (default list: the scalars the last run DEFINES -- a decoder-less run has no sigma / reconstruction / elbo; asked for by name they are NaN)
        names = names or [n for n in L.S_NAMES if not n.startswith("_") and n not in und]
        return {n: (float("nan") if n in und else float(s[L.S_INDEX[n]])) for n in names}

    def fetch_async(self, names=None):
        """fetch() without stalling the host: the scalars are copied into pinned host memory behind the kernels enqueued so far and an
        event marks the copy; `.get()` on the returned handle waits for THAT event only.  A caller that enqueues the next run before it
        reads the values keeps the GPU busy across the read (a synchronous fetch after every run left ~0.35 ms of bubbles per CelebA
        iteration: the device drains, then waits for the host to enqueue the next run's first kernels)."""
        pool = self.__dict__.setdefault("_pinned_pool", [])
        host = pool.pop() if pool else torch.empty(L.S_COUNT, dtype=torch.float32, pin_memory=True)
        src, stream = self._fetch_src if self._fetch_src is not None else (self.scalars, torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(stream):                  # (the stream the run was enqueued on: its scalars buffer, its order)
            host.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        und = getattr(self, "_undefined", ())
        return _AsyncFetch(host, ev, names or [n for n in L.S_NAMES if not n.startswith("_") and n not in und], pool, und)

    def std_dev_code(self):
        return (self.partials[L.P_FIXED:L.P_FIXED + self.Z] / self.Bg).cpu().numpy()

    def std_dev_representation(self):
        return (self.partials[L.P_FIXED + self.Z:L.P_FIXED + self.Z + self.R] / self.Bg).cpu().numpy()

    # -- backward pieces ------------------------------------------------------------------------
    def _sc(self, name):
        return self.scalars[L.S_INDEX[name]:]

    def _backward_ae(self):
        ctx, st, B, Z, R = self.ctx, self.ctx.stream, self.B, self.Z, self.R
        mu, sd, sd_raw, eps_z, z = self.lat_z
        vamp_on = self.vamp and not self.use_sg
        saved_enc = None
        if vamp_on:
            # the encoder weights receive two gradient contributions (data pass + pseudo-input pass); the filter-gradient kernels
            # overwrite, so the pseudo-pass result is parked and added back after the data pass
            self._vamp_backward(wgrad=True, need_input_dx=False)
            ctx.join_side()
            lo, hi = self._enc_range
            saved_enc = self.ps.grad["ae"][lo:hi].clone()
        dxhat = torch.empty_like(self.xhat)
        L.call("ladder_pixel_grad", _p(self.x), _p(self.xhat), _p(self._sc("_g_pix")), _p(dxhat), dxhat.numel(), st)
        dz = self.decoder.backward(dxhat)
        self._c1_pending = None
        if ctx.comm.on and self._dec_range is not None:
            # C1, first bucket: every decoder gradient is final here (~3/4 of the 72 MB); its all-reduce runs over xGMI
            # while the inner-VAE and encoder backward kernels keep the CUs busy.  The rest follows in _ae.
            lo, hi = self._dec_range
            ctx.join_side()                                       # the decoder's filter gradients are final only after the side stream
            self._c1_pending = ctx.comm.allreduce_async_(self.ps.grad["ae"][lo:hi], "C1 bucket 1 (decoder, async)")
        mode = 1
        if self.has_inner and not self.use_sg:
            mu_t, sd_t, sdraw_t, eps_t, t = self.lat_t
            dzhat = ctx.empty(B, Z)
            L.call("ladder_code_grad", _p(z), _p(self.zhat), _p(sd), int(self.use_mask), _p(self.scalars), _p(dz), _p(dzhat), B, Z, st)
            dt = self.inner.decode_backward(dzhat, wgrad=False)
            dmu_t, dsdraw_t = ctx.empty(B, R), ctx.empty(B, R)
            gm_mu, gm_sd = (None, None) if self.hier else self.gmm_grads       # hierarchical: closed-form N(0,I) term (mode bit 1)
            L.call("ladder_latent_bwd", _p(dt), _p(mu_t), _p(sd_t), _p(sdraw_t), _p(eps_t), _p(gm_mu),
                   _p(gm_sd), -1.0, _p(self.scalars), 3 if self.hier else 1, _p(dmu_t), _p(dsdraw_t), B, R, st)
            add_(ctx, dz, self.inner.encode_backward(dmu_t, dsdraw_t, wgrad=False, need_dz=True))
        else:
            mode = 3
        ex_mu, ex_sd = (None, None)
        if self.gmm_z or vamp_on:                             # -crossEntropy_prior = -(1/LB) sum log p(z_l): mixture grads, no SG term
            mode, (ex_mu, ex_sd) = 1, self.gmm_grads
        dmu, dsdraw = ctx.empty(B, Z), ctx.empty(B, Z)
        L.call("ladder_latent_bwd", _p(dz), _p(mu), _p(sd), _p(sd_raw), _p(eps_z), _p(ex_mu), _p(ex_sd), -1.0, _p(self.scalars), mode,
               _p(dmu), _p(dsdraw), B, Z, st)
        self.encoder.backward(dmu, dsdraw)
        ctx.join_side()
        if saved_enc is not None:
            lo, hi = self._enc_range
            add_(ctx, self.ps.grad["ae"][lo:hi], saved_enc)

    def _backward_prior(self):
        ctx, st, B, Z, R = self.ctx, self.ctx.stream, self.B, self.Z, self.R
        mu, sd, sd_raw, eps_z, z = self.lat_z
        mu_t, sd_t, sdraw_t, eps_t, t = self.lat_t
        dzhat = ctx.empty(B, Z)
        L.call("ladder_code_grad", _p(z), _p(self.zhat), _p(sd), int(self.use_mask), _p(self.scalars), None, _p(dzhat), B, Z, st)
        dt = self.inner.decode_backward(dzhat, wgrad=True)
        dmu_t, dsdraw_t = ctx.empty(B, R), ctx.empty(B, R)
        gm_mu, gm_sd = (None, None) if self.hier else self.gmm_grads
        L.call("ladder_latent_bwd", _p(dt), _p(mu_t), _p(sd_t), _p(sdraw_t), _p(eps_t), _p(gm_mu),
               _p(gm_sd), -1.0, _p(self.scalars), 3 if self.hier else 1, _p(dmu_t), _p(dsdraw_t), B, R, st)
        self.inner.encode_backward(dmu_t, dsdraw_t, wgrad=True, need_dz=False)

    # -- the four runs --------------------------------------------------------------------------
    def _ae(self, x, lr, noise, use_sg, use_mask, reuse_encoder=False):
        # in the SG regime the inner VAE does not enter loss_ae's gradient (tf.cond, base.py:318-320) nor its fetches
        parts = ("dec",) if (use_sg or not self.has_inner) else ("dec", "inner", "gmm")
        if self.gmm_z:
            parts = ("dec", "gmm", "gmm_grad")
        if self.vamp:
            parts = ("dec", "gmm")
        self.forward(x, noise, use_sg, use_mask, parts)
        self._backward_ae()
        g = self.ps.grad["ae"]                                    # C1 (sum of per-rank grads of the global-mean loss)
        if getattr(self, "_c1_pending", None) is not None:
            lo, hi = self._dec_range                              # decoder bucket already in flight: reduce what is left
            if lo > 0:
                self.ctx.comm.allreduce_(g[:lo], "C1 bucket 2 (rest)")
            if hi < g.numel():
                self.ctx.comm.allreduce_(g[hi:], "C1 bucket 2 (rest)")
            self._c1_pending.wait()
            self._c1_pending = None
        else:
            self.ctx.comm.allreduce_(g, "C1 (one bucket)")
        self.ps.adam("ae", lr)
        self._repack_filters()

    def _repack_filters(self):
        """Every split filter image the convolutions have asked for so far, re-packed from the just-updated weights in TWO launches
        (ladder_filter_pack_split_multi) instead of memset + absmax + pack per bank on first use (~25 banks on the CelebA nets).  Eager
        mode only: a hipGraph capture keeps the lazy per-bank path (it re-packs inside the graph)."""
        ctx = self.ctx
        banks = ctx.pack_banks
        if not banks or self.use_graphs or torch.cuda.is_current_stream_capturing():
            return
        tab = ctx._pack_table
        if tab is None or tab[0] != len(banks):
            import numpy as np
            dt = np.dtype([("w", "<u8"), ("packed", "<u8"), ("ntaps", "<i4"), ("cin", "<i4"), ("cout", "<i4"), ("flip", "<i4"),
                           ("block_begin", "<i4"), ("reserved", "<i4")])
            rows, blk = np.zeros(len(banks), dtype=dt), 0
            for r, (ent, w, taps, cin, cout, flip, ns, _grp) in zip(rows, banks):
                assert ns == ctx.ns
                r["w"], r["packed"], r["ntaps"], r["cin"], r["cout"], r["flip"], r["block_begin"] = w.data_ptr(), ent[1].data_ptr(), taps, cin, cout, flip, blk
                blk += L.query("ladder_filter_pack_job_blocks", taps, cin, cout)
            dev = torch.from_numpy(rows.view(np.uint8).copy()).to(ctx.device)
            scratch = torch.empty(L.query("ladder_filter_pack_split_multi_scratch_bytes", len(banks)), dtype=torch.uint8, device=ctx.device)
            tab = ctx._pack_table = (len(banks), dev, blk, scratch)
        L.call("ladder_filter_pack_split_multi", _p(tab[1]), tab[0], tab[2], ctx.ns, _p(tab[3]), tab[3].numel(), ctx.stream)
        for ent, *_, grp in banks:
            ent[0] = self.ps.version[grp]

    def _sigma(self, x, lr, noise, use_sg, use_mask, reuse_encoder=False):
        self.forward(x, noise, use_sg, use_mask, ("dec",), reuse_encoder, keep_acts=False)
        self.ps.adam("sigma", lr, grad=self._sc("_g_sigma_var"), n=1)

    def _prior(self, x, lr, noise, use_sg, use_mask, reuse_encoder=False):
        if self.vamp:
            # loss_prior = -elbo w.r.t. the pseudo-inputs (base.py:408-409, 474-481): only the mixture term depends on them; with
            # the standard-Gaussian switch on, the gradient is identically zero (the optimiser still steps, as tf.cond does)
            self.forward(x, noise, use_sg, use_mask, ("dec", "gmm"), reuse_encoder)
            g = self.ps.g["prior/Variable"]
            if use_sg:
                L.call("ladder_axpy", None, _p(g), g.numel(), 0.0, 2, self.ctx.stream)
            else:
                g.copy_(self._vamp_backward(wgrad=False, need_input_dx=True))
            self.ctx.comm.allreduce_(self.ps.grad["prior"], "C4 prior gradients")       # C4
            self.ps.adam("prior", lr)
            return
        self.forward(x, noise, use_sg, use_mask, ("inner", "gmm"), reuse_encoder)
        self._backward_prior()
        self.ctx.comm.allreduce_(self.ps.grad["prior"], "C4 prior gradients")           # C4
        self.ps.adam("prior", lr)

    def _inner_sigma(self, x, lr, noise, use_sg, use_mask, reuse_encoder=False):
        self.forward(x, noise, use_sg, use_mask, ("inner",), reuse_encoder)
        self.ps.adam("inner_sigma", lr, grad=self._sc("_g_inner_sigma_var"), n=1)

    _GROUP = {"ae": "ae", "sigma": "sigma", "prior": "prior", "inner_sigma": "inner_sigma"}
    _SNAP = ("x", "B", "Bg", "lat_z", "lat_t", "xhat", "zhat", "gmm_grads", "vamp_state", "use_sg", "use_mask")

    def _run(self, kind, x, lr, noise, use_sg, use_mask, reuse_encoder):
        """Eager, or -- with `use_graphs` -- one captured hipGraph per (run kind, regime, batch shape): after two eager
        warm-up calls the run's ~10^2..10^3 launches are replayed as a single graph launch.  Every per-step scalar (Adam step /
        lr_t, noise stream position) lives in device memory, so a replay is exactly the eager run."""
        fn = getattr(self, "_" + kind)
        # (also on a graph REPLAY, which does not pass through forward(): RUN#3 / RUN#4 evaluate no decoder -> their sigma-dependent scalars are NaN)
        self._undefined = DEC_SCALARS if (kind in ("prior", "inner_sigma") and not self.vamp) else ()
        if not self.use_graphs or noise is not None or self.ctx.comm.on:
            cache = getattr(self, "_enc_cache", None)
            if (self._aux_on and not self.use_graphs and kind in ("prior", "inner_sigma") and reuse_encoder and not self.vamp
                    and self._enc_event is not None and cache is not None and cache[0] == self.ps.step["ae"]
                    and cache[4] == self._batch_token(x)):
                return self._run_on_aux(fn, x, lr, noise, use_sg, use_mask)
            self._join_aux()
            self._fetch_src = (self.scalars, torch.cuda.current_stream(self.ctx.device))
            try:
                return fn(x, lr, noise, use_sg, use_mask, reuse_encoder)
            finally:
                self.ctx.join_side()
        self._join_aux()
        self._fetch_src = (self.scalars, torch.cuda.current_stream(self.ctx.device))
        group = self._GROUP[kind]
        tok = self._batch_token(x)
        xin = self._dev(x)
        cache = getattr(self, "_enc_cache", None)
        reuse = bool(reuse_encoder and cache is not None and cache[0] == self.ps.step["ae"] and cache[4] == tok)
        key = (kind, bool(use_sg), bool(use_mask), tuple(xin.shape), cache[2].data_ptr() if reuse else 0, self._gm_packed.data_ptr()
               if self._gm_packed is not None else 0, self.ctx.ws_generation)
        ent = self._graphs.get(key)
        if ent is None:
            if self._warm.get(key, 0) < 2:                         # eager warm-up: sizes the workspace and the allocator
                self._warm[key] = self._warm.get(key, 0) + 1
                self._x_token = tok
                return fn(xin, lr, None, use_sg, use_mask, reuse_encoder)
            static_x = cache[1] if reuse else xin.clone()
            self.ps.set_lr(group, lr)
            steps, calls = dict(self.ps.step), self._run_calls
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._x_token = tok
            with torch.cuda.graph(graph):                          # records the launches; nothing executes, host counters restored
                fn(static_x, lr, None, use_sg, use_mask, reuse_encoder)
            snap = {k: getattr(self, k, None) for k in self._SNAP}
            enc = None if reuse else self._enc_cache[1:4]
            self.ps.step, self._run_calls = steps, calls
            if self.ctx.ws_generation != key[-1]:                  # the workspace grew while recording: this graph is void
                self._warm[key[:-1] + (self.ctx.ws_generation,)] = 2
                return self._run(kind, x, lr, noise, use_sg, use_mask, reuse_encoder)
            ent = self._graphs[key] = (graph, static_x, snap, enc)
        graph, static_x, snap, enc = ent
        if not reuse and static_x.data_ptr() != xin.data_ptr():
            static_x.copy_(xin)
        self.ps.set_lr(group, lr)
        if enc is not None:
            self._enc_cache = (self.ps.step["ae"],) + tuple(enc) + (tok,)   # as the eager forward does (pre-update step)
        graph.replay()
        self.ps.step[group] += 1
        # the replay changed the group's weights on the device: packed split-filter images stamped with the old version are stale
        # for any EAGER forward that follows (val_step, fit_GMM_VI, decode, Session) -- graphs re-pack inside their own capture
        self.ps.version[group] += 1
        for k, v in snap.items():
            setattr(self, k, v)

    def _run_on_aux(self, fn, x, lr, noise, use_sg, use_mask):
        """One of RUN#3 / RUN#4 on the aux stream (see enable_prior_overlap)."""
        aux = self.ctx.aux
        aux.wait_event(self._enc_event)
        main_bufs = (self.partials, self.scalars)
        self.partials, self.scalars, self._on_aux = self.partials_aux, self.scalars_aux, True
        try:
            with torch.cuda.stream(aux):
                fn(x, lr, noise, use_sg, use_mask, True)
                self._aux_event = torch.cuda.Event()
                self._aux_event.record(aux)
        finally:
            self.partials, self.scalars = main_bufs
            self._on_aux = False
        self._fetch_src = (self.scalars_aux, aux)

    def run_ae(self, x, lr, noise=None, use_sg=True, use_mask=False):
        self._run("ae", x, lr, noise, use_sg, use_mask, False)

    def run_sigma(self, x, lr, noise=None, use_sg=True, use_mask=False, reuse_encoder=False):
        self._run("sigma", x, lr, noise, use_sg, use_mask, reuse_encoder)

    def run_prior(self, x, lr, noise=None, use_sg=True, use_mask=False, reuse_encoder=False):
        self._run("prior", x, lr, noise, use_sg, use_mask, reuse_encoder)

    def run_inner_sigma(self, x, lr, noise=None, use_sg=True, use_mask=False, reuse_encoder=False):
        self._run("inner_sigma", x, lr, noise, use_sg, use_mask, reuse_encoder)

    def evaluate(self, x, noise=None, use_sg=True, use_mask=False):
        parts = ("dec", "inner", "gmm") if (((self.has_inner or self.gmm_z) and self._gm_packed is not None) or self.vamp) else ("dec", "inner")
        self.forward(x, noise, use_sg, use_mask, parts, keep_acts=False)

    # -- generation -----------------------------------------------------------------------------
    def decode(self, code):
        """decoded given code_input (is_code_input=True; models.py:107,265,500)."""
        self._join_aux()
        self.ctx.keep_activations = False
        try:
            return self.decoder.forward(self._dev(code))
        finally:
            self.ctx.keep_activations = True

    def decode_representation(self, t):
        """decoded_code given representation_input (base.py:171-186)."""
        self._join_aux()
        return self.inner.decode(self._dev(t))

    def sample_code(self, x, noise=None):
        """code_sample for fit_GMM_VI(space="z") (base.py:699-710): encoder -> z."""
        self.forward(x, noise, True, False, ())
        return self.lat_z[4]

    def sample_representation(self, x, noise=None):
        """representation_sample for fit_GMM_VI (base.py:683-698): encoder -> z -> inner encoder -> t."""
        self.forward(x, noise, True, False, ("inner",))
        return self.lat_t[4]
