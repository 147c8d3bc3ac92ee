"""Exchange steps of the data-parallel scheme (C1-C4) -- over torch.distributed (RCCL on ROCm) between processes, or between VIRTUAL
ranks (threads of one process on one device: bit-exact reproduction of an N-process job), or not at all (replicated computations)."""
import os

import torch


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
