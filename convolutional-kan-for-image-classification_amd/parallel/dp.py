"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The conv-KAN path shards on the batch axis only (samples are independent through conv, per-sample
InstanceNorm and PReLU -- SURVEY.md section 8(e)), so the single exchange step per iteration is the
all-reduce (mean) of the parameter gradients: 331.9 MB fp32 for KAN-VGG11, 99.99 % of it conv weights.

Design for MI355X: xGMI is point-to-point, so a few LARGE messages beat many small ones.  Parameters are
packed into flat fp32 buckets in REVERSE registration order (the order backward produces gradients; the three
512->512 layers = 3 x 85 MB come first), each bucket is all-reduced on a side HIP stream as soon as its last
gradient has been written, and the compute stream only joins at ``finish()``.  The reference has no
distributed code at all, so there is no call pattern to mirror.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List, Optional

import weakref

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.seen = [False] * len(params)       # gradient of parameter i arrived this step
        self.work = None


class BucketedGradReducer:
    """Average gradients across ranks, overlapped with backward.

    usage:   red = BucketedGradReducer(model.parameters()); ...; loss.backward(); red.finish()
    After ``finish()`` every ``p.grad`` holds the mean gradient over the process group.

    Exactly ONE backward pass per ``finish()`` may run with the exchange armed.  Gradient accumulation over several
    backward passes goes through ``no_sync()`` (as with torch DDP): passes inside it only accumulate into ``p.grad``, the
    first pass outside it exchanges the accumulated sum.  A second armed pass before ``finish()`` raises -- the buckets
    of the first would already be in flight.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 96 << 20,
                 group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False):
        self.group = group
        self.always_reduce = always_reduce          # run the collective even in a 1-rank group (exercises the path)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        plist = [p for p in params if p.requires_grad]
        self.buckets: List[_Bucket] = []
        cur, size = [], 0
        for p in reversed(plist):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > bucket_bytes or p.device != cur[0].device or p.dtype != cur[0].dtype):
                self.buckets.append(_Bucket(cur)); cur, size = [], 0
            cur.append(p); size += nbytes
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[p] = (b, i)
        self.cuda = bool(plist) and plist[0].is_cuda
        # RCCL averages in the collective itself (ReduceOp.AVG); gloo (CPU tests) sums and the mean is taken afterwards
        self.avg_in_collective = bool(dist.is_initialized() and dist.get_backend(group) == "nccl")
        self.side = torch.cuda.Stream(device=plist[0].device) if self.cuda else None
        self._armed = True
        self._trace = None                          # see trace_step()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in plist]
        # conv-KAN weights: let the weight-gradient kernels write into the bucket directly (ops.GRAD_SINKS); the hook then
        # finds .grad already in place and skips its copy
        self._sinks = []
        if self.cuda:
            from .. import ops
            for b in self.buckets:
                for v, p in zip(b.views, b.params):
                    if p.dim() == 4:
                        ops.GRAD_SINKS[id(p)] = ops.GradSink(weakref.ref(p), v)
                        self._sinks.append(id(p))

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside only accumulate into ``p.grad`` (no bucket copy, no collective)."""
        prev, self._armed = self._armed, False
        try:
            yield self
        finally:
            self._armed = prev

    # -- called by autograd right after p.grad has been written
    def _on_grad(self, p: torch.nn.Parameter):
        if not self._armed:
            return
        b, i = self._where[p]
        if b.seen[i]:
            raise RuntimeError("BucketedGradReducer: a second gradient arrived for a parameter before finish() -- the bucket of the "
                               "first backward pass is already being all-reduced.  Wrap all but the last backward pass of an "
                               "accumulation step in reducer.no_sync().")
        b.seen[i] = True
        if p.grad.data_ptr() != b.views[i].data_ptr():      # (already there when the kernel wrote through a gradient sink)
            b.views[i].copy_(p.grad)
        b.pending -= 1
        assert b.pending >= 0
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        if self.world == 1 and not self.always_reduce:
            return
        if self.cuda:
            tr = self._trace
            cur = torch.cuda.current_stream(b.flat.device)
            if tr is not None:                      # when the bucket's last gradient is on the compute stream
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(cur)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                if tr is not None:
                    ev[1].record(self.side)
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM,
                                         group=self.group, async_op=True)
                if tr is not None:                  # the backend runs collectives on a stream of its own: make the (otherwise idle) side
                    b.work.wait()                   # stream wait for this one, so that an event on it marks the collective's end
                    ev[2].record(self.side)
                    tr["buckets"].append((self.buckets.index(b), ev))
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Flush buckets whose gradients never all arrived (unused parameters), wait, scale, publish as p.grad."""
        tr = self._trace if self.cuda else None
        if tr is not None:                          # the compute stream has every backward kernel enqueued: "backward ends" here
            tr["fin0"] = torch.cuda.Event(enable_timing=True); tr["fin0"].record(torch.cuda.current_stream(self.buckets[0].flat.device))
        self._finish()
        if tr is not None:
            tr["fin1"] = torch.cuda.Event(enable_timing=True); tr["fin1"].record(torch.cuda.current_stream(self.buckets[0].flat.device))

    @contextlib.contextmanager
    def trace_step(self, out: dict):
        """Diagnostics for ONE step (outside any timed region): HIP events around every bucket's collective and around finish().
        On exit (after a device sync) `out` holds, in ms relative to the event `out['t0']` the caller recorded before backward:
          exposed_ms          time the compute stream spends blocked in finish() after its last backward kernel (the un-overlapped tail),
          buckets[i]          {ready_ms: last gradient of the bucket written, start_ms / end_ms: its all-reduce on the side stream, mb}."""
        self._trace = {"buckets": []}
        try:
            yield self
        finally:
            tr, self._trace = self._trace, None
            if self.cuda and "fin1" in tr and out.get("t0") is not None:
                torch.cuda.synchronize()
                t0 = out.pop("t0")
                out["backward_end_ms"] = round(t0.elapsed_time(tr["fin0"]), 3)
                out["exposed_ms"] = round(tr["fin0"].elapsed_time(tr["fin1"]), 3)
                out["buckets"] = [{"bucket": i, "mb": round(self.buckets[i].flat.numel() * 4 / 2 ** 20, 1), "ready_ms": round(t0.elapsed_time(e[0]), 3),
                                   "start_ms": round(t0.elapsed_time(e[1]), 3), "end_ms": round(t0.elapsed_time(e[2]), 3)} for i, e in tr["buckets"]]

    def _finish(self):
        for b in self.buckets:
            if b.pending != 0:                      # some parameter had no gradient this step: send what we have
                for v, p in zip(b.views, b.params):
                    if p.grad is None:
                        v.zero_()
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()                       # on CUDA: makes the CURRENT stream wait for the collective
                b.work = None
            if self.world > 1 and not (self.avg_in_collective and self.cuda):
                b.flat.mul_(1.0 / self.world)
            for v, p in zip(b.views, b.params):
                if p.grad is not None:
                    p.grad = v
            b.pending = len(b.params)
            b.seen = [False] * len(b.params)
        if self._sinks:                                 # sinks are single-use per step (a parameter feeding two graph nodes)
            from .. import ops
            for k in self._sinks:
                ent = ops.GRAD_SINKS.get(k)
                if ent is not None:
                    ent.used = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._sinks:
            from .. import ops
            for k in self._sinks:
                ops.GRAD_SINKS.pop(k, None)
            self._sinks = []
