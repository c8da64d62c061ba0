"""Data-parallel gradient exchange: one flat fp32 bucket, one all-reduce per step.

The reference has no distributed code at all.  A PyG batch is
a block-diagonal union of per-sample graphs, so the hot path shards by sample with
no halo exchange; the only collective is the gradient average of the replicated
parameters - 1,033,219 fp32 = 4.13 MB for the full network, one bucket.  On
MI355X ``backend="nccl"`` is RCCL over xGMI; with 7 point-to-point links per GPU a
4 MB all-reduce is latency-bound (tens of microseconds), so it is issued once
after backward, not bucketed/overlapped.  The same code runs on ``gloo`` for the
CPU tests.

Gradients are *views* into the flat buffer (autograd accumulates in place), so
there is no pack/unpack copy and the buffer address is stable under hipGraph
capture.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def capture(graph: "torch.cuda.CUDAGraph", **kw):
    """``torch.cuda.graph(graph)`` for a step that is captured in a process that also owns a process group.
    With the RCCL ("nccl") backend a watchdog THREAD polls the events of issued collectives (``hipEventQuery``); a
    capture in the default ``capture_error_mode="global"`` makes that call from another thread an error (``operation
    not permitted when stream is capturing``), the watchdog thread dies with the exception and the process aborts -
    intermittently: it needs a poll to fall inside the ~0.1 s a step's capture takes (``tools/r05/rccl_step_probe.py``
    hit it on the first run).  ``thread_local`` keeps the check for the capturing thread only."""
    if "capture_error_mode" not in kw and dist.is_available() and dist.is_initialized():
        kw["capture_error_mode"] = "thread_local"
    return torch.cuda.graph(graph, **kw)


_HOST_STAGE = {}


def _host_stage(t: torch.Tensor) -> torch.Tensor:
    key = (t.numel(), t.dtype)
    if key not in _HOST_STAGE:
        if len(_HOST_STAGE) > 8:
            _HOST_STAGE.clear()
        _HOST_STAGE[key] = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
    return _HOST_STAGE[key].view(t.shape)


def staged_collective(fn, t: torch.Tensor, **kw) -> None:
    """``fn(t, **kw)`` (``dist.all_reduce`` / ``dist.broadcast``) for a DEVICE tensor over a backend without device
    support of its own worth using - gloo, the stand-in for RCCL on a one-GPU box and in the CPU tests: the tensor is
    staged through pinned host memory on the CURRENT stream and the collective runs on the host copy.
    ProcessGroupGloo's own path for device tensors copies on streams it takes from the high-priority pool (PyTorch 2.x
    source: ``initializeStreamsEvents``); with two
    processes time-sharing one device that is where round 5's (and round 4's) stalls sat - up to 15 s per 2.3 MB
    all-reduce, once a 60 s timeout (``profiles/r05/t_launch_flakes.txt``)."""
    if not t.is_cuda:
        fn(t, **kw)
        return
    host = _host_stage(t)
    host.copy_(t)                       # blocking: the current stream's work on `t` is done, the bytes are on the host
    fn(host, **kw)
    t.copy_(host)


def _device_collectives(group=None) -> bool:
    return dist.get_backend(group) == "nccl"


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s parameters and buffers."""
    if world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if _device_collectives(group):
                dist.broadcast(t, src=src, group=group)
            else:
                staged_collective(dist.broadcast, t.data, src=src, group=group)


def flat_offsets(params, align: int = 4) -> List[int]:
    """Element offsets of ``params`` in a flat buffer, each rounded up to ``align`` elements (16 bytes of fp32);
    the last entry is the buffer length."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    offs.append(off)
    return offs


class GradBucket:
    """``direct=True`` opts the parameters into direct gradient writes by the library's dW
    kernels (``ops.DIRECT_PARAM_GRAD``): those kernels then accumulate into this bucket's views
    on whatever stream the backward node runs on, and report each write here; the bucket's
    consumers wait for those streams before they read the flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, direct: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("GradBucket: parameters must share device and dtype")
        self.group = group
        # every view starts on a 16-byte boundary (one odd-sized parameter - the decoder's 3-element bias -
        # otherwise leaves everything behind it misaligned, and the dense kernels fall back to their scalar-load
        # forms for those operands: the attention heads of the reference network ran 2 x slower that way); the
        # padding elements stay zero in the gradients, the parameters and both Adam moments
        self.offsets = flat_offsets(self.params)
        self.numel = self.offsets[-1]
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self._views: List[torch.Tensor] = []
        for p, off in zip(self.params, self.offsets):
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self._views.append(v)
        self.direct = bool(direct)
        self._view_of = {id(p): v for p, v in zip(self.params, self._views)}
        self._pending = []          # (event, recorded under stream capture?) of direct writes
        if self.direct:
            import weakref
            ref = weakref.ref(self)
            for p in self.params:
                p._dc_grad_sink = ref

    def owns(self, p, g) -> bool:
        return self._view_of.get(id(p)) is g

    def note_direct_write(self, stream) -> None:
        """A library kernel on ``stream`` has just accumulated into this bucket."""
        from .graph import capture_id
        ev = torch.cuda.Event()
        ev.record(stream)
        self._pending.append((ev, capture_id(self.flat.device)))
        if len(self._pending) > 256:
            # repeated backward passes without a consumer (gradient accumulation): keep the list short -
            # eager events that have completed order nothing any more
            self._pending = [(e, c) for e, c in self._pending if c != 0 or not e.query()]

    def wait_direct_writes(self) -> None:
        """Order every reported direct write before later work on the current stream.  An event
        recorded under the capture the current stream is part of (or eagerly, seen eagerly) becomes a
        stream dependency; an event recorded under a capture that has ENDED is dropped (ending the
        capture required its streams to be joined); an EAGER event seen while capturing cannot become
        a graph dependency - the host waits for it instead, so the captured work starts behind it."""
        if not self._pending:
            return
        from .graph import capture_id
        pending, self._pending = self._pending, []
        cid = capture_id(self.flat.device)
        cur = torch.cuda.current_stream(self.flat.device)
        for ev, ecid in pending:
            if ecid == cid:
                cur.wait_event(ev)
            elif ecid == 0:
                ev.synchronize()

    def zero(self) -> None:
        """Zero all gradients in one memset and re-attach the views if something replaced them."""
        self.wait_direct_writes()               # the memset must not overtake a direct write still in flight
        self.flat.zero_()
        for p, v in zip(self.params, self._views):
            if p.grad is not v:
                p.grad = v

    def _repack(self) -> None:
        # autograd normally accumulates in place; if a .grad was swapped, copy it back in
        for p, v in zip(self.params, self._views):
            if p.grad is not None and p.grad is not v:
                v.copy_(p.grad)
                p.grad = v

    def all_reduce_mean(self) -> None:
        """Average gradients over ranks: one collective on the flat bucket."""
        self.wait_direct_writes()
        self._repack()
        ws = world_size(self.group)
        if ws == 1:
            return
        if dist.get_backend(self.group) == "nccl":       # RCCL averages in the collective itself
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:                                            # gloo (one-GPU stand-in, CPU tests) has no AVG
            staged_collective(dist.all_reduce, self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(ws)


class FlatAdam:
    """``torch.optim.Adam(params, lr)`` at its defaults (``/root/reference/train.py:20``) over flat
    buckets: parameters are re-pointed to views of one flat fp32 buffer (as ``GradBucket`` does for
    gradients), and a step is ONE elementwise HIP kernel (``dc_adam_flat``) instead of torch's
    multi-tensor launch sequence.  hipGraph-replayable (the step count lives on the device).
    ``zero_grad_in_step``: the same kernel also clears the gradients it has just consumed
    (``optimizer.zero_grad()`` of ``train.py:71`` without a separate memset)."""

    def __init__(self, bucket: GradBucket, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 zero_grad_in_step: bool = False):
        from . import _lib
        from .graph import _require_cuda
        self._lib = _lib
        self.bucket, self.lr, self.betas, self.eps = bucket, float(lr), betas, float(eps)
        _require_cuda(bucket.flat, "FlatAdam parameters")
        dev = bucket.flat.device
        self.flat_param = torch.zeros_like(bucket.flat)
        with torch.no_grad():
            for p, off in zip(bucket.params, bucket.offsets):      # same (16-byte aligned) layout as the gradients
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view                      # parameter storage now lives in the bucket
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.zero_grad_in_step = bool(zero_grad_in_step)
        self.step_count = torch.zeros(34, dtype=torch.float32, device=dev)  # [count, kernel tickets...]

    def zero_grad(self) -> None:
        self.bucket.zero()

    def step(self) -> None:
        from .graph import current_stream_ptr
        b = self.bucket
        b.wait_direct_writes()
        rc = self._lib.lib().dc_adam_flat(
            self.flat_param.data_ptr(), b.flat.data_ptr(), self.exp_avg.data_ptr(),
            self.exp_avg_sq.data_ptr(), b.numel, self.step_count.data_ptr(), self.lr,
            self.betas[0], self.betas[1], self.eps, int(self.zero_grad_in_step),
            current_stream_ptr(b.flat.device))
        self._lib.check(rc, "dc_adam_flat")
