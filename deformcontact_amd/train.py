"""Training step / loop with the reference's step maths (``/root/reference/train.py:46-58,71-73``).

    pred = model(rest, rigid);  pred.pos -= rest.pos;  target.pos = def.pos - rest.pos
    loss = L1(pred.pos, target.pos) + lambda_gradient * GradientConsistency(pred, target)
    Adam(lr = 4e-4)

No wandb (hard dependency of the reference's loop): scalars go to a JSONL file, and are only
pulled off the device when they are logged (the reference does 7 ``.item()`` syncs per step,
``train.py:60-70``).  With ``torch.distributed`` initialised the gradients are averaged through
one flat bucket (``deformcontact_amd.dp``).
"""
from __future__ import annotations

import json
import os
import time
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dp
from .graphnet import EVERYDAY_NETWORK, gradient_consistency_loss, load_model
from .loaders import InMemoryDataset, PrefetchLoader, SyntheticEverydayDataset, iterate_batches, to_batches


#: both losses (and their gradients) in ONE node pass of the library (``ops.contact_losses``) when the
#: batches live on a HIP device and prediction and target are KNOWN to share the rest batch's edge set.
#: The reference pairs the deformed mesh's own edges with the prediction's (``models/losses.py:12-13``) and
#: builds them from a separate mesh file (``loaders/everyday_deform.py``), so equality is a property of
#: the data, not of the code: it is decided ONCE PER BATCH ON THE HOST, where the tensors still are
#: (``loaders.mark_edge_equality``, called by ``to_batches`` / ``PrefetchLoader`` / ``synth.make_batch``),
#: and carried on the batch object.  Unknown or unequal -> the stock formulation of ``models/losses.py``.
#: ``DC_FUSED_LOSS=0`` keeps the stock formulation always; ``DC_LOSS_DEBUG=1`` adds a device-side
#: comparison that poisons both losses with NaN on a mismatch (a debug assert, two E-sized launches).
FUSED_LOSS = os.environ.get("DC_FUSED_LOSS", "1") != "0"
LOSS_DEBUG = os.environ.get("DC_LOSS_DEBUG", "0") == "1"


def _same_edges(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a is b or (a.shape == b.shape and a.data_ptr() == b.data_ptr())


def _edges_known_equal(batch, ei: torch.Tensor) -> bool:
    """Is ``batch.edge_index`` known - without looking at device memory - to equal ``ei``?  Same storage, a
    ``clone()`` of the batch that owns ``ei`` (``GraphNet.forward`` returns ``graph_resting.clone()``,
    ``models/model.py:91``), or marked equal on the host by the loader."""
    other = batch.edge_index
    if _same_edges(other, ei):
        return True
    src = getattr(batch, "_dc_cloned_edges", None)
    if src is not None and src[0] is ei and src[1] == ei._version and other.shape == ei.shape:
        return True
    mark = getattr(batch, "_dc_edges_equal", None)
    if mark is None or not mark[0] or other.shape != ei.shape:
        return False
    # a mark made before .to(device) cannot name the device copy; one that does name a tensor must still match it
    # (an in-place rewrite bumps the version, a replaced edge_index drops the mark in Batch.__setattr__)
    if len(mark) >= 4 and mark[1] == other.data_ptr():
        return mark[2] == other._version and tuple(mark[3]) == tuple(other.shape)
    return True


def losses(model, rest, deff, rig, lambda_gradient: float = 1.0) -> Dict[str, torch.Tensor]:
    pred = model(rest, rig)
    pred.pos = pred.pos - rest.pos
    tgt_pos = deff.pos - rest.pos
    ei = rest.edge_index
    if (FUSED_LOSS and pred.pos.is_cuda and pred.pos.dtype == torch.float32 and pred.pos.size(0) > 0
            and ei.size(1) > 0 and _edges_known_equal(pred, ei) and _edges_known_equal(deff, ei)):
        from . import ops
        from .graph import graph_index
        # The loss runs over the adjacency the encoder built for rest.edge_index (a cache hit): the
        # reference takes the edges from pred (a clone of the rest batch, model.py:91) and from the
        # deformed batch; both are known to equal rest's here (decided on the host, see FUSED_LOSS).
        g = graph_index(ei, pred.pos.size(0))
        if getattr(g, "row_offset", 0):
            raise RuntimeError("losses: the soft graph must be the first part of a merged adjacency")
        l1, gcl = ops.contact_losses(g, pred.pos, tgt_pos)
        if LOSS_DEBUG:
            bad = None
            for other in (pred.edge_index, deff.edge_index):
                if not _same_edges(other, ei):
                    ne = (other != ei).any()
                    bad = ne if bad is None else (bad | ne)
            if bad is not None:
                nan = torch.full_like(l1, float("nan"))
                l1, gcl = torch.where(bad, nan, l1), torch.where(bad, nan, gcl)
    else:
        tgt = deff.clone()
        tgt.pos = tgt_pos
        l1 = F.l1_loss(pred.pos, tgt.pos)
        gcl = gradient_consistency_loss(pred, tgt)
    return {"loss": l1 + lambda_gradient * gcl, "l1": l1, "consistency": gcl}


def train_step(model, optimizer, rest, deff, rig, lambda_gradient: float = 1.0,
               bucket: Optional[dp.GradBucket] = None) -> Dict[str, torch.Tensor]:
    out = losses(model, rest, deff, rig, lambda_gradient)
    if getattr(optimizer, "zero_grad_in_step", False):
        pass                                   # FlatAdam cleared the gradients it consumed
    elif bucket is not None:
        bucket.zero()
    else:
        optimizer.zero_grad(set_to_none=True)
    out["loss"].backward()
    if bucket is not None:
        bucket.all_reduce_mean()
    optimizer.step()
    return {k: v.detach() for k, v in out.items()}


_BATCH_TENSORS = ("x", "pos", "edge_index", "batch", "ptr")


class GraphedTrainStep:
    """The reference's whole train step (``train.py:46-58,71-73``: forward, both losses, backward,
    Adam) as ONE hipGraph over static input buffers.  Every call copies the new batches into the
    buffers and replays; the graph CONTAINS the per-batch topology work (sorted adjacency +
    gcn_norm build, first-layer hops), so any batch of the captured shape is handled correctly.
    The first ``eager_steps`` calls (and any call whose shapes differ from the captured ones) run
    eagerly - they are real training steps, nothing is replayed twice.  With ``world_size > 1``
    the gradient all-reduce and Adam stay outside the graph.  Returned loss tensors are the
    graph's static outputs: read them before the next call."""

    def __init__(self, model, optimizer: "dp.FlatAdam", bucket: dp.GradBucket,
                 lambda_gradient: float = 1.0, eager_steps: int = 2):
        self.model, self.opt, self.bucket, self.lam = model, optimizer, bucket, float(lambda_gradient)
        self.eager_steps, self._seen = int(eager_steps), 0
        self._sig = self._graph = self._static = self._out = None
        self._tail_in_graph = dp.world_size() == 1
        self.replays = 0

    @staticmethod
    def _signature(batches):
        # shapes + the host-side edge-equality mark of every batch: it selects the loss formulation the
        # captured graph contains, so a batch with a different mark must not replay it; + the batch layout
        # (node / edge offsets of its graphs): the one-launch adjacency build carries it in its kernel arguments
        return tuple((k, tuple(getattr(b, k).shape)) for b in batches for k in _BATCH_TENSORS
                     if isinstance(getattr(b, k, None), torch.Tensor)) + \
            tuple(bool((getattr(b, "_dc_edges_equal", None) or (False,))[0]) for b in batches) + \
            tuple(b.segments() if callable(getattr(b, "segments", None)) else None for b in batches)

    def _fwd_bwd(self, rest, deff, rig):
        out = losses(self.model, rest, deff, rig, self.lam)
        out["loss"].backward()
        return {k: v.detach() for k, v in out.items()}

    def _tail(self):
        self.bucket.all_reduce_mean()
        self.opt.step()

    def __call__(self, rest, deff, rig) -> Dict[str, torch.Tensor]:
        sig = self._signature((rest, deff, rig))
        if sig != self._sig:
            self._sig, self._seen, self._graph = sig, 0, None
        if self._graph is None and self._seen < self.eager_steps:
            self._seen += 1
            out = self._fwd_bwd(rest, deff, rig)
            self._tail()
            return out
        if self._graph is None:
            self._static = tuple(b.clone() for b in (rest, deff, rig))
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with dp.capture(graph):
                self._out = self._fwd_bwd(*self._static)
                if self._tail_in_graph:
                    self._tail()
            self._graph = graph
        for dst, src in zip(self._static, (rest, deff, rig)):
            for k in _BATCH_TENSORS:
                t = getattr(dst, k, None)
                if isinstance(t, torch.Tensor):
                    t.copy_(getattr(src, k), non_blocking=True)
        self._graph.replay()
        self.replays += 1
        if self.replays == 1:
            from .graph import validate_pending
            validate_pending()                  # DC_VALIDATE=1: adjacencies built under the capture
        if not self._tail_in_graph:
            self._tail()
        return self._out


def train(network_cfg=None, *, device="cuda", epochs: int = 2, num_train: int = 64, num_val: int = 16,
          batch_size: int = 4, lr: float = 4e-4, lambda_gradient: float = 1.0, out_dir: str = "runs/dc",
          soft_vertices: int = 1024, sphere_resolution: int = 20, log_every: int = 1, seed: int = 0,
          conv_module=None, capture: bool = True, stats: Optional[dict] = None):
    """Synthetic-data counterpart of ``train.py:15-135`` (best-val checkpoint + config dump).

    On a HIP device with the library's conv layers: Adam is ``dp.FlatAdam`` (one kernel over the
    flat bucket, ``train.py:20`` defaults), gradients are accumulated straight into the bucket
    (``GradBucket(direct=True)``), the loader prepares each batch's topology on its own stream
    (``loaders.prepare_for`` + ``TopologyCache``) and, with ``capture``, steps of a repeating shape
    are hipGraph replays (``GraphedTrainStep``).  ``stats`` (optional dict) receives step timings."""
    cfg = dict(EVERYDAY_NETWORK if network_cfg is None else network_cfg)
    torch.manual_seed(seed)
    model = load_model(cfg, conv_module=conv_module).to(device)
    dp.broadcast_parameters(model)
    on_gpu = torch.device(device).type == "cuda" and conv_module is None
    if on_gpu:
        bucket = dp.GradBucket(model.parameters(), direct=True)
        opt = dp.FlatAdam(bucket, lr=lr, zero_grad_in_step=True)
        bucket.zero()
        graphed = GraphedTrainStep(model, opt, bucket, lambda_gradient) if capture else None
    else:                                       # CPU oracle convs (tests): stock optimizer
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        bucket = dp.GradBucket(model.parameters()) if dp.world_size() > 1 else None
        graphed = None
    rank = torch.distributed.get_rank() if dp.world_size() > 1 else 0
    # graphs built once, before the loop (the reference's dataset object likewise holds its processed meshes)
    train_ds = InMemoryDataset(SyntheticEverydayDataset(num_train, rank * num_train, soft_vertices, sphere_resolution))
    val_ds = InMemoryDataset(SyntheticEverydayDataset(num_val, 10_000_000, soft_vertices, sphere_resolution))
    os.makedirs(out_dir, exist_ok=True)
    log = open(os.path.join(out_dir, f"log_rank{rank}.jsonl"), "a")
    best, step = float("inf"), 0
    # with capture the graph holds the topology work; otherwise the loader prepares it off-stream
    from .loaders import TopologyCache, prepare_for
    prepare = prepare_for(model, TopologyCache()) if (on_gpu and graphed is None) else None
    t_steps = []
    for epoch in range(epochs):
        model.train()
        # worker thread assembles + pins the next batches, uploads overlap the current step
        for collated, (rest, deff, rig) in PrefetchLoader(train_ds, batch_size, device, shuffle=True,
                                                          seed=seed + epoch, prepare=prepare):
            t0 = time.perf_counter()
            if graphed is not None:
                out = graphed(rest, deff, rig)
            else:
                out = train_step(model, opt, rest, deff, rig, lambda_gradient, bucket)
            if step % log_every == 0:
                log.write(json.dumps({"step": step, "epoch": epoch, "t": time.time(),
                                      **{k: float(v) for k, v in out.items()}}) + "\n")
            t_steps.append(time.perf_counter() - t0)
            step += 1
        model.eval()
        tot, nb = 0.0, 0
        with torch.no_grad():
            for collated in iterate_batches(val_ds, batch_size):
                rest, deff, rig = to_batches(collated, device)
                pred = model(rest, rig)                     # validation on absolute positions
                val = F.l1_loss(pred.pos, deff.pos) + lambda_gradient * gradient_consistency_loss(pred, deff)
                tot, nb = tot + float(val), nb + 1
        val_loss = tot / max(nb, 1)
        log.write(json.dumps({"epoch": epoch, "validation_loss": val_loss}) + "\n")
        log.flush()
        if rank == 0 and val_loss < best:
            best = val_loss
            torch.save(model.state_dict(), os.path.join(out_dir, "model_weights.pth"))
            with open(os.path.join(out_dir, "config.json"), "w") as f:
                json.dump({"network": cfg, "training": {"learning_rate": lr,
                                                        "lambda_gradient": lambda_gradient}}, f, indent=4)
    log.close()
    if stats is not None:
        stats.update(steps=step, step_seconds=t_steps,
                     graph_replays=graphed.replays if graphed is not None else 0)
    return model, best


if __name__ == "__main__":
    train()
