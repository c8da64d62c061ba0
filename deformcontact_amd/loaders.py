"""Batch assembly for the hot path: ``collate_fn`` and a synthetic stand-in for the dataset.

``collate_fn`` mirrors ``/root/reference/loaders/collate.py:4-16`` (zip the per-sample
5-tuples; stack tensor-valued meta entries, list the rest).  The real dataset
(``loaders/everyday_deform.py``) needs Open3D and a Google-Drive download that is absent, so
``SyntheticEverydayDataset`` yields items of the same shape
``(obj_name, soft_rest_graph, soft_def_graph, meta_data, rigid_graph)`` (``:72``) from
``deformcontact_amd.synth``.
"""
from __future__ import annotations

from typing import Iterator, List, Sequence, Tuple

import torch

from . import synth
from .data import Batch


def collate_fn(batch: Sequence[tuple]):
    names, rests, defs, metas, rigids = zip(*batch)
    keys = list(metas[0].keys())
    tensor_keys = [k for k in keys if isinstance(metas[0][k], torch.Tensor)]
    meta = {k: torch.stack([m[k] for m in metas]) for k in tensor_keys}
    for k in keys:
        if k not in meta:
            meta[k] = [m[k] for m in metas]
    return list(names), rests, defs, meta, rigids


class SyntheticEverydayDataset:
    OBJECTS = ["Box", "Cat", "Pillow", "Dog", "Donut", "Doritos", "Bottle", "Flipflop"]

    def __init__(self, num_samples: int, first_idx: int = 0, soft_vertices: int = 1024,
                 sphere_resolution: int = 20):
        self.num_samples, self.first_idx = num_samples, first_idx
        self.soft_vertices, self.sphere_resolution = soft_vertices, sphere_resolution

    def __len__(self) -> int:
        return self.num_samples

    def __getitem__(self, i: int):
        if not 0 <= i < self.num_samples:
            raise IndexError(i)
        idx = self.first_idx + i
        rest, deff, rig = synth.make_sample(idx, self.soft_vertices, self.sphere_resolution)
        meta = {"force": float(rig.x[0, 3]), "force_vector": rig.x[0, :3].clone(), "sample_idx": idx}
        return self.OBJECTS[idx % len(self.OBJECTS)], rest, deff, meta, rig


class InMemoryDataset:
    """Samples of ``dataset`` materialised once (the reference's ``EverydayDeformDataset`` likewise keeps its
    pre-processed graphs and only indexes them per step, ``loaders/everyday_deform.py``): ``__getitem__`` is a
    list lookup, so a training loop measures the loader + step, not the mesh synthesiser."""

    def __init__(self, dataset):
        self.samples = [dataset[i] for i in range(len(dataset))]

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, i: int):
        return self.samples[i]


def iterate_batches(dataset, batch_size: int, shuffle: bool = False, seed: int = 0,
                    drop_last: bool = False) -> Iterator[tuple]:
    """Minimal ``DataLoader(dataset, batch_size, shuffle, collate_fn=collate_fn)``."""
    order = list(range(len(dataset)))
    if shuffle:
        g = torch.Generator().manual_seed(seed)
        order = torch.randperm(len(dataset), generator=g).tolist()
    for a in range(0, len(order), batch_size):
        idx = order[a:a + batch_size]
        if drop_last and len(idx) < batch_size:
            break
        yield collate_fn([dataset[i] for i in idx])


def mark_edge_equality(rest, deff) -> None:
    """Decide ON THE HOST (the tensors are still there) whether the deformed batch's edge set equals the rest
    batch's, and record it on the batch: ``train.losses`` runs the fused loss kernel only then.  The
    reference builds the two graphs from two mesh files (``loaders/everyday_deform.py:48-72``): same
    triangles in the shipped data, but nothing enforces it."""
    a, b = rest.edge_index, deff.edge_index
    same = a.shape == b.shape and (a.data_ptr() == b.data_ptr() or
                                   (not a.is_cuda and not b.is_cuda and bool(torch.equal(a, b))))
    if a.is_cuda or b.is_cuda:
        same = a.shape == b.shape and a.data_ptr() == b.data_ptr()     # never read device memory here
    # the mark names the tensors it was decided on (address + version): train._edges_known_equal re-checks them
    deff.__dict__["_dc_edges_equal"] = (bool(same), b.data_ptr(), b._version, a.shape)


def to_batches(collated, device=None) -> Tuple[Batch, Batch, Batch]:
    """``train.py:36-44``: three ``Batch.from_data_list`` + ``.to(device)``."""
    _, rests, defs, _, rigids = collated
    out = tuple(Batch.from_data_list(list(g)) for g in (rests, defs, rigids))
    mark_edge_equality(out[0], out[1])
    if device is not None:
        out = tuple(b.to(device) for b in out)
    return out


class TopologyCache:
    """Per-topology cache of sorted adjacencies.

    The reference keeps one rest mesh per object (``loaders/everyday_deform.py:29-30``), so the
    same ``edge_index`` CONTENT comes back in fresh tensors batch after batch.  ``get`` hashes the
    tensor's content on the device (``dc_hash_i64``, one small kernel + an 8-byte read-back on the
    CURRENT stream - meant for the loader's side stream, not the training stream), returns the
    ``GraphIndex`` built for an earlier tensor with the same (content hash, shape, N, flags) or
    builds one, and registers it as the adjacency of THIS tensor so that the conv layers
    (``graph.graph_index``, keyed on the tensor's address) find it.  64-bit hash: a collision
    between two different topologies of identical shape is possible in principle (2^-64 per pair).
    """

    def __init__(self, max_entries: int = 64):
        from collections import OrderedDict
        self._d = OrderedDict()
        self.max_entries = max_entries
        self.hits = self.misses = 0

    def get(self, edge_index: torch.Tensor, num_nodes: int, *, self_loops: bool = False,
            normalize: bool = True, segments=None):
        from . import graph
        key = (graph.content_hash(edge_index), tuple(edge_index.shape), int(num_nodes),
               bool(self_loops), bool(normalize), edge_index.device.index)
        g = self._d.get(key)
        if g is None:
            self.misses += 1
            g = graph.GraphIndex(edge_index, num_nodes, self_loops=self_loops, normalize=normalize,
                                 segments=segments)
            self._d[key] = g
            while len(self._d) > self.max_entries:
                self._d.popitem(last=False)
        else:
            self.hits += 1
            self._d.move_to_end(key)
        graph.register(edge_index, g)
        return g


def prepare_for(model, cache: "TopologyCache | None" = None):
    """``prepare`` callback for ``PrefetchLoader``: for the freshly uploaded ``(rest, deff, rig)``
    batches build (or fetch from ``cache``) the sorted adjacency each branch's first conv layer
    will ask for and, for TAGConv, the hop slab of its no-grad input (``ops.precompute_input_hops``)
    - on the loader's stream, while the previous batch trains."""
    from . import ops

    def prepare(batches):
        rest, _, rig = batches
        built = []
        if getattr(model, "_mergeable", None) is not None and model._mergeable(rest.x, rig.x):
            # merged encoder path (ContactEncoder.merge_branches): ONE adjacency over both graphs, whose row windows
            # are what the first layers will ask for (the per-topology cache is per edge_index and does not apply)
            for g in model.topology(rest, rig):
                built.append(g)
                for i, (convs, b) in enumerate(((model.conv_layers_resting, rest), (model.conv_layers_rigid, rig))):
                    if hasattr(convs[0], "K"):
                        # (the window registered for this edge_index: the object the first layer will find)
                        ops.precompute_input_hops(convs[0].graph(b.edge_index, b.x.size(0)), b.x, convs[0].K)
            return built
        for convs, b in ((getattr(model, "conv_layers_resting", []), rest),
                         (getattr(model, "conv_layers_rigid", []), rig)):
            if not len(convs) or not hasattr(convs[0], "graph_flags"):
                continue
            conv, n = convs[0], b.x.size(0)
            seg = b.segments() if callable(getattr(b, "segments", None)) else None
            g = cache.get(b.edge_index, n, segments=seg, **conv.graph_flags()) if cache is not None \
                else conv.graph(b.edge_index, n, segments=seg)
            if hasattr(conv, "K") and getattr(conv, "supports_fused_relu", False):
                ops.precompute_input_hops(g, b.x, conv.K)
            built.append(g)
        return built

    return prepare


class PrefetchLoader:
    """Background batch assembly + overlapped upload.

    The reference loads with ``DataLoader(num_workers=0)`` and builds the three PyG batches and
    their ``.to(device)`` copies on the training thread (``train.py:25-44``), so every step waits
    for the per-sample Python work (mesh reading, ``mesh_to_graph``, collate).  Here a worker
    thread runs ``iterate_batches`` + ``Batch.from_data_list`` ``depth`` batches ahead and packs every
    tensor of a batch triple into ONE pinned staging buffer; the consumer uploads batch i+1 with one copy on
    a side HIP stream while batch i trains (the device tensors are views into that buffer);
    ``__iter__`` yields ``(collated, (rest, deff, rig))`` with the batches already on ``device``.
    Order and contents equal ``iterate_batches`` + ``to_batches``.
    """

    def __init__(self, dataset, batch_size: int, device=None, shuffle: bool = False, seed: int = 0,
                 drop_last: bool = False, depth: int = 2, prepare=None):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self.shuffle, self.seed, self.drop_last, self.depth = shuffle, seed, drop_last, max(1, depth)
        #: optional ``prepare((rest, deff, rig)) -> [GraphIndex]`` run on the upload stream right
        #: after the copies (``prepare_for(model, TopologyCache())``): the per-batch topology work
        #: (sorted adjacency, gcn_norm, first-layer hop slabs) leaves the training stream
        self.prepare = prepare

    #: pinned staging buffers in rotation: the worker may be ``depth`` queued + 1 in assembly ahead of the consumer,
    #: which has one upload in flight - a slot is refilled only after the event of its last upload has completed
    _EXTRA_SLOTS = 3

    @staticmethod
    def _layout(host):
        """Where every tensor of the three batches lives in one packed buffer (256-byte aligned)."""
        lay, off = [], 0
        for bi, b in enumerate(host):
            for k, v in b.__dict__.items():
                if isinstance(v, torch.Tensor):
                    nb = v.numel() * v.element_size()
                    lay.append((bi, k, off, nb, v.dtype, tuple(v.shape)))
                    off += (nb + 255) // 256 * 256
        return lay, off

    def _produce(self, q, stop):
        import queue
        threads = torch.get_num_threads()

        def put(item) -> bool:                  # False: the consumer has gone away (early break / close)
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False
        try:
            pin = self.device is not None and torch.device(self.device).type == "cuda"
            if pin:
                # batches are a few hundred KB of small tensors: on a many-core host the intra-op thread pool costs
                # more than it saves (measured on a 256-thread box: 10.4 ms per batch of 4 vs 2.5 ms single-threaded);
                # process-wide while this loader runs, restored when its worker ends
                torch.set_num_threads(1)
                slots = [{"buf": None, "event": None} for _ in range(self.depth + self._EXTRA_SLOTS)]
            for i, collated in enumerate(iterate_batches(self.dataset, self.batch_size, self.shuffle, self.seed,
                                                         self.drop_last)):
                host = to_batches(collated, None)
                pack = None
                if pin:
                    # ONE pinned buffer per batch triple and one host-to-device copy, instead of a pinned allocation
                    # and a copy per tensor (~45 of them)
                    slot = slots[i % len(slots)]
                    if slot["event"] is not None:
                        slot["event"].synchronize()
                        slot["event"] = None
                    lay, total = self._layout(host)
                    if slot["buf"] is None or slot["buf"].numel() < total:
                        slot["buf"] = torch.empty(max(total, 1) * 3 // 2, dtype=torch.uint8, pin_memory=True)
                    for bi, k, off, nb, dtype, shape in lay:
                        if nb:
                            slot["buf"][off:off + nb].view(dtype).view(shape).copy_(host[bi].__dict__[k])
                    pack = (slot, lay, total)
                if not put((collated, host, pack)):
                    return
            put(None)
        except BaseException as e:          # surface worker errors in the consumer
            put(e)
        finally:
            torch.set_num_threads(threads)      # the setting is process-wide: give it back when the epoch is over

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        worker = threading.Thread(target=self._produce, args=(q, stop), daemon=True)
        worker.start()
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        side = torch.cuda.Stream(device=self.device) if cuda else None

        def upload(item):
            collated, host, pack = item
            if self.device is None:
                return collated, host, None
            if not cuda:
                return collated, tuple(b.to(self.device) for b in host), None
            with torch.cuda.stream(side):
                slot, lay, total = pack
                dbuf = torch.empty(max(total, 1), dtype=torch.uint8, device=self.device)
                dbuf[:total].copy_(slot["buf"][:total], non_blocking=True)
                for bi, k, off, nb, dtype, shape in lay:       # device tensors = views into the uploaded buffer
                    host[bi].__dict__[k] = dbuf[off:off + nb].view(dtype).view(shape)
                dev = host
                ev = torch.cuda.Event()
                ev.record(side)
                slot["event"] = ev                             # the worker waits for it before refilling the slot
                built = self.prepare(dev) if self.prepare is not None else []
                ev = torch.cuda.Event()
                ev.record(side)
            return collated, dev, (ev, built)

        def get():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            return item

        try:
            nxt = get()
            pending = upload(nxt) if nxt is not None else None
            while pending is not None:
                collated, dev, ev = pending
                nxt = get()
                pending = upload(nxt) if nxt is not None else None      # next upload overlaps this step
                if ev is not None:
                    ev, built = ev
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for g in built or []:
                        g.record_stream(torch.cuda.current_stream(self.device))
                    for b in dev:                                    # all views of one buffer: one record is enough
                        for t in b.__dict__.values():
                            if isinstance(t, torch.Tensor) and t.is_cuda:
                                t.record_stream(torch.cuda.current_stream(self.device))
                                break
                yield collated, dev
        finally:                               # also on an early `break` / close of the generator: no worker left behind
            stop.set()
            worker.join()
