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


def to_batches(collated, device=None) -> Tuple[Batch, Batch, Batch]:
    """``train.py:36-44``: three ``Batch.from_data_list`` + ``.to(device)``."""
    _, rests, defs, _, rigids = collated
    out = tuple(Batch.from_data_list(list(g)) for g in (rests, defs, rigids))
    if device is not None:
        out = tuple(b.to(device) for b in out)
    return out


class PrefetchLoader:
    """Background batch assembly + overlapped upload.

    The reference loads with ``DataLoader(num_workers=0)`` and builds the three PyG batches and
    their ``.to(device)`` copies on the training thread (``train.py:25-44``), so every step waits
    for the per-sample Python work (mesh reading, ``mesh_to_graph``, collate).  Here a worker
    thread runs ``iterate_batches`` + ``Batch.from_data_list`` ``depth`` batches ahead, pins the
    host tensors, and the consumer uploads batch i+1 on a side HIP stream while batch i trains;
    ``__iter__`` yields ``(collated, (rest, deff, rig))`` with the batches already on ``device``.
    Order and contents equal ``iterate_batches`` + ``to_batches``.
    """

    def __init__(self, dataset, batch_size: int, device=None, shuffle: bool = False, seed: int = 0,
                 drop_last: bool = False, depth: int = 2):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self.shuffle, self.seed, self.drop_last, self.depth = shuffle, seed, drop_last, max(1, depth)

    def _produce(self, q):
        try:
            pin = self.device is not None and torch.device(self.device).type == "cuda"
            for collated in iterate_batches(self.dataset, self.batch_size, self.shuffle, self.seed,
                                            self.drop_last):
                host = to_batches(collated, None)
                if pin:
                    for b in host:
                        for k in ("x", "pos", "edge_index", "batch", "ptr"):
                            t = getattr(b, k, None)
                            if isinstance(t, torch.Tensor):
                                setattr(b, k, t.pin_memory())
                q.put((collated, host))
            q.put(None)
        except BaseException as e:          # surface worker errors in the consumer
            q.put(e)

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        worker = threading.Thread(target=self._produce, args=(q,), daemon=True)
        worker.start()
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        side = torch.cuda.Stream(device=self.device) if cuda else None

        def upload(item):
            collated, host = item
            if self.device is None:
                return collated, host, None
            if not cuda:
                return collated, tuple(b.to(self.device) for b in host), None
            with torch.cuda.stream(side):
                dev = tuple(b.to(self.device, non_blocking=True) for b in host)
                ev = torch.cuda.Event()
                ev.record(side)
            return collated, dev, ev

        def get():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            return item

        nxt = get()
        pending = upload(nxt) if nxt is not None else None
        while pending is not None:
            collated, dev, ev = pending
            nxt = get()
            pending = upload(nxt) if nxt is not None else None      # next upload overlaps this step
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for b in dev:
                    for k in ("x", "pos", "edge_index", "batch", "ptr"):
                        t = getattr(b, k, None)
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(torch.cuda.current_stream(self.device))
            yield collated, dev
        worker.join()
