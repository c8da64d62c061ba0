"""Input-side helpers of the hot path: mesh -> graph, positional encoding, rigid features.

Vectorised counterparts of ``/root/reference/utils/graph_utils.py:7-20``
(``mesh_to_graph``), ``utils/pos_encoding.py:6-44`` (``to_log_freq``) and
``loaders/common.py:6-19`` (``_feature_rigid``).  Same outputs (edge order
included), no per-triangle Python loop.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from .data import Data


def to_log_freq(x: Tensor, n_freqs: int = 3, dim: int = 1) -> Tensor:
    """``cat[x, sin(f0 x), cos(f0 x), ..., sin(f_{n-1} x), cos(f_{n-1} x)]`` along the
    last dim with ``f = 2 ** linspace(0, n_freqs-1, n_freqs)`` (``dim`` is only read
    for the input width in the reference, the concat is always on ``-1``)."""
    freqs = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    parts = [x]
    for f in freqs:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def triangles_to_edge_index(triangles) -> Tensor:
    """Directed edges (a,b),(b,c),(c,a) per triangle, triangle-major, no dedup:
    ``[2, 3T]`` int64 contiguous - identical to the reference's list comprehension."""
    t = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    src = t.reshape(-1)                              # a b c | a b c ...
    dst = np.roll(t, -1, axis=1).reshape(-1)         # b c a | ...
    return torch.from_numpy(np.stack([src, dst], axis=0)).contiguous()


def mesh_to_graph(mesh, encode: bool = True) -> Data:
    """Duck-typed mesh (``.vertices`` [V,3], ``.triangles`` [T,3]) -> ``Data(x, edge_index, pos)``."""
    pos = torch.tensor(np.asarray(mesh.vertices), dtype=torch.float32)
    edge_index = triangles_to_edge_index(mesh.triangles)
    x = to_log_freq(pos, 3, 1) if encode else pos
    return Data(x=x, edge_index=edge_index, pos=pos)


def feature_rigid(force_vector: Tensor, force: float, pos_enc: Tensor) -> Tensor:
    """``cat[force_vector x V (3), force x V (1), pos_enc (21)]`` -> ``[V, 25]``."""
    v = pos_enc.shape[0]
    fv = force_vector.to(torch.float32).reshape(1, -1).repeat(v, 1)
    fs = torch.tensor(force, dtype=torch.float32).repeat(v, 1)
    return torch.cat([fv, fs, pos_enc], dim=1)
