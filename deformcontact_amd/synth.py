"""Synthetic "everyday-deform shape" inputs (the dataset itself is a Google-Drive
download that is absent, ``/root/reference/README.md:36-37``).

Per sample:

* **soft** mesh: closed genus-0 triangle mesh, V=1024, T=2V-4=2044, directed
  E=3T=6132 (jittered Fibonacci sphere -> convex hull, outward orientation,
  smooth radial noise, extents ~0.3 m); features ``to_log_freq(pos)`` (21);
* **rigid** mesh: UV sphere with Open3D ``create_sphere(radius=0.05,
  resolution=20)`` enumeration: V=762, T=1520, E=4560 (poles have in-degree 40);
  features ``[force_dir(3), force(1), to_log_freq(pos)(21)]`` (25);
* **target**: rest mesh + Gaussian bump displacement around the contact point.

Seeds: geometry ``1234 + sample_idx``.  Also a dense radius-graph generator for
the 100k-node stress config (BASELINE.json configs[4]).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from .data import Batch, Data
from .features import feature_rigid, mesh_to_graph


class TriMesh:
    def __init__(self, vertices, triangles):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.triangles = np.asarray(triangles, dtype=np.int64)


def fibonacci_sphere(n: int) -> np.ndarray:
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    th = np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)


def soft_mesh(num_vertices: int = 1024, seed: int = 1234, extent: float = 0.15) -> TriMesh:
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(seed)
    p = fibonacci_sphere(num_vertices) + rng.normal(0, 0.15 / np.sqrt(num_vertices), (num_vertices, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    hull = ConvexHull(p)
    tri = hull.simplices.copy()
    # consistent outward orientation: flip triangles whose normal points inward
    n = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    flip = (n * p[tri[:, 0]]).sum(1) < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    # smooth radial noise (low-order harmonics) + anisotropic extents
    a = rng.normal(0, 0.08, 6)
    r = 1 + a[0] * p[:, 0] + a[1] * p[:, 1] + a[2] * p[:, 2] + a[3] * p[:, 0] * p[:, 1] \
        + a[4] * p[:, 1] * p[:, 2] + a[5] * p[:, 0] * p[:, 2]
    scale = extent * np.array([1.0, 0.8, 0.6]) * rng.uniform(0.9, 1.1, 3)
    return TriMesh(p * r[:, None] * scale, tri)


def uv_sphere(radius: float = 0.05, resolution: int = 20) -> TriMesh:
    """Open3D ``TriangleMesh.create_sphere`` vertex/triangle enumeration [3P-memory]
    (used by ``/root/reference/loaders/common.py:26``)."""
    r2 = 2 * resolution
    step = np.pi / resolution
    i = np.arange(1, resolution)[:, None] * step
    j = np.arange(r2)[None, :] * step
    ring = np.stack([np.sin(i) * np.cos(j), np.sin(i) * np.sin(j),
                     np.cos(i) * np.ones_like(j)], -1).reshape(-1, 3)
    v = np.concatenate([[[0, 0, 1.0], [0, 0, -1.0]], ring], 0) * radius
    jj = np.arange(r2)
    j1 = (jj + 1) % r2
    b_last = 2 + r2 * (resolution - 2)
    caps = np.stack([np.stack([np.zeros(r2, int), 2 + jj, 2 + j1], 1),
                     np.stack([np.ones(r2, int), b_last + j1, b_last + jj], 1)], 1).reshape(-1, 3)
    body = []
    for k in range(1, resolution - 1):
        b1 = 2 + r2 * (k - 1)
        b2 = b1 + r2
        body.append(np.stack([np.stack([b2 + jj, b1 + j1, b1 + jj], 1),
                              np.stack([b2 + jj, b2 + j1, b1 + j1], 1)], 1).reshape(-1, 3))
    return TriMesh(v, np.concatenate([caps] + body, 0))


def make_sample(idx: int, soft_vertices: int = 1024, sphere_resolution: int = 20
                ) -> Tuple[Data, Data, Data]:
    """(soft_rest, soft_def, rigid) graphs of synthetic sample ``idx``."""
    seed = 1234 + idx
    rng = np.random.default_rng(seed)
    soft = soft_mesh(soft_vertices, seed)
    contact = soft.vertices[rng.integers(0, len(soft.vertices))]
    rigid = uv_sphere(0.05, sphere_resolution)
    rigid.vertices = rigid.vertices + contact
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    force = float(rng.uniform(0, 1))
    d2 = ((soft.vertices - contact) ** 2).sum(1, keepdims=True)
    soft_def = TriMesh(soft.vertices + 0.03 * force * np.exp(-d2 / (2 * 0.08 ** 2)) * direction,
                       soft.triangles)
    g_rest, g_def, g_rig = mesh_to_graph(soft), mesh_to_graph(soft_def), mesh_to_graph(rigid)
    g_rig.x = feature_rigid(torch.tensor(direction, dtype=torch.float32), force, g_rig.x)
    return g_rest, g_def, g_rig


def make_batch(batch_size: int = 32, first_idx: int = 0, soft_vertices: int = 1024,
               sphere_resolution: int = 20) -> Tuple[Batch, Batch, Batch]:
    """Batched (soft_rest, soft_def, rigid) as ``train.py:36-38`` builds them."""
    triples = [make_sample(first_idx + i, soft_vertices, sphere_resolution)
               for i in range(batch_size)]
    out = (Batch.from_data_list([t[0] for t in triples]),
           Batch.from_data_list([t[1] for t in triples]),
           Batch.from_data_list([t[2] for t in triples]))
    from .loaders import mark_edge_equality
    mark_edge_equality(out[0], out[1])                 # host-side: selects the fused loss (train.losses)
    return out


def radius_graph_points(num_points: int = 100_000, radius: float = 0.02,
                        max_num_neighbors: int = 32, seed: int = 7,
                        blob_fraction: float = 0.3) -> Tuple[torch.Tensor, torch.Tensor]:
    """Point cloud with a dense blob + ``radius_graph(loop=False)``-style edges
    (``/root/reference/utils/pointcloud_utils.py:10``; neighbour cap as PyG's default
    ``max_num_neighbors=32`` [3P-memory]).  Returns ``(pos [N,3] f32, edge_index [2,E] i64)``;
    edges are (neighbour -> centre), grouped by centre."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    nb = int(num_points * blob_fraction)
    pts = np.concatenate([rng.uniform(0, 1, (num_points - nb, 3)),
                          0.5 + rng.normal(0, 0.03, (nb, 3))], 0).astype(np.float32)
    tree = cKDTree(pts)
    dist, nbr = tree.query(pts, k=max_num_neighbors + 1, distance_upper_bound=radius, workers=-1)
    centre = np.repeat(np.arange(num_points), max_num_neighbors + 1)
    nbr = nbr.reshape(-1)
    ok = (nbr < num_points) & (nbr != centre)
    ei = np.stack([nbr[ok], centre[ok]], 0).astype(np.int64)
    return torch.from_numpy(pts), torch.from_numpy(ei)
