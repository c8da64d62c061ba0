"""Tiny closed triangle meshes for golden fixtures (TEST INFRASTRUCTURE).

Duck-typed like the Open3D meshes the reference consumes
(``/root/reference/utils/graph_utils.py:8-9`` reads ``.vertices`` / ``.triangles``).
"""
from __future__ import annotations

import numpy as np


class Mesh:
    def __init__(self, vertices, triangles):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.triangles = np.asarray(triangles, dtype=np.int32)

    def translate(self, t):
        self.vertices = self.vertices + np.asarray(t, dtype=np.float64)
        return self


def _subdivide(v, t):
    v = [tuple(p) for p in v]
    cache = {}

    def mid(a, b):
        key = (min(a, b), max(a, b))
        if key not in cache:
            m = (np.asarray(v[a]) + np.asarray(v[b])) / 2.0
            v.append(tuple(m / np.linalg.norm(m)))
            cache[key] = len(v) - 1
        return cache[key]

    out = []
    for a, b, c in t:
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
    return np.asarray(v), np.asarray(out)


def octasphere(levels: int, radius: float = 0.15, squash=(1.0, 0.8, 0.6)) -> Mesh:
    """Subdivided octahedron: V = 4^levels*4+2, T = 8*4^levels, consistently oriented."""
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    t = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4],
                  [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    for _ in range(levels):
        v, t = _subdivide(v, t)
    return Mesh(v * radius * np.asarray(squash), t)


def uv_sphere(radius: float, resolution: int) -> Mesh:
    """Same vertex/triangle enumeration as Open3D ``TriangleMesh.create_sphere``
    [3P-memory] (``/root/reference/loaders/common.py:26``): two poles, then
    ``resolution-1`` rings of ``2*resolution`` vertices."""
    r2 = 2 * resolution
    v = [(0.0, 0.0, radius), (0.0, 0.0, -radius)]
    step = np.pi / resolution
    for i in range(1, resolution):
        a = step * i
        for j in range(r2):
            th = step * j
            v.append((radius * np.sin(a) * np.cos(th), radius * np.sin(a) * np.sin(th),
                      radius * np.cos(a)))
    t = []
    for j in range(r2):
        j1 = (j + 1) % r2
        base = 2
        t.append((0, base + j, base + j1))
        base = 2 + r2 * (resolution - 2)
        t.append((1, base + j1, base + j))
    for i in range(1, resolution - 1):
        b1 = 2 + r2 * (i - 1)
        b2 = b1 + r2
        for j in range(r2):
            j1 = (j + 1) % r2
            t.append((b2 + j, b1 + j1, b1 + j))
            t.append((b2 + j, b2 + j1, b1 + j1))
    return Mesh(v, t)
