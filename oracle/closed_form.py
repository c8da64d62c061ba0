"""Independent float64 dense closed forms of the three conv operators.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  These do not share a line of
arithmetic with ``oracle/pyg_ref.py`` (dense adjacency matrices, numpy float64,
no gather/scatter), so agreement between the two pins the restatement against
transcription slips.  Semantics follow SURVEY.md section 8(a) rows a3-a7 (PyG
2.5.2 ``tag_conv.py`` / ``gcn_conv.py`` / ``gat_conv.py``; call sites
``/root/reference/models/model.py:45,49,71,77``).
"""
from __future__ import annotations

import numpy as np


def dense_adjacency(edge_index: np.ndarray, n: int) -> np.ndarray:
    """``A[i, j]`` = number of edges ``j -> i`` (row = ``edge_index[1]`` target)."""
    a = np.zeros((n, n), dtype=np.float64)
    np.add.at(a, (edge_index[1], edge_index[0]), 1.0)
    return a


def sym_norm(a: np.ndarray) -> np.ndarray:
    """``A_hat = D^-1/2 A D^-1/2`` with ``D`` = in-degree (row sums), ``0`` for
    isolated targets."""
    deg = a.sum(axis=1)
    with np.errstate(divide="ignore"):
        dis = np.where(deg > 0, deg ** -0.5, 0.0)
    return dis[:, None] * a * dis[None, :]


def tagconv(x, edge_index, weights, bias):
    """``sum_k A_hat^k X W_k^T + b`` (no self loops added)."""
    x = np.asarray(x, dtype=np.float64)
    a_hat = sym_norm(dense_adjacency(np.asarray(edge_index), x.shape[0]))
    out = x @ np.asarray(weights[0], dtype=np.float64).T
    xk = x
    for w in weights[1:]:
        xk = a_hat @ xk
        out = out + xk @ np.asarray(w, dtype=np.float64).T
    return out + np.asarray(bias, dtype=np.float64)


def gcnconv(x, edge_index, weight, bias):
    """``A_hat' (X W^T) + b`` where ``A'`` = ``A`` with its diagonal replaced by 1."""
    x = np.asarray(x, dtype=np.float64)
    a = dense_adjacency(np.asarray(edge_index), x.shape[0])
    np.fill_diagonal(a, 1.0)
    return sym_norm(a) @ (x @ np.asarray(weight, dtype=np.float64).T) + np.asarray(bias, np.float64)


def gatconv(x, edge_index, weight, att_src, att_dst, bias, slope=0.2):
    """heads=1 GAT: per-target softmax over incoming edges (multi-edges counted),
    self loops replaced by exactly one per node."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    a = dense_adjacency(np.asarray(edge_index), n)
    np.fill_diagonal(a, 1.0)
    h = x @ np.asarray(weight, dtype=np.float64).T
    a_s = h @ np.asarray(att_src, dtype=np.float64).reshape(-1)
    a_d = h @ np.asarray(att_dst, dtype=np.float64).reshape(-1)
    e = a_d[:, None] + a_s[None, :]                       # e[i, j] for edge j -> i
    e = np.where(e > 0, e, slope * e)
    e = np.where(a > 0, e, -np.inf)
    e = e - e.max(axis=1, keepdims=True)
    p = a * np.exp(e)                                     # multiplicity-weighted
    p = p / (p.sum(axis=1, keepdims=True) + 1e-16)
    return p @ h + np.asarray(bias, dtype=np.float64)
