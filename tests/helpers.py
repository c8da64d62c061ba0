"""Shared test helpers (tests may import oracle/; the product may not)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    """max |a-b| / max |b|  -- the tolerance north_star states is 1e-5 on this."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


class G:
    """duck-typed graph batch"""

    def __init__(self, x, edge_index, pos=None):
        self.x, self.edge_index, self.pos = x, edge_index, pos

    def clone(self):
        return G(self.x.clone(), self.edge_index.clone(),
                 None if self.pos is None else self.pos.clone())


def golden_graphs(z, device="cpu"):
    t = lambda k, dt=None: torch.from_numpy(z[k]).to(device)
    rest = G(t("rest_x"), t("rest_edge_index"), t("rest_pos"))
    rig = G(t("rig_x"), t("rig_edge_index"), t("rig_pos"))
    return rest, rig


def random_multigraph(n, e, seed, self_loops=True, isolated=True):
    """edge_index with duplicate edges, self loops and zero-in-degree nodes."""
    rng = np.random.default_rng(seed)
    hi = n - (max(1, n // 10) if isolated else 0)      # last nodes never a target
    src = rng.integers(0, n, e)
    dst = rng.integers(0, max(hi, 1), e)
    if e >= 8:
        src[:4] = src[4:8]                              # duplicates
        dst[:4] = dst[4:8]
        if self_loops:
            src[8:12] = dst[8:12]
    return np.stack([src, dst]).astype(np.int64)
