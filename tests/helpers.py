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


def row_rel_err(a, b):
    """max over rows i of  max_j |a_ij - b_ij| / max_j |b_ij|  (rows of b that are all zero are
    compared absolutely against the global scale): stricter than ``rel_err`` for rows whose
    magnitude is far below the tensor's maximum."""
    a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
    b = np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    scale = np.abs(b).max(axis=1)
    scale = np.where(scale > 0, scale, max(np.abs(b).max(), 1e-30))
    return float((np.abs(a - b).max(axis=1) / scale).max())


#: every parity comparison of a test run: (test id, tensor, HIP-vs-fp32-oracle distance, did the float64 widening fire,
#: HIP-vs-float64, oracle-vs-float64) - tests/conftest.py writes it to gpurun_out/parity_distances.json at session end
#: ("nothing records how often the widening fires or the worst HIP-vs-fp32-oracle distance")
PARITY_LOG = []


def record_parity(name, d, widened=False, e_h=None, e_o=None, tol=1e-5, metric="rel_err", special=None):
    """``special``: why this comparison is not on north_star's 1e-5 scale (e.g. a gradient that is mathematically zero,
    parameters after several Adam steps) - listed apart and kept out of the summary's worst-case figures."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    PARITY_LOG.append({"test": test, "tensor": name, "special": special,
                       "hip_vs_fp32_oracle": None if d is None else float(d),
                       "tol": tol, "metric": metric, "float64_widening_fired": bool(widened),
                       "hip_vs_float64": None if e_h is None else float(e_h),
                       "fp32_oracle_vs_float64": None if e_o is None else float(e_o)})


def three_way(d, e_h_fn, e_o_fn, tol=1e-5, name="", metric="rel_err", special=None, special_bound=None):
    """The parity rule on already computed distances: ``d`` (HIP vs fp32 oracle) < tol, or - evaluated lazily - HIP
    within ``tol`` of the float64 truth (round 5: the ``2 x the oracle's own distance`` allowance is gone; every
    widened comparison of round 4 was within 4.4e-6 of float64).  Only a comparison registered ``special`` (a stated
    reason why it is not on north_star's scale) may pass a wider ``special_bound(e_o)``.  Records the outcome."""
    if d < tol:
        record_parity(name, d, tol=tol, metric=metric, special=special)
        return d
    e_h, e_o = e_h_fn(), e_o_fn()
    record_parity(name, d, True, e_h, e_o, tol=tol, metric=metric, special=special)
    bound = tol if special is None else max(tol, special_bound(e_o))
    assert e_h <= bound, (
        f"{name}: HIP vs fp32 oracle {d:.2e} >= {tol:g}, and vs the float64 truth the HIP path is off by "
        f"{e_h:.2e} > {bound:.2e} (the fp32 oracle is off by {e_o:.2e})")
    return d


def assert_parity(got, ref32, truth64=None, tol=1e-5, name="", metric=None, special=None, special_bound=None):
    """north_star's bar: ``metric(got, fp32 oracle) < tol`` (1e-5).  Where two correct fp32 evaluations legitimately
    differ by more (gradients through several layers: different but equally valid summation orders), the HIP result has
    to be within ``tol`` of a float64 evaluation of the same maths - nothing wider, unless the comparison is registered
    ``special`` with its own stated bound (``three_way``).  Returns the HIP-vs-oracle distance."""
    metric = metric or rel_err
    d = metric(got, ref32)
    if d >= tol:
        assert truth64 is not None, f"{name}: {d:.2e} >= {tol:g} vs the fp32 oracle and no float64 truth given"
    return three_way(d, lambda: metric(got, truth64), lambda: metric(ref32, truth64), tol=tol, name=name,
                     metric=metric.__name__, special=special, special_bound=special_bound)


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
