"""Device-side node reordering (dc_order.hip; BASELINE.json configs[4], /root/reference/utils/pointcloud_utils.py:7-13):
``dc_morton_order`` (Z-order permutation without a host synchronisation), ``dc_relabel_edges``, ``dc_gather_rows``."""
import numpy as np
import pytest
import torch

from deformcontact_amd.graph import NodeOrder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _codes(pos):
    """The kernel's 30-bit codes, restated in float32 numpy."""
    p = pos.astype(np.float32)
    lo, hi = p.min(0), p.max(0)
    sc = np.float32(1.0) / np.maximum(hi - lo, np.float32(1e-30))
    t = (p - lo) * sc * np.float32(1024.0)
    q = np.clip(t, 0, 1023).astype(np.uint32)

    def spread(v):
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


@pytest.mark.parametrize("n", [1, 2, 1000, 100_003])
def test_morton_order_is_a_stable_sort_of_the_codes(n):
    rng = np.random.default_rng(n)
    pos = np.concatenate([rng.uniform(-1, 2, (n - n // 3, 3)), 0.5 + rng.normal(0, 0.01, (n // 3, 3))]).astype(np.float32)
    if n > 10:
        pos[5] = pos[3]                                  # equal points: equal codes, original order kept
    o = NodeOrder.morton(torch.from_numpy(pos).to(DEV))
    perm, inv = o.perm.cpu().numpy(), o.inv.cpu().numpy()
    assert sorted(perm.tolist()) == list(range(n)) and np.array_equal(inv[perm], np.arange(n))
    c = _codes(pos)[perm].astype(np.int64)
    assert (np.diff(c) >= 0).all()
    ties = np.diff(c) == 0
    assert (np.diff(perm)[ties] > 0).all(), "equal codes must keep their original order (stable sort)"


def test_relabel_and_row_gathers():
    n, e = 5000, 40_000
    rng = np.random.default_rng(1)
    o = NodeOrder.morton(torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).to(DEV))
    ei = torch.from_numpy(rng.integers(0, n, (2, e))).to(DEV)
    assert torch.equal(o.relabel(ei), o.inv.long()[ei])
    for dt, f in ((torch.float32, 256), (torch.bfloat16, 256), (torch.float32, 4), (torch.float32, 21)):
        x = torch.randn(n, f, device=DEV).to(dt)
        assert torch.equal(o.apply(x), x.index_select(0, o.perm.long()))
        assert torch.equal(o.undo(o.apply(x)), x)
    wide = torch.randn(n, 300, device=DEV)[:, 16:272]     # a column slice: row stride != row width
    assert torch.equal(o.apply(wide), wide.index_select(0, o.perm.long()))


def test_ordering_a_new_cloud_is_capturable():
    """No host synchronisation anywhere: Morton order + relabelling + gathers inside ONE hipGraph, replayed on new points."""
    n = 20_000
    pos = torch.rand(n, 3, device=DEV)
    x = torch.randn(n, 64, device=DEV)
    ei = torch.randint(0, n, (2, 50_000), device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        NodeOrder.morton(pos)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = NodeOrder.morton(pos)
        xa, er = o.apply(x), o.relabel(ei)
        back = o.undo(xa)
    for seed in (1, 2):
        torch.manual_seed(seed)
        pos.copy_(torch.rand(n, 3, device=DEV))
        g.replay()
        torch.cuda.synchronize()
        want = NodeOrder.morton(pos.clone())
        assert torch.equal(o.perm, want.perm) and torch.equal(xa, x.index_select(0, want.perm.long()))
        assert torch.equal(er, want.inv.long()[ei]) and torch.equal(back, x)


def test_gathers_are_differentiable_through_undo_conv_apply():
    """apply / undo went from index_select (differentiable) to a raw-pointer gather with no grad_fn, which
    silently cut the graph of `undo(conv(apply(x), relabel(edge_index)))`.  The gathers are autograd functions now; the
    gradients of the reordered computation equal those of the unordered one bit for bit (same per-row sums)."""
    from deformcontact_amd import nn as dc_nn
    n, e = 3000, 18_000
    rng = np.random.default_rng(3)
    pos = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).to(DEV)
    ei = torch.from_numpy(rng.integers(0, n, (2, e))).to(DEV)
    o = NodeOrder.morton(pos)
    torch.manual_seed(0)
    conv = dc_nn.TAGConv(64, 64).to(DEV)
    x0 = torch.randn(n, 64, device=DEV)
    gy = torch.randn(n, 64, device=DEV)
    x1 = x0.clone().requires_grad_(True)
    y1 = conv(x1, ei)
    y1.backward(gy)
    gw1 = [p.grad.clone() for p in conv.parameters()]
    conv.zero_grad()
    x2 = x0.clone().requires_grad_(True)
    y2 = o.undo(conv(o.apply(x2), o.relabel(ei)))
    assert y2.grad_fn is not None
    y2.backward(gy)
    assert torch.equal(y1, y2) and x2.grad is not None
    assert torch.equal(x1.grad, x2.grad)
    for a, p in zip(gw1, conv.parameters()):
        # dW sums over nodes in memory order, which the relabelling changes: equal up to fp32 summation order
        assert p.grad is not None and float((a - p.grad).abs().max()) <= 1e-5 * float(a.abs().max())
    # no graph under no_grad / for inputs that need none
    with torch.no_grad():
        assert o.apply(x2).grad_fn is None
    assert o.apply(x0).grad_fn is None


def test_a_bad_user_permutation_is_refused():
    good = torch.tensor([2, 0, 1], device=DEV)
    assert torch.equal(NodeOrder(good).inv.cpu(), torch.tensor([1, 2, 0], dtype=torch.int32))
    for bad in ([0, 1, 3], [0, 0, 1], [-1, 0, 1]):
        with pytest.raises(IndexError):
            NodeOrder(torch.tensor(bad, device=DEV))
