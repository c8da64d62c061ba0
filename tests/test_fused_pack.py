"""``dc_spmm_f32_pack``: the packing pass and the first hop of a narrow layer's own input in ONE launch - the hop slab must
equal, bit for bit, what ``dc_tag_pack_input`` + ``dc_spmm_f32`` write (the first TAGConv of either branch,
/root/reference/models/model.py:71,77 with the raw ``graph.x``: F_in = 21 / 25, K = 3)."""
import numpy as np
import pytest
import torch

from deformcontact_amd import ops, synth
from deformcontact_amd.graph import GraphIndex
from tests.helpers import random_multigraph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,e,fi,k", [(300, 2000, 21, 3), (1000, 9000, 25, 3), (77, 0, 5, 1), (1, 0, 32, 2),
                                      (4099, 30011, 17, 3), (513, 4000, 8, 2)])
def test_fused_pack_and_first_hop_equal_two_launches_bitwise(n, e, fi, k, monkeypatch):
    ei = torch.from_numpy(random_multigraph(n, e, n + fi)).to(DEV) if e else torch.zeros((2, 0), dtype=torch.int64, device=DEV)
    g = GraphIndex(ei, n)
    torch.manual_seed(n)
    x = torch.randn(n, fi, device=DEV)
    slabs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_PACK", fused)
        slab, rowmax = ops._build_input_slab(g, x, k, False)
        assert rowmax is None
        slabs.append(slab.detach().cpu().numpy().copy())
    concat, width, wpad = ops.tag_slab_geometry(fi, k)
    assert np.array_equal(slabs[0][:, :wpad], slabs[1][:, :wpad])
    assert np.array_equal(slabs[0][:, :fi], x.cpu().numpy()) and np.all(slabs[0][:, width:wpad] == 0)


def test_encoder_first_layer_uses_the_fused_launch_and_matches(monkeypatch):
    from deformcontact_amd.graph import clear_cache
    from deformcontact_amd.graphnet import ContactEncoder
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(4))
    torch.manual_seed(0)
    enc = ContactEncoder([rest.x.size(1), rig.x.size(1)], 256, 2).to(DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_PACK", fused)
        clear_cache()
        for p in enc.parameters():
            p.grad = None
        a, b = enc(rest, rig)
        (a.square().sum() + b.sum()).backward()
        outs.append([a.detach().cpu().numpy(), b.detach().cpu().numpy()] + [p.grad.cpu().numpy() for p in enc.parameters()])
    for u, v in zip(*outs):
        assert np.array_equal(u, v)

