"""`ContactEncoder.prepare_weights_ahead`: the layers' weight preparation launched on a helper stream when a branch starts
(`ops.prepare_tag_weights`) instead of inline between the layers - same kernels, same operands, so the step must be
bit-identical with it on and off, eagerly and replayed from a hipGraph, and nothing may be left in the stash.
Reference: the encoder loops of /root/reference/models/model.py:69-78."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _steps(ahead: bool, overlap: bool, graphed: bool, steps: int = 4):
    from deformcontact_amd import dp, ops, synth
    from deformcontact_amd.graphnet import ContactEncoder
    rest, _, rig = (b.to(DEV) for b in synth.make_batch(3, soft_vertices=300, sphere_resolution=8))
    gen = torch.Generator(device=DEV).manual_seed(5)
    g_s = torch.randn(rest.x.shape[0], 256, device=DEV, generator=gen)
    g_r = torch.randn(rig.x.shape[0], 256, device=DEV, generator=gen)
    torch.manual_seed(0)
    enc = ContactEncoder([21, 25], 256).to(DEV)
    enc.overlap_branches, enc.prepare_weights_ahead = overlap, ahead
    bucket = dp.GradBucket(enc.parameters(), direct=True)
    opt = dp.FlatAdam(bucket, lr=1e-3, zero_grad_in_step=True)
    outs = []

    def step():
        a, b = enc(rest, rig)
        torch.autograd.backward([a, b], [g_s, g_r])
        bucket.all_reduce_mean()
        opt.step()
        return a, b

    if graphed:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                        # allocator / cache warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            a, b = step()
        for _ in range(steps - 1):
            gr.replay()
        outs = [a.clone(), b.clone()]
    else:
        for _ in range(steps):
            a, b = step()
        outs = [a.clone(), b.clone()]
    torch.cuda.synchronize()
    assert not ops._PREPARED, "prepared weight images left behind"
    return outs + [opt.flat_param.clone()]


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("graphed", [False, True])
def test_config1_weight_preparation_ahead_is_bit_identical(overlap, graphed):
    ref = _steps(False, overlap, graphed)
    got = _steps(True, overlap, graphed)
    for r, g in zip(ref, got):
        assert torch.equal(r, g)


def test_prepared_entries_are_consumed_or_discarded():
    """A layer whose weights were prepared with the wrong `want_t` (input gradient not expected, then needed) falls back
    to its inline preparation; the entry is gone either way."""
    from deformcontact_amd import ops
    import deformcontact_amd as dc
    from tests.helpers import random_multigraph
    n = 400
    ei = torch.from_numpy(random_multigraph(n, 2400, 3)).to(DEV)
    conv = dc.nn.TAGConv(256, 256).to(DEV)
    x = torch.randn(n, 256, device=DEV, requires_grad=True)
    ref = conv(x, ei)
    gref, = torch.autograd.grad(ref.sum(), x)
    helper = torch.cuda.Stream()
    helper.wait_stream(torch.cuda.current_stream())
    keys = ops.prepare_tag_weights([([lin.weight for lin in conv.lins], n, False)], helper)
    assert len(keys) == 1 and ops._PREPARED
    out = conv(x, ei)
    torch.cuda.current_stream().wait_stream(helper)
    assert not ops._PREPARED
    g, = torch.autograd.grad(out.sum(), x)
    assert torch.equal(out, ref) and torch.equal(g, gref)
    keys = ops.prepare_tag_weights([([lin.weight for lin in conv.lins], n, True)], helper)
    out2 = conv(x, ei)
    torch.cuda.current_stream().wait_stream(helper)
    g2, = torch.autograd.grad(out2.sum(), x)
    assert torch.equal(out2, ref) and torch.equal(g2, gref) and not ops._PREPARED
    keys = ops.prepare_tag_weights([([lin.weight for lin in conv.lins], n, True)], helper)
    ops.discard_prepared(keys)
    torch.cuda.current_stream().wait_stream(helper)
    assert not ops._PREPARED
