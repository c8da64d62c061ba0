"""`ops.DW_POSITION`: in a wide layer's backward dW is issued between the transposed chain and dX on every stream but the
encoder's side stream, where it stays behind dX (the two branches' dW kernels then do not run side by side).  Same kernels, same operands - every mode
must give the same bits, eagerly and replayed from a hipGraph.  Reference: the encoder loops of
/root/reference/models/model.py:69-78 (autograd orders nothing between a layer's weight and input gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _steps(mode: dict, overlap: bool, graphed: bool, steps: int = 3):
    from deformcontact_amd import dp, ops, synth
    from deformcontact_amd.graphnet import ContactEncoder
    keep = dict(ops.DW_POSITION)
    ops.DW_POSITION.update(mode)
    try:
        rest, _, rig = (b.to(DEV) for b in synth.make_batch(3, soft_vertices=300, sphere_resolution=8))
        gen = torch.Generator(device=DEV).manual_seed(5)
        g_s = torch.randn(rest.x.shape[0], 256, device=DEV, generator=gen)
        g_r = torch.randn(rig.x.shape[0], 256, device=DEV, generator=gen)
        torch.manual_seed(0)
        enc = ContactEncoder([21, 25], 256).to(DEV)
        enc.overlap_branches = overlap
        bucket = dp.GradBucket(enc.parameters(), direct=True)
        opt = dp.FlatAdam(bucket, lr=1e-3, zero_grad_in_step=True)

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
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                a, b = step()
            for _ in range(steps - 1):
                gr.replay()
        else:
            for _ in range(steps):
                a, b = step()
        torch.cuda.synchronize()
        return [a.clone(), b.clone(), opt.flat_param.clone()]
    finally:
        ops.DW_POSITION.update(keep)


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("graphed", [False, True])
def test_config1_weight_gradient_first_or_last_is_bit_identical(overlap, graphed):
    ref = _steps({"unlisted": "last", "listed": "last"}, overlap, graphed)           # the order of rounds 1 - 4
    for mode in ({"unlisted": "mid", "listed": "last"}, {"unlisted": "first", "listed": "first"},
                 {"unlisted": "first", "listed": "mid"}):
        for r, g in zip(ref, _steps(mode, overlap, graphed)):
            assert torch.equal(r, g), mode


def test_side_stream_is_registered_and_keeps_the_old_order():
    from deformcontact_amd import ops
    from deformcontact_amd.graphnet import ContactEncoder
    side = ContactEncoder._side_stream(DEV)
    assert side.cuda_stream in ops.DW_LAST_STREAMS and ops.DW_POSITION == {"unlisted": "mid", "listed": "last"}
    with torch.cuda.stream(side):
        assert ops._dw_position(DEV) == "last"
    assert ops._dw_position(DEV) == "mid"
