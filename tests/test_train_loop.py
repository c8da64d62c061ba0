"""Train-step maths (reference train.py:46-58,71-73) and the loss curve: CPU wiring test with the
oracle convs, and a GPU test that the HIP path follows the oracle's loss curve step for step."""
import json
import os

import numpy as np
import pytest
import torch

from deformcontact_amd import loaders
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model
from deformcontact_amd.train import losses, train, train_step
from oracle import pyg_ref

SMALL = dict(EVERYDAY_NETWORK, hidden_dim=32)


def _batches(n, bs):
    ds = loaders.SyntheticEverydayDataset(n, 0, soft_vertices=64, sphere_resolution=4)
    return list(loaders.iterate_batches(ds, bs))


def test_collate_fn_shapes():
    (names, rests, defs, meta, rigids), = _batches(3, 3)
    assert names == ["Box", "Cat", "Pillow"] and len(rests) == len(defs) == len(rigids) == 3
    assert meta["force_vector"].shape == (3, 3) and isinstance(meta["force"], list)
    rest, deff, rig = loaders.to_batches((names, rests, defs, meta, rigids))
    assert rest.num_graphs == 3 and rest.x.shape[1] == 21 and rig.x.shape[1] == 25
    assert torch.equal(rest.edge_index, deff.edge_index)


def test_train_step_cpu_oracle_loss_decreases(tmp_path):
    torch.manual_seed(0)
    torch.set_num_threads(1)
    model = load_model(SMALL, conv_module=pyg_ref)
    opt = torch.optim.Adam(model.parameters(), lr=4e-4)
    batch = loaders.to_batches(_batches(2, 2)[0])
    first = float(train_step(model, opt, *batch)["loss"])
    for _ in range(20):
        last = float(train_step(model, opt, *batch)["loss"])
    assert last < first
    # the loop writes a JSONL log, a checkpoint with the reference's state_dict keys and a config
    train(SMALL, device="cpu", epochs=1, num_train=4, num_val=2, batch_size=2, out_dir=str(tmp_path),
          soft_vertices=64, sphere_resolution=4, conv_module=pyg_ref)
    lines = [json.loads(l) for l in open(tmp_path / "log_rank0.jsonl")]
    assert any("validation_loss" in l for l in lines) and any("loss" in l for l in lines)
    sd = torch.load(tmp_path / "model_weights.pth")
    assert "conv_layers_resting.0.lins.3.weight" in sd and "decoder.9.bias" in sd
    assert os.path.exists(tmp_path / "config.json")


@pytest.mark.gpu
def test_config2_loss_curve_matches_oracle_on_gpu():
    """Same init, same batches, Adam(4e-4): HIP path vs CPU oracle, 8 steps."""
    torch.set_num_threads(1)
    torch.manual_seed(0)
    ref = load_model(SMALL, conv_module=pyg_ref)
    gpu = load_model(SMALL)
    gpu.load_state_dict(ref.state_dict())
    gpu = gpu.to("cuda:0")
    o_ref = torch.optim.Adam(ref.parameters(), lr=4e-4)
    o_gpu = torch.optim.Adam(gpu.parameters(), lr=4e-4)
    curve_ref, curve_gpu = [], []
    for collated in _batches(16, 2):
        b_cpu = loaders.to_batches(collated)
        b_gpu = loaders.to_batches(collated, "cuda:0")
        curve_ref.append(float(train_step(ref, o_ref, *b_cpu)["loss"]))
        curve_gpu.append(float(train_step(gpu, o_gpu, *b_gpu)["loss"]))
    assert np.allclose(curve_gpu, curve_ref, rtol=2e-4), (curve_gpu, curve_ref)
    assert np.allclose(curve_gpu[:2], curve_ref[:2], rtol=1e-5)


@pytest.mark.gpu
def test_config2_graphed_train_loop_follows_oracle_curve_with_changing_batches(tmp_path):
    """BASELINE.json configs[2]: the real loop - a NEW batch every step (new edge_index, new
    features), FlatAdam over the direct-gradient bucket, steps replayed from one hipGraph that
    contains the adjacency build - against the CPU oracle with torch.optim.Adam on the same
    batches.  Also: train() on the GPU writes its log / checkpoint and replays graphs."""
    from deformcontact_amd import dp
    from deformcontact_amd.train import GraphedTrainStep
    torch.set_num_threads(1)
    torch.manual_seed(0)
    ref = load_model(SMALL, conv_module=pyg_ref)
    gpu = load_model(SMALL)
    gpu.load_state_dict(ref.state_dict())
    gpu = gpu.to("cuda:0")
    o_ref = torch.optim.Adam(ref.parameters(), lr=4e-4)
    bucket = dp.GradBucket(gpu.parameters(), direct=True)
    o_gpu = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
    step = GraphedTrainStep(gpu, o_gpu, bucket, 1.0, eager_steps=2)
    curve_ref, curve_gpu = [], []
    for collated in _batches(20, 2):
        curve_ref.append(float(train_step(ref, o_ref, *loaders.to_batches(collated))["loss"]))
        curve_gpu.append(float(step(*loaders.to_batches(collated, "cuda:0"))["loss"]))
    assert step.replays == 8
    assert np.allclose(curve_gpu, curve_ref, rtol=2e-4), (curve_gpu, curve_ref)
    assert np.allclose(curve_gpu[:2], curve_ref[:2], rtol=1e-5)
    for a, b in zip(ref.parameters(), gpu.parameters()):
        assert np.allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=0, atol=2e-5)
    stats = {}
    train(SMALL, device="cuda:0", epochs=1, num_train=12, num_val=2, batch_size=2, out_dir=str(tmp_path),
          soft_vertices=64, sphere_resolution=4, stats=stats)
    assert stats["steps"] == 6 and stats["graph_replays"] == 4
    lines = [json.loads(l) for l in open(tmp_path / "log_rank0.jsonl")]
    assert sum("loss" in l for l in lines) == 6 and all(np.isfinite(l["loss"]) for l in lines if "loss" in l)
    stats = {}
    train(SMALL, device="cuda:0", epochs=1, num_train=6, num_val=2, batch_size=2, out_dir=str(tmp_path),
          soft_vertices=64, sphere_resolution=4, capture=False, stats=stats)   # loader-prepared topology
    assert stats["steps"] == 3 and stats["graph_replays"] == 0


@pytest.mark.gpu
def test_fused_contact_losses_match_reference_formulation_and_golden():
    """ops.contact_losses (one node pass: L1 + gradient consistency + both gradients) against the
    stock formulation of models/losses.py / nn.L1Loss in float64, the golden fixture produced by the
    reference's own losses.py, and - through train.losses - the device-side topology check."""
    from deformcontact_amd import ops
    from deformcontact_amd import train as dc_train
    from deformcontact_amd.graph import graph_index
    from deformcontact_amd.graphnet import gradient_consistency_loss
    from tests.helpers import G, load_golden, random_multigraph
    dev = "cuda:0"
    z = load_golden("gcl_loss.npz")
    keys = {k: z[k] for k in z}
    n, e = 500, 4000
    ei = torch.from_numpy(random_multigraph(n, e, 3)).to(dev)
    gen = torch.Generator().manual_seed(0)
    pred = torch.randn(n, 3, generator=gen).to(dev).requires_grad_(True)
    tgt = torch.randn(n, 3, generator=gen).to(dev)
    g = graph_index(ei, n)
    l1, gcl = ops.contact_losses(g, pred, tgt)
    (l1 + 0.7 * gcl).backward()
    p64 = pred.detach().double().cpu().requires_grad_(True)
    t64 = tgt.double().cpu()
    r_l1 = torch.nn.functional.l1_loss(p64, t64)
    r_gcl = gradient_consistency_loss(G(None, ei.cpu(), p64), G(None, ei.cpu(), t64))
    (r_l1 + 0.7 * r_gcl).backward()
    assert abs(float(l1) - float(r_l1)) <= 1e-6 * abs(float(r_l1))
    assert abs(float(gcl) - float(r_gcl)) <= 1e-6 * abs(float(r_gcl))
    assert np.abs(pred.grad.cpu().numpy() - p64.grad.numpy()).max() <= 1e-5 * np.abs(p64.grad.numpy()).max()
    # zero edge vectors (duplicates of a self loop): finite, zero-gradient terms as torch's norm
    ei0 = torch.tensor([[0, 1, 2, 2], [1, 0, 2, 2]], device=dev)
    p0 = torch.zeros(3, 3, device=dev, requires_grad=True)
    a, b = ops.contact_losses(graph_index(ei0, 3), p0, torch.zeros(3, 3, device=dev))
    (a + b).backward()
    assert float(a) == 0.0 and float(b) == 0.0 and torch.isfinite(p0.grad).all()
    # golden: the reference's GradientConsistencyLoss on a fixture
    gi = graph_index(torch.from_numpy(keys["edge_index"]).to(dev), keys["pred_pos"].shape[0])
    pp = torch.from_numpy(keys["pred_pos"]).to(dev).requires_grad_(True)
    _, gg = ops.contact_losses(gi, pp, torch.from_numpy(keys["tgt_pos"]).to(dev))
    gg.backward()
    assert abs(float(gg) - float(keys["loss"])) <= 1e-5 * abs(float(keys["loss"]))
    assert np.abs(pp.grad.cpu().numpy() - keys["grad_pred"]).max() <= 1e-5 * np.abs(keys["grad_pred"]).max()
    # through train.losses: fused == stock on a real batch (the loader marked the deformed batch's edges equal
    # to the rest batch's ON THE HOST); a deformed batch whose edges DIFFER - the reference builds them from a
    # second mesh file, nothing enforces equal triangle order - takes the stock formulation (it used to
    # turn both losses into NaN), and DC_LOSS_DEBUG keeps the device-side comparison as a debug assert
    model = load_model(SMALL).to(dev)
    collated = _batches(2, 2)[0]
    rest, deff, rig = loaders.to_batches(collated, dev)
    assert deff._dc_edges_equal[0] is True
    calls = []
    real = ops.contact_losses
    ops.contact_losses = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    fused = dc_train.losses(model, rest, deff, rig)
    assert len(calls) == 1
    old, dc_train.FUSED_LOSS = dc_train.FUSED_LOSS, False
    try:
        stock = dc_train.losses(model, rest, deff, rig)
    finally:
        dc_train.FUSED_LOSS = old
    for k in ("loss", "l1", "consistency"):
        assert abs(float(fused[k]) - float(stock[k])) <= 1e-5 * abs(float(stock[k])), k
    # a permuted deformed mesh: same edge SET per triangle order change -> different edge_index content
    names, rests, defs, meta, rigids = collated
    defs2 = []
    for d in defs:
        d2 = d.clone()
        d2.edge_index = torch.flip(d.edge_index, dims=[1])
        defs2.append(d2)
    rest2, deff2, rig2 = loaders.to_batches((names, rests, tuple(defs2), meta, rigids), dev)
    assert deff2._dc_edges_equal[0] is False
    out2 = dc_train.losses(model, rest2, deff2, rig2)
    assert len(calls) == 1                                               # stock formulation, no fused kernel
    from tests.helpers import G as _G
    with torch.no_grad():
        pred = model(rest2, rig2)
        pred.pos = pred.pos - rest2.pos
        tgt = _G(None, deff2.edge_index, deff2.pos - rest2.pos)
        want = torch.nn.functional.l1_loss(pred.pos, tgt.pos) + gradient_consistency_loss(pred, tgt)
    assert torch.isfinite(out2["loss"]) and abs(float(out2["loss"]) - float(want)) <= 1e-5 * abs(float(want))
    # debug assert: a batch MARKED equal whose device content differs poisons the losses
    old_dbg, dc_train.LOSS_DEBUG = dc_train.LOSS_DEBUG, True
    try:
        # (edited IN PLACE on the device copy, which the host-side mark cannot see; ASSIGNING a new edge_index would
        # drop the mark - tests/test_host_cpu.py - and take the stock formulation)
        deff.edge_index[0, 0] = (deff.edge_index[0, 0] + 1) % 5
        assert torch.isnan(dc_train.losses(model, rest, deff, rig)["loss"])
    finally:
        dc_train.LOSS_DEBUG = old_dbg
        ops.contact_losses = real


def _predict(model, rest, rig):
    with torch.no_grad():
        return model(rest, rig).pos


def test_per_graph_attention_mask_makes_predictions_independent_of_the_batch_cpu():
    """the reference's attention is unmasked across the batch, so a sample's prediction depends on
    which other samples share its batch.  With the opt-in `per_graph_mask` it does not: sample i predicted inside a batch of 3
    equals sample i predicted alone (CPU, oracle convs: host-side logic only).  With the option off (reference semantics)
    the two differ."""
    from deformcontact_amd.synth import make_batch
    torch.manual_seed(0)
    model = load_model(SMALL, conv_module=pyg_ref).eval()
    rest3, _, rig3 = make_batch(3, soft_vertices=64, sphere_resolution=4)
    n_s = rest3.x.shape[0] // 3
    for masked in (True, False):
        model.multihead_attention.per_graph_mask = masked
        full = _predict(model, rest3, rig3)
        diffs = []
        for i in range(3):
            r1, _, g1 = make_batch(1, first_idx=i, soft_vertices=64, sphere_resolution=4)
            alone = _predict(model, r1, g1)
            diffs.append(float((full[i * n_s:(i + 1) * n_s] - alone).abs().max()))
        if masked:
            assert max(diffs) < 1e-5, diffs
        else:
            assert max(diffs) > 1e-4, diffs


@pytest.mark.gpu
def test_per_graph_attention_mask_on_the_gpu_path_equals_single_sample_predictions():
    """The same property through the HIP encoder (batched layout taken from `Batch.segments()`), forward and gradients finite."""
    from deformcontact_amd.synth import make_batch
    torch.manual_seed(0)
    model = load_model(EVERYDAY_NETWORK).to("cuda:0")
    model.multihead_attention.per_graph_mask = True
    rest3, def3, rig3 = (b.to("cuda:0") for b in make_batch(3, soft_vertices=256, sphere_resolution=8))
    n_s = rest3.x.shape[0] // 3
    full = _predict(model, rest3, rig3)
    for i in range(3):
        r1, _, g1 = (b.to("cuda:0") for b in make_batch(1, first_idx=i, soft_vertices=256, sphere_resolution=8))
        alone = _predict(model, r1, g1)
        ref = float(alone.abs().max())
        assert float((full[i * n_s:(i + 1) * n_s] - alone).abs().max()) <= 1e-5 * ref
    out = losses(model, rest3, def3, rig3, 1.0)
    out["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
