"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own Python.

TEST INFRASTRUCTURE.  Runs only in the build container (needs ``/root/reference``);
the fixtures it writes are committed and are all that travels to the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

What is imported from ``/root/reference`` (unchanged, nothing copied):
  * ``models/model_loader.py: load_model`` -> ``models/model.py: GraphNet`` - the
    encoder loops, ReLU/dropout, unmasked cross-attention, decoder, ``clone()`` /
    ``pos +=`` wiring (``models/model.py:67-97``);
  * ``models/losses.py: GradientConsistencyLoss``; ``utils/pos_encoding.py:
    to_log_freq``; ``utils/graph_utils.py: mesh_to_graph``; ``loaders/collate.py:
    collate_fn``; ``configs/config.py: Config`` with ``configs/everyday.json``.
``torch_geometric`` is absent from the image, so ``oracle/pyg_ref.py`` is
registered under that name first; only the conv
internals are therefore a restatement.  ``loaders/common.py`` imports ``open3d`` at
module level (absent here): it is imported behind an EMPTY stub module of that name -
``_feature_rigid`` (``:6-19``) itself touches only torch - and its output is stored in
``feature_rigid.npz``; ``rigid_features`` below (used while building the other fixtures)
is asserted equal to it.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def install_oracle_as_pyg():
    from oracle import pyg_ref
    tg = types.ModuleType("torch_geometric")
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_data = types.ModuleType("torch_geometric.data")
    for n in ("TAGConv", "GCNConv", "GATConv", "knn"):
        setattr(tg_nn, n, getattr(pyg_ref, n))
    tg_data.Data, tg_data.Batch = pyg_ref.Data, pyg_ref.Batch
    tg.nn, tg.data = tg_nn, tg_data
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tg_nn,
                        "torch_geometric.data": tg_data})


def rigid_features(force_vector: torch.Tensor, force: float, pos_enc: torch.Tensor):
    """Restates ``loaders/common.py:6-19``: cat[force_vector x V, force x V, pos_enc]."""
    v = force_vector.repeat(pos_enc.shape[0], 1)
    s = torch.tensor(force, dtype=torch.float32).repeat(pos_enc.shape[0], 1)
    return torch.cat([v, s, pos_enc], dim=1)


def sample(i: int, mesh_to_graph):
    """One (soft_rest, soft_def, rigid) triple of tiny meshes, closed-form."""
    from oracle.meshes import octasphere, uv_sphere
    soft = octasphere(1 + (i % 2), radius=0.15 + 0.02 * i).translate([0.01 * i, -0.02, 0.03])
    contact = soft.vertices[(7 * i + 3) % len(soft.vertices)]
    rigid = uv_sphere(0.05, 4 + i).translate(contact)
    d = soft.vertices - contact
    bump = 0.03 * np.exp(-(d ** 2).sum(1, keepdims=True) / (2 * 0.08 ** 2))
    direction = np.array([0.3, -0.5, 0.81]) / np.linalg.norm([0.3, -0.5, 0.81])
    soft_def = type(soft)(soft.vertices + bump * direction, soft.triangles)
    g_rest, g_def, g_rig = mesh_to_graph(soft), mesh_to_graph(soft_def), mesh_to_graph(rigid)
    g_rig.x = rigid_features(torch.tensor(direction, dtype=torch.float32), 0.25 + 0.5 * i, g_rig.x)
    return g_rest, g_def, g_rig


def np_(t):
    return t.detach().cpu().numpy()


def graphnet_fixture(backbone: str, hidden: int, full: bool, fname: str):
    from configs.config import Config
    from models.model_loader import load_model
    from models.losses import GradientConsistencyLoss
    from utils.graph_utils import mesh_to_graph
    from torch_geometric.data import Batch
    from oracle.weights import fill_state_dict_

    cfg = Config(os.path.join(REF, "configs", "everyday.json"),
                 updates={"network": {"hidden_dim": hidden, "backbone": backbone}})
    model = load_model(cfg)
    fill_state_dict_(model)
    model.train()

    triples = [sample(i, mesh_to_graph) for i in range(2)]
    rest = Batch.from_data_list([t[0] for t in triples])
    deff = Batch.from_data_list([t[1] for t in triples])
    rig = Batch.from_data_list([t[2] for t in triples])

    acts = {}
    for br in ("resting", "rigid"):
        for li, conv in enumerate(getattr(model, f"conv_layers_{br}")):
            conv.register_forward_hook(
                lambda m, i, o, k=f"conv_{br}_{li}": acts.__setitem__(k, o.detach().clone()))

    out = dict(
        rest_x=np_(rest.x), rest_pos=np_(rest.pos), rest_edge_index=np_(rest.edge_index),
        rest_ptr=np_(rest.ptr), def_pos=np_(deff.pos),
        rig_x=np_(rig.x), rig_pos=np_(rig.pos), rig_edge_index=np_(rig.edge_index),
        rig_ptr=np_(rig.ptr), hidden=np.int64(hidden))

    pred = model(rest, rig)                                   # models/model.py:67-97
    for k, v in acts.items():
        out[k] = np_(v)
    if full:
        out["pred_pos"] = np_(pred.pos)
        # train.py:46-58 step maths
        pred.pos = pred.pos - rest.pos
        deff.pos = deff.pos - rest.pos
        l1 = torch.nn.L1Loss()(pred.pos, deff.pos)
        gcl = GradientConsistencyLoss()(pred, deff)
        loss = l1 + cfg.training.lambda_gradient * gcl
        loss.backward()
        out.update(loss_l1=np_(l1), loss_gcl=np_(gcl), loss=np_(loss))
        for name, p in model.named_parameters():
            out["grad." + name] = np_(p.grad)
    else:
        # encoder-only backward with a closed-form upstream gradient
        from oracle.weights import hashed_uniform
        gs = {}
        total = 0.0
        x_rest = rest.x
        for conv in model.conv_layers_resting:
            x_rest = torch.relu(conv(x_rest, rest.edge_index))
        x_rig = rig.x
        for conv in model.conv_layers_rigid:
            x_rig = torch.relu(conv(x_rig, rig.edge_index))
        for k, t in (("rest", x_rest), ("rig", x_rig)):
            g = torch.from_numpy(hashed_uniform(tuple(t.shape), 991 if k == "rest" else 997, 2.0))
            gs[k] = g
            total = total + (t * g).sum()
        total.backward()
        out["enc_rest"], out["enc_rig"] = np_(x_rest), np_(x_rig)
        for name, p in model.named_parameters():
            if name.startswith("conv_layers") and p.numel() <= 8192:
                out["grad." + name] = np_(p.grad)
            elif name.startswith("conv_layers"):
                # large matrices: keep a strided probe + exact float64 checksum
                g = np_(p.grad)
                out["gradprobe." + name] = g.reshape(-1)[::97].copy()
                out["gradsum." + name] = np.float64(g.astype(np.float64).sum())
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print("wrote", fname, {k: getattr(v, "shape", None) for k, v in list(out.items())[:4]})


def small_fixtures():
    from utils.pos_encoding import to_log_freq
    from models.losses import GradientConsistencyLoss
    from loaders.collate import collate_fn
    from utils.graph_utils import mesh_to_graph
    from torch_geometric.data import Batch
    from oracle.weights import hashed_uniform
    from oracle.meshes import octasphere, uv_sphere

    pos = torch.from_numpy(hashed_uniform((37, 3), 5, 0.6))
    np.savez_compressed(os.path.join(OUT, "pos_encoding.npz"),
                        pos=np_(pos), enc=np_(to_log_freq(pos, 3, 1)))

    # mesh_to_graph edge enumeration (utils/graph_utils.py:12-13) on both mesh kinds
    m1, m2 = octasphere(1), uv_sphere(0.05, 5)
    g1, g2 = mesh_to_graph(m1), mesh_to_graph(m2)
    b = Batch.from_data_list([g1, g2, g1])
    ei = np_(b.edge_index)
    perm = np.argsort(ei[1], kind="stable")
    perm_t = np.argsort(ei[0], kind="stable")
    n = b.x.shape[0]
    np.savez_compressed(
        os.path.join(OUT, "mesh_graph_csr.npz"),
        v1=m1.vertices, t1=m1.triangles, v2=m2.vertices, t2=m2.triangles,
        x1=np_(g1.x), ei1=np_(g1.edge_index), x2=np_(g2.x), ei2=np_(g2.edge_index),
        batch_edge_index=ei, batch_x=np_(b.x), batch_vec=np_(b.batch), batch_ptr=np_(b.ptr),
        rowptr=np.concatenate([[0], np.cumsum(np.bincount(ei[1], minlength=n))]).astype(np.int32),
        perm=perm.astype(np.int32), src_sorted=ei[0][perm].astype(np.int32),
        colptr=np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=n))]).astype(np.int32),
        perm_t=perm_t.astype(np.int32), dst_sorted=ei[1][perm_t].astype(np.int32))

    # GradientConsistencyLoss (models/losses.py:7-19) on duck-typed batches
    class G:
        pass
    p, t = G(), G()
    p.pos = torch.from_numpy(hashed_uniform((n, 3), 11, 0.5)).requires_grad_(True)
    t.pos = torch.from_numpy(hashed_uniform((n, 3), 13, 0.5))
    p.edge_index = t.edge_index = b.edge_index
    loss = GradientConsistencyLoss()(p, t)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "gcl_loss.npz"), pred_pos=np_(p.pos), tgt_pos=np_(t.pos),
                        edge_index=ei, loss=np_(loss), grad_pred=np_(p.pos.grad))

    # collate_fn (loaders/collate.py:4-16)
    items = [("Box", g1, g1, {"force": 0.5, "v": torch.ones(3)}, g2),
             ("Cat", g2, g2, {"force": 0.7, "v": torch.zeros(3)}, g1)]
    names, rest, deff, meta, rig = collate_fn(items)
    assert names == ["Box", "Cat"] and meta["force"] == [0.5, 0.7] and meta["v"].shape == (2, 3)
    assert rest == (g1, g2) and rig == (g2, g1)
    print("wrote small fixtures")


def feature_rigid_fixture():
    """``loaders/common.py:6-19`` ``_feature_rigid`` - the REFERENCE's own function, imported with a stub
    ``open3d`` (the module only needs the name at import time) - on closed-form inputs."""
    from oracle.weights import hashed_uniform
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    from loaders.common import _feature_rigid
    from utils.pos_encoding import to_log_freq
    cases = {}
    for i, (v, f) in enumerate([(7, 0.25), (1, 0.0), (42, 0.875)]):
        pos = torch.from_numpy(hashed_uniform((v, 3), 31 + i, 0.4))
        enc = to_log_freq(pos, 3, 1)
        fv = torch.from_numpy(hashed_uniform((3,), 41 + i, 1.0))
        out = _feature_rigid({"force_vector": fv, "force": f}, enc)
        assert torch.equal(out, rigid_features(fv, f, enc)), "make_golden.rigid_features != reference"
        cases.update({f"force_vector{i}": np_(fv), f"force{i}": np.float32(f), f"pos_enc{i}": np_(enc),
                      f"features{i}": np_(out)})
    np.savez_compressed(os.path.join(OUT, "feature_rigid.npz"), n_cases=3, **cases)
    print("wrote feature_rigid.npz")


def main():
    os.makedirs(OUT, exist_ok=True)
    install_oracle_as_pyg()
    sys.path.insert(0, REF)
    torch.set_num_threads(1)       # deterministic CPU summation order for the fixtures
    feature_rigid_fixture()
    if "--only-feature-rigid" in sys.argv:
        return
    small_fixtures()
    graphnet_fixture("TAGConv", 32, True, "graphnet_tag_h32.npz")
    graphnet_fixture("GCNConv", 32, True, "graphnet_gcn_h32.npz")
    graphnet_fixture("GATConv", 32, True, "graphnet_gat_h32.npz")
    graphnet_fixture("TAGConv", 256, False, "encoder_tag_h256.npz")


if __name__ == "__main__":
    main()
