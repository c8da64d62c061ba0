"""CPU: the Open3D-free dataset path (PLY reader, JSON meta, sample enumeration / split, item
tuples) against a tiny dataset written in the reference's on-disk layout
(/root/reference/loaders/everyday_deform.py:12-147)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from deformcontact_amd import dataset as ds
from deformcontact_amd import loaders, synth
from deformcontact_amd.features import triangles_to_edge_index

KEYS = ["force", "forceDirection", "objectWorldPos", "collisionPosition", "velocity", "angularVelocity",
        "inertiaTensorPosition", "inertiaTensorRotation", "deformerOrigin", "deformerCollisionPosition"]


def _meta_json(i):
    j = {"force": 2500.0 + i, "collisionImpulse": 1.5, "mass": 2.0, "gravity_enabled": True}
    for n, k in enumerate(KEYS[1:]):
        for a, ax in enumerate("XYZ"):
            j[k + ax] = 0.01 * (n + 1) + 0.001 * a + 0.0001 * i
    return [j]


@pytest.fixture()
def root(tmp_path):
    for o, obj in enumerate(["Box", "Cat"]):
        d = tmp_path / obj
        d.mkdir()
        rest = synth.soft_mesh(64, seed=o)
        ds.write_ply(str(d / "InitialMesh.ply"), rest, binary=bool(o))
        for i in range(5):
            deformed = synth.TriMesh(rest.vertices + 0.001 * (i + 1), rest.triangles)
            ds.write_ply(str(d / f"2024_{i:02d}.ply"), deformed, binary=(i % 2 == 0))
            (d / f"2024_{i:02d}.json").write_text(json.dumps(_meta_json(i)))
    return str(tmp_path)


def test_ply_roundtrip_ascii_and_binary(tmp_path):
    m = synth.uv_sphere(0.05, 5)
    for binary in (False, True):
        p = str(tmp_path / f"m{int(binary)}.ply")
        ds.write_ply(p, m, binary=binary)
        r = ds.read_ply(p)
        assert np.array_equal(r.triangles, m.triangles)
        assert np.allclose(r.vertices, m.vertices.astype(np.float32), atol=1e-7)
    # extra vertex properties, quads (fan-split) and an unknown element are tolerated
    p = str(tmp_path / "extra.ply")
    open(p, "w").write("ply\nformat ascii 1.0\nelement vertex 4\nproperty double x\nproperty double y\n"
                       "property double z\nproperty uchar red\nelement face 1\n"
                       "property list uchar uint vertex_index\nelement edge 1\nproperty int a\nend_header\n"
                       "0 0 0 9\n1 0 0 9\n1 1 0 9\n0 1 0 9\n4 0 1 2 3\n7\n")
    r = ds.read_ply(p)
    assert r.vertices.shape == (4, 3) and r.triangles.tolist() == [[0, 1, 2], [0, 2, 3]]
    with pytest.raises(ValueError):
        open(p, "w").write("plx\n")
        ds.read_ply(p)


def test_dataset_items_match_reference_semantics(root):
    train = ds.EverydayDeformDataset(root, ["Box", "Cat"], split="train")
    val = ds.EverydayDeformDataset(root, ["Box", "Cat"], split="val")
    assert len(train) == 8 and len(val) == 2                       # sorted 80/20 per object
    assert train.samples[0] == os.path.join("Box", "2024_00") and val.samples == [
        os.path.join("Box", "2024_04"), os.path.join("Cat", "2024_04")]
    name, rest, deff, meta, rig = train[1]
    assert name == "Box" and rest.x.shape == (64, 21) and rig.x.shape == (762, 25)
    j = _meta_json(1)[0]
    # Unity -> Open3D axis swap (x, y, z) -> (z, -x, y); force normalised by force_max
    assert meta["force"] == j["force"] / 10000
    assert np.allclose(meta["force_vector"].numpy(), [j["forceDirectionZ"], -j["forceDirectionX"], j["forceDirectionY"]])
    shift = np.array([j["objectWorldPosZ"], -j["objectWorldPosX"], j["objectWorldPosY"]], dtype=np.float32)
    rest0 = ds.read_ply(os.path.join(root, "Box", "InitialMesh.ply"))
    assert np.allclose(rest.pos.numpy(), (rest0.vertices + shift).astype(np.float32), atol=1e-6)
    assert np.allclose(deff.pos.numpy() - rest.pos.numpy(), 0.002, atol=1e-6)
    assert torch.equal(rest.edge_index, triangles_to_edge_index(rest0.triangles))
    # rigid sphere centred at the deformer collision position, features [dir(3), force(1), enc(21)]
    c = np.array([j["deformerCollisionPositionZ"], -j["deformerCollisionPositionX"], j["deformerCollisionPositionY"]])
    assert np.allclose(rig.pos.numpy().mean(0), c, atol=1e-5)
    assert torch.allclose(rig.x[:, :3], meta["force_vector"].expand(762, 3))
    assert torch.allclose(rig.x[:, 3], torch.full((762,), j["force"] / 10000))
    assert np.allclose(np.linalg.norm(rig.pos.numpy() - rig.pos.numpy().mean(0), axis=1), 0.05, atol=1e-6)
    # contact_position is collision - objectWorldPos (then swapped)
    cp = [j["collisionPosition" + a] - j["objectWorldPos" + a] for a in "XYZ"]
    assert np.allclose(meta["contact_position"].numpy(), [cp[2], -cp[0], cp[1]], atol=1e-7)


def test_load_dataset_and_batches(root):
    cfg = types.SimpleNamespace(
        dataset=types.SimpleNamespace(name="everyday", root_dir=root, obj_list=["Box", "Cat"], n_points=-1,
                                      graph_method="knn", neigbor_k=5, neigbor_radius=0.15,
                                      sphere_radius=0.05, force_max=10000),
        dataloader=types.SimpleNamespace(batch_size=4, shuffle=True))
    tr, va = ds.load_dataset(cfg)
    assert len(tr) == 2 and len(va) == 1
    batches = list(tr)
    assert sum(len(b[0]) for b in batches) == 8
    rest, deff, rig = loaders.to_batches(batches[0])
    assert rest.num_graphs == 4 and rest.x.shape == (256, 21) and rig.x.shape == (4 * 762, 25)
    assert isinstance(batches[0][3]["force"], list) and batches[0][3]["force_vector"].shape == (4, 3)


def test_sample_nearest_subsets_mesh(root):
    d = ds.EverydayDeformDataset(root, ["Box"], n_points=20, split="train")
    _, rest, deff, _, _ = d[0]
    assert rest.x.shape == (20, 21) and deff.x.shape == (20, 21)
    assert rest.edge_index.numel() > 0 and int(rest.edge_index.max()) < 20
    assert torch.equal(rest.edge_index, deff.edge_index)
