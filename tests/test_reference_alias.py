"""Build-container only (skipped where /root/reference is absent, e.g. the GPU box): the
reference's own models/model.py imports and constructs against this package through the
torch_geometric alias, with identical state_dict keys/shapes, and its forward reaches our ops."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")),
                                reason="reference checkout not present")


def test_reference_graphnet_binds_to_this_package():
    import deformcontact_amd as dc
    from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model as our_load
    sys.dont_write_bytecode = True
    dc.install_as_torch_geometric()
    sys.path.insert(0, REF)
    try:
        from configs.config import Config
        from models.model_loader import load_model
        import models.model as ref_model
        assert ref_model.TAGConv is dc.nn.TAGConv
        cfg = Config(os.path.join(REF, "configs", "everyday.json"))
        m = load_model(cfg)
        ours = our_load(EVERYDAY_NETWORK)
        sd_ref, sd_ours = m.state_dict(), ours.state_dict()
        assert list(sd_ref) == list(sd_ours)
        assert all(sd_ref[k].shape == sd_ours[k].shape for k in sd_ref)
        ours.load_state_dict(sd_ref)                     # reference checkpoint loads
        # forward reaches our conv (which refuses CPU tensors - no fallback)
        from torch_geometric.data import Batch, Data
        g = Data(x=torch.zeros(4, 21), edge_index=torch.tensor([[0, 1], [1, 2]]), pos=torch.zeros(4, 3))
        r = Data(x=torch.zeros(3, 25), edge_index=torch.tensor([[0, 1], [1, 2]]), pos=torch.zeros(3, 3))
        with pytest.raises(RuntimeError, match="HIP device"):
            m(Batch.from_data_list([g]), Batch.from_data_list([r]))
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k.split(".")[0] in ("torch_geometric", "models", "configs")]:
            sys.modules.pop(k, None)
