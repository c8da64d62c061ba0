"""deformcontact_amd - MI355X-native message-passing hot path of DeformContact.

Public surface (PyG-shaped, see ``INTEGRATION.md``):

* ``deformcontact_amd.nn``   - ``TAGConv``, ``GCNConv``, ``GATConv`` (HIP kernels underneath)
* ``deformcontact_amd.data`` - ``Data``, ``Batch``
* ``deformcontact_amd.install_as_torch_geometric()`` - register the two modules
  above as ``torch_geometric.nn`` / ``torch_geometric.data`` so the reference's
  ``models/model.py``, ``train.py`` and ``eval.py`` run unchanged.

There is no CPU fallback; the compute path is ``libdeformcontact_hip.so`` (gfx950).
"""
from . import data, nn  # noqa: F401
from .pyg_alias import install_as_torch_geometric  # noqa: F401

__version__ = "0.1.0"
