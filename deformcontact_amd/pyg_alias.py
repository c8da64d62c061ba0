"""Opt-in alias: make ``import torch_geometric`` resolve to this package.

The reference imports ``from torch_geometric.nn import GATConv, GCNConv, TAGConv, knn``
(``/root/reference/models/model.py:2``), ``from torch_geometric.data import Batch``
(``train.py:6``, ``eval.py:3``) and ``Data`` (``utils/graph_utils.py:2``).  Calling
``install_as_torch_geometric()`` before importing the reference's modules makes those
statements bind to the MI355X implementations - no edit to the reference needed.
"""
from __future__ import annotations

import sys
import types


def install_as_torch_geometric(force: bool = False) -> None:
    if "torch_geometric" in sys.modules and not force:
        mod = sys.modules["torch_geometric"]
        if getattr(mod, "__deformcontact_amd__", False):
            return
        raise RuntimeError("a real torch_geometric is already imported; pass force=True to shadow it")
    from . import data as _data, nn as _nn
    tg = types.ModuleType("torch_geometric")
    tg.__deformcontact_amd__ = True
    tg.nn, tg.data = _nn, _data
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = _nn
    sys.modules["torch_geometric.data"] = _data
