"""CPU oracle for the DeformContact message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deformcontact_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported CPU baseline.

What it restates
----------------
The reference (``/root/reference``) never implements the hot path's arithmetic
itself: ``models/model.py:2,39,45,49,71,77`` call ``torch_geometric.nn.TAGConv``
(default, ``configs/everyday.json:46``), ``GCNConv`` or ``GATConv``.  The pinned
dependency is ``pyg=2.5.2`` on ``pytorch=2.2.2`` CPU (``environment.yml:76,80``);
its source is NOT under ``/root/reference`` and the package is not installable
in the build container (no network).  ``oracle/pyg_ref.py`` therefore restates
PyG 2.5.2's published algorithm (``nn/conv/tag_conv.py``, ``gcn_conv.py``
(``gcn_norm``), ``gat_conv.py``, ``message_passing.py``, ``utils/_scatter.py``,
``utils/_softmax.py``, ``utils/loop.py``, ``data/batch.py``) with the same ATen op
sequence (``index_select`` -> ``mul`` -> ``scatter_add_``).

Parity status: **parity unpinned at the PyG boundary** -- the reference has no
tests, golden vectors or fixtures of its own, and real PyG
cannot be run here.  What IS pinned:

* the reference's own wiring (``models/model.py`` GraphNet, ``models/losses.py``,
  ``utils/pos_encoding.py``, ``loaders/collate.py``, ``configs/config.py``) is
  imported from ``/root/reference`` by ``oracle/make_golden.py`` (with this
  restatement injected as ``torch_geometric``) to generate ``tests/golden/*.npz``;
* the conv restatement is cross-checked against an independent float64 dense
  closed form (``oracle/closed_form.py``) and a scalar C restatement of the
  gather-scale-scatter hop (``oracle/hop_ref.c``).
"""
