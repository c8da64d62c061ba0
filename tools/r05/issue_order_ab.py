#!/usr/bin/env python3
"""Host ISSUE order of the two encoder branches (ORDER = rigid_first (shipped) | soft_first): the hipGraph executor keeps the
branch that was issued first on the launch queue; the other one starts a cross-queue dependency (~9 us) later.  Streams are
unchanged: soft on the caller's stream, rigid on the side stream."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from deformcontact_amd.graphnet import ContactEncoder, _segments_of  # noqa: E402


def encode_soft_first(self, graph_resting, graph_rigid):
    x_s, e_s = graph_resting.x, graph_resting.edge_index
    x_r, e_r = graph_rigid.x, graph_rigid.edge_index
    seg_s, seg_r = _segments_of(graph_resting), _segments_of(graph_rigid)
    main = torch.cuda.current_stream(x_s.device)
    side = self._side_stream(x_s.device)
    side.wait_stream(main)                       # the fork point, BEFORE anything of the soft branch is on `main`
    out_s = self._branch(self.conv_layers_resting, x_s, e_s, seg_s)
    with torch.cuda.stream(side):
        out_r = self._branch(self.conv_layers_rigid, x_r, e_r, seg_r)
    main.wait_stream(side)
    out_r.record_stream(main)
    return out_s, out_r


if os.environ.get("ORDER", "rigid_first") == "soft_first":
    ContactEncoder.encode = encode_soft_first
import bench  # noqa: E402

bench.main()
