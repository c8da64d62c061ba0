#!/usr/bin/env python3
"""Round 6: the whole train step at the shipped batch 4 (and at 8, 16) with the cross-attention on stock PyTorch (softmax + bmm:
the default below 2^26 scores) against this library's attention (DC_FUSED_ATTN=1).   python tools/r06/b4_attn_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from deformcontact_amd.graphnet import CrossAttention  # noqa: E402

dev = torch.device("cuda:0")
for batch in (4, 8, 16):
    for mode in ("0", "1"):
        CrossAttention.fused = mode
        r = bench.full_step_b4(dev, steps=20, batch=batch)
        print(f"batch {batch} fused={mode}: {r['ms_per_step']} ms per step (hipgraph {r['hipgraph']}), loss {r['loss']}", flush=True)
