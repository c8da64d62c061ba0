#!/usr/bin/env python3
"""bench.py with ContactEncoder.prepare_weights_ahead forced off (PREP_AHEAD=0) or left on: A/B of the weight
preparation launched on helper streams at the start of a branch against the inline launches."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402

ContactEncoder.prepare_weights_ahead = os.environ.get("PREP_AHEAD", "1") != "0"
if os.environ.get("PREP_HELPERS") == "1":          # one helper stream for both branches' preparation
    import torch
    ContactEncoder._helper_stream = classmethod(lambda cls, device, which: cls._helper_streams.setdefault(
        "one", torch.cuda.Stream(device=device)))
import bench  # noqa: E402

bench.main()
