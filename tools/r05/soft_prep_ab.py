#!/usr/bin/env python3
"""bench.py with ContactEncoder.prepare_soft_weights_on_side = $SOFTPREP (1 | 0)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from deformcontact_amd.graphnet import ContactEncoder  # noqa: E402

ContactEncoder.prepare_soft_weights_on_side = os.environ.get("SOFTPREP", "1") != "0"
import bench  # noqa: E402

bench.main()
