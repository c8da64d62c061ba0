#!/usr/bin/env python3
"""bench.py with ops.DW_POSITION set from $DWF: which branch issues its wide layer's dW in front
of the transposed chain + dX."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from deformcontact_amd import ops  # noqa: E402

# DWF = <soft position>,<rigid position>, each first | mid | last   (shipped: first,last)
_soft, _rigid = os.environ.get("DWF", "first,last").split(",")
ops.DW_POSITION.update({"unlisted": _soft, "listed": _rigid})
import bench  # noqa: E402

bench.main()
