#!/usr/bin/env python3
"""bench.py with ops.DW_FIRST_MODE = $DWF (none | listed | unlisted | all): which branch issues its wide layer's dW in front
of the transposed chain + dX."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from deformcontact_amd import ops  # noqa: E402

ops.DW_FIRST_MODE = os.environ.get("DWF", "listed")
import bench  # noqa: E402

bench.main()
