#!/usr/bin/env python3
"""bench.py with the shared status word of unvalidated segmented builds switched off (SHARED_STATUS=0: a private, cleared
word per new adjacency = one fill launch at the head of each branch's build, the behaviour before) or on."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from deformcontact_amd import graph  # noqa: E402

if os.environ.get("SHARED_STATUS", "1") == "0":
    graph._shared_status = lambda device: None
import bench  # noqa: E402

bench.main()
