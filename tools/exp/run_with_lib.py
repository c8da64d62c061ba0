#!/usr/bin/env python3
"""Run a script with another build of the C-ABI library (A/B experiments): run_with_lib.py <lib.so> <script.py> [args...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deformcontact_amd._lib as L  # noqa: E402

L.SO_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
