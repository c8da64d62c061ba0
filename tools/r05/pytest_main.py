#!/usr/bin/env python3
"""pytest under tools/exp/run_with_lib.py (a variant build of the C-ABI library): run_with_lib.py <lib.so> tools/r05/pytest_main.py <pytest args>"""
import sys

import pytest

sys.exit(pytest.main(sys.argv[1:]))
