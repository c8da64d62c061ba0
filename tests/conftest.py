import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the C-ABI library exists (hipcc cross-compiles without a GPU)
    from deformcontact_amd import build as dc_build
    dc_build.build()
    from oracle import hop_c
    hop_c.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    import torch
    # device_count() does not initialise the HIP runtime in this process (is_available() does):
    # tests/test_a_bench_launch.py starts child processes and must do so from a clean process
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
