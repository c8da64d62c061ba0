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


def pytest_sessionfinish(session, exitstatus):
    """Write the parity distances the run collected (tests/helpers.PARITY_LOG) next to the other GPU-box outputs:
    gpurun_out/parity_distances.json (or $DC_PARITY_LOG) - one record per compared tensor, plus a summary."""
    try:
        from tests import helpers
    except Exception:
        return
    log = helpers.PARITY_LOG
    if not log:
        return
    import json
    path = os.environ.get("DC_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity_distances.json"))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    special = [r for r in log if r.get("special")]
    log = [r for r in log if not r.get("special")]
    ds = [r["hip_vs_fp32_oracle"] for r in log if r["hip_vs_fp32_oracle"] is not None]
    fired = [r for r in log if r["float64_widening_fired"]]
    summary = {"comparisons": len(log), "float64_widening_fired": len(fired),
               "worst_hip_vs_fp32_oracle": max(ds) if ds else None,
               "worst_hip_vs_fp32_oracle_without_widening": max((r["hip_vs_fp32_oracle"] for r in log
                                                                  if not r["float64_widening_fired"]
                                                                  and r["hip_vs_fp32_oracle"] is not None), default=None),
               "worst_hip_vs_float64_where_widened": max((r["hip_vs_float64"] for r in fired), default=None),
               "worst_ratio_hip_to_oracle_distance_from_float64_where_widened": max(
                   (r["hip_vs_float64"] / max(r["fp32_oracle_vs_float64"], 1e-30) for r in fired
                    if r["hip_vs_float64"] is not None and r["hip_vs_float64"] > r["tol"]), default=None),
               "tests": len({r["test"] for r in log}), "special_comparisons": len(special)}
    with open(path, "w") as f:
        json.dump({"summary": summary, "widened": fired, "special": special, "all": log}, f, indent=1)
