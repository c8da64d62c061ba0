import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "launch: asserts on a multi-process GPU scenario run at session start "
                                       "(tests/launch_scenarios.py)")
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


def pytest_collection_finish(session):
    """The multi-process GPU scenarios run HERE - after collection / deselection, before the first test, while this
    process has not initialised HIP (child processes are never started from one that has) - and only when a selected
    test asserts on them (tests/test_z_launch.py, which sorts last)."""
    from tests import launch_scenarios as ls
    wanted = [it for it in session.items if "launch" in it.keywords]
    if not wanted or session.config.option.collectonly:
        return
    import torch
    if torch.cuda.device_count() == 0:
        ls.SKIPPED = "no GPU visible"
    elif os.environ.get("DC_SKIP_LAUNCH") == "1":
        ls.SKIPPED = "DC_SKIP_LAUNCH=1"
    elif torch.cuda.is_initialized():
        ls.SKIPPED = "HIP already initialised in the pytest process before the launch scenarios could start"
    else:
        names = []
        for it in wanted:                                   # only the scenarios the selected tests read
            for name in ls.SCENARIOS:
                tag = {"bench_two_rank_gloo": "direct_two_rank", "bench_torchrun_gloo": "torchrun_two_rank",
                       "bench_rccl_single": "bench_over_rccl",
                       "rccl_single": "rccl",
                       "dp_graphed_serial": "graphed_train_step", "dp_graphed_two_streams": "graphed_train_step"}[name]
                if tag in it.name and (not name.startswith("dp_graphed_") or name[len("dp_graphed_"):] in it.name):
                    if name not in names:
                        names.append(name)
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        if tr is not None:
            tr.write_line(f"launch scenarios before the first test: {names}")
        ls.run_all(names)
        if tr is not None:
            for k, v in ls.RESULTS.items():
                tr.write_line(f"  {k}: rc={v['rc']} timed_out={v['timed_out']} {v['wall_s']:.0f}s"
                              + (f" (attempt 2; the first: rc={v['first_attempt']['rc']} after "
                                 f"{v['first_attempt']['wall_s']:.0f}s)" if v.get("attempts", 1) > 1 else ""))


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
               # round 5: a widened comparison passes only with HIP within tol of float64; a wider bound needs `special`
               "widened_non_special_beyond_tol_of_float64": sum(
                   1 for r in fired if r["hip_vs_float64"] is not None and r["hip_vs_float64"] > r["tol"]),
               "tests": len({r["test"] for r in log}), "special_comparisons": len(special)}
    with open(path, "w") as f:
        json.dump({"summary": summary, "widened": fired, "special": special, "all": log}, f, indent=1)
