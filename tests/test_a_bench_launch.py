"""`python bench.py --gpus N` invoked directly (no torchrun): the parent spawns the N ranks before
anything touches the GPU and relays rank 0's JSON line (BASELINE.json configs[3] entry point).

This file sorts first on purpose: the GPU test starts child processes and only does so while
this pytest process has not initialised the HIP runtime."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_direct_multi_gpu_launch_fails_cleanly_without_devices():
    """CPU container: the ranks cannot find a HIP device; the launcher must come back with a
    non-zero exit code instead of hanging."""
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test below")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    assert b"HIP device" in r.stderr


@pytest.mark.gpu
def test_direct_two_rank_launch_on_one_device_over_gloo():
    """N = 2 ranks sharing the one device of the test box, gradient all-reduce over gloo
    (DC_DIST_BACKEND=gloo exists for exactly this smoke test; on the 8-GPU node it is RCCL)."""
    if torch.cuda.is_initialized():
        pytest.skip("HIP already initialised in this process; not starting child processes from it")
    env = dict(os.environ, DC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--kernel-reps", "4", "--no-pmc", "--no-cpu-baseline",
                        "--no-full-step", "--no-strict-fp32", "--no-radius100k", "--no-merged", "--no-backbones"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64
    assert out["value"] > 0 and out["value_cached_topology"] > 0
    assert 0 < out["roofline"]["frac"] <= 1.0
    d = out["dist"]                       # what the driver reads at N > 1 (world size as torch.distributed sees it)
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["allreduce_bytes"] >= 572416 * 4
    assert d["allreduce_us"] is not None and d["allreduce_us"] > 0
    # the devices the ranks ran on, all-gathered (here: two ranks, ONE physical device), and what
    # the eager all-reduce + Adam tail costs the host per step
    assert [r["rank"] for r in d["ranks_seen"]] == [0, 1] and all(r["device"] for r in d["ranks_seen"])
    assert d["distinct_devices"] == 1 and d["ranks_seen"][0]["device"] == d["ranks_seen"][1]["device"]
    assert d["tail"].startswith("eager") and d["tail_host_us_per_step"] > 0
