"""The multi-process paths (BASELINE.json configs[3] entry point): `python bench.py --gpus N` started
directly, the data-parallel hipGraph train step on two ranks, the RCCL collectives of the N-GPU path.

The GPU scenarios are run by tests/conftest.py BEFORE the first test (tests/launch_scenarios.py: child processes are
only ever started from a process that has not initialised HIP) and asserted on HERE, in the file that sorts last: a hung
launch can no longer stand between `pytest -x` and the parity tests.  Reference: /root/reference/train.py:36-58,71-73."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

from tests import launch_scenarios as ls

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
launch = pytest.mark.launch


def _slow_not_stuck(r, n=2):
    """A launch that ran into its deadline: were the ranks still making progress IN STEP when they were stopped (every rank
    recorded a collective within 45 s of the last marker of the launch, at least 3 each, counts within 1 of each other)?
    -> a description, or None (stuck, dead, or no evidence)."""
    from deformcontact_amd.launch import read_phase_times
    times = read_phase_times(r["dir"], n)
    marks = [[(t, p) for t, p in rank if p.startswith("collective ")] for rank in times]
    if any(len(m) < 3 for m in marks):
        return None
    end = max(t for rank in times for t, _ in rank)
    counts = [int(m[-1][1].split()[1]) for m in marks]
    if any(end - m[-1][0] > 45.0 for m in marks) or max(counts) - min(counts) > 1:
        return None
    per = [(m[-1][0] - m[0][0]) / max(len(m) - 1, 1) for m in marks]
    return (f"ranks stopped at the deadline while in step at collectives {counts}, {max(per):.1f} s per step with its "
            f"collective (a healthy box: the whole scenario takes 4 s)")


def _result(name):
    if ls.SKIPPED is not None:
        pytest.skip(ls.SKIPPED)
    assert name in ls.RESULTS, f"scenario {name} was not run at session start"
    r = ls.RESULTS[name]
    deadline = r["timed_out"] or "launch deadline" in (r.get("describe") or "") or "launch deadline" in (r.get("stdout") or "")
    if r["rc"] != 0 and deadline and name in ("bench_two_rank_gloo", "bench_torchrun_gloo"):
        # two ranks time-sharing ONE device over gloo is a stand-in whose speed depends on the box (host copies, TCP over
        # loopback, two processes on one GPU): a launch (both attempts) that was progressing in step when its DEADLINE came is a slow box,
        # not a failure of the N > 1 code - anything else (a dead rank, a stuck rank, ranks out of step) still fails
        slow = _slow_not_stuck(r)
        if slow is not None and os.environ.get("DC_ALLOW_SLOW_LAUNCH", "1") != "0":
            # reported as an expected FAILURE (xfailed in the summary), never as a pass or a plain skip;
            # DC_ALLOW_SLOW_LAUNCH=0 makes it a hard failure
            pytest.xfail(f"{name}: box too slow for the two-ranks-on-one-device smoke - {slow}")
    assert r["rc"] == 0, f"{name}: rc={r['rc']} timed_out={r['timed_out']} after {r['wall_s']:.0f}s\n{r['describe'][-6000:]}"
    if r.get("attempts", 1) > 1:
        # a launch that only passed on its second attempt is not silently green: a warning in the session
        # summary with the first attempt's record, a hard failure under DC_LAUNCH_STRICT=1
        first = r.get("first_attempt", {})
        msg = (f"{name}: passed only on attempt {r['attempts']}; first attempt rc={first.get('rc')} "
               f"timed_out={first.get('timed_out')} after {first.get('wall_s', 0):.0f}s - {str(first.get('describe'))[-600:]}")
        assert os.environ.get("DC_LAUNCH_STRICT", "0") != "1", msg
        import warnings
        warnings.warn(msg)
    return r


# --------------------------------------------------------------------------- CPU: the launcher itself
def _bench(args, env, timeout=120):
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    return r, time.time() - t0


def test_direct_multi_gpu_launch_fails_cleanly_when_a_rank_dies(tmp_path):
    """CPU container: the ranks cannot find a HIP device and exit; the launcher must come back non-zero (not hang),
    say why, and still print a JSON line naming the ranks it saw."""
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the ranks would run")
    r, _ = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], dict(os.environ, DC_RANK_LOG_DIR=str(tmp_path)))
    assert r.returncode != 0
    assert b"HIP device" in r.stderr
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["value"] is None and "error" in line
    assert [x["rank"] for x in line["dist"]["ranks_seen"]] == [0, 1]
    assert all(x["last_phase"] == "start" for x in line["dist"]["ranks_seen"])


def test_direct_multi_gpu_launch_stops_stalled_ranks_at_its_deadline(tmp_path):
    """Every rank stalls (DC_TEST_STALL_RANK=all): the launcher stops them at DC_LAUNCH_TIMEOUT and returns non-zero."""
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: GPU child processes are only started by tests/launch_scenarios.py")
    env = dict(os.environ, DC_RANK_LOG_DIR=str(tmp_path), DC_TEST_STALL_RANK="all", DC_LAUNCH_TIMEOUT="4")
    r, wall = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode != 0 and wall < 60
    assert b"launch deadline" in r.stderr
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["launch"]["timed_out"] and line["error"] == "launch deadline passed"
    assert all(x["last_phase"].startswith("stalled") for x in line["dist"]["ranks_seen"])


def test_one_stalled_rank_among_dead_ones_does_not_hold_the_launcher(tmp_path):
    """Rank 1 stalls, rank 0 dies (CPU: no device) or - on a GPU box - waits in the rendezvous: either way the launcher
    is back within its deadline with a non-zero code."""
    env = dict(os.environ, DC_RANK_LOG_DIR=str(tmp_path), DC_TEST_STALL_RANK="1", DC_LAUNCH_TIMEOUT="20",
               DC_DIST_BACKEND="gloo", DC_GLOO_TIMEOUT_S="5")
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: GPU child processes are only started by tests/launch_scenarios.py")
    r, wall = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-pmc"], env)
    assert r.returncode != 0 and wall < 90


def test_launch_ranks_reports_codes_phases_and_logs(tmp_path):
    from deformcontact_amd.launch import launch_ranks
    code = ("import os, sys; sys.path.insert(0, %r); from deformcontact_amd.launch import phase; phase('a'); "
            "print('out', os.environ['RANK']); print('err', file=sys.stderr); phase('b'); "
            "sys.exit(3 if os.environ['RANK'] == '1' else 0)") % ROOT
    res = launch_ranks(2, [sys.executable, "-c", code], timeout=60, log_dir=str(tmp_path))
    assert res.rc == 3 and not res.timed_out and res.rcs[1] == 3
    assert res.phases()[1] == ["a", "b"] and res.stdout0.strip() == "out 0"
    assert "rank 1: rc=3" in res.describe() and "err" in res.describe()
    ok = launch_ranks(2, [sys.executable, "-c", "pass"], timeout=60, log_dir=str(tmp_path))
    assert ok.rc == 0 and ok.rcs == [0, 0] and ok.phases() == [[], []]      # older markers cleared
    # the ranks' CPU thread pools are capped unless the caller chose (round 4's two-rank hang: two machine-wide pools)
    show = "import os; print(os.environ['OMP_NUM_THREADS'], os.environ['MKL_NUM_THREADS'])"
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    capped = launch_ranks(2, [sys.executable, "-c", show], timeout=60, log_dir=str(tmp_path), env=env)
    a, b = capped.stdout0.split()
    assert a == b and 1 <= int(a) <= 8
    kept = launch_ranks(2, [sys.executable, "-c", show], timeout=60, log_dir=str(tmp_path), env=dict(env, OMP_NUM_THREADS="3"))
    assert kept.stdout0.split()[0] == "3"


# --------------------------------------------------------------------------- GPU: what the scenarios left behind
@pytest.mark.gpu
@launch
def test_config3_direct_two_rank_launch_on_one_device_over_gloo():
    r = _result("bench_two_rank_gloo")
    lines = [l for l in r["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1, r["stdout"][-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64
    assert out["value"] > 0 and out["value_cached_topology"] > 0
    assert 0 < out["roofline"]["frac"] <= 1.0
    d = out["dist"]                       # what the driver reads at N > 1 (world size as torch.distributed sees it)
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["allreduce_bytes"] >= 572416 * 4
    assert d["allreduce_us"] is not None and d["allreduce_us"] > 0
    # the devices the ranks ran on, all-gathered (here: two ranks, ONE physical device), and what the eager
    # all-reduce + Adam tail costs the host per step
    assert [x["rank"] for x in d["ranks_seen"]] == [0, 1] and all(x["device"] for x in d["ranks_seen"])
    assert d["distinct_devices"] == 1 and d["ranks_seen"][0]["device"] == d["ranks_seen"][1]["device"]
    assert d["tail"].startswith("eager") and d["tail_host_us_per_step"] > 0
    # both ranks went through the same sequence of phases (same number of all-reduces at every marker)
    ph = r["phases"]
    assert len(ph) == 2 and ph[0] and [p for p in ph[0] if "all-reduces" in p] == [p for p in ph[1] if "all-reduces" in p]


@pytest.mark.gpu
@launch
def test_config3_torchrun_two_rank_launch_as_the_driver_starts_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 2 ...`: ONE JSON line, from rank 0, with the whole job's figures."""
    r = _result("bench_torchrun_gloo")
    lines = [l for l in r["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1, r["stdout"][-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    d = out["dist"]
    assert d["world_size"] == 2 and [x["rank"] for x in d["ranks_seen"]] == [0, 1]
    ph = r["phases"]
    assert len(ph) == 2 and ph[0] and [p for p in ph[0] if "all-reduces" in p] == [p for p in ph[1] if "all-reduces" in p]


@pytest.mark.gpu
@launch
def test_config3_bench_over_rccl_on_one_rank():
    """The N > 1 form of bench.py with the real collective library: backend "nccl" (= RCCL), world size 1."""
    r = _result("bench_rccl_single")
    lines = [l for l in r["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1, r["stdout"][-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["value_cached_topology"] > 0
    d = out["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["distinct_devices"] == 1
    assert d["tail"].startswith("eager") and d["allreduce_us"] is not None and d["allreduce_bytes"] >= 572416 * 4
    assert "RCCL grad all-reduce" in out["config"]["step"]


@pytest.mark.gpu
@launch
@pytest.mark.parametrize("branches", ["serial", "two_streams"])
def test_config3_two_rank_graphed_train_step_equals_single_process_mean_gradient(branches):
    """Two data-parallel ranks through `train.GraphedTrainStep` (forward + losses + backward replayed from one hipGraph,
    all-reduce + Adam outside it) hold, after 5 steps, exactly the parameters of ONE process that ran both ranks'
    batches and averaged the gradients: same kernels, same order -> the same bits, with the encoder branches on one
    stream and on two."""
    r = _result("dp_graphed_" + branches)
    out = os.path.join(r["dir"], "dp")
    r0, r1 = (json.load(open(f"{out}.rank{k}.json")) for k in range(2))
    assert r0["replays"] == r1["replays"] == 4
    assert r0["losses"] != r1["losses"]                                   # different batches per rank
    msg = (f"max |diff| {r0['max_abs_diff']:.3e} at scale {r0['scale']:.3e}; {r0['n_differing']} parameters differ, e.g. "
           f"{r0['differing']}")
    assert r0["bit_identical"], msg


@pytest.mark.gpu
@launch
def test_config3_rccl_collectives_of_the_n_gpu_path_on_one_rank():
    """backend "nccl" (= RCCL) with world_size 1 on the test box's GPU: init with device_id, broadcast, all_reduce(AVG)
    on the flat gradient bucket, float64 MAX, barrier - the ops `bench.py --gpus N` and `train.py` issue at N > 1."""
    r = _result("rccl_single")
    assert "RCCL_OK" in r["stdout"]


def test_slow_launch_is_told_from_a_stuck_one(tmp_path):
    """`_slow_not_stuck`: only ranks that were still recording collectives, in step, when the deadline came count as slow."""
    def write(rows0, rows1):
        d = tmp_path / f"case{len(list(tmp_path.iterdir()))}"
        d.mkdir()
        for r, rows in enumerate((rows0, rows1)):
            (d / f"rank{r}.phase").write_text("".join(f"{t:.3f} {p}\n" for t, p in rows))
        return {"dir": str(d)}
    prog = lambda n, dt, t0=100.0: [(t0, "start")] + [(t0 + 5 + dt * i, f"collective {i + 1}") for i in range(n)]
    assert "in step at collectives [9, 9]" in _slow_not_stuck(write(prog(9, 15.0), prog(9, 15.2)))
    assert _slow_not_stuck(write(prog(9, 15.0), prog(8, 15.0))) is not None                 # one collective apart: in flight
    assert _slow_not_stuck(write(prog(9, 15.0), prog(4, 15.0))) is None                     # rank 1 stopped 75 s earlier
    assert _slow_not_stuck(write(prog(9, 1.0) + [(300.0, "timed: warmup steps done")], prog(9, 1.0))) is None   # silent since
    assert _slow_not_stuck(write(prog(2, 15.0), prog(2, 15.0))) is None                     # too little evidence
    assert _slow_not_stuck(write([(100.0, "start")], [(100.0, "stalled (DC_TEST_STALL_RANK)")])) is None


def test_scenarios_retry_once_and_stop_at_their_budget(tmp_path, monkeypatch):
    """`launch_scenarios.run_all`: a failed scenario is run once more in a directory of its own; once the budget is spent
    nothing else is started (the parity tests must not be kept waiting by a box on which launches crawl)."""
    calls = []

    def make(name, rcs, seconds=0.0):
        def run(out):
            calls.append((name, os.path.basename(out)))
            time.sleep(seconds)
            return {"rc": rcs.pop(0), "timed_out": False, "wall_s": seconds, "dir": out, "stdout": "", "describe": "", "phases": []}
        return run
    monkeypatch.setattr(ls, "SCENARIOS", {"a": make("a", [1, 0]), "b": make("b", [0]), "c": make("c", [1, 1], 0.3),
                                          "d": make("d", [0])})
    monkeypatch.setattr(ls, "RESULTS", {})
    monkeypatch.setenv("DC_LAUNCH_BUDGET_S", "0.5")
    r = ls.run_all(outdir=str(tmp_path))
    assert calls == [("a", "a"), ("a", "a.attempt2"), ("b", "b"), ("c", "c"), ("c", "c.attempt2")]
    assert r["a"]["rc"] == 0 and r["a"]["attempts"] == 2 and r["a"]["first_attempt"]["rc"] == 1
    assert r["b"]["attempts"] == 1 and r["c"]["rc"] == 1 and r["c"]["attempts"] == 2
    assert r["d"]["rc"] == 126 and r["d"]["attempts"] == 0 and "not started" in r["d"]["describe"]
    assert json.load(open(tmp_path / "results.json"))["d"]["rc"] == 126
