"""One process per GPU without torchrun: start the ranks, watch them, stop them at a deadline.

SURVEY.md 8(e): the hot path shards by sample, one rank per GPU, one gradient all-reduce per step.  The reference
has no launcher (``/root/reference/train.py`` is single-process); ``python bench.py --gpus N`` and the multi-process
tests start their ranks through this module.  Rules it keeps (they come from the GPU pool this runs on):

* the parent never touches the GPU, and nothing is exec'ed from a process that has;
* a rank that dies takes the others down with it (exactly the PIDs started here, never a pattern);
* a stuck rendezvous or collective cannot hold the caller for ever: ``timeout`` seconds, then every rank is stopped;
* every rank's stdout / stderr go to their own files (``log_dir``) and every rank leaves PHASE MARKERS there, so a
  failure says which rank stopped where - round 4's only record of a two-rank hang was the last 2,000 bytes of two
  interleaved tracebacks;
* the ranks' CPU thread pools are capped (``OMP_NUM_THREADS`` / ``MKL_NUM_THREADS``, unless the caller set them) - as
  torchrun does.  THIS is what round 4's hang was: with N ranks each owning an OpenMP / MKL pool as wide as the machine, the
  pools' spinning workers starve each other and every tiny MKL vector call of the synthetic-batch generation
  (``features.to_log_freq``: sin / cos of a [1024, 3] tensor) takes ~0.1 s - round 5's watchdog caught both ranks of a
  two-rank launch still generating their first batches after 160 s (``profiles/r05/e_two_rank_launch_failure/``), a job that takes
  2 s alone; in round 4 the slower rank arrived at the first gradient all-reduce after the faster one's 240 s gloo timeout.

Environment a rank sees: ``RANK``, ``LOCAL_RANK``, ``WORLD_SIZE``, ``MASTER_ADDR=127.0.0.1``, ``MASTER_PORT``,
``DC_RANK_LOG_DIR`` (phase markers), ``DC_RANK_WATCHDOG_S`` (``install_watchdog``).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_cpu_threads(n: int) -> int:
    """CPU threads per rank: the ranks of one node share its cores (half of them, split N ways, at most 8 each)."""
    return max(1, min(8, (os.cpu_count() or 1) // (2 * max(n, 1))))


def phase(name: str) -> None:
    """Rank side: note that this rank has reached ``name`` (appends ``<seconds> <name>`` to
    ``$DC_RANK_LOG_DIR/rank<RANK>.phase``; nothing without that variable)."""
    d = os.environ.get("DC_RANK_LOG_DIR")
    if not d:
        return
    try:
        with open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}.phase"), "a") as f:
            f.write(f"{time.time():.3f} {name}\n")
    except OSError:
        pass


def install_watchdog(default_s: float = 0.0) -> None:
    """Rank side: ``faulthandler`` on fatal signals, and - with ``DC_RANK_WATCHDOG_S`` > 0 - a dump of every thread's
    Python stack to stderr each time that many seconds pass (a rank stuck in a collective shows where)."""
    import faulthandler
    faulthandler.enable()
    s = float(os.environ.get("DC_RANK_WATCHDOG_S", default_s) or 0)
    if s > 0:
        faulthandler.dump_traceback_later(s, repeat=True)


def _tail(path: str, nbytes: int = 3000) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            size = f.tell()
            f.seek(max(0, size - nbytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def read_phases(log_dir: str, n: int) -> List[List[str]]:
    """Per rank, the phase names it recorded, in order."""
    out = []
    for r in range(n):
        try:
            with open(os.path.join(log_dir, f"rank{r}.phase")) as f:
                out.append([ln.split(" ", 1)[1].strip() for ln in f if " " in ln])
        except OSError:
            out.append([])
    return out


def read_phase_times(log_dir: str, n: int) -> List[List[tuple]]:
    """Per rank, ``(seconds, phase name)`` of every marker it recorded, in order."""
    out = []
    for r in range(n):
        rows = []
        try:
            with open(os.path.join(log_dir, f"rank{r}.phase")) as f:
                for ln in f:
                    t, _, name = ln.partition(" ")
                    try:
                        rows.append((float(t), name.strip()))
                    except ValueError:
                        pass
        except OSError:
            pass
        out.append(rows)
    return out


class LaunchResult:
    def __init__(self, rc: int, rcs: List[Optional[int]], timed_out: bool, wall_s: float, log_dir: str, n: int):
        self.rc, self.rcs, self.timed_out, self.wall_s, self.log_dir, self.n = rc, rcs, timed_out, wall_s, log_dir, n

    @property
    def stdout0(self) -> str:
        try:
            with open(os.path.join(self.log_dir, "rank0.out")) as f:
                return f.read()
        except OSError:
            return ""

    def phases(self) -> List[List[str]]:
        return read_phases(self.log_dir, self.n)

    def summary(self) -> Dict:
        ph = self.phases()
        return {"rc": self.rc, "timed_out": self.timed_out, "wall_s": round(self.wall_s, 1),
                "ranks": [{"rank": r, "rc": self.rcs[r], "phases": len(ph[r]), "last_phase": ph[r][-1] if ph[r] else None}
                          for r in range(self.n)]}

    def describe(self, nbytes: int = 2500) -> str:
        """Human-readable record of a failed launch: per rank its exit code, last phase, and the end of its stderr."""
        lines = [f"launch rc={self.rc} timed_out={self.timed_out} wall={self.wall_s:.1f}s logs={self.log_dir}"]
        ph = self.phases()
        for r in range(self.n):
            lines.append(f"--- rank {r}: rc={self.rcs[r]} phases={ph[r][-6:]}")
            lines.append(_tail(os.path.join(self.log_dir, f"rank{r}.err"), nbytes))
        return "\n".join(lines)


def launch_ranks(n: int, argv: Sequence[str], *, timeout: float, log_dir: Optional[str] = None,
                 env: Optional[Dict[str, str]] = None, rank_env: Optional[Sequence[Dict[str, str]]] = None,
                 poll_s: float = 0.1) -> LaunchResult:
    """Run ``argv`` as ``n`` ranks and wait.  -> ``LaunchResult``; ``rc`` is 0 only when every rank returned 0 before
    ``timeout``; 124 when the deadline stopped them; else the first failing rank's code.  Never raises for a rank's
    failure; the ranks' output is in ``log_dir`` (a fresh temporary directory by default; kept)."""
    log_dir = log_dir or tempfile.mkdtemp(prefix="dc_launch_")
    os.makedirs(log_dir, exist_ok=True)
    for r in range(n):                                   # a reused directory must not show an older launch's markers
        for suffix in ("phase", "out", "err"):
            try:
                os.remove(os.path.join(log_dir, f"rank{r}.{suffix}"))
            except OSError:
                pass
    port = free_port()
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    threads = str(rank_cpu_threads(n))
    base.setdefault("OMP_NUM_THREADS", threads)
    base.setdefault("MKL_NUM_THREADS", threads)
    procs, files = [], []
    t0 = time.time()
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                     MASTER_PORT=str(port), DC_RANK_LOG_DIR=log_dir)
            if rank_env is not None:
                e.update(rank_env[r])
            fo = open(os.path.join(log_dir, f"rank{r}.out"), "wb")
            fe = open(os.path.join(log_dir, f"rank{r}.err"), "wb")
            files += [fo, fe]
            procs.append(subprocess.Popen(list(argv), env=e, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL))
        rc, timed_out = 0, False
        live = list(procs)
        deadline = t0 + float(timeout)
        while live and rc == 0:
            time.sleep(poll_s)
            if time.time() > deadline:
                rc, timed_out = 124, True
                break
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = rc or code
        for p in live:                                   # stop exactly the PIDs started here
            p.kill()
        for p in live:
            p.wait()
    finally:
        for f in files:
            f.close()
    rcs = [p.returncode for p in procs]
    return LaunchResult(rc, rcs, timed_out, time.time() - t0, log_dir, n)
