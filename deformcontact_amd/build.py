"""Compile the gfx950 C-ABI library in-tree with hipcc.

``python -m deformcontact_amd.build`` (or ``__graft_entry__.build()``) produces
``deformcontact_amd/libdeformcontact_hip.so``.  hipcc cross-compiles for gfx950
without a GPU, so this also runs in the CPU-only build container.  The ``.so``
is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
SO_PATH = os.path.join(PKG_DIR, "libdeformcontact_hip.so")
ARCH = "gfx950"

# -ffp-contract=off: the hop must round multiply and add separately (bit parity
# with a serial scatter_add_); the MFMA kernels are unaffected (intrinsics).
HIPCC_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-shared", "-std=c++17",
               "-ffp-contract=off", "-Wno-unused-value", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(os.path.dirname(PKG_DIR), "include", "deformcontact.h")]


def _digest(paths) -> str:
    """sha256 over the compile flags and the names + contents of ``paths``."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def source_hash() -> str:
    """What ``libdeformcontact_hip.so`` is a function of: every ``csrc/*.hip``, every header, the flags."""
    return _digest(sources() + headers())


HASH_PATH = SO_PATH + ".srchash"


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _stale() -> bool:
    """Content, not mtime: a tree pushed to another machine (gpurun snapshots, checkouts) carries arbitrary file
    times, and a pushed ``.so`` that matches its sources must never trigger a multi-minute rebuild there."""
    return not os.path.exists(SO_PATH) or _read(HASH_PATH) != source_hash()


def _compile_one(args):
    hipcc, src, obj, verbose = args
    cmd = [hipcc] + [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per ``csrc/*.hip`` (compiled in parallel, rebuilt only when the content of the source or of a
    header has changed), linked into ``libdeformcontact_hip.so``; the library's source hash is stored beside it."""
    if not force and not _stale():
        return SO_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdeformcontact_hip.so for gfx950")
    objdir = os.path.join(os.path.dirname(PKG_DIR), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    want = source_hash()
    hdrs = headers()
    jobs, objs, stamps = [], [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        d = _digest([src] + hdrs)
        if force or not os.path.exists(obj) or _read(obj + ".srchash") != d:
            jobs.append((hipcc, src, obj, verbose))
            stamps.append((obj + ".srchash", d))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 8)) as ex:
            list(ex.map(_compile_one, jobs))
        for path, d in stamps:
            with open(path, "w") as f:
                f.write(d + "\n")
    tmp = SO_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, SO_PATH)
    with open(HASH_PATH, "w") as f:
        f.write(want + "\n")
    return SO_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
