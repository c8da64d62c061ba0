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


def _stale() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(os.path.dirname(PKG_DIR), "include", "deformcontact.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_one(args):
    hipcc, src, obj, verbose = args
    cmd = [hipcc] + [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per ``csrc/*.hip`` (compiled in parallel, rebuilt only when the source or a
    header is newer), linked into ``libdeformcontact_hip.so``."""
    if not force and not _stale():
        return SO_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdeformcontact_hip.so for gfx950")
    objdir = os.path.join(os.path.dirname(PKG_DIR), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")) +
                [os.path.join(os.path.dirname(PKG_DIR), "include", "deformcontact.h")])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (force or not os.path.exists(obj)
                or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)):
            jobs.append((hipcc, src, obj, verbose))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1, 8)) as ex:
            list(ex.map(_compile_one, jobs))
    tmp = SO_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, SO_PATH)
    return SO_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
