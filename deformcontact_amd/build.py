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


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdeformcontact_hip.so for gfx950")
    tmp = SO_PATH + ".tmp"
    cmd = [hipcc] + HIPCC_FLAGS + sources() + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, SO_PATH)
    return SO_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
