"""Build + ctypes binding of ``oracle/hop_ref.c`` (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libhop_ref.so")
_lib = None


def build(force: bool = False) -> str:
    """Rebuilt when the CONTENT of hop_ref.c changed (file times do not survive a push to another machine)."""
    import hashlib
    src = os.path.join(_DIR, "hop_ref.c")
    with open(src, "rb") as f:
        want = hashlib.sha256(f.read()).hexdigest()
    stamp = _SO + ".srchash"
    have = None
    if os.path.exists(stamp):
        with open(stamp) as f:
            have = f.read().strip()
    if force or not os.path.exists(_SO) or have != want:
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                               src, "-o", _SO + ".tmp", "-lm"])
        os.replace(_SO + ".tmp", _SO)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def csr_build(edge_index: np.ndarray, n: int, key_row: int = 1):
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    e = ei.shape[1]
    ptr = np.zeros(n + 1, np.int32)
    other = np.zeros(e, np.int32)
    perm = np.zeros(e, np.int32)
    rc = lib().ref_csr_build(_p(ei, ctypes.c_int64), ctypes.c_int64(e), ctypes.c_int64(n),
                             ctypes.c_int(key_row), _p(ptr, ctypes.c_int32),
                             _p(other, ctypes.c_int32), _p(perm, ctypes.c_int32))
    if rc != 0:
        raise ValueError(f"ref_csr_build rc={rc}")
    return ptr, other, perm


def gcn_norm(edge_index: np.ndarray, n: int) -> np.ndarray:
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    w = np.zeros(ei.shape[1], np.float32)
    lib().ref_gcn_norm(_p(ei, ctypes.c_int64), ctypes.c_int64(ei.shape[1]), ctypes.c_int64(n),
                       _p(w, ctypes.c_float))
    return w


def hop(edge_index: np.ndarray, w: np.ndarray, x: np.ndarray) -> np.ndarray:
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, f = x.shape
    y = np.empty_like(x)
    lib().ref_hop(_p(ei, ctypes.c_int64), _p(w, ctypes.c_float), _p(x, ctypes.c_float),
                  ctypes.c_int64(f), ctypes.c_int64(ei.shape[1]), ctypes.c_int64(n),
                  ctypes.c_int64(f), _p(y, ctypes.c_float), ctypes.c_int64(f))
    return y


def hop_bf16(edge_index: np.ndarray, w, x_bits: np.ndarray, out_f32: bool) -> np.ndarray:
    """``x_bits``: uint16 bf16 bit patterns [N,F]; returns fp32 [N,F] or uint16 bit patterns."""
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    x_bits = np.ascontiguousarray(x_bits, dtype=np.uint16)
    n, f = x_bits.shape
    y = np.empty((n, f), np.float32 if out_f32 else np.uint16)
    wp = None
    if w is not None:
        w = np.ascontiguousarray(w, dtype=np.float32)
        wp = _p(w, ctypes.c_float)
    lib().ref_hop_bf16(_p(ei, ctypes.c_int64), wp, _p(x_bits, ctypes.c_uint16), ctypes.c_int64(f),
                       ctypes.c_int64(ei.shape[1]), ctypes.c_int64(n), ctypes.c_int64(f),
                       y.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(f),
                       ctypes.c_int(1 if out_f32 else 0))
    return y
