"""Machine-independent deterministic parameter / input fills (TEST INFRASTRUCTURE).

Golden fixtures must not depend on a RNG stream that could differ between the
container that generated them and the GPU box, and must stay small.  Every value
is an exact dyadic rational produced by integer hashing, so float64 -> float32
conversion is exact everywhere.
"""
from __future__ import annotations

import numpy as np
import torch


def hashed_uniform(shape, salt: int, scale: float = 1.0) -> np.ndarray:
    """Values in ``[-scale/2, scale/2)`` on a 2^-16 grid, as float32."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503) + np.uint64(12345)) \
        & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    u = (h & np.uint64(0xFFFF)).astype(np.float64) / 65536.0 - 0.5
    return (u * scale).astype(np.float32).reshape(shape)


def fill_state_dict_(module: torch.nn.Module, salt0: int = 0) -> None:
    """Overwrite every parameter with ``hashed_uniform`` scaled like its default
    init (``2/sqrt(fan_in)`` span for matrices, 0.2 span for vectors so biases
    are non-zero and their gradients are exercised)."""
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(module.named_parameters())):
            if p.dim() >= 2:
                fan_in = p.shape[-1]
                scale = 2.0 / np.sqrt(fan_in)
            else:
                scale = 0.2
            p.copy_(torch.from_numpy(hashed_uniform(tuple(p.shape), salt0 + 17 * k + 1, scale)))
