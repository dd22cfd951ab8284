"""``gymnasium.spaces.Box`` when gymnasium is installed, else a shape/dtype-only stand-in
(the manager stack only reads ``.shape``; genesis_forge/managers/observation_manager.py:182-216)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Space  # type: ignore
except Exception:  # gymnasium absent (this image)

    class Space:  # type: ignore
        shape: tuple = ()
        dtype = np.float32

    class Box(Space):  # type: ignore
        def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
            self.low = low
            self.high = high
            self.shape = tuple(shape) if shape is not None else ()
            self.dtype = dtype

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {np.dtype(self.dtype).name})"
