"""
TerrainManager — minimal mirror of genesis_forge/managers/terrain_manager.py.

Out of scope for the fused hot path: the reference uses it at reset time
(random spawn positions) and for the optional ``base_height(terrain_manager=…)`` lookup.  What the hot
path needs from it is ``get_bounds`` (Python floats consumed by ``terminations.out_of_bounds``) and a
height query, provided here for flat terrain / a uniform height field without the bilinear sampler.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import gs
from .base import BaseManager


class TerrainManager(BaseManager):
    def __init__(self, env, terrain_attr: str = "terrain", bounds: Optional[tuple] = None, height: float = 0.0):
        super().__init__(env, "terrain")
        self._terrain_attr = terrain_attr
        self._bounds = bounds
        self._height = height

    def build(self):
        terrain = getattr(self.env, self._terrain_attr, None)
        if self._bounds is None:
            b = getattr(terrain, "bounds", None)
            self._bounds = tuple(b) if b is not None else (-50.0, 50.0, -50.0, 50.0)

    def get_bounds(self, subterrain: str | None = None) -> tuple[float, float, float, float]:
        """(x_min, x_max, y_min, y_max) as Python floats (terrain_manager.py:168-199)."""
        return self._bounds

    def get_terrain_height(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return torch.full_like(x, self._height)

    def generate_random_env_pos(self, envs_idx=None, subterrain=None, height_offset: float = 0.0, output=None):
        n = self.env.num_envs if envs_idx is None else len(envs_idx)
        x0, x1, y0, y1 = self._bounds
        pos = torch.empty(n, 3, device=gs.device)
        pos[:, 0].uniform_(x0, x1)
        pos[:, 1].uniform_(y0, y1)
        pos[:, 2] = self._height + height_offset
        return pos
