"""
TerrainManager — API mirror of genesis_forge/managers/terrain_manager.py.

The reference maps the terrain once at ``build()`` (origin, size, bounds, per-subterrain bounds, the height field scaled to
metres and transposed to ``[y, x]``, :281-359) and answers two kinds of query:

* ``get_terrain_height(x, y)`` (:100-166) — ten in-place normalisation launches, a ``[n,1,1,2]`` grid and an ``n``-way expanded
  ``F.grid_sample``.  Here it is one ``gf_terrain_height`` launch reading ``x`` / ``y`` in place (strided views such as
  ``pos[:, 0]`` included); ``rewards.base_height(terrain_manager=…)`` and the fused reset sample the same field inside their
  own kernels through :meth:`gf_view`, with the same device function.
* ``generate_random_positions`` / ``generate_random_env_pos`` (:168-279) — random points of the usable centre area of the
  terrain or of a named subterrain, z = terrain height + offset.  ``mdp.reset.randomize_terrain_position`` on an entity with
  masked setters never calls these per step: the draw, the height lookup and the pose write happen inside the masked reset
  (``GfResetArgs.spawn_*``).  Called directly they behave like the reference (``rand_like`` draws + one height launch).

Extension kept from the earlier flat-terrain mirror: ``bounds=`` / ``height=`` constructor arguments describe a terrain without
a Genesis entity (tests, flat benchmarks).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native as nat
from .. import gs
from .base import BaseManager


class TerrainManager(BaseManager):
    def __init__(self, env, terrain_attr: str = "terrain", bounds: Optional[tuple] = None, height: float = 0.0):
        super().__init__(env, "terrain")
        self._terrain_attr = terrain_attr
        self._terrain = None
        self._origin = (0.0, 0.0, float(height))
        self._bounds = tuple(bounds) if bounds is not None else None   # x_min, x_max, y_min, y_max
        self._size = (0.0, 0.0)
        self._subterrain_size = None
        self._subterrain_bounds: dict[str, tuple] = {}
        self._height_field: torch.Tensor | None = None                 # [H (y), W (x)] metres
        self._explicit_bounds = bounds is not None
        N = env.num_envs
        self._env_pos_buffer = torch.zeros((N, 3), device=gs.device, dtype=gs.tc_float)
        self._heights_buffer = torch.zeros(N, device=gs.device, dtype=gs.tc_float)
        self._args = nat.GfTerrainHeightArgs()

    # -- build: map the terrain (terrain_manager.py:281-359) ---------------------------------------------------------------
    def build(self):
        self._terrain = getattr(self.env, self._terrain_attr, None)
        terrain = self._terrain
        geoms = getattr(terrain, "geoms", None)
        if not geoms:
            # no mapped geometry (plane / stub entity): flat terrain at the configured height
            if self._bounds is None:
                b = getattr(terrain, "bounds", None)
                self._bounds = tuple(b) if b is not None else (-50.0, 50.0, -50.0, 50.0)
            self._size = (self._bounds[1] - self._bounds[0], self._bounds[3] - self._bounds[2])
            self._origin = (self._bounds[0], self._bounds[2], self._origin[2])
            return
        (geom,) = geoms
        morph = terrain.morph
        aabb, pos = geom.get_AABB(), geom.get_pos()
        if aabb.ndim == 3:      # parallel envs: the first env's values
            aabb = aabb[0]
        if pos.ndim == 2:
            pos = pos[0]
        n_sub = getattr(morph, "n_subterrains", None)
        if hasattr(morph, "pos") and n_sub is not None:
            self._origin = tuple(morph.pos)
            sx, sy = morph.subterrain_size
            self._size = (sx * n_sub[0], sy * n_sub[1])
            x_min, y_min = self._origin[0], self._origin[1]
            bounds = (x_min, x_min + self._size[0], y_min, y_min + self._size[1])
        else:
            (x_min, y_min, _), (x_max, y_max, _) = (tuple(float(v) for v in aabb[0]), tuple(float(v) for v in aabb[1]))
            self._origin = tuple(float(v) for v in pos)
            self._size = (x_max - x_min, y_max - y_min)
            bounds = (x_min, x_max, y_min, y_max)
        if not self._explicit_bounds:
            self._bounds = bounds
        if n_sub is not None:
            self._subterrain_size = tuple(morph.subterrain_size)
            self._subterrain_bounds = {}
            for ix in range(n_sub[0]):
                for iy in range(n_sub[1]):
                    name = morph.subterrain_types[ix][iy]
                    x0 = self._origin[0] + ix * self._subterrain_size[0]
                    y0 = self._origin[1] + iy * self._subterrain_size[1]
                    self._subterrain_bounds[name] = (x0, x0 + self._subterrain_size[0], y0, y0 + self._subterrain_size[1])
        if "height_field" in geom.metadata:
            hf = torch.as_tensor(geom.metadata["height_field"], device=gs.device, dtype=gs.tc_float)
            hf = hf * morph.vertical_scale
            self._height_field = hf.T.contiguous()   # (x, y) -> [y rows, x cols], the layout grid_sample indexes (:354-359)

    # -- queries -------------------------------------------------------------------------------------------------------------
    def get_bounds(self, subterrain: str | None = None) -> tuple[float, float, float, float]:
        """(x_min, x_max, y_min, y_max) of the terrain or of a named subterrain, Python floats (terrain_manager.py:92-98)."""
        if subterrain is not None and subterrain in self._subterrain_bounds:
            return self._subterrain_bounds[subterrain]
        return self._bounds

    def gf_view(self, view: nat.GfTerrainView | None = None) -> nat.GfTerrainView:
        """The terrain map as the kernels see it (include/gf_step.h GfTerrainView)."""
        v = view if view is not None else nat.GfTerrainView()
        x_min, x_max, y_min, y_max = self._bounds
        hf = self._height_field
        v.height_field = hf.data_ptr() if hf is not None else None
        v.rows, v.cols = (hf.shape[0], hf.shape[1]) if hf is not None else (0, 0)
        v.x_min, v.x_span = x_min, x_max - x_min     # the double difference, rounded once when stored as f32
        v.y_min, v.y_span = y_min, y_max - y_min
        v.origin_z = self._origin[2]
        return v

    def get_terrain_height(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Interpolated terrain height at world (x, y), shape (n,) (terrain_manager.py:100-166).  Like the reference the
        result is a view of a persistent buffer when ``n <= num_envs``."""
        n = x.shape[0]
        out = self._heights_buffer[:n] if n <= self._heights_buffer.shape[0] else torch.empty(n, device=gs.device, dtype=gs.tc_float)
        if x.dtype != gs.tc_float or x.dim() != 1:
            x = x.to(gs.tc_float).reshape(-1)
        if y.dtype != gs.tc_float or y.dim() != 1:
            y = y.to(gs.tc_float).reshape(-1)
        a = self._args
        a.num = n
        a.x, a.y = x.data_ptr(), y.data_ptr()
        a.x_stride, a.y_stride = (x.stride(0) if n > 1 else 1), (y.stride(0) if n > 1 else 1)
        self.gf_view(a.terrain)
        a.out = out.data_ptr()
        self._keep = (x, y)  # strided views of temporaries must outlive the launch
        self.env.backend.call("terrain_height", a, owner=None)
        return out

    def usable_area(self, usable_ratio: float = 0.5, subterrain: str | None = None) -> tuple[float, float, float, float]:
        """(x_min, x_max, y_min, y_max) of the centre ``usable_ratio`` of the terrain / subterrain (terrain_manager.py:211-233)."""
        bounds, size = self._bounds, self._size
        if subterrain is not None and subterrain in self._subterrain_bounds:
            size = self._subterrain_size
            bounds = self._subterrain_bounds[subterrain]
        (x_origin, _x_max, y_origin, _y_max) = bounds
        (x_size, y_size) = size
        buffer_x = (x_size - x_size * usable_ratio) / 2
        buffer_y = (y_size - y_size * usable_ratio) / 2
        return (x_origin + buffer_x, x_origin + x_size - buffer_x, y_origin + buffer_y, y_origin + y_size - buffer_y)

    def generate_random_positions(self, num: int | None = None, usable_ratio: float = 0.5, subterrain: str | None = None,
                                  height_offset: float = 0.1e-3, output: torch.Tensor | None = None,
                                  out_idx: torch.Tensor | None = None) -> torch.Tensor:
        """Random X/Y on the terrain (or subterrain) with Z at the terrain height there (terrain_manager.py:168-248)."""
        assert output is not None or num is not None, "Either output or num must be provided"
        if output is None:
            output = torch.zeros(num, 3, device=gs.device)
        if out_idx is None:
            out_idx = torch.arange(output.shape[0], device=gs.device)
        x_min, x_max, y_min, y_max = self.usable_area(usable_ratio, subterrain)
        output[out_idx, 0] = torch.rand_like(output[out_idx, 0]) * (x_max - x_min) + x_min
        output[out_idx, 1] = torch.rand_like(output[out_idx, 1]) * (y_max - y_min) + y_min
        heights = self.get_terrain_height(output[out_idx, 0], output[out_idx, 1])
        output[out_idx, 2] = heights + height_offset
        return output

    def generate_random_env_pos(self, envs_idx=None, usable_ratio: float = 0.5, subterrain: str | None = None,
                                height_offset: float = 0.1e-3) -> torch.Tensor:
        """One random position per env of ``envs_idx`` (terrain_manager.py:250-279)."""
        if envs_idx is None:
            envs_idx = torch.arange(self.env.num_envs, device=gs.device)
        self.generate_random_positions(output=self._env_pos_buffer, out_idx=envs_idx, usable_ratio=usable_ratio, subterrain=subterrain,
                                       height_offset=height_offset)
        return self._env_pos_buffer[envs_idx]
