"""TerrainManager against fixtures recorded from the reference's own TerrainManager
(tools/gen_golden.py gen_terrain: get_terrain_height = F.grid_sample bilinear/border/align_corners on the height field,
bounds and subterrain bounds, generate_random_positions with given draws)."""
import ast

import numpy as np
import pytest
import torch

import helpers

HEIGHT_TOL = 1e-6  # metres; torch's vectorised grid_sample may fuse the four taps, the kernel rounds every op once


class _Env:
    """The smallest env a TerrainManager needs."""

    def __init__(self, n, terrain, backend):
        self.num_envs, self.terrain, self.backend = n, terrain, backend
        self.managers = {}

    def add_manager(self, kind, m):
        self.managers.setdefault(kind, []).append(m)


def _terrain_manager(fix, dev):
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd.managers import TerrainManager
    from genesis_forge_amd.scene import SyntheticScene, morphs

    tk = ast.literal_eval(str(fix["terrain_kwargs"]))
    scene = SyntheticScene(seed=123)  # the field itself comes from the fixture, not from the scene's generator
    terrain = scene.add_entity(morph=morphs.Terrain(height_field=fix["height_field"], **tk))
    env = _Env(int(fix["n"]), terrain, nat.get_backend())
    tm = TerrainManager(env)
    tm.build()
    return tm


def _check(tm, fix, dev, monkeypatch):
    assert tuple(fix["bounds"]) == tuple(tm.get_bounds())
    for name, b in zip(fix["sub_names"], fix["sub_bounds"]):
        assert tuple(b) == tuple(tm.get_bounds(str(name)))
    x, y = torch.from_numpy(fix["x"]).to(dev), torch.from_numpy(fix["y"]).to(dev)
    h = tm.get_terrain_height(x, y).cpu().numpy()
    np.testing.assert_allclose(h, fix["heights"], atol=HEIGHT_TOL, rtol=0)
    # strided inputs, as rewards.base_height passes them (pos[:, 0], pos[:, 1])
    pos = torch.stack([x, y, torch.zeros_like(x)], dim=-1).contiguous()
    h2 = tm.get_terrain_height(pos[:, 0], pos[:, 1]).cpu().numpy()
    assert np.array_equal(h, h2)
    import philox
    n, seed = int(fix["n"]), int(fix["seed"])
    for k in range(4):
        ratio, sub, off = ast.literal_eval(str(fix[f"spawn{k}_cfg"]))
        u = philox.draws(seed, 100 + k, 4, n, 5)
        calls = []

        def fake_rand_like(t, *a, **kw):
            calls.append(1)
            return torch.from_numpy(np.ascontiguousarray(u[:, len(calls) - 1])).to(t.device).reshape(t.shape)

        monkeypatch.setattr(torch, "rand_like", fake_rand_like)
        p = tm.generate_random_positions(num=n, usable_ratio=ratio, subterrain=sub, height_offset=off).cpu().numpy()
        monkeypatch.undo()
        np.testing.assert_allclose(p, fix[f"spawn{k}_pos"], atol=2e-6, rtol=0, err_msg=f"spawn case {k}")


def test_terrain_manager_matches_reference_cpu_oracle(oracle_backend, monkeypatch):
    fix = helpers.load("terrain")
    _check(_terrain_manager(fix, "cpu"), fix, "cpu", monkeypatch)


@pytest.mark.gpu
def test_terrain_manager_matches_reference_hip(hip_backend, monkeypatch):
    fix = helpers.load("terrain")
    _check(_terrain_manager(fix, "cuda"), fix, "cuda", monkeypatch)


def _flat_manager():
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd.managers import TerrainManager
    from genesis_forge_amd.scene import SyntheticScene, morphs

    scene = SyntheticScene()
    env = _Env(8, scene.add_entity(morphs.Plane()), nat.get_backend())
    tm = TerrainManager(env, height=0.125)
    tm.build()
    return tm


def test_no_height_field_returns_origin_height(oracle_backend):
    tm = _flat_manager()
    h = tm.get_terrain_height(torch.randn(8), torch.randn(8))
    assert torch.equal(h, torch.full((8,), 0.125))
    assert tm.get_bounds() == (-50.0, 50.0, -50.0, 50.0)


@pytest.mark.gpu
def test_hip_equals_oracle_on_random_queries(hip_backend, oracle_lib_path):
    """Bit-exact: the HIP sampler and the oracle restate the same operation sequence."""
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    fix = helpers.load("terrain")
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(50_000, generator=g) * 14.0 - 4.0)
    y = (torch.rand(50_000, generator=g) * 13.0 + 0.5)
    tm = _terrain_manager(fix, "cuda")
    tm._heights_buffer = torch.zeros(50_000, device="cuda")
    h_hip = tm.get_terrain_height(x.cuda(), y.cuda()).cpu()
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    try:
        tm_cpu = _terrain_manager(fix, "cpu")
        tm_cpu._heights_buffer = torch.zeros(50_000)
        h_cpu = tm_cpu.get_terrain_height(x, y).clone()
    finally:
        nat.set_backend(None)
        gs.set_device("cuda:0")
    assert torch.equal(h_hip, h_cpu)
