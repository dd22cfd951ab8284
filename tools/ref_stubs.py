"""
Stub modules that let the REFERENCE package (/root/reference/genesis_forge) be imported in the build
container, where genesis / gstaichi / gymnasium / tensordict / skrl / hid are not installed
(the reference's own docs build mocks the same imports, docs/conf.py:13-24).

Used only by tools/gen_golden.py, here, to generate tests/golden/*.npz.  Nothing from the reference is
copied: its code is imported from where it lies and executed; only inputs and outputs are recorded.

The quaternion helpers live in un-vendored genesis-world (parity unpinned upstream); they are defined
here from the mathematical definition with one torch op per arithmetic operation, the same operation
order oracle/gf_oracle.c uses.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def inv_quat(q):
    out = q.clone()
    out[..., 1:] = -out[..., 1:]
    return out


def transform_by_quat(v, q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    vx, vy, vz = v[..., 0], v[..., 1], v[..., 2]
    t0 = (y * vz - z * vy) * 2.0
    t1 = (z * vx - x * vz) * 2.0
    t2 = (x * vy - y * vx) * 2.0
    o0 = (vx + w * t0) + (y * t2 - z * t1)
    o1 = (vy + w * t1) + (z * t0 - x * t2)
    o2 = (vz + w * t2) + (x * t1 - y * t0)
    return torch.stack([o0, o1, o2], dim=-1)


def install():
    from genesis_forge_amd import compat, gs as my_gs
    from genesis_forge_amd import scene as my_scene
    from genesis_forge_amd.mdp.reset import xyz_to_quat

    my_gs.set_device("cpu")
    g = compat.make_genesis_shim()
    g.device = torch.device("cpu")
    sys.modules["genesis"] = g

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class RigidEntity:  # type annotations only
        pass

    class RigidLink:
        pass

    class Camera:
        pass

    mod("genesis.engine")
    mod("genesis.engine.entities", RigidEntity=RigidEntity)
    mod("genesis.engine.entities.rigid_entity")
    mod("genesis.engine.entities.rigid_entity.rigid_link", RigidLink=RigidLink)
    mod("genesis.utils")
    mod("genesis.utils.geom", transform_by_quat=transform_by_quat, inv_quat=inv_quat, xyz_to_quat=xyz_to_quat,
        ti_inv_transform_by_quat=lambda v, q: transform_by_quat(v, inv_quat(q)))
    mod("genesis.vis")
    mod("genesis.vis.camera", Camera=Camera)

    class _NdArray:
        def __call__(self, *a, **k):
            return None

    ti_types = types.SimpleNamespace(ndarray=_NdArray())

    # Just enough of the Taichi surface for kernel_get_contact_forces (managers/contact/kernel.py:5-90) to run
    # as ordinary Python over torch tensors: a serial ndrange, f32 vectors, ti.static as identity.  Serial
    # execution fixes the order of the kernel's atomic += (contact-slot order), which the oracle follows.
    import itertools

    class Vec:
        def __init__(self, vals):
            self.v = [np.float32(x) for x in vals]

        def __getitem__(self, i):
            return self.v[i]

        def __setitem__(self, i, x):
            self.v[i] = np.float32(float(x))

        def __neg__(self):
            return Vec([-x for x in self.v])

        def __len__(self):
            return len(self.v)

    class _VectorNS:
        @staticmethod
        def zero(dtype, n):
            return Vec([0.0] * n)

    def ti_inv_transform_by_quat(v, q):
        w, a, b, c = q[0], -q[1], -q[2], -q[3]
        t0 = (b * v[2] - c * v[1]) * np.float32(2.0)
        t1 = (c * v[0] - a * v[2]) * np.float32(2.0)
        t2 = (a * v[1] - b * v[0]) * np.float32(2.0)
        return Vec([(v[0] + w * t0) + (b * t2 - c * t1), (v[1] + w * t1) + (c * t0 - a * t2), (v[2] + w * t2) + (a * t1 - b * t0)])

    sys.modules["genesis.utils.geom"].ti_inv_transform_by_quat = ti_inv_transform_by_quat
    def ti_kernel(f):
        """`@ti.kernel` → the function itself; for large batches the env axis is cut into chunks that run in forked worker processes
        (GF_TI_JOBS, default: the CPU count).  The kernel's outermost index is the env (kernel.py:35-37) and nothing crosses envs, so
        every env still sees its contact slots in slot order — the order of the kernel's atomic += — and the result is the serial
        run's, bit for bit (checked: `tools/gen_golden.py check_ti_parallel`).  ≈ 45 min → ≈ 7 min for the 65 536-env fixture."""
        import functools

        @functools.wraps(f)
        def run(*args):
            import torch

            n = int(args[7].shape[0])   # output_forces: (n_envs, n_target_links, 3)
            jobs = int(os.environ.get("GF_TI_JOBS", os.cpu_count() or 1))
            if jobs <= 1 or n < 2048:
                return f(*args)
            import multiprocessing as mp

            per_env = [i for i, a in enumerate(args) if hasattr(a, "shape") and len(a.shape) >= 2 and int(a.shape[0]) == n and i not in (5, 6)]
            outs = (7, 8, 9)
            bounds = [(n * j // jobs, n * (j + 1) // jobs) for j in range(jobs)]
            ctx = mp.get_context("fork")
            pipes, procs = [], []
            for lo, hi in bounds:
                rd, wr = ctx.Pipe(duplex=False)
                pid = os.fork()
                if pid == 0:
                    try:
                        rd.close()
                        torch.set_num_threads(1)
                        sub = [a[lo:hi].clone() if i in per_env else a for i, a in enumerate(args)]
                        f(*sub)
                        wr.send([sub[i].numpy() if hasattr(sub[i], "numpy") else np.asarray(sub[i]) for i in outs])
                        wr.close()
                    finally:
                        os._exit(0)
                wr.close()
                pipes.append(rd); procs.append(pid)
            for (lo, hi), rd, pid in zip(bounds, pipes, procs):
                got = rd.recv()
                os.waitpid(pid, 0)
                for i, arr in zip(outs, got):
                    dst = args[i]
                    if hasattr(dst, "numpy"):
                        dst[lo:hi] = torch.from_numpy(arr)
                    else:
                        dst[lo:hi] = arr
            return None

        return run

    mod("gstaichi", kernel=ti_kernel, types=ti_types, i32=int, f32=np.float32, Vector=_VectorNS, static=lambda x: x,
        ndrange=lambda *dims: itertools.product(*[range(int(d)) for d in dims]))

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Space:
        pass

    spaces = mod("gymnasium.spaces", Box=Box, Space=Space)
    mod("gymnasium", spaces=spaces)

    class TensorDict(dict):
        def __init__(self, d=None, device=None, batch_size=None):
            super().__init__(d or {})

        def to(self, *a, **k):
            return self

    mod("tensordict", TensorDict=TensorDict)
    mod("skrl")
    mod("skrl.envs")
    mod("skrl.envs.wrappers")
    mod("skrl.envs.wrappers.torch")

    class _SkrlWrapper:
        def __init__(self, env):
            self._env = env

    mod("skrl.envs.wrappers.torch.base", Wrapper=_SkrlWrapper)
    mod("hid")
    sys.path.insert(0, "/root/reference")
    return my_scene
