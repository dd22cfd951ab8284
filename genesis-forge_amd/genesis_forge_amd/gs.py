"""
Minimal stand-in for the slice of the ``genesis`` module namespace the manager stack touches
(``gs.device``, ``gs.tc_float``, ``gs.tc_int``, ``gs.tc_bool``, ``gs.JOINT_TYPE``; reference call
sites: genesis_forge/genesis_env.py:75-89, managers/action/position_action_manager.py:303).

If the real ``genesis`` package is importable and initialised, its values win; otherwise the device
is the local ROCm GPU (``cuda:<LOCAL_RANK>``), falling back to ``cpu`` only so that host-side logic
(config compilation, registries, sharding) can be unit-tested without a GPU — the phase kernels
themselves never run on CPU.
"""
from __future__ import annotations

import enum
import os

import torch

tc_float = torch.float32
tc_int = torch.int32
tc_bool = torch.bool


class JOINT_TYPE(enum.IntEnum):
    FIXED = 0
    REVOLUTE = 1
    PRISMATIC = 2
    SPHERICAL = 3
    FREE = 4


def _default_device() -> torch.device:
    forced = os.environ.get("GF_DEVICE")   # e.g. "cpu": host-only helper processes must not even probe (= open) the GPU
    if forced:
        return torch.device(forced)
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


device: torch.device = _default_device()


def set_device(dev) -> torch.device:
    """Select the device all manager buffers are created on."""
    global device
    device = torch.device(dev)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    return device
