"""
Termination terms — same names, arguments and results as genesis_forge/mdp/terminations.py; each is a
descriptor for the fused ``gf_termination_step`` kernel and a directly callable function returning a
bool ``[N]`` tensor (see mdp/rewards.py for the pattern).
"""
from __future__ import annotations

import functools
import math

import torch

from .. import _native as nat
from ..managers._program import TermSpec, eval_termination_spec


def _entity(env, entity_attr, entity_manager):
    return entity_manager.entity if entity_manager is not None else getattr(env, entity_attr)


def _term(spec_fn):
    def deco(fn):
        def public(env, *args, **kwargs):
            return eval_termination_spec(env, spec_fn(env, *args, **kwargs))

        public.__name__ = fn.__name__
        public.__qualname__ = fn.__qualname__
        public.__doc__ = fn.__doc__
        public._gf_spec = spec_fn
        return public

    return deco


@functools.lru_cache(maxsize=64)
def tilt_threshold_sin(limit_angle_deg: float) -> tuple[float, float]:
    """Largest float32 ``x`` with ``asin(x) <= float32(radians(limit))`` and that float32 threshold.

    The reference tests ``torch.asin(clamp(|g_xy|, max=0.99)) > math.radians(limit)`` (terminations.py:64-71;
    the Python double is rounded to f32 by the comparison).  asin is monotone, so the test equals
    ``clamp(|g_xy|, max=0.99) > x``; evaluating asin only here, with the same torch CPU asin the reference's
    CPU path uses, means the device never computes asinf and cannot flip a mask by an ulp."""
    thr = torch.tensor(math.radians(limit_angle_deg), dtype=torch.float32)

    def ok(x: torch.Tensor) -> bool:  # asin(x) <= thr  → term does NOT fire at x
        return bool(torch.asin(x) <= thr)

    top = torch.tensor(0.99, dtype=torch.float32)
    if ok(top):
        return float(top), float(thr)       # never fires: clamp(...) > 0.99 is impossible
    zero = torch.tensor(0.0, dtype=torch.float32)
    if not ok(zero):
        return -1.0, float(thr)             # fires for every finite tilt
    lo, hi = int(zero.view(torch.int32)), int(top.view(torch.int32))  # positive floats order like their bits
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if ok(torch.tensor(mid, dtype=torch.int32).view(torch.float32)):
            lo = mid
        else:
            hi = mid
    return float(torch.tensor(lo, dtype=torch.int32).view(torch.float32)), float(thr)


@_term(lambda env: TermSpec(nat.GF_T_TIMEOUT))
def timeout(env):
    """``episode_length > max_episode_length`` (terminations.py:17-23)."""


def _spec_bad_orientation(env, limit_angle: float = 40.0, entity_attr: str = "robot", entity_manager=None, grace_steps: int = 0):
    x_thr, thr = tilt_threshold_sin(limit_angle)
    return TermSpec(nat.GF_T_BAD_ORIENTATION, p=[x_thr, thr], i=[int(grace_steps)], entity=_entity(env, entity_attr, entity_manager))


@_term(_spec_bad_orientation)
def bad_orientation(env, limit_angle=40.0, entity_attr="robot", entity_manager=None, grace_steps=0):
    """Tilt (from projected gravity) beyond ``limit_angle`` degrees, after a grace period (terminations.py:26-71)."""


def _spec_base_height_below(env, minimum_height: float = 0.05, entity_attr: str = "robot", entity_manager=None):
    return TermSpec(nat.GF_T_BASE_HEIGHT_BELOW, p=[float(minimum_height)], entity=_entity(env, entity_attr, entity_manager))


@_term(_spec_base_height_below)
def base_height_below_minimum(env, minimum_height=0.05, entity_attr="robot", entity_manager=None):
    """``base_z < minimum_height`` (terminations.py:74-99)."""


def _spec_out_of_bounds(env, terrain_manager, subterrain: str | None = None, border_margin: float = 0.5, entity_attr: str = "robot"):
    (x_min, x_max, y_min, y_max) = terrain_manager.get_bounds(subterrain)
    return TermSpec(nat.GF_T_OUT_OF_BOUNDS, p=[x_min + border_margin, x_max - border_margin, y_min + border_margin, y_max - border_margin],
                    entity=getattr(env, entity_attr))


@_term(_spec_out_of_bounds)
def out_of_bounds(env, terrain_manager, subterrain=None, border_margin=0.5, entity_attr="robot"):
    """Base position outside the terrain bounds minus a margin (terminations.py:102-137)."""


@_term(lambda _env, contact_manager, threshold=1.0, min_contacts=1: TermSpec(
    nat.GF_T_HAS_CONTACT, p=[float(threshold)], i=[0, int(min_contacts)], contact={0: contact_manager}))
def has_contact(_env, contact_manager, threshold=1.0, min_contacts=1):
    """At least ``min_contacts`` tracked links above the force threshold (terminations.py:139-155)."""


@_term(lambda _env, contact_manager, threshold=1.0: TermSpec(nat.GF_T_CONTACT_FORCE, p=[float(threshold)], contact={0: contact_manager}))
def contact_force(_env, contact_manager, threshold: float = 1.0):
    """Any tracked link above the force threshold (terminations.py:158-172)."""


@_term(lambda env, contact_manager, threshold=100.0, grace_steps=10: TermSpec(
    nat.GF_T_CONTACT_FORCE_GRACE, p=[float(threshold)], i=[0, int(grace_steps)], contact={0: contact_manager}))
def contact_force_with_grace_period(env, contact_manager, threshold: float = 100.0, grace_steps: int = 10):
    """Like ``contact_force`` but ignored for the first ``grace_steps`` steps of an episode (terminations.py:175-205)."""
