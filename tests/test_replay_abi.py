"""gf_replay_step's patch table through the raw C ABI (no launches: ops == NULL, so this runs without a GPU on BOTH libraries).

The table is what a recorded step applies natively before it replays its ops (include/gf_step.h: GfReplayPatch / GfReplay); the
product library and the oracle carry the same interpreter, and the host (_trace.py) relies on exactly these semantics."""
import ctypes as C
import os

import pytest

from genesis_forge_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libs(oracle_lib_path):
    out = []
    hip = C.CDLL(nat.lib_path())
    hip.gf_replay_step.restype = C.c_int
    hip.gf_replay_step.argtypes = [C.POINTER(nat.GfReplay), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    out.append(("gf", lambda r, act, par, n: hip.gf_replay_step(C.byref(r), act, par, n, None, None)))
    orc = C.CDLL(oracle_lib_path)
    orc.gfo_replay_step.restype = C.c_int
    orc.gfo_replay_step.argtypes = [C.POINTER(nat.GfReplay), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    out.append(("gfo", lambda r, act, par, n: orc.gfo_replay_step(C.byref(r), act, par, n, None)))
    return out


def _replay(patches, rng):
    tab = (nat.GfReplayPatch * max(1, len(patches)))(*patches)
    return nat.GfReplay(None, 0, len(patches), C.addressof(tab), C.addressof(rng) if rng is not None else None), tab


def test_every_patch_kind(oracle_lib_path):
    P = nat.GfReplayPatch
    for name, call in _libs(oracle_lib_path):
        act, obs, cmd = nat.GfActionArgs(), nat.GfObservationArgs(), nat.GfCommandArgs()
        scene, rst = nat.GfSynthSceneArgs(), nat.GfResetArgs()
        rng, tick = C.c_uint64(41), C.c_uint64(7)
        rotor = nat.GfRotor(0, 3, (C.c_void_p * 8)(0x1000, 0x2000, 0x3000))
        clock = nat.GfRingClock(0, 5)
        params = (C.c_void_p * 2)(0xAAA0, 0xBBB0)
        r, keep = _replay([
            P(nat.GF_PATCH_ACTIONS, 0, nat.field_addr(act, "actions_in"), None, None),
            P(nat.GF_PATCH_STREAM, 0, nat.field_addr(cmd, "stream"), None, None),
            P(nat.GF_PATCH_STREAM, 0, nat.field_addr(rst, "stream"), None, None),
            P(nat.GF_PATCH_COUNTER, 0, nat.field_addr(scene, "tick"), None, C.addressof(tick)),
            P(nat.GF_PATCH_ROTATE, 0, nat.field_addr(obs, "prev_obs"), nat.field_addr(obs, "obs"), C.addressof(rotor)),
            P(nat.GF_PATCH_PARAM, 1, nat.field_addr(act, "stats"), None, None),
            P(nat.GF_PATCH_COPY, 0, nat.field_addr(act, "stats_zero"), None, nat.field_addr(act, "stats")),
            P(nat.GF_PATCH_RING_SLOT, 0, nat.field_addr(obs, "history_ring"), None, C.addressof(clock)),
        ], rng)
        slots = []
        for step in range(7):
            assert call(r, 0xD000 + step, params, 2) == 0, name
            assert act.actions_in == 0xD000 + step
            assert (cmd.stream, rst.stream, rng.value) == (42 + 2 * step, 43 + 2 * step, 43 + 2 * step)      # draw order
            assert (scene.tick, tick.value) == (7 + step, 8 + step)
            assert (obs.prev_obs, obs.obs) == ((0x1000, 0x2000, 0x3000)[step % 3], (0x1000, 0x2000, 0x3000)[(step + 1) % 3])
            assert act.stats == 0xBBB0 and act.stats_zero == 0xBBB0
            slots.append(obs.history_ring - 1)
        assert slots == [0, 4, 3, 2, 1, 0, 4], "history ring slot = (H - calls % H) % H: newest-first is ascending from the slot"
        assert rotor.cur == 7 % 3 and clock.calls == 7


def test_patch_table_refusals(oracle_lib_path):
    P = nat.GfReplayPatch
    for name, call in _libs(oracle_lib_path):
        a = nat.GfActionArgs()
        rng = C.c_uint64(0)
        bad_rotor = nat.GfRotor(0, 0, (C.c_void_p * 8)())
        cases = [
            ([P(99, 0, nat.field_addr(a, "stats"), None, None)], rng, nat_err("GF_E_OPCODE")),
            ([P(nat.GF_PATCH_ACTIONS, 0, None, None, None)], rng, nat_err("GF_E_NULL")),
            ([P(nat.GF_PATCH_STREAM, 0, nat.field_addr(a, "stats"), None, None)], None, nat_err("GF_E_NULL")),       # no stream counter
            ([P(nat.GF_PATCH_ROTATE, 0, None, nat.field_addr(a, "stats"), C.addressof(bad_rotor))], rng, nat_err("GF_E_RANGE")),
            ([P(nat.GF_PATCH_PARAM, 3, nat.field_addr(a, "stats"), None, None)], rng, nat_err("GF_E_RANGE")),       # only 2 params passed
            ([P(nat.GF_PATCH_COPY, 0, nat.field_addr(a, "stats"), None, None)], rng, nat_err("GF_E_NULL")),
        ]
        params = (C.c_void_p * 2)(1, 2)
        for patches, r_, want in cases:
            r, keep = _replay(patches, r_)
            assert call(r, 0x10, params, 2) == want, (name, patches[0].kind)
        # an empty table is fine, a NULL descriptor is not
        r, keep = _replay([], rng)
        assert call(r, 0x10, params, 2) == 0


def nat_err(name):
    return {v: k for k, v in nat.GF_ERRORS.items()}[name]
