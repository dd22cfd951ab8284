"""gf_observe through the C ABI against gfo_observe (the oracle) on raw descriptors: frame widths around every vector width
and tile boundary (O = 1 … 255, O % 4 ∈ {0,1,2,3}), history lengths 1 … 5, partial tiles, every item kind — strided command
views, [N,D] rows, external columns, base position / quaternion, contact-force norms, body-frame vectors with a stale-quaternion
mask — with scale and noise.  Copies and scales are bit-exact; norms, rotations and noise use the same operation sequence on both
sides and are compared at 1e-6.  History is checked over three consecutive calls (ping-pong buffers), so a unit of the shift
that lands one float off shows up as a frame in the wrong slot.
"""
import ctypes as C
import random

import numpy as np
import pytest
import torch


def _case(seed):
    rnd = random.Random(seed)
    n = rnd.choice([1, 5, 63, 64, 65, 130, 200])
    D = rnd.choice([12, 12, 7, 28])
    L = rnd.choice([1, 4, 6])
    H = rnd.choice([1, 1, 2, 3, 5])
    target = rnd.choice([1, 2, 3, 4, 5, 7, 13, 31, 62, 64, 65, 130, 200, 255])
    items, width = [], 0
    pool = ["cmd_strided", "cmd_dense", "dof_pos", "dof_vel", "dof_force", "actions", "raw_actions", "ext", "base_pos", "base_quat",
            "norm", "ang", "lin", "grav"]
    widths = {"cmd_strided": 14, "cmd_dense": 3, "dof_pos": D, "dof_vel": D, "dof_force": D, "actions": D, "raw_actions": D, "base_pos": 3,
              "base_quat": 4, "norm": L, "ang": 3, "lin": 3, "grav": 3}
    n_ext = 0
    while width < target and len(items) < 24:
        room = target - width
        kind = rnd.choice(pool)
        if kind == "ext":
            if n_ext >= 8:
                continue
            w = min(room, rnd.choice([1, 2, 5, 17, 40]))
            n_ext += 1
        else:
            w = widths[kind]
            if w > room:
                kind, w = "ext", min(room, 40)
                if n_ext >= 8:
                    break
                n_ext += 1
        items.append((kind, w, rnd.choice([1.0, 1.0, 0.05, 2.0]), rnd.choice([0.0, 0.0, 0.0, 0.02])))
        width += w
    return dict(n=n, D=D, L=L, H=H, items=items, O=width, seed=seed)


def _run(case, dev, backend, calls=3):
    from genesis_forge_amd import _native as nat

    n, D, L, H, O = case["n"], case["D"], case["L"], case["H"], case["O"]
    g = torch.Generator().manual_seed(case["seed"])
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    quat = torch.randn(n, 4, generator=g)
    quat = (quat / quat.norm(dim=1, keepdim=True)).contiguous().to(dev)   # normalised on the host: both sides get the same bits
    bufs = dict(pos=mk(n, 3), quat=quat, lin=mk(n, 3), ang=mk(n, 3), dof_pos=mk(n, D), dof_vel=mk(n, D), dof_force=mk(n, D), targets=mk(n, D),
                raw=mk(n, D), gait=mk(n, 16), cmd=mk(n, 3), contacts=mk(n, L, 3), stale=mk(n, 4),
                mask=(torch.rand(n, generator=g) < 0.3).to(torch.uint8).to(dev), mask2=(torch.rand(n, generator=g) < 0.1).to(torch.uint8).to(dev))
    ext = []
    a = nat.GfObservationArgs()
    a.num_envs, a.num_dofs, a.num_items, a.obs_width, a.history_len = n, D, len(case["items"]), O, H
    a.entity.pos, a.entity.quat, a.entity.lin_vel, a.entity.ang_vel = (bufs[k].data_ptr() for k in ("pos", "quat", "lin", "ang"))
    a.dof_pos, a.dof_vel, a.dof_force, a.targets, a.env_actions = (bufs[k].data_ptr() for k in ("dof_pos", "dof_vel", "dof_force", "targets", "raw"))
    a.command[0].command, a.command[0].width, a.command[0].stride = bufs["gait"].data_ptr(), 14, 16
    a.command[1].command, a.command[1].width, a.command[1].stride = bufs["cmd"].data_ptr(), 3, 0
    a.contact[2].contacts, a.contact[2].num_links = bufs["contacts"].data_ptr(), L
    a.seed, a.stream, a.env_offset = 77, 5, 1000
    a.stale_quat, a.stale_mask, a.stale_mask2 = bufs["stale"].data_ptr(), bufs["mask"].data_ptr(), bufs["mask2"].data_ptr()
    ops = {"cmd_strided": (nat.GF_O_COMMAND, 0), "cmd_dense": (nat.GF_O_COMMAND, 1), "dof_pos": (nat.GF_O_DOF_POS, 0), "dof_vel": (nat.GF_O_DOF_VEL, 0),
           "dof_force": (nat.GF_O_DOF_FORCE, 0), "actions": (nat.GF_O_ACTIONS, 0), "raw_actions": (nat.GF_O_RAW_ACTIONS, 0),
           "base_pos": (nat.GF_O_BASE_POS, 0), "base_quat": (nat.GF_O_BASE_QUAT, 0), "norm": (nat.GF_O_CONTACT_FORCE_NORM, 2),
           "ang": (nat.GF_O_ANG_VEL_BODY, 0), "lin": (nat.GF_O_LIN_VEL_BODY, 0), "grav": (nat.GF_O_PROJ_GRAVITY, 0)}
    for i, (kind, w, scale, noise) in enumerate(case["items"]):
        it = a.items[i]
        if kind == "ext":
            t = mk(n, w)
            a.ext[len(ext)] = t.data_ptr()
            it.op, it.i0 = nat.GF_O_EXTERNAL, len(ext)
            ext.append(t)
        else:
            it.op, it.i0 = ops[kind]
        it.width, it.scale, it.noise = w, scale, noise
    out = [torch.full((n, O * H), float("nan"), device=dev) for _ in range(2)]
    out[1].zero_()   # the first call's history source
    res = []
    for c in range(calls):
        a.obs, a.prev_obs = out[c & 1].data_ptr(), (out[(c + 1) & 1].data_ptr() if H > 1 else None)
        a.stream = 5 + c
        bufs["dof_pos"].add_(1.0)   # a new frame every call
        backend.call("observe", a)
        res.append(out[c & 1].cpu().clone())
    return res


SEEDS = list(range(40))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_observe_hip_equals_oracle(hip_backend, oracle_lib_path, seed):
    from oracle_backend import OracleBackend

    case = _case(seed)
    got = _run(case, "cuda", hip_backend)
    torch.cuda.synchronize()
    want = _run(case, "cpu", OracleBackend(oracle_lib_path))
    O, H = case["O"], case["H"]
    exact = np.zeros(O, dtype=bool)
    col = 0
    for kind, w, scale, noise in case["items"]:
        exact[col:col + w] = kind not in ("norm", "ang", "lin", "grav") and noise == 0.0
        col += w
    for c, (a, b) in enumerate(zip(got, want)):
        assert not torch.isnan(a).any(), f"{case}: call {c} left part of the output unwritten"
        a, b = a.numpy().reshape(case["n"], H, O), b.numpy().reshape(case["n"], H, O)
        assert np.array_equal(a[:, :, exact], b[:, :, exact]), f"{case}: copied columns differ in call {c}"
        np.testing.assert_allclose(a, b, atol=1e-6, rtol=0, err_msg=f"{case}: call {c}")
    if H > 1:  # frame slot 1 of a call is frame slot 0 of the call before
        a = got[2].numpy().reshape(case["n"], H, O)
        p = got[1].numpy().reshape(case["n"], H, O)
        assert np.array_equal(a[:, 1:], p[:, :-1])


def _run_ring_slots(case, dev, backend, slots, calls):
    """A ring with more slots than frames (GfObservationArgs.ring_slots): `obs` is [N, slots, O], a call writes ONLY frame slot
    history_ring - 1 — every other float of the buffer must keep its value."""
    from genesis_forge_amd import _native as nat

    n, D, O, H = case["n"], case["D"], case["O"], case["H"]
    g = torch.Generator().manual_seed(case["seed"])
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    dof_pos, raw, cmd, ext = mk(n, D), mk(n, D), mk(n, 3), mk(n, max(1, O))
    a = nat.GfObservationArgs()
    a.num_envs, a.num_dofs, a.obs_width, a.history_len, a.ring_slots = n, D, O, H, slots
    a.dof_pos, a.env_actions = dof_pos.data_ptr(), raw.data_ptr()
    a.command[1].command, a.command[1].width, a.command[1].stride = cmd.data_ptr(), 3, 0
    a.ext[0] = ext.data_ptr()
    a.seed, a.stream, a.env_offset = 77, 5, 1000
    col, k = 0, 0
    for op, i0, w in ((nat.GF_O_DOF_POS, 0, D), (nat.GF_O_COMMAND, 1, 3), (nat.GF_O_RAW_ACTIONS, 0, D)):
        if col + w <= O:
            a.items[k].op, a.items[k].i0, a.items[k].width, a.items[k].scale, a.items[k].noise = op, i0, w, (0.5 if k == 1 else 1.0), (0.02 if k == 2 else 0.0)
            col += w; k += 1
    if col < O:
        ext_t = mk(n, O - col)
        a.ext[0] = ext_t.data_ptr()
        a.items[k].op, a.items[k].i0, a.items[k].width, a.items[k].scale = nat.GF_O_EXTERNAL, 0, O - col, 1.0
        k += 1
    a.num_items = k
    buf = torch.arange(n * slots * O, dtype=torch.float32).reshape(n, slots * O).to(dev)   # recognisable content everywhere
    a.obs = buf.data_ptr()
    res = []
    for c in range(calls):
        a.history_ring = (slots - 1 - (c * 3) % slots) + 1
        a.stream = 5 + c
        dof_pos.add_(1.0)
        backend.call("observe", a)
        res.append((a.history_ring - 1, buf.cpu().clone()))
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("seed,O,H,slots", [(0, 62, 5, 14), (1, 16, 5, 14), (2, 45, 3, 8), (3, 3, 2, 5), (4, 130, 2, 2), (5, 64, 4, 11)])
def test_observe_ring_with_more_slots_than_frames_hip_equals_oracle(hip_backend, oracle_lib_path, seed, O, H, slots):
    from oracle_backend import OracleBackend

    case = dict(n=[200, 65, 64, 130, 5, 257][seed], D=12, O=O, H=H, seed=seed)
    got = _run_ring_slots(case, "cuda", hip_backend, slots, calls=5)
    torch.cuda.synchronize()
    want = _run_ring_slots(case, "cpu", OracleBackend(oracle_lib_path), slots, calls=5)
    before = torch.arange(case["n"] * slots * O, dtype=torch.float32).reshape(case["n"], slots, O)
    for c, ((slot, a), (_s, b)) in enumerate(zip(got, want)):
        a3, b3 = a.reshape(case["n"], slots, O), b.reshape(case["n"], slots, O)
        np.testing.assert_allclose(a3.numpy(), b3.numpy(), atol=1e-6, rtol=0, err_msg=f"call {c}")
        untouched = [s for s in range(slots) if s != slot]
        assert torch.equal(a3[:, untouched], before[:, untouched]), f"call {c} wrote outside frame slot {slot}"
        assert not torch.equal(a3[:, slot], before[:, slot])
        before = a3.clone()


def test_ring_slots_refusals(oracle_backend):
    """ring_slots without a ring, fewer slots than frames, a slot beyond the buffer: GF_E_RANGE on both sides of the boundary
    (the HIP side's check is observe_prep, host code: tests/test_error_codes.py runs it without a GPU through gf_observe_check)."""
    from genesis_forge_amd import _native as nat

    a = nat.GfObservationArgs()
    buf, src = torch.zeros(4, 40), torch.zeros(4, 12)
    a.num_envs, a.num_dofs, a.obs_width, a.history_len, a.num_items = 4, 12, 4, 3, 1
    a.ext[0], a.obs = src.data_ptr(), buf.data_ptr()
    a.items[0].op, a.items[0].i0, a.items[0].width, a.items[0].scale = nat.GF_O_EXTERNAL, 0, 4, 1.0
    lib = oracle_backend.lib
    for ring, slots, rc in ((1, 10, 0), (10, 10, 0), (11, 10, -2), (0, 10, -2), (1, 2, -2), (3, 0, 0), (4, 0, -2)):
        a.history_ring, a.ring_slots = ring, slots
        a.prev_obs = None
        assert lib.gfo_observe(C.byref(a)) == rc, (ring, slots)


def test_observe_cases_cover_the_widths():
    cases = [_case(s) for s in SEEDS]
    assert {c["O"] % 4 for c in cases} == {0, 1, 2, 3}
    assert any(c["O"] < 4 and c["H"] > 1 for c in cases) and any(c["O"] > 128 for c in cases) and any(c["H"] == 5 for c in cases)
    assert any(c["n"] % 64 not in (0,) and c["n"] > 64 for c in cases)
