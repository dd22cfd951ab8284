"""Error behaviour at the C boundary: a malformed descriptor is refused with the documented GF_E_* code before anything is
launched — the same code from the oracle twin (gfo_*, CPU) and from the library (gf_*, GPU).  The cases follow the header's
contract (include/gf_step.h:44-51): NULL required pointers, counts out of range, unknown opcodes, terms that address an
unbound view, layouts the kernels do not take."""
import ctypes as C

import pytest
import torch

E_NULL, E_RANGE, E_OPCODE, E_SLOT, E_UNSUPPORTED = -1, -2, -3, -4, -5


def _cases(dev):
    """[(phase function, descriptor, expected code, what is wrong)] on buffers of device `dev`."""
    from genesis_forge_amd import _native as nat

    n, D = 8, 12
    f = lambda *s: torch.zeros(*s, device=dev)
    keep = dict(quat=f(n, 4), pos=f(n, 3), lin=f(n, 3), ang=f(n, 3), dof=f(n, D), act=f(n, D), last=f(n, D), tgt=f(n, D), rew=f(n), secs=f(n),
                sums=f(4, n), obs=f(n, 16), cmd=f(n, 3), ep=torch.zeros(n, dtype=torch.int32, device=dev),
                term=torch.zeros(n, dtype=torch.uint8, device=dev), trunc=torch.zeros(n, dtype=torch.uint8, device=dev),
                scale=f(D), state=f(n, 16), sel=torch.zeros(n, dtype=torch.int64, device=dev))
    p = {k: v.data_ptr() for k, v in keep.items()}
    out = []

    def term_args():
        a = nat.GfTerminationArgs()
        a.num_envs, a.num_terms = n, 1
        a.terms[0].op = nat.GF_T_TIMEOUT
        a.episode_length, a.terminated, a.truncated = p["ep"], p["term"], p["trunc"]
        return a

    a = term_args(); a.terminated = None
    out.append(("termination_step", a, E_NULL, "no terminated buffer"))
    a = term_args(); a.num_terms = nat.GF_MAX_TERM_TERMS + 1
    out.append(("termination_step", a, E_RANGE, "too many terms"))
    a = term_args(); a.terms[0].op = 9999
    out.append(("termination_step", a, E_OPCODE, "unknown termination opcode"))
    a = term_args(); a.terms[0].op = nat.GF_T_CONTACT_FORCE; a.terms[0].i[0] = 1
    out.append(("termination_step", a, E_SLOT, "contact term on an unbound view"))
    a = term_args(); a.terms[0].op = nat.GF_T_BAD_ORIENTATION
    out.append(("termination_step", a, E_NULL, "orientation term without a quaternion"))

    def reward_args():
        a = nat.GfRewardArgs()
        a.num_envs, a.num_dofs, a.num_terms, a.mode, a.dt = n, D, 1, nat.GF_REWARD_MODE_STEP, 0.02
        a.terms[0].op, a.terms[0].w = nat.GF_R_IS_ALIVE, 1.0
        a.terminated, a.reward, a.episode_seconds = p["term"], p["rew"], p["secs"]
        return a

    a = reward_args(); a.reward = None
    out.append(("reward_step", a, E_NULL, "no reward buffer"))
    a = reward_args(); a.terms[0].op = 9999
    out.append(("reward_step", a, E_OPCODE, "unknown reward opcode"))
    a = reward_args(); a.num_terms = nat.GF_MAX_TERMS + 1
    out.append(("reward_step", a, E_RANGE, "too many terms"))
    a = reward_args(); a.terms[0].op = nat.GF_R_HAS_CONTACT; a.terms[0].i[0] = 2
    out.append(("reward_step", a, E_SLOT, "contact term on an unbound view"))
    a = reward_args(); a.terms[0].op = nat.GF_R_LIN_VEL_Z_L2
    out.append(("reward_step", a, E_NULL, "velocity term without entity buffers"))

    def obs_args():
        a = nat.GfObservationArgs()
        a.num_envs, a.num_dofs, a.num_items, a.obs_width, a.history_len = n, D, 1, D, 1
        a.items[0].op, a.items[0].width, a.items[0].scale = nat.GF_O_DOF_POS, D, 1.0
        a.dof_pos, a.obs = p["dof"], p["obs"]
        return a

    a = obs_args(); a.obs = None
    out.append(("observe", a, E_NULL, "no output buffer"))
    a = obs_args(); a.obs_width = D + 1
    out.append(("observe", a, E_RANGE, "item widths do not add up to the frame width"))
    a = obs_args(); a.items[0].op = 9999
    out.append(("observe", a, E_OPCODE, "unknown item opcode"))
    a = obs_args(); a.history_len = 3
    out.append(("observe", a, E_NULL, "history without a previous buffer"))
    a = obs_args(); a.items[0].op = nat.GF_O_COMMAND; a.items[0].i0 = 1; a.items[0].width = 3; a.obs_width = 3
    out.append(("observe", a, E_SLOT, "command item on an unbound view"))
    a = obs_args(); a.num_items = nat.GF_MAX_OBS_ITEMS + 1
    out.append(("observe", a, E_RANGE, "too many items"))

    def cmd_args():
        a = nat.GfCommandArgs()
        a.num_envs, a.num_ranges, a.mode, a.resample_steps = n, 3, nat.GF_CMD_STEP, 10
        a.episode_length, a.command = p["ep"], p["cmd"]
        return a

    a = cmd_args(); a.command = None
    out.append(("command_step", a, E_NULL, "no command buffer"))
    a = cmd_args(); a.num_ranges = nat.GF_MAX_RANGES + 1
    out.append(("command_step", a, E_RANGE, "too many ranges"))
    a = cmd_args(); a.mode = nat.GF_CMD_MASKED
    out.append(("command_step", a, E_NULL, "masked resample without a mask"))

    def gait_args():
        a = nat.GfGaitArgs()
        a.num_envs, a.mode, a.resample_steps, a.num_gaits = n, nat.GF_CMD_STEP, 10, 1
        a.episode_length, a.state, a.selected = p["ep"], p["state"], p["sel"]
        return a

    a = gait_args(); a.state = None
    out.append(("gait_step", a, E_NULL, "no state rows"))
    a = gait_args(); a.num_gaits = nat.GF_MAX_GAITS + 1
    out.append(("gait_step", a, E_RANGE, "more gaits than the table holds"))
    a = gait_args(); a.resample_steps = 0
    out.append(("gait_step", a, E_RANGE, "resample period of zero steps"))

    a = nat.GfResetArgs(); a.num_envs, a.num_dofs = n, D
    out.append(("masked_reset", a, E_NULL, "no mask"))
    a = nat.GfResetArgs(); a.num_envs, a.num_dofs, a.mask, a.num_contact = n, D, p["term"], nat.GF_MAX_CONTACT_VIEWS + 1
    out.append(("masked_reset", a, E_RANGE, "too many air-time managers"))
    a = nat.GfResetArgs(); a.num_envs, a.num_dofs, a.mask, a.env_actions = n, D, p["term"], p["act"]
    out.append(("masked_reset", a, E_NULL, "actions without last_actions"))

    def action_args():
        a = nat.GfActionArgs()
        a.num_envs, a.num_dofs, a.mode = n, D, nat.GF_ACTION_POSITION
        a.actions_in, a.scale, a.offset, a.clip_lo, a.clip_hi, a.targets = p["act"], p["scale"], p["scale"], p["scale"], p["scale"], p["tgt"]
        return a

    a = action_args(); a.actions_in = None
    out.append(("action_step", a, E_NULL, "no input actions"))
    a = action_args(); a.num_dofs = 0
    out.append(("action_step", a, E_RANGE, "no DOFs"))
    return out, keep


def _empty_cases(dev):
    """Well-formed descriptors over ZERO envs: accepted, nothing to do (GF_OK), on both sides."""
    cases, keep = _cases(dev)
    seen, out = set(), []
    for fn, a, _want, _what in cases:
        if fn in seen:
            continue
        seen.add(fn)
    # rebuild one valid descriptor per phase from the malformed ones' builders: undo the single defect
    from genesis_forge_amd import _native as nat

    p = {k: v.data_ptr() for k, v in keep.items()}
    t = nat.GfTerminationArgs(); t.num_terms = 1; t.terms[0].op = nat.GF_T_TIMEOUT
    t.episode_length, t.terminated, t.truncated = p["ep"], p["term"], p["trunc"]
    r = nat.GfRewardArgs(); r.num_dofs, r.num_terms, r.mode, r.dt = 12, 1, nat.GF_REWARD_MODE_STEP, 0.02
    r.terms[0].op, r.terms[0].w = nat.GF_R_IS_ALIVE, 1.0
    r.terminated, r.reward, r.episode_seconds = p["term"], p["rew"], p["secs"]
    o = nat.GfObservationArgs(); o.num_dofs, o.num_items, o.obs_width, o.history_len = 12, 1, 12, 1
    o.items[0].op, o.items[0].width, o.items[0].scale = nat.GF_O_DOF_POS, 12, 1.0
    o.dof_pos, o.obs = p["dof"], p["obs"]
    c = nat.GfCommandArgs(); c.num_ranges, c.mode, c.resample_steps = 3, nat.GF_CMD_STEP, 10
    c.episode_length, c.command = p["ep"], p["cmd"]
    g = nat.GfGaitArgs(); g.mode, g.resample_steps, g.num_gaits = nat.GF_CMD_STEP, 10, 1
    g.episode_length, g.state, g.selected = p["ep"], p["state"], p["sel"]
    m = nat.GfResetArgs(); m.num_dofs, m.mask = 12, p["term"]
    ac = nat.GfActionArgs(); ac.num_dofs, ac.mode = 12, nat.GF_ACTION_POSITION
    ac.actions_in, ac.scale, ac.offset, ac.clip_lo, ac.clip_hi, ac.targets = p["act"], p["scale"], p["scale"], p["scale"], p["scale"], p["tgt"]
    for fn, a in (("termination_step", t), ("reward_step", r), ("observe", o), ("command_step", c), ("gait_step", g), ("masked_reset", m), ("action_step", ac)):
        a.num_envs = 0
        out.append((fn, a, 0, "zero envs"))
    return out, keep


def _codes(lib, prefix, cases, stream):
    from genesis_forge_amd import _native as nat

    got = []
    for fn, a, _want, _what in cases:
        f = getattr(lib, prefix + fn)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(nat.PHASE_FUNCS[fn])] + ([C.c_void_p] if stream is not None else [])
        got.append(f(C.byref(a), stream) if stream is not None else f(C.byref(a)))
    return got


def test_oracle_refuses_malformed_descriptors(oracle_lib_path):
    cases, _keep = _cases("cpu")
    empty, _keep2 = _empty_cases("cpu")
    cases = cases + empty
    got = _codes(C.CDLL(oracle_lib_path), "gfo_", cases, None)
    for (fn, _a, want, what), rc in zip(cases, got):
        assert rc == want, f"gfo_{fn}: {what}: returned {rc}, the header documents {want}"


@pytest.mark.gpu
def test_library_refuses_malformed_descriptors_like_the_oracle(hip_backend, oracle_lib_path):
    cases, _keep = _cases("cuda")
    empty, _keep2 = _empty_cases("cuda")
    cases = cases + empty
    got = _codes(hip_backend.lib, "gf_", cases, C.c_void_p(0))
    torch.cuda.synchronize()
    for (fn, _a, want, what), rc in zip(cases, got):
        assert rc == want, f"gf_{fn}: {what}: returned {rc}, the header documents {want}"
    assert all(t.eq(0).all() for t in _keep.values()), "a refused call must not have touched a buffer"
