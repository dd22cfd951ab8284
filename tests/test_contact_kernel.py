"""ContactManager.step against the reference on ARBITRARY contact arrays (tests/golden/contact_kernel.npz, recorded by
tools/gen_golden.py from the reference's ContactManager + its Taichi kernel executed from source): link ids over every link
on both sides (force and reaction branches, kernel.py:74-78), empty slots, NaN / ±Inf forces (contact_manager.py:399-403),
with-filters on another entity and on own links (kernel.py:47-57), air time (contact_manager.py:434-477).

On the GPU the three managers also run as ONE launch (gf_run_ops folds consecutive contact ops over the same scene arrays)
and must give the same bits as three separate launches."""
import ast
import ctypes as C

import numpy as np
import pytest
import torch

import helpers


def _env(n, Cn, dev):
    from genesis_forge_amd import ManagedEnvironment
    from genesis_forge_amd.managers import ContactManager
    from genesis_forge_amd.scene import SyntheticScene, morphs

    fix = helpers.load("contact_kernel")
    cases = ast.literal_eval(str(fix["cases"]))

    class E(ManagedEnvironment):
        def __init__(self):
            super().__init__(num_envs=n, dt=1 / 50, max_episode_length_sec=20)
            self.scene = SyntheticScene(dt=self.dt, max_collision_pairs=Cn)
            self.terrain = self.scene.add_entity(morphs.Plane())
            self.robot = self.scene.add_entity(morphs.URDF(file="go2"))

        def config(self):
            self.cms = [ContactManager(self, **kw) for kw in cases]

    env = E()
    env.build()
    return env, fix


def _load_step(env, fix, t, dev):
    sc = env.scene
    sc.contact_force[:] = torch.from_numpy(fix["force"][t]).to(dev)
    sc.contact_pos[:] = torch.from_numpy(fix["position"][t]).to(dev)
    sc.link_a[:] = torch.from_numpy(fix["link_a"][t]).to(dev)
    sc.link_b[:] = torch.from_numpy(fix["link_b"][t]).to(dev)
    sc.links_quat[:] = torch.from_numpy(fix["links_quat"][t]).to(dev)


def _check(env, fix, t):
    for m, cm in enumerate(env.cms):
        np.testing.assert_allclose(cm.contacts.cpu().numpy(), fix[f"m{m}_contacts"][t], atol=1e-5, rtol=0, err_msg=f"manager {m} forces, step {t}")
        np.testing.assert_allclose(cm.contact_positions.cpu().numpy(), fix[f"m{m}_contact_positions"][t], atol=1e-5, rtol=0,
                                   err_msg=f"manager {m} positions, step {t}")
        if cm.last_air_time is not None:
            got = np.stack([x.cpu().numpy() for x in (cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time)])
            np.testing.assert_allclose(got, fix[f"m{m}_air"][t], atol=1e-6, rtol=0, err_msg=f"manager {m} air time, step {t}")


def _run(dev):
    fix0 = helpers.load("contact_kernel")
    env, fix = _env(int(fix0["n"]), int(fix0["C"]), dev)
    for t in range(int(fix["steps"])):
        _load_step(env, fix, t, dev)
        env.stats.clear(env.backend)
        for cm in env.cms:
            cm.step()
        _check(env, fix, t)
        flags = env.stats.snapshot().wait().contact_flags
        assert flags == (1 if t == 2 else 0), "the non-finite force flag is raised exactly on the step that has NaN / Inf forces"


def test_contact_managers_match_reference_cpu_oracle(oracle_backend):
    _run("cpu")


@pytest.mark.gpu
def test_contact_managers_match_reference_hip(hip_backend):
    _run("cuda")


@pytest.mark.gpu
def test_folded_launch_equals_separate_launches(hip_backend):
    """gf_run_ops with three consecutive contact ops = one launch of the multi-manager kernel; same bits as three launches."""
    from genesis_forge_amd import _native as nat

    fix0 = helpers.load("contact_kernel")
    env, fix = _env(int(fix0["n"]), int(fix0["C"]), "cuda")
    ops = (nat.GfOp * 3)()
    for t in range(int(fix["steps"])):
        _load_step(env, fix, t, "cuda")
        air_before = [[x.clone() for x in (cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time)]
                      if cm.last_air_time is not None else None for cm in env.cms]
        for cm in env.cms:
            cm.step()
        separate = [(cm.contacts.clone(), cm.contact_positions.clone(),
                     [x.clone() for x in (cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time)]
                     if cm.last_air_time is not None else None) for cm in env.cms]
        for cm, before in zip(env.cms, air_before):  # rewind the air-time state, poison the outputs
            cm.contacts.fill_(float("nan"))
            cm.contact_positions.fill_(float("nan"))
            if before is not None:
                for dst, src in zip((cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time), before):
                    dst.copy_(src)
        for k, cm in enumerate(env.cms):
            ops[k].phase, ops[k].args = nat.GF_PHASE_CONTACT, C.addressof(cm._args)
        env.backend.run_ops(ops, 3)
        torch.cuda.synchronize()
        for cm, (f, p, air) in zip(env.cms, separate):
            assert torch.equal(cm.contacts, f) and torch.equal(cm.contact_positions, p)
            if air is not None:
                for got, want in zip((cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time), air):
                    assert torch.equal(got, want)
        _check(env, fix, t)


@pytest.mark.gpu
def test_contact_step_as_first_phase_of_the_fused_launch_raw_abi(hip_backend):
    """gf_post_physics_step_contacts through the raw ABI on the reference-recorded contact arrays: link ids on both sides, empty slots,
    NaN / Inf forces, a with-filter on another entity and one on own links, air time — the three managers run as the first phase of a
    fused post-physics launch (of an unrelated Go2 env of the same size: any fusable step will do) and must leave exactly what three
    gf_contact_step launches leave, which is what the reference left (fixture)."""
    import envs
    from genesis_forge_amd import _native as nat

    fix0 = helpers.load("contact_kernel")
    n = int(fix0["n"])
    env, fix = _env(n, int(fix0["C"]), "cuda")
    host = envs.Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3, contacts=True, scene_kwargs=dict(ang_noise=0.3, seed=3))
    host.build(); host.seed(3); host.reset()
    g = torch.Generator().manual_seed(0)
    for _ in range(6):
        host.step(torch.randn(n, 12, generator=g).to("cuda"))
    refs = host._trace.post_refs
    assert refs is not None
    hip_backend.set_option(nat.GF_OPT_FOLD_CONTACT, 2)
    try:
        for t in range(int(fix["steps"])):
            _load_step(env, fix, t, "cuda")
            air_before = [[x.clone() for x in (cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time)]
                          if cm.last_air_time is not None else None for cm in env.cms]
            for cm in env.cms:
                cm.step()
            separate = [(cm.contacts.clone(), cm.contact_positions.clone(), cm._contact_position_counts.clone(),
                         [x.clone() for x in (cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time)]
                         if cm.last_air_time is not None else None) for cm in env.cms]
            for cm, before in zip(env.cms, air_before):   # rewind the air-time state, poison the outputs
                cm.contacts.fill_(float("nan")); cm.contact_positions.fill_(float("nan")); cm._contact_position_counts.fill_(-1.0)
                if before is not None:
                    for dst, src in zip((cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time), before):
                        dst.copy_(src)
            descs = []
            for cm in env.cms:
                a = nat.GfContactArgs()
                C.memmove(C.addressof(a), C.addressof(cm._args), C.sizeof(a))
                a.stats = None   # (the managers of another env: no statistics block of the host env's step)
                descs.append(a)
            rc = hip_backend.post_step_contacts(refs, descs)
            assert rc == 0, f"not folded: {nat.GF_ERRORS.get(rc, rc)}"
            torch.cuda.synchronize()
            for cm, (f, p, c, air) in zip(env.cms, separate):
                assert torch.equal(cm.contacts, f) and torch.equal(cm.contact_positions, p) and torch.equal(cm._contact_position_counts, c)
                if air is not None:
                    for got, want in zip((cm.last_air_time, cm.current_air_time, cm.last_contact_time, cm.current_contact_time), air):
                        assert torch.equal(got, want)
            _check(env, fix, t)
        # what cannot be folded is refused, nothing launched: a descriptor over another number of envs
        bad = nat.GfContactArgs()
        C.memmove(C.addressof(bad), C.addressof(env.cms[0]._args), C.sizeof(bad))
        bad.num_envs = n + 1
        bad.stats = None
        assert hip_backend.post_step_contacts(refs, [bad]) == -5
    finally:
        hip_backend.set_option(nat.GF_OPT_FOLD_CONTACT, 1)
