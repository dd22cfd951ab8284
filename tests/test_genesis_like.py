"""The manager step on a scene with ONLY Genesis' public surface (tests/genesis_like.py: fresh getter tensors, envs_idx setters,
collider.get_contacts() dict, rigid_solver.get_links_quat(), no gf_* extras) — the L0 side of the boundary.

The double runs the same stand-in physics in private, so an env built on it must produce, step for step and bit for bit, what the
same env produces on the synthetic scene (whose fast path is pinned to the reference by the golden trajectories): outputs, manager
state AND the simulator's own state (the reset rows have to arrive through the setters).  Checked for the ordinary step and for
the recorded + fused step, which must exist on such a scene: at most the one nonzero() the setters force, one call per getter
and tick."""
import os

import pytest
import torch

import envs
from genesis_forge_amd import tasks
from genesis_like import GenesisLikeScene

CASES = {
    # name: (factory, action width)
    "go2_cmd": (lambda n: envs.Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3, contacts=True, history=2,
                                                      obs_noise=True, scene_kwargs=dict(ang_noise=0.3, seed=3)), 12),
    "go2_plain": (lambda n: envs.Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3,
                                                        scene_kwargs=dict(ang_noise=0.3, seed=3)), 12),
    "rough_terrain": (lambda n: envs.Go2RoughTerrainEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3,
                                                        scene_kwargs=dict(ang_noise=0.3, seed=5, contact_prob=0.2)), 12),
    "gait": (lambda n: envs.Go2GaitTrainingEnv(num_envs=n, max_episode_length_s=1,
                                               scene_kwargs=dict(ang_noise=0.25, seed=7, contact_prob=0.05)), 12),
    "gait_curriculum": (lambda n: envs.Go2GaitTrainingCurriculumEnv(num_envs=n, max_episode_length_s=1,
                                                                    scene_kwargs=dict(ang_noise=0.25, seed=7, contact_prob=0.05)), 12),
    "humanoid28": (lambda n: envs.HumanoidGaitLikeEnv(num_envs=n, dofs=28), 28),
    "humanoid": (lambda n: envs.BerkeleyHumanoidEnv(num_envs=n, max_episode_length_s=1,
                                                    scene_kwargs=dict(ang_noise=0.2, seed=9, contact_prob=0.02, max_collision_pairs=30)), 12),
}


def _sim_of(env):
    sc = env.scene
    sim = sc._sim if isinstance(sc, GenesisLikeScene) else sc
    r = sim.robot
    return [r.pos, r.quat, r.lin_vel, r.ang_vel, r.dof_pos, r.dof_vel]


def run(name, dev, scene_cls, trace, n=70, steps=60, poison=True):
    make, width = CASES[name]
    if scene_cls is None:
        env = make(n)
    else:
        with tasks.use_scene(scene_cls):
            env = make(n)
        env.scene.poison = poison
    env.trace_enabled = trace
    env.build()
    env.seed(11)
    env.reset()
    g = torch.Generator().manual_seed(0)
    outs = []
    for t in range(steps):
        o, r, te, tr, ex = env.step(torch.randn(n, width, generator=g).to(dev))
        rec = [o, r, te, tr, env.episode_length] + _sim_of(env)
        for m in env.managers["command"]:
            rec.append(m.command)
        for name_, ob in ex["observations"].items():
            if name_ != "policy":
                rec.append(ob)
        for cm in env.managers["contact"]:
            rec.append(cm.contacts)
            if cm.current_air_time is not None:
                rec += [cm.current_air_time, cm.last_air_time]
        outs.append(([x.detach().cpu().clone() for x in rec], {k: float(v) for k, v in ex["episode"].items()}))
    return outs, env


def same(a, b):
    assert len(a) == len(b)
    for t, ((xs, lx), (ys, ly)) in enumerate(zip(a, b)):
        assert len(xs) == len(ys)
        for k, (x, y) in enumerate(zip(xs, ys)):
            assert torch.equal(x, y), f"output {k} differs at step {t}: max |d| = {(x.float() - y.float()).abs().max()}"
        assert lx == ly, f"log differs at step {t}: {lx} vs {ly}"


@pytest.mark.parametrize("name", sorted(CASES))
def test_ordinary_step_on_genesis_like_scene_cpu(oracle_backend, name):
    a, _ = run(name, "cpu", None, False)
    b, env = run(name, "cpu", GenesisLikeScene, False)
    same(a, b)
    dones = sum(int((x[0][2] | x[0][3]).sum()) for x in b)
    assert dones > 20, "the case should reset envs"
    assert not any(hasattr(env.robot, k) for k in ("gf_views", "gf_dofs", "gf_masked_dofs", "gf_masked_base"))
    assert not hasattr(env.scene, "gf_static_buffers") and not hasattr(env.scene.rigid_solver, "gf_contacts")


@pytest.mark.parametrize("name", sorted(CASES))
def test_recorded_step_on_genesis_like_scene_cpu(oracle_backend, name):
    a, _ = run(name, "cpu", None, False)
    before = oracle_backend.replays
    b, env = run(name, "cpu", GenesisLikeScene, True)
    same(a, b)
    tr = env._trace
    assert tr is not None, f"the step on a Genesis-shaped scene was not recorded: {env._untraceable}"
    assert oracle_backend.replays - before >= 50
    if name != "gait_curriculum":
        assert tr.post_refs is not None, "the post-physics phases should run as the fused launch"
    else:
        assert tr.tail_python and sorted(tr.tail_seg) == ["obs", "reset"]


@pytest.mark.parametrize("name", ["gait", "rough_terrain", "humanoid"])
def test_contact_arrays_that_change_shape_every_tick_cpu(oracle_backend, name):
    """Genesis pads the collider's contact arrays to the CURRENT tick's contact count, so their shape varies from
    tick to tick; a recorded step re-patched only the pointers and ran the kernel with the recorded step's slot count.  With the
    double trimming its arrays the same way (poisoned stale tensors included), the recorded step must still equal the synthetic run."""
    _varying_contacts(name, "cpu")


def _varying_contacts(name, dev):
    a, _ = run(name, dev, None, False)
    GenesisLikeScene.trim_contacts = True
    try:
        b, env = run(name, dev, GenesisLikeScene, True)
    finally:
        GenesisLikeScene.trim_contacts = False
    same(a, b)
    assert env._trace is not None, env._untraceable
    counts = env.scene.contact_counts
    assert len(set(counts)) > 2, f"the contact arrays should change shape from tick to tick: {sorted(set(counts))}"


def test_setter_assumption_is_checked_against_the_simulator(oracle_backend):
    """"after set_pos / set_quat / set_dofs_position the getters return what was set, velocities zeroed" was a comment
    (_scene_adapter.py).  It is now a run-time check before the first full reset: one env is probed through the
    setters and read back through the getters.  On the double it holds; on a variant whose setters ignore ``zero_velocity`` the env
    warns, takes the scene-side resets off the masked path (index lists, the reference's reset(ids)) and — recorded or ordinary —
    observes what the SIMULATOR holds (non-zero base velocities of a just-reset env), not what the masked reset would have assumed."""
    import warnings

    _, env = run("go2_plain", "cpu", GenesisLikeScene, True, steps=12)
    ad = env._adapter
    assert ad.setters_verified is True and ad.setter_report == []
    assert env.managers["action"]._can_fuse_reset() and all(m._can_fuse_reset() for m in env.managers["entity"])
    assert ad.fetches_last_tick == len(env._trace.scene_plan) > 0   # one call per getter of the plan in a replayed tick

    GenesisLikeScene.sloppy_setters = True
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            rec, env_r = run("go2_plain", "cpu", GenesisLikeScene, True, steps=40)
        assert any("envs_idx setters" in str(x.message) and "base vel" in str(x.message) for x in w), [str(x.message) for x in w]
        assert env_r._adapter.setters_verified is False
        assert not any(m._can_fuse_reset() for m in env_r.managers["entity"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ordi, env_o = run("go2_plain", "cpu", GenesisLikeScene, False, steps=40)
        same(rec, ordi)
        # a just-reset env keeps its base velocity on this simulator, and the observation shows it (go2_plain: obs[3:6] = ang vel, [6:9] = lin vel)
        seen = 0
        for xs, _log in rec:
            obs, term, trunc = xs[0], xs[2], xs[3]
            done = term | trunc
            if done.any():
                seen += int((obs[done][:, 3:9].abs().sum(1) > 0).sum())
        assert seen > 0
    finally:
        GenesisLikeScene.sloppy_setters = False


def test_one_getter_call_per_tick_and_one_index_list(oracle_backend, monkeypatch):
    """A replayed step on the double: control_dofs_position + scene.step() once, every getter of the plan once, setters only on
    steps that reset an env, and exactly one index-list compaction (gf_done_compact + sync: what the envs_idx setters need, where the
    reference has its nonzero(); managed_env.py:308-310)."""
    _, env = run("go2_cmd", "cpu", GenesisLikeScene, True, steps=20)
    tr = env._trace
    assert tr is not None and len(tr.scene_plan) > 0
    sc = env.scene
    nz = {"n": 0}
    real = torch.Tensor.nonzero

    def counting(self, *a, **k):
        nz["n"] += 1
        return real(self, *a, **k)

    monkeypatch.setattr(torch.Tensor, "nonzero", counting)
    g = torch.Generator().manual_seed(5)
    resets = 0
    for _ in range(25):
        sc.calls.clear()
        nz["n"] = 0
        f0 = env._adapter.fetches
        calls0 = len(oracle_backend.calls)
        o, r, te, tru, ex = env.step(torch.randn(70, 12, generator=g))
        assert sc.calls["step"] == 1 and sc.calls["control_dofs_position"] == 1
        assert env._adapter.fetches - f0 == len(tr.scene_plan)
        for getter in ("get_pos", "get_quat", "get_vel", "get_ang", "get_dofs_position", "get_dofs_velocity", "get_contacts", "get_links_quat"):
            assert sc.calls[getter] == 1, f"{getter} called {sc.calls[getter]} times in one tick"
        # the index list the envs_idx setters need (managed_env.py:308-310): ONE compaction + sync per tick, and not torch's nonzero()
        assert nz["n"] == 0 and oracle_backend.calls[calls0:].count("done_compact") == 1, (nz["n"], oracle_backend.calls[calls0:])
        done = bool((te | tru).any())
        resets += done
        for setter in ("set_dofs_position", "set_pos", "set_quat"):
            assert sc.calls[setter] == (1 if done else 0)
    assert resets > 3 and env._trace is tr


def test_unexplained_pointer_change_refuses_the_recording(oracle_backend, monkeypatch):
    """The recording's safety net: a descriptor pointer that changes from tick to tick and is neither scene state read through
    the snapshot nor a known per-step field must refuse the recording (here: link velocities handed to the reward descriptor from
    alternating buffers behind the snapshot's back) — the env keeps stepping phase by phase and says why."""
    from genesis_forge_amd.managers import ContactManager

    make, _ = CASES["humanoid28"]   # feet_slide reads the tracked links' velocities (GfContactView.link_vel)
    with tasks.use_scene(GenesisLikeScene):
        env = make(40)
    real_view = ContactManager.view
    bufs = {}

    def view(self, v, need_link_vel=False, need_link_pos=False):
        keep = real_view(self, v, need_link_vel=need_link_vel, need_link_pos=need_link_pos)
        if need_link_vel:
            pair = bufs.setdefault(id(self), [torch.zeros(40, self.contacts.shape[1], 3), torch.zeros(40, self.contacts.shape[1], 3)])
            pair.reverse()
            v.link_vel = pair[0].data_ptr()
        return keep

    monkeypatch.setattr(ContactManager, "view", view)
    env.build()
    env.reset()
    for _ in range(6):
        env.step(torch.zeros(40, 28))
    assert env._trace is None
    assert env._untraceable is not None and "link_vel" in env._untraceable, env._untraceable


def _fuzz_on(scene_cls, seed, dev, trace, steps=40):
    import test_fuzz_configs as fz

    fz.SCENE_CLS[0] = scene_cls
    if not trace:
        os.environ["GF_NO_TRACE"] = "1"
    try:
        return fz._run(seed, dev, steps=steps)
    finally:
        fz.SCENE_CLS[0] = None
        os.environ.pop("GF_NO_TRACE", None)


def _gl_seeds(default):
    """``GF_GL_SEEDS=a:b`` soaks other fuzz seeds on the double (as GF_FUZZ_SEEDS does for tests/test_fuzz_configs.py)."""
    spec = os.environ.get("GF_GL_SEEDS")
    if not spec:
        return default
    a, b = spec.split(":")
    return list(range(int(a), int(b)))


@pytest.mark.parametrize("seed", _gl_seeds(list(range(0, 28, 2)) + [123, 466]))
def test_random_configs_on_genesis_like_scene_cpu(oracle_backend, seed):
    """The randomised task configs (tests/test_fuzz_configs.py: drawn reward / termination / observation tables, Python-level terms,
    a third ObservationManager, reset() overrides, output / history modes) on the double, recorded, against the same config on the
    synthetic scene stepped phase by phase: bit for bit — every output, every manager buffer, the simulator's pose."""
    import test_fuzz_configs as fz

    want, _ = _fuzz_on(None, seed, "cpu", trace=False)
    got, info = _fuzz_on(GenesisLikeScene, seed, "cpu", trace=True)
    # (a reset(ids) override of the action manager writes joint positions through the simulator's setters in the middle of the step: on such
    #  a scene the step stays ordinary — every getter call fetches again — and must equal the synthetic run all the same)
    assert info["recorded"] or "action" in info["env"].user_reset_cls, info["env"]._untraceable
    fz._compare(got, want, 0, f"seed {seed} on the Genesis-shaped double")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gait", "humanoid"])
def test_contact_arrays_that_change_shape_every_tick_hip(hip_backend, name):
    _varying_contacts(name, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _gl_seeds(list(range(1, 28, 3)) + [140, 466]))
def test_random_configs_on_genesis_like_scene_hip(hip_backend, seed):
    """-m gpu: the same on the HIP kernels — recorded step on the double == recorded step on the synthetic scene, bit for bit."""
    import test_fuzz_configs as fz

    want, _ = _fuzz_on(None, seed, "cuda", trace=True)
    got, info = _fuzz_on(GenesisLikeScene, seed, "cuda", trace=True)
    assert info["recorded"] or "action" in info["env"].user_reset_cls, info["env"]._untraceable
    fz._compare(got, want, 0, f"seed {seed} on the Genesis-shaped double (HIP)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_recorded_step_on_genesis_like_scene_hip_equals_oracle(hip_backend, oracle_lib_path, name):
    """-m gpu: HIP on the Genesis-shaped double == oracle on the synthetic scene (masks / counters bit-exact, floats <= 1e-5)."""
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    n = 130
    b, env = run(name, "cuda", GenesisLikeScene, True, n=n)
    tr = env._trace
    assert tr is not None, f"not recorded: {env._untraceable}"
    c, _ = run(name, "cuda", None, True, n=n)
    same(b, c)   # HIP on the double == HIP on the synthetic scene, bit for bit
    torch.cuda.synchronize()
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    try:
        a, _ = run(name, "cpu", None, False, n=n)
    finally:
        nat.set_backend(None)
        gs.set_device("cuda:0")
    for t, ((xs, lx), (ys, ly)) in enumerate(zip(a, b)):
        for k, (x, y) in enumerate(zip(xs, ys)):
            if x.dtype in (torch.bool, torch.int32, torch.int64, torch.uint8):
                assert torch.equal(x, y), f"output {k} differs at step {t}"
            else:
                assert torch.allclose(x, y, atol=1e-5, rtol=0), f"output {k} at step {t}: {(x - y).abs().max()}"
        assert set(lx) == set(ly)
