"""The recorded step (genesis_forge_amd/_trace.py) must be indistinguishable from the ordinary path."""
import pytest
import torch

from envs import Go2CommandDirectionEnv


def _run(dev, trace, n=70, steps=50, mutate_at=None, cls=None):
    env = (cls or Go2CommandDirectionEnv)(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3, contacts=True, history=2, obs_noise=True,
                                 scene_kwargs=dict(ang_noise=0.3, seed=3))
    env.trace_enabled = trace
    env.build()
    env.seed(5)
    env.reset()
    g = torch.Generator().manual_seed(0)
    outs = []
    for t in range(steps):
        if mutate_at is not None and t == mutate_at:
            env.reward_manager.cfg["action_rate"].weight = -0.5          # curriculum-style mutation
            env.velocity_command.range["lin_vel_x"][1] = 3.0              # in-place range edit (no setter involved)
            env.reward_manager.cfg["base_height_target"].params["target_height"] = 0.33
        o, r, te, tr, ex = env.step(torch.randn(n, 12, generator=g).to(dev))
        outs.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                     env.velocity_command._command.cpu().clone()))
    return outs, env


def _same(a, b):
    for t, (x, y) in enumerate(zip(a, b)):
        for k in (0, 1, 2, 3, 5):
            assert torch.equal(x[k], y[k]), f"output {k} differs at step {t}"
        assert x[4] == y[4], f"log differs at step {t}: {x[4]} vs {y[4]}"


def test_traced_step_equals_ordinary_cpu(oracle_backend):
    a, _ = _run("cpu", False)
    before = oracle_backend.replays
    b, env = _run("cpu", True)
    assert oracle_backend.replays - before >= 45, "the step was not recorded"
    _same(a, b)


def test_trace_invalidated_by_mutation_cpu(oracle_backend):
    a, _ = _run("cpu", False, mutate_at=20)
    b, env = _run("cpu", True, mutate_at=20)
    _same(a, b)
    assert env._trace is not None, "trace should have been re-recorded after the mutation"


def test_reset_override_with_curriculum_mutation_cpu(oracle_backend):
    """An env that overrides reset() AND is mutated in mid-run (a curriculum): the recording — main segment and the two native tail
    segments — is dropped and rebuilt, and every step equals the ordinary path's."""
    class HookEnv(Go2CommandDirectionEnv):
        hooks = 0

        def reset(self, env_ids=None):
            out = super().reset(env_ids)
            if env_ids is not None:
                self.hooks += 1
            return out

    a, e0 = _run("cpu", False, mutate_at=20, cls=HookEnv)
    b, env = _run("cpu", True, mutate_at=20, cls=HookEnv)
    _same(a, b)
    tr = env._trace
    assert tr is not None and tr.tail_python and sorted(tr.tail_seg) == ["obs", "reset"]
    assert env.hooks == e0.hooks > 0


def test_reset_override_tail_is_adopted_when_the_first_reset_comes_late_cpu(oracle_backend):
    """No env is done while the step is being recorded (1 s episodes, no early terminations): the recording starts without native tail
    segments and adopts them from the Python walk of the first replayed step that resets an env."""
    class HookEnv(Go2CommandDirectionEnv):
        def reset(self, env_ids=None):
            return super().reset(env_ids)

    def run(trace):
        env = HookEnv(num_envs=70, max_episode_length_s=1, cmd_resample_s=0.3, history=2, scene_kwargs=dict(ang_noise=0.0, seed=3))
        env.trace_enabled = trace
        env.build()
        env.seed(5)
        env.reset()
        g = torch.Generator().manual_seed(0)
        outs, seen = [], []
        for t in range(75):
            o, r, te, tr, ex = env.step(torch.randn(70, 12, generator=g))
            outs.append((o.clone(), r.clone(), te.clone(), tr.clone(), {k: float(v) for k, v in ex["episode"].items()}))
            seen.append((int((te | tr).sum()), bool(env._trace is not None and env._trace.tail_seg)))
        return outs, seen

    a, _ = run(False)
    b, seen = run(True)
    first = next(t for t, (done, _) in enumerate(seen) if done)
    assert first > 5, "the config should not reset an env while the step is being recorded"
    assert not any(seg for _, seg in seen[:first]) and all(seg for _, seg in seen[first:]), "tail segments from the first step with a reset on"
    for t, (x, y) in enumerate(zip(a, b)):
        for k in range(4):
            assert torch.equal(x[k], y[k]), f"output {k} differs at step {t}"
        assert x[4] == y[4], f"log differs at step {t}"


def test_reset_override_with_python_on_reset_entry_keeps_the_python_tail(oracle_backend):
    """an env that overrides reset() AND whose EntityManager has a second, Python-level on_reset entry.  That
    manager needs ``reset(ids)`` — it never goes through the masked-reset launch — so the tail must not be replayed natively:
    the user's function runs on every reset step and the trajectory equals the ordinary path's."""
    class HookEnv(Go2CommandDirectionEnv):
        calls_now = 0

        def config(self):
            super().config()
            from genesis_forge_amd.managers.config import ConfigItem

            def custom(env, entity, envs_idx):
                type(self).calls_now += 1

            self.robot_manager.on_reset["custom"] = ConfigItem({"fn": custom}, self, on_dirty=self.invalidate_trace)

        def reset(self, env_ids=None):
            return super().reset(env_ids)

    def run(trace):
        HookEnv.calls_now = 0
        out, env = _run("cpu", trace, cls=HookEnv)
        return out, env, HookEnv.calls_now

    a, _, calls_a = run(False)
    b, env, calls_b = run(True)
    _same(a, b)
    assert calls_a == calls_b > 10, f"the Python on_reset entry ran {calls_b} times in the recorded run, {calls_a} in the ordinary one"
    assert env._trace is not None and env._trace.tail_python and not env._trace.tail_seg, "no native tail for a manager that needs reset(ids)"


def test_mutation_inside_reset_override_reaches_the_same_steps_observation(oracle_backend):
    """the reference gait example's pattern — reset() → update_curriculum() — mutating something the OBSERVATION
    depends on inside the override.  The ordinary path reads the live value in the same step; the replayed step must too (its frozen
    'obs' tail segment is stale from the moment of the mutation)."""
    class HookEnv(Go2CommandDirectionEnv):
        resets = 0

        def reset(self, env_ids=None):
            out = super().reset(env_ids)
            if env_ids is not None:
                self.resets += 1
                if self.resets in (12, 20):
                    self.observation_manager.noise = 0.05 if self.resets == 12 else 0.0
            return out

    a, e0 = _run("cpu", False, cls=HookEnv)
    b, env = _run("cpu", True, cls=HookEnv)
    assert e0.resets == env.resets >= 20
    _same(a, b)


@pytest.mark.parametrize("trace", [False, True])
def test_params_dict_bulk_mutations_take_effect(oracle_backend, trace):
    """the reference re-reads **params every step, so params.update(...) / pop / setdefault / |= / clear take effect
    there; here they must mark the compiled term table dirty (and drop a recorded step) just like item assignment does."""
    def run(how):
        env = Go2CommandDirectionEnv(num_envs=40, max_episode_length_s=1, scene_kwargs=dict(ang_noise=0.3, seed=3))
        env.trace_enabled = trace
        env.build()
        env.seed(5)
        env.reset()
        g = torch.Generator().manual_seed(0)
        out = []
        for t in range(16):
            if t == 8:
                p = env.reward_manager.cfg["base_height_target"].params
                q = env.reward_manager.cfg["tracking_lin_vel"].params
                if how == "setitem":
                    p["target_height"] = 0.37
                    q["sensitivity"] = 0.5
                elif how == "update":
                    p.update(target_height=0.37)
                    q.setdefault("sensitivity", 0.5)
                elif how == "ior":
                    p |= {"target_height": 0.37}
                    q.update({"sensitivity": 0.5})
                elif how == "pop":       # back to the term's defaults: target_height has none -> set again; sensitivity default 0.25
                    p.pop("target_height")
                    p["target_height"] = 0.37
                    q["sensitivity"] = 0.9
                    q.pop("sensitivity")
                    q.update(sensitivity=0.5)
            out.append(env.step(torch.randn(40, 12, generator=g))[1].clone())
        assert (env._trace is not None) == trace
        return out

    want = run("setitem")
    assert not torch.equal(want[7], want[9])
    for how in ("update", "ior", "pop"):
        got = run(how)
        for t, (a, b) in enumerate(zip(want, got)):
            assert torch.equal(a, b), f"{how}: reward differs at step {t} — the mutation did not reach the term table"


def test_assignments_that_change_nothing_keep_the_recorded_step(oracle_backend):
    """The reference's documented curriculum recipe assigns its params on EVERY step, mostly with the value they already have
    (docs/guide/managers/termination.md, "Curriculum-Based Termination": step() → update_curriculum() → params[...] = value).  Such an
    assignment must not drop the compiled tables / the recorded step; a changed value must; re-assigning a MUTABLE value (the documented
    way to announce an in-place edit) still must."""
    env = Go2CommandDirectionEnv(num_envs=33, contacts=True, history=2, scene_kwargs=dict(ang_noise=0.3, seed=3))
    env.build()
    env.seed(5)
    env.reset()
    for _ in range(4):
        env.step(torch.zeros(33, 12))
    tm, rm, om, vc = env.termination_manager, env.reward_manager, env.observation_manager, env.velocity_command
    item = next(iter(om.cfg))

    def curriculum():
        tm.term_cfg["fall_over"].params["limit_angle"] = 10.0
        tm.term_cfg["fall_over"].params.update(limit_angle=10)
        tm.term_cfg["timeout"].time_out = True
        rm.cfg["action_rate"].weight = rm.cfg["action_rate"].weight
        rm.cfg["base_height_target"].params["target_height"] = 0.3
        om.cfg[item].scale = om.cfg[item].scale
        om.cfg[item].noise = om.cfg[item].noise
        om.noise = om.noise
        rm.logging_enabled = True
        env.foot_contacts.enabled = True
        vc.resample_time_sec = vc.resample_time_sec

    before = oracle_backend.replays
    for _ in range(12):
        curriculum()
        env.step(torch.zeros(33, 12))
    assert env._trace is not None and oracle_backend.replays - before == 12, "an assignment that changes nothing dropped the recorded step"
    tm.term_cfg["fall_over"].params["limit_angle"] = 12.5      # a real change of a NUMBER: the recorded step's term table is refreshed in place
    rm.cfg["action_rate"].weight = -0.02
    before = oracle_backend.replays
    for _ in range(3):
        env.step(torch.zeros(33, 12))
    assert env._trace is not None and oracle_backend.replays - before == 3
    import math
    assert any(abs(tm._program.args.terms[k].p[j] - math.sin(math.radians(12.5))) < 1e-6 for k in range(tm._program.n) for j in range(4)), \
        "the termination table the recorded step uses did not take the new limit"
    rm.cfg["action_rate"].weight = 0.0                           # … to zero: the term leaves the table — a structural change
    env.step(torch.zeros(33, 12))
    assert oracle_backend.replays - before == 3, "a weight set to zero must drop the recorded step"
    for _ in range(3):
        env.step(torch.zeros(33, 12))
    assert env._trace is not None
    pos = env.robot_manager.on_reset["position"].params
    vec = list(pos["position"])   # (a list of this test's own: the config's is a module constant other envs share)
    pos["position"] = vec
    for _ in range(3):
        env.step(torch.zeros(33, 12))
    assert env._trace is not None
    vec[2] = 0.45
    pos["position"] = vec     # the same mutable object again: "I edited it in place"
    assert env._trace is None


def test_weights_and_params_annealed_every_step_stay_on_the_recorded_step(oracle_backend):
    """A curriculum that changes NUMBERS on every step (a weight annealed, a limit tightened): the term tables are compiled again and
    their numbers go into the descriptors that exist — the step is recorded after the usual two steps and stays recorded; every step
    equals the ordinary one, logs included."""
    class Annealed(Go2CommandDirectionEnv):
        def step(self, actions):
            self.reward_manager.cfg["action_rate"].weight = -0.005 * (1.0 + 1e-2 * self.step_count)
            self.termination_manager.term_cfg["fall_over"].params["limit_angle"] = 10.0 + 0.01 * self.step_count
            return super().step(actions)

    a, _ = _run("cpu", False, cls=Annealed)
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Annealed)
    _same(a, b)
    assert env._trace is not None and oracle_backend.replays - before >= 47


def test_a_weight_assigned_by_user_code_in_the_middle_of_a_step_reaches_the_phases_behind_it(oracle_backend):
    """A Python-level termination term that also anneals a reward weight: the reward phase runs behind it and must use the new weight in
    the same step — recorded (the term is user code between two native pieces) exactly as phase by phase."""
    from genesis_forge_amd.managers import TerminationManager

    class Env(Go2CommandDirectionEnv):
        def config(self):
            super().config()
            tc = {k: {"fn": v.fn, "params": dict(v.params), "time_out": v.time_out} for k, v in self.termination_manager.term_cfg.items()}

            def far_and_anneal(env):
                env.reward_manager.cfg["action_rate"].weight = -0.005 * (1.0 + 0.05 * env.step_count)
                return env.robot.get_pos()[:, :2].abs().sum(dim=1) > 0.6

            tc["far"] = {"fn": far_and_anneal}
            self.managers["termination"] = None
            self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg=tc)

    a, _ = _run("cpu", False, cls=Env)
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Env)
    _same(a, b)
    assert env._trace is not None and oracle_backend.replays - before >= 40


def test_parity_draws_disable_trace(oracle_backend):
    env = Go2CommandDirectionEnv(num_envs=8)
    env.build()
    env.reset()
    for _ in range(4):
        env.step(torch.zeros(8, 12))
    assert env._trace is not None
    env.set_draws(**{"command:0": torch.rand(8, 3)})
    assert env._trace is None
    env.step(torch.zeros(8, 12))
    assert env._trace is None  # the step that consumed draws ran the ordinary path and is not a recording candidate twice yet


def test_user_overrides_limit_what_is_recorded(oracle_backend):
    """A reset() override is honoured by index list: the step is recorded only up to the reset (the rest stays Python).
    A step() override that wraps `super().step()` is code around the step, like the training loop's: the step inside is recorded and
    fused all the same, and a manager call the override makes between steps drops the recording like one the script makes.
    A get_observations() override is user code BEHIND the step's native phases (round 4): the step is recorded and fused without the
    observations, the override runs where the ordinary step calls it."""
    class ResetEnv(Go2CommandDirectionEnv):
        def reset(self, env_ids=None):
            return super().reset(env_ids)

    class StepEnv(Go2CommandDirectionEnv):
        seen = 0

        def step(self, actions):
            self.seen += 1   # (the reference's gait example renders a camera here)
            return super().step(actions * 1.0)

    class ObsEnv(Go2CommandDirectionEnv):
        def get_observations(self):
            return super().get_observations()

    ref = Go2CommandDirectionEnv(num_envs=8, scene_kwargs=dict(seed=3))
    ref.trace_enabled = False
    ref.build()
    ref.seed(5)
    ref.reset()
    g = torch.Generator().manual_seed(0)
    acts = [torch.randn(8, 12, generator=g) for _ in range(6)]
    want = [ref.step(a)[0].clone() for a in acts]
    for cls, how in ((ResetEnv, "tail"), (StepEnv, "fused"), (ObsEnv, "obs")):
        env = cls(num_envs=8, scene_kwargs=dict(seed=3))
        env.build()
        env.seed(5)
        env.reset()
        got = [env.step(a)[0].clone() for a in acts]
        for t, (x, y) in enumerate(zip(want, got)):
            assert torch.equal(x, y), f"{cls.__name__}: observation {t} differs from the plain env"
        if how == "tail":
            # (recorded up to the reset: termination, rewards and the command step are ONE launch that resets nothing — GF_POST_NO_RESET)
            assert env._trace is not None and env._trace.tail_python
            from genesis_forge_amd import _native as nat

            assert env._trace.post_refs is not None and env._trace.post_refs.flags == nat.GF_POST_NO_RESET and env._trace.post_refs.num_observe == 0
        elif how == "fused":
            assert env._trace is not None and not env._trace.tail_python and env._trace.post_refs is not None and env.seen == 6
        else:
            tr = env._trace
            assert tr is not None and len(tr.py_marks) == 1 and tr.post_refs is not None and tr.post_refs.num_observe == 0


def _user_manager_env():
    """A user-defined CommandManager class (its own step() and reset(), Python + torch inside — the shape of the reference's
    examples/gait_trainer/gait_command_manager.py), a reward term and an observation item that read it."""
    from genesis_forge_amd.managers import CommandManager

    class PhaseClock(CommandManager):
        steps = resets = 0

        def __init__(self, env):
            super().__init__(env, range=(0.5, 1.5), resample_time_sec=0.2)
            self.phase = torch.zeros(env.num_envs)

        def step(self):
            type(self).steps += 1
            super().step()                                    # (the base class' native resample launch, made from user code)
            self.phase = (self.phase + self.env.dt * self._command[:, 0]) % 1.0

        def reset(self, env_ids=None):
            type(self).resets += 1
            super().reset(env_ids)
            if env_ids is None:
                self.phase = torch.zeros_like(self.phase)
            else:
                self.phase[env_ids] = 0.0

    class Env(Go2CommandDirectionEnv):
        def config(self):
            super().config()
            from genesis_forge_amd.managers import ObservationManager, RewardManager
            self.clock = PhaseClock(self)
            rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in self.reward_manager.cfg.items()}
            rc["in_phase"] = {"weight": 0.2, "fn": lambda env: torch.cos(6.2831853 * self.clock.phase)}
            self.managers["reward"] = None
            self.reward_manager = RewardManager(self, logging_enabled=True, cfg=rc)
            om = self.observation_manager
            oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
            oc["clock"] = {"fn": lambda env: torch.stack([torch.sin(6.2831853 * self.clock.phase), self.clock.phase], dim=-1)}
            self.managers["observation"].remove(om)
            self.observation_manager = ObservationManager(self, cfg=oc, history_len=2)

    return Env, PhaseClock


def test_user_manager_class_is_replayed_between_native_phases(oracle_backend):
    """Round 3: a user-defined manager class no longer takes the whole step off the recording.  Its step() / reset(ids) run as user
    code between the native phases — same place, same arguments, as often as in the ordinary step — and the trajectory is the
    ordinary step's bit for bit."""
    Env, Clock = _user_manager_env()
    Clock.steps = Clock.resets = 0
    a, _ = _run("cpu", False, cls=Env)
    counts_a = (Clock.steps, Clock.resets)
    Clock.steps = Clock.resets = 0
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Env)
    counts_b = (Clock.steps, Clock.resets)
    _same(a, b)
    tr = env._trace
    assert tr is not None and len(tr.py_marks) == 2 and tr.post_refs is None, "recorded, with the user manager's step() and reset(ids) as splits"
    assert oracle_backend.replays - before >= 40
    assert counts_a == counts_b and counts_a[0] == 50 and counts_a[1] > 10, (counts_a, counts_b)


def _user_term_reward_env(which):
    """User-defined TerminationManager / RewardManager CLASSES with a step() of their own around the library's (`super().step()`, then
    torch on the manager's buffers) — managers that produce the step's native outputs themselves."""
    from genesis_forge_amd.managers import RewardManager, TerminationManager

    class CappedRewards(RewardManager):
        steps = 0

        def step(self):
            type(self).steps += 1
            r = super().step()
            r.clamp_(min=-0.5)           # in place on the manager's buffer: what the env returns and what a rollout storage copies
            return r

    class GracefulTerminations(TerminationManager):
        steps = 0

        def step(self):
            type(self).steps += 1
            te, tr = super().step()
            te &= self.env.episode_length > 2    # a grace period applied to EVERY termination, in place on the manager's mask
            return te, tr

    class Env(Go2CommandDirectionEnv):
        def config(self):
            super().config()
            if "reward" in which:
                rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in self.reward_manager.cfg.items()}
                self.managers["reward"] = None
                self.reward_manager = CappedRewards(self, logging_enabled=True, cfg=rc)
            if "termination" in which:
                tc = {k: {"fn": v.fn, "params": dict(v.params), "time_out": v.time_out} for k, v in self.termination_manager.term_cfg.items()}
                self.managers["termination"] = None
                self.termination_manager = GracefulTerminations(self, logging_enabled=True, term_cfg=tc)

    return Env, CappedRewards, GracefulTerminations


@pytest.mark.parametrize("which", ["reward", "termination", "reward+termination"])
def test_user_reward_and_termination_manager_classes_are_python_phases_of_a_recorded_step(oracle_backend, which):
    """Round 4 (the part of it that pays): a user RewardManager / TerminationManager class with its own step() used to
    keep the env on the 127 us ordinary step for good.  Its step() is now user code BETWEEN native pieces of a recorded step, like a
    user command manager's: it runs where the ordinary step calls it, as often, with the launches it makes itself pointed at the
    step's statistics slot; the phases on either side run as phase chains.  Bit-identical to the ordinary step, logs included."""
    Env, R, T = _user_term_reward_env(which)
    R.steps = T.steps = 0
    a, _ = _run("cpu", False, cls=Env)
    counts_a = (R.steps, T.steps)
    R.steps = T.steps = 0
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Env)
    _same(a, b)
    tr = env._trace
    assert tr is not None, f"not recorded: {env._untraceable}"
    assert len(tr.py_marks) == which.count("+") + 1 and tr.post_refs is None
    assert oracle_backend.replays - before >= 40
    assert (R.steps, T.steps) == counts_a and sum(counts_a) == 50 * (which.count("+") + 1)
    assert any("Terminations / " in k for row in b for k in row[4]) and any("Rewards / " in k for row in b for k in row[4])


def _user_obs_env(which):
    """A user-defined ObservationManager CLASS whose get_observations() post-processes the library's, and / or an env whose own
    get_observations() does (normalisation, clipping — the usual reasons)."""
    from genesis_forge_amd.managers import ObservationManager

    class ClippedObs(ObservationManager):
        calls = 0

        def get_observations(self):
            type(self).calls += 1
            return super().get_observations().clamp(-3.0, 3.0)

    class Env(Go2CommandDirectionEnv):
        env_calls = 0

        def config(self):
            super().config()
            if "manager" in which:
                om = self.observation_manager
                oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
                self.managers["observation"].remove(om)
                self.observation_manager = ClippedObs(self, cfg=oc, history_len=2)

        if "env" in which:
            def get_observations(self):
                type(self).env_calls += 1
                o = super().get_observations()
                return None if o is None else o * 0.5

    return Env, ClippedObs


@pytest.mark.parametrize("which", ["manager", "env", "manager+env"])
def test_user_observation_code_is_a_python_phase_behind_the_fused_launch(oracle_backend, which):
    """Round 4: a user ObservationManager class (its own get_observations()) and an env-level get_observations() override no longer keep
    the env on the ordinary step: termination ... reset stay ONE fused launch (without that manager's observation) and the user's code
    runs behind it, where the ordinary step calls it, as often.  Bit-identical to the ordinary step."""
    Env, Obs = _user_obs_env(which)
    Obs.calls = Env.env_calls = 0
    a, _ = _run("cpu", False, cls=Env)
    counts_a = (Obs.calls, Env.env_calls)
    Obs.calls = Env.env_calls = 0
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Env)
    _same(a, b)
    tr = env._trace
    assert tr is not None, f"not recorded: {env._untraceable}"
    assert tr.post_refs is not None and tr.post_refs.num_observe == 0 and len(tr.py_marks) == 1
    assert oracle_backend.replays - before >= 40
    assert (Obs.calls, Env.env_calls) == counts_a and sum(counts_a) >= 50


def _user_reset_env(which):
    """User manager classes that override reset(ids) of the ACTION / REWARD / TERMINATION manager (around the library's): such a manager
    is reset by index list behind the masked reset, the others stay sections of it."""
    from genesis_forge_amd.managers import PositionActionManager, RewardManager, TerminationManager

    seen = {"action": 0, "reward": 0, "termination": 0}

    class CountingActions(PositionActionManager):
        def reset(self, envs_idx=None):
            seen["action"] += 1 if envs_idx is None else int(len(envs_idx))
            return super().reset(envs_idx)

    class CountingRewards(RewardManager):
        def reset(self, envs_idx=None):
            seen["reward"] += 1 if envs_idx is None else int(len(envs_idx))
            return super().reset(envs_idx)

    class CountingTerminations(TerminationManager):
        def reset(self, envs_idx=None):
            seen["termination"] += 1 if envs_idx is None else int(len(envs_idx))
            return super().reset(envs_idx)

    class Env(Go2CommandDirectionEnv):
        if "action" in which:
            action_cls = CountingActions

        def config(self):
            super().config()
            if "reward" in which:
                rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in self.reward_manager.cfg.items()}
                self.managers["reward"] = None
                self.reward_manager = CountingRewards(self, logging_enabled=True, cfg=rc)
            if "termination" in which:
                tc = {k: {"fn": v.fn, "params": dict(v.params), "time_out": v.time_out} for k, v in self.termination_manager.term_cfg.items()}
                self.managers["termination"] = None
                self.termination_manager = CountingTerminations(self, logging_enabled=True, term_cfg=tc)

    return Env, seen


@pytest.mark.parametrize("which", ["reward", "action", "termination", "action+reward+termination"])
def test_manager_reset_overrides_are_reset_by_index_list_in_a_recorded_step(oracle_backend, which):
    Env, seen = _user_reset_env(which)
    a, _ = _run("cpu", False, cls=Env)
    counts_a = dict(seen)
    for k in seen:
        seen[k] = 0
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Env)
    _same(a, b)
    assert env._trace is not None, f"not recorded: {env._untraceable}"
    assert oracle_backend.replays - before >= 40
    assert dict(seen) == counts_a and all(counts_a[k] > 0 for k in which.split("+")), (seen, counts_a)


@pytest.mark.gpu
def test_manager_reset_overrides_recorded_hip(hip_backend):
    Env, seen = _user_reset_env("action+reward+termination")
    a, _ = _run("cuda", False, n=1000, cls=Env)
    b, env = _run("cuda", True, n=1000, cls=Env)
    assert env._trace is not None
    _same(a, b)


def _user_action_env(which):
    """A user-defined action manager CLASS: `handle_actions()` overridden (the reference's documented extension point,
    position_action_manager.py:389-392 — here a first-order low-pass on the incoming actions in front of the library's processing) or
    `step()` wrapped (targets post-processed)."""
    from genesis_forge_amd.managers import PositionActionManager

    class SmoothedActions(PositionActionManager):
        calls = 0

        def handle_actions(self, actions):
            type(self).calls += 1
            prev = getattr(self, "_lp", None)
            self._lp = actions.clone() if prev is None else 0.7 * prev + 0.3 * actions
            return super().handle_actions(self._lp)

    class HalvedTargets(PositionActionManager):
        calls = 0

        def step(self, actions):
            type(self).calls += 1
            return super().step(actions * 0.5)

    Cls = SmoothedActions if which == "handle_actions" else HalvedTargets

    class Env(Go2CommandDirectionEnv):
        action_cls = Cls

    return Env, Cls


@pytest.mark.parametrize("which", ["handle_actions", "step"])
def test_user_action_manager_class_is_a_python_phase_of_a_recorded_step(oracle_backend, which):
    """A user action manager class no longer keeps the env on the ordinary step: the env's bookkeeping launch takes the action kernel's
    place in the recording, the user's step() / handle_actions() follows it as user code (its own `super()` launch included), and the
    rest of the step is the fused launch as ever.  `handle_actions()` is honoured at all (it is what the reference's step() calls)."""
    Env, Cls = _user_action_env(which)
    Cls.calls = 0
    a, env_a = _run("cpu", False, cls=Env)
    calls_a = Cls.calls
    assert calls_a >= 50
    plain, _ = _run("cpu", False)
    assert not torch.equal(a[-1][0], plain[-1][0]), "the user's action code has no effect on the step"
    Cls.calls = 0
    before = oracle_backend.replays
    b, env = _run("cpu", True, cls=Env)
    _same(a, b)
    tr = env._trace
    assert tr is not None, f"not recorded: {env._untraceable}"
    assert tr.post_refs is not None and len(tr.py_marks) == 1 and tr.action_owner is None
    assert oracle_backend.replays - before >= 40 and Cls.calls == calls_a


@pytest.mark.gpu
def test_user_action_manager_class_recorded_hip(hip_backend):
    Env, _Cls = _user_action_env("handle_actions")
    a, _ = _run("cuda", False, n=1000, cls=Env)
    b, env = _run("cuda", True, n=1000, cls=Env)
    assert env._trace is not None and env._trace.post_refs is not None and env._trace.action_owner is None
    _same(a, b)


def test_env_get_observations_override_next_to_a_reset_override(oracle_backend):
    """Both overridden: the recorded step ends in front of the reset, the Python tail calls the user's get_observations() — and the
    step returns ITS value, not the policy manager's (fuzz seed 216)."""
    class Env(Go2CommandDirectionEnv):
        def reset(self, env_ids=None):
            return super().reset(env_ids)

        def get_observations(self):
            o = super().get_observations()
            return None if o is None else o * 0.5

    a, _ = _run("cpu", False, cls=Env)
    b, env = _run("cpu", True, cls=Env)
    assert env._trace is not None and env._trace.tail_python
    _same(a, b)


def test_perform_observation_override_is_honoured(oracle_backend):
    """The reference's get_observations() takes the new frame from self._perform_observation() (observation_manager.py:218-256): a
    subclass that overrides it (here: clipping the frame before it enters the history) is honoured — the frame list and the concat are
    then the reference's — and such a manager is user code behind the fused launch of a recorded step."""
    from genesis_forge_amd.managers import ObservationManager

    class ClippedFrames(ObservationManager):
        def _perform_observation(self):
            return super()._perform_observation().clamp(-0.5, 0.5)

    class Env(Go2CommandDirectionEnv):
        def config(self):
            super().config()
            om = self.observation_manager
            oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
            self.managers["observation"].remove(om)
            self.observation_manager = ClippedFrames(self, cfg=oc, history_len=2, noise=om.noise)

    plain, _ = _run("cpu", False)
    a, _ = _run("cpu", False, cls=Env)
    b, env = _run("cpu", True, cls=Env)
    _same(a, b)
    assert env._trace is not None and env._trace.post_refs is not None and len(env._trace.py_marks) == 1
    for t, (x, y) in enumerate(zip(a, plain)):
        assert torch.equal(x[0], y[0].clamp(-0.5, 0.5)), f"step {t}: the frames are not the clipped library frames"


@pytest.mark.gpu
def test_user_observation_code_recorded_hip(hip_backend):
    Env, _Obs = _user_obs_env("manager+env")
    a, _ = _run("cuda", False, n=1000, cls=Env)
    b, env = _run("cuda", True, n=1000, cls=Env)
    assert env._trace is not None and env._trace.post_refs is not None
    _same(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["reward+termination"])
def test_user_reward_and_termination_manager_classes_recorded_hip(hip_backend, which):
    Env, _R, _T = _user_term_reward_env(which)
    a, _ = _run("cuda", False, n=1000, cls=Env)
    b, env = _run("cuda", True, n=1000, cls=Env)
    assert env._trace is not None and len(env._trace.py_marks) == 2
    _same(a, b)


@pytest.mark.gpu
def test_user_manager_class_recorded_hip(hip_backend):
    Env, _Clock = _user_manager_env()

    def run(trace):
        env = Env(num_envs=1000, max_episode_length_s=1, cmd_resample_s=0.3, contacts=True, history=2, scene_kwargs=dict(ang_noise=0.3, seed=3))
        env.clock_dev = "cuda"
        env.trace_enabled = trace
        env.build()
        env.clock.phase = env.clock.phase.cuda()
        env.seed(5)
        env.reset()
        g = torch.Generator().manual_seed(0)
        outs = []
        for t in range(50):
            o, r, te, tr, ex = env.step(torch.randn(1000, 12, generator=g).cuda())
            outs.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                         env.velocity_command._command.cpu().clone()))
        return outs, env

    a, _ = run(False)
    b, env = run(True)
    assert env._trace is not None and len(env._trace.py_marks) == 2
    _same(a, b)


@pytest.mark.gpu
def test_traced_step_equals_ordinary_hip(hip_backend):
    a, _ = _run("cuda", False, n=1000)
    b, env = _run("cuda", True, n=1000)
    assert env._trace is not None
    assert env._trace.post_refs is not None, "the post-physics phases should have been fused into one launch"
    _same(a, b)


@pytest.fixture
def post_variant(hip_backend, request):
    """Selects the fused kernel's variant (0 = interpreter, one wave per tile; 1 = interpreter, four specialised waves;
    2 = static programs where the config matches one) for one test."""
    from genesis_forge_amd import _native as nat

    hip_backend.set_option(nat.GF_OPT_POST_VARIANT, request.param)
    yield request.param
    hip_backend.set_option(nat.GF_OPT_POST_VARIANT, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("post_variant,obs_noise", [(0, True), (1, True), (2, True), (2, False)], indirect=["post_variant"])
@pytest.mark.parametrize("n", [1, 65, 4096])
def test_fused_post_physics_equals_phase_by_phase_hip(hip_backend, post_variant, obs_noise, n):
    """gf_post_physics_step (one launch) against the same recorded step replayed phase by phase.  Without observation noise
    this is the structure of examples/command_direction, which variant 2 runs as a static program."""
    outs = []
    for fuse in (False, True):
        env = Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3, obs_noise=obs_noise, scene_kwargs=dict(ang_noise=0.3, seed=11))
        env.fuse_post_physics = fuse
        env.build()
        env.seed(77)
        env.reset()
        g = torch.Generator().manual_seed(3)
        seq = []
        for t in range(80):
            o, r, te, tr, ex = env.step(torch.randn(n, 12, generator=g).to("cuda"))
            seq.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                        env.velocity_command._command.cpu().clone(), env.episode_length.cpu().clone(), env.max_episode_length.cpu().clone(),
                        env.reward_manager._episode_sums.cpu().clone(), env.robot.dof_pos.cpu().clone(), env.robot.quat.cpu().clone()))
        assert env._trace is not None and (env._trace.post_refs is not None) == fuse
        if fuse:
            static = post_variant == 2 and not obs_noise
            what = hip_backend.post_describe(env._trace.post_refs).split(":")[0]
            # (a noisy config matches no built-in program: the interpreter — or, when an earlier test of this process has compiled this
            # very structure at run time, tests/test_jit_programs.py, that program; GF_OPT_POST_VARIANT < 2 always selects the interpreter)
            assert what == "program 1 (go2_command_direction)" if static else (what == "program 0 (interpreter)" or (post_variant == 2 and "(jit_" in what)), what
        outs.append(seq)
    for t, (x, y) in enumerate(zip(*outs)):
        for k in (0, 1, 2, 3, 5, 6, 7, 8, 9, 10):
            assert torch.equal(x[k], y[k]), f"output {k} differs at step {t}"
        assert x[4].keys() == y[4].keys()
        for key in x[4]:
            assert abs(x[4][key] - y[4][key]) <= 1e-6 + 1e-6 * abs(y[4][key]), (t, key)


def _run_rough(dev, mode, n, steps=60):
    from envs import Go2RoughTerrainEnv

    env = Go2RoughTerrainEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3, scene_kwargs=dict(ang_noise=0.4, seed=9))
    env.trace_enabled = mode != "ordinary"
    env.fuse_post_physics = mode == "fused"
    env.build()
    env.seed(13)
    env.reset()
    g = torch.Generator().manual_seed(2)
    seq = []
    for t in range(steps):
        o, r, te, tr, ex = env.step(torch.randn(n, 12, generator=g).to(dev))
        seq.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                    env.robot.get_pos().cpu().clone(), env.robot.get_quat().cpu().clone(), env.velocity_command._command.cpu().clone(),
                    env.max_episode_length.cpu().clone(), env.foot_contact_manager.current_air_time.cpu().clone(),
                    env.reward_manager._episode_sums.cpu().clone()))
    return seq, env


def test_rough_terrain_traced_equals_ordinary_cpu(oracle_backend):
    a, _ = _run_rough("cpu", "ordinary", 70)
    b, env = _run_rough("cpu", "fused", 70)
    assert env._trace is not None and env._trace.post_refs is not None
    _same_h(a, b)
    assert sum(int(x[2].sum() + x[3].sum()) for x in a) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("post_variant", [0, 1, 2], indirect=True)
@pytest.mark.parametrize("n", [70, 4097])
def test_rough_terrain_fused_hip(hip_backend, post_variant, n):
    """Terrain spawn + base_height over the height field inside the fused launch (both kernel layouts) against the phase list."""
    a, _ = _run_rough("cuda", "ordinary", n)
    c, env_c = _run_rough("cuda", "fused", n)
    assert env_c._trace is not None and env_c._trace.post_refs is not None, "this config must take the fused kernel"
    _same_h(a, c)


def _run_humanoid(dev, mode, n, dofs, steps=50):
    from envs import HumanoidGaitLikeEnv

    env = HumanoidGaitLikeEnv(num_envs=n, dofs=dofs)
    env.trace_enabled = mode != "ordinary"
    env.fuse_post_physics = mode == "fused"
    env.build()
    env.seed(31)
    env.reset()
    g = torch.Generator().manual_seed(5)
    seq = []
    for t in range(steps):
        o, r, te, tr, ex = env.step(torch.randn(n, dofs, generator=g).to(dev))
        crit = ex["observations"]["critic"]
        seq.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                    crit.cpu().clone(), env.velocity_command._command.cpu().clone(), env.height_command._command.cpu().clone(),
                    env.max_episode_length.cpu().clone(), env.feet.current_air_time.cpu().clone(), env.reward_manager._episode_sums.cpu().clone()))
    return seq, env


def _same_h(a, b):
    for t, (x, y) in enumerate(zip(a, b)):
        for k in (0, 1, 2, 3, 5, 6, 7, 8, 9, 10):
            assert torch.equal(x[k], y[k]), f"output {k} differs at step {t}"
        assert x[4].keys() == y[4].keys(), f"log keys differ at step {t}"
        for key in x[4]:
            assert abs(x[4][key] - y[4][key]) <= 1e-6 + 1e-6 * abs(y[4][key]), (t, key)


@pytest.mark.parametrize("dofs", [12, 28])
def test_humanoid_config_traced_equals_ordinary_cpu(oracle_backend, dofs):
    a, _ = _run_humanoid("cpu", "ordinary", 50, dofs)
    b, env = _run_humanoid("cpu", "fused", 50, dofs)
    assert env._trace is not None and env._trace.post_refs is not None
    _same_h(a, b)
    assert sum(int(x[2].sum() + x[3].sum()) for x in a) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("post_variant", [0, 1, 2], indirect=True)
@pytest.mark.parametrize("dofs,n", [(12, 200), (28, 200), (28, 1), (12, 4097)])
def test_humanoid_config_fused_hip(hip_backend, post_variant, dofs, n):
    """Two command managers, two observation managers, three contact managers, 13 reward terms, D=28 variant."""
    a, _ = _run_humanoid("cuda", "ordinary", n, dofs)
    b, env_b = _run_humanoid("cuda", "unfused", n, dofs)
    c, env_c = _run_humanoid("cuda", "fused", n, dofs)
    assert env_b._trace is not None and env_b._trace.post_refs is None
    assert env_c._trace is not None and env_c._trace.post_refs is not None, "this config must take the fused kernel"
    _same_h(a, b)
    _same_h(a, c)


def test_static_program_selection_is_host_side(oracle_backend):
    """gf_post_physics_describe needs no GPU: the command_direction structure selects the static program, any structural change
    (here: observation noise, a history ring) falls back to the table interpreter."""
    from genesis_forge_amd import _native as nat

    hip = nat.HipBackend()  # loads libgf_step.so; describe() is host-only
    for kwargs, want in ((dict(), "program 1 (go2_command_direction)"), (dict(obs_noise=True), "program 0 (interpreter)"),
                         (dict(history=2), "program 0 (interpreter)")):
        env = Go2CommandDirectionEnv(num_envs=8, **kwargs)
        env.build()
        env.reset()
        for _ in range(3):
            env.step(torch.zeros(8, 12))
        assert env._trace is not None and env._trace.post_refs is not None
        text = hip.post_describe(env._trace.post_refs)
        assert text.startswith(want), text
        assert "n_rew = 6" in text


@pytest.mark.gpu
def test_hipgraph_replay_equals_plain_launches(hip_backend):
    """GF_OPT_GRAPH: the recorded step as one hipGraphLaunch (node arguments refreshed every step) gives the same bits."""
    from genesis_forge_amd import _native as nat

    a, _ = _run("cuda", True, n=1000)
    hip_backend.set_option(nat.GF_OPT_GRAPH, 1)
    try:
        b, env = _run("cuda", True, n=1000)
        assert env._trace is not None and env._trace.graph is not None and env._trace.graph.value, "no graph was built"
    finally:
        hip_backend.set_option(nat.GF_OPT_GRAPH, 0)
    _same(a, b)


def _run_simple(dev, trace, n=40, steps=14, edit_at=8):
    """examples/simple passes VIEWS of env.target_command to its tracking terms (environment.py:160,168); the reference sees
    later in-place edits of the base tensor, so the recorded step must read the views in place, not a frozen copy."""
    from envs import Go2SimpleEnv

    env = Go2SimpleEnv(num_envs=n, scene_kwargs=dict(seed=3, ang_noise=0.2))
    env.trace_enabled = trace
    env.build()
    env.seed(1)
    env.reset()
    g = torch.Generator().manual_seed(0)
    out = []
    for t in range(steps):
        if t == edit_at:
            env.target_command[:, 0] = 2.0
            env.target_command[:, 2] = -0.7
        o, r, *_ = env.step(torch.randn(n, 12, generator=g).to(dev))
        out.append((o.cpu().clone(), r.cpu().clone()))
    return out, env


def test_recorded_step_reads_command_views_in_place_cpu(oracle_backend):
    a, _ = _run_simple("cpu", False)
    b, env = _run_simple("cpu", True)
    assert env._trace is not None
    for t, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), f"step {t}"
    assert not torch.equal(a[7][1], a[9][1])


@pytest.mark.gpu
def test_recorded_step_reads_command_views_in_place_hip(hip_backend):
    a, _ = _run_simple("cuda", False, n=1000)
    b, env = _run_simple("cuda", True, n=1000)
    assert env._trace is not None and env._trace.post_refs is not None, "the simple config should still fuse (strided views are read in place)"
    for t, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), f"step {t}"


def test_recorded_step_follows_resample_time_changes(oracle_backend):
    """`CommandManager.resample_time_sec` is a settable property in the reference (command_manager.py:121-130): changing it
    in the middle of a run must reach the recorded step."""
    def run(trace):
        env = Go2CommandDirectionEnv(num_envs=48, max_episode_length_s=20, cmd_resample_s=5.0, scene_kwargs=dict(seed=3))
        env.trace_enabled = trace
        env.build()
        env.seed(2)
        env.reset()
        cmds = []
        for t in range(30):
            if t == 10:
                env.velocity_command.resample_time_sec = 0.1   # every 5 steps from now on
            env.step(torch.zeros(48, 12))
            cmds.append(env.velocity_command._command.clone())
        return cmds, env

    a, _ = run(False)
    b, env = run(True)
    assert env._trace is not None
    assert any(not torch.equal(a[t], a[t + 1]) for t in range(12, 29)), "the shorter period should resample within the run"
    for t, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"commands differ at step {t}"


def test_recorded_step_follows_live_manager_attributes(oracle_backend):
    """Attributes the reference re-reads every step — ObservationManager.noise, RewardManager.logging_enabled,
    env.set_max_episode_length() — must reach a recorded step when they are assigned in the middle of a run."""
    def run(trace):
        env = Go2CommandDirectionEnv(num_envs=40, max_episode_length_s=1, cmd_resample_s=0.3, scene_kwargs=dict(seed=4, ang_noise=0.3))
        env.trace_enabled = trace
        env.build()
        env.seed(6)
        env.reset()
        g = torch.Generator().manual_seed(0)
        out = []
        for t in range(36):
            if t == 10:
                env.observation_manager.noise = 0.05
            if t == 16:
                env.reward_manager.logging_enabled = False
            if t == 22:
                env.set_max_episode_length(0.4)
            o, r, te, tr, ex = env.step(torch.randn(40, 12, generator=g))
            out.append((o.clone(), r.clone(), te.clone(), tr.clone(), env.max_episode_length.clone(), {k: float(v) for k, v in ex["episode"].items()}))
        return out, env

    a, _ = run(False)
    b, env = run(True)
    assert env._trace is not None
    for t, (x, y) in enumerate(zip(a, b)):
        for k in range(5):
            assert torch.equal(x[k], y[k]), f"output {k} differs at step {t}"
        assert x[5] == y[5], f"log differs at step {t}"
    assert not any(k.startswith("Rewards /") for k in a[-1][5]) and int(a[-1][4].min()) < 30


@pytest.mark.parametrize("dofs", [7, 16])
def test_other_dof_counts_use_phase_chains_cpu(oracle_backend, dofs):
    """Other DOF counts: the step is still recorded; on the GPU every count up to 28 runs the fused launch."""
    a, _ = _run_humanoid("cpu", "ordinary", 50, dofs)
    b, env = _run_humanoid("cpu", "fused", 50, dofs)
    assert env._trace is not None
    _same_h(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("dofs,n", [(7, 300), (10, 65), (16, 1000), (20, 129), (8, 257), (24, 1000)])
def test_other_dof_counts_hip(hip_backend, oracle_lib_path, dofs, n):
    """Scalar-row (D % 4 != 0) variants of the reward / action / scene kernels (ordinary steps) and the fused launch's interpreter for
    7 / 8 / 10 / 16 / 20 / 24 DOF (recorded steps): recorded == ordinary on HIP, and HIP == oracle."""
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    a, _ = _run_humanoid("cuda", "ordinary", n, dofs)
    b, env = _run_humanoid("cuda", "fused", n, dofs)
    assert env._trace is not None
    assert env._trace.post_refs is not None, "every DOF count up to 28 runs the fused launch (ceil(D / 4) row chunks, the last one element by element)"
    _same_h(a, b)
    torch.cuda.synchronize()
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    try:
        ref, _ = _run_humanoid("cpu", "ordinary", n, dofs)
    finally:
        nat.set_backend(None)
        gs.set_device("cuda:0")
    for t, (x, y) in enumerate(zip(a, ref)):
        for k in (2, 3, 8):
            assert torch.equal(x[k], y[k]), f"integer state {k} differs at step {t}"
        for k in (0, 1, 5, 6, 7, 9, 10):
            assert torch.allclose(x[k], y[k], atol=1e-5, rtol=0), f"float state {k} differs at step {t}: {(x[k] - y[k]).abs().max()}"


def _run_with_user_resets(dev, trace, n=70, steps=60):
    """A training script's own resets between steps: the whole batch (env.reset()), and some envs by index list
    (env.reset([…]), managed_env.py:336-371), while the step is recorded."""
    env = Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=1, cmd_resample_s=0.3, contacts=True, history=2, obs_noise=True,
                                 scene_kwargs=dict(ang_noise=0.3, seed=3))
    env.trace_enabled = trace
    env.build()
    env.seed(5)
    env.reset()
    g = torch.Generator().manual_seed(0)
    outs = []
    for t in range(steps):
        if t == 20:
            obs, _ = env.reset()
            outs.append((obs.cpu().clone(),))
        if t in (33, 34, 50):
            env.reset([1, 5, n - 1] if t != 34 else torch.tensor([0, 2]))
        if t == 42:  # a manager method of the public API, through the very descriptor the recorded step replays
            env.velocity_command.resample_command([0, 3, n - 2])
        if t == 45:
            env.seed(99)                                  # a re-seed in mid-run
        if t == 47:
            env.foot_contacts.enabled = False             # manager toggles (base_manager.py: `enabled`)
            env.velocity_command.enabled = False
        if t == 52:
            env.foot_contacts.enabled = True
            env.velocity_command.enabled = True
        if t in (30, 48):
            outs.append((env.get_observations().cpu().clone(),))   # between steps: the last step's observation, nothing is launched
        o, r, te, tr, ex = env.step(torch.randn(n, 12, generator=g).to(dev))
        outs.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                     env.velocity_command._command.cpu().clone(), env.episode_length.cpu().clone(), env.reward_manager._episode_sums.cpu().clone()))
    return outs, env


def _same_user_resets(a, b):
    for t, (x, y) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for k, (u, v) in enumerate(zip(x, y)):
            if isinstance(u, dict):
                assert u == v, f"log differs at entry {t}: {u} vs {v}"
            else:
                assert torch.equal(u, v), f"output {k} differs at entry {t}"


def test_user_resets_between_recorded_steps_cpu(oracle_backend):
    a, _ = _run_with_user_resets("cpu", False)
    b, env = _run_with_user_resets("cpu", True)
    assert env._trace is not None
    _same_user_resets(a, b)


@pytest.mark.gpu
def test_user_resets_between_recorded_steps_hip(hip_backend):
    a, _ = _run_with_user_resets("cuda", False, n=1000)
    b, env = _run_with_user_resets("cuda", True, n=1000)
    assert env._trace is not None and env._trace.post_refs is not None
    _same_user_resets(a, b)


def test_two_recorded_envs_interleaved(oracle_backend):
    """A training env and an evaluation env alive at once (one backend, two recorded steps replayed alternately, one of them reset by
    its script in mid-run) behave exactly as each would alone."""
    def make(seed):
        env = Go2CommandDirectionEnv(num_envs=40, max_episode_length_s=0.6, cmd_resample_s=0.3, contacts=seed % 2 == 0,
                                     scene_kwargs=dict(ang_noise=0.3, seed=seed))
        env.build()
        env.seed(seed)
        env.reset()
        return env

    def drive(env, t, g):
        if t == 15 and env is not None and getattr(env, "_is_eval", False):
            env.reset()
        o, r, te, tr, ex = env.step(torch.randn(40, 12, generator=g))
        return (o.clone(), r.clone(), te.clone(), tr.clone(), {k: float(v) for k, v in ex["episode"].items()})

    alone = []
    for seed in (2, 3):
        env, g = make(seed), torch.Generator().manual_seed(seed)
        env._is_eval = seed == 3
        alone.append([drive(env, t, g) for t in range(30)])
    envs = [make(2), make(3)]
    envs[1]._is_eval = True
    gens = [torch.Generator().manual_seed(2), torch.Generator().manual_seed(3)]
    both = [[], []]
    for t in range(30):
        for i in (0, 1):
            both[i].append(drive(envs[i], t, gens[i]))
    assert envs[0]._trace is not None and envs[1]._trace is not None
    for i in (0, 1):
        for t, (x, y) in enumerate(zip(alone[i], both[i])):
            for k in range(4):
                assert torch.equal(x[k], y[k]), f"env {i}: output {k} differs at step {t}"
            assert x[4] == y[4], f"env {i}: log differs at step {t}"


def test_logs_read_late_across_ring_wraps(oracle_backend):
    """The statistics ring has 64 slots and logs are lazy: extras dicts kept from steps 10, 70, 100 and 149 of a 150-step recorded
    run, read only at the end (two wrap-arounds later), hold what the ordinary path logged at those steps; dicts nobody kept cost
    nothing."""
    def run(trace):
        env = Go2CommandDirectionEnv(num_envs=70, max_episode_length_s=0.5, cmd_resample_s=0.3, scene_kwargs=dict(ang_noise=0.3, seed=8))
        env.trace_enabled = trace
        env.build()
        env.seed(8)
        env.reset()
        g = torch.Generator().manual_seed(8)
        kept = {}
        for t in range(150):
            _, _, _, _, ex = env.step(torch.randn(70, 12, generator=g))
            if t in (10, 70, 100, 149):
                kept[t] = ex["episode"] if trace else {k: float(v) for k, v in ex["episode"].items()}
        return {t: {k: float(v) for k, v in d.items()} for t, d in kept.items()}, env

    want, _ = run(False)
    got, env = run(True)
    assert env._trace is not None
    assert want == got
    assert any(len(d) > 0 for d in want.values())


def _ring_run(dev, kind, output, n, steps=14):
    from genesis_forge_amd import tasks
    from genesis_forge_amd.managers import ObservationManager

    old = ObservationManager.default_output
    ObservationManager.default_output = output
    try:
        if kind == "gait":
            env = tasks.Go2GaitTrainingEnv(num_envs=n, max_episode_length_s=0.4, scene_kwargs=dict(ang_noise=0.3, seed=3, contact_prob=0.05))
        else:
            env = Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=0.4, cmd_resample_s=0.2, history=3, contacts=True, obs_noise=True,
                                         scene_kwargs=dict(ang_noise=0.3, seed=3))
        env.build()
    finally:
        ObservationManager.default_output = old
    env.seed(9)
    env.reset()
    g = torch.Generator().manual_seed(1)
    d = env.action_space.shape[0]
    outs = []
    for _ in range(steps):
        obs, rew, te, tr, ex = env.step(torch.randn(n, d, generator=g).to(dev))
        frames = [m.ordered(ex["observations"][m.name]).cpu().clone() for m in env.managers["observation"]]
        outs.append(frames + [rew.cpu().clone(), te.cpu().clone()])
    return outs, env


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,steps", [("go2_hist", 1000, 14), ("gait", 1000, 14), ("gait", 65536 + 37, 7)])
def test_in_place_history_ring_hip(hip_backend, kind, n, steps):
    """The in-place ring through the fused post-physics kernel (interpreter for the noisy Go2 config, the gait static program):
    only the new frame is written; gathered newest first it equals the default output bit for bit.  At 65 573 envs the gait
    policy history is 81 MB: the in-kernel shift of the static output runs with the streaming hint there (kObsStreamBytes)."""
    want, _ = _ring_run("cuda", kind, "static", n, steps)
    got, env = _ring_run("cuda", kind, "ring", n, steps)
    assert env._trace is not None and env._trace.post_refs is not None
    for t, (a, b) in enumerate(zip(want, got)):
        for k, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"output {k} differs at step {t}"


@pytest.mark.gpu
def test_in_place_history_ring_stand_alone_kernel_hip(hip_backend, monkeypatch):
    """… and through the stand-alone observation kernel (no recording: every phase its own launch)."""
    monkeypatch.setenv("GF_NO_TRACE", "1")
    want, e0 = _ring_run("cuda", "gait", "static", 130)
    got, e1 = _ring_run("cuda", "gait", "ring", 130)
    assert e0._trace is None and e1._trace is None
    for t, (a, b) in enumerate(zip(want, got)):
        for k, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"output {k} differs at step {t}"


def _window_check(dev, kind, n, steps, trace=True, slack=None):
    """output="window" against the default output, step by step: equal values in the reference's layout, a strided VIEW of one
    persistent buffer, intact for window_slack + 1 further observations."""
    from genesis_forge_amd.managers import ObservationManager

    old_slack = ObservationManager.window_slack
    ObservationManager.window_slack = slack
    try:
        want, _ = _ring_run(dev, kind, "fresh", n, steps)
        from genesis_forge_amd import tasks

        old = ObservationManager.default_output
        ObservationManager.default_output = "window"
        try:
            if kind == "gait":
                env = tasks.Go2GaitTrainingEnv(num_envs=n, max_episode_length_s=0.4, scene_kwargs=dict(ang_noise=0.3, seed=3, contact_prob=0.05))
            else:
                env = Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=0.4, cmd_resample_s=0.2, history=3, contacts=True, obs_noise=True,
                                             scene_kwargs=dict(ang_noise=0.3, seed=3))
            env.trace_enabled = trace
            env.build()
        finally:
            ObservationManager.default_output = old
    finally:
        ObservationManager.window_slack = old_slack
    env.seed(9)
    env.reset()
    g = torch.Generator().manual_seed(1)
    d = env.action_space.shape[0]
    held = []   # (step, manager index, the tensor the step returned, a copy of it)
    oms = env.managers["observation"]
    for t in range(steps):
        obs, rew, te, tr, ex = env.step(torch.randn(n, d, generator=g).to(dev))
        for k, m in enumerate(oms):
            o = ex["observations"][m.name]
            H, O = m._history_len, m._frame
            assert o.shape == (n, H * O) and o.stride() == (m._win_slots * O, 1) and o.untyped_storage().data_ptr() == m._win.untyped_storage().data_ptr()
            assert torch.equal(o.cpu(), want[t][k]), f"manager {k}: window differs from the fresh output at step {t}"
            held.append((t, k, o, o.cpu().clone()))
        assert torch.equal(obs, ex["observations"]["policy"]) and torch.equal(rew.cpu(), want[t][len(oms)])
        for t0, k, o, copy in held:   # every tensor handed out up to window_slack + 1 observations ago is still what it was
            if t - t0 <= oms[k]._win_cycle - oms[k]._history_len:
                assert torch.equal(o.cpu(), copy), f"the observation of step {t0} (manager {k}) changed after {t - t0} further steps"
    m = oms[0]
    t0, k, o, copy = next(h for h in held if h[1] == 0 and steps - 1 - h[0] > m._win_cycle)
    assert not torch.equal(o.cpu(), copy), "an observation older than the window's slack is expected to have been overwritten"
    return env


@pytest.mark.parametrize("trace", [True, False])
def test_history_window_output_cpu(oracle_backend, trace):
    env = _window_check("cpu", "go2_hist", 70, 24, trace=trace)
    assert (env._trace is not None) == trace
    env = _window_check("cpu", "gait", 70, 30, trace=trace, slack=2)
    assert (env._trace is not None) == trace


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,steps", [("go2_hist", 1000, 24), ("gait", 1000, 36), ("gait", 65536 + 37, 14)])
def test_history_window_output_hip(hip_backend, kind, n, steps):
    """The window through the fused post-physics kernel (interpreter for the noisy Go2 config, the gait static program — its structure
    is unchanged: the buffer's extra slots are a stride, GfObservationArgs.ring_slots)."""
    env = _window_check("cuda", kind, n, steps)
    assert env._trace is not None and env._trace.post_refs is not None
    if kind == "gait":
        text = hip_backend.post_describe(env._trace.post_refs)
        assert text.startswith("program") and "interpreter" not in text.split("\n")[0] and "gait" in text.split("\n")[0], text.split("\n")[0]


@pytest.mark.gpu
def test_history_window_stand_alone_kernel_hip(hip_backend):
    """… and through the stand-alone observation kernel (no recording: every phase its own launch)."""
    env = _window_check("cuda", "gait", 130, 30, trace=False)
    assert env._trace is None


def _run_dof_variant(dev, seed, trace):
    """The Go2 command stack over a synthetic D-joint robot with drawn D, env count, history, noise and episode / resample periods."""
    import random

    rnd = random.Random(9000 + seed)
    dofs = rnd.choice([7, 8, 10, 16, 19, 20, 23, 24, 28, 29, 32])
    n = rnd.choice([63, 64, 65, 257, 1000])
    env = Go2CommandDirectionEnv(num_envs=n, dofs=dofs, max_episode_length_s=rnd.choice([0.3, 0.5]), cmd_resample_s=rnd.choice([0.1, 0.3]),
                                 history=rnd.choice([None, 2, 3]), obs_noise=rnd.random() < 0.5,
                                 scene_kwargs=dict(ang_noise=rnd.choice([0.1, 0.3]), seed=seed))
    env.trace_enabled = trace
    env.build()
    env.seed(100 + seed)
    env.reset()
    g = torch.Generator().manual_seed(seed)
    outs = []
    for _ in range(36):
        o, r, te, tr, ex = env.step(torch.randn(n, dofs, generator=g).to(dev))
        outs.append((o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone(), {k: float(v) for k, v in ex["episode"].items()},
                     env.velocity_command._command.cpu().clone(), env.action_manager.get_dofs_position().cpu().clone()))
    return outs, env, dofs


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GF_DOF_SEEDS", "16"))))
def test_random_dof_variant_hip_equals_oracle(hip_backend, oracle_lib_path, seed):
    """7 … 32 DOF on the interpreter variants of the fused launch (rows as ceil(D / 4) float4 chunks; D % 4 != 0: the last chunk element by
    element, rows only dword aligned) — recorded on HIP == phase by phase on the oracle, and every variant resets envs."""
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    hip, env, dofs = _run_dof_variant("cuda", seed, True)
    torch.cuda.synchronize()
    assert env._trace is not None and env._trace.post_refs is not None
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    try:
        ref, _, _ = _run_dof_variant("cpu", seed, False)
    finally:
        nat.set_backend(None)
        gs.set_device("cuda:0")
    resets = 0
    for t, (x, y) in enumerate(zip(hip, ref)):
        for k in (2, 3):
            assert torch.equal(x[k], y[k]), f"mask {k} differs at step {t} (D = {dofs})"
        for k in (0, 1, 5, 6):
            assert torch.allclose(x[k], y[k], atol=1e-5, rtol=0), f"output {k} differs at step {t} (D = {dofs}): {(x[k] - y[k]).abs().max()}"
        assert set(x[4]) == set(y[4]), f"log keys differ at step {t}"
        for key in x[4]:
            assert abs(x[4][key] - y[4][key]) <= 1e-5 + 1e-5 * abs(y[4][key]), (t, key)
        resets += int(y[2].sum() + y[3].sum())
    assert resets > 0


def test_split_fused_launch_flags_are_checked_by_the_library(oracle_backend):
    """GF_POST_NO_RESET / GF_POST_OBSERVE_ONLY (the two ends of a step whose reset runs through user code): the descriptors a recorded
    override env builds pass the HIP library's own validation (gf_post_physics_check packs them — host work, no GPU), and the
    combinations the flags exclude are refused."""
    import copy

    import envs
    from genesis_forge_amd import _native as nat

    env = envs.Go2GaitTrainingCurriculumEnv(num_envs=70, max_episode_length_s=0.3, scene_kwargs=dict(ang_noise=0.35, seed=11, contact_prob=0.05))
    env.build()
    env.seed(3)
    env.reset()
    g = torch.Generator().manual_seed(1)
    for _ in range(40):
        env.step(torch.randn(70, 12, generator=g))
    tr = env._trace
    assert tr is not None and tr.tail_python and tr.post_refs is not None and tr.tail_seg.get("obs", {}).get("fused_obs")
    hip = nat.HipBackend()   # loads the library; nothing is launched
    front, back = tr.post_refs, tr._tail_refs
    assert front.flags == nat.GF_POST_NO_RESET and not front.reset and front.num_observe == 0 and front.num_gait == 1
    assert back.flags == nat.GF_POST_OBSERVE_ONLY and back.num_observe == 2 and not back.reward
    assert hip.post_check(front) and hip.post_check(back)

    def variant(refs, **kw):
        r = nat.GfPostRefs.from_buffer_copy(refs)
        for k, v in kw.items():
            setattr(r, k, v)
        return r

    assert not hip.post_check(variant(front, flags=nat.GF_POST_NO_RESET | nat.GF_POST_OBSERVE_ONLY))
    assert not hip.post_check(variant(front, flags=0))                      # without the flag the reset descriptor is required
    assert not hip.post_check(variant(front, num_observe=1))                # nothing is observed in front of the reset
    assert not hip.post_check(variant(back, reward=front.reward))           # the observation-only launch has no reward phase
    assert not hip.post_check(variant(back, num_command=1))
    assert not hip.post_check(variant(back, num_observe=0))
