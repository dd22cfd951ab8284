#!/usr/bin/env python3
"""Step time of a config with one user-level (Python) term — a reward lambda (default) or an observation item (`obs`): recorded step
(cut around the call) vs phase by phase — or (`manager`) a user-defined CommandManager CLASS with its own step() / reset(), a reward
term and an observation item reading it (the shape of the reference's examples/gait_trainer/gait_command_manager.py) — or (`classes`)
user-defined RewardManager and TerminationManager CLASSES whose step() wraps the library's (round 4: python phases of a recorded step).
or (`action`) a user-defined ACTION manager class overriding handle_actions(), the reference's extension point.
or (`curriculum`) the reference's documented curriculum recipe: step() assigns termination params on EVERY step, mostly unchanged values.
    python tools/bench_user_term.py [num_envs] [reward|obs|manager|classes|obsclass|action|curriculum|anneal]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from genesis_forge_amd import gs
from genesis_forge_amd.tasks import Go2CommandDirectionEnv


KIND = sys.argv[2] if len(sys.argv) > 2 else "reward"


def run(n, trace, steps=400):
    if trace:
        os.environ.pop("GF_NO_TRACE", None)
    else:
        os.environ["GF_NO_TRACE"] = "1"

    env_cls = Go2CommandDirectionEnv
    if KIND == "obsclass":
        class env_cls(Go2CommandDirectionEnv):   # an env-level get_observations() override
            def get_observations(self):
                o = super().get_observations()
                return None if o is None else o * 0.5
    if KIND == "curriculum":   # docs/guide/managers/termination.md "Curriculum-Based Termination", verbatim in shape
        class env_cls(Go2CommandDirectionEnv):   # noqa: F811
            def step(self, actions):
                self.update_curriculum()
                return super().step(actions)

            def update_curriculum(self):
                limit = 10.0 if self.step_count > 200 else 12.0
                self.termination_manager.term_cfg["fall_over"].params["limit_angle"] = limit
    if KIND == "anneal":   # a reward weight annealed on EVERY step: each step refreshes the recorded step's term table in place
        class env_cls(Go2CommandDirectionEnv):   # noqa: F811
            def step(self, actions):
                self.reward_manager.cfg["action_rate"].weight = -0.005 * (1.0 + 1e-4 * self.step_count)
                return super().step(actions)
    if KIND == "action":   # a low-pass on the incoming actions in front of the library's processing
        from genesis_forge_amd.managers import PositionActionManager

        class SmoothedActions(PositionActionManager):
            def handle_actions(self, actions):
                prev = getattr(self, "_lp", None)
                self._lp = actions.clone() if prev is None else torch.lerp(actions, prev, 0.7)
                return super().handle_actions(self._lp)

        class env_cls(Go2CommandDirectionEnv):   # noqa: F811
            action_cls = SmoothedActions
    env = env_cls(num_envs=n, scene_kwargs=dict(ang_noise=0.05, seed=1))
    cfg_add = {"user_height": {"weight": 0.3, "fn": lambda env: torch.tanh(env.robot.get_pos()[:, 2])}}
    orig = env.config

    def config():
        orig()
        from genesis_forge_amd.managers import RewardManager
        rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in env.reward_manager.cfg.items()}
        if KIND == "reward":
            rc.update(cfg_add)
        env.managers["reward"] = None
        env.reward_manager = RewardManager(env, logging_enabled=True, cfg=rc)
        if KIND == "obs":   # a Python-level observation item: the manager observes behind the fused launch, after the callable
            from genesis_forge_amd.managers import ObservationManager
            om = env.observation_manager
            oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
            oc["user_xy"] = {"fn": lambda env: env.robot.get_pos()[:, :2] * 2.0}
            env.managers["observation"].remove(om)
            env.observation_manager = ObservationManager(env, cfg=oc)

    if KIND == "manager":
        from genesis_forge_amd.managers import CommandManager, ObservationManager, RewardManager

        class PhaseClock(CommandManager):
            def __init__(self, env):
                super().__init__(env, range=(0.5, 1.5), resample_time_sec=0.2)
                self.phase = torch.zeros(env.num_envs, device=gs.device)

            def step(self):
                super().step()
                self.phase = (self.phase + self.env.dt * self._command[:, 0]) % 1.0

            def reset(self, env_ids=None):
                super().reset(env_ids)
                if env_ids is None:
                    self.phase = torch.zeros_like(self.phase)
                else:
                    self.phase[env_ids] = 0.0

        def config():   # noqa: F811
            orig()
            env.clock = PhaseClock(env)
            rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in env.reward_manager.cfg.items()}
            rc["in_phase"] = {"weight": 0.2, "fn": lambda env: torch.cos(6.2831853 * env.clock.phase)}
            env.managers["reward"] = None
            env.reward_manager = RewardManager(env, logging_enabled=True, cfg=rc)
            om = env.observation_manager
            oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
            oc["clock"] = {"fn": lambda env: torch.stack([torch.sin(6.2831853 * env.clock.phase), env.clock.phase], dim=-1)}
            env.managers["observation"].remove(om)
            env.observation_manager = ObservationManager(env, cfg=oc)

    if KIND == "classes":
        from genesis_forge_amd.managers import RewardManager, TerminationManager

        class CappedRewards(RewardManager):
            def step(self):
                r = super().step()
                r.clamp_(min=-0.5)
                return r

        class GracefulTerminations(TerminationManager):
            def step(self):
                te, tr = super().step()
                te &= self.env.episode_length > 2
                return te, tr

        def config():   # noqa: F811
            orig()
            rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in env.reward_manager.cfg.items()}
            env.managers["reward"] = None
            env.reward_manager = CappedRewards(env, logging_enabled=True, cfg=rc)
            tc = {k: {"fn": v.fn, "params": dict(v.params), "time_out": v.time_out} for k, v in env.termination_manager.term_cfg.items()}
            env.managers["termination"] = None
            env.termination_manager = GracefulTerminations(env, logging_enabled=True, term_cfg=tc)

    if KIND == "obsclass":   # a user ObservationManager class + an env-level get_observations() override (round 4: python phases behind the fused launch)
        from genesis_forge_amd.managers import ObservationManager

        class ClippedObs(ObservationManager):
            def get_observations(self):
                return super().get_observations().clamp(-3.0, 3.0)

        def config():   # noqa: F811
            orig()
            om = env.observation_manager
            oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
            env.managers["observation"].remove(om)
            env.observation_manager = ClippedObs(env, cfg=oc)


    env.config = config
    env.build()
    env.seed(1)
    env.reset()
    acts = [torch.randn(n, 12, device=gs.device) for _ in range(4)]
    for i in range(30):
        env.step(acts[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e6
    if os.environ.get("GF_PROFILE") == "1" and trace:   # where the host time of the recorded step goes
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(steps):
            env.step(acts[i % 4])
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
    tr = env._trace
    return dt, (tr is not None), (len(tr.splits) if tr else None), (tr.n_ops if tr else None)


if __name__ == "__main__":
    gs.set_device("cuda:0")
    for n in ([int(sys.argv[1])] if len(sys.argv) > 1 and sys.argv[1].isdigit() else [4096, 65536]):
        for trace in (True, False):
            dt, rec, splits, ops = run(n, trace)
            print(f"{KIND:6s} N={n:6d} recorded={rec!s:5s} cuts={splits} ops={ops}  {dt:8.1f} us/step  {n / dt:8.1f} M env-steps/s", flush=True)
