"""The six task configs the reference ships (examples/*/environment.py) as parity cases.

One table drives three things, so they cannot drift apart:
  * tools/gen_golden.py runs the REFERENCE package on the reference's own example file (unchanged) over the synthetic
    scene and records tests/golden/traj_ex_<name>.npz;
  * tests/test_examples.py replays the fixture through this package's restatement of the same task config
    (tests/envs.py) on the CPU oracle and on the HIP kernels;
  * tests/test_examples.py::test_reference_example_file_drops_in loads the reference's example FILE itself under the
    ``genesis_forge`` alias (only where /root/reference exists) — "task configs drop in unchanged".

``scene``: SyntheticScene options (the examples construct ``gs.Scene(...)`` themselves; the options are injected through
compat.SCENE_OVERRIDES).  ``resample``: command managers' ``resample_time_sec`` set through the public property after
build() so resampling happens inside short trajectories.  ``events``: {step: name} curriculum actions.
"""

CASES = {
    # BASELINE config 1 (examples/simple): static target command, clip ±100, no command manager, O = 45
    "simple": dict(n=8, steps=180, episode_s=1.5, dofs=12,
                   scene=dict(ang_noise=0.33, lin_noise=0.05, seed=61), resample={}, events={}),
    # BASELINE config 2 (examples/command_direction)
    "command_direction": dict(n=8, steps=180, episode_s=1.5, dofs=12,
                              scene=dict(ang_noise=0.33, lin_noise=0.05, seed=62), resample={"velocity_command": 0.6}, events={}),
    # examples/contacts: feet_air_time on the calves, bad_orientation 20°
    "contacts": dict(n=8, steps=180, episode_s=1.5, dofs=12,
                     scene=dict(ang_noise=0.5, lin_noise=0.05, seed=63, contact_prob=0.3, contact_force=30.0, max_collision_pairs=12),
                     resample={"velocity_command": 0.6}, events={}),
    # BASELINE config 3 (examples/rough_terrain): height-field terrain, terrain spawn, out_of_bounds, two contact managers
    "rough_terrain": dict(n=8, steps=180, episode_s=1.5, dofs=12,
                          scene=dict(ang_noise=0.8, lin_noise=0.05, seed=64, contact_prob=0.3, contact_force=30.0, max_collision_pairs=12),
                          resample={"velocity_command": 0.6}, events={}),
    # BASELINE config 4 (examples/berkeley_humanoid): 12 actuated joints in the reference's MJCF, torso contact termination
    "berkeley_humanoid": dict(n=8, steps=180, episode_s=1.5, dofs=12,
                              scene=dict(ang_noise=0.3, lin_noise=0.05, seed=65, contact_prob=0.03, contact_force=6.0, max_collision_pairs=10),
                              resample={"velocity_command": 0.6}, events={}),
    # BASELINE config 5 (examples/gait_trainer): velocity + gait command managers, three contact managers, policy (62x5)
    # and critic (16x5) observations, user-level gait rewards; the curriculum events widen the gait set / ranges
    "gait_trainer": dict(n=8, steps=220, episode_s=1.5, dofs=12,
                         scene=dict(ang_noise=0.45, lin_noise=0.05, seed=66, contact_prob=0.03, contact_force=4.0, max_collision_pairs=14),
                         resample={"velocity_command": 0.5, "gait_command_manager": 0.7},
                         events={40: "more_gaits", 41: "more_gaits", 42: "wider_ranges", 90: "more_gaits", 91: "more_gaits",
                                 92: "more_gaits", 93: "wider_ranges"}),
}

# The same example files at more than one 64-env tile (fixtures traj_ex_<key>.npz, `example` names the reference directory):
# the n = 8 cases above only ever exercise one partial tile of the HIP kernels.  130 envs = two full tiles + a 2-env tail,
# 70 = one full tile + 6; short episodes (0.5 s = 25 steps +- jitter) so time-outs, resets and resamples all occur.
for _ex, _n, _steps in (("command_direction", 130, 56), ("rough_terrain", 130, 56), ("berkeley_humanoid", 130, 48), ("gait_trainer", 70, 44)):
    _base = CASES[_ex]
    CASES[f"{_ex}_n{_n}"] = dict(_base, example=_ex, n=_n, steps=_steps, episode_s=0.5,
                                  resample={k: 0.2 + 0.1 * i for i, k in enumerate(_base["resample"])},
                                  events={5: "more_gaits", 6: "more_gaits", 7: "wider_ranges", 20: "more_gaits"} if _base["events"] else {})


# … and BASELINE config 5's example at the size one rank of an 8-GPU strong-scaling run holds (8 192 envs = 128 tiles): the fixture is
# COMPACT (helpers.compact_example: per-step sums, sums of squares and a strided 64-env sample of every recorded field; the actions are
# regenerated from the generator's stream) — the full trajectory is 300 MB.  The reference run takes ≈ 10 minutes (three ContactManagers
# through the serial Taichi emulation): `python tools/gen_golden.py examples gait_trainer_n8192`.
CASES["gait_trainer_n8192"] = dict(CASES["gait_trainer"], example="gait_trainer", n=8192, steps=20, episode_s=0.3, compact=True,
                                   resample={"velocity_command": 0.2, "gait_command_manager": 0.3},
                                   events={5: "more_gaits", 6: "more_gaits", 7: "wider_ranges", 12: "more_gaits"})
# the headline config's example file at the headline size, and BASELINE config 4's at its size (compact fixtures as above)
CASES["command_direction_n65536"] = dict(CASES["command_direction"], example="command_direction", n=65536, steps=20, episode_s=0.3, compact=True,
                                         resample={"velocity_command": 0.2}, events={})
CASES["berkeley_humanoid_n8192"] = dict(CASES["berkeley_humanoid"], example="berkeley_humanoid", n=8192, steps=20, episode_s=0.3, compact=True,
                                        resample={"velocity_command": 0.2}, events={})


def example_of(key: str) -> str:
    """The reference example directory (and restated task config) a case runs."""
    return CASES[key].get("example", key)


GOLDEN_SEED = 20251017


def apply_event(env, what: str) -> None:
    """Curriculum actions through the gait manager's public methods (examples/gait_trainer/gait_command_manager.py:146-180)."""
    g = env.gait_command_manager
    if what == "more_gaits":
        g.increment_num_gaits()
    elif what == "wider_ranges":
        g.increment_gait_period_range()
        g.increment_foot_clearance_range()
    else:
        raise KeyError(what)
