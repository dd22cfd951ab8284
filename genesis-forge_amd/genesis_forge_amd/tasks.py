"""Task configs in the reference's style — this package's restatements of the reference's six shipped examples
(examples/*/environment.py: same managers, cfg dicts, weights; pinned to the reference by tests/test_examples.py) plus two
stress configs — and ``BASELINE_CONFIGS``, the table of workloads bench.py, tools/bench_configs.py and the parity tests at
the timed sizes (tests/test_bench_parity.py) share."""
import os
import torch

from genesis_forge_amd import ManagedEnvironment
from genesis_forge_amd.managers import (ContactManager, EntityManager, ObservationManager, PositionActionManager, RewardManager,
                                        TerminationManager, VelocityCommandManager)
from genesis_forge_amd.mdp import reset, rewards, terminations, observations
from genesis_forge_amd.scene import SyntheticScene, morphs

#: the scene class the task configs below build on.  Default: the synthetic stand-in with persistent, shared state buffers;
#: ``use_scene(cls)`` swaps in another one with the same constructor (tests/genesis_like.py: Genesis' public surface only)
_SCENE_CLS = [SyntheticScene]


def new_scene(**kw):
    return _SCENE_CLS[-1](**kw)


class use_scene:
    """``with use_scene(GenesisLikeScene): env = Go2CommandDirectionEnv(...)`` — envs constructed inside build on ``cls``."""

    def __init__(self, cls):
        self.cls = cls

    def __enter__(self):
        _SCENE_CLS.append(self.cls)
        return self

    def __exit__(self, *exc):
        _SCENE_CLS.pop()


INITIAL_BODY_POSITION = [0.0, 0.0, 0.4]
INITIAL_QUAT = [1.0, 0.0, 0.0, 0.0]


GO2_JOINTS = ["FL_.*_joint", "FR_.*_joint", "RL_.*_joint", "RR_.*_joint"]
GO2_DEFAULT_POS = {".*_hip_joint": 0.0, "FL_thigh_joint": 0.8, "FR_thigh_joint": 0.8, "RL_thigh_joint": 1.0, "RR_thigh_joint": 1.0,
                   ".*_calf_joint": -1.5}


def _std_obs(self, velocity_cmd=True, scales=None):
    """The observation layout every shipped Go2 / humanoid example uses (e.g. examples/command_direction/environment.py:223-247)."""
    sc = scales or {}
    cfg = {}
    if velocity_cmd:
        cfg["velocity_cmd"] = {"fn": self.velocity_command.observation}
    cfg.update({
        "angle_velocity": {"fn": lambda env: self.robot_manager.get_angular_velocity(), "scale": sc.get("ang", 1.0)},
        "linear_velocity": {"fn": lambda env: self.robot_manager.get_linear_velocity(), "scale": sc.get("lin", 1.0)},
        "projected_gravity": {"fn": lambda env: self.robot_manager.get_projected_gravity()},
        "dof_position": {"fn": lambda env: self.action_manager.get_dofs_position()},
        "dof_velocity": {"fn": lambda env: self.action_manager.get_dofs_velocity(), "scale": 0.05},
        "actions": {"fn": lambda env: self.action_manager.get_actions()},
    })
    return cfg


class Go2SimpleEnv(ManagedEnvironment):
    """BASELINE config 1 (cf. examples/simple/environment.py:95-243): static target command, action clip ±100, no command
    manager, observation scales 0.25 / 2.0 / 0.05 (O = 45)."""

    def __init__(self, num_envs=1, dt=1 / 50, max_episode_length_s=20, scene_kwargs=None):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.1)
        from genesis_forge_amd import gs
        self.target_command = torch.zeros((self.num_envs, 3), device=gs.device, dtype=gs.tc_float)
        self.target_command[:, 0] = 0.5
        self.scene = new_scene(dt=self.dt, substeps=2, **dict(dict(max_collision_pairs=30), **(scene_kwargs or {})))
        self.terrain = self.scene.add_entity(morphs.Plane())
        self.robot = self.scene.add_entity(morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=INITIAL_BODY_POSITION, quat=INITIAL_QUAT))

    def config(self):
        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={
            "position": {"fn": reset.position, "params": {"position": INITIAL_BODY_POSITION, "quat": INITIAL_QUAT, "zero_velocity": True}}})
        self.action_manager = PositionActionManager(self, joint_names=GO2_JOINTS, default_pos=GO2_DEFAULT_POS, scale=0.25,
                                                    clip=(-100.0, 100.0), use_default_offset=True, pd_kp=20, pd_kv=0.5)
        RewardManager(self, logging_enabled=True, cfg={
            "base_height_target": {"weight": -50.0, "fn": rewards.base_height, "params": {"target_height": 0.3, "entity_attr": "robot"}},
            "tracking_lin_vel": {"weight": 1.0, "fn": rewards.command_tracking_lin_vel,
                                 "params": {"command": self.target_command[:, :2], "entity_manager": self.robot_manager}},
            "tracking_ang_vel": {"weight": 0.2, "fn": rewards.command_tracking_ang_vel,
                                 "params": {"commanded_ang_vel": self.target_command[:, 2], "entity_manager": self.robot_manager}},
            "lin_vel_z": {"weight": -1.0, "fn": rewards.lin_vel_z_l2, "params": {"entity_manager": self.robot_manager}},
            "action_rate": {"weight": -0.005, "fn": rewards.action_rate_l2},
            "similar_to_default": {"weight": -0.1, "fn": rewards.dof_similar_to_default, "params": {"action_manager": self.action_manager}},
        })
        self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg={
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "fall_over": {"fn": terminations.bad_orientation, "params": {"limit_angle": 10.0, "entity_manager": self.robot_manager}},
        })
        ObservationManager(self, cfg=_std_obs(self, velocity_cmd=False, scales={"ang": 0.25, "lin": 2.0}))


class Go2ContactsEnv(ManagedEnvironment):
    """cf. examples/contacts/environment.py:95-290: feet_air_time on the calves (threshold 0.5 s), flat_orientation, 20° limit."""

    def __init__(self, num_envs=1, dt=1 / 50, max_episode_length_s=20, scene_kwargs=None):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.1)
        self.scene = new_scene(dt=self.dt, substeps=2, **dict(dict(max_collision_pairs=30), **(scene_kwargs or {})))
        self.terrain = self.scene.add_entity(morphs.Plane())
        self.robot = self.scene.add_entity(morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=[0.0, 0.0, 0.35], quat=INITIAL_QUAT))

    def config(self):
        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={
            "position": {"fn": reset.position, "params": {"position": [0.0, 0.0, 0.35], "quat": INITIAL_QUAT}}})
        self.action_manager = PositionActionManager(self, joint_names=GO2_JOINTS, default_pos=GO2_DEFAULT_POS, scale=0.5,
                                                    use_default_offset=True, pd_kp=20, pd_kv=0.5)
        self.velocity_command = VelocityCommandManager(
            self, range={"lin_vel_x": [-1.0, 1.0], "lin_vel_y": [0, 0], "ang_vel_z": [-0.5, 0.5]}, standing_probability=0.0,
            resample_time_sec=5.0, debug_visualizer=True, debug_visualizer_cfg={"envs_idx": [0]})
        self.foot_contact_manager = ContactManager(self, link_names=[".*_calf"], track_air_time=True, air_time_contact_threshold=5.0)
        em, vc = self.robot_manager, self.velocity_command
        RewardManager(self, logging_enabled=True, cfg={
            "foot_air_time": {"weight": 2.5, "fn": rewards.feet_air_time,
                              "params": {"contact_manager": self.foot_contact_manager, "vel_cmd_manager": vc, "time_threshold": 0.5}},
            "tracking_lin_vel": {"weight": 1.0, "fn": rewards.command_tracking_lin_vel, "params": {"vel_cmd_manager": vc, "entity_manager": em}},
            "tracking_ang_vel": {"weight": 0.5, "fn": rewards.command_tracking_ang_vel, "params": {"vel_cmd_manager": vc, "entity_manager": em}},
            "lin_vel_z": {"weight": -1.0, "fn": rewards.lin_vel_z_l2, "params": {"entity_manager": em}},
            "ang_vel_xy": {"weight": -0.05, "fn": rewards.ang_vel_xy_l2, "params": {"entity_manager": em}},
            "action_rate": {"weight": -0.005, "fn": rewards.action_rate_l2},
            "similar_to_default": {"weight": -0.1, "fn": rewards.dof_similar_to_default, "params": {"action_manager": self.action_manager}},
            "flat_orientation": {"weight": -2.5, "fn": rewards.flat_orientation_l2},
        })
        self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg={
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "fall_over": {"fn": terminations.bad_orientation, "params": {"limit_angle": 20.0, "entity_manager": em}},
        })
        ObservationManager(self, cfg=_std_obs(self))


class BerkeleyHumanoidEnv(ManagedEnvironment):
    """BASELINE config 4 (cf. examples/berkeley_humanoid/environment.py:82-274): the reference's MJCF has 12 actuated joints;
    torso contact termination (default threshold 1.0), feet_air_time clamped to 0.2 … 0.5 s."""

    POS = [0.0, 0.0, 0.515]

    def __init__(self, num_envs=1, dt=1 / 50, max_episode_length_s=20, scene_kwargs=None):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.1)
        self.scene = new_scene(dt=self.dt, substeps=2, **(scene_kwargs or {}))
        self.terrain = self.scene.add_entity(morphs.Plane())
        self.robot = self.scene.add_entity(morphs.MJCF(file="./model/berkeley_humanoid.xml", pos=self.POS, quat=INITIAL_QUAT))

    def config(self):
        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={
            "position": {"fn": reset.position, "params": {"position": self.POS, "quat": INITIAL_QUAT, "zero_velocity": True}}})
        self.action_manager = PositionActionManager(
            self, joint_names=[".*"],
            default_pos={"LL_HR": -0.071, "LR_HR": 0.071, "LL_HAA": 0.103, "LR_HAA": -0.103, "LL_HFE": -0.463, "LR_HFE": -0.463,
                         "LL_KFE": 0.983, "LR_KFE": 0.983, "LL_FFE": -0.350, "LR_FFE": -0.350, "LL_FAA": 0.126, "LR_FAA": -0.126},
            scale=0.5, use_default_offset=True, pd_kp=15.0, pd_kv=1.0,
            max_force={".*_HR": 20.0, ".*_HAA": 20.0, ".*_HFE": 30.0, ".*_KFE": 30.0, ".*_FFE": 20.0, ".*_FAA": 5.0})
        self.velocity_command = VelocityCommandManager(
            self, range={"lin_vel_x": [0.0, 1.0], "lin_vel_y": [0.0, 0.0], "ang_vel_z": [-0.5, 0.5]}, standing_probability=0.02,
            resample_time_sec=5.0, debug_visualizer=True, debug_visualizer_cfg={"envs_idx": [0], "arrow_offset": 0.12})
        self.torso_contact_manager = ContactManager(self, link_names=["torso"])
        self.feet_contact_manager = ContactManager(self, link_names=[".*_faa"], track_air_time=True)
        em, vc = self.robot_manager, self.velocity_command
        RewardManager(self, logging_enabled=True, cfg={
            "tracking_lin_vel": {"weight": 1.0, "fn": rewards.command_tracking_lin_vel, "params": {"vel_cmd_manager": vc, "entity_manager": em}},
            "tracking_ang_vel": {"weight": 0.5, "fn": rewards.command_tracking_ang_vel, "params": {"vel_cmd_manager": vc, "entity_manager": em}},
            "lin_vel_z": {"weight": -2.0, "fn": rewards.lin_vel_z_l2, "params": {"entity_manager": em}},
            "ang_vel_xy_l2": {"weight": -0.05, "fn": rewards.ang_vel_xy_l2, "params": {"entity_manager": em}},
            "action_rate": {"weight": -0.005, "fn": rewards.action_rate_l2},
            "similar_to_default": {"weight": -0.05, "fn": rewards.dof_similar_to_default, "params": {"action_manager": self.action_manager}},
            "feet_air_time": {"weight": 2.0, "fn": rewards.feet_air_time,
                              "params": {"time_threshold": 0.2, "time_threshold_max": 0.5, "contact_manager": self.feet_contact_manager,
                                         "vel_cmd_manager": vc}},
        })
        self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg={
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "torso_contact": {"fn": terminations.contact_force, "params": {"contact_manager": self.torso_contact_manager}},
        })
        ObservationManager(self, cfg=_std_obs(self))


class Go2GaitTrainingEnv(ManagedEnvironment):
    """BASELINE config 5 (cf. examples/gait_trainer/environment.py:26-380): velocity + gait command managers, three contact
    managers, the gait manager's two reward methods, policy (62 x 5) and critic (16 x 5) observations, and — with
    ``curriculum=True`` — the example's ``reset`` override that widens the gait set from ``last_episode_mean_reward``."""

    CHECK_EVERY = 100

    def __init__(self, num_envs=1, dt=1 / 50, max_episode_length_s=20, scene_kwargs=None, curriculum=True):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.4)
        self._curriculum = curriculum
        self._next_curriculum_check_step = self.CHECK_EVERY
        self.scene = new_scene(dt=self.dt, substeps=2, **dict(dict(max_collision_pairs=60), **(scene_kwargs or {})))
        self.terrain = self.scene.add_entity(morphs.Plane())
        self.robot = self.scene.add_entity(morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=INITIAL_BODY_POSITION, quat=INITIAL_QUAT,
                                                       links_to_keep=["FL_foot", "FR_foot", "RL_foot", "RR_foot"]))

    def config(self):
        from genesis_forge_amd.managers import GaitCommandManager

        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={
            "position": {"fn": reset.position, "params": {"position": INITIAL_BODY_POSITION, "quat": INITIAL_QUAT, "zero_velocity": True}}})
        self.action_manager = PositionActionManager(self, joint_names=GO2_JOINTS, default_pos=GO2_DEFAULT_POS, scale=0.25,
                                                    use_default_offset=True, pd_kp=20, pd_kv=0.5)
        self.foot_contact_manager = ContactManager(self, link_names=[".*_foot"], air_time_contact_threshold=1.0)
        self.body_contact_manager = ContactManager(self, link_names=["base"], air_time_contact_threshold=1.0)
        self.bad_contact_manager = ContactManager(self, link_names=[".*_thigh", ".*_calf"])
        self.velocity_command = VelocityCommandManager(
            self, range={"lin_vel_x": [-1.0, 1.0], "lin_vel_y": [0.0, 0.0], "ang_vel_z": [-1.0, 1.0]}, standing_probability=0.00,
            resample_time_sec=3.0)
        self.gait_command_manager = GaitCommandManager(
            self, foot_names={"FL": "FL_foot", "FR": "FR_foot", "RL": "RL_foot", "RR": "RR_foot"}, resample_time_sec=4.0)
        em, vc, gait = self.robot_manager, self.velocity_command, self.gait_command_manager
        self.reward_manager = RewardManager(self, logging_enabled=True, cfg={
            "gait_phase_reward": {"weight": 1.5, "fn": gait.gait_phase_reward, "params": {"contact_manager": self.foot_contact_manager}},
            "foot_height_reward": {"weight": 0.9, "fn": gait.foot_height_reward},
            "base_height_target": {"weight": -25.0, "fn": rewards.base_height, "params": {"target_height": 0.35, "entity_attr": "robot"}},
            "tracking_lin_vel": {"weight": 1.0, "fn": rewards.command_tracking_lin_vel, "params": {"vel_cmd_manager": vc, "entity_manager": em}},
            "tracking_ang_vel": {"weight": 0.5, "fn": rewards.command_tracking_ang_vel, "params": {"vel_cmd_manager": vc, "entity_manager": em}},
            "body_acceleration": {"weight": -0.1, "fn": rewards.body_acceleration_exp, "params": {"entity_manager": em}},
            "lin_vel_z": {"weight": -0.1, "fn": rewards.lin_vel_z_l2, "params": {"entity_manager": em}},
            "action_rate": {"weight": -0.01, "fn": rewards.action_rate_l2},
            "bad_contact": {"weight": -1.0, "fn": rewards.contact_force, "params": {"contact_manager": self.bad_contact_manager}},
        })
        self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg={
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "fall_over": {"fn": terminations.bad_orientation, "params": {"limit_angle": 20.0, "entity_manager": em}},
            "body_contact": {"fn": terminations.contact_force, "params": {"contact_manager": self.body_contact_manager, "threshold": 1.0}},
        })
        ocfg = {"gait_command": {"fn": gait.observation}}
        ocfg.update(_std_obs(self))
        ObservationManager(self, name="policy", history_len=5, cfg=ocfg)
        ObservationManager(self, name="critic", history_len=5, cfg={
            "foot_contact_force": {"fn": observations.contact_force, "params": {"contact_manager": self.foot_contact_manager}},
            "dof_force": {"fn": observations.entity_dofs_force, "params": {"action_manager": self.action_manager}, "scale": 0.1},
        })

    def update_curriculum(self):
        """cf. examples/gait_trainer/environment.py:354-380"""
        if self.step_count < self._next_curriculum_check_step:
            return
        self._next_curriculum_check_step = self.step_count + self.CHECK_EVERY
        if self.reward_manager.last_episode_mean_reward("gait_phase_reward", before_weight=True) > 0.75:
            self.gait_command_manager.increment_num_gaits()
            self.gait_command_manager.increment_gait_period_range()
        if self.reward_manager.last_episode_mean_reward("foot_height_reward", before_weight=True) > 0.8:
            self.gait_command_manager.increment_foot_clearance_range()


def _gait_reset_with_curriculum(self, envs_idx=None):
    out = ManagedEnvironment.reset(self, envs_idx)
    if envs_idx is not None and self._curriculum:
        self.update_curriculum()
    return out


class Go2GaitTrainingCurriculumEnv(Go2GaitTrainingEnv):
    """The example's ``reset`` override (examples/gait_trainer/environment.py:347-352).  Kept in a subclass: an env that
    overrides ``reset`` gets the reference's index-list reset path: its step is recorded up to the reset, the reset and the
    observations run phase by phase (_trace.StepTrace.tail_python)."""

    reset = _gait_reset_with_curriculum


def make_example(name: str, case: dict):
    """This package's restatement of the reference example ``name`` with the case's scene options (tests/example_cases.py)."""
    kw = dict(num_envs=case["n"], max_episode_length_s=case["episode_s"], scene_kwargs=dict(case["scene"]))
    if name == "simple":
        return Go2SimpleEnv(**kw)
    if name == "command_direction":
        kw["scene_kwargs"].setdefault("max_collision_pairs", 30)
        return Go2CommandDirectionEnv(**kw)
    if name == "contacts":
        return Go2ContactsEnv(**kw)
    if name == "rough_terrain":
        return Go2RoughTerrainEnv(height_reward=False, **kw)
    if name == "berkeley_humanoid":
        return BerkeleyHumanoidEnv(**kw)
    if name == "gait_trainer":
        return Go2GaitTrainingCurriculumEnv(**kw)
    raise KeyError(name)


class Go2CommandDirectionEnv(ManagedEnvironment):
    """BASELINE config 2: Go2 12-DOF, 6 rewards, 2 terminations, velocity command, 7 observation items (O=48)."""

    def __init__(self, num_envs=1, dt=1 / 50, max_episode_length_s=20, scene_kwargs=None, obs_noise=False, fused_obs=True,
                 contacts=False, history=None, cmd_resample_s=5.0, dofs=12):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.1)
        kw = dict(scene_kwargs or {})
        if contacts:
            kw.setdefault("max_collision_pairs", 12)
        self._contacts, self._history, self._cmd_resample_s, self._dofs = contacts, history, cmd_resample_s, dofs
        self.scene = new_scene(dt=self.dt, substeps=2, **kw)
        self.terrain = self.scene.add_entity(morphs.Plane())
        if dofs == 12:
            self.robot = self.scene.add_entity(morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=INITIAL_BODY_POSITION, quat=INITIAL_QUAT))
        else:  # the same manager stack over a synthetic `dofs`-joint robot (BASELINE.md plan B2: D in {12, 28})
            from genesis_forge_amd.scene import humanoid_model
            self.robot = self.scene.add_entity(model=humanoid_model(dofs))
        self._obs_noise = obs_noise
        self._fused_obs = fused_obs

    def config(self):
        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={
            "position": {"fn": reset.position, "params": {"position": INITIAL_BODY_POSITION, "quat": INITIAL_QUAT, "zero_velocity": True}}})
        if self._dofs == 12:
            joint_names, default_pos = GO2_JOINTS, GO2_DEFAULT_POS
        else:
            joint_names, default_pos = [".*"], {".*": 0.1}
        self.action_manager = getattr(self, "action_cls", PositionActionManager)(   # (tests plug a user-defined action manager class in)
            self, joint_names=joint_names, default_pos=default_pos, scale=0.25, use_default_offset=True, pd_kp=20, pd_kv=0.5)
        self.velocity_command = VelocityCommandManager(
            self, range={"lin_vel_x": [-1.0, 1.0], "lin_vel_y": [-1.0, 1.0], "ang_vel_z": [-1.0, 1.0]}, standing_probability=0.02,
            resample_time_sec=self._cmd_resample_s)
        rcfg = {
            "base_height_target": {"weight": -50.0, "fn": rewards.base_height, "params": {"target_height": 0.3, "entity_attr": "robot"}},
            "tracking_lin_vel": {"weight": 1.0, "fn": rewards.command_tracking_lin_vel,
                                 "params": {"vel_cmd_manager": self.velocity_command, "entity_manager": self.robot_manager}},
            "tracking_ang_vel": {"weight": 0.5, "fn": rewards.command_tracking_ang_vel,
                                 "params": {"vel_cmd_manager": self.velocity_command, "entity_manager": self.robot_manager}},
            "lin_vel_z": {"weight": -1.0, "fn": rewards.lin_vel_z_l2, "params": {"entity_manager": self.robot_manager}},
            "action_rate": {"weight": -0.005, "fn": rewards.action_rate_l2},
            "similar_to_default": {"weight": -0.1, "fn": rewards.dof_similar_to_default, "params": {"action_manager": self.action_manager}},
        }
        tcfg = {
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "fall_over": {"fn": terminations.bad_orientation, "params": {"limit_angle": 10.0, "entity_manager": self.robot_manager}},
        }
        if self._contacts:
            self.foot_contacts = ContactManager(self, link_names=[".*_foot"], track_air_time=True, air_time_contact_threshold=5.0)
            self.body_contacts = ContactManager(self, link_names=[".*_thigh", "base"])
            rcfg["foot_air_time"] = {"weight": 2.5, "fn": rewards.feet_air_time,
                                     "params": {"contact_manager": self.foot_contacts, "time_threshold": 0.05,
                                                "vel_cmd_manager": self.velocity_command}}
            rcfg["undesired_contacts"] = {"weight": -1.0, "fn": rewards.has_contact,
                                          "params": {"contact_manager": self.body_contacts, "threshold": 5.0}}
            rcfg["ang_vel_xy"] = {"weight": -0.05, "fn": rewards.ang_vel_xy_l2, "params": {"entity_manager": self.robot_manager}}
            rcfg["flat_orientation"] = {"weight": -2.5, "fn": rewards.flat_orientation_l2, "params": {"entity_manager": self.robot_manager}}
            rcfg["terminated"] = {"weight": -100.0, "fn": rewards.terminated}
            rcfg["zero_weight"] = {"weight": 0.0, "fn": rewards.is_alive}
            tcfg["body_contact"] = {"fn": terminations.contact_force, "params": {"contact_manager": self.body_contacts, "threshold": 30.0}}
        self.reward_manager = RewardManager(self, logging_enabled=True, cfg=rcfg)
        self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg=tcfg)
        noise = 0.01 if self._obs_noise else None
        ocfg = {
            "velocity_cmd": {"fn": self.velocity_command.observation},
            "angle_velocity": {"fn": lambda env: self.robot_manager.get_angular_velocity(), "noise": noise},
            "linear_velocity": {"fn": lambda env: self.robot_manager.get_linear_velocity()},
            "projected_gravity": {"fn": lambda env: self.robot_manager.get_projected_gravity()},
            "dof_position": {"fn": lambda env: self.action_manager.get_dofs_position()},
            "dof_velocity": {"fn": lambda env: self.action_manager.get_dofs_velocity(), "scale": 0.05},
            "actions": {"fn": lambda env: self.action_manager.get_actions()},
        }
        if self._contacts:
            ocfg["foot_force"] = {"fn": observations.contact_force, "params": {"contact_manager": self.foot_contacts}, "scale": 0.1}
        self.observation_manager = ObservationManager(self, fused=self._fused_obs, cfg=ocfg, history_len=self._history)


class HumanoidGaitLikeEnv(ManagedEnvironment):
    """Stress config for the fused post-physics kernel: synthetic 28-DOF humanoid (BASELINE config 4's size), two
    command managers, two observation managers (policy with history, critic), three contact managers, the stateful
    body_acceleration_exp term, PositionWithinLimitsActionManager, zero-weight and contact terms."""

    def __init__(self, num_envs=1, dofs=28, dt=1 / 50, max_episode_length_s=1, scene_kwargs=None):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.4)
        from genesis_forge_amd.scene import humanoid_model
        kw = dict(max_collision_pairs=10, ang_noise=0.3, contact_prob=0.3, seed=21)
        kw.update(scene_kwargs or {})
        self.scene = new_scene(dt=self.dt, substeps=2, **kw)
        self.terrain = self.scene.add_entity(morphs.Plane())
        self.robot = self.scene.add_entity(model=humanoid_model(dofs))

    def config(self):
        from genesis_forge_amd.managers import CommandManager, PositionWithinLimitsActionManager
        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={
            "position": {"fn": reset.position, "params": {"position": [0.0, 0.0, 0.55], "quat": [1.0, 0.0, 0.0, 0.0]}}})
        self.action_manager = PositionWithinLimitsActionManager(self, joint_names=".*", default_pos={".*": 0.1}, noise_scale=0.02)
        self.velocity_command = VelocityCommandManager(self, range={"lin_vel_x": [-1.0, 1.0], "lin_vel_y": [0.0, 0.0], "ang_vel_z": [-1.0, 1.0]},
                                                       resample_time_sec=0.3)
        self.height_command = CommandManager(self, range=(0.3, 0.6), resample_time_sec=0.4)
        self.feet = ContactManager(self, link_names=[".*_faa"], track_air_time=True, air_time_contact_threshold=1.0)
        self.torso = ContactManager(self, link_names=["torso"])
        self.legs = ContactManager(self, link_names=[".*_kfe", ".*_hfe"], track_air_time=True)
        self.reward_manager = RewardManager(self, cfg={
            "lin": {"weight": 1.0, "fn": rewards.command_tracking_lin_vel, "params": {"vel_cmd_manager": self.velocity_command, "entity_manager": self.robot_manager}},
            "ang": {"weight": 0.5, "fn": rewards.command_tracking_ang_vel, "params": {"vel_cmd_manager": self.velocity_command, "entity_manager": self.robot_manager}},
            "accel": {"weight": -0.1, "fn": rewards.body_acceleration_exp, "params": {"entity_manager": self.robot_manager}},
            "rate": {"weight": -0.005, "fn": rewards.action_rate_l2},
            "pose": {"weight": -0.05, "fn": rewards.dof_similar_to_default, "params": {"action_manager": self.action_manager}},
            "still": {"weight": -0.2, "fn": rewards.stand_still_joint_deviation_l1, "params": {"vel_cmd_manager": self.velocity_command, "action_manager": self.action_manager, "command_threshold": 0.3}},
            "air": {"weight": 2.0, "fn": rewards.feet_air_time, "params": {"contact_manager": self.feet, "time_threshold": 0.04, "time_threshold_max": 0.5, "vel_cmd_manager": self.velocity_command}},
            "bad_contact": {"weight": -1.0, "fn": rewards.contact_force, "params": {"contact_manager": self.legs, "threshold": 2.0}},
            "slide": {"weight": -0.1, "fn": rewards.feet_slide, "params": {"contact_manager": self.feet}},
            "alive": {"weight": 0.3, "fn": rewards.is_alive},
            "off": {"weight": 0.0, "fn": rewards.flat_orientation_l2, "params": {"entity_manager": self.robot_manager}},
            "flat": {"weight": -1.0, "fn": rewards.flat_orientation_l2, "params": {"entity_manager": self.robot_manager}},
            "height": {"weight": -5.0, "fn": rewards.base_height, "params": {"target_height": 0.5}},
        })
        self.termination_manager = TerminationManager(self, term_cfg={
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "fall": {"fn": terminations.bad_orientation, "params": {"limit_angle": 20.0, "entity_manager": self.robot_manager, "grace_steps": 3}},
            "torso": {"fn": terminations.contact_force, "params": {"contact_manager": self.torso, "threshold": 25.0}},
            "low": {"fn": terminations.base_height_below_minimum, "params": {"minimum_height": 0.2, "entity_manager": self.robot_manager}},
        })
        self.observation_manager = ObservationManager(self, name="policy", history_len=3, noise=0.01, cfg={
            "height_cmd": {"fn": self.height_command.observation},
            "velocity_cmd": {"fn": self.velocity_command.observation},
            "ang": {"fn": lambda env: self.robot_manager.get_angular_velocity(), "scale": 0.25},
            "grav": {"fn": lambda env: self.robot_manager.get_projected_gravity()},
            "pos": {"fn": lambda env: self.action_manager.get_dofs_position()},
            "vel": {"fn": lambda env: self.action_manager.get_dofs_velocity(), "scale": 0.05},
            "act": {"fn": lambda env: self.action_manager.get_actions()},
        })
        self.critic_manager = ObservationManager(self, name="critic", cfg={
            "feet": {"fn": observations.contact_force, "params": {"contact_manager": self.feet}, "scale": 0.1},
            "lin": {"fn": lambda env: self.robot_manager.get_linear_velocity()},
            "force": {"fn": observations.entity_dofs_force, "params": {"action_manager": self.action_manager}},
            "raw": {"fn": observations.current_actions},
        })


class Go2RoughTerrainEnv(ManagedEnvironment):
    """BASELINE config 3 (cf. examples/rough_terrain/environment.py:87-287): Go2 on a height-field terrain, TerrainManager,
    random terrain spawn with random yaw on reset, out_of_bounds termination, two contact managers, 9 reward terms — plus
    (``height_reward=True``) ``base_height(terrain_manager=…)`` so the in-kernel height lookup is on the path."""

    def __init__(self, num_envs=1, dt=1 / 50, max_episode_length_s=20, scene_kwargs=None, height_reward=True, rotation="default",
                 terrain_kwargs=None, cmd_resample_s=5.0):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_s, max_episode_random_scaling=0.1)
        kw = dict(scene_kwargs or {})
        kw.setdefault("max_collision_pairs", 12)
        self._height_reward, self._rotation, self._cmd_resample_s = height_reward, rotation, cmd_resample_s
        self.scene = new_scene(dt=self.dt, substeps=2, **kw)
        tk = dict(pos=(-12, -12, 0), n_subterrains=(1, 1), subterrain_size=(24, 24), vertical_scale=0.001,
                  subterrain_types=[["random_uniform_terrain"]],
                  subterrain_parameters={"random_uniform_terrain": {"min_height": 0.0, "max_height": 0.1, "step": 0.05, "downsampled_scale": 0.25}})
        tk.update(terrain_kwargs or {})
        self.terrain = self.scene.add_entity(morph=morphs.Terrain(**tk))
        self.robot = self.scene.add_entity(morphs.URDF(file="urdf/go2/urdf/go2.urdf", pos=INITIAL_BODY_POSITION, quat=INITIAL_QUAT))

    def config(self):
        from genesis_forge_amd.managers import TerrainManager

        self.terrain_manager = TerrainManager(self)
        params = {"height_offset": 0.4, "terrain_manager": self.terrain_manager}
        if self._rotation != "default":
            params["rotation"] = self._rotation
        self.robot_manager = EntityManager(self, entity_attr="robot", on_reset={"position": {"fn": reset.randomize_terrain_position, "params": params}})
        self.action_manager = PositionActionManager(
            self, joint_names=["FL_.*_joint", "FR_.*_joint", "RL_.*_joint", "RR_.*_joint"],
            default_pos={".*_hip_joint": 0.0, "FL_thigh_joint": 0.8, "FR_thigh_joint": 0.8, "RL_thigh_joint": 1.0, "RR_thigh_joint": 1.0,
                         ".*_calf_joint": -1.5},
            scale=0.25, use_default_offset=True, pd_kp=20, pd_kv=0.5, max_force=23.5)
        self.velocity_command = VelocityCommandManager(
            self, range={"lin_vel_x": [-1.0, 1.0], "lin_vel_y": [-1.0, 1.0], "ang_vel_z": [-0.5, 0.5]}, standing_probability=0.05,
            resample_time_sec=self._cmd_resample_s)
        self.foot_contact_manager = ContactManager(self, link_names=[".*_calf"], track_air_time=True, air_time_contact_threshold=5.0)
        self.undesired_contacts = ContactManager(self, link_names=[".*_thigh", "base"])
        rcfg = {
            "tracking_lin_vel": {"weight": 1.5, "fn": rewards.command_tracking_lin_vel,
                                 "params": {"vel_cmd_manager": self.velocity_command, "entity_manager": self.robot_manager}},
            "tracking_ang_vel": {"weight": 0.75, "fn": rewards.command_tracking_ang_vel,
                                 "params": {"vel_cmd_manager": self.velocity_command, "entity_manager": self.robot_manager}},
            "lin_vel_z": {"weight": -2.0, "fn": rewards.lin_vel_z_l2, "params": {"entity_manager": self.robot_manager}},
            "ang_vel_xy": {"weight": -0.05, "fn": rewards.ang_vel_xy_l2, "params": {"entity_manager": self.robot_manager}},
            "undesired_contacts": {"weight": -1.0, "fn": rewards.has_contact, "params": {"contact_manager": self.undesired_contacts, "threshold": 5.0}},
            "action_rate": {"weight": -0.01, "fn": rewards.action_rate_l2},
            "similar_to_default": {"weight": -0.1, "fn": rewards.dof_similar_to_default, "params": {"action_manager": self.action_manager}},
            "flat_orientation": {"weight": -1.5, "fn": rewards.flat_orientation_l2},
            "terminated": {"weight": -100.0, "fn": rewards.terminated},
        }
        if self._height_reward:
            rcfg["base_height"] = {"weight": -30.0, "fn": rewards.base_height,
                                   "params": {"target_height": 0.35, "terrain_manager": self.terrain_manager, "entity_manager": self.robot_manager}}
        self.reward_manager = RewardManager(self, logging_enabled=True, cfg=rcfg)
        self.termination_manager = TerminationManager(self, logging_enabled=True, term_cfg={
            "timeout": {"fn": terminations.timeout, "time_out": True},
            "out_of_bounds": {"fn": terminations.out_of_bounds, "params": {"terrain_manager": self.terrain_manager}},
            "bad_orientation": {"fn": terminations.bad_orientation,
                                "params": {"limit_angle": 30.0, "entity_manager": self.robot_manager, "grace_steps": 20}},
        })
        self.observation_manager = ObservationManager(self, cfg={
            "velocity_cmd": {"fn": self.velocity_command.observation},
            "angle_velocity": {"fn": lambda env: self.robot_manager.get_angular_velocity()},
            "linear_velocity": {"fn": lambda env: self.robot_manager.get_linear_velocity()},
            "projected_gravity": {"fn": lambda env: self.robot_manager.get_projected_gravity()},
            "dof_position": {"fn": lambda env: self.action_manager.get_dofs_position()},
            "dof_velocity": {"fn": lambda env: self.action_manager.get_dofs_velocity(), "scale": 0.05},
            "actions": {"fn": lambda env: self.action_manager.get_actions()},
        })


# ----------------------------------------------------------------------------------------------------------------------------
# The workloads that get TIMED (bench.py: "go2_cmd"; tools/bench_configs.py: all of them).  tests/test_bench_parity.py walks the
# same table and compares every one of them with the oracle at the size it is timed at.
# Stand-in physics settings: small attitude noise (about 0.2 % of the envs fall over per step; SURVEY.md §8d's input spec would
# reset 2-8 % per step, which times the reset path rather than the step) and, for the configs with ContactManagers, the WALKING
# contact model of the synthetic scene (GfSynthSceneArgs.foot_link_mask): the links the config's foot ContactManager tracks touch
# the ground in a trot pattern — two to four feet of a quadruped (one or two of a biped) per tick, the ground on either side of the
# contact pair, so the Taichi kernel's matching branches (managers/contact/kernel.py:47-78), the air-time bookkeeping and
# feet_air_time (mdp/rewards.py:431-469) do real work every step — while body contacts (base / torso / thigh / calf) stay rare
# enough that the configs that terminate on them reset 0.3-0.5 % of the envs per step (episodes of a few hundred steps).
# ----------------------------------------------------------------------------------------------------------------------------
_SC = dict(ang_noise=0.05, seed=1234)
_CON = dict(_SC, contact_prob=0.15, contact_force=40.0)


def _walk(foot_links, body_prob, **kw):
    if os.environ.get("GF_SPARSE_CONTACTS") == "1":   # round 3's near-empty contact tables (0.03-0.06 contacts per env and step), for comparison only
        return dict(_SC, contact_force=40.0, contact_prob=body_prob, **kw)
    return dict(_SC, contact_force=40.0, foot_links=foot_links, foot_contact_prob=0.5, contact_prob=body_prob, **kw)


def bench_env(num_envs: int, **kw):
    """bench.py's workload: BASELINE.json's Go2 12-DOF config with the full Reward / Termination / Command manager stack."""
    return Go2CommandDirectionEnv(num_envs=num_envs, max_episode_length_s=kw.pop("max_episode_length_s", 20), scene_kwargs=dict(_SC), **kw)


#: name -> (BASELINE.json size, factory(num_envs, **env_kwargs))
BASELINE_CONFIGS = {
    "simple": (4096, lambda n, **kw: Go2SimpleEnv(num_envs=n, scene_kwargs=dict(_SC), **kw)),
    "go2_cmd": (4096, lambda n, **kw: bench_env(n, **kw)),
    "go2_cmd_65536": (65536, lambda n, **kw: bench_env(n, **kw)),
    "contacts": (4096, lambda n, **kw: Go2ContactsEnv(num_envs=n, scene_kwargs=_walk(".*_calf", 0.01), **kw)),
    "rough_terrain": (16384, lambda n, **kw: Go2RoughTerrainEnv(num_envs=n, height_reward=False,
                                                               scene_kwargs=_walk(".*_calf", 0.01, max_collision_pairs=30), **kw)),
    "humanoid": (8192, lambda n, **kw: BerkeleyHumanoidEnv(num_envs=n, scene_kwargs=_walk(".*_faa", 0.002, max_collision_pairs=30), **kw)),
    # BASELINE config 4 as stated ("~28-DOF"): the same kind of manager stack over a synthetic 28-joint humanoid (the reference's
    # berkeley_humanoid example, "humanoid" above, has 12 actuated joints)
    "humanoid28": (8192, lambda n, **kw: HumanoidGaitLikeEnv(num_envs=n, dofs=28, **kw)),
    "gait": (65536, lambda n, **kw: Go2GaitTrainingEnv(num_envs=n, scene_kwargs=_walk(".*_foot", 0.001), **kw)),
    # a USER's config: the Go2 task with observation noise — a structure none of the library's built-in programs has, so its fused
    # launch runs the program compiled for it at run time (genesis_forge_amd/_programs.py; GF_JIT) instead of the interpreter
    "go2_user": (65536, lambda n, **kw: bench_env(n, obs_noise=True, **kw)),
    "gait_8192": (8192, lambda n, **kw: Go2GaitTrainingEnv(num_envs=n, scene_kwargs=_walk(".*_foot", 0.001), **kw)),
    # the gait example AS SHIPPED: with its reset() override (curriculum hook, examples/gait_trainer/environment.py:347-352) — the reset
    # runs through user code by index list; what is recorded around it: GF_POST_NO_RESET in front, GF_POST_OBSERVE_ONLY behind
    "gait_override_8192": (8192, lambda n, **kw: Go2GaitTrainingCurriculumEnv(num_envs=n, scene_kwargs=_walk(".*_foot", 0.001), **kw)),
}
