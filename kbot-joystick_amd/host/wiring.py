"""Read-only views of the task wiring the reference builds in its `get_*` methods (train.py:1059-1276).

In the reference these methods return ksim objects that ksim's runtime calls; here the wiring is COMPILED into `kbj_config`
(spec/layout.py) and executed by the HIP kernels. The views below describe that compiled wiring with the reference's names and
parameters, so a user of the reference finds every knob where they expect it. Editing = passing the corresponding
`HumanoidWalkingTaskConfig` field (e.g. `reward_scales`, `reward_params`, `command_ranges`) and rebuilding the task; the views
themselves are frozen. Semantics of the un-vendored ksim pieces are this build's own (DESIGN.md "Spec decisions").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

from ..spec import constants, layout as L


@dataclass(frozen=True)
class OptimizerSpec:            # train.py:1059-1077
    kind: str                   # "adamw" | "adam"
    learning_rate: float
    weight_decay: float
    b1: float
    b2: float
    eps: float
    max_grad_norm: float        # ksim's global-norm clip (restated default)
    schedule: Optional[str]     # None | "cosine_decay"
    decay_steps: Optional[int] = None
    alpha: Optional[float] = None


@dataclass(frozen=True)
class PositionActuatorsSpec:    # train.py:1091-1105
    kp: Tuple[float, ...]
    kd: Tuple[float, ...]
    soft_torque_limit: Tuple[float, ...]
    kp_scale: float
    kd_scale: float
    torque_limit_scale_low: float
    action_bias_scale: float
    torque_bias_scale: float
    action_latency_range: Tuple[float, float]   # train.py:1780
    drop_action_prob: float                     # train.py:1781


@dataclass(frozen=True)
class RandomizerSpec:           # train.py:1107-1132
    name: str
    params: Dict[str, float]
    note: str = ""


@dataclass(frozen=True)
class EventSpec:                # train.py:1134-1144
    name: str
    body_name: str
    max_force: float
    max_torque: float
    duration_range: Tuple[float, float]
    interval_range: Tuple[float, float]
    enabled: bool


@dataclass(frozen=True)
class ResetSpec:                # train.py:1146-1153
    name: str
    params: Dict[str, float]


@dataclass(frozen=True)
class ObservationSpec:          # train.py:1155-1204
    name: str
    dim: int
    noise: str = ""
    used_by: str = ""           # which packed vector / reward consumes it


@dataclass(frozen=True)
class CommandSpec:              # train.py:1206-1222, 710-785
    vx_range: Tuple[float, float]
    vy_range: Tuple[float, float]
    wz_range: Tuple[float, float]
    bh_range: Tuple[float, float]
    rx_range: Tuple[float, float]
    ry_range: Tuple[float, float]
    arms_range: Tuple[Tuple[float, ...], Tuple[float, ...]]
    ctrl_dt: float
    switch_prob: float
    fixed_command: Optional[Tuple[float, ...]]   # BASELINE configs[1]; None = the 6-mode sampler


def _example_rewards():
    """`examples/reference_rewards.py` as a module (repository checkout layout: `<root>/examples`, `<root>/kbot-joystick_amd`)."""
    import importlib.util, os, sys
    name = "kbj_examples_reference_rewards"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "reference_rewards.py")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: RewardSpec.build() needs the repository's examples/ directory (the reward classes are example code, not part of the package)")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@dataclass(frozen=True)
class RewardSpec:               # train.py:1224-1256
    name: str
    scale: float
    params: Dict[str, float] = field(default_factory=dict)

    def build(self, model):
        """This entry as an executable reward term in ksim's protocol (`scale`, `get_reward(trajectory)` / `initial_carry`,
        `get_reward_stateful`): the class the reference constructs for this key (train.py:1224-1256), restated in torch on
        `host/trajectory.Trajectory`, with THIS configuration's scale and constructor arguments. The kernel (`rewards_kernel`) remains
        what the task runs; the built term is what a user edits and passes back as `extra_rewards` (with the built-in scale set to 0).
        The classes are example material (`examples/reference_rewards.py` at the repository root), loaded from there on demand: the product
        package holds no restatement of the reference's reward code."""
        return _example_rewards().build_reward(self.name, self.scale, self.params, model)


@dataclass(frozen=True)
class TerminationSpec:          # train.py:1258-1269
    name: str
    params: Dict[str, float]


@dataclass(frozen=True)
class CurriculumSpec:           # train.py:1271-1276 ("disable curriculum": the level is pinned at 1.0)
    kind: str = "LinearCurriculum"
    step_size: float = 1
    step_every_n_epochs: int = 1
    min_level: float = 1.0
    level: float = 1.0


# (config field, reference keyword) of every reward constructor argument besides `scale`, KBJ_REW_* order
REWARD_PARAM_FIELDS = {
    "linvel": {"error_scale": "rew_linvel_err"},
    "angvel": {"error_scale": "rew_angvel_err"},
    "roll_pitch": {"error_scale": "rew_rollpitch_err", "error_scale_zero_cmd": "rew_rollpitch_err_zero"},
    "base_height": {"error_scale": "rew_height_err", "standard_height": "rew_standard_height", "foot_origin_height": "rew_foot_origin_height"},
    "arm_pos": {"error_scale": "rew_armpos_err"},
    "single_contact": {"grace_period": "rew_grace_period"},
    "no_contact_p": {},
    "feet_airtime": {"touchdown_penalty": "rew_touchdown_penalty"},
    "feet_orient": {"error_scale": "rew_feetorient_err"},
    "com_distance": {"error_scale": "rew_comdist_err"},
    "base_accel": {"error_scale": "rew_baseaccel_err"},
    "torque": {"error_scale": "rew_torque_err"},
}


def apply_reward_overrides(kcfg: L.Config, scales: Optional[dict], params: Optional[dict]) -> None:
    """HumanoidWalkingTaskConfig.reward_scales / reward_params -> kbj_config (the editable part of get_rewards())."""
    for name, v in (scales or {}).items():
        if name not in constants.REWARD_NAMES:
            raise KeyError(f"unknown reward {name!r}; known: {constants.REWARD_NAMES}")
        kcfg.reward_scale[constants.REWARD_NAMES.index(name)] = float(v)
    for name, kv in (params or {}).items():
        if name not in REWARD_PARAM_FIELDS:
            raise KeyError(f"unknown reward {name!r}; known: {constants.REWARD_NAMES}")
        for k, v in kv.items():
            if k not in REWARD_PARAM_FIELDS[name]:
                raise KeyError(f"reward {name!r} has no parameter {k!r}; it has {sorted(REWARD_PARAM_FIELDS[name])}")
            v = float(v)
            # error scales sit in exp(-x / scale): zero, negative or non-finite values make the reward inf / NaN, GAE passes that into the
            # gradient and the optimizer's non-finite guard then skips every step - refuse here, with the name
            if ("error_scale" in k or k.endswith("_scale")) and not (v > 0.0 and v < float("inf")):
                raise ValueError(f"reward {name!r}: {k} must be a positive finite number, got {v!r}")
            if v != v or v in (float("inf"), float("-inf")):
                raise ValueError(f"reward {name!r}: {k} must be finite, got {v!r}")
            setattr(kcfg, REWARD_PARAM_FIELDS[name][k], v)


def rewards(kcfg: L.Config) -> Dict[str, RewardSpec]:
    out = {}
    for i, name in enumerate(constants.REWARD_NAMES):
        p = {k: float(getattr(kcfg, f)) for k, f in REWARD_PARAM_FIELDS[name].items()}
        if name in ("single_contact", "feet_airtime"):
            p["ctrl_dt"] = float(kcfg.ctrl_dt)
        out[name] = RewardSpec(name, float(kcfg.reward_scale[i]), p)
    return out


def optimizer(cfg, kcfg: L.Config) -> OptimizerSpec:
    kind = "adam" if cfg.adam_weight_decay == 0.0 else "adamw"
    if cfg.use_lr_decay and cfg.adam_weight_decay == 0.0:
        kind = "scale_by_adam+scale_by_schedule"      # train.py:1074-1075 as written: no sign flip (host/task.py update())
    if cfg.use_lr_decay:
        return OptimizerSpec(kind, cfg.learning_rate, cfg.adam_weight_decay, kcfg.adam_b1, kcfg.adam_b2, kcfg.adam_eps, kcfg.max_grad_norm,
                             "cosine_decay", cfg.lr_decay_steps, cfg.lr_final_multiplier)
    return OptimizerSpec(kind, cfg.learning_rate, cfg.adam_weight_decay, kcfg.adam_b1, kcfg.adam_b2, kcfg.adam_eps, kcfg.max_grad_norm, None)


def actuators(model: L.Model, kcfg: L.Config) -> PositionActuatorsSpec:
    return PositionActuatorsSpec(tuple(model.kp), tuple(model.kd), tuple(model.tau_limit), kcfg.kp_scale, kcfg.kd_scale, kcfg.torque_limit_scale_low,
                                 kcfg.action_bias_scale, kcfg.torque_bias_scale, (kcfg.latency_lo, kcfg.latency_hi), kcfg.drop_action_prob)


def physics_randomizers(kcfg: L.Config) -> Dict[str, RandomizerSpec]:
    on = bool(kcfg.enable_randomizers)
    off = "" if on else " (enable_randomizers = 0: off)"
    return {
        "static_friction": RandomizerSpec("StaticFrictionRandomizer", dict(scale_lower=kcfg.fricloss_scale_lo, scale_upper=kcfg.fricloss_scale_hi),
                                          "dof_frictionloss x U[lo, hi]; ksim default range restated" + off),
        "armature": RandomizerSpec("ArmatureRandomizer", dict(scale_lower=kcfg.armature_scale_lo, scale_upper=kcfg.armature_scale_hi),
                                   "dof_armature x U[lo, hi]; ksim default range restated" + off),
        "joint_damping": RandomizerSpec("JointDampingRandomizer", dict(scale_lower=0.5, scale_upper=2.5),
                                        "no-op: the MJCF defines no joint damping (robot.mjcf:4-19)"),
        "floor_friction": RandomizerSpec("FloorFrictionRandomizer", dict(scale_lower=kcfg.floor_friction_lo, scale_upper=kcfg.floor_friction_hi),
                                         "no-op: capsule priority 1 beats the floor, contact friction is the capsule's (robot.mjcf:24)"),
        "all_body_COM": RandomizerSpec("AllBodiesCOMRandomizer", dict(scale=kcfg.com_jitter), "body_ipos + U(+-scale)" + off),
        "all_body_inertia": RandomizerSpec("AllBodiesInertiaRandomizer", dict(scale=kcfg.inertia_scale), "mass and inertia x U[1-s, 1+s]" + off),
        "collision_body": RandomizerSpec("CollisionBodyRandomizer", dict(radius_scale=kcfg.cap_radius_scale, length_scale=kcfg.cap_length_scale,
                                                                         position_jitter_x=kcfg.cap_jitter[0], position_jitter_y=kcfg.cap_jitter[1],
                                                                         position_jitter_z=kcfg.cap_jitter[2]),
                                         "the four foot capsules " + ", ".join(constants.COLLISION_CAPSULES) + off),
    }


def events(kcfg: L.Config) -> Dict[str, EventSpec]:
    return {"force_push": EventSpec("ForcePushEvent", constants.BASE_BODY, kcfg.push_max_force, kcfg.push_max_torque, (kcfg.push_dur_lo, kcfg.push_dur_hi),
                                    (kcfg.push_int_lo, kcfg.push_int_hi), bool(kcfg.enable_pushes))}


def resets(kcfg: L.Config):
    return [ResetSpec("RandomJointPositionReset", dict(scale=kcfg.reset_joint_pos_scale)), ResetSpec("RandomJointVelocityReset", dict(scale=kcfg.reset_joint_vel_scale)),
            ResetSpec("RandomBaseVelocityXYReset", dict(scale=kcfg.reset_base_vel_xy_scale)), ResetSpec("RandomHeadingReset", {}),
            ResetSpec("PlaneXYPositionReset", dict(x_range=kcfg.reset_xy_range, y_range=kcfg.reset_xy_range))]


def observations(kcfg: L.Config) -> Dict[str, ObservationSpec]:
    n = bool(kcfg.enable_noise)
    deg = math.degrees
    u = lambda mag: f"uniform +-{deg(mag):.3g} deg" if n else ""
    g = lambda std: f"gaussian std {deg(std):.3g} deg" if n else ""
    O = ObservationSpec
    return {
        "joint_position": O("joint_position", 20, "", "critic[0:20]"),
        "biased_joint_position": O("biased_joint_position", 20, (f"episode bias +-{deg(kcfg.jpos_bias_range):.3g} deg; " + u(kcfg.jpos_noise)) if n else "", "actor[0:20]"),
        "joint_velocity": O("joint_velocity", 20, u(kcfg.jvel_noise) + ("/s" if n else ""), "actor[20:40] (noisy), critic[20:40]"),
        "actuator_force": O("actuator_force", 20, "", "critic[454:474] (/4)"),
        "center_of_mass_inertia": O("center_of_mass_inertia", 230, "", "critic[80:310]"),
        "center_of_mass_velocity": O("center_of_mass_velocity", 138, "", "critic[310:448]"),
        "base_position": O("base_position", 3, "", "critic[73:76]"),
        "base_orientation": O("base_orientation", 4, "", "critic[76:80]"),
        "base_linear_velocity": O("base_linear_velocity", 3, "", "critic[448:451]"),
        "base_angular_velocity": O("base_angular_velocity", 3, "", "critic[451:454]"),
        "base_linear_acceleration": O("base_linear_acceleration", 3, "", "unused by actor, critic and rewards (train.py:1381-1433): not computed"),
        "base_angular_acceleration": O("base_angular_acceleration", 3, "", "unused: not computed"),
        "actuator_acceleration": O("actuator_acceleration", 20, "", "unused: not computed"),
        "imu_gyro": O("imu_gyro", 3, g(kcfg.gyro_noise_std) + ("/s" if n else ""), "actor[45:48] (noisy), critic[45:48]"),
        "left_foot_touch": O("left_foot_touch", 1, "", "critic[65], rewards"),
        "right_foot_touch": O("right_foot_touch", 1, "", "critic[66], rewards"),
        "feet_position": O("feet_position", 6, "", "critic[67:73]"),
        "base_height": O("base_height", 1, "", "critic[474]"),
        "imu_projected_gravity": O("imu_projected_gravity", 3, (g(kcfg.pg_noise_std) + f", lag U[{kcfg.pg_lag_lo:.3g}, {kcfg.pg_lag_hi:.3g}], bias +-{deg(kcfg.pg_bias):.3g} deg") if n else "",
                                   "actor[40:45] (roll, pitch, unit vector)"),
        "projected_gravity": O("projected_gravity", 3, "", "critic[40:45]"),
        "com_distance": O("com_distance", 1, "", "com_distance reward"),
    }


def commands(model: L.Model, kcfg: L.Config) -> Dict[str, CommandSpec]:
    lo = tuple(model.dof_range[16 + j][0] for j in range(10))
    hi = tuple(model.dof_range[16 + j][1] for j in range(10))
    fixed = tuple(kcfg.fixed_command) if kcfg.command_mode == 1 else None
    return {"unified_command": CommandSpec((kcfg.vx_lo, kcfg.vx_hi), (kcfg.vy_lo, kcfg.vy_hi), (kcfg.wz_lo, kcfg.wz_hi), (kcfg.bh_lo, kcfg.bh_hi),
                                           (kcfg.rx_lo, kcfg.rx_hi), (kcfg.ry_lo, kcfg.ry_hi), (lo, hi), kcfg.ctrl_dt, kcfg.switch_prob, fixed)}


def terminations(kcfg: L.Config) -> Dict[str, TerminationSpec]:
    return {"bad_z": TerminationSpec("TerrainBadZTermination", dict(unhealthy_z=kcfg.unhealthy_z)),
            "not_upright": TerminationSpec("NotUprightTermination", dict(max_radians=kcfg.max_tilt_rad)),
            "episode_length": TerminationSpec("EpisodeLengthTermination", dict(max_length_sec=kcfg.max_episode_steps * kcfg.ctrl_dt))}
