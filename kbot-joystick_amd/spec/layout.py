"""ctypes mirrors of include/kbj_model.h (kbj_model, kbj_config) and the record offsets.

The C header is the source of truth; tests/test_abi.py checks sizeof() of both structs against the
compiled library (kbj_sizeof_model / kbj_sizeof_config) so the two cannot drift silently.
"""
from __future__ import annotations

import ctypes as C
import math

NBODY, NQ, NV, NU, NCAP, NCON, NCMD, NREW = 24, 27, 26, 20, 4, 8, 16, 12
NOBS_ACTOR, NOBS_CRITIC, LD_ACTOR, LD_CRITIC = 65, 475, 68, 476     # the reference's rows (train.py:1281-1312) and their 16-byte aligned strides
MAX_EXTRA_OBS = 64                                                  # kbj_model.h KBJ_MAX_EXTRA_OBS: user columns behind the reference's


def ld_of(nobs: int) -> int:
    """KBJ_LD_OF: row stride (floats) of an observation row of `nobs` used columns."""
    return (nobs + 3) & ~3


def obs_widths(cfg) -> tuple[int, int, int, int]:
    """(nobs_actor, nobs_critic, ld_actor, ld_critic) of a kbj_config: the reference's 65 / 475 columns plus the user columns
    (extra_obs_actor / extra_obs_critic) appended behind them."""
    na, nc = NOBS_ACTOR + int(cfg.extra_obs_actor), NOBS_CRITIC + int(cfg.extra_obs_critic)
    return na, nc, ld_of(na), ld_of(nc)

MAGIC, VERSION = 0x4D4A424B, 3

f32, i32, u32 = C.c_float, C.c_int32, C.c_uint32


class Model(C.Structure):
    _fields_ = [
        ("magic", u32), ("version", u32),
        ("nbody", i32), ("nq", i32), ("nv", i32), ("nu", i32), ("ncap", i32), ("reserved0", i32),
        ("body_parent", i32 * NBODY), ("body_dofadr", i32 * NBODY), ("body_dofnum", i32 * NBODY),
        ("dof_body", i32 * NV), ("dof_parent", i32 * NV), ("cap_body", i32 * NCAP),
        ("base_body", i32), ("torso_body", i32), ("lfoot_body", i32), ("rfoot_body", i32), ("imu_body", i32),
        ("reserved1", i32 * 3),
        ("body_pos", (f32 * 3) * NBODY), ("body_quat", (f32 * 4) * NBODY), ("body_ipos", (f32 * 3) * NBODY),
        ("body_mass", f32 * NBODY), ("body_inertia", (f32 * 3) * NBODY), ("jnt_axis", (f32 * 3) * NBODY),
        ("qpos0", f32 * NQ), ("reserved2", f32),
        ("dof_armature", f32 * NV), ("dof_frictionloss", f32 * NV), ("dof_invweight0", f32 * NV),
        ("dof_range", (f32 * 2) * NV), ("act_range", (f32 * 2) * NU), ("body_invweight0", (f32 * 2) * NBODY),
        ("cap_pos", (f32 * 3) * NCAP), ("cap_axis", (f32 * 3) * NCAP), ("cap_halflen", f32 * NCAP),
        ("cap_radius", f32 * NCAP),
        ("site_pos", (f32 * 3) * 2), ("site_size", (f32 * 3) * 2), ("imu_quat", f32 * 4),
        ("contact_mu", f32), ("contact_solref", f32 * 2), ("contact_solimp", f32 * 5),
        ("limit_solref", f32 * 2), ("limit_solimp", f32 * 5), ("fric_solref", f32 * 2), ("fric_solimp", f32 * 5),
        ("gravity", f32 * 3),
        ("kp", f32 * NU), ("kd", f32 * NU), ("tau_limit", f32 * NU),
        ("joint_bias", f32 * NU), ("joint_lo", f32 * NU), ("joint_hi", f32 * NU),
        ("total_mass", f32), ("meaninertia", f32), ("reserved3", f32 * 2),
    ]


class Config(C.Structure):
    _fields_ = [
        ("num_envs", i32), ("env_id_offset", i32), ("rollout_len", i32), ("substeps", i32),
        ("solver_iterations", i32), ("ls_iterations", i32), ("hidden_size", i32), ("depth", i32),
        ("batch_size", i32), ("num_passes", i32), ("command_mode", i32), ("enable_randomizers", i32),
        ("enable_pushes", i32), ("enable_noise", i32), ("max_episode_steps", i32), ("solver_newton", i32), ("deterministic", i32), ("extra_obs_actor", i32), ("extra_obs_critic", i32), ("gemm_bf16x3", i32),
        ("dt", f32), ("ctrl_dt", f32), ("solver_tolerance", f32), ("latency_lo", f32), ("latency_hi", f32),
        ("drop_action_prob", f32), ("fixed_command", f32 * NCMD),
        ("vx_lo", f32), ("vx_hi", f32), ("vy_lo", f32), ("vy_hi", f32), ("wz_lo", f32), ("wz_hi", f32),
        ("bh_lo", f32), ("bh_hi", f32), ("rx_lo", f32), ("rx_hi", f32), ("ry_lo", f32), ("ry_hi", f32),
        ("switch_prob", f32),
        ("reset_joint_pos_scale", f32), ("reset_joint_vel_scale", f32), ("reset_base_vel_xy_scale", f32),
        ("reset_xy_range", f32),
        ("unhealthy_z", f32), ("max_tilt_rad", f32),
        ("kp_scale", f32), ("kd_scale", f32), ("torque_limit_scale_low", f32), ("action_bias_scale", f32),
        ("torque_bias_scale", f32),
        ("fricloss_scale_lo", f32), ("fricloss_scale_hi", f32), ("armature_scale_lo", f32), ("armature_scale_hi", f32),
        ("floor_friction_lo", f32), ("floor_friction_hi", f32), ("com_jitter", f32), ("inertia_scale", f32),
        ("cap_radius_scale", f32), ("cap_length_scale", f32), ("cap_jitter", f32 * 3),
        ("push_max_force", f32), ("push_max_torque", f32), ("push_dur_lo", f32), ("push_dur_hi", f32),
        ("push_int_lo", f32), ("push_int_hi", f32),
        ("jpos_bias_range", f32), ("jpos_noise", f32), ("jvel_noise", f32), ("gyro_noise_std", f32),
        ("pg_noise_std", f32), ("pg_lag_lo", f32), ("pg_lag_hi", f32), ("pg_bias", f32),
        ("min_std", f32), ("max_std", f32), ("var_scale", f32), ("lpf_alpha", f32),
        ("gamma", f32), ("lam", f32), ("clip_param", f32), ("value_loss_coef", f32), ("entropy_coef", f32),
        ("log_ratio_clip", f32), ("max_grad_norm", f32),
        ("learning_rate", f32), ("adam_b1", f32), ("adam_b2", f32), ("adam_eps", f32), ("weight_decay", f32),
        ("adv_eps", f32), ("value_clip", f32), ("actor_mirror_loss_scale", f32), ("critic_mirror_loss_scale", f32),
        ("terrain_amp", f32), ("terrain_wavelength", f32), ("reserved_f", f32 * 4),
        ("reward_scale", f32 * NREW),
        ("rew_linvel_err", f32), ("rew_angvel_err", f32), ("rew_rollpitch_err", f32), ("rew_rollpitch_err_zero", f32),
        ("rew_height_err", f32), ("rew_standard_height", f32), ("rew_foot_origin_height", f32),
        ("rew_armpos_err", f32), ("rew_grace_period", f32), ("rew_touchdown_penalty", f32),
        ("rew_feetorient_err", f32), ("rew_comdist_err", f32), ("rew_baseaccel_err", f32), ("rew_torque_err", f32),
        ("reserved_r", f32 * 2),
    ]


# record offsets (enum values of kbj_model.h)
EP = dict(IPOS=0, MASS=72, INERTIA=96, ARMATURE=168, FRICLOSS=194, CAP_POS=220, CAP_HALF=232, CAP_RAD=236,
          KP=240, KD=260, TAULIM=280, ACTBIAS=300, JPBIAS=320, PGBIAS=340, PGLAG=343, LATENCY=344, MU=345, SIZE=352)
ES = dict(QPOS=0, QVEL=28, WARM=54, ACT_PREV=80, CMD=100, PUSH=116, PUSH_REM=122, PUSH_NXT=123, TIME=124,
          PGLAG=125, EPISODE=128, STEP=129, SIZE=136)
RC = dict(TSINGLE=0, AIRTIME=1, CONTACT=3, SIZE=8)
QSTATE = dict(QPOS=0, QVEL=27, QPOS_KIN=53, SIZE=80)      # kbj_model.h KBJ_QSTATE_*: optional per env-step state record
# pieces of a packed observation row in the reference's concatenation order (kbj_model.h KBJ_OBS_*; train.py:1367-1374, 1410-1430):
# name -> (offset, width); the first six are common to the actor and the critic row
OBS = dict(JPOS=(0, 20), JVEL=(20, 20), PG=(40, 5), GYRO=(45, 3), ZEROCMD=(48, 1), CMD=(49, 16), TOUCH=(65, 2), FEETPOS=(67, 6), BASEPOS=(73, 3),
           BASEQUAT=(76, 4), CINERT=(80, 230), CVEL=(310, 138), LINVEL=(448, 3), ANGVEL=(451, 3), ACTFRC=(454, 20), HEIGHT=(474, 1))
OBS_JVEL_DIV, OBS_ACTFRC_DIV = 10.0, 4.0
AUX = dict(QVEL=0, BQUAT=6, BASEZ=10, LFZ=11, RFZ=12, LFQUAT=13, RFQUAT=17, ARMQ=21, CTRL=31, TOUCH=51,
           COMDIST=53, CMD=54, DONE=70, SIZE=72)


def default_config(**overrides) -> Config:
    """Launch configuration of the reference (train.py:1759-1792) + task wiring constants.

    Values not visible in the reference tree (ksim/MuJoCo defaults) are this build's documented
    choices, see DESIGN.md "Spec decisions".
    """
    c = Config()
    c.num_envs, c.env_id_offset, c.rollout_len, c.substeps = 4096, 0, 100, 5     # train.py:1763,1766,1775-1776
    c.solver_iterations, c.ls_iterations = 8, 8                                   # train.py:1777-1778
    c.hidden_size, c.depth, c.batch_size, c.num_passes = 256, 2, 512, 3           # train.py:1773,82-85,1764-1765
    c.command_mode, c.enable_randomizers, c.enable_pushes, c.enable_noise = 0, 1, 1, 1
    c.solver_newton = 1
    c.max_episode_steps = 600                                                     # 12 s, train.py:1268
    c.dt, c.ctrl_dt, c.solver_tolerance = 0.004, 0.02, 1e-6   # fp32 gradient noise floor sits above MuJoCo's 1e-8
    c.latency_lo, c.latency_hi, c.drop_action_prob = 0.003, 0.01, 0.05            # train.py:1780-1781
    c.vx_lo, c.vx_hi, c.vy_lo, c.vy_hi, c.wz_lo, c.wz_hi = -0.5, 1.2, -0.5, 0.5, -1.0, 1.0   # train.py:1212-1214
    c.bh_lo, c.bh_hi, c.rx_lo, c.rx_hi, c.ry_lo, c.ry_hi = -0.25, 0.05, -0.25, 0.25, -0.25, 0.25
    c.switch_prob = 0.02 / 5                                                      # train.py:1220
    c.reset_joint_pos_scale, c.reset_joint_vel_scale = 0.1, 2.0                   # train.py:1148-1149
    c.reset_base_vel_xy_scale, c.reset_xy_range = 0.2, 0.1                        # train.py:1150,1152
    c.unhealthy_z, c.max_tilt_rad = 0.4, math.radians(45)                         # train.py:1265,1267
    c.kp_scale, c.kd_scale, c.torque_limit_scale_low = 1.4, 1.4, 0.5              # train.py:1100-1102
    c.action_bias_scale, c.torque_bias_scale = 0.02, 0.0                          # train.py:1103-1104
    c.fricloss_scale_lo, c.fricloss_scale_hi = 0.5, 2.0
    c.armature_scale_lo, c.armature_scale_hi = 1.0, 1.05
    c.floor_friction_lo, c.floor_friction_hi = 0.5, 1.5                           # train.py:1113
    c.com_jitter, c.inertia_scale = 0.05, 0.15                                    # train.py:1115-1116
    c.cap_radius_scale, c.cap_length_scale = 0.01, 0.03                           # train.py:1126-1127
    c.cap_jitter[0], c.cap_jitter[1], c.cap_jitter[2] = 0.020, 0.005, 0.005       # train.py:1128-1130
    c.push_max_force, c.push_max_torque = 200.0, 10.0                             # train.py:1139-1140
    c.push_dur_lo, c.push_dur_hi, c.push_int_lo, c.push_int_hi = 0.1, 0.5, 0.0, 6.0
    c.jpos_bias_range = c.jpos_noise = math.radians(3)                            # train.py:1159-1160
    c.jvel_noise, c.gyro_noise_std = math.radians(15), math.radians(10)           # train.py:1162,1176
    c.pg_noise_std, c.pg_lag_lo, c.pg_lag_hi, c.pg_bias = math.radians(3), 0.0, 0.75, math.radians(4)
    c.min_std, c.max_std, c.var_scale = 0.01, 1.0, 0.5                            # train.py:1320-1322
    fc = 10.0                                                                     # train.py:90-93
    c.lpf_alpha = c.ctrl_dt / (c.ctrl_dt + 1.0 / (2.0 * math.pi * fc))
    c.gamma, c.lam, c.entropy_coef = 0.94, 0.94, 0.004                            # train.py:1767-1770
    c.clip_param, c.value_loss_coef, c.log_ratio_clip, c.max_grad_norm = 0.2, 0.5, 10.0, 2.0
    c.learning_rate, c.weight_decay = 5e-4, 1e-5                                  # train.py:95-102
    c.adam_b1, c.adam_b2, c.adam_eps, c.adv_eps, c.value_clip = 0.9, 0.999, 1e-8, 1e-6, 0.2
    c.terrain_amp, c.terrain_wavelength = 0.0, 2.0      # flat ground; BASELINE configs[4] sets terrain_amp = 0.05
    # reward stack, KBJ_REW_* order (train.py:1225-1256)
    for i, v in enumerate((0.2, 0.1, 0.2, 0.2, 0.2, 0.1, 0.1, 1.5, 0.1, 0.05, 0.1, 0.1)):
        c.reward_scale[i] = v
    c.rew_linvel_err, c.rew_angvel_err, c.rew_rollpitch_err, c.rew_rollpitch_err_zero = 0.2, 0.2, 0.03, 0.01
    c.rew_height_err, c.rew_standard_height, c.rew_foot_origin_height = 0.02, 0.80, 0.06
    c.rew_armpos_err, c.rew_grace_period, c.rew_touchdown_penalty = 0.1, 2.0, 0.4
    c.rew_feetorient_err, c.rew_comdist_err, c.rew_baseaccel_err, c.rew_torque_err = 0.02, 0.04, 5.0, 5.0
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise AttributeError(f"kbj_config has no field {k!r}")
        cur = getattr(c, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(c, k, v)
    return c


def param_count(hidden: int, depth: int = 2, extra_obs: tuple[int, int] = (0, 0)) -> tuple[int, int]:
    """(actor, critic) parameter counts (train.py:878-903, 964-989); extra_obs = (actor, critic) user columns."""
    lstm = depth * (4 * hidden * 2 * hidden + 4 * hidden)
    actor = (NOBS_ACTOR + extra_obs[0]) * hidden + hidden + lstm + hidden * 2 * NU + 2 * NU
    critic = (NOBS_CRITIC + extra_obs[1]) * hidden + hidden + lstm + hidden + 1
    return actor, critic


def param_leaves(hidden_size: int, depth: int = 2, extra_obs: tuple[int, int] = (0, 0)):
    """(name, shape) of every leaf of the flat parameter vector, in equinox leaf order of Model(actor, critic)
    (include/kbj.h; train.py:847-1046; convert.py:44-46 takes `model.actor`). extra_obs = (actor, critic) user columns: they widen the
    input projections, as appending a term to run_actor / run_critic's concatenation does in the reference (train.py:1351-1433)."""
    H, leaves = hidden_size, []
    for net, nin, nout in (("actor", NOBS_ACTOR + extra_obs[0], 2 * NU), ("critic", NOBS_CRITIC + extra_obs[1], 1)):
        leaves.append((f"{net}.input_proj.weight", (H, nin)))
        leaves.append((f"{net}.input_proj.bias", (H,)))
        for l in range(depth):
            leaves.append((f"{net}.rnns.{l}.weight_ih", (4 * H, H)))
            leaves.append((f"{net}.rnns.{l}.weight_hh", (4 * H, H)))
            leaves.append((f"{net}.rnns.{l}.bias", (4 * H,)))
        leaves.append((f"{net}.output_proj.weight", (nout, H)))
        leaves.append((f"{net}.output_proj.bias", (nout,)))
    return leaves
