/* kbj_model.h — shared DATA FORMATS of the K-Bot joystick hot path (no algorithms).
 *
 * Everything here is a plain-C description of bytes that cross the C ABI in kbj.h:
 *   - kbj_model   : the compiled robot ("model blob"), replaces mujoco.MjModel for this path
 *                   (reference: train.py:1079-1089 get_mujoco_model / get_mujoco_model_metadata,
 *                    robot/<name>/robot.mjcf, robot/<name>/metadata.json)
 *   - kbj_config  : run-time configuration (reference: train.py:73-122 config dataclass,
 *                   train.py:1759-1792 launch values, train.py:1091-1276 task wiring constants)
 *   - per-env parameter / state records and the trajectory record layouts (float offsets)
 *
 * The kbot topology is fixed (24 bodies, 1 free + 20 hinge joints, 4 foot capsules); the HIP
 * kernels are specialised on it and kbj_create() rejects blobs with a different topology.
 */
#ifndef KBJ_MODEL_H
#define KBJ_MODEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBJ_MAGIC   0x4D4A424Bu /* "KBJM" little endian */
#define KBJ_VERSION 3u

#define KBJ_NBODY 24 /* incl. world (0) */
#define KBJ_NQ    27
#define KBJ_NV    26
#define KBJ_NU    20
#define KBJ_NCAP  4  /* collision capsules (2 per foot) */
#define KBJ_NCON  8  /* capsule-plane: one contact per capsule end */
#define KBJ_NCMD  16
#define KBJ_NREW  12

/* observation vector sizes (train.py:1281-1312) and their row strides in HBM (16-B aligned rows) */
#define KBJ_NOBS_ACTOR   65
#define KBJ_NOBS_CRITIC  475
#define KBJ_LD_ACTOR     68
#define KBJ_MAX_DEPTH    4   /* LSTM layers per net the library's workspaces are laid out for */
#define KBJ_LD_CRITIC    476
/* User observations INTO the network rows (kbj_config.extra_obs_actor / extra_obs_critic, the f3 row of SURVEY.md section 8): the reference's
 * user appends terms to the lists run_actor / run_critic concatenate (train.py:1351-1433); here up to KBJ_MAX_EXTRA_OBS floats are appended
 * BEHIND the reference's 65 / 475 (every KBJ_OBS_* offset stays put). Row widths and strides of a context:
 *   nobs = KBJ_NOBS_x + extra_obs_x,  ld = KBJ_LD_OF(nobs)  (rows stay 16-byte aligned; the pad columns are zero)
 * The env kernels write the reference's columns and zero [KBJ_NOBS_x, ld); the host writes its terms' outputs behind them. */
#define KBJ_MAX_EXTRA_OBS 64
#define KBJ_LD_OF(nobs)  (((nobs) + 3) & ~3)

/* Offsets of the pieces of a packed observation row, in the order run_actor / run_critic concatenate them (train.py:1367-1374,
 * 1410-1430); the first 65 entries are common to both rows, the critic's privileged pieces follow. Divisors as train.py:1336, 1427.
 * tests/test_ref_constants.py derives this table from the reference's source text (ast) and asserts it. */
enum {
  KBJ_OBS_JPOS     = 0,    /* [20] normalize_joint_pos(joint position) */
  KBJ_OBS_JVEL     = 20,   /* [20] joint velocity / KBJ_OBS_JVEL_DIV */
  KBJ_OBS_PG       = 40,   /* [5]  roll, pitch, unit projected gravity */
  KBJ_OBS_GYRO     = 45,   /* [3]  */
  KBJ_OBS_ZEROCMD  = 48,   /* [1]  |cmd[0:3]| < 1e-3 */
  KBJ_OBS_CMD      = 49,   /* [16] unified command */
  KBJ_OBS_TOUCH    = 65,   /* [2]  left, right foot touch (critic only from here) */
  KBJ_OBS_FEETPOS  = 67,   /* [6]  */
  KBJ_OBS_BASEPOS  = 73,   /* [3]  */
  KBJ_OBS_BASEQUAT = 76,   /* [4]  */
  KBJ_OBS_CINERT   = 80,   /* [23][10] */
  KBJ_OBS_CVEL     = 310,  /* [23][6]  */
  KBJ_OBS_LINVEL   = 448,  /* [3]  */
  KBJ_OBS_ANGVEL   = 451,  /* [3]  */
  KBJ_OBS_ACTFRC   = 454,  /* [20] actuator force / KBJ_OBS_ACTFRC_DIV */
  KBJ_OBS_HEIGHT   = 474   /* [1]  */
};
#define KBJ_OBS_JVEL_DIV   10.0f
#define KBJ_OBS_ACTFRC_DIV 4.0f

typedef struct kbj_model {
  uint32_t magic, version;
  int32_t  nbody, nq, nv, nu, ncap, reserved0;
  int32_t  body_parent[KBJ_NBODY];
  int32_t  body_dofadr[KBJ_NBODY]; /* first dof of the body's joint, -1 if none */
  int32_t  body_dofnum[KBJ_NBODY]; /* 0 (welded), 1 (hinge), 6 (free) */
  int32_t  dof_body[KBJ_NV];
  int32_t  dof_parent[KBJ_NV];     /* previous dof up the tree, -1 at the root */
  int32_t  cap_body[KBJ_NCAP];
  int32_t  base_body, torso_body, lfoot_body, rfoot_body, imu_body, reserved1[3];
  float    body_pos[KBJ_NBODY][3];
  float    body_quat[KBJ_NBODY][4];   /* w,x,y,z relative to parent */
  float    body_ipos[KBJ_NBODY][3];
  float    body_mass[KBJ_NBODY];
  float    body_inertia[KBJ_NBODY][3]; /* diagonal, inertial frame == body frame for this robot */
  float    jnt_axis[KBJ_NBODY][3];     /* hinge axis in the body frame (bodies with dofnum==1) */
  float    qpos0[KBJ_NQ];
  float    reserved2;
  float    dof_armature[KBJ_NV];
  float    dof_frictionloss[KBJ_NV];
  float    dof_invweight0[KBJ_NV];
  float    dof_range[KBJ_NV][2];       /* joint limits by dof (first 6 unused) */
  float    act_range[KBJ_NU][2];       /* motor ctrlrange == joint actuatorfrcrange */
  float    body_invweight0[KBJ_NBODY][2];
  float    cap_pos[KBJ_NCAP][3];       /* capsule centre in body frame */
  float    cap_axis[KBJ_NCAP][3];      /* unit axis in body frame */
  float    cap_halflen[KBJ_NCAP];
  float    cap_radius[KBJ_NCAP];
  float    site_pos[2][3];             /* foot touch boxes (left, right), body frame, axis aligned */
  float    site_size[2][3];
  float    imu_quat[4];                /* imu_site orientation in the imu body */
  float    contact_mu;
  float    contact_solref[2];
  float    contact_solimp[5];
  float    limit_solref[2];
  float    limit_solimp[5];
  float    fric_solref[2];
  float    fric_solimp[5];
  float    gravity[3];
  float    kp[KBJ_NU], kd[KBJ_NU], tau_limit[KBJ_NU]; /* metadata.json per-joint gains */
  float    joint_bias[KBJ_NU];  /* train.py:24-45  */
  float    joint_lo[KBJ_NU];    /* train.py:47-68  */
  float    joint_hi[KBJ_NU];
  float    total_mass;
  float    meaninertia;
  float    reserved3[2];
} kbj_model;

/* Run-time configuration. Defaults (kbj_config_default in the host layer) follow the launch block
 * train.py:1759-1792 and the task wiring train.py:1091-1276; ksim-internal defaults that are not
 * visible in the reference tree are our own documented choices (DESIGN.md "Spec decisions"). */
typedef struct kbj_config {
  int32_t num_envs;          /* envs on THIS gpu */
  int32_t env_id_offset;     /* global id of local env 0 (rank * num_envs) -> RNG streams independent of gpu count */
  int32_t rollout_len;       /* T control steps per rollout (100 = 2.0 s / 0.02 s) */
  int32_t substeps;          /* ctrl_dt / dt = 5 */
  int32_t solver_iterations; /* 8 */
  int32_t ls_iterations;     /* 8 */
  int32_t hidden_size;       /* 256 launch / 128 dataclass default */
  int32_t depth;             /* LSTM layers per net, 1..KBJ_MAX_DEPTH (train.py:82-85: 2) */
  int32_t batch_size;        /* envs per minibatch */
  int32_t num_passes;
  int32_t command_mode;      /* 0 = UnifiedCommand sampler (train.py:710-785); 1 = fixed command; 2 = the sampler (and PlaneXYPositionReset,
                                train.py:834-836) with jax.random's own key handling below the call key: split / uniform / bernoulli / randint as jax
                                0.6.0 derives them from a threefry key (csrc/kbj_env_core.h "jax.random key handling"; UNVERIFIED against a live JAX) */
  int32_t enable_randomizers;
  int32_t enable_pushes;
  int32_t enable_noise;
  int32_t max_episode_steps; /* 12 s / 0.02 s = 600 */
  int32_t solver_newton;     /* 1 = Newton direction (default), 0 = Polak-Ribiere CG */
  int32_t deterministic;     /* 1 = the PPO update reduces in a fixed order (split-K slabs, per-block partials) instead of with fp32 / fp64 atomics:
                                two updates from the same state give bit-identical parameters, as the reference's XLA program does; default 0 */
  int32_t extra_obs_actor;   /* floats the host appends to every actor / critic observation row (0..KBJ_MAX_EXTRA_OBS, default 0): the input */
  int32_t extra_obs_critic;  /* projections are [H][65 + extra] / [H][475 + extra], the parameter vector grows accordingly */
  int32_t gemm_bf16x3;       /* 1 = the PPO update's large backward GEMMs (input gradients, weight-gradient pairs) run on the bf16 matrix cores through an
                                EXACT three-way split of their fp32 operands (6 bf16 products per fp32 product, fp32 accumulation): measured more
                                accurate than the fp32-MFMA chain and faster. Default 0: the plain fp32-MFMA kernels, which is
                                what every headline number of this library is measured with. Ignored in deterministic mode. */
  float dt;                  /* 0.004 */
  float ctrl_dt;             /* 0.02  */
  float solver_tolerance;    /* 1e-8 */
  float latency_lo, latency_hi; /* seconds (0.003, 0.01) */
  float drop_action_prob;    /* 0.05 */
  float fixed_command[KBJ_NCMD];
  /* command ranges train.py:1211-1221 */
  float vx_lo, vx_hi, vy_lo, vy_hi, wz_lo, wz_hi, bh_lo, bh_hi, rx_lo, rx_hi, ry_lo, ry_hi;
  float switch_prob;         /* ctrl_dt / 5 */
  /* resets train.py:1146-1153 */
  float reset_joint_pos_scale, reset_joint_vel_scale, reset_base_vel_xy_scale, reset_xy_range;
  /* terminations train.py:1258-1269 */
  float unhealthy_z, max_tilt_rad;
  /* actuator randomisation train.py:1097-1105 */
  float kp_scale, kd_scale, torque_limit_scale_low, action_bias_scale, torque_bias_scale;
  /* physics randomisers train.py:1107-1132 */
  float fricloss_scale_lo, fricloss_scale_hi, armature_scale_lo, armature_scale_hi;
  float floor_friction_lo, floor_friction_hi, com_jitter, inertia_scale;
  float cap_radius_scale, cap_length_scale, cap_jitter[3];
  /* push event train.py:1134-1144 */
  float push_max_force, push_max_torque, push_dur_lo, push_dur_hi, push_int_lo, push_int_hi;
  /* observation noise train.py:1156-1204 */
  float jpos_bias_range, jpos_noise, jvel_noise, gyro_noise_std, pg_noise_std, pg_lag_lo, pg_lag_hi, pg_bias;
  /* actor head train.py:1320-1326 */
  float min_std, max_std, var_scale, lpf_alpha;
  /* ppo (ksim defaults restated) */
  float gamma, lam, clip_param, value_loss_coef, entropy_coef, log_ratio_clip, max_grad_norm;
  float learning_rate, adam_b1, adam_b2, adam_eps, weight_decay, adv_eps;
  float value_clip;          /* clipped value loss range */
  float actor_mirror_loss_scale;  /* train.py:115-122 (dataclass defaults 1.0 / 0.01; launch config 0.0 / 0.0, train.py:1771-1772) */
  float critic_mirror_loss_scale;
  /* terrain ("sine" scene, train.py:1081; the surface is this build's own definition):
   * z = terrain_amp * sin(2 pi x / terrain_wavelength) * sin(2 pi y / terrain_wavelength); terrain_amp = 0 is the plane z = 0 */
  float terrain_amp, terrain_wavelength;
  float reserved_f[4];
  /* reward stack (train.py:1224-1256): the scale of every term in KBJ_REW_* order, then the constructor arguments the reference
   * passes to its reward classes (error scales, heights, grace period, touchdown penalty). User-editable like the reference's
   * get_rewards(); a scale of 0 switches a term off. */
  float reward_scale[12];
  float rew_linvel_err, rew_angvel_err, rew_rollpitch_err, rew_rollpitch_err_zero;   /* train.py:1227-1229 */
  float rew_height_err, rew_standard_height, rew_foot_origin_height;                 /* train.py:1230-1239 */
  float rew_armpos_err, rew_grace_period, rew_touchdown_penalty;                     /* train.py:1240-1244 */
  float rew_feetorient_err, rew_comdist_err, rew_baseaccel_err, rew_torque_err;      /* train.py:1245-1255 */
  float reserved_r[2];
} kbj_config;

/* ---- per-env randomised model parameters ("EP" record, floats, one contiguous row per env) ---- */
enum {
  KBJ_EP_IPOS     = 0,    /* [24][3] */
  KBJ_EP_MASS     = 72,   /* [24]    */
  KBJ_EP_INERTIA  = 96,   /* [24][3] */
  KBJ_EP_ARMATURE = 168,  /* [26]    */
  KBJ_EP_FRICLOSS = 194,  /* [26]    */
  KBJ_EP_CAP_POS  = 220,  /* [4][3]  */
  KBJ_EP_CAP_HALF = 232,  /* [4]     */
  KBJ_EP_CAP_RAD  = 236,  /* [4]     */
  KBJ_EP_KP       = 240,  /* [20]    */
  KBJ_EP_KD       = 260,
  KBJ_EP_TAULIM   = 280,
  KBJ_EP_ACTBIAS  = 300,
  KBJ_EP_JPBIAS   = 320,  /* [20] biased joint position observation offset */
  KBJ_EP_PGBIAS   = 340,  /* [3]  projected-gravity bias */
  KBJ_EP_PGLAG    = 343,
  KBJ_EP_LATENCY  = 344,  /* action latency in substeps (integer valued) */
  KBJ_EP_MU       = 345,  /* contact friction */
  KBJ_EP_SIZE     = 352
};

/* ---- per-env persistent state ("ES" record, floats; uint32 fields are bit-cast) ---- */
enum {
  KBJ_ES_QPOS     = 0,    /* [27] */
  KBJ_ES_QVEL     = 28,   /* [26] */
  KBJ_ES_WARM     = 54,   /* [26] qacc warm start */
  KBJ_ES_ACT_PREV = 80,   /* [20] action applied during the previous control step */
  KBJ_ES_CMD      = 100,  /* [16] */
  KBJ_ES_PUSH     = 116,  /* [6] force(3) torque(3) world frame */
  KBJ_ES_PUSH_REM = 122,  /* substeps of push remaining */
  KBJ_ES_PUSH_NXT = 123,  /* substeps until the next push starts */
  KBJ_ES_TIME     = 124,  /* control steps since episode start */
  KBJ_ES_PGLAG    = 125,  /* [3] lagged projected gravity */
  KBJ_ES_EPISODE  = 128,  /* uint32 episode counter (RNG stream) */
  KBJ_ES_STEP     = 129,  /* uint32 global control-step counter  */
  KBJ_ES_SIZE     = 136
};

/* ---- reward carry per env (StatefulReward carries, train.py:135-136,175-178) ---- */
enum {
  KBJ_RC_TSINGLE  = 0,   /* time since single contact, initial 0 */
  KBJ_RC_AIRTIME  = 1,   /* [2] initial 0 */
  KBJ_RC_CONTACT  = 3,   /* [2] previous contact flags, initial 1 (True) */
  KBJ_RC_SIZE     = 8
};

/* ---- per env-step "aux" record: everything the reward stack reads (train.py:125-506) ---- */
enum {
  KBJ_AUX_QVEL    = 0,   /* [6] base qvel after the step */
  KBJ_AUX_BQUAT   = 6,   /* [4] xquat[base] */
  KBJ_AUX_BASEZ   = 10,
  KBJ_AUX_LFZ     = 11,
  KBJ_AUX_RFZ     = 12,
  KBJ_AUX_LFQUAT  = 13,  /* [4] */
  KBJ_AUX_RFQUAT  = 17,  /* [4] */
  KBJ_AUX_ARMQ    = 21,  /* [10] qpos of the arm joints */
  KBJ_AUX_CTRL    = 31,  /* [20] torque command of the last substep */
  KBJ_AUX_TOUCH   = 51,  /* [2] observation (pre-step) */
  KBJ_AUX_COMDIST = 53,  /* observation (pre-step) */
  KBJ_AUX_CMD     = 54,  /* [16] command the policy saw */
  KBJ_AUX_DONE    = 70,  /* -1 failure, 0 running, +1 episode-length truncation */
  KBJ_AUX_SIZE    = 72
};

/* ---- optional per env-step state record (kbj_traj.qstate_d / kbj_env_record_state): the generalised state a ksim Trajectory step holds
 * (`trajectory.qpos`, `.qvel`: train.py:262, 286, 301, 488) plus the positions the step's LAST forward pass ran on, from which the body poses
 * of that pass (`trajectory.xpos`, `.xquat`: train.py:276, 317, 378-383, 419-444 - as in mj_step, the derived quantities a step leaves behind
 * are those of its last substep's kinematics, one integration behind qpos) follow by forward kinematics over the model blob's tree ---- */
enum {
  KBJ_QSTATE_QPOS     = 0,    /* [27] qpos after the step (before any reset) */
  KBJ_QSTATE_QVEL     = 27,   /* [26] qvel after the step */
  KBJ_QSTATE_QPOS_KIN = 53,   /* [27] qpos the last substep's forward pass (xpos, xquat, sensors, contacts) was computed from */
  KBJ_QSTATE_SIZE     = 80
};

/* reward component order (train.py:1225-1256) */
enum {
  KBJ_REW_LINVEL = 0, KBJ_REW_ANGVEL, KBJ_REW_ROLL_PITCH, KBJ_REW_BASE_HEIGHT, KBJ_REW_ARM_POS,
  KBJ_REW_SINGLE_CONTACT, KBJ_REW_NO_CONTACT, KBJ_REW_FEET_AIRTIME, KBJ_REW_FEET_ORIENT,
  KBJ_REW_COM_DISTANCE, KBJ_REW_BASE_ACCEL, KBJ_REW_TORQUE
};

/* RNG stream ids (threefry2x32 key = (seed ^ stream * 0x9E3779B9, global env id)) */
enum {
  KBJ_RNG_RESET = 1, KBJ_RNG_RANDOMIZE = 2, KBJ_RNG_OBS_NOISE = 3, KBJ_RNG_COMMAND = 4,
  KBJ_RNG_ACTION = 5, KBJ_RNG_DROP = 6, KBJ_RNG_PUSH = 7, KBJ_RNG_INIT = 8, KBJ_RNG_SHUFFLE = 9
};

#ifdef __cplusplus
}
#endif
#endif /* KBJ_MODEL_H */
