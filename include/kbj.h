/* kbj.h — C ABI of the MI355X-native K-Bot joystick hot path (rollout + PPO update).
 *
 * The reference has no FFI: its boundary is ksim's Python Task API (a `ksim.PPOTask` subclass,
 * train.py:1058) whose engine, rollout loop and PPO update run as jitted JAX programs. This
 * library is what a ksim-style host would bind instead of those programs; every entry point
 * cites the reference interface it replaces. The Python host layer
 * (kbot-joystick_amd/host) binds it with ctypes and re-exposes the Task API names.
 *
 * Conventions
 *  - All functions return 0 on success, <0 on error; kbj_last_error() gives the message.
 *    No exceptions cross the ABI, no torch types appear in signatures.
 *  - One kbj_ctx per (process, GPU). Not thread-safe; distinct contexts may be driven from
 *    distinct host threads. Work is enqueued on the hipStream_t given to kbj_create and is
 *    asynchronous unless stated otherwise.
 *  - Pointers named *_d are DEVICE pointers owned by the caller (e.g. torch tensor data_ptr());
 *    the library never frees caller memory. Pointers named *_h are host pointers.
 *  - Trajectory arrays are time-major, row-major: [T][N][dim] with dims/strides from kbj_model.h.
 *  - Multi-GPU: the library only produces/consumes flat fp32 buffers; the RCCL communicator is
 *    owned by the host (torch.distributed).
 */
#ifndef KBJ_H
#define KBJ_H

#include <stddef.h>
#include <stdint.h>
#include "kbj_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kbj_ctx kbj_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
/* replaces: HumanoidWalkingTask.launch(config) set-up — get_mujoco_model / metadata / mjx.put_model
 * (train.py:1079-1089, 1760-1792). model_blob is a kbj_model produced by the spec compiler.
 * Model fields served (train.py:78-85): cfg->hidden_size in 1..512 (multiples of 64 run natively, above 256 on an untuned schedule; any other size runs zero padded to the
 * next one INSIDE the library - parameters, gradients, carries and trajectory start carries keep the caller's hidden_size layout and are
 * converted at every entry point, exact for this network), cfg->depth in 1..KBJ_MAX_DEPTH; anything else fails here. */
int kbj_create(kbj_ctx** out, const void* model_blob, size_t model_bytes, const kbj_config* cfg, int device, void* hip_stream);
int kbj_destroy(kbj_ctx* ctx);
const char* kbj_last_error(const kbj_ctx* ctx); /* ctx may be NULL: error of a failed kbj_create */
/* Host-only check of the sizes a configuration asks for (no device needed; kbj_create applies it first): 0 = served, -1 = refused with the
 * reason in `why`. Besides the range checks (hidden_size 1..512, depth 1..4, Newton solver) it refuses configurations whose largest
 * operand - the minibatch stash [rollout_len x batch_size][4 hidden_size | 476] or one control step's [num_envs][...] rows - reaches
 * 2 GiB: those arrays are fetched with 32-bit byte offsets. (The reference has no such limit: XLA addresses with 64 bits.) */
int kbj_check_config(const kbj_config* cfg, char* why, size_t why_bytes);
int kbj_sizeof_model(void);
int kbj_sizeof_config(void);
int kbj_sizeof_traj(void);    /* sizeof(kbj_traj), sizeof(kbj_carry): let a binding verify its struct mirrors */
int kbj_sizeof_carry(void);
int kbj_synchronize(kbj_ctx* ctx);

/* ---- environment (physics + task), SURVEY §8 rows a1-a3, a14-a22 -------------------------- */
/* replaces: ksim reset path — get_resets / get_physics_randomizers / initial commands and the first
 * get_observations call (train.py:1107-1132, 1146-1204, 724-766). Writes observation row 0. */
int kbj_env_reset_all(kbj_ctx* ctx, uint32_t seed, float* actor0_d, float* critic0_d, float* aux0_d);
/* replaces: one ksim engine control step = action latency/drop + ctrl_dt/dt x (PositionActuators.get_ctrl,
 * ForcePushEvent, mjx.step) + terminations + reset + command update + next observations
 * (train.py:1091-1105, 1134-1144, 1155-1222, 1258-1269, 1775-1781).
 * action_d [N][20]; aux_t_d [N][72] row of this step (completed); *_next_d rows of step t+1. */
int kbj_env_step(kbj_ctx* ctx, const float* action_d, float* aux_t_d, float* actor_next_d, float* critic_next_d, float* aux_next_d);
/* One-shot: the NEXT kbj_env_step of this context also writes the step's state record (KBJ_QSTATE_*: qpos, qvel after the step and the
 * positions its last forward pass ran on) into qstate_t_d [N][KBJ_QSTATE_SIZE]. kbj_rollout does the same per step for kbj_traj.qstate_d;
 * this entry serves hosts that drive the steps themselves (user Termination / Observation / Command terms). NULL cancels. */
int kbj_env_record_state(kbj_ctx* ctx, float* qstate_t_d);
/* replaces: ksim's reset of the envs a Termination finished, for terminations decided OUTSIDE the step kernel - user-written terms in
 * the reference's protocol (`Termination.__call__(physics_data, curriculum_level) -> {-1, 0, 1}`, train.py:817) evaluated by the host on
 * the post-step record. mask_d [N] float: envs with a non-zero entry are re-initialised exactly as kbj_env_step re-initialises an env its
 * own terminations finish (same reset / randomiser / command streams, episode counter + 1) and their rows of the NEXT observation arrays
 * are rewritten; the others are untouched. The caller then writes the term's value into the step's KBJ_AUX_DONE column before
 * kbj_carry_reset / kbj_rewards / kbj_gae read it. */
int kbj_env_reset_where(kbj_ctx* ctx, const float* mask_d, float* actor_next_d, float* critic_next_d, float* aux_next_d);
/* replaces: the command update of a user-written Command term in the reference's protocol (`Command.initial_command(physics_data,
 * curriculum_level, rng)` / `Command.__call__(prev_command, physics_data, curriculum_level, rng)`, train.py:724, 768) evaluated by the host
 * between two control steps. cmd_d [N][16]; mask_d [N] float or NULL (= every env): the envs with a non-zero entry get cmd_d's row as
 * their joystick command - in the env state (the next kbj_env_step starts from it; run with command_mode = 1 so the kernel's own
 * switch draw does not replace it) and in the command columns + zero-command flag of the NEXT observation rows and aux record. */
int kbj_env_set_command(kbj_ctx* ctx, const float* mask_d, const float* cmd_d, float* actor_next_d, float* critic_next_d, float* aux_next_d);
/* replaces: user-written Reset terms in the reference's protocol (`Reset.__call__(data, curriculum_level, rng) -> data`, train.py:833-844: the in-tree
 * PlaneXYPositionReset reads `data.qpos` and returns the data with a new one), evaluated by the host on the envs that have just been re-initialised
 * (by the step kernel's own terminations, kbj_env_reset_where or kbj_env_reset_all), AFTER the built-in resets of train.py:1146-1153 as in the
 * reference's list order. kbj_env_get_qstate copies every env's generalised positions / velocities into DEVICE arrays (qpos_d [N][27], qvel_d [N][26];
 * asynchronous, on the context's stream); kbj_env_set_qstate writes them back for the envs with a non-zero mask_d entry (NULL = all), clears their
 * solver warm start, re-runs the forward pass of the new state (kinematics, sensors, lagged projected gravity) and rewrites their NEXT observation
 * rows and aux record, exactly as the reset path does for the state it draws itself. Episode counter, randomised parameters and command are untouched. */
int kbj_env_get_qstate(kbj_ctx* ctx, float* qpos_d, float* qvel_d);
int kbj_env_set_qstate(kbj_ctx* ctx, const float* mask_d, const float* qpos_d, const float* qvel_d, float* actor_next_d, float* critic_next_d, float* aux_next_d);
/* state save/restore (checkpointing, tests): ep [N][KBJ_EP_SIZE], es [N][KBJ_ES_SIZE]; synchronous */
int kbj_env_get_state(kbj_ctx* ctx, float* ep_h, float* es_h);
int kbj_env_set_state(kbj_ctx* ctx, const float* ep_h, const float* es_h);
/* StatefulReward carries (train.py:135-136, 175-178: single-contact timer, feet airtime, previous contact flags) that kbj_rewards
 * keeps inside the context between rollouts: rc [N][KBJ_RC_SIZE]; synchronous. A resumed run needs them with ep/es. */
int kbj_env_get_reward_carry(kbj_ctx* ctx, float* rc_h);
int kbj_env_set_reward_carry(kbj_ctx* ctx, const float* rc_h);

/* ---- rewards, row a23 --------------------------------------------------------------------- */
/* replaces: get_rewards() stack evaluated over the trajectory (train.py:125-506, 1224-1256).
 * aux_d [T][N][72] -> reward_d [T][N] (sum of scale*term), comps_d [T][N][12] unscaled terms or NULL.
 * The StatefulReward carries persist inside the context across calls. */
int kbj_rewards(kbj_ctx* ctx, const float* aux_d, int T, float* reward_d, float* comps_d);

/* ---- actor-critic, rows a4-a11 ------------------------------------------------------------- */
/* Flat parameter vector layout (floats), equinox leaf order of Model(actor, critic)
 * (train.py:847-1046; convert.py:44-46):
 *   actor : input_proj.weight [H][65], input_proj.bias [H],
 *           rnns[l].weight_ih [4H][H], rnns[l].weight_hh [4H][H], rnns[l].bias [4H]   (l = 0..depth-1)
 *           output_proj.weight [40][H], output_proj.bias [40]
 *   critic: input_proj.weight [H][475], input_proj.bias [H], rnns[l]..., output_proj.weight [1][H], output_proj.bias [1]
 */
size_t kbj_param_count(const kbj_config* cfg);
size_t kbj_actor_param_count(const kbj_config* cfg);
/* replaces: get_model(InitParams(key)) (train.py:1278-1327): U(+-1/sqrt(fan_in)) init from a threefry stream */
int kbj_init_params(kbj_ctx* ctx, uint32_t seed, float* params_d);

/* The sagittal mirror of a PACKED observation row (mirror_obs / mirror_cmd / mirror_joints, train.py:1574-1756, applied to the rows
 * run_actor / run_critic pack, train.py:1351-1433) as the table the kernels use: out[k] = mul[k] * in[src[k]] + add[k] (the affine part
 * re-normalises joint positions whose source and destination joints have different biases / ranges). Host-only, needs no device:
 * fills KBJ_LD_ACTOR (critic = 0) or KBJ_LD_CRITIC (critic != 0) entries and returns that count (< 0 on error).
 * tests/test_ref_constants.py compares it with the table derived from the reference's source text. */
int kbj_mirror_table(const void* model_blob, size_t model_bytes, int critic, int32_t* src_h, float* mul_h, float* add_h);

/* Model carry (train.py:1049-1055, 1526-1543), device arrays owned by the caller:
 *   actor_hc_d / critic_hc_d [depth][2][N][H] (h then c per layer), lpf_d [N][20] */
typedef struct kbj_carry {
  float* actor_hc_d;
  float* critic_hc_d;
  float* lpf_d;
  /* mirror branches of the aux losses (train.py:1051-1055, 1463-1481); may be NULL when both mirror scales are 0 */
  float* actor_mirror_hc_d;
  float* critic_mirror_hc_d;
  float* lpf_mirror_d;
} kbj_carry;

/* replaces: sample_action() (train.py:1545-1572) for all envs at one control step, fused with what the
 * on-policy get_ppo_variables() pass would recompute for this step (log-prob, value; train.py:1435-1508):
 *   actor_obs_d [N][68], critic_obs_d [N][476] -> action_d [N][20], logp_d [N], value_d [N]; carry updated in place.
 * step_index seeds the Gaussian draw (RNG stream KBJ_RNG_ACTION); argmax!=0 returns the mode (train.py:1564). */
int kbj_policy_step(kbj_ctx* ctx, const float* params_d, const float* actor_obs_d, const float* critic_obs_d, kbj_carry* carry,
                    uint32_t seed, uint32_t step_index, int argmax, float* action_d, float* logp_d, float* value_d);
/* carry <- initial carry where done (train.py:1502-1506): done_d [N] float (aux KBJ_AUX_DONE column, stride in floats) */
int kbj_carry_reset(kbj_ctx* ctx, kbj_carry* carry, const float* done_d, int done_stride);

/* ---- rollout, §3.2 ------------------------------------------------------------------------- */
typedef struct kbj_traj {
  int32_t T, N;
  float* actor_obs_d;  /* [T+1][N][68]  row T = observation after the last step (row 0 of the next rollout) */
  float* critic_obs_d; /* [T+1][N][476] */
  float* aux_d;        /* [T+1][N][72]  */
  float* action_d;     /* [T][N][20] */
  float* logp_d;       /* [T][N] on-policy log-prob  */
  float* value_d;      /* [T][N] on-policy value     */
  float* reward_d;     /* [T][N] */
  float* carry0_actor_hc_d;  /* [depth][2][N][H] carry at the start of the trajectory (BPTT initial state) */
  float* carry0_critic_hc_d;
  float* carry0_lpf_d;       /* [N][20] */
  float* carry0_actor_mirror_hc_d;   /* mirror-branch carries at the start of the trajectory (NULL when the mirror losses are off) */
  float* carry0_critic_mirror_hc_d;
  float* carry0_lpf_mirror_d;
  float* reward_comps_d;     /* optional [T][N][12]: unscaled reward terms of the rollout (logging), or NULL */
  float* qstate_d;           /* optional [T][N][KBJ_QSTATE_SIZE]: qpos / qvel after every step + the positions its last forward pass ran on (the
                              * fields a ksim Trajectory carries for user reward terms: trajectory.qpos / .qvel / .xpos / .xquat, train.py:262-506), or NULL */
} kbj_traj;
/* replaces: ksim's jitted rollout scan (vmap over envs, scan over T). Copies observation row T to row 0,
 * snapshots the carry, then T x (policy_step, env_step, carry_reset), then rewards. */
int kbj_rollout(kbj_ctx* ctx, const float* params_d, kbj_carry* carry, uint32_t seed, uint32_t first_step_index, kbj_traj* traj);

/* replaces: `argmax=True` of sample_action during ksim's validation rollouts (train.py:1564, valid_every_n_steps train.py:1789): the following
 * kbj_rollout calls of this context act with the distribution's mode (the filtered mean) instead of a sample; the stored log-probs are those of the
 * mode. 0 restores sampling. (kbj_policy_step takes the flag per call.) */
int kbj_set_rollout_argmax(kbj_ctx* ctx, int argmax);

/* ---- PPO update, rows a9, a12, a13 ---------------------------------------------------------- */
/* replaces: ksim GAE (gamma, lam: train.py:1769-1770). done from aux; adv_d/target_d [T][N] */
int kbj_gae(kbj_ctx* ctx, const kbj_traj* traj, float* adv_d, float* target_d);
/* replaces: the loss/grad of one minibatch: get_ppo_variables under grad (BPTT through T LSTM steps,
 * train.py:1435-1524) + ksim's clipped PPO loss. env_idx_d [B] int32 env indices of the minibatch.
 * When config.actor_mirror_loss_scale / critic_mirror_loss_scale are non-zero the mirror branches (train.py:1463-1481) run
 * under the gradient too and their aux losses are added; the carry / trajectory mirror arrays must then be non-NULL.
 * grad_d [P] flat gradient (overwritten), metrics_d [10]: loss, policy, value, entropy, clipfrac, kl, adv_mean, adv_std,
 * action_mirror_loss, value_mirror_loss */
int kbj_ppo_grad(kbj_ctx* ctx, const float* params_d, const kbj_traj* traj, const int32_t* env_idx_d, int B, const float* adv_d,
                 const float* target_d, float* grad_d, float* metrics_d);
/* replaces: get_ppo_variables(model, trajectory, model_carry, rng) (train.py:1510-1524) = the scan of _ppo_scan_fn (train.py:1435-1508)
 * WITHOUT taking gradients, for the B = config.batch_size envs env_idx_d names of ANY trajectory of the context's shape (one kbj_rollout
 * produced, or one reloaded into the same arrays): both nets run through the T steps from the trajectory's start carries with the carry
 * reset where `done` (train.py:1502-1506), and the per-step variables of PPOVariables come back as [T][B] arrays in env_idx order
 * (row t * B + b belongs to env env_idx_d[b]): log_probs of the stored actions (train.py:1452), values (:1455), entropy (:1486),
 * action_std [T][B][20] (:1487). entropy_d / action_std_d / action_mean_d may be NULL. The mirror aux losses are not evaluated here. */
typedef struct kbj_ppo_vars {
  float* logp_d;         /* [T][B] */
  float* value_d;        /* [T][B] */
  float* entropy_d;      /* [T][B] or NULL */
  float* action_std_d;   /* [T][B][20] or NULL */
  float* action_mean_d;  /* [T][B][20] or NULL: the filtered mean (the distribution's mode, train.py:936-939) */
} kbj_ppo_vars;
int kbj_ppo_forward(kbj_ctx* ctx, const float* params_d, const kbj_traj* traj, const int32_t* env_idx_d, int B, kbj_ppo_vars* out);
/* Next-minibatch hint (no reference counterpart: XLA schedules its whole update as one program): the NEXT kbj_ppo_grad / kbj_ppo_forward of
 * this context will be called with this trajectory and these indices (the same pointers). What that call gathers before its first
 * recurrence and that does not depend on the parameters - the actor's observation rows, the keep flags, the start carries - is queued now,
 * on a side lane behind everything enqueued on the context's stream so far, so it runs under the optimizer step between the two calls.
 * Purely a scheduling hint: results are identical with and without it; a call with other arguments simply ignores the prefetch. Call it
 * right after kbj_ppo_grad(k) with minibatch k + 1's indices. */
int kbj_ppo_prefetch(kbj_ctx* ctx, const kbj_traj* traj, const int32_t* env_idx_d);
/* Data-parallel overlap (no reference counterpart: the reference has no collective call site). After kbj_ppo_grad, work enqueued on
 * `hip_stream` behind this call starts once the ACTOR's slice of the gradient, grad_d[0, kbj_actor_param_count()), is final - about half
 * a millisecond before the call's own stream sees the whole gradient - so a host may all-reduce that slice on a second stream under the
 * critic's tail and the rest on the call's stream (host/task.py `overlap_allreduce`). */
int kbj_stream_wait_actor_grad(kbj_ctx* ctx, void* hip_stream);
/* Data-parallel variant of the advantage normalisation (ksim normalises the advantages of the batch it is given and
 * the reference has no multi-device call site to compare with): the following kbj_ppo_grad calls normalise with the statistics the caller
 * supplies - three doubles on the device, (sum adv, sum adv^2, sample count), e.g. the minibatch's sums all-reduced over the ranks so that
 * every rank normalises with the mean / variance of the GLOBAL minibatch - instead of their own minibatch's. NULL restores the default.
 * The pointer is read when kbj_ppo_grad runs (stream-ordered): fill it on the context's stream before the call. */
int kbj_set_advantage_sums(kbj_ctx* ctx, const double* sums_d);
/* Residency of the persistent recurrences (no reference counterpart: XLA has no persistent kernels). The T-step LSTM recurrences of
 * kbj_ppo_grad / kbj_ppo_forward are persistent launches whose workgroups hand tiles to each other, so a launch's whole grid must be resident:
 * *grid_wgs = workgroups of ONE recurrence launch at this configuration, *concurrent = launches this context keeps in flight at once (2: actor-
 * and critic-type net; 1 on the one-stream schedule), *slots = workgroups of the worst-fitting recurrence kernel the device holds (occupancy
 * query x CUs). kbj_create refuses a configuration with concurrent x grid > slots; a host that puts SEVERAL contexts (ranks) on one GPU must keep
 * the sum over them within slots (bench.py --share-gpu). A grid that cannot be placed fails within the wait bound (2 s, KBJ_SEQ_TIMEOUT_MS) and
 * every later launch of the call aborts at once: fail-stop in seconds, never a crawl. */
int kbj_recurrence_residency(kbj_ctx* ctx, int* grid_wgs, int* concurrent, int* slots);
/* replaces: optax.adamw + global-norm clip (train.py:1059-1077). step is 1-based. grad_scale multiplies the
 * gradient first (1/world_size after an all-reduce sum). */
int kbj_adamw_step(kbj_ctx* ctx, float* params_d, float* m_d, float* v_d, const float* grad_d, int64_t step, float grad_scale);
/* replaces: optax.cosine_decay_schedule feeding adamw (train.py:1067-1077): the host evaluates the schedule and sets the
 * learning rate used by the following kbj_adamw_step calls. Any finite value; a negative one moves the parameters ALONG the Adam direction,
 * which is what train.py:1074-1075 (scale_by_adam chained with scale_by_schedule, no sign flip) does as written. */
int kbj_set_learning_rate(kbj_ctx* ctx, float learning_rate);

/* per-launch timing of the dominant kernels, measured with HIP events on the context's stream (bench.py roofline) */
int kbj_profile_begin(kbj_ctx* ctx);
int kbj_profile_end(kbj_ctx* ctx, float* env_step_ms, int* env_step_launches, float* nn_ms, int* nn_launches);
/* per-kernel totals of the matrix-core kernels launched between kbj_profile_begin and kbj_profile_end: every launch is
 * bracketed by two HIP events on the stream it is launched on; flops = algorithmic 2*M*N*K of those launches */
typedef struct kbj_kernel_stat {
  char name[96];     /* kernel name as rocprofv3 prints it */
  int32_t launches;
  float total_ms;    /* sum of the launch durations */
  double flops;
} kbj_kernel_stat;
int kbj_profile_kernel_stats(kbj_ctx* ctx, kbj_kernel_stat* out, int capacity, int* count);

#ifdef __cplusplus
}
#endif
#endif /* KBJ_H */
