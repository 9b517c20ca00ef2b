// kbj_env.hip — HIP kernels + C-ABI entry points of the environment side of the hot path:
// kbj_env_reset_all / kbj_env_step (one wavefront = one env, state in LDS; kbj_env_*.h) and kbj_rewards.
#include <hip/hip_runtime.h>
#include "kbj_env_task.h"
#include "kbj_ctx.h"

using namespace kbj;

namespace {



// grid = N workgroups of one wavefront; env state rows are read/written lane-contiguously (coalesced)
__global__ __launch_bounds__(64) void env_reset_kernel(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ mc, const PhysConst* __restrict__ pcp, uint32_t seed,
                                                       float* __restrict__ ep, float* __restrict__ es, float* actor0, float* critic0, float* aux0) {
  __shared__ KbjShared S;
  const int env = blockIdx.x;
  PFOR(k, (int)(sizeof(KbjModelLds) / sizeof(float))) reinterpret_cast<float*>(&S.mc)[k] = mc[k];
  PFOR(k, (int)(sizeof(PhysConst) / sizeof(float))) reinterpret_cast<float*>(&S.pc)[k] = reinterpret_cast<const float*>(pcp)[k];
  PFOR(k, KBJ_ES_SIZE) S.es[k] = 0;
  PFOR(k, 12) S.zrow[k] = 0;
  KBJ_SYNC();
  Rng rng{seed, (uint32_t)(c->env_id_offset + env)};
  const PhysConst& pc = S.pc;
  task_reset(S, *m, *c, pc, rng);
  task_write_obs(S, *m, *c, rng, actor0 + (size_t)env * KBJ_LD_OF(KBJ_NOBS_ACTOR + c->extra_obs_actor), critic0 + (size_t)env * KBJ_LD_OF(KBJ_NOBS_CRITIC + c->extra_obs_critic), aux0 + (size_t)env * KBJ_AUX_SIZE);
  PFOR(k, KBJ_EP_SIZE) ep[(size_t)env * KBJ_EP_SIZE + k] = S.ep[k];
  PFOR(k, KBJ_ES_SIZE) es[(size_t)env * KBJ_ES_SIZE + k] = S.es[k];
}

// re-initialise the envs whose mask entry is non-zero, exactly as env_step_kernel does for an env its own terminations finish (same
// task_reset / task_write_obs on the env's state row: the episode counter advances, the randomisers, reset distributions and the first
// command are drawn from the env's streams) and rewrite their next observation rows; other envs are untouched. For terminations decided
// OUTSIDE the kernel (user-written Termination terms on the host, train.py:817 protocol).
__global__ __launch_bounds__(64) void env_reset_where_kernel(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ mc, const PhysConst* __restrict__ pcp, uint32_t seed,
                                                             float* __restrict__ ep, float* __restrict__ es, const float* __restrict__ mask, float* actor_next,
                                                             float* critic_next, float* aux_next) {
  __shared__ KbjShared S;
  const int env = blockIdx.x;
  if (mask[env] == 0.0f) return;     // uniform over the workgroup
  PFOR(k, (int)(sizeof(KbjModelLds) / sizeof(float))) reinterpret_cast<float*>(&S.mc)[k] = mc[k];
  PFOR(k, (int)(sizeof(PhysConst) / sizeof(float))) reinterpret_cast<float*>(&S.pc)[k] = reinterpret_cast<const float*>(pcp)[k];
  PFOR(k, KBJ_EP_SIZE) S.ep[k] = ep[(size_t)env * KBJ_EP_SIZE + k];
  PFOR(k, KBJ_ES_SIZE) S.es[k] = es[(size_t)env * KBJ_ES_SIZE + k];
  PFOR(k, 12) S.zrow[k] = 0;
  KBJ_SYNC();
  Rng rng{seed, (uint32_t)(c->env_id_offset + env)};
  const PhysConst& pc = S.pc;
  task_reset(S, *m, *c, pc, rng);
  task_write_obs(S, *m, *c, rng, actor_next + (size_t)env * KBJ_LD_OF(KBJ_NOBS_ACTOR + c->extra_obs_actor), critic_next + (size_t)env * KBJ_LD_OF(KBJ_NOBS_CRITIC + c->extra_obs_critic), aux_next + (size_t)env * KBJ_AUX_SIZE);
  PFOR(k, KBJ_EP_SIZE) ep[(size_t)env * KBJ_EP_SIZE + k] = S.ep[k];
  PFOR(k, KBJ_ES_SIZE) es[(size_t)env * KBJ_ES_SIZE + k] = S.es[k];
}

// user-written Reset terms (train.py:833-844 protocol): the generalised state of every env as device arrays ...
__global__ __launch_bounds__(256) void env_get_qstate_kernel(int N, const float* __restrict__ es, float* __restrict__ qpos, float* __restrict__ qvel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, env = i / 64, k = i % 64;
  if (env >= N) return;
  if (k < KBJ_NQ) qpos[(size_t)env * KBJ_NQ + k] = es[(size_t)env * KBJ_ES_SIZE + KBJ_ES_QPOS + k];
  if (k < KBJ_NV) qvel[(size_t)env * KBJ_NV + k] = es[(size_t)env * KBJ_ES_SIZE + KBJ_ES_QVEL + k];
}
// ... and back for the masked envs: new positions / velocities, cleared warm start, then what task_reset does behind the state it draws itself -
// one forward pass (PD on the held action, kinematics, sensors), the lagged projected gravity re-seeded, the next observation rows rewritten
__global__ __launch_bounds__(64) void env_set_qstate_kernel(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ mc, const PhysConst* __restrict__ pcp, uint32_t seed,
                                                            float* __restrict__ ep, float* __restrict__ es, const float* __restrict__ mask, const float* __restrict__ qpos,
                                                            const float* __restrict__ qvel, float* actor_next, float* critic_next, float* aux_next) {
  __shared__ KbjShared S;
  const int env = blockIdx.x;
  if (mask && mask[env] == 0.0f) return;     // uniform over the workgroup
  PFOR(k, (int)(sizeof(KbjModelLds) / sizeof(float))) reinterpret_cast<float*>(&S.mc)[k] = mc[k];
  PFOR(k, (int)(sizeof(PhysConst) / sizeof(float))) reinterpret_cast<float*>(&S.pc)[k] = reinterpret_cast<const float*>(pcp)[k];
  PFOR(k, KBJ_EP_SIZE) S.ep[k] = ep[(size_t)env * KBJ_EP_SIZE + k];
  PFOR(k, KBJ_ES_SIZE) S.es[k] = es[(size_t)env * KBJ_ES_SIZE + k];
  PFOR(k, 12) S.zrow[k] = 0;
  KBJ_SYNC();
  PFOR(k, KBJ_NQ) S.es[KBJ_ES_QPOS + k] = qpos[(size_t)env * KBJ_NQ + k];
  PFOR(k, KBJ_NV) { S.es[KBJ_ES_QVEL + k] = qvel[(size_t)env * KBJ_NV + k]; S.es[KBJ_ES_WARM + k] = 0; }
  KBJ_SYNC();
  PFOR(w, 1) {   // a unit base quaternion whatever the term returned
    float* q = S.es + KBJ_ES_QPOS + 3;
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0)) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else if (fabsf(n - 1.0f) > 1e-6f) { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }     // (a term that leaves the state alone leaves its bits alone)
    S.pushing = 0;
  }
  KBJ_SYNC();
  Rng rng{seed, (uint32_t)(c->env_id_offset + env)};
  const PhysConst& pc = S.pc;
  task_pd(S, S.es + KBJ_ES_ACT_PREV);
  phys_forward(S, S.mc, pc);
  PFOR(k, 3) S.es[KBJ_ES_PGLAG + k] = S.pg[k];
  KBJ_SYNC();
  task_write_obs(S, *m, *c, rng, actor_next + (size_t)env * KBJ_LD_OF(KBJ_NOBS_ACTOR + c->extra_obs_actor), critic_next + (size_t)env * KBJ_LD_OF(KBJ_NOBS_CRITIC + c->extra_obs_critic), aux_next + (size_t)env * KBJ_AUX_SIZE);
  PFOR(k, KBJ_ES_SIZE) es[(size_t)env * KBJ_ES_SIZE + k] = S.es[k];
}

// overwrite the joystick command of the envs whose mask entry is non-zero (mask == nullptr: all envs): the env's state row (the next step's
// rewards and its command-switch draw start from it) and the command columns of the NEXT observation rows + aux record, zero-command flag
// included — what task_write_obs wrote there from the kernel's own command. One thread per (env, command slot).
__global__ __launch_bounds__(256) void env_set_command_kernel(int N, int lda, int ldc, float* __restrict__ es, const float* __restrict__ mask, const float* __restrict__ cmd, float* actor_next,
                                                              float* critic_next, float* aux_next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, env = i / KBJ_NCMD, k = i % KBJ_NCMD;
  if (env >= N || (mask && mask[env] == 0.0f)) return;
  const float* c = cmd + (size_t)env * KBJ_NCMD;
  const float v = c[k];
  es[(size_t)env * KBJ_ES_SIZE + KBJ_ES_CMD + k] = v;
  actor_next[(size_t)env * lda + KBJ_OBS_CMD + k] = v;
  critic_next[(size_t)env * ldc + KBJ_OBS_CMD + k] = v;
  aux_next[(size_t)env * KBJ_AUX_SIZE + KBJ_AUX_CMD + k] = v;
  if (k == 0) {
    const float zc = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) < 1e-3f ? 1.0f : 0.0f;
    actor_next[(size_t)env * lda + KBJ_OBS_ZEROCMD] = zc;
    critic_next[(size_t)env * ldc + KBJ_OBS_ZEROCMD] = zc;
  }
}

// register budget of the step kernel: 13.4 KB of LDS lets 12 single-wavefront workgroups share a CU (3 waves/SIMD), which
// needs <= 168 VGPRs. `amdgpu_num_vgpr(N)` makes hipcc allocate 2 N registers for this wave64 kernel (floor 129): N = 84 gives
// exactly 168 with 61 spilled values. Measured (8192 envs, ms/step): no cap (2 waves/SIMD, no spills) 3.26 -> N = 62 (129 VGPRs,
// 90 spills) 1.98 -> N = 72 (144, 70 spills) 1.83 -> N = 84 (168, 61 spills) 1.80. `amdgpu_waves_per_eu(3,3)` also lands on 168
// registers but schedules worse (2.15).
#ifndef KBJ_ENV_NUM_VGPR
#define KBJ_ENV_NUM_VGPR 84
#endif
// REC: the form that also writes the per-step state record (kbj_traj.qstate_d / kbj_env_record_state). The default rollout runs the other
// one, whose code is exactly the kernel without the feature (two more pointers live through the substep loop cost 2 VGPR / 8 SGPR spills);
// the two are separate __global__ functions so that the hot kernel keeps its name in every profile.
template <bool REC>
__device__ __forceinline__ void env_step_body(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ mc, const PhysConst* __restrict__ pcp, uint32_t seed,
                                              float* __restrict__ ep, float* __restrict__ es, const float* __restrict__ action,
                                              float* aux_t, float* actor_next, float* critic_next, float* aux_next, int env0, float* qstate_t) {
  __shared__ KbjShared S;
  const int env = env0 + blockIdx.x;   // a launch covers the env range [env0, env0 + gridDim.x)
  PFOR(k, (int)(sizeof(KbjModelLds) / sizeof(float))) reinterpret_cast<float*>(&S.mc)[k] = mc[k];
  PFOR(k, (int)(sizeof(PhysConst) / sizeof(float))) reinterpret_cast<float*>(&S.pc)[k] = reinterpret_cast<const float*>(pcp)[k];
  PFOR(k, KBJ_EP_SIZE) S.ep[k] = ep[(size_t)env * KBJ_EP_SIZE + k];
  PFOR(k, KBJ_ES_SIZE) S.es[k] = es[(size_t)env * KBJ_ES_SIZE + k];
  PFOR(k, 12) S.zrow[k] = 0;
  KBJ_SYNC();
  Rng rng{seed, (uint32_t)(c->env_id_offset + env)};
  const PhysConst& pc = S.pc;
  KBJ_STAMP(18);
  task_step(S, *m, *c, pc, rng, action + (size_t)env * KBJ_NU, aux_t + (size_t)env * KBJ_AUX_SIZE, actor_next + (size_t)env * KBJ_LD_OF(KBJ_NOBS_ACTOR + c->extra_obs_actor),
            critic_next + (size_t)env * KBJ_LD_OF(KBJ_NOBS_CRITIC + c->extra_obs_critic), aux_next + (size_t)env * KBJ_AUX_SIZE,
            REC ? qstate_t + (size_t)env * KBJ_QSTATE_SIZE : nullptr);
  if (S.done) PFOR(k, KBJ_EP_SIZE) ep[(size_t)env * KBJ_EP_SIZE + k] = S.ep[k];
  PFOR(k, KBJ_ES_SIZE) es[(size_t)env * KBJ_ES_SIZE + k] = S.es[k];
  KBJ_STAMP(19);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(KBJ_ENV_NUM_VGPR))) void env_step_kernel(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ mc, const PhysConst* __restrict__ pcp, uint32_t seed,
                                                      float* __restrict__ ep, float* __restrict__ es, const float* __restrict__ action,
                                                      float* aux_t, float* actor_next, float* critic_next, float* aux_next, int env0) {
  env_step_body<false>(m, c, mc, pcp, seed, ep, es, action, aux_t, actor_next, critic_next, aux_next, env0, nullptr);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(KBJ_ENV_NUM_VGPR))) void env_step_record_kernel(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ mc, const PhysConst* __restrict__ pcp, uint32_t seed,
                                                      float* __restrict__ ep, float* __restrict__ es, const float* __restrict__ action,
                                                      float* aux_t, float* actor_next, float* critic_next, float* aux_next, int env0, float* qstate_t) {
  env_step_body<true>(m, c, mc, pcp, seed, ep, es, action, aux_t, actor_next, critic_next, aux_next, env0, qstate_t);
}

#ifdef KBJ_ENV_STAMPS
extern "C" int kbj_debug_env_stamps(unsigned long long* out32, int clear) {   // diagnostics build only (tools/env_stamps.py)
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(kbj_env_stamp_acc), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (clear) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(kbj_env_stamp_acc), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

// ---- reward stack (train.py:125-506, weights train.py:1225-1256): one thread scans one env's trajectory ----
__global__ void rewards_kernel(const kbj_model* __restrict__ m, const kbj_config* __restrict__ c, const float* __restrict__ aux, int T, int N,
                               float* __restrict__ carry, float* __restrict__ reward, float* __restrict__ comps) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  const float* scales = c->reward_scale;   // user-editable like the reference's get_rewards() (train.py:1224-1256)
  const float ctrl_dt = c->ctrl_dt;
  float* rc = carry + (size_t)env * KBJ_RC_SIZE;
  float tsingle = rc[KBJ_RC_TSINGLE], air[2] = {rc[KBJ_RC_AIRTIME], rc[KBJ_RC_AIRTIME + 1]};
  bool pcon[2] = {rc[KBJ_RC_CONTACT] != 0, rc[KBJ_RC_CONTACT + 1] != 0};
  float pq[6];
  bool pdone = false;
  for (int t = 0; t < T; ++t) {
    const float* a = aux + ((size_t)t * N + env) * KBJ_AUX_SIZE;
    float r[KBJ_NREW];
    const float* cmd = a + KBJ_AUX_CMD;
    bool zc = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]) < 1e-3f;
    bool done = a[KBJ_AUX_DONE] != 0;
    float bq[4] = {a[KBJ_AUX_BQUAT], a[KBJ_AUX_BQUAT + 1], a[KBJ_AUX_BQUAT + 2], a[KBJ_AUX_BQUAT + 3]}, be[3];
    quat_to_euler(bq, be);
    {  // linvel (train.py:274-292)
      float ye[3] = {0, 0, be[2]}, yq[4], v[3] = {cmd[0], cmd[1], 0}, g[3];
      euler_to_quat(ye, yq); rotate_by_quat(v, yq, false, g);
      float ex = a[KBJ_AUX_QVEL] - g[0], ey = a[KBJ_AUX_QVEL + 1] - g[1], err = sqrtf(ex * ex + ey * ey);
      r[KBJ_REW_LINVEL] = expf(-(zc ? err : err * err) / c->rew_linvel_err);
    }
    r[KBJ_REW_ANGVEL] = expf(-fabsf(a[KBJ_AUX_QVEL + 5] - cmd[2]) / c->rew_angvel_err);  // train.py:301-306
    {  // roll_pitch (train.py:316-334)
      float e1[3] = {be[0], be[1], 0}, q1[4], e2[3] = {cmd[4], cmd[5], 0}, q2[4];
      euler_to_quat(e1, q1); euler_to_quat(e2, q2);
      float d_ = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
      r[KBJ_REW_ROLL_PITCH] = expf(-(1 - d_ * d_) / (zc ? c->rew_rollpitch_err_zero : c->rew_rollpitch_err));
    }
    {  // base_height (train.py:377-388)
      float low = fminf(a[KBJ_AUX_LFZ] - c->rew_foot_origin_height, a[KBJ_AUX_RFZ] - c->rew_foot_origin_height);
      float h = a[KBJ_AUX_BASEZ] - low;
      r[KBJ_REW_BASE_HEIGHT] = expf(-fabsf(h - (cmd[3] + c->rew_standard_height)) / c->rew_height_err);
    }
    {  // arm_pos (train.py:261-265)
      float e = 0;
      for (int j = 0; j < 10; ++j) { float dq = a[KBJ_AUX_ARMQ + j] - (cmd[6 + j] + m->joint_bias[10 + j]); e += dq * dq; }
      r[KBJ_REW_ARM_POS] = expf(-e / c->rew_armpos_err);
    }
    bool cl = a[KBJ_AUX_TOUCH] > 0.1f, cr = a[KBJ_AUX_TOUCH + 1] > 0.1f;
    {  // single_contact (train.py:138-154), grace period 2.0 s
      float ts = (cl != cr) ? 0.0f : tsingle + ctrl_dt;
      if (zc) ts = c->rew_grace_period;
      tsingle = ts;
      r[KBJ_REW_SINGLE_CONTACT] = zc ? 1.0f : (ts < c->rew_grace_period ? 1.0f : 0.0f);
    }
    r[KBJ_REW_NO_CONTACT] = zc ? 0.0f : ((cl || cr) ? 0.0f : 1.0f);  // train.py:161-165
    {  // feet_airtime (train.py:197-213)
      bool con[2] = {cl, cr};
      float rew = 0;
      for (int f = 0; f < 2; ++f) {
        bool first = con[f] && !pcon[f] && !done;
        rew += (air[f] - c->rew_touchdown_penalty) * (first ? 1.0f : 0.0f);
        air[f] = (con[f] || done) ? 0.0f : air[f] + ctrl_dt;
        pcon[f] = con[f];
      }
      r[KBJ_REW_FEET_AIRTIME] = zc ? 0.0f : rew;
    }
    {  // feet_orient (train.py:418-457)
      float rpy = 0, rp = 0;
      for (int f = 0; f < 2; ++f) {
        const float* fq_ = a + (f ? KBJ_AUX_RFQUAT : KBJ_AUX_LFQUAT);
        float fq[4] = {fq_[0], fq_[1], fq_[2], fq_[3]};
        float te[3] = {f ? 1.5707963267948966f : -1.5707963267948966f, 0, be[2] - 3.141592653589793f}, tq[4];
        euler_to_quat(te, tq);
        float d1 = tq[0] * fq[0] + tq[1] * fq[1] + tq[2] * fq[2] + tq[3] * fq[3];
        rpy += 1 - d1 * d1;
        float fe[3]; quat_to_euler(fq, fe); fe[2] = 0;
        float fq0[4]; euler_to_quat(fe, fq0);
        te[2] = 0; euler_to_quat(te, tq);
        float d2 = tq[0] * fq0[0] + tq[1] * fq0[1] + tq[2] * fq0[2] + tq[3] * fq0[3];
        rp += 1 - d2 * d2;
      }
      r[KBJ_REW_FEET_ORIENT] = expf(-(fabsf(cmd[2]) > 1e-3f ? rp : rpy) / c->rew_feetorient_err);
    }
    {  // com_distance (train.py:466-478)
      float cd = a[KBJ_AUX_COMDIST];
      r[KBJ_REW_COM_DISTANCE] = (cd >= 0 && zc) ? expf(-cd / c->rew_comdist_err) : 0.0f;
    }
    {  // base_accel (train.py:487-494)
      float e = 0;
      if (t > 0 && !pdone) for (int k = 0; k < 6; ++k) e += fabsf(a[KBJ_AUX_QVEL + k] - pq[k]);
      for (int k = 0; k < 6; ++k) pq[k] = a[KBJ_AUX_QVEL + k];
      pdone = done;
      r[KBJ_REW_BASE_ACCEL] = expf(-e / c->rew_baseaccel_err);
    }
    {  // torque (train.py:503-506)
      float s = 0;
      for (int u = 0; u < KBJ_NU; ++u) s += expf(-fabsf(a[KBJ_AUX_CTRL + u]) / c->rew_torque_err);
      r[KBJ_REW_TORQUE] = zc ? s / KBJ_NU : 1.0f;
    }
    float tot = 0;
    for (int k = 0; k < KBJ_NREW; ++k) { tot += scales[k] * r[k]; if (comps) comps[((size_t)t * N + env) * KBJ_NREW + k] = r[k]; }
    reward[(size_t)t * N + env] = tot;
  }
  rc[KBJ_RC_TSINGLE] = tsingle; rc[KBJ_RC_AIRTIME] = air[0]; rc[KBJ_RC_AIRTIME + 1] = air[1];
  rc[KBJ_RC_CONTACT] = pcon[0] ? 1.0f : 0.0f; rc[KBJ_RC_CONTACT + 1] = pcon[1] ? 1.0f : 0.0f;
}

__global__ void init_reward_carry_kernel(float* carry, int N) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  float* rc = carry + (size_t)env * KBJ_RC_SIZE;
  for (int k = 0; k < KBJ_RC_SIZE; ++k) rc[k] = 0;
  rc[KBJ_RC_CONTACT] = 1.0f; rc[KBJ_RC_CONTACT + 1] = 1.0f;  // train.py:175-178 initial contact carry True
}

}  // namespace

void kbj_nn_drop_prefetch(kbj_ctx* ctx);   // kbj_nn.hip: a pending next-minibatch hint dies with the trajectory contents it was made from

// one control step of the envs [env0, env0 + count) on stream s; row pointers are those of env 0
int kbj_env_step_range(kbj_ctx* ctx, hipStream_t s, int env0, int count, const float* action_d, float* aux_t_d, float* actor_next_d, float* critic_next_d,
                       float* aux_next_d, float* qstate_t_d) {
  KbjKernelTimer timer(s, KBJ_KIND_ENV_STEP, 0.0);
  if (qstate_t_d)
    hipLaunchKernelGGL(env_step_record_kernel, dim3(count), dim3(64), 0, s, ctx->model_d, ctx->cfg_d, ctx->mc_d, (const PhysConst*)ctx->pc_d, ctx->seed, ctx->ep_d, ctx->es_d, action_d, aux_t_d,
                       actor_next_d, critic_next_d, aux_next_d, env0, qstate_t_d);
  else
    hipLaunchKernelGGL(env_step_kernel, dim3(count), dim3(64), 0, s, ctx->model_d, ctx->cfg_d, ctx->mc_d, (const PhysConst*)ctx->pc_d, ctx->seed, ctx->ep_d, ctx->es_d, action_d, aux_t_d,
                       actor_next_d, critic_next_d, aux_next_d, env0);
  KBJ_CHECK_LAUNCH(ctx, "env_step_kernel");
  return 0;
}

extern "C" {

int kbj_env_reset_all(kbj_ctx* ctx, uint32_t seed, float* actor0_d, float* critic0_d, float* aux0_d) {
  if (!ctx) return kbj_fail(nullptr, "kbj_env_reset_all: null ctx");
  if (!actor0_d || !critic0_d || !aux0_d) return kbj_fail(ctx, "kbj_env_reset_all: null observation pointer");
  ctx->seed = seed;
  int N = ctx->cfg_h.num_envs;
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  kbj_nn_drop_prefetch(ctx);
  hipLaunchKernelGGL(init_reward_carry_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, ctx->rcarry_d, N);
  KBJ_CHECK_LAUNCH(ctx, "init_reward_carry_kernel");
  hipLaunchKernelGGL(env_reset_kernel, dim3(N), dim3(64), 0, ctx->stream, ctx->model_d, ctx->cfg_d, ctx->mc_d, (const PhysConst*)ctx->pc_d, seed, ctx->ep_d, ctx->es_d, actor0_d,
                     critic0_d, aux0_d);
  KBJ_CHECK_LAUNCH(ctx, "env_reset_kernel");
  return 0;
}

int kbj_env_step(kbj_ctx* ctx, const float* action_d, float* aux_t_d, float* actor_next_d, float* critic_next_d, float* aux_next_d) {
  if (!ctx) return kbj_fail(nullptr, "kbj_env_step: null ctx");
  float* q = ctx->qstate_next;
  ctx->qstate_next = nullptr;     // one-shot (kbj_env_record_state), consumed by THIS call whether it succeeds or not: a call that fails early must not
                                  // leave the pointer armed for a later step, when the host may have freed or reused the row
  if (!action_d || !aux_t_d || !actor_next_d || !critic_next_d || !aux_next_d) return kbj_fail(ctx, "kbj_env_step: null pointer");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  kbj_nn_drop_prefetch(ctx);
  return kbj_env_step_range(ctx, ctx->stream, 0, ctx->cfg_h.num_envs, action_d, aux_t_d, actor_next_d, critic_next_d, aux_next_d, q);
}

int kbj_env_record_state(kbj_ctx* ctx, float* qstate_t_d) {
  if (!ctx) return kbj_fail(nullptr, "kbj_env_record_state: null ctx");
  ctx->qstate_next = qstate_t_d;
  return 0;
}

int kbj_env_reset_where(kbj_ctx* ctx, const float* mask_d, float* actor_next_d, float* critic_next_d, float* aux_next_d) {
  if (!ctx || !mask_d || !actor_next_d || !critic_next_d || !aux_next_d) return kbj_fail(ctx, "kbj_env_reset_where: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  kbj_nn_drop_prefetch(ctx);
  hipLaunchKernelGGL(env_reset_where_kernel, dim3(ctx->cfg_h.num_envs), dim3(64), 0, ctx->stream, ctx->model_d, ctx->cfg_d, ctx->mc_d, (const PhysConst*)ctx->pc_d, ctx->seed, ctx->ep_d, ctx->es_d,
                     mask_d, actor_next_d, critic_next_d, aux_next_d);
  KBJ_CHECK_LAUNCH(ctx, "env_reset_where_kernel");
  return 0;
}

int kbj_env_set_command(kbj_ctx* ctx, const float* mask_d, const float* cmd_d, float* actor_next_d, float* critic_next_d, float* aux_next_d) {
  if (!ctx || !cmd_d || !actor_next_d || !critic_next_d || !aux_next_d) return kbj_fail(ctx, "kbj_env_set_command: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  kbj_nn_drop_prefetch(ctx);
  const int N = ctx->cfg_h.num_envs, n = N * KBJ_NCMD;
  hipLaunchKernelGGL(env_set_command_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, N, KBJ_LD_OF(KBJ_NOBS_ACTOR + ctx->cfg_h.extra_obs_actor),
                     KBJ_LD_OF(KBJ_NOBS_CRITIC + ctx->cfg_h.extra_obs_critic), ctx->es_d, mask_d, cmd_d, actor_next_d, critic_next_d, aux_next_d);
  KBJ_CHECK_LAUNCH(ctx, "env_set_command_kernel");
  return 0;
}

int kbj_env_get_qstate(kbj_ctx* ctx, float* qpos_d, float* qvel_d) {
  if (!ctx || !qpos_d || !qvel_d) return kbj_fail(ctx, "kbj_env_get_qstate: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  const int N = ctx->cfg_h.num_envs;
  hipLaunchKernelGGL(env_get_qstate_kernel, dim3((N * 64 + 255) / 256), dim3(256), 0, ctx->stream, N, ctx->es_d, qpos_d, qvel_d);
  KBJ_CHECK_LAUNCH(ctx, "env_get_qstate_kernel");
  return 0;
}

int kbj_env_set_qstate(kbj_ctx* ctx, const float* mask_d, const float* qpos_d, const float* qvel_d, float* actor_next_d, float* critic_next_d, float* aux_next_d) {
  if (!ctx || !qpos_d || !qvel_d || !actor_next_d || !critic_next_d || !aux_next_d) return kbj_fail(ctx, "kbj_env_set_qstate: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  kbj_nn_drop_prefetch(ctx);
  hipLaunchKernelGGL(env_set_qstate_kernel, dim3(ctx->cfg_h.num_envs), dim3(64), 0, ctx->stream, ctx->model_d, ctx->cfg_d, ctx->mc_d, (const PhysConst*)ctx->pc_d, ctx->seed, ctx->ep_d, ctx->es_d,
                     mask_d, qpos_d, qvel_d, actor_next_d, critic_next_d, aux_next_d);
  KBJ_CHECK_LAUNCH(ctx, "env_set_qstate_kernel");
  return 0;
}

int kbj_env_get_state(kbj_ctx* ctx, float* ep_h, float* es_h) {
  if (!ctx) return kbj_fail(nullptr, "kbj_env_get_state: null ctx");
  size_t N = ctx->cfg_h.num_envs;
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  KBJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ep_h) KBJ_HIP(ctx, hipMemcpy(ep_h, ctx->ep_d, N * KBJ_EP_SIZE * sizeof(float), hipMemcpyDeviceToHost));
  if (es_h) KBJ_HIP(ctx, hipMemcpy(es_h, ctx->es_d, N * KBJ_ES_SIZE * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int kbj_env_set_state(kbj_ctx* ctx, const float* ep_h, const float* es_h) {
  if (!ctx) return kbj_fail(nullptr, "kbj_env_set_state: null ctx");
  size_t N = ctx->cfg_h.num_envs;
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  KBJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ep_h) KBJ_HIP(ctx, hipMemcpy(ctx->ep_d, ep_h, N * KBJ_EP_SIZE * sizeof(float), hipMemcpyHostToDevice));
  if (es_h) KBJ_HIP(ctx, hipMemcpy(ctx->es_d, es_h, N * KBJ_ES_SIZE * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int kbj_env_get_reward_carry(kbj_ctx* ctx, float* rc_h) {
  if (!ctx || !rc_h) return kbj_fail(ctx, "kbj_env_get_reward_carry: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  KBJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  KBJ_HIP(ctx, hipMemcpy(rc_h, ctx->rcarry_d, (size_t)ctx->cfg_h.num_envs * KBJ_RC_SIZE * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int kbj_env_set_reward_carry(kbj_ctx* ctx, const float* rc_h) {
  if (!ctx || !rc_h) return kbj_fail(ctx, "kbj_env_set_reward_carry: null argument");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  KBJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
  KBJ_HIP(ctx, hipMemcpy(ctx->rcarry_d, rc_h, (size_t)ctx->cfg_h.num_envs * KBJ_RC_SIZE * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int kbj_rewards(kbj_ctx* ctx, const float* aux_d, int T, float* reward_d, float* comps_d) {
  if (!ctx) return kbj_fail(nullptr, "kbj_rewards: null ctx");
  if (!aux_d || !reward_d || T <= 0) return kbj_fail(ctx, "kbj_rewards: bad arguments");
  KBJ_HIP(ctx, hipSetDevice(ctx->device));
  kbj_nn_drop_prefetch(ctx);
  int N = ctx->cfg_h.num_envs;
  hipLaunchKernelGGL(rewards_kernel, dim3((N + 63) / 64), dim3(64), 0, ctx->stream, ctx->model_d, ctx->cfg_d, aux_d, T, N, ctx->rcarry_d,
                     reward_d, comps_d);
  KBJ_CHECK_LAUNCH(ctx, "rewards_kernel");
  return 0;
}

}  // extern "C"
