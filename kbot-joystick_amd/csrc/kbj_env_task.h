// kbj_env_task.h — task layer of one env per wavefront: randomisers, resets, command sampler, PD actuators,
// push events, terminations and observation packing.
// In-tree reference pieces are cited per function (train.py line ranges); definitions of the un-vendored
// ksim-fork pieces are the ones frozen in DESIGN.md "Spec decisions".
#pragma once
#include "kbj_env_phys.h"

namespace kbj {

KBJ_DEV uint32_t f2u(float f) { union { float f; uint32_t u; } x; x.f = f; return x.u; }
KBJ_DEV float u2f(uint32_t u) { union { float f; uint32_t u; } x; x.u = u; return x.f; }

// physics randomisers + per-episode actuator / sensor draws (train.py:1097-1132, 1158-1161, 1191-1198, 1780)
KBJ_DEV void task_randomize(KbjShared& S, const kbj_model& m, const kbj_config& c, const Rng& rng) {
  uint32_t e = f2u(S.es[KBJ_ES_EPISODE]);
  bool on = c.enable_randomizers != 0, noise = c.enable_noise != 0;
  float* ep = S.ep;
  PFOR(b, NB) {
    float s = on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 140 + b, 1 - c.inertia_scale, 1 + c.inertia_scale) : 1.0f;
    ep[KBJ_EP_MASS + b] = m.body_mass[b] * s;
    for (int k = 0; k < 3; ++k) {
      ep[KBJ_EP_INERTIA + 3 * b + k] = m.body_inertia[b][k] * s;
      ep[KBJ_EP_IPOS + 3 * b + k] = m.body_ipos[b][k] + ((b && on) ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 60 + 3 * b + k, -c.com_jitter, c.com_jitter) : 0.0f);
    }
  }
  PFOR(d, NV) {
    ep[KBJ_EP_FRICLOSS + d] = m.dof_frictionloss[d] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, d, c.fricloss_scale_lo, c.fricloss_scale_hi) : 1.0f);
    ep[KBJ_EP_ARMATURE + d] = m.dof_armature[d] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 26 + d, c.armature_scale_lo, c.armature_scale_hi) : 1.0f);
  }
  PFOR(cp, 4) {
    ep[KBJ_EP_CAP_RAD + cp] = m.cap_radius[cp] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 170 + cp, 1 - c.cap_radius_scale, 1 + c.cap_radius_scale) : 1.0f);
    ep[KBJ_EP_CAP_HALF + cp] = m.cap_halflen[cp] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 174 + cp, 1 - c.cap_length_scale, 1 + c.cap_length_scale) : 1.0f);
    for (int k = 0; k < 3; ++k)
      ep[KBJ_EP_CAP_POS + 3 * cp + k] = m.cap_pos[cp][k] + (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 180 + 3 * cp + k, -c.cap_jitter[k], c.cap_jitter[k]) : 0.0f);
  }
  PFOR(u, NU) {
    ep[KBJ_EP_KP + u] = m.kp[u] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 200 + u, 1.0f / c.kp_scale, c.kp_scale) : 1.0f);
    ep[KBJ_EP_KD + u] = m.kd[u] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 220 + u, 1.0f / c.kd_scale, c.kd_scale) : 1.0f);
    ep[KBJ_EP_TAULIM + u] = m.tau_limit[u] * (on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 240 + u, c.torque_limit_scale_low, 1.0f) : 1.0f);
    ep[KBJ_EP_ACTBIAS + u] = on ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 260 + u, -c.action_bias_scale, c.action_bias_scale) : 0.0f;
    ep[KBJ_EP_JPBIAS + u] = noise ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 280 + u, -c.jpos_bias_range, c.jpos_bias_range) : 0.0f;
  }
  PFOR(w, 1) {
    for (int k = 0; k < 3; ++k) ep[KBJ_EP_PGBIAS + k] = noise ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 300 + k, -c.pg_bias, c.pg_bias) : 0.0f;
    ep[KBJ_EP_PGLAG] = noise ? rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 303, c.pg_lag_lo, c.pg_lag_hi) : 0.0f;
    float lat = rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 304, c.latency_lo, c.latency_hi);
    ep[KBJ_EP_LATENCY] = floorf(lat / c.dt + 0.5f);
    // capsule priority 1 beats the floor's 0, so the capsule friction is the contact friction and the
    // floor-friction randomiser (train.py:1112-1114) has nothing to scale
    ep[KBJ_EP_MU] = m.contact_mu;
    for (int k = KBJ_EP_MU + 1; k < KBJ_EP_SIZE; ++k) ep[k] = 0;
  }
  KBJ_SYNC();
}

// one of the six command templates (train.py:741-763)
KBJ_DEV void task_command_template(int mode, float vx, float vy, float wz, float bh, float rx, float ry, const float* arms, float* cmd) {
  for (int k = 0; k < KBJ_NCMD; ++k) cmd[k] = 0;
  if (mode == 0) cmd[0] = vx;
  else if (mode == 1) cmd[1] = vy;
  else if (mode == 2) cmd[2] = wz;
  else if (mode == 3) { cmd[0] = vx; cmd[1] = vy; cmd[2] = wz; for (int j = 0; j < 10; ++j) cmd[6 + j] = arms[j]; }
  else if (mode == 4) { cmd[3] = bh; cmd[4] = rx; cmd[5] = ry; for (int j = 0; j < 10; ++j) cmd[6 + j] = arms[j]; }
}
// UnifiedCommand.initial_command with jax.random's key handling (command_mode == 2; kbj_env_core.h): `rng` is the key the method is called with
KBJ_DEV void task_sample_command_jax(const kbj_model& m, const kbj_config& c, const JaxKey& rng, float* cmd) {
  JaxKey ks[9];   // rng_a .. rng_i = jax.random.split(rng, 9)   (train.py:725)
  for (uint32_t i = 0; i < 9; ++i) ks[i] = jax_split(rng, i);
  const float vx = jax_uniform(ks[1], 0, c.vx_lo, c.vx_hi), vy = jax_uniform(ks[2], 0, c.vy_lo, c.vy_hi), wz = jax_uniform(ks[3], 0, c.wz_lo, c.wz_hi);
  const float bh = jax_uniform(ks[4], 0, c.bh_lo, c.bh_hi), rx = jax_uniform(ks[5], 0, c.rx_lo, c.rx_hi), ry = jax_uniform(ks[6], 0, c.ry_lo, c.ry_hi);
  float arms[10];
  for (uint32_t j = 0; j < 10; ++j) {   // uniform(rng_h, (10,), lo, hi) * bernoulli(rng_h, shape=(10,)): the SAME key, hence the same bits (train.py:734-738)
    const float lo = m.dof_range[16 + j][0], hi = m.dof_range[16 + j][1];
    arms[j] = jax_uniform(ks[7], j, lo, hi) * (jax_u01(ks[7], j) < 0.5f ? 1.0f : 0.0f);
  }
  task_command_template((int)jax_randint(ks[0], 6u), vx, vy, wz, bh, rx, ry, arms, cmd);   // mode = randint(rng_a, (), 0, 6)   (train.py:752)
}
// UnifiedCommand.initial_command (train.py:724-766); one lane
KBJ_DEV void task_sample_command(const kbj_model& m, const kbj_config& c, const Rng& rng, uint32_t a, uint32_t off, float* cmd) {
  if (c.command_mode == 1) { for (int k = 0; k < KBJ_NCMD; ++k) cmd[k] = c.fixed_command[k]; return; }
  if (c.command_mode == 2) { task_sample_command_jax(m, c, jax_call_key(rng, KBJ_RNG_COMMAND, a, off), cmd); return; }
  float vx = rng_uniform(rng, KBJ_RNG_COMMAND, a, off + 2, c.vx_lo, c.vx_hi), vy = rng_uniform(rng, KBJ_RNG_COMMAND, a, off + 3, c.vy_lo, c.vy_hi);
  float wz = rng_uniform(rng, KBJ_RNG_COMMAND, a, off + 4, c.wz_lo, c.wz_hi), bh = rng_uniform(rng, KBJ_RNG_COMMAND, a, off + 5, c.bh_lo, c.bh_hi);
  float rx = rng_uniform(rng, KBJ_RNG_COMMAND, a, off + 6, c.rx_lo, c.rx_hi), ry = rng_uniform(rng, KBJ_RNG_COMMAND, a, off + 7, c.ry_lo, c.ry_hi);
  float arms[10];
  for (int j = 0; j < 10; ++j) {
    // the reference draws uniform and bernoulli from one key (train.py:734-737): the same u decides both
    float u = rng_u01(rng, KBJ_RNG_COMMAND, a, off + 8 + j);
    float lo = m.dof_range[16 + j][0], hi = m.dof_range[16 + j][1];
    arms[j] = u < 0.5f ? fmaf(hi - lo, u, lo) : 0.0f;
  }
  uint32_t b0, b1;
  rng_bits(rng, KBJ_RNG_COMMAND, a, off + 1, b0, b1);
  task_command_template((int)(b0 % 6u), vx, vy, wz, bh, rx, ry, arms, cmd);
}

// PositionActuators (train.py:1097-1105): tau = kp (a + bias - q) - kd qdot, clipped to the randomised soft limit
KBJ_DEV void task_pd(KbjShared& S, const float* action) {
  const float* ep = S.ep;
  PFOR(u, NU) {
    float t = ep[KBJ_EP_KP + u] * (action[u] + ep[KBJ_EP_ACTBIAS + u] - S.es[KBJ_ES_QPOS + 7 + u]) - ep[KBJ_EP_KD + u] * S.es[KBJ_ES_QVEL + 6 + u];
    float lim = ep[KBJ_EP_TAULIM + u];
    S.ctrl[u] = fminf(fmaxf(t, -lim), lim);
  }
  KBJ_SYNC();
}

// resets (train.py:1146-1153, 833-844) + per-episode re-initialisation, then one forward pass for the first observation
KBJ_DEV void task_reset(KbjShared& S, const kbj_model& m, const kbj_config& c, const PhysConst& pc, const Rng& rng) {
  float* es = S.es;
  PFOR(w, 1) es[KBJ_ES_EPISODE] = u2f(f2u(es[KBJ_ES_EPISODE]) + 1u);
  KBJ_SYNC();
  task_randomize(S, m, c, rng);
  uint32_t e = f2u(es[KBJ_ES_EPISODE]);
  PFOR(u, NU) {
    es[KBJ_ES_QPOS + 7 + u] = m.joint_bias[u] + rng_uniform(rng, KBJ_RNG_RESET, e, u, -c.reset_joint_pos_scale, c.reset_joint_pos_scale);
    es[KBJ_ES_QVEL + 6 + u] = rng_uniform(rng, KBJ_RNG_RESET, e, 20 + u, -c.reset_joint_vel_scale, c.reset_joint_vel_scale);
    es[KBJ_ES_ACT_PREV + u] = m.joint_bias[u];
  }
  PFOR(i, NV) { es[KBJ_ES_WARM + i] = 0; if (i < 6) es[KBJ_ES_QVEL + i] = 0; }
  KBJ_SYNC();
  PFOR(w, 1) {
    es[KBJ_ES_QVEL + 0] = rng_uniform(rng, KBJ_RNG_RESET, e, 40, -c.reset_base_vel_xy_scale, c.reset_base_vel_xy_scale);
    es[KBJ_ES_QVEL + 1] = rng_uniform(rng, KBJ_RNG_RESET, e, 41, -c.reset_base_vel_xy_scale, c.reset_base_vel_xy_scale);
    float yaw = rng_uniform(rng, KBJ_RNG_RESET, e, 42, -3.14159265358979323846f, 3.14159265358979323846f);
    es[KBJ_ES_QPOS + 3] = cosf(yaw / 2); es[KBJ_ES_QPOS + 4] = 0; es[KBJ_ES_QPOS + 5] = 0; es[KBJ_ES_QPOS + 6] = sinf(yaw / 2);
    es[KBJ_ES_QPOS + 0] = rng_uniform(rng, KBJ_RNG_RESET, e, 43, -c.reset_xy_range, c.reset_xy_range);
    es[KBJ_ES_QPOS + 1] = rng_uniform(rng, KBJ_RNG_RESET, e, 44, -c.reset_xy_range, c.reset_xy_range);
    if (c.command_mode == 2) {   // PlaneXYPositionReset with jax.random's key handling: keyx, keyy = split(rng); uniform(key, (1,), -r, r)   (train.py:834-836)
      const JaxKey k = jax_call_key(rng, KBJ_RNG_RESET, e, 43);
      es[KBJ_ES_QPOS + 0] = jax_uniform(jax_split(k, 0), 0, -c.reset_xy_range, c.reset_xy_range);
      es[KBJ_ES_QPOS + 1] = jax_uniform(jax_split(k, 1), 0, -c.reset_xy_range, c.reset_xy_range);
    }
    es[KBJ_ES_QPOS + 2] = m.qpos0[2];
    if (pc.tamp != 0) {  // stand on the highest of five terrain samples under the robot (centre, +-0.15 m in x and y)
      const float sx[5] = {0, 0.15f, -0.15f, 0, 0}, sy[5] = {0, 0, 0, 0.15f, -0.15f};
      float hmax = 0, nn[3];
      for (int k = 0; k < 5; ++k) { float h; terrain_eval(pc, es[KBJ_ES_QPOS + 0] + sx[k], es[KBJ_ES_QPOS + 1] + sy[k], h, nn); hmax = k == 0 ? h : fmaxf(hmax, h); }
      es[KBJ_ES_QPOS + 2] = m.qpos0[2] + hmax;
    }
    for (int k = 0; k < 6; ++k) es[KBJ_ES_PUSH + k] = 0;
    es[KBJ_ES_PUSH_REM] = 0;
    es[KBJ_ES_PUSH_NXT] = floorf(rng_uniform(rng, KBJ_RNG_RANDOMIZE, e, 310, c.push_int_lo, c.push_int_hi) / c.ctrl_dt);
    es[KBJ_ES_TIME] = 0;
    task_sample_command(m, c, rng, f2u(es[KBJ_ES_STEP]), 32, es + KBJ_ES_CMD);
    S.pushing = 0;
  }
  KBJ_SYNC();
  task_pd(S, es + KBJ_ES_ACT_PREV);
  phys_forward(S, S.mc, pc);
  PFOR(k, 3) es[KBJ_ES_PGLAG + k] = S.pg[k];
  KBJ_SYNC();
}

// COMDistanceObservation (train.py:509-659): monotone-chain hull of the 8 contact slots, shoelace centroid with the
// mean-point fallback, distance to subtree_com[2].xy. One lane.
KBJ_DEV float task_com_distance(const KbjShared& S) {
  int order[NCON];
  for (int i = 0; i < NCON; ++i) order[i] = i;
  for (int i = 1; i < NCON; ++i) {  // stable insertion sort, lexicographic (x, y)
    int v = order[i], j = i - 1;
    while (j >= 0) {
      int o = order[j];
      bool less = S.conpos[v][0] != S.conpos[o][0] ? S.conpos[v][0] < S.conpos[o][0] : S.conpos[v][1] < S.conpos[o][1];
      if (!less) break;
      order[j + 1] = o; --j;
    }
    order[j + 1] = v;
  }
  float sp[NCON][2];
  for (int i = 0; i < NCON; ++i) { sp[i][0] = S.conpos[order[i]][0]; sp[i][1] = S.conpos[order[i]][1]; }
  float poly[2 * NCON][2];
  int cnt = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int stack[NCON], ptr = 0;
    for (int k = 0; k < NCON; ++k) {
      int idx = pass ? NCON - 1 - k : k;
      while (ptr >= 2) {
        int a = stack[ptr - 2], b = stack[ptr - 1];
        float cr = (sp[b][0] - sp[a][0]) * (sp[idx][1] - sp[a][1]) - (sp[b][1] - sp[a][1]) * (sp[idx][0] - sp[a][0]);
        if (cr <= 0) --ptr; else break;
      }
      stack[ptr++] = idx;
    }
    for (int i = 0; i < ptr - 1; ++i) { poly[cnt][0] = sp[stack[i]][0]; poly[cnt][1] = sp[stack[i]][1]; ++cnt; }
  }
  float area = 0, sx = 0, sy = 0, mx = 0, my = 0;
  for (int i = 0; i < cnt; ++i) {
    int j = (i + 1 < cnt) ? i + 1 : 0;
    float cr = poly[i][0] * poly[j][1] - poly[j][0] * poly[i][1];
    area += cr; sx += (poly[i][0] + poly[j][0]) * cr; sy += (poly[i][1] + poly[j][1]) * cr;
    mx += poly[i][0]; my += poly[i][1];
  }
  area *= 0.5f;
  float cx, cy;
  if (fabsf(area) < 1e-12f) { float n = (float)(cnt > 1 ? cnt : 1); cx = mx / n; cy = my / n; }
  else { cx = sx / (6 * area); cy = sy / (6 * area); }
  return sqrtf((cx - S.com2[0]) * (cx - S.com2[0]) + (cy - S.com2[1]) * (cy - S.com2[1]));
}

KBJ_DEV void encode_pg(const float* g, float* o) {  // train.py:1338-1349
  float n = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  o[0] = atan2f(g[1], -g[2]); o[1] = atan2f(-g[0], sqrtf(g[1] * g[1] + g[2] * g[2]));
  o[2] = g[0] / n; o[3] = g[1] / n; o[4] = g[2] / n;
}

// observation packing (train.py:1329-1433) from the derived data of the last forward pass; rows live in HBM
KBJ_DEV void task_write_obs(KbjShared& S, const kbj_model& m, const kbj_config& c, const Rng& rng, float* actor, float* critic, float* aux) {
  float* es = S.es;
  uint32_t st = f2u(es[KBJ_ES_STEP]);
  bool noise = c.enable_noise != 0;
  const float* cmd = es + KBJ_ES_CMD;
  float zc = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]) < 1e-3f ? 1.0f : 0.0f;
  PFOR(u, NU) {
    float range = fmaxf(m.joint_bias[u] - m.joint_lo[u], m.joint_hi[u] - m.joint_bias[u]);
    float q = es[KBJ_ES_QPOS + 7 + u], v = es[KBJ_ES_QVEL + 6 + u];
    float qn = q + S.ep[KBJ_EP_JPBIAS + u] + (noise ? rng_uniform(rng, KBJ_RNG_OBS_NOISE, st, u, -c.jpos_noise, c.jpos_noise) : 0.0f);
    float vn = v + (noise ? rng_uniform(rng, KBJ_RNG_OBS_NOISE, st, 20 + u, -c.jvel_noise, c.jvel_noise) : 0.0f);
    actor[KBJ_OBS_JPOS + u] = (qn - m.joint_bias[u]) / range; actor[KBJ_OBS_JVEL + u] = vn / KBJ_OBS_JVEL_DIV;
    critic[KBJ_OBS_JPOS + u] = (q - m.joint_bias[u]) / range; critic[KBJ_OBS_JVEL + u] = v / KBJ_OBS_JVEL_DIV;
    critic[KBJ_OBS_ACTFRC + u] = S.qfrc_act[6 + u] / KBJ_OBS_ACTFRC_DIV;
  }
  // The four one-lane jobs below are different code, so the wavefront runs them one after the other: keep each short. The six normal
  // draws of the imu noise (Box-Muller: threefry + log + cos + sqrt, ~150 vector instructions each) run on six lanes at once.
  float* nrm6 = S.u.cfrc[0];     // scratch: the RNE buffers are dead once the solver has run
  PFOR(w, 6) nrm6[w] = noise ? rng_normal(rng, KBJ_RNG_OBS_NOISE, st, 40 + w) : 0.0f;    // 40..42 gyro, 43..45 projected gravity
  KBJ_SYNC();
  PFOR(w, 4) {
    if (w == 0) {  // lagged / biased / noisy projected gravity for the actor, clean one for the critic
      float lag = S.ep[KBJ_EP_PGLAG], pgn[3], o[5];
      for (int k = 0; k < 3; ++k) {
        float pgl = lag * es[KBJ_ES_PGLAG + k] + (1 - lag) * S.pg[k];
        es[KBJ_ES_PGLAG + k] = pgl;
        pgn[k] = pgl + S.ep[KBJ_EP_PGBIAS + k] + (noise ? c.pg_noise_std * nrm6[3 + k] : 0.0f);
      }
      encode_pg(pgn, o);
      for (int k = 0; k < 5; ++k) actor[KBJ_OBS_PG + k] = o[k];
      for (int k = 0; k < 3; ++k) {
        actor[KBJ_OBS_GYRO + k] = S.gyro[k] + (noise ? c.gyro_noise_std * nrm6[k] : 0.0f);
        critic[KBJ_OBS_GYRO + k] = S.gyro[k];
      }
    } else if (w == 1) {
      float o[5];
      encode_pg(S.pg, o);
      for (int k = 0; k < 5; ++k) critic[KBJ_OBS_PG + k] = o[k];
      actor[KBJ_OBS_ZEROCMD] = zc; critic[KBJ_OBS_ZEROCMD] = zc;
      // behind the reference's columns: user observation slots (the host fills them after the step) and the pad to the 16-byte row stride
      for (int k = KBJ_NOBS_ACTOR; k < KBJ_LD_OF(KBJ_NOBS_ACTOR + c.extra_obs_actor); ++k) actor[k] = 0;
      for (int k = KBJ_NOBS_CRITIC; k < KBJ_LD_OF(KBJ_NOBS_CRITIC + c.extra_obs_critic); ++k) critic[k] = 0;
      critic[KBJ_OBS_TOUCH] = S.touch[0]; critic[KBJ_OBS_TOUCH + 1] = S.touch[1];
      aux[KBJ_AUX_TOUCH] = S.touch[0]; aux[KBJ_AUX_TOUCH + 1] = S.touch[1];
      for (int k = 0; k < 3; ++k) { critic[KBJ_OBS_BASEPOS + k] = es[KBJ_ES_QPOS + k]; critic[KBJ_OBS_LINVEL + k] = es[KBJ_ES_QVEL + k]; critic[KBJ_OBS_ANGVEL + k] = es[KBJ_ES_QVEL + 3 + k]; }
      for (int k = 0; k < 4; ++k) critic[KBJ_OBS_BASEQUAT + k] = es[KBJ_ES_QPOS + 3 + k];
      critic[KBJ_OBS_HEIGHT] = S.xpos[1][2];  // BaseHeightObservation (train.py:706-707)
    } else if (w == 2) {  // FeetPositionObservation (train.py:682-699): foot offsets in the base's yaw frame
      // the reference goes quat -> euler -> (0, 0, yaw) -> quat -> rotate; the yaw rotation is cos / sin of atan2(sn, cs), i.e. (cs, sn)
      // normalised - no trigonometry needed
      const float* q = S.xquat[1];
      const float sn = 2 * (q[0] * q[3] + q[1] * q[2]), cs = 1 - 2 * (q[2] * q[2] + q[3] * q[3]);
      const float h2 = sn * sn + cs * cs, inv = h2 > 0 ? kbj_rsqrt(h2) : 0.0f;
      const float cy = h2 > 0 ? cs * inv : 1.0f, sy = sn * inv;
      for (int f = 0; f < 2; ++f) {
        int fb = f ? 12 : 7;
        float rel[3] = {S.xpos[fb][0] - S.xpos[1][0], S.xpos[fb][1] - S.xpos[1][1], S.xpos[fb][2] - S.xpos[1][2]};
        critic[KBJ_OBS_FEETPOS + 3 * f + 0] = cy * rel[0] + sy * rel[1];
        critic[KBJ_OBS_FEETPOS + 3 * f + 1] = cy * rel[1] - sy * rel[0];
        critic[KBJ_OBS_FEETPOS + 3 * f + 2] = rel[2];
      }
    } else aux[KBJ_AUX_COMDIST] = task_com_distance(S);
  }
  PFOR(k, KBJ_NCMD) { actor[KBJ_OBS_CMD + k] = cmd[k]; critic[KBJ_OBS_CMD + k] = cmd[k]; aux[KBJ_AUX_CMD + k] = cmd[k]; }
  PFOR(k, 230) critic[KBJ_OBS_CINERT + k] = S.cinert[1 + k / 10][k % 10];
  PFOR(k, 138) critic[KBJ_OBS_CVEL + k] = S.cvel[1 + k / 6][k % 6];
  KBJ_SYNC();
}

// one control step of one env (mirrors ksim's engine + rollout bookkeeping):
// latency/drop -> push event -> substeps x (PD, forward, integrate) -> termination -> reward inputs ->
// reset or command switch -> next observation.
KBJ_DEV void task_step(KbjShared& S, const kbj_model& m, const kbj_config& c, const PhysConst& pc, const Rng& rng, const float* action,
                       float* aux_t, float* actor_next, float* critic_next, float* aux_next, float* qstate = nullptr) {
  float* es = S.es;
  uint32_t st = f2u(es[KBJ_ES_STEP]);
  bool drop = rng_u01(rng, KBJ_RNG_DROP, st, 0) < c.drop_action_prob;
  PFOR(u, NU) S.act_eff[u] = drop ? es[KBJ_ES_ACT_PREV + u] : action[u];
  PFOR(w, 1) {  // ForcePushEvent (train.py:1134-1144)
    int pushing = 0;
    if (c.enable_pushes) {
      if (es[KBJ_ES_PUSH_REM] > 0) { es[KBJ_ES_PUSH_REM] -= 1; pushing = 1; }
      else if (es[KBJ_ES_PUSH_NXT] <= 0) {
        for (int k = 0; k < 3; ++k) {
          es[KBJ_ES_PUSH + k] = rng_uniform(rng, KBJ_RNG_PUSH, st, k, -c.push_max_force, c.push_max_force);
          es[KBJ_ES_PUSH + 3 + k] = rng_uniform(rng, KBJ_RNG_PUSH, st, 3 + k, -c.push_max_torque, c.push_max_torque);
        }
        es[KBJ_ES_PUSH_REM] = floorf(rng_uniform(rng, KBJ_RNG_PUSH, st, 6, c.push_dur_lo, c.push_dur_hi) / c.ctrl_dt);
        es[KBJ_ES_PUSH_NXT] = floorf(rng_uniform(rng, KBJ_RNG_PUSH, st, 7, c.push_int_lo, c.push_int_hi) / c.ctrl_dt);
        pushing = 1;
      } else es[KBJ_ES_PUSH_NXT] -= 1;
    }
    S.pushing = pushing;
    for (int k = 0; k < 6; ++k) S.push[k] = es[KBJ_ES_PUSH + k];
  }
  KBJ_SYNC();
  int lat = (int)S.ep[KBJ_EP_LATENCY];
  for (int s = 0; s < c.substeps; ++s) {
    if (qstate && s == c.substeps - 1) PFOR(k, KBJ_NQ) qstate[KBJ_QSTATE_QPOS_KIN + k] = es[KBJ_ES_QPOS + k];   // what the last forward pass's kinematics run on
    task_pd(S, s >= lat ? S.act_eff : es + KBJ_ES_ACT_PREV);
    phys_forward(S, S.mc, pc, s == c.substeps - 1);
    phys_integrate(S, pc);
    KBJ_STAMP(17);
  }
  PFOR(u, NU) { es[KBJ_ES_ACT_PREV + u] = S.act_eff[u]; aux_t[KBJ_AUX_CTRL + u] = S.ctrl[u]; }
  if (qstate) {   // the state a Trajectory step holds: after the step, before any reset (uniform branch: null unless the caller records)
    PFOR(k, KBJ_NQ) qstate[KBJ_QSTATE_QPOS + k] = es[KBJ_ES_QPOS + k];
    PFOR(k, KBJ_NV) qstate[KBJ_QSTATE_QVEL + k] = es[KBJ_ES_QVEL + k];
  }
  PFOR(w, 1) {
    es[KBJ_ES_TIME] += 1;
    es[KBJ_ES_STEP] = u2f(st + 1u);
    // terminations (train.py:817-823, 1267-1268)
    float height = S.xpos[1][2] - fminf(S.xpos[7][2], S.xpos[12][2]);
    float qx = es[KBJ_ES_QPOS + 4], qy = es[KBJ_ES_QPOS + 5];
    float zz = 1 - 2 * (qx * qx + qy * qy);
    int done = 0;
    if (es[KBJ_ES_TIME] >= (float)c.max_episode_steps) done = 1;
    if (height < c.unhealthy_z || zz < pc.cos_max_tilt) done = -1;
    S.done = done;
    for (int k = 0; k < 6; ++k) aux_t[KBJ_AUX_QVEL + k] = es[KBJ_ES_QVEL + k];
    for (int k = 0; k < 4; ++k) { aux_t[KBJ_AUX_BQUAT + k] = S.xquat[1][k]; aux_t[KBJ_AUX_LFQUAT + k] = S.xquat[7][k]; aux_t[KBJ_AUX_RFQUAT + k] = S.xquat[12][k]; }
    aux_t[KBJ_AUX_BASEZ] = S.xpos[1][2]; aux_t[KBJ_AUX_LFZ] = S.xpos[7][2]; aux_t[KBJ_AUX_RFZ] = S.xpos[12][2];
    for (int j = 0; j < 10; ++j) aux_t[KBJ_AUX_ARMQ + j] = es[KBJ_ES_QPOS + 17 + j];
    aux_t[KBJ_AUX_DONE] = (float)done;
    aux_t[KBJ_AUX_DONE + 1] = 0;
  }
  KBJ_SYNC();
  if (S.done) task_reset(S, m, c, pc, rng);
  else {
    PFOR(w, 1) {  // UnifiedCommand.__call__ (train.py:768-785)
      if (c.command_mode == 0 && rng_u01(rng, KBJ_RNG_COMMAND, st + 1, 0) < c.switch_prob) task_sample_command(m, c, rng, st + 1, 0, es + KBJ_ES_CMD);
      if (c.command_mode == 2) {   // the same with jax.random's key handling: rng_a, rng_b = split(rng); bernoulli(rng_a, switch_prob); initial_command(rng_b)
        const JaxKey k = jax_call_key(rng, KBJ_RNG_COMMAND, st + 1, 0);
        if (jax_u01(jax_split(k, 0), 0) < c.switch_prob) task_sample_command_jax(m, c, jax_split(k, 1), es + KBJ_ES_CMD);
      }
    }
    KBJ_SYNC();
  }
  task_write_obs(S, m, c, rng, actor_next, critic_next, aux_next);
}

}  // namespace kbj
