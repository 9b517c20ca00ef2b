// kbj_env_phys.h — rigid-body forward dynamics + soft-constraint Newton solve for one env per wavefront.
//
// the "MuJoCo-XLA physics step" (reference: mjx.step under ksim's engine, configured at
// train.py:1775-1778). Pipeline per substep: kinematics -> tree com / cinert / cdof -> composite inertia ->
// mass matrix -> velocities + bias forces (RNE) -> actuation -> plane-capsule contacts -> constraint rows ->
// Newton solve with exact line search -> sensors -> semi-implicit Euler.
//
// Data layout: everything lives in KbjShared (LDS). Linear algebra exploits the kbot tree: four 5-dof limb
// chains hang off the 6-dof base, so M and the Newton Hessian H = M + J^T D J are "arrow" matrices; each chain
// is eliminated in its own 11x11 local block (5 chain dofs + 6 base dofs) in parallel, the four Schur
// complements are summed into the 6x6 base block, and the right-hand side rides along as an extra row.
#pragma once
#include "kbj_env_core.h"
#include "kbj_wave.h"
#include <cstddef>
#include <type_traits>

// diagnostics (tools/env_stamps.py, -DKBJ_ENV_STAMPS): shader-clock cycles of env 0 per phase, accumulated in a device array
#if defined(KBJ_ENV_STAMPS) && !defined(KBJ_EMU)
// every 32nd env (256 of 8192) adds its per-phase cycles, so the profile averages over easy and hard envs alike
__device__ unsigned long long kbj_env_stamp_acc[32];
__shared__ unsigned long long kbj_env_stamp_last;
#define KBJ_STAMP(k) do { if ((blockIdx.x & 31) == 0 && threadIdx.x == 0) { unsigned long long t_ = clock64(); if ((k) != 18) atomicAdd(&kbj_env_stamp_acc[k], t_ - kbj_env_stamp_last); kbj_env_stamp_last = clock64(); } } while (0)
#else
#define KBJ_STAMP(k) ((void)0)
#endif

namespace kbj {

// terrain height and unit normal at (x, y)
KBJ_DEV void terrain_eval(const PhysConst& pc, float x, float y, float& h, float n[3]) {
  float sx = sinf(pc.tkw * x), cx = cosf(pc.tkw * x), sy = sinf(pc.tkw * y), cy = cosf(pc.tkw * y);
  h = pc.tamp * sx * sy;
  float hx = pc.tamp * pc.tkw * cx * sy, hy = pc.tamp * pc.tkw * sx * cy;
  float inv = 1 / sqrtf(1 + hx * hx + hy * hy);
  n[0] = -hx * inv; n[1] = -hy * inv; n[2] = inv;
}

// ---- position stage ----------------------------------------------------------------------------------------
// v rotated by the unit quaternion q: v + w t + u x t with t = 2 u x v
KBJ_DEV void quat_rot(const float* q, const float* v, float* o) {
  float t[3] = {2 * (q[2] * v[2] - q[3] * v[1]), 2 * (q[3] * v[0] - q[1] * v[2]), 2 * (q[1] * v[1] - q[2] * v[0])};
  o[0] = v[0] + q[0] * t[0] + (q[2] * t[2] - q[3] * t[1]);
  o[1] = v[1] + q[0] * t[1] + (q[3] * t[0] - q[1] * t[2]);
  o[2] = v[2] + q[0] * t[2] + (q[1] * t[1] - q[2] * t[0]);
}
KBJ_DEV void quat_norm_fast(float* q) {   // one reciprocal square root (v_rsq_f32 + a Newton step) instead of a square root and four divisions
  float inv = kbj_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] *= inv;
}

// Kinematics in three phases: (1) every joint's local rotation body_quat * rot(axis, angle) in parallel (the trigonometry of all
// 20 joints at once), (2) five walkers (four limbs + the imu body) compose them from the torso outwards - a quaternion product, a
// normalisation and one vector rotation per body, (3) everything that hangs off a body's pose (inertial frame origin, world hinge
// axis) in parallel over bodies, see phys_com.
KBJ_DEV void phys_kinematics(KbjShared& S, const KbjModelLds& m) {
  const float* qpos = S.es + KBJ_ES_QPOS;
  PFOR(u, NU) {
    const int b = 3 + u;
    float sn, co;
    kbj_sincos(0.5f * qpos[7 + u], sn, co);
    const float jq[4] = {co, sn * m.jnt_axis[b][0], sn * m.jnt_axis[b][1], sn * m.jnt_axis[b][2]};
    float q[4];
    quat_mul(m.body_quat[b], jq, q);
    for (int k = 0; k < 4; ++k) S.xquat[b][k] = q[k];   // staged: the walker of this limb replaces it by the world orientation
  }
  KBJ_SYNC();
  PFOR(c, 5) {
    // base and torso (welded to the base), recomputed by every walker instead of a phase of their own
    float pq[4] = {qpos[3], qpos[4], qpos[5], qpos[6]}, pp[3] = {qpos[0], qpos[1], qpos[2]}, t[3], q[4];
    quat_norm_fast(pq);
    if (c == 4) {
      for (int k = 0; k < 3; ++k) { S.xpos[0][k] = 0; S.xpos[1][k] = pp[k]; }
      S.xquat[0][0] = 1; S.xquat[0][1] = S.xquat[0][2] = S.xquat[0][3] = 0;
      for (int k = 0; k < 4; ++k) S.xquat[1][k] = pq[k];
    }
    quat_rot(pq, m.body_pos[2], t);
    for (int k = 0; k < 3; ++k) pp[k] += t[k];
    quat_mul(pq, m.body_quat[2], q);
    quat_norm_fast(q);
    for (int k = 0; k < 4; ++k) pq[k] = q[k];
    if (c == 4) {
      for (int k = 0; k < 3; ++k) S.xpos[2][k] = pp[k];
      for (int k = 0; k < 4; ++k) S.xquat[2][k] = pq[k];
    }
    const int nb = c < 4 ? 5 : 1;
    for (int k5 = 0; k5 < nb; ++k5) {
      const int b = c < 4 ? 3 + 5 * c + k5 : 23;
      quat_rot(pq, m.body_pos[b], t);
      for (int k = 0; k < 3; ++k) pp[k] += t[k];
      const float* ql = c < 4 ? S.xquat[b] : m.body_quat[b];
      const float qloc[4] = {ql[0], ql[1], ql[2], ql[3]};
      quat_mul(pq, qloc, q);
      quat_norm_fast(q);
      for (int k = 0; k < 4; ++k) pq[k] = q[k];
      for (int k = 0; k < 3; ++k) S.xpos[b][k] = pp[k];
      for (int k = 0; k < 4; ++k) S.xquat[b][k] = pq[k];
    }
  }
  KBJ_SYNC();
}

// centre of mass, body inertias about it, motion axes (cdof) - parallel over bodies / dofs, the two centre-of-mass sums as wave sums
KBJ_DEV void phys_com(KbjShared& S, const KbjModelLds& m) {
  const float* mass = S.ep + KBJ_EP_MASS;
  PFOR(b, NB) {
    float mat[9], t[3] = {0, 0, 0};
    if (b > 0) { quat_to_mat(S.xquat[b], mat); mat_vec(mat, S.ep + KBJ_EP_IPOS + 3 * b, t); }
    for (int k = 0; k < 3; ++k) S.xipos[b][k] = S.xpos[b][k] + t[k];
  }
  KBJ_SYNC();
  {
    float s2[3], m2;
    for (int w = 0; w < 3; ++w) s2[w] = wsum(NB, [&](int l) { return l >= 2 ? mass[l] * S.xipos[l][w] : 0.0f; });
    m2 = wsum(NB, [&](int l) { return l >= 2 ? mass[l] : 0.0f; });
    const float m1 = m2 + mass[1];
    KBJ_SYNC();   // (emulation: the sums above read what the stores below overwrite nothing of; on the GPU a no-op for one wavefront)
    PFOR(w, 3) {
      S.com2[w] = s2[w] * kbj_frcp(m2);
      S.com[w] = (s2[w] + mass[1] * S.xipos[1][w]) * kbj_frcp(m1);
      if (w == 0) S.com[3] = m1;
    }
  }
  KBJ_SYNC();
  PFOR(b, NB) {
    float* c = S.cinert[b];
    if (b == 0) { for (int k = 0; k < 10; ++k) c[k] = 0; }
    else {
      float mat[9];
      quat_to_mat(S.xquat[b], mat);
      const float* in = S.ep + KBJ_EP_INERTIA + 3 * b;
      float dif[3] = {S.xipos[b][0] - S.com[0], S.xipos[b][1] - S.com[1], S.xipos[b][2] - S.com[2]};
      float ms = mass[b];
      c[0] = mat[0] * mat[0] * in[0] + mat[1] * mat[1] * in[1] + mat[2] * mat[2] * in[2];
      c[1] = mat[3] * mat[3] * in[0] + mat[4] * mat[4] * in[1] + mat[5] * mat[5] * in[2];
      c[2] = mat[6] * mat[6] * in[0] + mat[7] * mat[7] * in[1] + mat[8] * mat[8] * in[2];
      c[3] = mat[0] * mat[3] * in[0] + mat[1] * mat[4] * in[1] + mat[2] * mat[5] * in[2];
      c[4] = mat[0] * mat[6] * in[0] + mat[1] * mat[7] * in[1] + mat[2] * mat[8] * in[2];
      c[5] = mat[3] * mat[6] * in[0] + mat[4] * mat[7] * in[1] + mat[5] * mat[8] * in[2];
      c[0] += ms * (dif[1] * dif[1] + dif[2] * dif[2]);
      c[1] += ms * (dif[0] * dif[0] + dif[2] * dif[2]);
      c[2] += ms * (dif[0] * dif[0] + dif[1] * dif[1]);
      c[3] -= ms * dif[0] * dif[1];
      c[4] -= ms * dif[0] * dif[2];
      c[5] -= ms * dif[1] * dif[2];
      c[6] = ms * dif[0]; c[7] = ms * dif[1]; c[8] = ms * dif[2]; c[9] = ms;
    }
  }
  PFOR(d, NV) {
    float* cd = S.cdof[d];
    int b = dof_body(d);
    float off[3] = {S.com[0] - S.xpos[b][0], S.com[1] - S.xpos[b][1], S.com[2] - S.xpos[b][2]};
    if (d < 3) { for (int k = 0; k < 6; ++k) cd[k] = 0; cd[3 + d] = 1; }
    else {
      float mat[9], ax[3];
      quat_to_mat(S.xquat[b], mat);
      if (d < 6) { int i = d - 3; ax[0] = mat[i]; ax[1] = mat[3 + i]; ax[2] = mat[6 + i]; }
      else mat_vec(mat, m.jnt_axis[b], ax);   // world hinge axis = angular part of the motion axis
      for (int k = 0; k < 3; ++k) cd[k] = ax[k];
      cross3(ax, off, cd + 3);
    }
  }
  KBJ_SYNC();
}

// sum of a per-body quantity over the subtree of body b (fixed kbot tree)
template <int W> KBJ_DEV float subtree_sum(const float (*q)[W], int b, int k) {
  float s = 0;
  if (b == 1 || b == 2) { for (int bb = b; bb < NB; ++bb) s += q[bb][k]; }
  else if (b == 23) s = q[23][k];
  else if (b >= 3) { int last = 3 + 5 * ((b - 3) / 5) + 4; for (int bb = b; bb <= last; ++bb) s += q[bb][k]; }
  return s;
}

// the stored entries of the tree-sparse mass matrix: (dof i, ancestor-or-self dof j, byte offset of the entry from S.Mb)
struct MPair { unsigned char i, j; unsigned short off; };
constexpr int MPAIR_N = 21 + 20 * 6 + 4 * 15;
struct MPairTab { MPair e[MPAIR_N]; };
constexpr MPairTab make_mpair_tab() {
  MPairTab t{};
  int n = 0;
  for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) t.e[n++] = MPair{(unsigned char)i, (unsigned char)j, (unsigned short)(4 * (i * 6 + j))};
  for (int c = 0; c < 4; ++c)
    for (int a = 0; a < 5; ++a) {
      const int i = 6 + 5 * c + a, row = 36 + (c * 5 + a) * 11;
      for (int j = 0; j < 6; ++j) t.e[n++] = MPair{(unsigned char)i, (unsigned char)j, (unsigned short)(4 * (row + j))};
      for (int a2 = 0; a2 <= a; ++a2) t.e[n++] = MPair{(unsigned char)i, (unsigned char)(6 + 5 * c + a2), (unsigned short)(4 * (row + 6 + a2))};
    }
  return t;
}
#ifdef KBJ_EMU
static const MPairTab MPAIR = make_mpair_tab();
#else
__device__ const MPairTab MPAIR = make_mpair_tab();
#endif

KBJ_DEV void phys_crb_mass(KbjShared& S) {
  // composite inertias: suffix sums from the tip of each limb towards the torso (one lane per limb and component), then torso and base
  PFOR(w, 50) {
    const int c = w / 10, k = w % 10;
    if (c < 4) {
      float acc = 0;
      for (int k5 = 4; k5 >= 0; --k5) { const int b = 3 + 5 * c + k5; acc += S.cinert[b][k]; S.u.crb[b][k] = acc; }
    } else S.u.crb[23][k] = S.cinert[23][k];
  }
  KBJ_SYNC();
  PFOR(k, 10) {
    const float t = S.cinert[2][k] + (((S.u.crb[3][k] + S.u.crb[8][k]) + (S.u.crb[13][k] + S.u.crb[18][k])) + S.u.crb[23][k]);
    S.u.crb[2][k] = t; S.u.crb[1][k] = t + S.cinert[1][k]; S.u.crb[0][k] = 0;
  }
  KBJ_SYNC();
  // M[i][j] = cdof_j . (crb[body_i] cdof_i) for j = i and its ancestors (the six base dofs, then the limb's dofs up to i): first the 26
  // spatial forces crb cdof_i (one lane per dof), then one lane per STORED entry (201 of them: 4 rounds of 64 instead of one lane per row
  // walking up to 11 dot products, with the base and the limb rows diverging). The forces sit in the constraint-row arrays D / aref / force,
  // which are dead from the end of a solve (and its sensors) to the next phys_make_constraints.
  float* fbuf = S.D;
  static_assert(sizeof(S.D) + sizeof(S.aref) + sizeof(S.force) >= NV * 6 * sizeof(float) && offsetof(KbjShared, aref) == offsetof(KbjShared, D) + sizeof(S.D) &&
                offsetof(KbjShared, force) == offsetof(KbjShared, aref) + sizeof(S.aref), "D, aref, force are used as one scratch array here");
  PFOR(i, NV) {
    float buf[6];
    inert_mul(S.u.crb[dof_body(i)], S.cdof[i], buf);
    for (int k = 0; k < 6; ++k) fbuf[6 * i + k] = buf[k];
  }
  KBJ_SYNC();
  PFOR(w, MPAIR_N) {
    const MPair e = MPAIR.e[w];
    const float* cj = S.cdof[e.j];
    const float* bi = fbuf + 6 * e.i;
    float x = 0;
    for (int k = 0; k < 6; ++k) x += cj[k] * bi[k];
    if (e.i == e.j) x += S.ep[KBJ_EP_ARMATURE + e.i];
    *reinterpret_cast<float*>(reinterpret_cast<char*>(&S.Mb[0][0]) + e.off) = x;
  }
  KBJ_SYNC();
}

// ---- contacts + velocity stage -------------------------------------------------------------------------------
KBJ_DEV void phys_collide_vel(KbjShared& S, const KbjModelLds& m, const PhysConst& pc) {
  const float* qvel = S.es + KBJ_ES_QVEL;
  PFOR(ci, NCON) {  // capsule end ci%2 of capsule ci/2 against the plane z = 0
    int c = ci / 2, b = c < 2 ? 7 : 12;
    float t[3], ax[3], mat[9];
    quat_to_mat(S.xquat[b], mat);
    mat_vec(mat, S.ep + KBJ_EP_CAP_POS + 3 * c, t);
    mat_vec(mat, m.cap_axis[c], ax);
    float sgn = (ci & 1) ? 1.0f : -1.0f, hl = S.ep[KBJ_EP_CAP_HALF + c], rad = S.ep[KBJ_EP_CAP_RAD + c];
    float end[3];
    for (int k = 0; k < 3; ++k) end[k] = S.xpos[b][k] + t[k] + sgn * hl * ax[k];
    if (pc.tamp == 0) {
      float dist = end[2] - rad;
      S.condist[ci] = dist;
      S.conpos[ci][0] = end[0]; S.conpos[ci][1] = end[1]; S.conpos[ci][2] = end[2] - (rad + dist / 2);
      S.connrm[ci][0] = 0; S.connrm[ci][1] = 0; S.connrm[ci][2] = 1;
      S.conact[ci] = dist < 0;
    } else {  // sphere (capsule end) against the tangent plane of the sine surface below its centre
      float h, n[3];
      terrain_eval(pc, end[0], end[1], h, n);
      float dist = (end[2] - h) * n[2] - rad;
      S.condist[ci] = dist;
      for (int k = 0; k < 3; ++k) { S.conpos[ci][k] = end[k] - n[k] * (rad + dist / 2); S.connrm[ci][k] = n[k]; }
      S.conact[ci] = dist < 0;
    }
  }
  // RNE forward pass: spatial velocity and bias acceleration of every body. The base part (three translations, then three rotations that
  // share the pre-rotation velocity) is computed by every lane; then one lane per hinge dof builds its body's velocity from the base's by
  // adding the limb's dofs up to its own IN THE WALKER'S ORDER (base .. tip: same sums, bit for bit, as walking the limb), and the
  // velocity-product axis cdof_dot = v_parent x cdof; a last phase adds the limb's cdof_dot qvel terms per (body, component). The serial
  // walk per limb (five bodies x ~35 instructions on 5 lanes) becomes two short data-parallel phases.
  {
    float v[6] = {0, 0, 0, 0, 0, 0}, a[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
    float cdd[6];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 6; ++k) v[k] += S.cdof[i][k] * qvel[i];
    float vb[6];
    for (int k = 0; k < 6; ++k) vb[k] = v[k];
    for (int i = 3; i < 6; ++i) {
      cross_motion(vb, S.cdof[i], cdd);
      for (int k = 0; k < 6; ++k) { v[k] += S.cdof[i][k] * qvel[i]; a[k] += cdd[k] * qvel[i]; }
    }
    PFOR(w, 1) {   // world body at rest; base, torso and imu move together
      for (int k = 0; k < 6; ++k) {
        S.cvel[0][k] = 0; S.u.cfrc_acc[0][k] = 0;
        S.cvel[1][k] = v[k]; S.u.cfrc_acc[1][k] = a[k]; S.cvel[2][k] = v[k]; S.u.cfrc_acc[2][k] = a[k]; S.cvel[23][k] = v[k]; S.u.cfrc_acc[23][k] = a[k];
      }
    }
    PFOR(u, NU) {
      const int c = u / 5, pos = u % 5, d = 6 + u, b = 3 + u;
      float vl[6], cd[6];
      for (int k = 0; k < 6; ++k) vl[k] = v[k];
      for (int e = 0; e < pos; ++e) { const int de = 6 + 5 * c + e; for (int k = 0; k < 6; ++k) vl[k] += S.cdof[de][k] * qvel[de]; }
      cross_motion(vl, S.cdof[d], cd);
      for (int k = 0; k < 6; ++k) { S.u.cfrc[b][k] = cd[k]; vl[k] += S.cdof[d][k] * qvel[d]; S.cvel[b][k] = vl[k]; }   // cfrc: cdof_dot until the body forces overwrite it
    }
  }
  KBJ_SYNC();
  PFOR(w, NU * 6) {
    const int u = w / 6, k = w % 6, c = u / 5, pos = u % 5;
    float a = S.u.cfrc_acc[1][k];
    for (int e = 0; e <= pos; ++e) { const int de = 6 + 5 * c + e; a += S.u.cfrc[3 + 5 * c + e][k] * qvel[de]; }
    S.u.cfrc_acc[3 + u][k] = a;
  }
  KBJ_SYNC();
  PFOR(b, NB) {
    float v[6], a[6], Ia[6], Iv[6], x[6];
    for (int k = 0; k < 6; ++k) { v[k] = S.cvel[b][k]; a[k] = S.u.cfrc_acc[b][k]; }
    inert_mul(S.cinert[b], a, Ia); inert_mul(S.cinert[b], v, Iv); cross_force(v, Iv, x);
    for (int k = 0; k < 6; ++k) S.u.cfrc[b][k] = b ? Ia[k] + x[k] : 0.0f;
  }
  KBJ_SYNC();
  // subtree sums of the body forces: suffix sums along the limbs, then torso and base
  PFOR(w, 30) {
    const int c = w / 6, k = w % 6;
    if (c < 4) {
      float acc = 0;
      for (int k5 = 4; k5 >= 0; --k5) { const int b = 3 + 5 * c + k5; acc += S.u.cfrc[b][k]; S.u.cfrc_acc[b][k] = acc; }
    } else S.u.cfrc_acc[23][k] = S.u.cfrc[23][k];
  }
  KBJ_SYNC();
  PFOR(k, 6) {
    const float t = S.u.cfrc[2][k] + (((S.u.cfrc_acc[3][k] + S.u.cfrc_acc[8][k]) + (S.u.cfrc_acc[13][k] + S.u.cfrc_acc[18][k])) + S.u.cfrc_acc[23][k]);
    S.u.cfrc_acc[2][k] = t; S.u.cfrc_acc[1][k] = t + S.u.cfrc[1][k]; S.u.cfrc_acc[0][k] = 0;
  }
  KBJ_SYNC();
}

KBJ_DEV void phys_smooth_forces(KbjShared& S, const KbjModelLds& m) {
  PFOR(i, NV) {
    float s = 0;
    for (int k = 0; k < 6; ++k) s += S.cdof[i][k] * S.u.cfrc_acc[dof_body(i)][k];
    float act = 0, app = 0;
    if (i >= 6) act = fminf(fmaxf(S.ctrl[i - 6], m.act_range[i - 6][0]), m.act_range[i - 6][1]);
    else if (S.pushing) {
      if (i < 3) app = S.push[i];
      else {
        float arm[3] = {S.xipos[1][0] - S.xpos[1][0], S.xipos[1][1] - S.xpos[1][1], S.xipos[1][2] - S.xpos[1][2]}, t[3], tq[3], loc[3], mat[9];
        cross3(arm, S.push, t);
        for (int k = 0; k < 3; ++k) tq[k] = S.push[3 + k] + t[k];
        quat_to_mat(S.xquat[1], mat);
        matT_vec(mat, tq, loc);
        app = loc[i - 3];
      }
    }
    S.qfrc_act[i] = act;
    S.qfrc_smooth[i] = act + app - s;
  }
  KBJ_SYNC();
}

// ---- arrow-matrix LDL^T with the right-hand side carried as an extra row -----------------------------------------
// G = M (+ J^T D J over rows in their quadratic zone when `hess`); solves G x = rhs, result in S.vec.
// local index li of chain c: li 0..4 <-> dof 10+5c-li (ankle first), li 5..10 <-> base dof li-5.
// Overwrites the union S.u (crb / cfrc are dead by the time a solve runs).
// The 77 stored entries (i, j) of one augmented 12 x 11 lower-triangular block, sorted by column j DESCENDING: the trailing
// sub-block a pivot p updates (j > p) is then the contiguous prefix of n_p = TRI_N[p] entries, so no lane idles.
struct TriTab { unsigned char i[77], j[77]; };
constexpr TriTab make_tri() {
  TriTab t{};
  int n = 0;
  for (int j = 10; j >= 0; --j) for (int i = j; i <= 11; ++i) { t.i[n] = (unsigned char)i; t.j[n] = (unsigned char)j; ++n; }
  return t;
}
#ifdef KBJ_EMU
static const TriTab TRI = make_tri();
#else
__device__ const TriTab TRI = make_tri();
#endif
KBJ_DEV int tri_count(int p) { return 65 - p * (23 - p) / 2; }  // entries with j > p: 65, 54, 44, 35, 27 for p = 0..4

#ifdef KBJ_EMU
#define KBJ_RCP(x) (1.0f / (x))
#else
#define KBJ_RCP(x) __frcp_rn(x)
#endif

#if defined(KBJ_ARROW_LDS)
// LDS formulation (host emulation / A-B builds): one phase per pivot over the block entries spread across the lanes
KBJ_DEV void arrow_solve(KbjShared& S, const float* rhs, bool hess) {
  PFOR(w, 4 * 77) {
    int c = w / 77, e = w % 77, i = TRI.i[e], j = TRI.j[e];
    float v;
    if (i == 11) v = j < 5 ? rhs[10 + 5 * c - j] : 0.0f;
    else {
      int di = i < 5 ? 10 + 5 * c - i : i - 5, dj = j < 5 ? 10 + 5 * c - j : j - 5;
      v = (i >= 5 && j >= 5) ? 0.0f : M_get(S, di, dj);
      if (hess) {
        if (c < 2) {  // legs: contact rows of this leg
          int ci_ = i < 5 ? 10 - i : i - 5, cj_ = j < 5 ? 10 - j : j - 5;
          for (int r = 16 * c; r < 16 * c + 16; ++r)
            if (S.quad[ROW_CON + r]) v += S.D[ROW_CON + r] * S.Jc[r][ci_] * S.Jc[r][cj_];
        }
        if (i == j && i < 5) {
          int u = di - 6;
          if (S.quad[u]) v += S.D[u];
          if (S.quad[ROW_LIM + u]) v += S.D[ROW_LIM + u];
        }
      }
    }
    S.u.A[c][i][j] = v;
  }
  KBJ_SYNC();
  // column p keeps its unscaled entries (L_ip D_p), so each pivot is ONE phase
  for (int p = 0; p < 5; ++p) {
    const int np = tri_count(p);
    PFOR(w, 4 * np) {
      int c = w / np, e = w % np, i = TRI.i[e], j = TRI.j[e];
      S.u.A[c][i][j] -= S.u.A[c][i][p] * S.u.A[c][j][p] * KBJ_RCP(S.u.A[c][p][p]);
    }
    KBJ_SYNC();
  }
  PFOR(w, 7 * 6) {
    int i = w / 6, j = w % 6;
    if (i < 6 && j > i) continue;
    float v = i < 6 ? S.Mb[i][j] : rhs[j];
    int ai = i < 6 ? 5 + i : 11;
    for (int c = 0; c < 4; ++c) v += S.u.A[c][ai][5 + j];
    S.u.B[i][j] = v;
  }
  KBJ_SYNC();
  for (int p = 0; p < 6; ++p) {
    PFOR(w, 36) { int i = p + 1 + w / 6, j = p + 1 + w % 6; if (i <= 6 && j <= 5 && j <= i) S.u.B[i][j] -= S.u.B[i][p] * S.u.B[j][p] * KBJ_RCP(S.u.B[p][p]); }
    KBJ_SYNC();
  }
  PFOR(w, 1) {
    float x[6];
    for (int p = 5; p >= 0; --p) {
      float s = S.u.B[6][p];
      for (int i = p + 1; i < 6; ++i) s -= S.u.B[i][p] * x[i];
      x[p] = s * KBJ_RCP(S.u.B[p][p]);
      S.vec[p] = x[p];
    }
  }
  KBJ_SYNC();
  // chains: the base part of every back-substitution row is independent -> 20 lanes; only the 10 in-chain products stay serial
  PFOR(w, 4 * 5) {
    int c = w / 5, p = w % 5;
    float s = S.u.A[c][11][p];
    for (int i = 5; i < 11; ++i) s -= S.u.A[c][i][p] * S.vec[i - 5];
    S.u.A[c][11][p] = s;
  }
  KBJ_SYNC();
  PFOR(c, 4) {
    float x[5];
    for (int p = 4; p >= 0; --p) {
      float s = S.u.A[c][11][p];
      for (int i = p + 1; i < 5; ++i) s -= S.u.A[c][i][p] * x[i];
      x[p] = s * KBJ_RCP(S.u.A[c][p][p]);
      S.vec[10 + 5 * c - p] = x[p];
    }
  }
  KBJ_SYNC();
}
#endif

#if defined(KBJ_ARROW_LDS)
// y = M v using the tree sparsity
KBJ_DEV float mul_M_row(const KbjShared& S, int i, const float* v) {
  float s = 0;
  if (i < 6) {
    for (int j = 0; j < 6; ++j) s += M_get(S, i, j) * v[j];
    for (int j = 6; j < NV; ++j) s += S.Mc[(j - 6) / 5][(j - 6) % 5][i] * v[j];
  } else {
    int c = (i - 6) / 5, a = (i - 6) % 5;
    for (int j = 0; j < 6; ++j) s += S.Mc[c][a][j] * v[j];
    for (int b = 0; b < 5; ++b) s += (b <= a ? S.Mc[c][a][6 + b] : S.Mc[c][b][6 + a]) * v[6 + 5 * c + b];
  }
  return s;
}
#endif

// ---- constraint rows -------------------------------------------------------------------------------------------
KBJ_DEV float impedance(float dist, const ImpConst& ic) {
  float x = fabsf(dist) * ic.iwidth;
  if (x >= 1) return ic.dmax;
  if (x <= 0) return ic.dmin;
  float y;
  if (ic.power == 1.0f) y = x;
  else if (ic.power == 2.0f) y = x <= ic.mid ? x * x * ic.imid : 1 - (1 - x) * (1 - x) * ic.i1mid;   // MuJoCo's default power, without pow() or a division
  else if (x <= ic.mid) y = powf(x, ic.power) / powf(ic.mid, ic.power - 1);
  else y = 1 - powf(1 - x, ic.power) / powf(1 - ic.mid, ic.power - 1);
  return ic.dmin + y * (ic.dmax - ic.dmin);
}

KBJ_DEV void phys_make_constraints(KbjShared& S, const KbjModelLds& m, const PhysConst& pc) {
  const float* qpos = S.es + KBJ_ES_QPOS;
  const float* qvel = S.es + KBJ_ES_QVEL;
  PFOR(u, NU) {
    int dof = 6 + u;
    const float b = pc.fric.b, imp = pc.fric.dmin;   // impedance at distance 0
    float fl = S.ep[KBJ_EP_FRICLOSS + dof];
    float Rr = fmaxf(1e-15f, pc.fric_ratio * m.dof_invweight0[dof]);
    S.Rf[u] = Rr; S.D[u] = fl > 0 ? kbj_frcp(Rr) : 0.0f; S.aref[u] = -b * qvel[dof]; S.floss[u] = fl;
    float q = qpos[7 + u];
    float dlo = q - m.dof_range[dof][0], dhi = m.dof_range[dof][1] - q;
    float pos = fminf(dlo, dhi), sgn = dlo < dhi ? 1.0f : -1.0f;
    int r = ROW_LIM + u;
    S.lsign[u] = sgn;
    if (pos < 0) {
      const float impl = impedance(pos, pc.lim);
      float Rl = fmaxf(1e-15f, (1 - impl) * kbj_frcp(impl) * m.dof_invweight0[dof]);
      S.D[r] = kbj_frcp(Rl); S.aref[r] = -pc.lim.b * sgn * qvel[dof] - pc.lim.k * impl * pos;
    } else { S.D[r] = 0; S.aref[r] = 0; }
  }
  // point Jacobian of every active contact in its contact frame, once per (contact, column) - the four pyramid rows of a contact
  // are combinations of the same three components (normal, two tangents)
  PFOR(w, NCON * 11) {
    const int ci = w / 11, k = w % 11, leg = ci / 4;
    if (!S.conact[ci]) continue;
    const int dk = k < 6 ? k : 6 + 5 * leg + (k - 6);
    const float off[3] = {S.conpos[ci][0] - S.com[0], S.conpos[ci][1] - S.com[1], S.conpos[ci][2] - S.com[2]};
    float t[3];
    cross3(S.cdof[dk], off, t);
    const float jp[3] = {S.cdof[dk][3] + t[0], S.cdof[dk][4] + t[1], S.cdof[dk][5] + t[2]};
    float jn = jp[2], j1 = jp[0], j2 = jp[1];   // plane z = 0: normal = world z, tangents = world x, y
    if (pc.tamp != 0) {  // contact frame: n = surface normal, t1 = world x made orthogonal to n, t2 = n x t1
      const float nr[3] = {S.connrm[ci][0], S.connrm[ci][1], S.connrm[ci][2]};
      const float inv = 1 / sqrtf(1 - nr[0] * nr[0]);
      const float t1[3] = {(1 - nr[0] * nr[0]) * inv, -nr[0] * nr[1] * inv, -nr[0] * nr[2] * inv};
      const float t2[3] = {nr[1] * t1[2] - nr[2] * t1[1], nr[2] * t1[0] - nr[0] * t1[2], nr[0] * t1[1] - nr[1] * t1[0]};
      jn = nr[0] * jp[0] + nr[1] * jp[1] + nr[2] * jp[2];
      j1 = t1[0] * jp[0] + t1[1] * jp[1] + t1[2] * jp[2];
      j2 = t2[0] * jp[0] + t2[1] * jp[1] + t2[2] * jp[2];
    }
    S.u.jp[ci][k][0] = jn; S.u.jp[ci][k][1] = j1; S.u.jp[ci][k][2] = j2;
  }
  KBJ_SYNC();
  PFOR(r, 32) {
    const int ci = r / 4, e = r % 4, leg = ci / 4, row = ROW_CON + r;
    if (!S.conact[ci]) {
      S.D[row] = 0; S.aref[row] = 0;
      for (int k = 0; k < 11; ++k) S.Jc[r][k] = 0;
      continue;
    }
    const float mu = S.ep[KBJ_EP_MU];
    const int ax = e / 2;
    const float sg = (e & 1) ? -mu : mu;
    float vel = 0;
    for (int k = 0; k < 11; ++k) {
      const int dk = k < 6 ? k : 6 + 5 * leg + (k - 6);
      const float j = S.u.jp[ci][k][0] + sg * S.u.jp[ci][k][1 + ax];
      S.Jc[r][k] = j;
      vel += j * qvel[dk];
    }
    const float k_ = pc.con.k, b_ = pc.con.b, imp = impedance(S.condist[ci], pc.con);
    float tran = m.body_invweight0[leg ? 12 : 7][0];
    float invw = (tran + mu * mu * tran) * 2 * mu * mu;
    float Rc = fmaxf(1e-15f, (1 - imp) * kbj_frcp(imp) * invw);
    S.D[row] = kbj_frcp(Rc); S.aref[row] = -b_ * vel - k_ * imp * S.condist[ci];
  }
  KBJ_SYNC();
}

#if defined(KBJ_ARROW_LDS)   // ---- LDS formulation of the solver (A/B builds on the GPU and in the host emulation) ----
// J q for row r (r active)
KBJ_DEV float row_dot(const KbjShared& S, int r, const float* q) {
  if (r < ROW_LIM) return q[6 + r];
  if (r < ROW_CON) return S.lsign[r - ROW_LIM] * q[6 + r - ROW_LIM];
  int rr = r - ROW_CON, leg = rr / 16;
  float x = 0;
  for (int k = 0; k < 6; ++k) x += S.Jc[rr][k] * q[k];
  for (int k = 6; k < 11; ++k) x += S.Jc[rr][k] * q[6 + 5 * leg + (k - 6)];
  return x;
}

// residual of every row for a candidate acceleration q: jar = J q - aref
KBJ_DEV void rows_residual(KbjShared& S, const float* q) {
  PFOR(i, NV) S.Ma[i] = mul_M_row(S, i, q);
  PFOR(r, NROW) S.jar[r] = S.D[r] != 0 ? row_dot(S, r, q) - S.aref[r] : 0.0f;
  KBJ_SYNC();
}

// per-lane cost / derivative contributions: lane l < 20 owns friction+limit rows of joint l, lanes 32..63 one contact row
KBJ_DEV float row_cost(const KbjShared& S, int r, float x) {
  float D = S.D[r];
  if (D == 0) return 0.0f;
  if (r < ROW_LIM) {
    float f = S.floss[r], Rr = S.Rf[r];
    if (x <= -Rr * f) return f * (-0.5f * Rr * f - x);
    if (x >= Rr * f) return f * (-0.5f * Rr * f + x);
    return 0.5f * D * x * x;
  }
  return x < 0 ? 0.5f * D * x * x : 0.0f;
}
KBJ_DEV void row_deriv(const KbjShared& S, int r, float a, float& d1, float& d2) {
  float D = S.D[r];
  if (D == 0) return;
  float jv = S.jv[r], x = S.jar[r] + a * jv;
  if (r < ROW_LIM) {
    float f = S.floss[r], Rr = S.Rf[r];
    if (x <= -Rr * f) d1 -= f * jv;
    else if (x >= Rr * f) d1 += f * jv;
    else { d1 += D * x * jv; d2 += D * jv * jv; }
  } else if (x < 0) { d1 += D * x * jv; d2 += D * jv * jv; }
}
KBJ_DEV float total_cost(const KbjShared& S, const float* q) {
  return wsum(64, [&](int l) {
    float c = 0;
    if (l < NV) c += 0.5f * (S.Ma[l] - S.qfrc_smooth[l]) * (q[l] - S.qacc_smooth[l]);
    if (l < NU) c += row_cost(S, l, S.jar[l]) + row_cost(S, ROW_LIM + l, S.jar[ROW_LIM + l]);
    if (l >= 32) c += row_cost(S, ROW_CON + l - 32, S.jar[ROW_CON + l - 32]);
    return c;
  });
}

KBJ_DEV void rows_force(KbjShared& S) {
  PFOR(r, NROW) {
    float f = 0, D = S.D[r];
    int quad = 0;
    if (D != 0) {
      float x = S.jar[r];
      if (r < ROW_LIM) {
        float fl = S.floss[r], Rr = S.Rf[r];
        if (x <= -Rr * fl) f = fl;
        else if (x >= Rr * fl) f = -fl;
        else { f = -D * x; quad = 1; }
      } else if (x < 0) { f = -D * x; quad = 1; }
    }
    S.force[r] = f; S.quad[r] = quad;
  }
  KBJ_SYNC();
}

// Newton iterations on the convex constraint cost with an exact (safeguarded Newton) line search
KBJ_DEV void phys_solve(KbjShared& S, const KbjModelLds& m, const PhysConst& pc) {
  float* warm = S.es + KBJ_ES_WARM;
  arrow_solve(S, S.qfrc_smooth, false);
  PFOR(i, NV) S.qacc_smooth[i] = S.vec[i];
  KBJ_SYNC();
  KBJ_STAMP(7);
  // warm-start selection: the cheaper of the previous step's acceleration and the unconstrained one
  rows_residual(S, S.qacc_smooth);
  float cs = total_cost(S, S.qacc_smooth);
  KBJ_SYNC();
  rows_residual(S, warm);
  float cw = total_cost(S, warm);
  KBJ_SYNC();
  bool use_warm = cw < cs;
  PFOR(i, NV) S.qacc[i] = use_warm ? warm[i] : S.qacc_smooth[i];
  KBJ_SYNC();
  if (!use_warm) rows_residual(S, S.qacc);
  KBJ_STAMP(8);
  float scale = 1.0f / (m.meaninertia * NV);
  int iters = 0;
  for (int it = 0; it < pc.iterations; ++it) {
    rows_force(S);
    PFOR(i, NV) {
      float g = S.Ma[i] - S.qfrc_smooth[i];
      if (i < 6) { for (int r = 0; r < 32; ++r) g -= S.Jc[r][i] * S.force[ROW_CON + r]; }
      else {
        int u = i - 6;
        g -= S.force[u] + S.lsign[u] * S.force[ROW_LIM + u];
        if (i < 16) { int leg = u / 5, col = 6 + u % 5; for (int r = 16 * leg; r < 16 * leg + 16; ++r) g -= S.Jc[r][col] * S.force[ROW_CON + r]; }
      }
      S.grad[i] = g;
      S.mv[i] = -g;  // right-hand side of the Newton system
    }
    KBJ_SYNC();
    float gg = wsum(NV, [&](int l) { return S.grad[l] * S.grad[l]; });
    KBJ_STAMP(9);
    if (scale * sqrtf(gg) < pc.tolerance) break;
    arrow_solve(S, S.mv, true);
    KBJ_STAMP(10);
    PFOR(i, NV) S.search[i] = S.vec[i];
    KBJ_SYNC();
    PFOR(i, NV) S.mv[i] = mul_M_row(S, i, S.search);
    PFOR(r, NROW) S.jv[r] = S.D[r] != 0 ? row_dot(S, r, S.search) : 0.0f;
    KBJ_SYNC();
    KBJ_STAMP(11);
    float g1, g2;
    wsum2(NV, [&](int l, float& a, float& b) { a = S.search[l] * (S.Ma[l] - S.qfrc_smooth[l]); b = S.search[l] * S.mv[l]; }, g1, g2);
    auto eval = [&](float a, float& d1, float& d2) {
      wsum2(64, [&](int l, float& x1, float& x2) {
        if (l < NU) { row_deriv(S, l, a, x1, x2); row_deriv(S, ROW_LIM + l, a, x1, x2); }
        if (l >= 32) row_deriv(S, ROW_CON + l - 32, a, x1, x2);
      }, d1, d2);
      d1 += g1 + a * g2; d2 += g2;
    };
    float d1, d2, alpha = 0;
    eval(0.0f, d1, d2);
    if (d1 < 0 && d2 > 0) {
      float lo = 0, hi = 0;
      bool hi_valid = false;
      float a = -d1 / d2;
      const float d1_stop = 0.01f * fabsf(d1);  // MuJoCo's default ls_tolerance: relative slope reduction
      for (int ls = 0; ls < pc.ls_iterations; ++ls) {
        eval(a, d1, d2);
        if (fabsf(d1) <= d1_stop) break;
        if (d1 < 0) lo = a; else { hi = a; hi_valid = true; }
        float an = a - d1 / d2;
        if (an <= lo || (hi_valid && an >= hi)) an = hi_valid ? 0.5f * (lo + hi) : 2 * a;
        a = an;
      }
      alpha = a;
    }
    KBJ_STAMP(12);
    PFOR(i, NV) { S.qacc[i] += alpha * S.search[i]; S.Ma[i] += alpha * S.mv[i]; }
    PFOR(r, NROW) S.jar[r] += alpha * S.jv[r];
    KBJ_SYNC();
    KBJ_STAMP(13);
    iters = it + 1;
    if (alpha == 0) break;
  }
  rows_force(S);
  S.iters = iters;
}
#else
// ---- register-resident Newton solver (the product kernel; the host emulation runs the same source lane by lane, kbj_wave.h) ----
//
// SOLVER LAYOUT of the wavefront: lane 16 c + r of DPP row c (= limb c: two legs, two arms).
//   r = 0..4   chain dof 10 + 5 c - r (ankle .. hip of limb c), together with the friction-loss and limit rows of that joint (unit rows)
//   r = 5..10  base dof r - 5, replicated in the four rows; its base-base part of M lives in row 0 only (the rows are summed)
//   r = 11     idle (a zero row)
//   r = 12..15 the right-hand side of an arrow solve (DPP bank 3, so ONE bank-masked `row_newbcast` move per column writes it)
//   lanes 0..31 additionally own pyramid row `lane` (row 16 c + k of leg c sits in DPP row c, where the leg's dofs are)
// A lane keeps: its row of M against (its chain's dofs ankle..hip | the six base dofs) m[11]; the same row of the contact part of the
// Hessian h[11] (kept incrementally; bank 3 carries the right-hand side); its pyramid row jc[11] and - transposed - column `col(r)` of its
// leg's sixteen pyramid rows jt[16]; the scalars of its dof and rows. Matrix-vector products are `v_fmac_f32_dpp` chains: M v and J v
// broadcast v along the DPP row, J^T f broadcasts f (16 instructions instead of eleven 16-lane reductions). Instruction counts matter
// here: the kernel is bound by vector issue.

// byte offsets (from S.Mb) of the eleven entries of a lane's row of M; entries a lane does not have point at S.zrow
struct MRowTab { unsigned short off[11][64]; };
constexpr MRowTab make_mrow_tab() {
  MRowTab t{};
  for (int lane = 0; lane < 64; ++lane) {
    const int c = lane >> 4, r = lane & 15;
    for (int j = 0; j < 11; ++j) {
      int idx = 36 + 220;   // zrow[0], in floats from Mb[0][0]
      if (r < 5) idx = j < 5 ? 36 + (c * 5 + (4 - (r < j ? r : j))) * 11 + (10 - (r > j ? r : j)) : 36 + (c * 5 + (4 - r)) * 11 + (j - 5);
      else if (r <= 10) {
        const int br = r - 5, bj = j - 5;
        if (j < 5) idx = 36 + (c * 5 + (4 - j)) * 11 + br;
        else if (c == 0) idx = (br > bj ? br : bj) * 6 + (br < bj ? br : bj);
      }
      t.off[j][lane] = (unsigned short)(4 * idx);
    }
  }
  return t;
}
#ifdef KBJ_EMU
static const MRowTab MROW = make_mrow_tab();
#else
__device__ const MRowTab MROW = make_mrow_tab();
#endif
static_assert(offsetof(KbjShared, Mc) - offsetof(KbjShared, Mb) == 36 * 4 && offsetof(KbjShared, zrow) - offsetof(KbjShared, Mb) == (36 + 220) * 4,
              "MRowTab addresses Mb, Mc and zrow as one array");

KBJ_DEV float kbj_fdiv(float a, float b) {   // a / b: hardware reciprocal + one Newton step on the quotient (wave-uniform scalars of the line search)
#ifdef KBJ_EMU
  return a / b;
#else
  const float r = __builtin_amdgcn_rcpf(b), q = a * r;
  return fmaf(fmaf(-q, b, a), r, q);
#endif
}

// Solves G x = g, G = M + (contact part of the Hessian in h) + (dnow on the diagonal of the chain dofs), in the solver layout: g comes
// in and x goes out as one value per lane (lanes r <= 10), nothing touches memory. Arrow-matrix LDL^T: every DPP row eliminates its
// chain's five dofs from its augmented block (rows = lanes), the four Schur complements are summed over the rows with lane swaps, every
// row factors the 6 x 6 base block redundantly. A pivot column is stored as the NEGATED multipliers -L_ip, so both back-substitutions
// are plain row sums of (multiplier x known unknowns) with the right-hand-side lane holding -1.
KBJ_DEV WF arrow_solve_w(const WF (&m)[11], const WF (&h)[11], const WF (&oh)[5], const WF& dnow, const WF& g) {
  WF a[11];
  WLANES(l) {
#pragma unroll
    for (int j = 0; j < 5; ++j) WL(a[j], l) = fmaf(WL(oh[j], l), WL(dnow, l), WL(m[j], l) + WL(h[j], l));
#pragma unroll
    for (int j = 5; j < 11; ++j) WL(a[j], l) = WL(m[j], l) + WL(h[j], l);
  }
  // right-hand side into bank 3 (m, h and oh are zero there); its base part in DPP row 0 only: the four rows are summed
  static_for<0, 5>([&](auto J_) { constexpr int j = decltype(J_)::value; wset_rhs<j, 0xF>(a[j], g); });
  static_for<5, 11>([&](auto J_) { constexpr int j = decltype(J_)::value; wset_rhs<j, 0x1>(a[j], g); });
  static_for<0, 5>([&](auto P_) {
    constexpr int p = decltype(P_)::value;
    const WF lp = wneg_div_bcast<p>(a[p]);
    static_for<p + 1, 11>([&](auto J_) { constexpr int j = decltype(J_)::value; wfmac_bcast<j>(a[j], a[p], lp); });
    a[p] = lp;
  });
  wrows_sum2(a[5], a[6]); wrows_sum2(a[7], a[8]); wrows_sum2(a[9], a[10]);
  static_for<0, 6>([&](auto Q_) {
    constexpr int q = decltype(Q_)::value;
    const WF lp = wneg_div_bcast<5 + q>(a[5 + q]);
    static_for<q + 1, 6>([&](auto J_) { constexpr int j = decltype(J_)::value; wfmac_bcast<5 + j>(a[5 + j], a[5 + q], lp); });
    a[5 + q] = lp;
  });
  WF xm;
  WLANES(l) WL(xm, l) = (l & 15) == 12 ? -1.0f : 0.0f;
  static_for<0, 11>([&](auto K_) {
    constexpr int p = 10 - decltype(K_)::value;   // lane (= block row) whose unknown this step produces: base dofs 10..5, then hip..ankle 4..0
    WF prod;
    WLANES(l) WL(prod, l) = WL(a[p], l) * WL(xm, l);
    wopaque(prod);
    xm = wsel<wmask_r(p)>(wrow_sum16(prod), xm);
  });
  return xm;
}

KBJ_DEV void phys_solve(KbjShared& S, const KbjModelLds& mdl, const PhysConst& pc) {
  constexpr unsigned long long CHAIN = wmask_r_below(5), BASE = wmask_r_below(11) & ~CHAIN, OWN = CHAIN | (BASE & 0xFFFFull);
  WF m[11], h[11], jc[11], jt[16], oh[5];
  WF qs, warm, Df, fl, thr, aref_f, actf, Dl, aref_l, lsa, Dc, aref_c;
  WLANES(l) {
    const int c = l >> 4, r = l & 15;
    const bool is_chain = r < 5;
    const int d = is_chain ? 10 + 5 * c - r : (r <= 10 ? r - 5 : 0), u = is_chain ? d - 6 : 0;
    const char* mb = reinterpret_cast<const char*>(&S.Mb[0][0]);
#pragma unroll
    for (int j = 0; j < 11; ++j) { WL(m[j], l) = *reinterpret_cast<const float*>(mb + MROW.off[j][l]); WL(h[j], l) = 0.0f; }
#pragma unroll
    for (int j = 0; j < 5; ++j) WL(oh[j], l) = r == j ? 1.0f : 0.0f;
    WL(qs, l) = S.qfrc_smooth[d]; WL(warm, l) = S.es[KBJ_ES_WARM + d];
    // friction-loss and limit row of this lane's joint; rows that do not exist (or are inactive: D = 0) carry zeros throughout
    const float D_f = is_chain ? S.D[u] : 0.0f, D_l = is_chain ? S.D[ROW_LIM + u] : 0.0f, f_l = is_chain ? S.floss[u] : 0.0f;
    WL(Df, l) = D_f; WL(fl, l) = f_l; WL(thr, l) = is_chain ? S.Rf[u] * f_l : 0.0f;
    WL(actf, l) = D_f != 0 ? 1.0f : 0.0f; WL(aref_f, l) = D_f != 0 ? S.aref[u] : 0.0f;
    WL(Dl, l) = D_l; WL(lsa, l) = D_l != 0 ? S.lsign[u] : 0.0f; WL(aref_l, l) = D_l != 0 ? S.aref[ROW_LIM + u] : 0.0f;
    // pyramid row `lane` (lanes 0..31; an inactive contact's row is all zeros, aref and D included) and, transposed, column col(r) of
    // the sixteen rows of leg c. The arms' rows (c = 2, 3) read leg c & 1: finite values that only ever meet a zero force.
    const float* jrow = l < 32 ? S.Jc[l] : S.zrow;
#pragma unroll
    for (int k = 0; k < 11; ++k) WL(jc[k], l) = jrow[k];
    WL(Dc, l) = l < 32 ? S.D[ROW_CON + (l & 31)] : 0.0f; WL(aref_c, l) = l < 32 ? S.aref[ROW_CON + (l & 31)] : 0.0f;
    const int colr = r < 5 ? 10 - r : (r <= 10 ? r - 5 : 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) WL(jt[k], l) = S.Jc[16 * (c & 1) + k][colr];
  }
  auto mul_M = [&](const WF& v) {       // (M v) of this lane's dof: v broadcast along the DPP row, the base dofs summed over the four limbs
    WF q = wmul_bcast<0>(v, m[0]);
    static_for<1, 11>([&](auto J_) { constexpr int j = decltype(J_)::value; wfmac_bcast<j>(q, v, m[j]); });
    return wsel<BASE>(wrows_sum1(q), q);
  };
  auto jdot = [&](const WF& v) {        // pyramid row . v  (Jc columns: 0..5 base dofs = lanes 5..10, 6..10 hip..ankle = lanes 4..0)
    WF x = wmul_bcast<5>(v, jc[0]);
    static_for<1, 6>([&](auto K_) { constexpr int k = decltype(K_)::value; wfmac_bcast<5 + k>(x, v, jc[k]); });
    static_for<0, 5>([&](auto A_) { constexpr int a = decltype(A_)::value; wfmac_bcast<4 - a>(x, v, jc[6 + a]); });
    return x;
  };
  // ---- unconstrained acceleration ----
  WF zero;
  WLANES(l) WL(zero, l) = 0.0f;
  const WF qas = arrow_solve_w(m, h, oh, zero, qs);
  KBJ_STAMP(7);
  // ---- warm start: the cheaper of the previous step's acceleration and the unconstrained one (both costs on one reduction tree) ----
  const WF Ma_s = mul_M(qas), Ma_w = mul_M(warm), jd_s = jdot(qas), jd_w = jdot(warm);
  WF jf_s, jl_s, jc_s, jf_w, jl_w, jc_w, cs_l, cw_l;
  WLANES(l) {
    auto cost_f = [&](float x) { const float ax = fabsf(x); return ax >= WL(thr, l) ? WL(fl, l) * (ax - 0.5f * WL(thr, l)) : 0.5f * WL(Df, l) * x * x; };
    auto cost_u = [&](float D, float x) { const float n = wmin0(x); return 0.5f * D * n * n; };
    WL(jf_s, l) = fmaf(WL(actf, l), WL(qas, l), -WL(aref_f, l)); WL(jl_s, l) = fmaf(WL(lsa, l), WL(qas, l), -WL(aref_l, l)); WL(jc_s, l) = WL(jd_s, l) - WL(aref_c, l);
    WL(jf_w, l) = fmaf(WL(actf, l), WL(warm, l), -WL(aref_f, l)); WL(jl_w, l) = fmaf(WL(lsa, l), WL(warm, l), -WL(aref_l, l)); WL(jc_w, l) = WL(jd_w, l) - WL(aref_c, l);
    WL(cs_l, l) = cost_f(WL(jf_s, l)) + cost_u(WL(Dl, l), WL(jl_s, l)) + cost_u(WL(Dc, l), WL(jc_s, l));          // the Gauss term vanishes at qacc_smooth
    WL(cw_l, l) = cost_f(WL(jf_w, l)) + cost_u(WL(Dl, l), WL(jl_w, l)) + cost_u(WL(Dc, l), WL(jc_w, l));
  }
  {
    WF gauss;
    WLANES(l) WL(gauss, l) = 0.5f * (WL(Ma_w, l) - WL(qs, l)) * (WL(warm, l) - WL(qas, l));
    gauss = wsel0<OWN>(gauss);            // every dof exactly once (the base dofs are replicated per row)
    WLANES(l) WL(cw_l, l) += WL(gauss, l);
  }
  float cs, cw;
  wsum2(cs_l, cw_l, cs, cw);
  const bool use_warm = cw < cs;
  WF qa, Ma, jar_f, jar_l, jar_c;
  WLANES(l) {
    WL(qa, l) = use_warm ? WL(warm, l) : WL(qas, l); WL(Ma, l) = use_warm ? WL(Ma_w, l) : WL(Ma_s, l);
    WL(jar_f, l) = use_warm ? WL(jf_w, l) : WL(jf_s, l); WL(jar_l, l) = use_warm ? WL(jl_w, l) : WL(jl_s, l); WL(jar_c, l) = use_warm ? WL(jc_w, l) : WL(jc_s, l);
  }
  KBJ_STAMP(8);
  const float tol2 = pc.tol2;
  int iters = 0;
  WF ff, flm, fc, dnow, w_prev;
  WLANES(l) WL(w_prev, l) = 0.0f;     // weight (D in the quadratic zone, else 0) with which this lane's pyramid row currently sits in h
  // forces of the rows at the current residuals; dnow = friction-loss + limit rows in their quadratic zone (unit rows: diagonal of H)
  auto rows_force = [&]() {
    WLANES(l) {
      const float t = wclamp(WL(jar_f, l), WL(thr, l));     // Huber: force = -D clamp(residual, +-R f)
      const bool qf = fabsf(WL(jar_f, l)) < WL(thr, l), ql = WL(jar_l, l) < 0.0f;
      WL(ff, l) = -WL(Df, l) * t;
      WL(flm, l) = -WL(Dl, l) * wmin0(WL(jar_l, l));
      WL(fc, l) = -WL(Dc, l) * wmin0(WL(jar_c, l));
      WL(dnow, l) = (qf ? WL(Df, l) : 0.0f) + (ql ? WL(Dl, l) : 0.0f);
    }
  };
  for (int it = 0; it < pc.iterations; ++it) {
    rows_force();
    // contact part of the Hessian, incrementally: only rows whose quadratic-zone flag flipped since the last solve change h (all active
    // rows on the first iteration). The weight changes go through LDS (S.force is free until the solve ends), the changed rows are a
    // ballot mask, so an iteration without flips costs nothing here.
    WF dw;
    WLANES(l) {
      const float w_now = WL(jar_c, l) < 0.0f ? WL(Dc, l) : 0.0f;
      WL(dw, l) = w_now - WL(w_prev, l); WL(w_prev, l) = w_now;
      if (l < 32) S.force[ROW_CON + l] = WL(dw, l);
    }
    const unsigned long long chg = wballot([&](int l) { return l < 32 && WL(dw, l) != 0.0f; });
    // gradient: M qacc - qfrc_smooth - J^T force; J^T f broadcasts the leg's sixteen forces over the transposed rows
    WF gcon = wmul_bcast<0>(fc, jt[0]);
    static_for<1, 16>([&](auto K_) { constexpr int k = decltype(K_)::value; wfmac_bcast<k>(gcon, fc, jt[k]); });
    gcon = wsel<BASE>(wrows_sum1(gcon), gcon);     // base dofs: both legs
    WF gr, g2;
    WLANES(l) {
      WL(gr, l) = ((WL(Ma, l) - WL(qs, l)) - (WL(ff, l) + WL(lsa, l) * WL(flm, l))) - WL(gcon, l);
      WL(g2, l) = WL(gr, l) * WL(gr, l);
    }
    const float gg = wsum(wsel0<OWN>(g2));
    KBJ_SYNC();
    KBJ_STAMP(9);
    if (gg < tol2) break;
    {
      unsigned rows = (unsigned)(chg & 0xFFFFu) | (unsigned)((chg >> 16) & 0xFFFFu);   // row k of either leg changed
      while (rows) {
        const int k = __builtin_ctz(rows);
        rows &= rows - 1;
        WLANES(l) {
          const int c = l >> 4, r = l & 15;
          if (c < 2 && r <= 10) {
            const int row = 16 * c + k, col = r < 5 ? 10 - r : r - 5;     // column of Jc this lane's block row stands for
            const float* J = S.Jc[row];
            const float t = S.force[ROW_CON + row] * J[col];
#pragma unroll
            for (int j = 0; j < 5; ++j) WL(h[j], l) = fmaf(t, J[10 - j], WL(h[j], l));
#pragma unroll
            for (int j = 5; j < 11; ++j) WL(h[j], l) = fmaf(t, J[j - 5], WL(h[j], l));
          }
        }
      }
    }
    WF ngr;
    WLANES(l) WL(ngr, l) = -WL(gr, l);
    const WF se = arrow_solve_w(m, h, oh, dnow, ngr);
    KBJ_STAMP(10);
    const WF mv = mul_M(se), jv_c = jdot(se);
    KBJ_STAMP(11);
    // exact line search on the piecewise-quadratic cost along se: phi'(a) = g1 + a g2 + sum over rows of D jv x(a) [in the quadratic
    // zone; the Huber rows saturate at +-f jv], phi''(a) = g2 + sum of D jv^2 over the rows in their quadratic zone
    WF jv_f, jv_l, Dfjv, Dfjv2, Dljv, Dljv2, Dcjv, Dcjv2, x1, x2;
    WLANES(l) {
      WL(jv_f, l) = WL(actf, l) * WL(se, l); WL(jv_l, l) = WL(lsa, l) * WL(se, l);
      WL(Dfjv, l) = WL(Df, l) * WL(jv_f, l); WL(Dfjv2, l) = WL(Dfjv, l) * WL(jv_f, l);
      WL(Dljv, l) = WL(Dl, l) * WL(jv_l, l); WL(Dljv2, l) = WL(Dljv, l) * WL(jv_l, l);
      WL(Dcjv, l) = WL(Dc, l) * WL(jv_c, l); WL(Dcjv2, l) = WL(Dcjv, l) * WL(jv_c, l);
      WL(x1, l) = WL(se, l) * (WL(Ma, l) - WL(qs, l)); WL(x2, l) = WL(se, l) * WL(mv, l);
    }
    float g1, g2s;
    wsum2(wsel0<OWN>(x1), wsel0<OWN>(x2), g1, g2s);
    auto eval = [&](float a, float& d1, float& d2) {
      WF y1, y2;
      WLANES(l) {
        const float xf = fmaf(a, WL(jv_f, l), WL(jar_f, l)), xl = fmaf(a, WL(jv_l, l), WL(jar_l, l)), xc = fmaf(a, WL(jv_c, l), WL(jar_c, l));
        const float tf = wclamp(xf, WL(thr, l));
        float s1 = WL(Dfjv, l) * tf, s2 = fabsf(xf) < WL(thr, l) ? WL(Dfjv2, l) : 0.0f;
        s1 = fmaf(WL(Dljv, l), wmin0(xl), s1); s2 += xl < 0.0f ? WL(Dljv2, l) : 0.0f;
        s1 = fmaf(WL(Dcjv, l), wmin0(xc), s1); s2 += xc < 0.0f ? WL(Dcjv2, l) : 0.0f;
        WL(y1, l) = s1; WL(y2, l) = s2;
      }
      float r1, r2;
      wsum2(y1, y2, r1, r2);
      d1 = r1 + fmaf(a, g2s, g1); d2 = r2 + g2s;
    };
    float d1, d2, alpha = 0;
    eval(0.0f, d1, d2);
    if (d1 < 0 && d2 > 0) {
      float lo = 0, hi = 0;
      bool hi_valid = false;
      float a = -kbj_fdiv(d1, d2);
      const float d1_stop = 0.01f * fabsf(d1);  // MuJoCo's default ls_tolerance: relative slope reduction
      for (int lsi = 0; lsi < pc.ls_iterations; ++lsi) {
        eval(a, d1, d2);
        if (fabsf(d1) <= d1_stop) break;
        if (d1 < 0) lo = a; else { hi = a; hi_valid = true; }
        float an = a - kbj_fdiv(d1, d2);
        if (an <= lo || (hi_valid && an >= hi)) an = hi_valid ? 0.5f * (lo + hi) : 2 * a;
        a = an;
      }
      alpha = a;
    }
    KBJ_STAMP(12);
    WLANES(l) {
      WL(qa, l) = fmaf(alpha, WL(se, l), WL(qa, l)); WL(Ma, l) = fmaf(alpha, WL(mv, l), WL(Ma, l));
      WL(jar_f, l) = fmaf(alpha, WL(jv_f, l), WL(jar_f, l)); WL(jar_l, l) = fmaf(alpha, WL(jv_l, l), WL(jar_l, l)); WL(jar_c, l) = fmaf(alpha, WL(jv_c, l), WL(jar_c, l));
    }
    KBJ_STAMP(13);
    iters = it + 1;
    if (alpha == 0) break;
  }
  rows_force();
  WLANES(l) {
    const int c = l >> 4, r = l & 15;
    if (r < 5) S.qacc[10 + 5 * c - r] = WL(qa, l);
    else if (r <= 10 && c == 0) S.qacc[r - 5] = WL(qa, l);
    if (l < 32) S.force[ROW_CON + l] = WL(fc, l);
  }
  PFOR(w, 1) S.iters = iters;
  KBJ_SYNC();
}
#endif

KBJ_DEV void phys_sensors(KbjShared& S, const KbjModelLds& m) {
  PFOR(w, 3) {
    if (w == 0) {
      float iq[4], sm[9];
      quat_mul(S.xquat[23], m.imu_quat, iq);
      for (int k = 0; k < 4; ++k) S.imuquat[k] = iq[k];
      quat_to_mat(iq, sm);
      matT_vec(sm, S.cvel[23], S.gyro);
      float g[3] = {0, 0, -1};
      rotate_by_quat(g, iq, true, S.pg);
    } else {
      int foot = w - 1, body = foot ? 12 : 7;
      float tot = 0, mat[9];
      quat_to_mat(S.xquat[body], mat);
      for (int ci = 4 * foot; ci < 4 * foot + 4; ++ci) {
        if (!S.conact[ci]) continue;
        float rel[3] = {S.conpos[ci][0] - S.xpos[body][0], S.conpos[ci][1] - S.xpos[body][1], S.conpos[ci][2] - S.xpos[body][2]}, loc[3];
        matT_vec(mat, rel, loc);
        bool inside = true;
        for (int k = 0; k < 3; ++k) inside = inside && fabsf(loc[k] - m.site_pos[foot][k]) <= m.site_size[foot][k];
        if (!inside) continue;
        for (int e = 0; e < 4; ++e) tot += S.force[ROW_CON + 4 * ci + e];
      }
      S.touch[foot] = tot;
    }
  }
  KBJ_SYNC();
}

// full forward pass on the state in S.es with torques S.ctrl and (if S.pushing) wrench S.push
// `sensors`: gyro / projected gravity / touch only feed the observations, i.e. they are needed after the LAST substep of a control step
KBJ_DEV void phys_forward(KbjShared& S, const KbjModelLds& m, const PhysConst& pc, bool sensors = true) {
  KBJ_STAMP(0);
  phys_kinematics(S, m); KBJ_STAMP(1);
  phys_com(S, m); KBJ_STAMP(2);
  phys_crb_mass(S); KBJ_STAMP(3);
  phys_collide_vel(S, m, pc); KBJ_STAMP(4);
  phys_smooth_forces(S, m); KBJ_STAMP(5);
  phys_make_constraints(S, m, pc); KBJ_STAMP(6);
  phys_solve(S, m, pc); KBJ_STAMP(15);
  if (sensors) phys_sensors(S, m);
  KBJ_STAMP(16);
}

// semi-implicit Euler; also refreshes the warm start (kept in the state row)
KBJ_DEV void phys_integrate(KbjShared& S, const PhysConst& pc) {
  float* qpos = S.es + KBJ_ES_QPOS;
  float* qvel = S.es + KBJ_ES_QVEL;
  PFOR(i, NV) { qvel[i] += pc.dt * S.qacc[i]; S.es[KBJ_ES_WARM + i] = S.qacc[i]; }
  KBJ_SYNC();
  PFOR(i, NV) {
    if (i < 3) qpos[i] += pc.dt * qvel[i];
    else if (i == 3) {
      // base orientation: q <- normalise(q * [cos h, sin(h) w / |w|]), h = |w| dt / 2. In z = h^2: cos h = 1 - z/2 + ..., sin(h) / |w| =
      // (dt / 2) (1 - z/6 + ...): no square root, no sin / cos (truncation < 1e-10 for |w| dt < 1; beyond that the closed form).
      float w[3] = {qvel[3], qvel[4], qvel[5]};
      const float hd = 0.5f * pc.dt, z = hd * hd * (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      if (z > 0) {
        float co, s;
        if (z < 0.25f) {
          co = fmaf(fmaf(fmaf(fmaf(fmaf(-1.0f / 3628800, z, 1.0f / 40320), z, -1.0f / 720), z, 1.0f / 24), z, -0.5f), z, 1.0f);
          s = hd * fmaf(fmaf(fmaf(fmaf(1.0f / 362880, z, -1.0f / 5040), z, 1.0f / 120), z, -1.0f / 6), z, 1.0f);
        } else { const float nrm = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), ang = nrm * pc.dt; co = cosf(ang / 2); s = sinf(ang / 2) / nrm; }
        float dq[4] = {co, s * w[0], s * w[1], s * w[2]}, q[4];
        quat_mul(qpos + 3, dq, q);
        quat_norm_fast(q);
        for (int k = 0; k < 4; ++k) qpos[3 + k] = q[k];
      }
    } else if (i >= 6) qpos[1 + i] += pc.dt * qvel[i];
  }
  KBJ_SYNC();
}

}  // namespace kbj
