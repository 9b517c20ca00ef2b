// kbj_oracle_physics.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
//
// Scalar restatement of the rigid-body step the reference obtains from mujoco-mjx 3.3.5
// (requirements.lock:113-114; configured at train.py:1775-1778: dt 0.004, iterations 8, ls_iterations 8).
// mujoco-mjx is NOT vendored in /root/reference and cannot be installed here, so this file restates
// MuJoCo's published pipeline (kinematics -> com -> CRB -> factor -> bias forces -> actuation ->
// plane-capsule collision -> constraint rows (frictionloss, limits, pyramidal contacts) -> CG solver ->
// semi-implicit Euler) from its documentation. PARITY WITH THE JAX REFERENCE IS UNPINNED (no reference
// tests / golden vectors exist); this oracle is pinned by analytic known-answer tests
// (tests/test_oracle_physics.py) and by an independent numpy mass-matrix in the model compiler.
//
// Written for clarity: one env at a time, dense matrices, fp32 and fp64 instantiations.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include "../include/kbj_model.h"

namespace kbjo {

constexpr int NB = KBJ_NBODY, NQ = KBJ_NQ, NV = KBJ_NV, NU = KBJ_NU, NCAP = KBJ_NCAP, NCON = KBJ_NCON;
constexpr int NEFC = 20 + 20 + 4 * NCON;  // frictionloss | limits | pyramidal contacts
constexpr int ROW_FRIC = 0, ROW_LIM = 20, ROW_CON = 40;

template <class R> struct Vec3 { R v[3]; };

template <class R> inline void cross3(const R* a, const R* b, R* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
template <class R> inline R dot3(const R* a, const R* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class R> inline void quat_mul(const R* a, const R* b, R* o) {
  R w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  R x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  R y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  R z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
template <class R> inline void quat_norm(R* q) {
  R n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}
template <class R> inline void quat_to_mat(const R* q, R* m) {  // row-major 3x3
  R w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = 1 - 2 * (x * x + y * y);
}
template <class R> inline void mat_vec(const R* m, const R* v, R* o) {
  for (int i = 0; i < 3; ++i) o[i] = m[3 * i] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
}
template <class R> inline void matT_vec(const R* m, const R* v, R* o) {
  for (int i = 0; i < 3; ++i) o[i] = m[i] * v[0] + m[3 + i] * v[1] + m[6 + i] * v[2];
}

// spatial (6D: [angular; linear]) helpers, MuJoCo conventions
template <class R> inline void inert_mul(const R* I, const R* v, R* o) {  // I[10] = Ixx Iyy Izz Ixy Ixz Iyz mdx mdy mdz m
  o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  o[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  o[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  o[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
template <class R> inline void cross_motion(const R* vel, const R* v, R* o) {
  cross3(vel, v, o);
  R t1[3], t2[3];
  cross3(vel, v + 3, t1); cross3(vel + 3, v, t2);
  for (int k = 0; k < 3; ++k) o[3 + k] = t1[k] + t2[k];
}
template <class R> inline void cross_force(const R* vel, const R* f, R* o) {
  R t1[3], t2[3];
  cross3(vel, f, t1); cross3(vel + 3, f + 3, t2);
  for (int k = 0; k < 3; ++k) o[k] = t1[k] + t2[k];
  cross3(vel, f + 3, o + 3);
}

// per-env randomised parameters, typed view of the EP record
template <class R> struct EnvParams {
  R ipos[NB][3], mass[NB], inertia[NB][3], armature[NV], fricloss[NV];
  R cap_pos[NCAP][3], cap_half[NCAP], cap_rad[NCAP];
  R kp[NU], kd[NU], taulim[NU], actbias[NU], jpbias[NU], pgbias[3], pglag, mu;
  int latency;
  void load(const float* ep) {
    for (int b = 0; b < NB; ++b) {
      for (int k = 0; k < 3; ++k) { ipos[b][k] = ep[KBJ_EP_IPOS + 3 * b + k]; inertia[b][k] = ep[KBJ_EP_INERTIA + 3 * b + k]; }
      mass[b] = ep[KBJ_EP_MASS + b];
    }
    for (int d = 0; d < NV; ++d) { armature[d] = ep[KBJ_EP_ARMATURE + d]; fricloss[d] = ep[KBJ_EP_FRICLOSS + d]; }
    for (int c = 0; c < NCAP; ++c) {
      for (int k = 0; k < 3; ++k) cap_pos[c][k] = ep[KBJ_EP_CAP_POS + 3 * c + k];
      cap_half[c] = ep[KBJ_EP_CAP_HALF + c]; cap_rad[c] = ep[KBJ_EP_CAP_RAD + c];
    }
    for (int u = 0; u < NU; ++u) {
      kp[u] = ep[KBJ_EP_KP + u]; kd[u] = ep[KBJ_EP_KD + u]; taulim[u] = ep[KBJ_EP_TAULIM + u];
      actbias[u] = ep[KBJ_EP_ACTBIAS + u]; jpbias[u] = ep[KBJ_EP_JPBIAS + u];
    }
    for (int k = 0; k < 3; ++k) pgbias[k] = ep[KBJ_EP_PGBIAS + k];
    pglag = ep[KBJ_EP_PGLAG]; latency = (int)ep[KBJ_EP_LATENCY]; mu = ep[KBJ_EP_MU];
  }
};

// everything one forward pass derives (mjData analogue, only the fields this path reads)
template <class R> struct Derived {
  R xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3];
  R xaxis[NB][3];                       // world hinge axis of the body's joint
  R subtree_com[NB][3], subtree_mass[NB];
  R cinert[NB][10], cvel[NB][6], cdof[NV][6], cdof_dot[NV][6];
  R M[NV][NV], Lc[NV][NV];              // mass matrix and its Cholesky factor (lower)
  R qfrc_bias[NV], qfrc_actuator[NV], qfrc_applied[NV], qfrc_smooth[NV], qacc_smooth[NV], qacc[NV];
  R con_pos[NCON][3], con_dist[NCON];   // all 8 slots always filled (fixed-size contact array semantics)
  R con_n[NCON][3];                     // contact normal (terrain surface normal below the capsule end)
  int con_active[NCON];
  R efc_J[NEFC][NV], efc_D[NEFC], efc_R[NEFC], efc_aref[NEFC], efc_floss[NEFC], efc_force[NEFC];
  int efc_active[NEFC];
  R qfrc_constraint[NV];
  R gyro[3], imu_quat[4], touch[2];
  int solver_iters;
};

inline long long g_solver_hist[16] = {};   // diagnostics: histogram of solver iterations per forward pass (kbj_cpu_solver_hist)

struct SolverOpts { int iterations = 8, ls_iterations = 8; double tolerance = 1e-8; int newton = 1; };

template <class R> struct Physics {
  const kbj_model* m;
  EnvParams<R> p;
  R dt;
  SolverOpts opt;
  R terrain_amp = 0, terrain_kw = 0;   // z = amp sin(kw x) sin(kw y), kw = 2 pi / wavelength (kbj_config); amp = 0: plane

  // terrain height and unit normal at (x, y)
  void terrain(R x, R y, R& h, R n[3]) const {
    R sx = std::sin(terrain_kw * x), cx = std::cos(terrain_kw * x), sy = std::sin(terrain_kw * y), cy = std::cos(terrain_kw * y);
    h = terrain_amp * sx * sy;
    R hx = terrain_amp * terrain_kw * cx * sy, hy = terrain_amp * terrain_kw * sx * cy;
    R inv = 1 / std::sqrt(1 + hx * hx + hy * hy);
    n[0] = -hx * inv; n[1] = -hy * inv; n[2] = inv;
  }
  // tangents of the contact frame: t1 = world x made orthogonal to n, t2 = n x t1 (world x, y on the plane)
  static void contact_tangents(const R n[3], R t1[3], R t2[3]) {
    R inv = 1 / std::sqrt(1 - n[0] * n[0]);
    t1[0] = (1 - n[0] * n[0]) * inv; t1[1] = -n[0] * n[1] * inv; t1[2] = -n[0] * n[2] * inv;
    t2[0] = n[1] * t1[2] - n[2] * t1[1]; t2[1] = n[2] * t1[0] - n[0] * t1[2]; t2[2] = n[0] * t1[1] - n[1] * t1[0];
  }

  // ---- position-dependent stage -------------------------------------------------------------
  void kinematics(const R* qpos, Derived<R>& d) const {
    for (int k = 0; k < 3; ++k) d.xpos[0][k] = 0;
    d.xquat[0][0] = 1; d.xquat[0][1] = d.xquat[0][2] = d.xquat[0][3] = 0;
    quat_to_mat(d.xquat[0], d.xmat[0]);
    int qadr = 0;
    for (int b = 1; b < NB; ++b) {
      int par = m->body_parent[b];
      if (m->body_dofnum[b] == 6) {
        for (int k = 0; k < 3; ++k) d.xpos[b][k] = qpos[k];
        for (int k = 0; k < 4; ++k) d.xquat[b][k] = qpos[3 + k];
        quat_norm(d.xquat[b]);
        qadr = 7;
      } else {
        R bp[3] = {(R)m->body_pos[b][0], (R)m->body_pos[b][1], (R)m->body_pos[b][2]}, t[3];
        mat_vec(d.xmat[par], bp, t);
        for (int k = 0; k < 3; ++k) d.xpos[b][k] = d.xpos[par][k] + t[k];
        R bq[4] = {(R)m->body_quat[b][0], (R)m->body_quat[b][1], (R)m->body_quat[b][2], (R)m->body_quat[b][3]};
        R q[4];
        quat_mul(d.xquat[par], bq, q);
        if (m->body_dofnum[b] == 1) {
          R ang = qpos[qadr++];
          R s = std::sin(ang / 2), c = std::cos(ang / 2);
          R jq[4] = {c, s * (R)m->jnt_axis[b][0], s * (R)m->jnt_axis[b][1], s * (R)m->jnt_axis[b][2]}, q2[4];
          quat_mul(q, jq, q2);
          for (int k = 0; k < 4; ++k) q[k] = q2[k];
        }
        quat_norm(q);
        for (int k = 0; k < 4; ++k) d.xquat[b][k] = q[k];
      }
      quat_to_mat(d.xquat[b], d.xmat[b]);
      R t[3];
      mat_vec(d.xmat[b], p.ipos[b], t);
      for (int k = 0; k < 3; ++k) d.xipos[b][k] = d.xpos[b][k] + t[k];
      if (m->body_dofnum[b] == 1) {
        R ax[3] = {(R)m->jnt_axis[b][0], (R)m->jnt_axis[b][1], (R)m->jnt_axis[b][2]};
        mat_vec(d.xmat[b], ax, d.xaxis[b]);
      }
    }
  }

  void com_pos(Derived<R>& d) const {
    for (int b = 0; b < NB; ++b) {
      d.subtree_mass[b] = b ? p.mass[b] : 0;
      for (int k = 0; k < 3; ++k) d.subtree_com[b][k] = b ? p.mass[b] * d.xipos[b][k] : 0;
    }
    for (int b = NB - 1; b >= 1; --b) {
      int par = m->body_parent[b];
      d.subtree_mass[par] += d.subtree_mass[b];
      for (int k = 0; k < 3; ++k) d.subtree_com[par][k] += d.subtree_com[b][k];
    }
    for (int b = 0; b < NB; ++b)
      for (int k = 0; k < 3; ++k) d.subtree_com[b][k] = d.subtree_mass[b] > 0 ? d.subtree_com[b][k] / d.subtree_mass[b] : d.xipos[b][k];
    const R* root = d.subtree_com[1];  // all moving bodies hang off body 1
    for (int b = 1; b < NB; ++b) {
      const R* mat = d.xmat[b];
      const R* in = p.inertia[b];
      R dif[3] = {d.xipos[b][0] - root[0], d.xipos[b][1] - root[1], d.xipos[b][2] - root[2]};
      R ms = p.mass[b];
      R* c = d.cinert[b];
      // rotate the diagonal body inertia to world axes: mat * diag(in) * mat^T
      c[0] = mat[0] * mat[0] * in[0] + mat[1] * mat[1] * in[1] + mat[2] * mat[2] * in[2];
      c[1] = mat[3] * mat[3] * in[0] + mat[4] * mat[4] * in[1] + mat[5] * mat[5] * in[2];
      c[2] = mat[6] * mat[6] * in[0] + mat[7] * mat[7] * in[1] + mat[8] * mat[8] * in[2];
      c[3] = mat[0] * mat[3] * in[0] + mat[1] * mat[4] * in[1] + mat[2] * mat[5] * in[2];
      c[4] = mat[0] * mat[6] * in[0] + mat[1] * mat[7] * in[1] + mat[2] * mat[8] * in[2];
      c[5] = mat[3] * mat[6] * in[0] + mat[4] * mat[7] * in[1] + mat[5] * mat[8] * in[2];
      // parallel axis to the tree's centre of mass
      c[0] += ms * (dif[1] * dif[1] + dif[2] * dif[2]);
      c[1] += ms * (dif[0] * dif[0] + dif[2] * dif[2]);
      c[2] += ms * (dif[0] * dif[0] + dif[1] * dif[1]);
      c[3] -= ms * dif[0] * dif[1];
      c[4] -= ms * dif[0] * dif[2];
      c[5] -= ms * dif[1] * dif[2];
      c[6] = ms * dif[0]; c[7] = ms * dif[1]; c[8] = ms * dif[2]; c[9] = ms;
    }
    for (int k = 0; k < 10; ++k) d.cinert[0][k] = 0;
    // motion axes about the tree com
    for (int b = 1; b < NB; ++b) {
      int adr = m->body_dofadr[b];
      R off[3] = {root[0] - d.xpos[b][0], root[1] - d.xpos[b][1], root[2] - d.xpos[b][2]};  // joint anchor == body origin
      if (m->body_dofnum[b] == 6) {
        for (int i = 0; i < 3; ++i) {
          for (int k = 0; k < 6; ++k) d.cdof[adr + i][k] = 0;
          d.cdof[adr + i][3 + i] = 1;
          R ax[3] = {d.xmat[b][i], d.xmat[b][3 + i], d.xmat[b][6 + i]};  // body-local rotation axis i
          for (int k = 0; k < 3; ++k) d.cdof[adr + 3 + i][k] = ax[k];
          cross3(ax, off, d.cdof[adr + 3 + i] + 3);
        }
      } else if (m->body_dofnum[b] == 1) {
        for (int k = 0; k < 3; ++k) d.cdof[adr][k] = d.xaxis[b][k];
        cross3(d.xaxis[b], off, d.cdof[adr] + 3);
      }
    }
  }

  void crb(Derived<R>& d) const {
    R crbI[NB][10];
    for (int b = 0; b < NB; ++b) for (int k = 0; k < 10; ++k) crbI[b][k] = d.cinert[b][k];
    for (int b = NB - 1; b >= 1; --b) {
      int par = m->body_parent[b];
      if (par > 0) for (int k = 0; k < 10; ++k) crbI[par][k] += crbI[b][k];
    }
    for (int i = 0; i < NV; ++i) for (int j = 0; j < NV; ++j) d.M[i][j] = 0;
    for (int i = 0; i < NV; ++i) {
      R buf[6];
      inert_mul(crbI[m->dof_body[i]], d.cdof[i], buf);
      d.M[i][i] = p.armature[i];
      for (int j = i; j >= 0; j = m->dof_parent[j]) {
        R s = 0;
        for (int k = 0; k < 6; ++k) s += d.cdof[j][k] * buf[k];
        d.M[i][j] += s;
        d.M[j][i] = d.M[i][j];
      }
    }
  }

  void factor(Derived<R>& d) const {  // dense Cholesky M = Lc Lc^T
    for (int i = 0; i < NV; ++i) {
      for (int j = 0; j <= i; ++j) {
        R s = d.M[i][j];
        for (int k = 0; k < j; ++k) s -= d.Lc[i][k] * d.Lc[j][k];
        d.Lc[i][j] = (i == j) ? std::sqrt(s) : s / d.Lc[j][j];
      }
      for (int j = i + 1; j < NV; ++j) d.Lc[i][j] = 0;
    }
  }
  void solve_M(const Derived<R>& d, const R* b, R* x) const {
    R y[NV];
    for (int i = 0; i < NV; ++i) { R s = b[i]; for (int k = 0; k < i; ++k) s -= d.Lc[i][k] * y[k]; y[i] = s / d.Lc[i][i]; }
    for (int i = NV - 1; i >= 0; --i) { R s = y[i]; for (int k = i + 1; k < NV; ++k) s -= d.Lc[k][i] * x[k]; x[i] = s / d.Lc[i][i]; }
  }
  void mul_M(const Derived<R>& d, const R* v, R* o) const {
    for (int i = 0; i < NV; ++i) { R s = 0; for (int j = 0; j < NV; ++j) s += d.M[i][j] * v[j]; o[i] = s; }
  }

  // ---- velocity-dependent stage ---------------------------------------------------------------
  void com_vel(const R* qvel, Derived<R>& d) const {
    for (int k = 0; k < 6; ++k) d.cvel[0][k] = 0;
    for (int b = 1; b < NB; ++b) {
      int par = m->body_parent[b], adr = m->body_dofadr[b], n = m->body_dofnum[b];
      R v[6];
      for (int k = 0; k < 6; ++k) v[k] = d.cvel[par][k];
      if (n == 6) {
        for (int i = 0; i < 3; ++i) {
          for (int k = 0; k < 6; ++k) { d.cdof_dot[adr + i][k] = 0; v[k] += d.cdof[adr + i][k] * qvel[adr + i]; }
        }
        for (int i = 3; i < 6; ++i) cross_motion(v, d.cdof[adr + i], d.cdof_dot[adr + i]);
        for (int i = 3; i < 6; ++i) for (int k = 0; k < 6; ++k) v[k] += d.cdof[adr + i][k] * qvel[adr + i];
      } else if (n == 1) {
        cross_motion(v, d.cdof[adr], d.cdof_dot[adr]);
        for (int k = 0; k < 6; ++k) v[k] += d.cdof[adr][k] * qvel[adr];
      }
      for (int k = 0; k < 6; ++k) d.cvel[b][k] = v[k];
    }
  }

  void rne_bias(const R* qvel, Derived<R>& d) const {  // Coriolis/centrifugal + gravity
    R cacc[NB][6], cfrc[NB][6];
    for (int k = 0; k < 3; ++k) { cacc[0][k] = 0; cacc[0][3 + k] = -(R)m->gravity[k]; }
    for (int b = 1; b < NB; ++b) {
      int par = m->body_parent[b], adr = m->body_dofadr[b], n = m->body_dofnum[b];
      for (int k = 0; k < 6; ++k) cacc[b][k] = cacc[par][k];
      for (int i = 0; i < n; ++i) for (int k = 0; k < 6; ++k) cacc[b][k] += d.cdof_dot[adr + i][k] * qvel[adr + i];
      R Ia[6], Iv[6], vxIv[6];
      inert_mul(d.cinert[b], cacc[b], Ia);
      inert_mul(d.cinert[b], d.cvel[b], Iv);
      cross_force(d.cvel[b], Iv, vxIv);
      for (int k = 0; k < 6; ++k) cfrc[b][k] = Ia[k] + vxIv[k];
    }
    for (int k = 0; k < 6; ++k) cfrc[0][k] = 0;
    for (int b = NB - 1; b >= 1; --b) {
      int par = m->body_parent[b];
      for (int k = 0; k < 6; ++k) cfrc[par][k] += cfrc[b][k];
    }
    for (int i = 0; i < NV; ++i) {
      R s = 0;
      for (int k = 0; k < 6; ++k) s += d.cdof[i][k] * cfrc[m->dof_body[i]][k];
      d.qfrc_bias[i] = s;
    }
  }

  // translational Jacobian row set of a world point rigidly attached to `body`: out[3][NV]
  void jac_point(const Derived<R>& d, int body, const R* point, R out[3][NV]) const {
    for (int r = 0; r < 3; ++r) for (int i = 0; i < NV; ++i) out[r][i] = 0;
    R off[3] = {point[0] - d.subtree_com[1][0], point[1] - d.subtree_com[1][1], point[2] - d.subtree_com[1][2]};
    int b = body;
    while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
    if (b <= 0) return;
    for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parent[i]) {
      R t[3];
      cross3(d.cdof[i], off, t);  // angular x offset
      for (int r = 0; r < 3; ++r) out[r][i] = d.cdof[i][3 + r] + t[r];
    }
  }

  // ---- constraints ---------------------------------------------------------------------------
  static R impedance(R dist, const float* solimp) {
    R dmin = std::min<R>(std::max<R>(solimp[0], (R)0.0001), (R)0.9999), dmax = std::min<R>(std::max<R>(solimp[1], (R)0.0001), (R)0.9999);
    R width = std::max<R>(solimp[2], (R)1e-15), mid = std::min<R>(std::max<R>(solimp[3], (R)0.0001), (R)0.9999), power = std::max<R>(solimp[4], 1);
    R x = std::fabs(dist) / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    R y;
    if (power == 1) y = x;
    else if (x <= mid) y = std::pow(x, power) / std::pow(mid, power - 1);
    else y = 1 - std::pow(1 - x, power) / std::pow(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
  }
  void kbi(const float* solref, const float* solimp, R dist, R& k, R& b, R& imp) const {
    R dmax = std::min<R>(std::max<R>(solimp[1], (R)0.0001), (R)0.9999);
    R tc = std::max<R>(solref[0], 2 * dt), dr = solref[1];
    k = 1 / (dmax * dmax * tc * tc * dr * dr);
    b = 2 / (dmax * tc);
    imp = impedance(dist, solimp);
  }

  void collide(Derived<R>& d) const {  // capsule ends against the plane z = 0 (normal +z) or the sine terrain
    for (int c = 0; c < NCAP; ++c) {
      int b = m->cap_body[c];
      R ctr[3], ax[3], t[3];
      mat_vec(d.xmat[b], p.cap_pos[c], t);
      for (int k = 0; k < 3; ++k) ctr[k] = d.xpos[b][k] + t[k];
      R a0[3] = {(R)m->cap_axis[c][0], (R)m->cap_axis[c][1], (R)m->cap_axis[c][2]};
      mat_vec(d.xmat[b], a0, ax);
      for (int e = 0; e < 2; ++e) {
        R sgn = e ? 1 : -1;
        R end[3] = {ctr[0] + sgn * p.cap_half[c] * ax[0], ctr[1] + sgn * p.cap_half[c] * ax[1], ctr[2] + sgn * p.cap_half[c] * ax[2]};
        int ci = 2 * c + e;
        if (terrain_amp == 0) {
          R dist = end[2] - p.cap_rad[c];
          d.con_dist[ci] = dist;
          d.con_pos[ci][0] = end[0]; d.con_pos[ci][1] = end[1]; d.con_pos[ci][2] = end[2] - (p.cap_rad[c] + dist / 2);
          d.con_n[ci][0] = 0; d.con_n[ci][1] = 0; d.con_n[ci][2] = 1;
          d.con_active[ci] = dist < 0;
        } else {  // sphere (capsule end) against the tangent plane of the surface below its centre
          R h, n[3];
          terrain(end[0], end[1], h, n);
          R dist = (end[2] - h) * n[2] - p.cap_rad[c];
          d.con_dist[ci] = dist;
          for (int k = 0; k < 3; ++k) { d.con_pos[ci][k] = end[k] - n[k] * (p.cap_rad[c] + dist / 2); d.con_n[ci][k] = n[k]; }
          d.con_active[ci] = dist < 0;
        }
      }
    }
  }

  void make_constraints(const R* qpos, const R* qvel, Derived<R>& d) const {
    for (int r = 0; r < NEFC; ++r) {
      for (int i = 0; i < NV; ++i) d.efc_J[r][i] = 0;
      d.efc_D[r] = 0; d.efc_R[r] = 0; d.efc_aref[r] = 0; d.efc_floss[r] = 0; d.efc_active[r] = 0; d.efc_force[r] = 0;
    }
    // dof friction loss (Huber rows)
    for (int u = 0; u < NU; ++u) {
      int dof = 6 + u, r = ROW_FRIC + u;
      R k, b, imp;
      kbi(m->fric_solref, m->fric_solimp, 0, k, b, imp);
      d.efc_J[r][dof] = 1;
      d.efc_R[r] = std::max<R>((R)1e-15, (1 - imp) / imp * (R)m->dof_invweight0[dof]);
      d.efc_D[r] = 1 / d.efc_R[r];
      d.efc_aref[r] = -b * qvel[dof];
      d.efc_floss[r] = p.fricloss[dof];
      d.efc_active[r] = p.fricloss[dof] > 0;
    }
    // joint limits (one row per hinge, the nearer side)
    for (int u = 0; u < NU; ++u) {
      int dof = 6 + u, r = ROW_LIM + u;
      R q = qpos[7 + u];
      R dlo = q - (R)m->dof_range[dof][0], dhi = (R)m->dof_range[dof][1] - q;
      R pos = std::min(dlo, dhi), sgn = dlo < dhi ? 1 : -1;
      if (pos < 0) {
        R k, b, imp;
        kbi(m->limit_solref, m->limit_solimp, pos, k, b, imp);
        d.efc_J[r][dof] = sgn;
        d.efc_R[r] = std::max<R>((R)1e-15, (1 - imp) / imp * (R)m->dof_invweight0[dof]);
        d.efc_D[r] = 1 / d.efc_R[r];
        d.efc_aref[r] = -b * sgn * qvel[dof] - k * imp * pos;
        d.efc_active[r] = 1;
      }
    }
    // pyramidal frictional contacts: rows (n + mu t1, n - mu t1, n + mu t2, n - mu t2), t1 = +x, t2 = +y
    for (int ci = 0; ci < NCON; ++ci) {
      if (!d.con_active[ci]) continue;
      int body = m->cap_body[ci / 2];
      R Jp[3][NV];
      jac_point(d, body, d.con_pos[ci], Jp);
      R k, b, imp;
      kbi(m->contact_solref, m->contact_solimp, d.con_dist[ci], k, b, imp);
      R mu = p.mu;
      R tran = (R)m->body_invweight0[body][0];  // world side contributes 0
      R invw = (tran + mu * mu * tran) * 2 * mu * mu;  // common pyramid-edge weight (impratio = 1)
      R tg[2][3];
      if (terrain_amp != 0) contact_tangents(d.con_n[ci], tg[0], tg[1]);
      for (int e = 0; e < 4; ++e) {
        int r = ROW_CON + 4 * ci + e;
        int ax = e / 2;
        R s = (e & 1) ? -mu : mu;
        R vel = 0;
        for (int i = 0; i < NV; ++i) {
          if (terrain_amp == 0) d.efc_J[r][i] = Jp[2][i] + s * Jp[ax][i];
          else {
            R jn = d.con_n[ci][0] * Jp[0][i] + d.con_n[ci][1] * Jp[1][i] + d.con_n[ci][2] * Jp[2][i];
            R jt = tg[ax][0] * Jp[0][i] + tg[ax][1] * Jp[1][i] + tg[ax][2] * Jp[2][i];
            d.efc_J[r][i] = jn + s * jt;
          }
          vel += d.efc_J[r][i] * qvel[i];
        }
        d.efc_R[r] = std::max<R>((R)1e-15, (1 - imp) / imp * invw);
        d.efc_D[r] = 1 / d.efc_R[r];
        d.efc_aref[r] = -b * vel - k * imp * d.con_dist[ci];
        d.efc_active[r] = 1;
      }
    }
  }

  // constraint cost pieces for row r at residual x: force, and (optionally) cost
  inline void row_eval(const Derived<R>& d, int r, R x, R& force, R& cost) const {
    force = 0; cost = 0;
    if (!d.efc_active[r]) return;
    R D = d.efc_D[r];
    if (r < ROW_LIM) {  // Huber
      R f = d.efc_floss[r], Rr = d.efc_R[r];
      if (x <= -Rr * f) { force = f; cost = f * (-(R)0.5 * Rr * f - x); }
      else if (x >= Rr * f) { force = -f; cost = f * (-(R)0.5 * Rr * f + x); }
      else { force = -D * x; cost = (R)0.5 * D * x * x; }
    } else if (x < 0) { force = -D * x; cost = (R)0.5 * D * x * x; }
  }
  R total_cost(const Derived<R>& d, const R* qacc) const {
    R Ma[NV], c = 0;
    mul_M(d, qacc, Ma);
    for (int i = 0; i < NV; ++i) c += (R)0.5 * (Ma[i] - d.qfrc_smooth[i]) * (qacc[i] - d.qacc_smooth[i]);
    for (int r = 0; r < NEFC; ++r) {
      if (!d.efc_active[r]) continue;
      R x = -d.efc_aref[r];
      for (int i = 0; i < NV; ++i) x += d.efc_J[r][i] * qacc[i];
      R f, cr; row_eval(d, r, x, f, cr); c += cr;
    }
    return c;
  }

  // Polak-Ribiere CG with M^-1 preconditioner and a safeguarded-Newton exact line search
  // (fixed ls_iterations evaluations; spec in DESIGN.md "Solver")
  void solve(Derived<R>& d, const R* warm) const {
    R qacc[NV], Ma[NV], jar[NEFC], grad[NV], Mgrad[NV], grad_old[NV], Mgrad_old[NV], search[NV], mv[NV], jv[NEFC];
    R cw = total_cost(d, warm), cs = total_cost(d, d.qacc_smooth);
    for (int i = 0; i < NV; ++i) qacc[i] = cw < cs ? warm[i] : d.qacc_smooth[i];
    mul_M(d, qacc, Ma);
    for (int r = 0; r < NEFC; ++r) { R x = -d.efc_aref[r]; for (int i = 0; i < NV; ++i) x += d.efc_J[r][i] * qacc[i]; jar[r] = x; }
    R scale = 1 / ((R)m->meaninertia * NV);
    d.solver_iters = 0;
    for (int it = 0; it < opt.iterations; ++it) {
      for (int i = 0; i < NV; ++i) grad[i] = Ma[i] - d.qfrc_smooth[i];
      for (int r = 0; r < NEFC; ++r) {
        R f, c; row_eval(d, r, jar[r], f, c);
        if (f != 0) for (int i = 0; i < NV; ++i) grad[i] -= d.efc_J[r][i] * f;
      }
      solve_M(d, grad, Mgrad);
      R gg = 0; for (int i = 0; i < NV; ++i) gg += grad[i] * grad[i];
      if (scale * std::sqrt(gg) < (R)opt.tolerance) break;
      if (opt.newton) {
        // Newton direction: H = M + sum over rows in their quadratic zone of D J^T J (dense Cholesky)
        static thread_local R H[NV][NV], Lh[NV][NV];
        for (int i = 0; i < NV; ++i) for (int j = 0; j < NV; ++j) H[i][j] = d.M[i][j];
        for (int r = 0; r < NEFC; ++r) {
          if (!d.efc_active[r]) continue;
          bool quad = r < ROW_LIM ? std::fabs(jar[r]) < d.efc_R[r] * d.efc_floss[r] : jar[r] < 0;
          if (!quad) continue;
          for (int i = 0; i < NV; ++i) {
            if (d.efc_J[r][i] == 0) continue;
            R s = d.efc_D[r] * d.efc_J[r][i];
            for (int j = 0; j < NV; ++j) H[i][j] += s * d.efc_J[r][j];
          }
        }
        for (int i = 0; i < NV; ++i)
          for (int j = 0; j <= i; ++j) {
            R s = H[i][j];
            for (int k = 0; k < j; ++k) s -= Lh[i][k] * Lh[j][k];
            Lh[i][j] = (i == j) ? std::sqrt(s) : s / Lh[j][j];
          }
        R y[NV];
        for (int i = 0; i < NV; ++i) { R s = -grad[i]; for (int k = 0; k < i; ++k) s -= Lh[i][k] * y[k]; y[i] = s / Lh[i][i]; }
        for (int i = NV - 1; i >= 0; --i) { R s = y[i]; for (int k = i + 1; k < NV; ++k) s -= Lh[k][i] * search[k]; search[i] = s / Lh[i][i]; }
      } else if (it == 0) { for (int i = 0; i < NV; ++i) search[i] = -Mgrad[i]; }
      else {
        R num = 0, den = 0;
        for (int i = 0; i < NV; ++i) { num += grad[i] * (Mgrad[i] - Mgrad_old[i]); den += grad_old[i] * Mgrad_old[i]; }
        R beta = den > (R)1e-30 ? std::max<R>(0, num / den) : 0;
        for (int i = 0; i < NV; ++i) search[i] = -Mgrad[i] + beta * search[i];
      }
      for (int i = 0; i < NV; ++i) { grad_old[i] = grad[i]; Mgrad_old[i] = Mgrad[i]; }
      mul_M(d, search, mv);
      for (int r = 0; r < NEFC; ++r) { R s = 0; if (d.efc_active[r]) for (int i = 0; i < NV; ++i) s += d.efc_J[r][i] * search[i]; jv[r] = s; }
      R g1 = 0, g2 = 0;
      for (int i = 0; i < NV; ++i) { g1 += search[i] * (Ma[i] - d.qfrc_smooth[i]); g2 += search[i] * mv[i]; }
      auto eval = [&](R a, R& d1, R& d2) {
        d1 = g1 + a * g2; d2 = g2;
        for (int r = 0; r < NEFC; ++r) {
          if (!d.efc_active[r]) continue;
          R x = jar[r] + a * jv[r], D = d.efc_D[r];
          if (r < ROW_LIM) {
            R f = d.efc_floss[r], Rr = d.efc_R[r];
            if (x <= -Rr * f) d1 -= f * jv[r];
            else if (x >= Rr * f) d1 += f * jv[r];
            else { d1 += D * x * jv[r]; d2 += D * jv[r] * jv[r]; }
          } else if (x < 0) { d1 += D * x * jv[r]; d2 += D * jv[r] * jv[r]; }
        }
      };
      R d1, d2, alpha = 0;
      eval(0, d1, d2);
      if (d1 < 0 && d2 > 0) {
        R lo = 0, hi = 0; bool hi_valid = false;
        R a = -d1 / d2;
        const R d1_stop = (R)0.01 * std::fabs(d1);  // MuJoCo's default ls_tolerance: relative slope reduction
        for (int ls = 0; ls < opt.ls_iterations; ++ls) {
          eval(a, d1, d2);
          if (std::fabs(d1) <= d1_stop) break;
          if (d1 < 0) lo = a; else { hi = a; hi_valid = true; }
          R an = a - d1 / d2;
          if (an <= lo || (hi_valid && an >= hi)) an = hi_valid ? (R)0.5 * (lo + hi) : 2 * a;
          a = an;
        }
        alpha = a;
      }
      for (int i = 0; i < NV; ++i) { qacc[i] += alpha * search[i]; Ma[i] += alpha * mv[i]; }
      for (int r = 0; r < NEFC; ++r) jar[r] += alpha * jv[r];
      d.solver_iters = it + 1;
      if (alpha == 0) break;
    }
    for (int i = 0; i < NV; ++i) { d.qacc[i] = qacc[i]; d.qfrc_constraint[i] = 0; }
    for (int r = 0; r < NEFC; ++r) {
      R f, c; row_eval(d, r, jar[r], f, c);
      d.efc_force[r] = f;
      if (f != 0) for (int i = 0; i < NV; ++i) d.qfrc_constraint[i] += d.efc_J[r][i] * f;
    }
  }

  // ---- full forward pass: qpos, qvel, ctrl, external wrench on the base -> qacc + sensors ---------
  void forward(const R* qpos, const R* qvel, const R* ctrl, const R* push, const R* warm, Derived<R>& d) const {
    kinematics(qpos, d);
    com_pos(d);
    crb(d);
    factor(d);
    collide(d);
    com_vel(qvel, d);
    rne_bias(qvel, d);
    for (int i = 0; i < NV; ++i) { d.qfrc_actuator[i] = 0; d.qfrc_applied[i] = 0; }
    for (int u = 0; u < NU; ++u)
      d.qfrc_actuator[6 + u] = std::min<R>(std::max<R>(ctrl[u], (R)m->act_range[u][0]), (R)m->act_range[u][1]);
    if (push) {  // wrench at the base body's inertial frame origin (ipos), world frame
      int b = m->base_body;
      R arm[3] = {d.xipos[b][0] - d.xpos[b][0], d.xipos[b][1] - d.xpos[b][1], d.xipos[b][2] - d.xpos[b][2]}, t[3], tq[3], loc[3];
      cross3(arm, push, t);
      for (int k = 0; k < 3; ++k) { d.qfrc_applied[k] = push[k]; tq[k] = push[3 + k] + t[k]; }
      matT_vec(d.xmat[b], tq, loc);
      for (int k = 0; k < 3; ++k) d.qfrc_applied[3 + k] = loc[k];
    }
    for (int i = 0; i < NV; ++i) d.qfrc_smooth[i] = d.qfrc_actuator[i] + d.qfrc_applied[i] - d.qfrc_bias[i];
    solve_M(d, d.qfrc_smooth, d.qacc_smooth);
    make_constraints(qpos, qvel, d);
    solve(d, warm);
    __atomic_fetch_add(&g_solver_hist[d.solver_iters < 15 ? d.solver_iters : 15], 1LL, __ATOMIC_RELAXED);
    sensors(d);
  }

  void sensors(Derived<R>& d) const {
    int ib = m->imu_body;
    R iq[4] = {(R)m->imu_quat[0], (R)m->imu_quat[1], (R)m->imu_quat[2], (R)m->imu_quat[3]}, sm[9];
    quat_mul(d.xquat[ib], iq, d.imu_quat);
    quat_to_mat(d.imu_quat, sm);
    matT_vec(sm, d.cvel[ib], d.gyro);  // angular velocity is origin independent
    d.touch[0] = d.touch[1] = 0;
    for (int ci = 0; ci < NCON; ++ci) {
      if (!d.con_active[ci]) continue;
      int foot = ci / 4, body = m->cap_body[ci / 2];
      R rel[3] = {d.con_pos[ci][0] - d.xpos[body][0], d.con_pos[ci][1] - d.xpos[body][1], d.con_pos[ci][2] - d.xpos[body][2]}, loc[3];
      matT_vec(d.xmat[body], rel, loc);
      bool inside = true;
      for (int k = 0; k < 3; ++k) inside = inside && std::fabs(loc[k] - (R)m->site_pos[foot][k]) <= (R)m->site_size[foot][k];
      if (!inside) continue;
      R fn = 0;
      for (int e = 0; e < 4; ++e) fn += d.efc_force[ROW_CON + 4 * ci + e];
      d.touch[foot] += fn;
    }
  }

  // semi-implicit Euler (== MuJoCo "implicitfast" for this model: no joint damping, motors have no velocity term)
  void integrate(R* qpos, R* qvel, const Derived<R>& d) const {
    for (int i = 0; i < NV; ++i) qvel[i] += dt * d.qacc[i];
    for (int k = 0; k < 3; ++k) qpos[k] += dt * qvel[k];
    R w[3] = {qvel[3], qvel[4], qvel[5]};
    R ang = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) * dt;
    if (ang > 0) {
      R s = std::sin(ang / 2) / (ang / dt);
      R dq[4] = {std::cos(ang / 2), s * w[0], s * w[1], s * w[2]}, q[4];
      quat_mul(qpos + 3, dq, q);  // body-local angular velocity: right multiplication
      quat_norm(q);
      for (int k = 0; k < 4; ++k) qpos[3 + k] = q[k];
    }
    for (int u = 0; u < NU; ++u) qpos[7 + u] += dt * qvel[6 + u];
  }

  R kinetic_energy(const Derived<R>& d, const R* qvel) const {
    R Mv[NV], e = 0; mul_M(d, qvel, Mv);
    for (int i = 0; i < NV; ++i) e += (R)0.5 * qvel[i] * Mv[i];
    return e;
  }
  R potential_energy(const Derived<R>& d) const {
    R e = 0;
    for (int b = 1; b < NB; ++b) for (int k = 0; k < 3; ++k) e -= p.mass[b] * (R)m->gravity[k] * d.xipos[b][k];
    return e;
  }
};

}  // namespace kbjo
